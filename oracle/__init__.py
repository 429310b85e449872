"""CPU oracle for the lightmotif scoring hot path -- TEST INFRASTRUCTURE ONLY.

Two independent restatements of the reference's *Generic* pipeline
(``lightmotif/src/pli/mod.rs:72-221``):

* :mod:`oracle.c_oracle` -- ctypes front-end of ``lm_oracle.c`` (plain C), plus
  the AVX2 port ``lm_avx2.c`` that bench.py times as ``cpu_baseline``;
* :mod:`oracle.np_oracle` -- a numpy-float32 restatement used to cross-check
  the C one.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package.  ``lightmotif_amd`` never does.
"""
