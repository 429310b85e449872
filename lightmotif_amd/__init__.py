"""lightmotif_amd -- MI355X (gfx950) back-end for lightmotif's PWM scoring hot path.

``pssm.score(striped) -> StripedScores -> argmax / max / threshold`` runs in
hand-written HIP kernels behind the C ABI of ``include/lightmotif_hip.h``;
:mod:`lightmotif_amd.lib` mirrors the reference's host-side interface.
The shared library must be built first (``python -m lightmotif_amd.build``);
there is no CPU fallback.
"""
from .lib import *  # noqa: F401,F403
from .lib import __all__ as _lib_all

__version__ = "0.1.0"
__all__ = list(_lib_all)
