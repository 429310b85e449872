#!/usr/bin/env python3
"""Headline benchmark: ``pssm.score()`` of a len-20 DNA PSSM over a 1 Gbp striped
sequence per GPU (BASELINE.json configs[1]; at N GPUs the job is an N Gbp sequence
row-sharded with an M-1-row halo, configs[3]).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...      (no launcher: re-executes itself under torch.distributed.run)

A *step* is one full ``score_into`` (pli/mod.rs:109-117) of the rank's shard into a
resident StripedScores matrix: 1 B read + 4 B written per position.  At N > 1 every step
also produces what ``scores.argmax()`` returns for the WHOLE sequence: the shard's argmax
(tracked by the store kernel) merged over RCCL through the C ABI's own communicator
("wall time of the slowest rank incl. the RCCL merge").  Inputs (SplitMix64
stream) are resident in HBM before the timed region; a time-based preheat runs
before the counted warm-up.  Rank 0 prints ONE JSON line.

PyTorch is plumbing only (device buffers, stream, process group); every timed
kernel is the hand-written HIP code behind include/lightmotif_hip.h.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

# dmabuf IPC: RCCL / device-memory sharing between the ranks' processes needs it on these hosts (set before HIP loads)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import lightmotif_amd as lm  # noqa: E402
from lightmotif_amd import distributed as D  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_POS = 5      # 1 B symbol read + 4 B f32 score written
COLS = 32
# Secondary ceiling: every position gathers M f32 weights from LDS, 4*M bytes at
# 256 B/clk/CU (ds_read_b128, MI355X_MICROARCH.md "LDS") x 256 CUs x 2.4 GHz
LDS_PEAK_BYTES_PER_S = 256 * 256 * 2.4e9

_M64 = (1 << 64) - 1


def _i64(x: int) -> int:
    """Python int -> the int64 with the same 64 bits (torch has no uint64 arithmetic)."""
    x &= _M64
    return x - (1 << 64) if x >> 63 else x


def _lsr(x: torch.Tensor, n: int) -> torch.Tensor:
    """Logical right shift of int64 bit patterns."""
    return (x >> n) & ((1 << (64 - n)) - 1)


def splitmix64_bases(pos: torch.Tensor, seed: int) -> torch.Tensor:
    """base at sequence position `pos` (int64 tensor) of the counter-based
    SplitMix64 stream -- word w = mix(seed + (w + 1) * 0x9E3779B97F4A7C15) holds positions
    32w .. 32w+31, two bits each, low bits first; uniform over {A, C, T, G} = {0, 1, 2, 3}."""
    w = pos >> 5
    z = (w + 1) * _i64(0x9E3779B97F4A7C15) + _i64(seed)         # int64 arithmetic wraps like u64
    z = (z ^ _lsr(z, 30)) * _i64(0xBF58476D1CE4E5B9)
    z = (z ^ _lsr(z, 27)) * _i64(0x94D049BB133111EB)
    z = z ^ _lsr(z, 31)
    return ((z >> ((pos & 31) * 2)) & 3).to(torch.uint8)


def synth_shard(rows: int, row0: int, total_rows: int, total_length: int, halo: int, dev, seed: int = 0x5EED0001,
                chunk: int = 1 << 21) -> torch.Tensor:
    """Rows [row0, row0 + rows) of the striped matrix (pli/mod.rs:191-196: position i at
    [i % R][i / R], N past the end) of the synthetic sequence, + `halo` uninitialised rows."""
    shard = torch.empty((rows + halo, COLS), dtype=torch.uint8, device=dev)
    cols = torch.arange(COLS, device=dev, dtype=torch.int64)[None, :] * total_rows
    for a in range(0, rows, chunk):
        b = min(a + chunk, rows)
        pos = cols + (torch.arange(a, b, device=dev, dtype=torch.int64)[:, None] + row0)
        v = splitmix64_bases(pos, seed)
        v[pos >= total_length] = 4
        shard[a:b] = v
    return shard


_JSON_OUT = None
_HARD_EXIT = False   # a helper thread is stuck in a collective that will never complete: leave with os._exit


def claim_stdout() -> None:
    """Rank 0 owes the driver ONE JSON line on stdout, but RCCL prints a version banner there when a
    communicator comes up (torch's and the library's alike), buffered until the process exits.  So the
    real stdout is set aside for the JSON line and file descriptor 1 is pointed at stderr for everything
    else this process or its libraries print."""
    global _JSON_OUT
    if _JSON_OUT is None:
        sys.stdout.flush()
        _JSON_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def emit(obj: dict) -> None:
    out = _JSON_OUT or sys.stdout
    out.write(json.dumps(obj) + "\n")
    out.flush()


def synth_pssm(m: int, seed: int = 0x5EED0002) -> lm.ScoringMatrix:
    """counts of 10 pseudo-random m-mers -> to_freq(0.1) -> to_scoring(uniform)."""
    rng = np.random.default_rng(seed)
    sites = ["".join("ACTG"[i] for i in rng.integers(0, 4, m)) for _ in range(10)]
    return lm.create(sites).counts.normalize(0.1).log_odds()


def store_kernel_digest() -> str:
    """sha256 over the sources the store kernel is compiled from: `profiles/pmc_traffic.json` records it at
    collection time (tools/collect_pmc.sh), and a line printed from other sources says `traffic_current: false`
    instead of quoting a stale counter figure silently."""
    import hashlib
    h = hashlib.sha256()
    for name in ("score_kernels.hpp", "score_inst.hip", "score_plan.hip", "score_store.hip"):
        h.update((ROOT / "lightmotif_amd" / "csrc" / name).read_bytes())
    return h.hexdigest()[:16]


def cpu_topology() -> dict:
    """{sockets, cores, threads, model} of the host from /proc/cpuinfo: `cores` are physical cores
    (distinct (physical id, core id) pairs), `threads` the hardware threads the OS schedules on."""
    sockets, cores, threads, model = set(), set(), 0, ""
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            key, _, val = line.partition(":")
            key, val = key.strip(), val.strip()
            if key == "processor":
                threads += 1
                phys = core = None
            elif key == "model name" and not model:
                model = val
            elif key == "physical id":
                phys = val
                sockets.add(val)
            elif key == "core id":
                core = val
                cores.add((phys, core))
    except OSError:
        pass
    threads = threads or (os.cpu_count() or 1)
    return {"sockets": len(sockets) or 1, "cores": len(cores) or threads, "threads": threads, "model": model}


def self_launch(n: int) -> None:
    """`python bench.py --gpus N` without a launcher: run the same command line as N ranks of one
    node under torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1) and leave with
    its exit status.  Rank 0's JSON line reaches this process's stdout through the inherited
    descriptor."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on these hosts
    env.setdefault("OMP_NUM_THREADS", "1")
    env["LM_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    print(f"bench.py: --gpus {n} without WORLD_SIZE: launching {n} ranks under torch.distributed.run "
          f"(port {port})", file=sys.stderr, flush=True)
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def device_identity(dev: torch.device) -> str:
    """PCI bus id (or uuid / name) of the GPU a rank runs on: gathered into `config.devices` so the
    line shows N DISTINCT devices, not N ranks on one."""
    p = torch.cuda.get_device_properties(dev)
    for attr in ("pci_bus_id", "uuid"):
        v = getattr(p, attr, None)
        if v not in (None, ""):
            if attr == "pci_bus_id":
                return f"{getattr(p, 'pci_domain_id', 0):04x}:{int(v):02x}:{int(getattr(p, 'pci_device_id', 0)):02x}"
            return str(v)
    return f"{p.name}#{dev.index}"


def cpu_baseline(seq_sample: np.ndarray, length: int, pssm: np.ndarray, gpu_scores: np.ndarray,
                 seconds: float) -> dict:
    """Times the AVX2 port of the reference back-end (oracle/lm_avx2.c follows
    avx2.rs:104-199) on a bounded sample of the same workload, after checking that the
    GPU produced bit-identical scores for that sample."""
    from oracle import c_oracle as co
    m = pssm.shape[0]
    rows = seq_sample.shape[0] - (m - 1)
    data = co.aligned_empty(seq_sample.shape, np.uint8)
    data[:] = seq_sample
    s = co.Striped(data, length, m - 1, COLS, 5)
    p = co.aligned_empty(pssm.shape, np.float32)
    p[:] = pssm
    out = co.aligned_empty((rows, COLS), np.float32)
    threads = os.cpu_count() or 1
    co.avx2_score_rows(s, p, out=out, row_end=rows, threads=threads)
    verified = bool(np.array_equal(out.view(np.uint32), gpu_scores.view(np.uint32)))

    def run(nthreads: int, budget: float) -> float:
        n, t0 = 0, time.perf_counter()
        while True:
            co.avx2_score_rows(s, p, out=out, row_end=rows, threads=nthreads)
            n += 1
            dt = time.perf_counter() - t0
            if dt >= budget:
                return rows * COLS * n / dt / 1e9

    one = run(1, seconds / 2)
    allc = run(threads, seconds / 2)
    # the Generic pipeline (pli/mod.rs:72-106; oracle/lm_oracle.c), one thread, on the head of the same sample: BASELINE.md
    # section 3 lists it next to the AVX2 leg (it is what lightmotif-bench/dna.rs times)
    grows = min(rows, 1 << 19)
    gs = co.Striped(data[:grows + m - 1], length, m - 1, COLS, 5)
    co.score_rows(gs, p, 0, min(grows, 4096))
    t0 = time.perf_counter()
    gout, _ = co.score_rows(gs, p, 0, grows)
    generic = grows * COLS / (time.perf_counter() - t0) / 1e9
    generic_ok = bool(np.array_equal(gout[:, :COLS].view(np.uint32), gpu_scores[:grows].view(np.uint32)))
    topo = cpu_topology()
    return {
        "value": round(allc, 3), "unit": "Gpos/s", "cores": topo["cores"], "threads": threads,
        "sockets": topo["sockets"], "kind": "port",
        "sample": f"first {rows * COLS} positions of rank 0's shard, AVX2 port of avx2.rs:104-199 "
                  f"(oracle/lm_avx2.c), rows split over {threads} software threads = every hardware thread of "
                  f"{topo['sockets']} socket(s) x {topo['cores'] // max(topo['sockets'], 1)} cores "
                  f"({topo['cores']} physical cores, SMT {threads // max(topo['cores'], 1)}), ~{seconds:.0f} s of CPU work",
        "single_thread_gpos": round(one, 3), "cpu_model": topo["model"], "gpu_matches_cpu_bitwise": verified,
        "generic_single_thread_gpos": round(generic, 4),
        "generic_sample": f"first {grows * COLS} positions, Generic restatement of pli/mod.rs:72-106 (oracle/lm_oracle.c), 1 thread",
        "gpu_matches_generic_bitwise": generic_ok,
    }


def init_ranks(args):
    """World of this run.  Under a launcher (torch.distributed.run: WORLD_SIZE / RANK / LOCAL_RANK in
    the environment) the environment is authoritative -- `--gpus` that disagrees is noted in the
    line, not fatal.  Without one, `--gpus N > 1` has already launched the N ranks itself (main -> self_launch).  The
    process group gets a bounded timeout so that a rank that died raises on the others instead of
    hanging them."""
    import datetime
    note = None
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.single_device else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        note = f"--gpus {args.gpus} but the launcher started {world} rank(s): the launcher's world is used"
    if not args.single_device and local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    coll_dev = dev if args.dist_backend == "nccl" else torch.device("cpu")  # where collectives run
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        timeout = datetime.timedelta(seconds=args.comm_timeout_s)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=timeout)
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world, timeout=timeout)
    # the C ABI's collectives wait this long at most (csrc/comm.hip: bounded waits)
    os.environ.setdefault("LM_HIP_COMM_TIMEOUT_MS", str(int(args.comm_timeout_s * 1000)))
    return world, rank, local_rank, dev, coll_dev, note


def best_kmer_score(p) -> np.float32:
    """the sequential f32 sum of the row maxima: no score of the matrix exceeds it (score_plan.hip best_kmer_score)"""
    b = np.float32(0)
    for row in p.data[:, :4]:
        b = np.float32(b + row.max())
    return b


def c3_setup(pli, dev, world: int, rank: int, length: int, motifs: int = 0) -> dict:
    """BASELINE.json configs[2] on this rank: the JASPAR 2024 CORE matrices (the reference's bench fixture, converted like
    the CLI, main.rs:469-498), thresholds at p = 1e-5, the WHOLE synthetic sequence resident on every rank, and the motif
    list cut into `world` shares balanced on the expected scan cost (lightmotif-cli main.rs:502-561 fans (motif, sequence)
    jobs out to threads; across GPUs the motif list is the unit)."""
    from lightmotif_amd import io as lmio
    pssms = [r.matrix.normalize(0.1).log_odds() for r in lmio.read(ROOT / "tests" / "golden" / "JASPAR2024.pwm.gz")]
    if motifs:
        pssms = pssms[:motifs]
    lengths = [len(p) for p in pssms]
    max_m = max(lengths)
    rows = -(-length // COLS)
    shard = synth_shard(rows, 0, rows, length, max_m - 1, dev, seed=0x5EED0003)
    pli.configure_wrap_dptr(shard.data_ptr(), rows, COLS, COLS, max_m - 1, 4)
    torch.cuda.synchronize()
    seq = pli.adopt_sequence(shard.data_ptr(), rows, max_m - 1, COLS, COLS, length, keepalive=shard)
    ts = [p.score_for_pvalue(1e-5) for p in pssms]
    unreachable = [i for i, (p, t) in enumerate(zip(pssms, ts)) if np.float32(t) > best_kmer_score(p)]
    # shard on the expected cost of a motif's scan (profiles/r02_c3_per_length.txt: ~16-byte table reads per pair of
    # positions), not on its bare length: a motif that cannot reach the threshold costs (almost) nothing
    skip = set(unreachable)
    cost = [0.02 if i in skip else (1 + (m | 3) // 8) for i, m in enumerate(lengths)]
    parts = D.shard_motifs(cost, world)
    for i in parts[rank]:
        pssms[i]._device(pli)
    cells = sum(length + 1 - m for m in lengths)
    scanned = cells - sum(length + 1 - lengths[i] for i in unreachable)
    # LDS bytes the pair-prefilter scans gather (score_prefilter2.hpp: one table row of (M | 3) + 1 u16 per pair of input
    # rows = (M | 3) + 1 bytes per position), summed over the motifs that are scanned
    lds_bytes = sum(((m | 3) + 1) * (length + 1 - m) for i, m in enumerate(lengths) if i not in skip)
    return {"pssms": pssms, "lengths": lengths, "ts": ts, "seq": seq, "shard": shard, "parts": parts,
            "unreachable": unreachable, "rows": rows, "max_m": max_m, "length": length, "cells": cells,
            "scanned_cells": scanned, "lds_bytes": lds_bytes}


C3_LDS_MODEL = ("pair-prefilter scans (score_prefilter2.hpp): one LDS table row of (M | 3) + 1 u16 entries per pair of input "
                "rows = (M | 3) + 1 bytes gathered per scanned (motif, position) cell, against 256 B/clk/CU x 256 CUs x 2.4 GHz; "
                "whole call on the wall clock (re-scoring, ordering and read-back of the hits included)")


def at_sustained_clock(mhz: float, lds_bytes_per_s: float, valu_lane_ops_per_s: float, valu_model: str) -> dict:
    """The LDS and VALU fractions of a kernel at the shader clock it was MEASURED to sustain (lm_hip_device_clock_mhz
    beside the calls): 256 B/clk/CU of LDS reads and 64 lanes/clk/CU of 32-bit VALU issue on 256 CUs."""
    if not mhz or mhz <= 0:
        return {"sclk_mhz_sustained": None}
    hz = mhz * 1e6
    return {"sclk_mhz_sustained": round(mhz, 1),
            "lds_frac_at_sustained_clock": round(lds_bytes_per_s / (256 * 256 * hz), 4),
            "valu_frac_at_sustained_clock": round(valu_lane_ops_per_s / (256 * 64 * hz), 4),
            "valu_model": valu_model + "; 64 lanes/clk/CU x 256 CUs"}


def scan_kernel_ms(pli, call, reps: int = 25, warm: int = 40, phases=None):
    """Median duration of the scan kernel alone inside a fused call: HIP events around it on the library's stream
    (context option "time_scan", lm_hip_ctx_last_scan_kernel_ms) -- measured in calls of their own, not in the timed ones,
    after `warm` calls (behind a pause the device runs at a lower clock: 0.25 against 0.22 ms).  `phases` (a dict): filled
    with the medians of the threshold call's phases (lm_hip_ctx_last_phases_ms)."""
    pli.set_option("time_scan", 1)
    try:
        ms, ph = [], []
        for i in range(warm + reps):
            call()
            k = pli.last_scan_kernel_ms
            if k is not None and i >= warm:
                ms.append(k)
                p = pli.last_phases_ms
                if p and p[1] >= 0 and p[2] >= 0:
                    ph.append(p)
    finally:
        pli.set_option("time_scan", 0)
    if phases is not None and ph:
        med = np.median(np.asarray(ph), axis=0)
        phases.update({"scan_us": round(float(med[0]) * 1e3, 1), "rescore_us": round(float(med[1]) * 1e3, 1),
                       "order_us": round(float(med[2]) * 1e3, 1), "host_us": round(float(med[3]) * 1e3, 1),
                       "what": "medians over calls with events recorded (HIP events on the library's stream: scan kernel | exact "
                               "re-scoring | ordering kernels; host = the C call minus those: enqueueing, the synchronisation's "
                               "wake-up, the copy of the hits out of the pinned block)"})
    return float(np.median(ms)) if ms else None


def fused_roofline(ms: float, rows: int, m: int, kernel: str, mhz: float = 0.0, kernel_ms=None, scan_info=(0, 0),
                   timing: str = "minimum of 5 single calls, each between two device synchronisations, after 60 untimed calls") -> dict:
    """A fused score+argmax / score+threshold call over `rows` x 32 positions, whole call on the wall clock (scan, re-scoring,
    reductions, read-back).  SURVEY 8(d): no score matrix is written, so the binding ceiling is the LDS gather -- the table
    bytes per position the scan kernel that RAN looks up (`scan_info` = lm_hip_ctx_last_scan_info: the motif rows it scanned
    and what they cost; a pair scan reads one row of ((rows | 3) + 1) u16 entries per two positions, the protein block scan a
    row of prefilter_mp entries per position) against 256 B/clk/CU x 256 CUs x 2.4 GHz; the one byte per position the scan must
    still read from HBM is reported beside it (`hbm_read_frac`), not as the bound."""
    scanned, row_bytes = scan_info
    if not row_bytes:           # (no prefilter / exact scan kernel ran: price the motif's own pair table)
        scanned, row_bytes = m, (m | 3) + 1
    lds = row_bytes * rows * COLS / (ms * 1e-3)
    ach = rows * COLS / (ms * 1e-3) / 1e9
    return {"ms": round(ms, 4), "kernel": kernel, "Gpos_s": round(rows * COLS / ms / 1e6, 1), "timing": timing,
            # what a call spends outside its scan kernel: re-scoring, ordering / reduction of the hit list, the launches'
            # gaps, the synchronisation, the copy of the results out of the pinned block, the Python wrapper
            **({"tail_us": round((ms - kernel_ms) * 1e3, 1)} if kernel_ms else {}),
            "roofline": {"bound": "lds", "achieved": round(lds / 1e12, 2), "peak": round(LDS_PEAK_BYTES_PER_S / 1e12, 1),
                         "unit": "TB/s", "frac": round(lds / LDS_PEAK_BYTES_PER_S, 4),
                         "lds_bytes_per_position": row_bytes, "motif_rows_scanned": scanned, "hbm_read_gbs": round(ach, 1),
                         "hbm_read_frac": round(ach / HBM_PEAK_GBS, 4), "algorithmic_hbm_bytes_per_call": rows * COLS,
                         # the scan kernel alone (what the LDS ceiling bounds), without the call's launch-bound tail
                         **({"kernel_ms": round(kernel_ms, 4),
                             "kernel_frac": round(row_bytes * rows * COLS / (kernel_ms * 1e-3) / LDS_PEAK_BYTES_PER_S, 4)}
                            if kernel_ms else {}),
                         # the pair scan's issue per position and lane: (NP + 1) accumulate operations + 6 of decode per
                         # pair of super-steps = 4 positions (score_prefilter2.hpp), NP = row bytes / 2
                         **at_sustained_clock(mhz, lds, (row_bytes // 2 + 7) / 4 * rows * COLS / (ms * 1e-3),
                                              f"{(row_bytes // 2 + 7) / 4:.2f} VALU operations per position "
                                              "(v_add3_u32 accumulation + register decode of the pair scan)")}}


def lds_roofline(lds_bytes: float, seconds: float, note: str) -> dict:
    """scans that never touch HBM per cell are priced against the LDS-gather ceiling."""
    ach = lds_bytes / seconds / 1e12
    peak = LDS_PEAK_BYTES_PER_S / 1e12
    return {"bound": "lds", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TB/s", "frac": round(ach / peak, 4),
            "model": note}



def main_c3(args) -> None:
    """BASELINE.json configs[2]: the 2 346 DNA matrices of JASPAR 2024 CORE (the reference's own
    bench fixture, tests/golden/JASPAR2024.pwm.gz; converted like the CLI, main.rs:469-498) over a
    100 Mbp resident sequence.  The reference fans (motif, sequence) jobs out to worker threads
    (lightmotif-cli main.rs:502-561); across GPUs the motif list is sharded (balanced on sum M),
    every rank holds the whole sequence, and the per-motif results are gathered in motif order.
    Step = one batched fused threshold scan at p = 1e-5 per motif of this rank's share + the
    gather.  Value = (motif, position) cells per second over all ranks; scaling is STRONG (the
    total work is fixed)."""
    world, rank, local_rank, dev, coll_dev, launch_note = init_ranks(args)
    stream = torch.cuda.current_stream()
    pli = lm.Pipeline.hip(local_rank, stream=stream.cuda_stream)
    length = args.length if args.length != 1_000_000_000 else 100_000_000
    c3 = c3_setup(pli, dev, world, rank, length, args.motifs)
    pssms, lengths, ts, seq, shard, parts, unreachable, rows, max_m = (c3[k] for k in (
        "pssms", "lengths", "ts", "seq", "shard", "parts", "unreachable", "rows", "max_m"))

    # the rank's motif list in the form the C ABI takes it, built once (the CLI converts its motifs once as well,
    # main.rs:469-498, and then loops over sequences)
    prepared = D.prepare_sharded_batch(pli, pssms, ts, parts=parts)

    def step():
        return D.scan_threshold_batch_sharded(pli, pssms, ts, seq, device=coll_dev, parts=parts, prepared=prepared)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        res = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    am = D.scan_argmax_batch_sharded(pli, pssms, seq, device=coll_dev, parts=parts)
    t1 = time.perf_counter()
    for _ in range(max(args.steps // 2, 1)):
        am = D.scan_argmax_batch_sharded(pli, pssms, seq, device=coll_dev, parts=parts)
    barrier()
    am_s = (time.perf_counter() - t1) / max(args.steps // 2, 1)
    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    cells, scanned_cells = c3["cells"], c3["scanned_cells"]
    value = cells * args.steps / elapsed / 1e9
    out = {
        "metric": "scored (motif, position) cells/sec, fused threshold scan", "value": round(value, 1), "unit": "Gcell/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u16 prefilter + f32 re-scoring",
        "data": "synthetic sequence (SplitMix64), JASPAR 2024 CORE matrices (reference fixture)",
        "config": {"workload": f"configs[2]: {len(pssms)} JASPAR DNA PSSMs (sum M = {sum(lengths)}) x {length} bp resident, "
                               "fused threshold at p = 1e-5 per motif, hits in the reference's order; "
                               f"{len(unreachable)} motifs (all of length <= {max([lengths[i] for i in unreachable] or [0])}) "
                               "cannot reach p = 1e-5 -- their threshold exceeds the score of their best k-mer -- and "
                               "are answered (no hits) without a scan; `value` counts "
                               "their cells as done, `extras.scanned_Gcell_s` does not",
                   "parallelism": f"motif-shard x{world} (LPT on the expected scan cost per motif), sequence replicated",
                   "motifs_per_rank": [len(p) for p in parts]},
        "extras": {"hits_total": int(sum(len(c) for c, _ in res)),
                   "motifs_unreachable_at_p": len(unreachable),
                   "scanned_Gcell_s": round(scanned_cells * args.steps / elapsed / 1e9, 1),
                   "fused_argmax_ms": round(am_s * 1e3, 3),
                   "fused_argmax_Gcell_s": round(cells / am_s / 1e9, 1),
                   "argmax_found": int(sum(a is not None for a in am))},
        "roofline": lds_roofline(c3["lds_bytes"] * args.steps, elapsed, C3_LDS_MODEL),
        "roofline_note": "LDS-gather-bound scans over a cache-resident sequence: the PMC-based LDS / VALU utilisation "
                         "per kernel is in profiles/r02_c3_record.md; no HBM fraction applies",
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_c3(shard, rows, length, max_m, pssms, res, am, args.cpu_seconds)
    emit(out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline_c3(shard, rows, length, max_m, pssms, gpu_thr, gpu_am, seconds: float) -> dict:
    """The AVX2 port on all host threads over a stratified subset of the motifs (every length once
    if time allows) and the first 20 Mbp-equivalent rows of the sequence: score_rows + argmax per
    motif (what one reference worker does per job, main.rs:554-561), GPU results of the same
    motifs checked on that sample beforehand."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import c_oracle as co
    srows = min(rows, 625_000)                      # 20 Mbp worth of striped rows
    lengths = [len(p) for p in pssms]
    subset = [lengths.index(m) for m in sorted(set(lengths))]
    threads = os.cpu_count() or 1
    workers = min(threads, 128)
    data = co.aligned_empty((srows + max_m - 1, COLS), np.uint8)
    data[:] = shard[:srows + max_m - 1].cpu().numpy()
    block = 8192                                    # rows per call: the f32 block (1 MB) stays in cache, like the
                                                    # reference Scanner's 256-row blocks (scan.rs:174-178)
    tpv = {i: np.float32(pssms[i].score_for_pvalue(1e-5)) for i in subset}

    def job(i):
        """One (motif, sequence) job of a CLI worker (main.rs:554-561): score block by block, keep the
        best cell and count the hits; single-threaded, the fan-out is over jobs like rayon's."""
        m = lengths[i]
        s = co.Striped(data[:srows + m - 1], length, m - 1, COLS, 5)
        p = co.aligned_empty(pssms[i].data.shape, np.float32)
        p[:] = pssms[i].data
        out = co.aligned_empty((block, COLS), np.float32)
        hits = []
        for a in range(0, srows, block):
            b = min(a + block, srows)
            co.avx2_score_rows(s, p, out=out[:b - a], row_begin=a, row_end=b, threads=1)
            co.avx2_argmax(out[:b - a], (b - a) * COLS)
            rc = np.argwhere(out[:b - a, :COLS] >= tpv[i])
            if rc.size:
                rc[:, 0] += a
                hits.append(rc)
        return i, (np.concatenate(hits) if hits else np.zeros((0, 2), np.int64))

    ok, cells, n, t0 = True, 0, 0, time.perf_counter()
    with ThreadPoolExecutor(max_workers=workers) as ex:
        while time.perf_counter() - t0 < seconds:
            jobs = [subset[k % len(subset)] for k in range(workers)]
            for i, rc in ex.map(job, jobs):
                cells += srows * COLS
                if n == 0:                              # the GPU's hits of this motif inside the sample == the CPU's
                    g = gpu_thr[i][0]
                    ok = ok and np.array_equal(g[g[:, 0] < srows], rc)
            n += 1
    dt = time.perf_counter() - t0
    return {"value": round(cells / dt / 1e9, 2), "unit": "Gcell/s", "cores": workers, "kind": "port",
            "sample": f"{len(subset)} motifs (one per length) x first {srows * COLS} positions; one single-threaded job per "
                      f"(motif, sequence) like the CLI's workers, {workers} jobs in flight: AVX2 port (oracle/lm_avx2.c) "
                      f"score_rows in cache-resident {block}-row blocks + argmax + threshold, {n} rounds, ~{seconds:.0f} s",
            "gpu_hits_match_on_sample": bool(ok)}


def host_pointer_bench():
    """tools/host_pointer_bench.py as a module: the measurements of the literal drop-in path are shared with that tool"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("host_pointer_bench", ROOT / "tools" / "host_pointer_bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def measured_d2h_gbs(dev, nbytes: int = 1 << 28) -> float:
    """The floor of the host-pointer path: device -> pinned host memory, GB/s (what a copy command reaches on this box)."""
    host = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
    src = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    best = 0.0
    for _ in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        host.copy_(src, non_blocking=True)
        b.record()
        torch.cuda.synchronize()
        best = max(best, nbytes / (a.elapsed_time(b) * 1e-3) / 1e9)
    return best


def end_to_end(shard, rows: int, m: int, length: int, pssm, scores_h, dev) -> dict:
    """SURVEY 8(d) "End-to-end ... reported separately and labelled": the same 1 Gbp job through the host-pointer entry
    point the reference-side shim binds (`lm_hip_score_f32`: striped sequence and score matrix in pageable HOST memory,
    1 B per position up + 4 B per position down over PCIe per call, pwm/mod.rs:640-648).  Never `value`."""
    hpb = host_pointer_bench()
    mat = shard.cpu().numpy()                       # rows + M - 1 wrap rows, as seq.rs:369-381 leaves them on the host

    def check(out):                                 # against the resident scores of the timed region: head, tail, a middle window
        ok = True
        for a in (0, max(rows // 2 - 50_000, 0), max(rows - 100_000, 0)):
            b = min(a + 100_000, rows)
            ok = ok and np.array_equal(out[a:b].view(np.uint32), scores_h.rows_matrix(a, b)[:, :COLS].view(np.uint32))
        return ok
    res = hpb.bench_big(length, m, mat=mat, pssm=pssm.data, reps=4, check=check)
    d2h = measured_d2h_gbs(dev)
    floor_ms = 4 * rows * COLS / d2h / 1e6
    return {"host_pointer_1gbp_ms": res["host_pointer_ms"], "host_pointer_median_ms": res["host_pointer_median_ms"],
            "gpos": res["gpos"], "link_gbs": res["link_gbs"], "d2h_gbs": res["d2h_gbs"],
            "measured_pinned_d2h_gbs": round(d2h, 1), "d2h_floor_ms": round(floor_ms, 2),
            "frac_of_d2h_floor": round(floor_ms / res["host_pointer_ms"], 4), "matches_resident_scores": res["verified"],
            "positions": rows * COLS,
            "what": "lm_hip_score_f32 on pageable host matrices (what INTEGRATION.md 2 binds): tiles of 262144 rows, pageable "
                    "H2D by an uploader thread | store kernel | D2H into a pinned ring + 4 copier threads; the floor is the 4 B "
                    "per position that must come back over PCIe at the pinned D2H rate measured in this run"}


def c3_leg(pli, dev, coll_dev, world: int, rank: int, reps: int = 5) -> dict:
    """configs[2] after the headline run: ms per batched fused threshold scan (p = 1e-5 per motif) and per batched fused
    argmax of the JASPAR matrices x 100 Mbp.  world > 1: the motif list sharded over the ranks (every rank holds the
    sequence, results gathered in motif order: `scan_threshold_batch_sharded`), slowest rank counts; all ranks call this."""
    length = 100_000_000
    c3 = c3_setup(pli, dev, world, rank, length)
    pssms, ts, seq, parts = c3["pssms"], c3["ts"], c3["seq"], c3["parts"]
    res = [None]

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, n, warm=2):
        for _ in range(warm):
            fn()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        sync_all()
        dt = (time.perf_counter() - t0) / n
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # the motif list in the form the C ABI takes it, built once (a job loop scans sequence after sequence with the same
    # motifs, lightmotif-cli main.rs:502-561)
    prepared = pli.prepare_batch(pssms, ts) if world == 1 else D.prepare_sharded_batch(pli, pssms, ts, parts=parts)

    def scan():
        res[0] = (pli.scan_threshold_batch(prepared, None, seq) if world == 1 else
                  D.scan_threshold_batch_sharded(pli, pssms, ts, seq, device=coll_dev, parts=parts, prepared=prepared))
    t_th = timed(scan, reps)
    k_th = pli.last_kernel
    t_am = timed(lambda: (pli.scan_argmax_batch(pssms, seq) if world == 1 else
                          D.scan_argmax_batch_sharded(pli, pssms, seq, device=coll_dev, parts=parts)), reps)
    lengths = c3["lengths"]
    k_am = pli.last_kernel
    # the threshold batch by phase, from the library's own events (context option "time_scan"; calls of their own, after the
    # timed ones): scan kernels | exact re-scoring | ordering kernels on the stream, the rest of the call on the host clock
    phases = None
    if world == 1:
        pli.set_option("time_scan", 1)
        try:
            rec = []
            for _ in range(3):
                t0 = time.perf_counter()
                scan()
                ph = pli.last_phases_ms
                if ph:
                    rec.append(list(ph) + [(time.perf_counter() - t0) * 1e3])
            if rec:
                med = np.median(np.asarray(rec), axis=0)
                phases = {"scan_ms": round(float(med[0]), 3), "rescore_ms": round(float(med[1]), 3), "order_ms": round(float(med[2]), 3),
                          "host_ms": round(float(med[3]), 3), "python_call_ms": round(float(med[4]), 3),
                          "scan_frac_of_lds_ceiling": round(c3["lds_bytes"] / (float(med[0]) * 1e-3) / LDS_PEAK_BYTES_PER_S, 4),
                          "what": "medians of 3 calls with events recorded; host_ms = the C call minus the three event spans "
                                  "(enqueueing ~50 launches, the synchronisation, the copy of the hit lists out of the pinned block); "
                                  "python_call_ms adds the wrapper (per-motif numpy views)"}
        finally:
            pli.set_option("time_scan", 0)
    realistic = None
    if world == 1:
        # the same batch on a NON-i.i.d. sequence of the same length (5 % N in runs, microsatellites / homopolymers, 35 % /
        # 65 % GC isochores: tools/realistic_inputs.py, which also checks it against the oracle): candidate density is what
        # the fused scans' cost depends on beyond the scan itself
        sys.path.insert(0, str(ROOT / "tools"))
        import realistic_inputs as ri
        enc = ri.realistic_dna(length)
        rseq = pli.stripe(lm.EncodedSequence(enc))
        rseq.configure_wrap(c3["max_m"] - 1)
        rres = [None]

        def rscan():
            rres[0] = pli.scan_threshold_batch(prepared, None, rseq)
        rt_th = timed(rscan, max(reps - 1, 2))
        rhits, rcands = pli.last_scan_counts
        rphases = None
        pli.set_option("time_scan", 1)   # where the non-i.i.d. input's extra time goes: the same phases as above
        try:
            rec = []
            for _ in range(2):
                rscan()
                if pli.last_phases_ms:
                    rec.append(list(pli.last_phases_ms))
            if rec:
                med = np.median(np.asarray(rec), axis=0)
                rphases = {"scan_ms": round(float(med[0]), 3), "rescore_ms": round(float(med[1]), 3),
                           "order_ms": round(float(med[2]), 3), "host_ms": round(float(med[3]), 3)}
        finally:
            pli.set_option("time_scan", 0)
        rt_am = timed(lambda: pli.scan_argmax_batch(pssms, rseq), max(reps - 1, 2))
        scan()
        uhits, ucands = pli.last_scan_counts
        realistic = {"sequence": ri.describe(enc), "fused_threshold_ms": round(rt_th * 1e3, 3), "fused_argmax_ms": round(rt_am * 1e3, 3),
                     "hits_total": int(sum(len(c) for c, _ in rres[0])), "candidate_pieces_per_hit": round(rcands / max(rhits, 1), 2),
                     "uniform_candidate_pieces_per_hit": round(ucands / max(uhits, 1), 2),
                     "threshold_ms_over_uniform": round(rt_th / t_th, 3), "argmax_ms_over_uniform": round(rt_am / t_am, 3),
                     **({"phases": rphases} if rphases else {}),
                     "parity": "tools/realistic_inputs.py -> profiles/r05_realistic_inputs.json (whole-sequence check against the AVX2 port)"}
        del rseq, enc
    return {"workload": f"configs[2]: {len(pssms)} JASPAR 2024 CORE DNA PSSMs (sum M = {sum(lengths)}) x "
                        f"{length} bp resident, one batched fused threshold scan at p = 1e-5 per motif",
            "parallelism": f"motif-shard x{world} (LPT on the expected scan cost per motif), sequence replicated",
            "motifs_per_rank": [len(p) for p in parts],
            "fused_threshold_ms": round(t_th * 1e3, 3), "Gcell_per_s": round(c3["cells"] / t_th / 1e9, 1),
            "scanned_Gcell_per_s": round(c3["scanned_cells"] / t_th / 1e9, 1),
            "motifs_skipped_unreachable": len(c3["unreachable"]),
            "hits_total": int(sum(len(c) for c, _ in res[0])), "kernel": k_th,
            "roofline": lds_roofline(c3["lds_bytes"], t_th, C3_LDS_MODEL),
            **({"phases": phases} if phases else {}),
            "fused_argmax_ms": round(t_am * 1e3, 3), "fused_argmax_kernel": k_am,
            **({"realistic": realistic} if realistic else {})}


def secondary_configs(pli, dev) -> dict:
    """BASELINE.json configs[0], [2], [4] on this box, after the timed region of the headline run (a
    few seconds in all): the driver's line then carries every single-GPU configuration, not only
    configs[1].  Times are wall time per call from this process (launch + synchronisation included)
    unless marked `kernel_ms` (HIP events on the launch stream)."""
    out = {}
    stream = torch.cuda.current_stream()

    def wall(fn, reps, warm=3):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts))

    def events(fn, reps, warm=5):
        for _ in range(warm):
            fn()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in ev:
            a.record(stream)
            fn()
            b.record(stream)
        torch.cuda.synchronize()
        return float(np.median([a.elapsed_time(b) for a, b in ev]))

    # --- configs[0]: lightmotif-bench/dna.rs:81-109 as is -- MX000001 (M = 15) over the first tenth of
    # E. coli K12 (464 165 bp; a seeded random stand-in with the site planted at 391 677: the genome file is
    # absent from the reference mount), one StripedScores re-used, timed body = score_into + argmax
    length = 464_165
    rng = np.random.default_rng(0xEC011)
    enc = rng.integers(0, 4, length, dtype=np.uint8)
    enc[391_677:391_677 + 15] = lm.EncodedSequence("GTTGACCTTATCAAC").data
    pssm = lm.create(["GTTGACCTTATCAAC", "GTTGATCCAGTCAAC"]).counts.normalize(0.1).log_odds()
    c1 = {"workload": "configs[0]: MX000001 (M = 15) x 464165 bp stand-in for E. coli/10 (site planted at 391677), "
                      "score_into + argmax per iteration into one re-used StripedScores (dna.rs:104-107)"}
    # (a context of its own on a stream the library owns, like the reference-side shim's default context: launches on
    #  torch's legacy default stream, which the headline run borrows, cost ~10 us more each)
    pli_main, pli = pli, lm.Pipeline.hip(dev.index)
    for cols, tag in ((32, "C32_dispatch_geometry"), (1, "C1_generic_bench_geometry")):
        seq = pli.stripe(lm.EncodedSequence(enc), cols)
        seq.configure(pssm)
        scores = lm.StripedScores.empty(pli, cols)
        best = [None]

        def it():
            pli.score_into(pssm, seq, scores)
            best[0] = pli.argmax(scores)

        def loop(fn, reps=2000, warm=200):      # the calls return their result: no device-wide synchronise in the loop
            for _ in range(warm):
                fn()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            return (time.perf_counter() - t0) / reps
        t = loop(it)
        k_store = pli.last_kernel
        tf = loop(lambda: pli.score_argmax(pssm, seq))
        c1[tag] = {"us_per_iter": round(t * 1e6, 2), "Mpos_per_s": round(length / t / 1e6, 1),
                   "best_position": int(scores.offset(*best[0])), "kernel": k_store,
                   "fused_score_argmax_us": round(tf * 1e6, 2), "fused_kernel": pli.last_kernel}
        del seq, scores
    # ... and the same loop as the reference-side shim runs it: `lm_hip_score_f32` + `lm_hip_argmax_f32` on
    # HOST matrices, next to the 1-thread AVX2 port of that loop on this box; Scanner's 256-row u8 block (scan.rs:174-178)
    hpb = host_pointer_bench()
    c1.update(hpb.bench_c1())
    c1.update(hpb.bench_block())
    out["c1"] = c1
    out["readme_10kb"] = hpb.bench_readme_10kb()

    # --- the reference's own published benchmark (README.md:102-108): `score` of MX000001 (M = 15) over
    # the WHOLE E. coli K12 genome, 4 641 652 bp -- AVX2 4.51 ms, Generic 317.7 ms on an i7-10710U, one thread.  The genome
    # file is absent from the reference mount, so a seeded random stand-in of that length; the same-box AVX2 port beside it.
    from oracle import c_oracle as co
    glen = 4_641_652
    genc = np.random.default_rng(0xEC012).integers(0, 4, glen, dtype=np.uint8)
    gseq = pli.stripe(lm.EncodedSequence(genc))
    gseq.configure(pssm)
    gscores = lm.StripedScores.empty(pli, COLS)
    pli.set_track_argmax(False)
    t_res = loop(lambda: (pli.score_into(pssm, gseq, gscores), pli.sync()), reps=300, warm=30)
    gmat = gseq.matrix()
    grows = gseq.rows
    gout = np.zeros((grows, COLS), np.float32)
    t_host, _ = hpb.loop_us(lambda: hpb.score_f32(gmat, grows, glen, pssm.data, gout), 40, 5)
    cs = co.Striped(co.aligned_empty(gmat.shape, np.uint8), glen, pssm.data.shape[0] - 1, COLS, 5)
    cs.data[:] = gmat
    cp = co.aligned_empty(pssm.data.shape, np.float32)
    cp[:] = pssm.data
    cout = co.aligned_empty((grows, COLS), np.float32)
    t_cpu, _ = hpb.loop_us(lambda: co.avx2_score_rows(cs, cp, out=cout, row_end=grows, threads=1), 10, 2)
    out["readme_benchmark"] = {
        "workload": "README.md:102-108 `score` f32 of MX000001 (M = 15) over a whole-E.-coli-sized sequence (4641652 bp, random "
                    "stand-in), one call",
        "published_i7_10710U_avx2_ms": 4.511, "published_i7_10710U_generic_ms": 317.7,
        "resident_score_into_ms": round(t_res * 1e3, 4), "host_pointer_score_f32_ms": round(t_host / 1e3, 4),
        "same_box_avx2_port_1_thread_ms": round(t_cpu / 1e3, 3),
        "host_pointer_matches_avx2_port_bitwise": bool(np.array_equal(gout.view(np.uint32), cout.view(np.uint32)))}
    del gseq, gscores
    pli = pli_main

    # --- configs[4]: protein (K = 21) len-12 PSSM x 200 Mres: score() materialised + fused threshold
    length, m = 200_000_000, 12
    rows = -(-length // COLS)
    prng = np.random.default_rng(5)
    sym = lm.lib.PROTEIN_SYMBOLS[:-1]
    sites = ["".join(sym[i] for i in prng.integers(0, len(sym), m)) for _ in range(6)]
    ppssm = lm.create(sites, protein=True).counts.normalize(0.1).log_odds()
    gen = torch.Generator(device=dev)
    gen.manual_seed(55)
    pseq = torch.empty((rows + m - 1, COLS), dtype=torch.uint8, device=dev)
    pseq[:rows] = torch.randint(0, 20, (rows, COLS), dtype=torch.uint8, device=dev, generator=gen)
    pli.configure_wrap_dptr(pseq.data_ptr(), rows, COLS, COLS, m - 1, 20)
    pout = torch.empty((rows, COLS), dtype=torch.float32, device=dev)

    def pscore():
        pli.score_dptr(ppssm, pseq.data_ptr(), rows + m - 1, COLS, COLS, m - 1, length, 0, rows, pout.data_ptr(), COLS)
    for _ in range(30):
        pscore()
    kms = events(pscore, 50)
    k_store = pli.last_kernel
    sample = pout[: 1 << 18].flatten()
    thr = float(torch.quantile(sample[torch.isfinite(sample)].float(), 1 - 1e-5))
    fth = lambda: pli.score_threshold_dptr(ppssm, pseq.data_ptr(), rows + m - 1, COLS, COLS, m - 1, length, 0, rows, thr)  # noqa: E731
    t_th = wall(fth, 20, warm=40)
    k_th, i_th = pli.last_kernel, pli.last_scan_info
    n_hits = len(fth()[0])
    ph_th = {}
    km_th = scan_kernel_ms(pli, fth, phases=ph_th)
    fam = lambda: pli.score_argmax_dptr(ppssm, pseq.data_ptr(), rows + m - 1, COLS, COLS, m - 1, length, 0, rows)  # noqa: E731
    t_am = wall(fam, 20, warm=40)
    k_am, i_am = pli.last_kernel, pli.last_scan_info
    km_am = scan_kernel_ms(pli, fam)
    c5_timing = "median of 20 single calls, each followed by a device synchronisation, after 40 untimed calls"
    out["c5"] = {"workload": "configs[4]: protein (K = 21) len-12 PSSM x 200 Mres, score() materialised",
                 "kernel": k_store, "kernel_ms": round(kms, 4), "Gpos_per_s": round(rows * COLS / kms / 1e6, 1),
                 "hbm_frac": round(BYTES_PER_POS * rows * COLS / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                 "roofline": {"bound": "hbm", "achieved": round(BYTES_PER_POS * rows * COLS / (kms * 1e-3) / 1e9, 1),
                              "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(BYTES_PER_POS * rows * COLS / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                              "algorithmic_bytes_per_launch": BYTES_PER_POS * rows * COLS,
                              "lds_frac": round(4 * m * rows * COLS / (kms * 1e-3) / LDS_PEAK_BYTES_PER_S, 4)},
                 "fused_threshold_ms": round(t_th * 1e3, 4), "fused_threshold_kernel": k_th, "fused_threshold_hits": n_hits,
                 "fused_argmax_ms": round(t_am * 1e3, 4), "fused_argmax_kernel": k_am,
                 # the fused scans against the LDS-gather ceiling at the table bytes their kernel reads per residue
                 # (score_prefilter_blk.hpp: a row of prefilter_mp u16 entries per step), call and kernel alone
                 "fused_threshold": {**fused_roofline(t_th * 1e3, rows, m, k_th, 0.0, km_th, i_th, c5_timing),
                                     **({"phases": ph_th} if ph_th else {})},
                 "fused_argmax": fused_roofline(t_am * 1e3, rows, m, k_am, 0.0, km_am, i_am, c5_timing)}
    del pseq, pout

    # --- configs[2]: the 2 346 JASPAR 2024 CORE matrices x 100 Mbp, fused threshold at p = 1e-5 per motif
    if (ROOT / "tests" / "golden" / "JASPAR2024.pwm.gz").exists():
        out["c3"] = c3_leg(pli, dev, torch.device("cpu"), 1, 0, reps=5)
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2", choices=["c2", "c3"],
                    help="c2 (default): the headline score() of configs[1] / configs[3]; c3: the JASPAR batch of configs[2]")
    ap.add_argument("--motifs", type=int, default=0, help="c3: use only the first N matrices (development)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--length", type=int, default=1_000_000_000, help="positions per GPU")
    ap.add_argument("--motif-len", type=int, default=20)
    ap.add_argument("--preheat-ms", type=float, default=300.0,
                    help="untimed launches for at least this long BEFORE the counted warm-up: the first "
                         "~50 launches of a fresh process run ~6 %% slow while the clocks ramp")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-sample", type=int, default=256_000_000, help="positions in the CPU sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rows-per-stream", type=int, default=0)
    ap.add_argument("--dist-backend", default="nccl",
                    help="development: 'gloo' + --single-device exercises the N>1 control flow on a 1-GPU box")
    ap.add_argument("--single-device", action="store_true", help="development: every rank uses cuda:0")
    ap.add_argument("--require-distinct-devices", action="store_true",
                    help="development: with --single-device, keep the N-ranks-on-N-devices check (the line becomes `invalid`, "
                         "exit status 3) -- the check is always on without --single-device")
    ap.add_argument("--sync-merge", action="store_true",
                    help="N > 1 through the C ABI: wait for every step's merge before the next score_into "
                         "(default: pipelined, lm_hip_argmax_sharded_begin / _end)")
    ap.add_argument("--merge", default="auto", choices=["auto", "cabi", "torch"],
                    help="transport of the per-step argmax merge at N > 1: the C ABI's own RCCL communicator "
                         "(lm_hip_comm_*, default on nccl) or torch.distributed (gloo runs)")
    ap.add_argument("--comm-timeout-s", type=float, default=120.0,
                    help="N > 1: bound on every collective's wait (process group and the C ABI's communicator)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip extras.configs (configs[0], [2], [4] of BASELINE.json after the timed region)")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:    # no launcher: become one (before stdout is set aside)
        if not args.single_device and torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"--gpus {args.gpus}: this node shows {torch.cuda.device_count()} GPU(s)")
        self_launch(args.gpus)                              # does not return
    claim_stdout()
    if args.config == "c3":
        if args.steps == 200 and args.warmup == 50:
            args.steps, args.warmup = 10, 3
        return main_c3(args)

    world, rank, local_rank, dev, coll_dev, launch_note = init_ranks(args)

    m = args.motif_len
    rows = -(-args.length // COLS)            # striped rows owned by this rank
    total_rows = rows * world
    total_length = args.length * world
    row0 = rows * rank
    pssm = synth_pssm(m)

    # --- resident inputs (SplitMix64 stream, seed 0x5EED0001, 2 bits per base) -------
    shard = synth_shard(rows, row0, total_rows, total_length, m - 1, dev)
    D.exchange_halo(shard, m - 1, COLS, 4)              # RCCL all_gather of (M-1) x 32 bytes per rank
    torch.cuda.synchronize()
    # self-check of the hand-over: the halo the exchange delivered == the same rows drawn straight from the
    # generator (the successor's first M-1 rows; on the last rank rank 0's rows as wrap rows, seq.rs:373-378)
    if rank < world - 1:
        want_halo = synth_shard(m - 1, row0 + rows, total_rows, total_length, 0, dev)
    else:
        head = synth_shard(m - 1, 0, total_rows, total_length, 0, dev)
        want_halo = torch.full_like(head, 4)
        want_halo[:, :COLS - 1] = head[:, 1:]
    halo_ok = bool(torch.equal(shard[rows:], want_halo)) if m > 1 else True
    ident = device_identity(dev)
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, (ident, halo_ok))
        devices, halo_ok = [g[0] for g in gathered], all(g[1] for g in gathered)
    else:
        devices = [ident]
    if not halo_ok:
        raise SystemExit("halo exchange delivered rows that differ from the generator's: refusing to time a wrong job")

    stream = torch.cuda.current_stream()
    pli = lm.Pipeline.hip(local_rank, stream=stream.cuda_stream)
    if args.rows_per_stream:
        pli.set_rows_per_stream(args.rows_per_stream)
    seq = pli.adopt_sequence(shard.data_ptr(), rows, m - 1, COLS, COLS, total_length)  # borrowed device matrix
    scores_h = lm.StripedScores.empty(pli, COLS)
    pli.score_into(pssm, seq, scores_h)                  # allocates the resident StripedScores
    torch.cuda.synchronize()

    # (--merge cabi on ONE rank runs the sharded step through a communicator of one: a functional check of
    #  the C-ABI merge path on a single-GPU box, not the headline configuration)
    use_cabi = args.merge == "cabi" or (world > 1 and args.merge == "auto" and args.dist_backend == "nccl")
    comm, comm_note = None, None
    if use_cabi:
        try:
            comm = D.CabiComm.from_torch(pli, device=coll_dev, deadline_s=args.comm_timeout_s)
        except lm.LightmotifHipError as e:      # e.g. no librccl next to a non-torch host: say so, use torch's
            comm_note = f"C-ABI communicator unavailable ({e}); merge carried by torch.distributed"
        except TimeoutError as e:               # a rank never arrived in ncclCommInitRank: a helper thread is stuck there
            comm_note = f"C-ABI communicator not created ({e}); merge carried by torch.distributed"
            globals()["_HARD_EXIT"] = True
        if world > 1:
            flag = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=coll_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)   # all ranks or none
            if int(flag.item()) == 0 and comm is not None:
                comm.close()
                comm = None
    sharded = world > 1 or comm is not None
    rccl_ranks = comm.info()[1] if comm is not None else None     # what RCCL itself was initialised with
    scores_h.set_first_cell_rule(rank == 0)
    pli.set_track_argmax(sharded)              # N = 1 times the plain store kernel (configs[1])

    # merge transport of the sharded step: 2 = C ABI, pipelined; 1 = C ABI, one wait per step; 0 = torch.distributed.
    # Chosen by a probe of three steps before the preheat: a transport that raises on any rank is dropped on all.
    mode = {"v": 2 if (comm is not None and not args.sync_merge) else 1 if comm is not None else 0}

    def run_steps(n, events=None, lat=None):
        """N = 1: one score_into (pli/mod.rs:109-117) into the resident StripedScores per step.
        N > 1 (configs[3]): the same on this rank's row shard, plus the argmax of the shard
        (tracked by the store kernel, first-cell rule on rank 0 only) and its merge over RCCL --
        SURVEY 8(d): "wall time of the slowest rank incl. the RCCL merge".  Through the C ABI's
        communicator the merge is pipelined: lm_hip_argmax_sharded_begin enqueues record +
        all_gather + read-back (the last two on the communicator's own stream) and the host
        collects step i's result after enqueueing step i + 1, so the all_gather and the host's
        wait overlap the next store kernel; every step's result is collected before this returns."""
        merged, pending = None, None
        for i in range(n):
            if events is not None:
                events[i][0].record(stream)
            pli.score_into(pssm, seq, scores_h)
            if events is not None:
                events[i][1].record(stream)
            if not sharded:
                continue
            tm = time.perf_counter()
            if mode["v"] == 2:
                ticket = comm.argmax_sharded_begin(scores_h, row0)
                if pending is not None:
                    merged = comm.argmax_sharded_end(pending[0])
                    if lat is not None:     # from the merge's begin to its result on the host (the next step's enqueue lies inside)
                        lat.append(time.perf_counter() - pending[1])
                pending = (ticket, tm)
            elif mode["v"] == 1:
                merged = comm.argmax_sharded(scores_h, row0)   # device-side all_gather + combine, one read-back
                if lat is not None:
                    lat.append(time.perf_counter() - tm)
            else:
                loc = pli.argmax_handle_shard(scores_h, first_cell_rule=rank == 0)
                merged = D.merge_argmax(loc, row0, device=coll_dev)
                if lat is not None:
                    lat.append(time.perf_counter() - tm)
        if pending is not None:
            merged = comm.argmax_sharded_end(pending[0])
            if lat is not None:
                lat.append(time.perf_counter() - pending[1])
        return merged

    def barrier() -> None:
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    while sharded and mode["v"] > 0:            # probe the transport (identical failures on every rank are the expected kind)
        try:
            run_steps(3)
            ok = 1
        except lm.LightmotifHipError as e:
            ok = 0
            comm_note = f"C-ABI merge (mode {mode['v']}) failed in the probe ({e}); degraded"
        if world > 1:
            flag = torch.tensor([ok], dtype=torch.int32, device=coll_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item())
        if ok:
            break
        mode["v"] -= 1
    pipelined = mode["v"] == 2

    # time-based preheat (outside the counted warm-up), then the W counted warm-up steps
    t_pre = time.perf_counter()
    n_pre = 0
    while (time.perf_counter() - t_pre) * 1e3 < args.preheat_ms:
        for _ in range(8):
            pli.score_into(pssm, seq, scores_h)
        torch.cuda.synchronize()
        n_pre += 8
    preheat_ms = (time.perf_counter() - t_pre) * 1e3
    run_steps(args.warmup)
    barrier()
    kernel_name = pli.last_kernel

    # --- timed region: exactly K steps ---------------------------------------------------
    # HIP events on the launch stream bracket the score kernel of every step (the context
    # enqueues on torch's current stream, so torch.cuda.Event sees it)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    merge_lat = []
    barrier()
    t0 = time.perf_counter()
    merged = run_steps(args.steps, ev, merge_lat)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = [a.elapsed_time(b) for a, b in ev]
    kernel_avg_ms = float(np.mean(kernel_ms))
    kernel_med_ms = float(np.median(kernel_ms))
    # every rank's own kernel time and merge latency (the line reports the spread, and the aggregate rate as a sum)
    mine = {"kernel_avg_ms": kernel_avg_ms,
            "merge_us": [float(np.median(merge_lat)) * 1e6, float(np.max(merge_lat)) * 1e6] if merge_lat else None}
    if world > 1:
        by_rank = [None] * world
        dist.all_gather_object(by_rank, mine)
    else:
        by_rank = [mine]

    # --- reductions / merge on the last step's matrix (outside the timed region; "extras") --------
    def timed(fn, reps=5, warm=0):
        best, out = None, None
        # the fused scans are quoted in steady state: the first ~60 calls of a process (or behind a pause of the device)
        # take 0.30-0.33 ms where later ones take 0.27 -- host-side pools filling, then the clock governor
        # (tools/scan_kernel_timer_check.py, profiles/r05_fused_call_warmup.txt)
        for _ in range(warm):
            fn()
        for _ in range(reps):
            torch.cuda.synchronize()
            t = time.perf_counter()
            out = fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) * 1e3
            best = dt if best is None else min(best, dt)
        return best, out

    # The shader clock each kind of call sustains: a second thread runs lm_hip_device_clock_mhz windows (one mostly-sleeping
    # wavefront on a stream of its own) while this one repeats the call, after the timed region.  The part clocks to its power budget, so every LDS / VALU fraction below is also
    # quoted at this clock, not only at the 2.4 GHz of the data sheet.
    def sustained_clock(fn, seconds=0.25):
        mhz, beside_ms, alone_ms = pli.sustained_clock_mhz(fn, seconds)
        # a probe that slows what it measures reports the clock of something else: dropped (None) beyond 15 %
        return (mhz if mhz and beside_ms <= 1.15 * alone_ms else None), beside_ms

    store_mhz, store_clock_ms = sustained_clock(lambda: pli.score_into(pssm, seq, scores_h))

    sc_ptr = scores_h.data_ptr
    am_ms, am = timed(lambda: pli.argmax_dptr(sc_ptr, rows, COLS, COLS, first_cell_rule=rank == 0))
    fam_ms, fam = timed(lambda: pli.score_argmax_dptr(pssm, shard.data_ptr(), rows + m - 1, COLS, COLS,
                                                      m - 1, total_length, 0, rows, first_cell_rule=rank == 0), warm=60)
    fam_kernel, fam_info = pli.last_kernel, pli.last_scan_info
    assert am == fam, (am, fam)
    mg_ms, best = timed(lambda: D.merge_argmax(am, row0, device=coll_dev))
    if merged is not None:
        assert merged == best, ("per-step merge differs from the merge of the materialised argmax", merged, best)
    # threshold ~ the p = 1e-5 tail the CLI defaults to (main.rs:487): estimated from a sample
    sample = torch.from_numpy(scores_h.rows_matrix(0, min(rows, 1 << 18))[:, :COLS].reshape(-1))
    thr_t = float(torch.quantile(sample[torch.isfinite(sample)][:8_000_000].float(), 1 - 1e-5))
    if world > 1:   # one threshold for the whole job: rank 0's estimate
        tt = torch.tensor([thr_t], dtype=torch.float64, device=coll_dev)
        dist.broadcast(tt, src=0)
        thr_t = float(tt.item())
    th_ms, hits = timed(lambda: pli.threshold_dptr(sc_ptr, rows, COLS, COLS, thr_t), reps=5)
    fth_ms, fhits = timed(lambda: pli.score_threshold_dptr(pssm, shard.data_ptr(), rows + m - 1, COLS, COLS,
                                                          m - 1, total_length, 0, rows, thr_t), reps=5, warm=60)
    fth_kernel, fth_info = pli.last_kernel, pli.last_scan_info
    assert np.array_equal(hits, fhits[0]), "fused threshold differs from materialised threshold"
    fam_kms = scan_kernel_ms(pli, lambda: pli.score_argmax_dptr(pssm, shard.data_ptr(), rows + m - 1, COLS, COLS, m - 1,
                                                                 total_length, 0, rows))
    fth_phases = {}
    fth_kms = scan_kernel_ms(pli, lambda: pli.score_threshold_dptr(pssm, shard.data_ptr(), rows + m - 1, COLS, COLS, m - 1,
                                                                    total_length, 0, rows, thr_t), phases=fth_phases)
    fam_mhz, _ = sustained_clock(lambda: pli.score_argmax_dptr(pssm, shard.data_ptr(), rows + m - 1, COLS, COLS, m - 1,
                                                               total_length, 0, rows, first_cell_rule=rank == 0))
    fth_mhz, _ = sustained_clock(lambda: pli.score_threshold_dptr(pssm, shard.data_ptr(), rows + m - 1, COLS, COLS, m - 1,
                                                                  total_length, 0, rows, thr_t))
    # Scanner::max as the reference walks it (scan.rs:200-249; csrc/scanmax.hip), a fresh scanner at the same threshold
    smax_ms, smax = timed(lambda: lm.Scanner(pssm, seq, threshold=thr_t).max(), reps=5, warm=3)
    smax_kernel = pli.last_kernel
    mt_ms, all_hits = timed(lambda: (comm.merge_threshold(hits, row0) if comm is not None else
                                     D.merge_threshold(hits, row0, device=coll_dev)), reps=3)
    # the same list through the other transport (one timed merge per variant of the step)
    mt_torch_ms, all_hits_t = (timed(lambda: D.merge_threshold(hits, row0, device=coll_dev), reps=3)
                               if comm is not None else (mt_ms, all_hits))
    assert np.array_equal(np.asarray(all_hits), np.asarray(all_hits_t)), "merge_threshold: the two transports disagree"
    # configs[2] with the motif list sharded over the ranks (every rank takes part)
    c3_sharded = (c3_leg(pli, dev, coll_dev, world, rank, reps=3)
                  if world > 1 and not args.no_extras and (ROOT / "tests" / "golden" / "JASPAR2024.pwm.gz").exists() else None)

    if rank != 0:
        if comm is not None:
            comm.close()
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    positions = rows * COLS * world * args.steps
    value = positions / elapsed / 1e9
    achieved = BYTES_PER_POS * rows * COLS / (kernel_avg_ms * 1e-3) / 1e9
    rank_kernel_ms = [r["kernel_avg_ms"] for r in by_rank]
    achieved_aggregate = sum(BYTES_PER_POS * rows * COLS / (k * 1e-3) / 1e9 for k in rank_kernel_ms)
    rank_merge = [r["merge_us"] for r in by_rank if r["merge_us"]]
    # a line that does not describe N ranks on N devices over RCCL must not pass for one (exit status 3 below)
    invalid = []
    if world > 1 and len(set(devices)) != world and (not args.single_device or args.require_distinct_devices):
        invalid.append(f"{world} ranks on {len(set(devices))} distinct device(s)")
    if world > 1 and args.dist_backend == "nccl" and mode["v"] > 0 and rccl_ranks != world:
        invalid.append(f"the C-ABI communicator was initialised with {rccl_ranks} rank(s), world is {world}")
    lds_bytes_per_s = 4 * m * rows * COLS / (kernel_avg_ms * 1e-3)
    traffic, traffic_source, traffic_current = None, None, None
    pmc = ROOT / "profiles" / "pmc_traffic.json"
    if pmc.exists():
        try:
            j = json.loads(pmc.read_text())
            # PMC counters come from separate rocprofv3 passes over this workload
            # (profiles/README.md); only quoted for the launch shape they were taken on
            if (j.get("algorithmic_bytes_per_launch") == BYTES_PER_POS * rows * COLS and m == 20
                    and not args.rows_per_stream):
                traffic = j.get("hbm_bytes_per_launch")
                traffic_current = j.get("store_kernel_digest") == store_kernel_digest()
                traffic_source = "offline PMC: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this " \
                                 "command (profiles/pmc_traffic.json), not measured by this run"
        except (OSError, ValueError):
            traffic = None
    step_desc = ("score_into" if not sharded else
                 "score_into of the rank's row shard + tracked shard argmax + RCCL merge of the argmax records "
                 f"({'C-ABI communicator' if mode['v'] > 0 else 'torch.distributed ' + args.dist_backend}"
                 + ("; pipelined: the merge of step i overlaps the scoring of step i+1, every step's result is "
                    "collected on the host inside the timed region" if pipelined else "") + ")")
    out = {
        "metric": "scored positions/sec", "value": round(value, 2), "unit": "Gpos/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": f"score(): len-{m} DNA PSSM x {args.length} bp striped sequence per GPU "
                        f"(C=32, K=5, {rows} rows + {m - 1} halo rows), scores materialised in HBM; "
                        f"step = {step_desc}",
            "positions_per_gpu": rows * COLS, "motif_len": m, "parallelism": f"row-shard x{world}",
            "devices": devices, "distinct_devices": len(set(devices)), "halo_verified": halo_ok,
            "merge_transport": (None if not sharded else
                                ["torch.distributed:" + args.dist_backend, "C-ABI communicator (lm_hip_argmax_sharded)",
                                 "C-ABI communicator, pipelined (lm_hip_argmax_sharded_begin/_end)"][mode["v"]]),
            "rccl_ranks": rccl_ranks if mode["v"] > 0 else (world if world > 1 and args.dist_backend == "nccl" else None),
            "process_group": None if world == 1 else f"{dist.get_backend()} x{dist.get_world_size()}",
            **({"launch_note": launch_note} if launch_note else {}),
            "inputs": "SplitMix64 stream seed 0x5EED0001, 2 bits/base (SURVEY 8d), PSSM seed 0x5EED0002",
            "preheat_ms": round(preheat_ms, 1), "preheat_launches": n_pre,
            **({"merge_note": comm_note} if comm_note else {}),
        },
        "roofline": {
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source, "traffic_current": traffic_current,
            "achieved_aggregate": round(achieved_aggregate, 1), "peak_aggregate": HBM_PEAK_GBS * world,
            "kernel_ms_by_rank": [round(min(rank_kernel_ms), 4), round(max(rank_kernel_ms), 4)],
            "kernel": kernel_name, "kernel_avg_ms": round(kernel_avg_ms, 4),
            "kernel_median_ms": round(kernel_med_ms, 4),
            "frac_at_median": round(BYTES_PER_POS * rows * COLS / (kernel_med_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "algorithmic_bytes_per_launch": BYTES_PER_POS * rows * COLS,
            "read_gbs": round(achieved / 5, 1), "write_gbs": round(achieved * 4 / 5, 1),
            "frac_of_measured_copy": round(achieved / 6290.0, 4),   # MI355X_MICROARCH.md: best measured copy 6.29 TB/s
            "lds_frac": round(lds_bytes_per_s / LDS_PEAK_BYTES_PER_S, 4),
            "lds_note": f"secondary ceiling: {4 * m} B of LDS gathers per position against 256 B/clk/CU x 256 CUs x 2.4 GHz",
            **at_sustained_clock(store_mhz, lds_bytes_per_s, m * rows * COLS / (kernel_avg_ms * 1e-3),
                                 f"{m} v_add_f32 per position (the algorithm's adds alone)"),
            "clock_pass_ms_per_launch": round(store_clock_ms, 4),
        },
        "extras": {
            "argmax_ms": round(am_ms, 4), "fused_score_argmax_ms": round(fam_ms, 4),
            "fused_score_argmax_gpos": round(rows * COLS / fam_ms / 1e6, 1),
            "merge_argmax_ms": round(mg_ms, 4), "threshold_ms": round(th_ms, 4),
            "fused_score_threshold_ms": round(fth_ms, 4), "threshold_t": round(thr_t, 4),
            "merge_threshold_ms": round(mt_ms, 4),
            "scanner_max_ms": round(smax_ms, 4), "scanner_max_kernel": smax_kernel,
            "scanner_max": None if smax is None else [int(smax.position), round(float(smax.score), 4)],
            "threshold_hits": int(len(all_hits)), "argmax_global": [int(best[0][0]), int(best[0][1])],
            "kernel_ms_min": round(min(kernel_ms), 4), "kernel_ms_max": round(max(kernel_ms), 4),
            # the fused forms never write the score matrix -- priced against the LDS-gather ceiling (the pair
            # table's (M | 3) + 1 bytes per position), the 1 B per position of HBM reads beside it
            "fused_score_argmax": fused_roofline(fam_ms, rows, m, fam_kernel, fam_mhz, fam_kms, fam_info),
            "fused_score_threshold": {**fused_roofline(fth_ms, rows, m, fth_kernel, fth_mhz, fth_kms, fth_info),
                                      **({"phases": fth_phases} if fth_phases else {})},
            "merge_threshold_ms_torch": round(mt_torch_ms, 4),
            "merge_us": (None if not rank_merge else
                         {"p50": round(float(np.median([x[0] for x in rank_merge])), 1),
                          "max": round(max(x[1] for x in rank_merge), 1),
                          "what": "per timed step, host clock: from the merge's begin to the merged argmax on the host (the wait for "
                                  "the shard's own store kernel, whose tracked record is what gets merged, lies inside); "
                                  "median of the ranks' medians, maximum over ranks and steps"
                                  + ("; pipelined -- the next step's enqueue lies inside" if pipelined else "")}),
        },
    }
    if invalid:
        out["invalid"] = invalid
    if world == 1 and not args.no_extras:
        out["extras"]["end_to_end"] = end_to_end(shard, rows, m, total_length, pssm, scores_h, dev)
        out["extras"]["configs"] = secondary_configs(pli, dev)
        out["extras"]["crossover_positions"] = host_pointer_bench().crossover_positions()
    elif c3_sharded is not None:
        out["extras"]["configs"] = {"c3": c3_sharded}
    if not args.no_cpu_baseline:
        # rank 0's host cores, whatever N (the other ranks wait at the closing barrier): the sample is the head of
        # rank 0's shard; a window that reaches the shard's end needs the halo rows the exchange delivered
        srows = min(rows, max(args.cpu_sample // COLS, 1))
        seq_sample = shard[:srows + m - 1].cpu().numpy()
        gpu_sample = scores_h.rows_matrix(0, srows)
        out["cpu_baseline"] = cpu_baseline(seq_sample, total_length, pssm.data, gpu_sample,
                                           args.cpu_seconds)
    else:
        out["cpu_baseline"] = None
    emit(out)
    if comm is not None:
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if invalid:
        print("bench.py: " + "; ".join(invalid), file=sys.stderr, flush=True)
        sys.stdout.flush()
        os._exit(3)


if __name__ == "__main__":
    main()
    if _HARD_EXIT:
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)
