"""bench.py as the driver runs it: the N = 1 line, and the N > 1 launch path rehearsed on ONE GPU.

An 8-GPU node is the driver's to use, not the builder's, so the multi-rank control flow of
``bench.py`` (self-launch under torch.distributed.run, halo hand-over checked against the
generator, per-step argmax merge, the JSON fields that say what actually ran) is exercised here
with two ranks that share ``cuda:0`` and merge over gloo -- RCCL refuses two ranks on one
device, everything around the transport is the code the 8-GPU run takes."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def run_bench(*args, timeout=900):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)                                   # no launcher: bench.py must cope on its own
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True,
                       timeout=timeout, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                       # ONE JSON line on stdout, nothing else
    return json.loads(lines[0])


def test_two_ranks_without_a_launcher():
    """`python bench.py --gpus 2 ...` with no WORLD_SIZE in the environment re-executes itself under
    torch.distributed.run and prints one line for a 2-rank job."""
    out = run_bench("--gpus", "2", "--single-device", "--dist-backend", "gloo", "--steps", "3", "--warmup", "1",
                    "--length", "20000000", "--preheat-ms", "20")
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak"
    cfg = out["config"]
    assert cfg["parallelism"] == "row-shard x2" and cfg["halo_verified"] is True
    assert cfg["process_group"] == "gloo x2" and cfg["merge_transport"].startswith("torch.distributed:gloo")
    assert len(cfg["devices"]) == 2 and cfg["distinct_devices"] == 1     # both ranks on cuda:0 in this rehearsal
    assert out["value"] > 0 and out["roofline"]["kernel"].startswith("score_c32<20")
    # the merged argmax of the 40 Mbp job is the same cell a single process finds on the whole sequence
    whole = run_bench("--gpus", "1", "--steps", "2", "--warmup", "1", "--length", "40000000", "--preheat-ms", "20",
                      "--no-cpu-baseline", "--no-extras")
    assert out["extras"]["argmax_global"] == whole["extras"]["argmax_global"]
    assert out["extras"]["threshold_hits"] > 0
    # what the first real N > 1 run will be read by: a CPU leg on rank 0, the per-rank kernel spread, the aggregate rate as
    # a sum over ranks, the merge latency of the timed steps, both merge_threshold transports, the motif-sharded configs[2]
    cb = out["cpu_baseline"]
    assert cb is not None and cb["gpu_matches_cpu_bitwise"] is True and cb["gpu_matches_generic_bitwise"] is True
    assert cb["value"] > 0 and cb["generic_single_thread_gpos"] > 0
    rf = out["roofline"]
    lo, hi = rf["kernel_ms_by_rank"]
    assert 0 < lo <= hi and rf["peak_aggregate"] == 2 * rf["peak"]
    assert rf["achieved"] < rf["achieved_aggregate"] < 4 * rf["achieved"]     # the sum over both ranks (which share ONE device here)
    mu = out["extras"]["merge_us"]
    assert mu["p50"] > 0 and mu["max"] >= mu["p50"]
    assert out["extras"]["merge_threshold_ms"] > 0 and out["extras"]["merge_threshold_ms_torch"] > 0
    for key in ("fused_score_argmax", "fused_score_threshold"):
        fr = out["extras"][key]["roofline"]
        assert fr["bound"] == "lds" and 0 < fr["frac"] < 1.2 and 0 < fr["hbm_read_frac"] < 1
    c3 = out["extras"]["configs"]["c3"]
    assert c3["parallelism"].startswith("motif-shard x2") and len(c3["motifs_per_rank"]) == 2
    assert sum(c3["motifs_per_rank"]) == 2346 and c3["hits_total"] > 0 and c3["roofline"]["bound"] == "lds"
    assert "invalid" not in out


def test_eight_ranks_rehearsed_on_one_device():
    """The first real 8-GPU run must not also be the first 8-rank run: the launcher-less line at --gpus 8 with all ranks
    on cuda:0 over gloo -- 8 row shards + halos, 8 merge records per step, the threshold lists of 8 ranks concatenated,
    and the motif-sharded configs[2] leg (LPT over 2 346 motifs, 8 shares).  NOT a scaling number: one device."""
    out = run_bench("--gpus", "8", "--single-device", "--dist-backend", "gloo", "--steps", "2", "--warmup", "1",
                    "--length", "50000000", "--preheat-ms", "20", "--cpu-seconds", "1", "--cpu-sample", "4000000", timeout=1500)
    assert out["n_gpus"] == 8 and "invalid" not in out
    cfg = out["config"]
    assert cfg["parallelism"] == "row-shard x8" and cfg["halo_verified"] is True and cfg["process_group"] == "gloo x8"
    assert len(cfg["devices"]) == 8 and cfg["distinct_devices"] == 1
    whole = run_bench("--gpus", "1", "--steps", "2", "--warmup", "1", "--length", "400000000", "--preheat-ms", "20",
                      "--no-cpu-baseline", "--no-extras")
    assert out["extras"]["argmax_global"] == whole["extras"]["argmax_global"]     # 8 shards merge to the whole job's cell
    assert out["extras"]["threshold_hits"] > 0
    c3 = out["extras"]["configs"]["c3"]
    assert c3["parallelism"].startswith("motif-shard x8") and len(c3["motifs_per_rank"]) == 8
    assert sum(c3["motifs_per_rank"]) == 2346 and min(c3["motifs_per_rank"]) > 0 and c3["hits_total"] > 0
    one = run_bench("--config", "c3", "--gpus", "1", "--steps", "1", "--warmup", "1", "--no-cpu-baseline")
    assert c3["hits_total"] == one["extras"]["hits_total"]     # 8 shares of the motif list find what one rank finds
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / "bench_8rank_gloo_single_device.json").write_text(json.dumps(
        {"note": "8 ranks on ONE device over gloo: a rehearsal of the control flow, NOT a scaling number", **out}) + "\n")


def test_a_line_that_is_not_n_ranks_on_n_devices_fails_loudly():
    """Two ranks on one device WITHOUT the rehearsal flag's exemption cannot happen by accident (LOCAL_RANK picks the
    device), but a launcher that maps both ranks to one GPU can: the line carries `invalid` and the exit status is 3."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--single-device", "--require-distinct-devices",
                        "--dist-backend", "gloo",
                        "--steps", "2", "--warmup", "1", "--length", "10000000", "--preheat-ms", "20", "--no-cpu-baseline",
                        "--no-extras"], capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
    assert r.returncode != 0
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][0])
    assert line["invalid"] and "distinct device" in line["invalid"][0]


def test_single_gpu_line_has_every_contract_field():
    out = run_bench("--steps", "5", "--warmup", "2", "--length", "50000000", "--preheat-ms", "20", "--cpu-seconds", "2",
                    "--cpu-sample", "16000000")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["dtype"] == "f32" and out["config"]["merge_transport"] is None
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = out["cpu_baseline"]
    assert cb["gpu_matches_cpu_bitwise"] is True and cb["kind"] == "port"
    assert cb["sockets"] >= 1 and cb["cores"] >= 1 and cb["threads"] >= cb["cores"]
    ex = out["extras"]["configs"]                           # configs[0], [2], [4] ride along on the driver's line
    assert ex["c1"]["C32_dispatch_geometry"]["best_position"] == 391_677
    assert ex["c1"]["C1_generic_bench_geometry"]["best_position"] == 391_677
    assert ex["c5"]["kernel"].startswith("score_c32<12") and 0 < ex["c5"]["hbm_frac"] < 1
    assert ex["c3"]["hits_total"] > 0 and ex["c3"]["motifs_skipped_unreachable"] >= 0
    # the literal drop-in path rides along: host-pointer loop of dna.rs, Scanner block, the labelled end-to-end figure
    assert ex["c1"]["host_pointer_us_per_iter"] > 0 and ex["c1"]["avx2_port_1_thread_us_per_iter"] > 0
    assert ex["c1"]["scores_match_avx2_port_bitwise"] is True and ex["c1"]["scanner_block_us"] > 0
    rb = ex["readme_benchmark"]                             # the reference's published benchmark at its own size
    assert rb["host_pointer_matches_avx2_port_bitwise"] is True and 0 < rb["resident_score_into_ms"] < rb["host_pointer_score_f32_ms"]
    e2e = out["extras"]["end_to_end"]
    assert e2e["matches_resident_scores"] is True and 0 < e2e["frac_of_d2h_floor"] <= 1.05 and e2e["gpos"] > 0
    assert cb["gpu_matches_generic_bitwise"] is True and cb["generic_single_thread_gpos"] > 0
    for key in ("fused_score_argmax", "fused_score_threshold"):
        fr = out["extras"][key]["roofline"]
        assert fr["bound"] == "lds" and 0 < fr["frac"] < 1.2 and 0 < fr["hbm_read_frac"] < 1
    assert ex["c3"]["roofline"]["bound"] == "lds" and 0 < ex["c3"]["roofline"]["frac"] < 1.2
    assert ex["c5"]["roofline"]["bound"] == "hbm" and ex["c5"]["roofline"]["frac"] == ex["c5"]["hbm_frac"]
    # round 5: the clock every LDS / VALU fraction is also quoted at, the reference's 10 kb benchmark, the shipped crossovers
    assert 500 < rf["sclk_mhz_sustained"] < 3000 and 0 < rf["lds_frac_at_sustained_clock"] < 1.2 and 0 < rf["valu_frac_at_sustained_clock"] < 1.2
    for key in ("fused_score_argmax", "fused_score_threshold"):
        fr = out["extras"][key]["roofline"]
        assert fr["sclk_mhz_sustained"] is None or 0 < fr["lds_frac_at_sustained_clock"] < 1.5
        # the scan kernel alone, by events inside the library: shorter than the call, and what the ceiling bounds
        # (absent when the call took a route without a scan kernel: small test lengths)
        if "kernel_ms" in fr:
            assert 0 < fr["kernel_ms"] < 1.1 * out["extras"][key]["ms"] and 0 < fr["kernel_frac"] < 1.2
    r10 = ex["readme_10kb"]
    assert r10["published_avx2_us"] == 12.797 and r10["host_pointer_us"] > 0 and r10["avx2_port_us"] > 0
    assert r10["scores_match_avx2_port_bitwise"] is True
    xo = out["extras"]["crossover_positions"]["cells"]
    assert xo["encode"] is None and xo["stripe"] is None and xo["maximum_u8"] is None and xo["threshold_u8"] is None
    assert 10_000 < xo["score_f32"] < xo["score_u8"] and xo["scan"] > 10_000 and xo["maximum_f32"] > 10_000
    assert ex["c1"]["C1_generic_bench_geometry"]["fused_score_argmax_us"] <= 1.25 * ex["c1"]["C1_generic_bench_geometry"]["us_per_iter"]


def test_one_rank_through_the_c_abi_communicator():
    """`--merge cabi` on one rank: the sharded step through the library's own RCCL communicator (a world of one);
    the line reports what RCCL was initialised with."""
    out = run_bench("--merge", "cabi", "--steps", "5", "--warmup", "2", "--length", "50000000", "--preheat-ms", "20",
                    "--no-cpu-baseline", "--no-extras")
    assert out["config"]["rccl_ranks"] == 1 and "C-ABI communicator" in out["config"]["merge_transport"]
