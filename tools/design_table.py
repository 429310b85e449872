#!/usr/bin/env python3
"""Regenerates the numbers tables of DESIGN.md (section 8) and README.md from ONE bench line, so that the documents quote what the
driver measures and nothing else:

    python tools/design_table.py [profiles/r06_bench_default.json] [--write]

Without --write the tables are printed; with it the regions between `<!-- bench-table:begin -->` / `<!-- bench-table:end -->`
in DESIGN.md and README.md are replaced.  Every row names the JSON field it comes from."""
import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def load(path: Path) -> dict:
    lines = [l for l in path.read_text().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


def rows_of(d: dict):
    e = d.get("extras", {})
    c = e.get("configs", {})
    r = d["roofline"]
    out = []
    out.append(("`score_c32<20,0>` store, 1 Gbp x M = 20 (headline)", f"{r.get('kernel_avg_ms', d['ms_per_step']):.4f} ms per launch (events, average; median {r.get('kernel_median_ms')}); step {d['ms_per_step']:.4f} ms = {d['value']:.0f} Gpos/s",
                "HBM, 5 B per position", f"**{r['frac']:.3f}** of 8 TB/s; counter traffic {r.get('traffic') and round(r['traffic'] / 5e9, 3)} x algorithmic"
                + (f"; sustained clock {r.get('sclk_mhz_sustained')} MHz: LDS {r.get('lds_frac_at_sustained_clock')}, VALU {r.get('valu_frac_at_sustained_clock')}" if r.get("sclk_mhz_sustained") else ""),
                "`roofline`"))
    for key, name in (("fused_score_threshold", "fused score + threshold call (p = 1e-5), 1 Gbp"), ("fused_score_argmax", "fused score + argmax call, 1 Gbp")):
        f = e.get(key)
        if not f:
            continue
        rf = f["roofline"]
        ph = f.get("phases")
        out.append((name + f" (`{f['kernel']}`)", f"call {f['ms']:.4f} ms; scan kernel {rf.get('kernel_ms')} ms; tail {f.get('tail_us')} us"
                    + (f" (re-score {ph['rescore_us']} + order {ph['order_us']} + host {ph['host_us']} us)" if ph else ""),
                    f"LDS gather, {rf['lds_bytes_per_position']} B per position ({rf['motif_rows_scanned']} rows scanned)",
                    f"call **{rf['frac']:.3f}**, kernel **{rf.get('kernel_frac')}** of the LDS ceiling; {rf['hbm_read_frac']:.3f} of HBM reads",
                    f"`extras.{key}`"))
    if "scanner_max_ms" in e:
        out.append(("`Scanner::max` (scan.rs:200-249), 1 Gbp, p = 1e-5", f"{e['scanner_max_ms']:.3f} ms (`{e.get('scanner_max_kernel')}`)", "one flag scan + the walk of a short list", "—",
                    "`extras.scanner_max_ms`"))
    out.append(("`argmax` / `threshold` on the stored matrix, 1 Gbp", f"{e.get('argmax_ms')} / {e.get('threshold_ms')} ms", "HBM, 4 B per position",
                f"{4e9 / (e['argmax_ms'] * 1e-3) / 8e12:.2f} / {4e9 / (e['threshold_ms'] * 1e-3) / 8e12:.2f} of 8 TB/s (calls)" if e.get("argmax_ms") else "—", "`extras.argmax_ms`, `threshold_ms`"))
    c5 = c.get("c5")
    if c5:
        out.append(("protein `score_c32<12,0>` WIDE store, 200 Mres (configs[4])", f"{c5['kernel_ms']} ms", "HBM, 5 B per residue", f"**{c5['roofline']['frac']:.3f}**", "`extras.configs.c5`"))
        ft = c5.get("fused_threshold")
        if ft:
            rf = ft["roofline"]
            ph = ft.get("phases")
            out.append((f"protein fused threshold call (`{ft['kernel']}`), 200 Mres", f"call {ft['ms']:.4f} ms; scan kernel {rf.get('kernel_ms')} ms; tail {ft.get('tail_us')} us"
                        + (f" (re-score {ph['rescore_us']} + order {ph['order_us']} + host {ph['host_us']} us)" if ph else ""),
                        f"LDS gather, {rf['lds_bytes_per_position']} B per residue", f"call **{rf['frac']:.3f}**, kernel **{rf.get('kernel_frac')}**", "`extras.configs.c5.fused_threshold`"))
    c3 = c.get("c3")
    if c3:
        ph = c3.get("phases")
        out.append(("JASPAR 2024 batch, 2 346 motifs x 100 Mbp, threshold at p = 1e-5 (configs[2])", f"{c3['fused_threshold_ms']} ms" + (f" (scan {ph['scan_ms']} + re-score {ph['rescore_ms']} + order {ph['order_ms']} + host {ph['host_ms']} ms)" if ph else "")
                    + f"; argmax batch {c3['fused_argmax_ms']} ms", "LDS gather of the pair tables", f"call **{c3['roofline']['frac']:.3f}**" + (f", scan kernels {ph['scan_frac_of_lds_ceiling']}" if ph else "")
                    + (f"; non-i.i.d. input {c3['realistic']['threshold_ms_over_uniform']} x" if c3.get("realistic") else ""), "`extras.configs.c3`"))
    c1 = c.get("c1")
    if c1:
        g32 = c1.get("C32_dispatch_geometry", {})
        out.append(("the reference's bench (dna.rs:81-116: `score_into` + `argmax`, 464 165 bp)", f"{g32.get('us_per_iter')} us on handles; {c1.get('host_pointer_us_per_iter')} us on host matrices" + (f" ({c1['host_pointer_reuse_us_per_iter']} with `lm_hip_host_reuse_scores`)" if 'host_pointer_reuse_us_per_iter' in c1 else "") + f"; AVX2 port, one core: {c1.get('avx2_port_1_thread_us_per_iter')} us",
                    "launch + link latency", "—", "`extras.configs.c1`"))
    ee = e.get("end_to_end")
    if ee:
        out.append(("host-pointer `lm_hip_score_f32`, 1 Gbp (pageable host matrices)", f"{ee.get('host_pointer_1gbp_ms')} ms = {ee.get('gpos')} Gpos/s", "PCIe D2H, 4 B per position", f"{ee.get('frac_of_d2h_floor')} of the pinned-copy floor",
                    "`extras.end_to_end`"))
    cb = d.get("cpu_baseline")
    if cb:
        out.append(("CPU beside it: AVX2 port of the reference kernel", f"{cb['value']} {cb['unit']} on {cb['cores']} threads; one thread {cb.get('single_thread_gpos')}; Generic {cb.get('generic_single_thread_gpos')}", "—", "—", "`cpu_baseline`"))
    return out


def table(d: dict) -> str:
    lines = ["| kernel / call | time | bound | of the bound | bench field |", "|---|---|---|---|---|"]
    for r in rows_of(d):
        lines.append("| " + " | ".join(str(x) for x in r) + " |")
    return "\n".join(lines)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    path = Path(args[0]) if args else ROOT / "profiles" / "r06_bench_default.json"
    d = load(path)
    t = f"(generated by `tools/design_table.py {path.relative_to(ROOT) if path.is_absolute() and ROOT in path.parents else path}`; one MI355X, N = 1)\n\n" + table(d)
    if "--write" not in sys.argv:
        print(t)
        return
    for doc in ("DESIGN.md", "README.md"):
        p = ROOT / doc
        s = p.read_text()
        new, n = re.subn(r"(<!-- bench-table:begin -->\n).*?(\n<!-- bench-table:end -->)", lambda m: m.group(1) + t + m.group(2), s, flags=re.S)
        if n:
            p.write_text(new)
            print(f"{doc}: table rewritten")


if __name__ == "__main__":
    main()
