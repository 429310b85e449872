#!/usr/bin/env python3
"""Busy fractions per kernel from a `tools/collect_stalls.sh` summary (gpurun_out/stalls_<tag>/summary.txt):
VALU busy = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x cycles), LDS busy = SQ_LDS_IDX_ACTIVE / (256 CUs x cycles),
cycles = GRBM_GUI_ACTIVE / 8 XCDs, MHz = cycles / duration (the clock under counter collection),
waitLDS = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES.
    python tools/stalls_table.py gpurun_out/stalls_r06_protein/summary.txt [name filter]"""
import re
import sys


def parse(path):
    out, cur = {}, None
    for line in open(path):
        if not line.startswith(" "):
            cur = out.setdefault(line.strip(), {})
            continue
        m = re.match(r"\s+(\S.*?)\s+([0-9.e+]+)\s+\(n=(\d+)\)", line)
        if m and cur is not None:
            cur[m.group(1)] = (float(m.group(2)), int(m.group(3)))
    return out


def main():
    kernels = parse(sys.argv[1])
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    print("kernel | launches | avg us | VALU busy | LDS busy | bankconf | VALU/LDS | MHz | waitLDS/wavecyc | HBM read requests")
    for name, c in kernels.items():
        if flt not in name or "GRBM_GUI_ACTIVE" not in c or "duration_ns(all passes)" not in c:
            continue
        g = lambda k: c.get(k, (0.0, 0))[0]  # noqa: E731
        cycles = g("GRBM_GUI_ACTIVE") / 8
        dur_us = g("duration_ns(all passes)") / 1e3
        if cycles <= 0 or dur_us <= 0:
            continue
        valu = g("SQ_INSTS_VALU") * 4 / (1024 * cycles)
        lds = g("SQ_LDS_IDX_ACTIVE") / (256 * cycles)
        bank = g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1)
        ratio = g("SQ_INSTS_VALU") / max(g("SQ_INSTS_LDS"), 1)
        wait = g("SQ_WAIT_INST_LDS") / max(g("SQ_WAVE_CYCLES"), 1)
        print(f"{name[:60]:60s} {c['GRBM_GUI_ACTIVE'][1]:4d} {dur_us:9.1f} {valu:5.2f} {lds:5.2f} {bank:6.3f} {ratio:6.2f} {cycles / dur_us:6.0f} {wait:5.2f} {g('TCC_EA0_RDREQ_sum'):12.0f}")


if __name__ == "__main__":
    main()
