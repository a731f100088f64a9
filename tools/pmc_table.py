#!/usr/bin/env python
"""Per-kernel table from rocprofv3 --pmc ... --kernel-trace result databases (any command, any kernels).
usage: python tools/pmc_table.py [--match SUBSTR] db1 [db2 ...]
Derived columns (MI355X_MICROARCH.md): clock = GRBM_GUI_ACTIVE / 8 XCDs / duration; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES /
(1024 SIMDs x GRBM_GUI_ACTIVE / 8); LDS conflicts = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; waiting = SQ_WAIT_ANY /
SQ_WAVE_CYCLES; FETCH_SIZE doubled (gfx950 under-reports wide coalesced reads by 2x), sizes in MB per dispatch."""
import sqlite3
import sys


def main(argv):
    match = None
    if argv and argv[0] == "--match":
        match, argv = argv[1], argv[2:]
    rows = {}
    for path in argv:
        c = sqlite3.connect(path)
        q = ("select kernel_name, counter_name, count(*), avg(value), avg(end-start) from counters_collection "
             "group by kernel_name, counter_name")
        for name, counter, n, val, dur in c.execute(q):
            rows.setdefault(name.split("(")[0], {})[counter] = (n, val, dur)
    print(f"{'kernel':44s} {'n':>5s} {'avg_us':>9s} {'GHz':>5s} {'mfma':>6s} {'ldsconf':>7s} {'wait':>6s} {'fetchMB':>9s} {'writeMB':>9s}")
    for k, r in sorted(rows.items(), key=lambda kv: -max(v[2] * v[0] for v in kv[1].values())):
        if match and match not in k:
            continue
        any_ = next(iter(r.values()))
        act = r.get("GRBM_GUI_ACTIVE")
        dur = act[2] if act else any_[2]
        g = lambda name: r[name][1] if name in r else None
        clk = act[1] / 8 / dur if act else None
        mf = g("SQ_VALU_MFMA_BUSY_CYCLES") / 1024 / (act[1] / 8) if act and g("SQ_VALU_MFMA_BUSY_CYCLES") is not None else None
        lc = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE") if g("SQ_LDS_IDX_ACTIVE") else None
        wt = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES") if g("SQ_WAVE_CYCLES") else None
        fe = g("FETCH_SIZE") * 2 * 1024 / 1e6 if g("FETCH_SIZE") is not None else None
        wr = g("WRITE_SIZE") * 1024 / 1e6 if g("WRITE_SIZE") is not None else None
        f = lambda v, fmt: (fmt % v) if v is not None else "-"
        print(f"{k.split('::')[-1][:44]:44s} {any_[0]:5d} {dur / 1e3:9.1f} {f(clk, '%.2f'):>5s} {f(mf, '%.3f'):>6s} {f(lc, '%.3f'):>7s} "
              f"{f(wt, '%.3f'):>6s} {f(fe, '%.1f'):>9s} {f(wr, '%.1f'):>9s}")


if __name__ == "__main__":
    main(sys.argv[1:])
