#!/usr/bin/env python
"""Per-kernel record of one profiled bench.py command -> profiles/kernel_times.json (read back by bench.py's `coarse_roofline`).
usage: python tools/kernel_times.py OUT_JSON CONFIG SOURCE_NOTE kernel_trace.db [pmc.db ...]
The record is keyed by configuration (A, E) and carries the hash of csrc/ it was measured with; bench.py ignores it when the
sources have changed since.  avg_us from the --kernel-trace pass; MFMA busy, effective clock and LDS conflicts from the --pmc
passes of the same command (formulas of tools/pmc_summary.py)."""
import json
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    m = re.search(r"p2p::([A-Za-z0-9_]+)", name)
    return m.group(1) if m else None


def main(out_json, config, note, trace_db, pmc_dbs):
    sys.path.insert(0, ROOT)
    import bench
    kernels = {}
    c = sqlite3.connect(trace_db)
    for name, calls, total, avg, pct in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        k = short(name)
        if k:
            e = kernels.setdefault(k, {"calls": 0, "total_us": 0.0})
            e["calls"] += calls
            e["total_us"] += total
    for e in kernels.values():
        e["avg_us"] = e["total_us"] / max(e["calls"], 1)
    pmc = {}
    for path in pmc_dbs:
        c = sqlite3.connect(path)
        q = ("select kernel_name, counter_name, count(*), avg(value), avg(end-start) from counters_collection "
             "group by kernel_name, counter_name")
        for name, counter, n, val, dur in c.execute(q):
            k = short(name)
            if k:
                pmc.setdefault(k, {})[counter] = (n, val, dur)
    for k, rg in pmc.items():
        if k not in kernels or "GRBM_GUI_ACTIVE" not in rg:
            continue
        act = rg["GRBM_GUI_ACTIVE"]
        kernels[k]["effective_clock_ghz"] = act[1] / 8 / act[2]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in rg:
            kernels[k]["mfma_busy_fraction"] = rg["SQ_VALU_MFMA_BUSY_CYCLES"][1] / 1024 / (act[1] / 8)
        if "SQ_LDS_BANK_CONFLICT" in rg and rg.get("SQ_LDS_IDX_ACTIVE", (0, 0, 0))[1]:
            kernels[k]["lds_bank_conflict_fraction"] = rg["SQ_LDS_BANK_CONFLICT"][1] / rg["SQ_LDS_IDX_ACTIVE"][1]
        if "FETCH_SIZE" in rg:
            kernels[k]["fetch_bytes_corrected_x2"] = rg["FETCH_SIZE"][1] * 1024 * 2
        if "WRITE_SIZE" in rg:
            kernels[k]["write_bytes"] = rg["WRITE_SIZE"][1] * 1024
    allrec = json.load(open(out_json)) if os.path.exists(out_json) else {}
    allrec[config] = {"config": config, "pairs_per_step": bench.CONFIGS[config]["pairs_per_step"], "source_hash": bench.source_hash(),
                      "source": note, "kernels": kernels}
    json.dump(allrec, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5:])
