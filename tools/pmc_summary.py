#!/usr/bin/env python
"""Summarise rocprofv3 --pmc result databases (one per counter group) into text + the regress traffic json.
usage: python tools/pmc_summary.py OUT_TXT OUT_JSON MODE db1 db2 ...   (or: ... MODE OUT_TXT to rebuild the json record from the text)
OUT_JSON is keyed by regressor mode ({"fp16x2": {...}, "f32": {...}}); an existing file is updated.  Every record carries
the hash of the kernel sources it was measured with; bench.py reports `roofline.traffic` only when it matches."""
import json
import os
import sqlite3
import sys


def rows_from_txt(path):
    """The per-kernel averages of a summary written by this script (profiles/*_pmc.txt) back into {kernel: {counter: (n, avg, ns)}}."""
    rows, cur = {}, None
    for line in open(path):
        if line.startswith("#") or not line.strip():
            continue
        if not line.startswith(" "):
            cur = rows.setdefault(line.strip(), {})
        else:
            name, n, avg, dur = line.split()[0], line.split("n=")[1].split("avg=")[0], line.split("avg=")[1].split()[0], line.split("avg_duration_ns=")[1]
            cur[name] = (int(n), float(avg), float(dur))
    return rows


def main(out_txt, out_json, mode, dbs):
    rows = {}
    if len(dbs) == 1 and dbs[0].endswith(".txt"):      # re-derive the json record from a committed summary (same numbers)
        rows = rows_from_txt(dbs[0])
        dbs = []
    for path in dbs:
        c = sqlite3.connect(path)
        q = ("select kernel_name, counter_name, count(*), avg(value), avg(end-start) from counters_collection "
             "group by kernel_name, counter_name")
        for name, counter, n, val, dur in c.execute(q):
            rows.setdefault(name.split("(")[0], {})[counter] = (n, val, dur)
    lines = [f"# rocprofv3 --pmc <counters> --kernel-trace -- python bench.py --config {os.environ.get('P2P_CONFIG', 'A')} --mode {mode} --steps 3 --warmup 1 (one pass per group)",
             "# per-dispatch averages; FETCH_SIZE/WRITE_SIZE in KiB as reported; duration in ns; SQ_* wave counters in quad-cycles",
             "# gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE under-reports wide coalesced reads by 2x -> doubled in the json"]
    for k, v in rows.items():
        if "p2p::" in k:
            lines.append(k)
            for cn, (n, val, dur) in sorted(v.items()):
                lines.append(f"    {cn:32s} n={n:4d} avg={val:.5g} avg_duration_ns={dur:.0f}")
    if dbs:
        open(out_txt, "w").write("\n".join(lines) + "\n")
    for k, rg in rows.items():
        if "regress" in k and "FETCH_SIZE" in rg and "GRBM_GUI_ACTIVE" in rg:
            fetch = rg["FETCH_SIZE"][1] * 1024 * 2
            write = rg.get("WRITE_SIZE", (0, 0, 0))[1] * 1024
            act = rg["GRBM_GUI_ACTIVE"]
            clk = act[1] / 8 / (act[2] * 1e-9) / 1e9
            mf = rg["SQ_VALU_MFMA_BUSY_CYCLES"][1] / 1024 / (act[1] / 8) if "SQ_VALU_MFMA_BUSY_CYCLES" in rg else None
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            import bench
            config = os.environ.get("P2P_CONFIG", "A")
            cfg = bench.CONFIGS[config]
            pairs = int(os.environ.get("P2P_PAIRS_PER_STEP", cfg["pairs_per_step"]))
            per_pair = cfg["ptmax"] * cfg["panc"]
            allrec = json.load(open(out_json)) if os.path.exists(out_json) else {}
            if "kernel" in allrec:                                        # round-1 format (one un-keyed record)
                allrec = {}
            lds = None
            if "SQ_LDS_BANK_CONFLICT" in rg and "SQ_LDS_IDX_ACTIVE" in rg and rg["SQ_LDS_IDX_ACTIVE"][1] > 0:
                lds = rg["SQ_LDS_BANK_CONFLICT"][1] / rg["SQ_LDS_IDX_ACTIVE"][1]
            allrec[mode if config == "A" else f"{mode}@{config}"] = {
                            "kernel": k.split("::")[-1], "proposals_per_launch": pairs * per_pair, "config": config,
                            "launch": f"{pairs * per_pair} proposals ({pairs} pairs x {per_pair}), 2 levels",
                            "hbm_bytes_per_launch": fetch + write, "fetch_bytes_corrected_x2": fetch, "write_bytes": write,
                            "effective_clock_ghz": clk, "mfma_busy_fraction": mf, "lds_bank_conflict_fraction": lds,
                            "source_hash": bench.source_hash(), "source": "profiles/" + os.path.basename(out_txt)}
            json.dump(allrec, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4:])
