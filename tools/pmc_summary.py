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
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    config = os.environ.get("P2P_CONFIG", "A")
    cfg = bench.CONFIGS[config]
    pairs = int(os.environ.get("P2P_PAIRS_PER_STEP", cfg["pairs_per_step"]))
    per_pair = cfg["ptmax"] * cfg["panc"]
    # the fine stage of a step is one kernel (fp16x2, f32) or a sequence of kernels per regress call (fp16x2w)
    names = bench.MODES[mode].get("kernels", [bench.MODES[mode]["kernel"]])
    found = {}
    for nm in names:
        for k, rg in rows.items():
            if k.split("::")[-1] == nm or (len(names) == 1 and nm in k):
                found[nm] = rg
    if len(found) == len(names) and all("FETCH_SIZE" in rg and "GRBM_GUI_ACTIVE" in rg for rg in found.values()):
        # dispatches of a kernel per regress call: the FC kernel runs once per level (2 per call); a single kernel once
        calls = found[names[-1]]["GRBM_GUI_ACTIVE"][0] / (2.0 if len(names) > 1 else 1.0)
        fetch = write = busy_cycles = gui = dur = conf = ldsact = 0.0
        per_kernel = {}
        for nm, rg in found.items():
            mult = rg["GRBM_GUI_ACTIVE"][0] / calls
            f, w = rg["FETCH_SIZE"][1] * 1024 * 2 * mult, rg.get("WRITE_SIZE", (0, 0, 0))[1] * 1024 * mult
            act = rg["GRBM_GUI_ACTIVE"]
            fetch += f; write += w
            gui += act[1] * mult; dur += act[2] * mult
            busy_cycles += rg.get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0, 0))[1] * mult
            conf += rg.get("SQ_LDS_BANK_CONFLICT", (0, 0, 0))[1] * mult
            ldsact += rg.get("SQ_LDS_IDX_ACTIVE", (0, 0, 0))[1] * mult
            per_kernel[nm] = {"dispatches_per_call": mult, "avg_us": act[2] / 1e3, "fetch_bytes_corrected_x2": f, "write_bytes": w,
                              "effective_clock_ghz": act[1] / 8 / act[2],
                              "mfma_busy_fraction": (rg["SQ_VALU_MFMA_BUSY_CYCLES"][1] / 1024 / (act[1] / 8))
                              if "SQ_VALU_MFMA_BUSY_CYCLES" in rg else None}
        allrec = json.load(open(out_json)) if os.path.exists(out_json) else {}
        if "kernel" in allrec:                                        # round-1 format (one un-keyed record)
            allrec = {}
        allrec[mode if config == "A" else f"{mode}@{config}"] = {
            "kernel": "+".join(names), "proposals_per_launch": pairs * per_pair, "config": config,
            "launch": f"{pairs * per_pair} proposals ({pairs} pairs x {per_pair}), 2 levels",
            "hbm_bytes_per_launch": fetch + write, "fetch_bytes_corrected_x2": fetch, "write_bytes": write,
            "kernel_time_per_call_ms": dur / 1e6, "effective_clock_ghz": gui / 8 / dur,
            "mfma_busy_fraction": busy_cycles / 1024 / (gui / 8) if busy_cycles else None,
            "lds_bank_conflict_fraction": conf / ldsact if ldsact else None,
            "per_kernel": per_kernel if len(names) > 1 else None,
            "source_hash": bench.source_hash(), "source": "profiles/" + os.path.basename(out_txt)}
        json.dump(allrec, open(out_json, "w"), indent=1)

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4:])
