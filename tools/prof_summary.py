#!/usr/bin/env python
"""Turn a rocprofv3 results .db (rocpd sqlite) into the text summary committed under profiles/.
usage: python tools/prof_summary.py gpurun_out/prof/bench_results.db > profiles/rNN_kernel_stats.txt"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    print(f"# rocprofv3 --kernel-trace --stats summary of {path}")
    print(f"# durations in microseconds")
    print(f"{'calls':>6} {'total_us':>12} {'avg_us':>10} {'pct':>6}  kernel")
    for name, calls, total, avg, pct in rows:
        short = name if len(name) < 110 else name[:107] + "..."
        print(f"{calls:>6} {total:>12.1f} {avg:>10.2f} {pct:>6.2f}  {short}")
    try:
        cur = c.execute("select name, count(*), avg(value) from pmc_events group by name")
        pmc = cur.fetchall()
        if pmc:
            print("\n# PMC counters (avg per dispatch)")
            for r in pmc:
                print(r)
    except Exception:
        pass


if __name__ == "__main__":
    main(sys.argv[1])
