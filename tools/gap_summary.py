#!/usr/bin/env python
"""Idle time of the GPU between kernels, from a rocprofv3 --kernel-trace results .db (rocpd sqlite).
Sweeps the dispatches in start order, keeps the running maximum of the end times, and charges every interval in which
nothing ran to the pair (kernel that ended last, kernel that starts next).
usage: python tools/gap_summary.py results.db [first_kernel_substring]   (trace cut in front of the first match)"""
import sqlite3
import sys
from collections import defaultdict


def short(n):
    n = n.replace("void ", "").replace("p2p::", "")
    return n.split("(")[0][:40]


def main(path, first=None):
    c = sqlite3.connect(path)
    try:
        rows = c.execute("select name, start, end from kernels order by start").fetchall()
    except Exception as e:                                   # schema differs: show what there is
        print("no view 'kernels':", e)
        for r in c.execute("select type, name from sqlite_master"):
            print(r)
        return
    if first:
        idx = [i for i, r in enumerate(rows) if first in r[0]]
        rows = rows[idx[len(idx) // 4]:] if idx else rows        # skip the warm-up quarter
    busy, gaps, cnt = 0, defaultdict(int), defaultdict(int)
    t_end, last = rows[0][1], "(start)"
    for name, s, e in rows:
        if s > t_end:
            gaps[(last, short(name))] += s - t_end
            cnt[(last, short(name))] += 1
            busy += e - s
            t_end, last = e, short(name)
        else:
            if e > t_end:
                busy += e - t_end
                t_end, last = e, short(name)
    if first:                                                # per occurrence of the anchor kernel: period, busy, idle since the previous one
        print(f"# per {first} dispatch: start_ms  duration_ms  since_previous_end_ms  idle_in_between_ms  dispatches_in_between")
        prev_end, t_run, idle_acc, n_between, t0 = None, None, 0, 0, rows[0][1]
        for name, s, e in rows:
            if t_run is not None and s > t_run:
                idle_acc += s - t_run
            t_run = e if t_run is None else max(t_run, e)
            n_between += 1
            if first in name:
                if prev_end is not None:
                    print(f"{(s - t0) / 1e6:>10.2f} {(e - s) / 1e6:>8.3f} {(s - prev_end) / 1e6:>8.3f} {idle_acc / 1e6:>8.3f} {n_between:>5}")
                prev_end, idle_acc, n_between = e, 0, 0
    span = t_end - rows[0][1]
    idle = sum(gaps.values())
    print(f"# {path}: {len(rows)} dispatches, span {span / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, idle {idle / 1e6:.2f} ms "
          f"({100.0 * idle / span:.1f} %)")
    print(f"{'idle_us':>10} {'count':>6} {'avg_us':>8}  after -> before")
    for k, v in sorted(gaps.items(), key=lambda kv: -kv[1])[:25]:
        print(f"{v / 1e3:>10.1f} {cnt[k]:>6} {v / 1e3 / cnt[k]:>8.1f}  {k[0]} -> {k[1]}")


if __name__ == "__main__":
    main(*sys.argv[1:3])
