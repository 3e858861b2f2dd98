#!/usr/bin/env python3
"""Per-kernel mean of every PMC counter in one or more rocprofv3 (rocpd) databases -> markdown table.

  python tools/rocpd_pmc_summary.py gpurun_out/pmc_sq/pmc_results.db [more.db ...] > profiles/rNN_pmc.md
"""
import sqlite3
import sys
from collections import defaultdict


def main():
    acc = defaultdict(lambda: defaultdict(list))
    for path in sys.argv[1:]:
        con = sqlite3.connect(path)
        per_dispatch = defaultdict(float)
        meta = {}
        for did, kname, cname, value in con.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection"):
            per_dispatch[(did, cname)] += value   # one row per (dispatch, counter, dimension instance): sum over instances
            meta[did] = kname
        for (did, cname), v in per_dispatch.items():
            acc[meta[did]][cname].append(v)
    counters = sorted({c for k in acc.values() for c in k})
    print("| kernel | dispatches | " + " | ".join(counters) + " |")
    print("|---|---|" + "---|" * len(counters))
    for kname in sorted(acc, key=lambda k: -max(len(v) for v in acc[k].values())):
        short = kname.split("(")[0][-60:]
        n = max(len(v) for v in acc[kname].values())
        cells = []
        for c in counters:
            v = acc[kname].get(c)
            cells.append(f"{sum(v) / len(v):.4g}" if v else "")
        print(f"| {short} | {n} | " + " | ".join(cells) + " |")


if __name__ == "__main__":
    main()
