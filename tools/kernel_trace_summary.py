#!/usr/bin/env python3
"""usage: kernel_trace_summary.py <rocprofv3 *_kernel_trace.csv> [min share %]
Per (kernel, launch geometry): calls and average duration.  `rocprofv3 --stats` averages by kernel NAME; one bench.py run
launches the same sweep kernel for several workloads (C3's 101^4 and C4's 151^4 grid share k_sweep_lean4<2, uchar>), which
the launch geometry tells apart."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
floor = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
acc = collections.defaultdict(list)
for r in rows:
    key = (r["Kernel_Name"].split("(")[0], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")),
           r.get("LDS_Block_Size", "?"))
    acc[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
total = sum(sum(v) for v in acc.values())
print("%-72s %12s %6s %8s %8s %14s %7s" % ("kernel", "grid", "block", "lds", "calls", "avg ns", "share"))
for key, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    share = 100.0 * sum(v) / total
    if share < floor:
        continue
    print("%-72s %12s %6s %8s %8d %14.1f %6.2f%%" % (key[0][:72], key[1], key[2], key[3], len(v), sum(v) / len(v), share))
