#!/usr/bin/env python3
"""tools/timeline_gaps.py TRACE.csv [first_kernel_substring]: one steady-state step of a rocprofv3 --kernel-trace run as a
timeline -- start (us from the step's first kernel), duration, gap to the previous kernel's end, queue, kernel -- to see
where a step with collectives loses time (launch gaps at stream switches vs the collectives' own kernels)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
first = sys.argv[2] if len(sys.argv) > 2 else "preprocess_fwd_kernel"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
if len(starts) < 4:
    sys.exit("not enough steps in the trace")
a, b = starts[-3], starts[-2]          # the second-to-last step
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-60:]
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  q{r.get('Queue_Id', '?'):>3}  {name}")
    busy += e - s
    prev_end = max(prev_end, e)
print(f"step span {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us")
