#!/usr/bin/env python3
"""Per-dispatch kernel durations of the last bench step from a `rocprofv3 --kernel-trace --output-format csv` directory.

usage: python tools/dispatch_times.py <dir> [kernels_per_step]
Prints, in launch order, the kernels of the last complete step with their durations in microseconds (a step starts at
preprocess_fwd_kernel), averaged over the last few steps position by position.
"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if "preprocess_fwd_kernel" in r[2]]
steps = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
steps = [s for s in steps if len(s) == len(steps[-1])][-8:]
acc = defaultdict(list)
for s in steps:
    for pos, (t0, t1, name) in enumerate(s):
        acc[pos].append(((t1 - t0) / 1e3, name, (s[pos + 1][0] - t1) / 1e3 if pos + 1 < len(s) else 0.0))
tot = 0.0
for pos in sorted(acc):
    v = acc[pos]
    dur = sum(x[0] for x in v) / len(v)
    gap = sum(x[2] for x in v) / len(v)
    tot += dur + gap
    name = v[0][1].split("(")[0].replace("segs::", "")[:70]
    print(f"{pos:3d} {dur:9.1f} us  gap after {gap:6.1f}  {name}")
print(f"steps averaged: {len(steps)}; kernels+gaps per step {tot:.1f} us")
