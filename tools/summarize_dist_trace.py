#!/usr/bin/env python3
"""Launch-by-launch table of one rank's share (tools/exp_dist_rank.py under `rocprofv3 --kernel-trace --output-format csv`):
    summarize_dist_trace.py <dir with *kernel_trace.csv>  ->  markdown rows (kernel, stream, mean / min us over the last cycles)
The dispatches of the library repeat with the period of one cycle; the period is found from the kernel-name sequence."""
import csv
import glob
import os
import statistics
import sys

src = sys.argv[1]
f = max(glob.glob(os.path.join(src, "*", "*kernel_trace.csv")), key=os.path.getmtime)
rows = [r for r in csv.DictReader(open(f)) if "aggmg" in r["Kernel_Name"] or "loopback" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(r):
    return r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("aggmg::", "")


names = [short(r) for r in rows]
W = 6
P = next(S for S in range(4, 64) if all(names[-S * (k + 1):len(names) - S * k] == names[-S:] for k in range(1, W)))
tail = rows[-P * W:]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tail]
first = max(range(P), key=lambda i: statistics.mean(dur[i::P][:W - 1]))   # the fine-level descent opens the cycle
tail, dur = tail[first:first + P * (W - 1)], dur[first:first + P * (W - 1)]
print(f"period {P} dispatches; {W - 1} cycles")
print("| # | kernel | stream | us mean | us min |")
print("|---|---|---|---|---|")
for i in range(P):
    d = dur[i::P]
    print(f"| {i + 1} | `{short(tail[i])}` | {tail[i]['Stream_Id']} | {statistics.mean(d):.1f} | {min(d):.1f} |")
t0 = [int(r["Start_Timestamp"]) for r in tail[0::P]]
per = [(t0[i + 1] - t0[i]) / 1e3 for i in range(len(t0) - 1)]
print(f"\nperiod of the cycle in the trace: mean {statistics.mean(per):.1f} us, min {min(per):.1f} us; sum of the mean durations {sum(statistics.mean(dur[i::P]) for i in range(P)):.1f} us")
