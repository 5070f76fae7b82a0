#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of one profiled bench run (gpurun_out/<dir>/{kt,fetch,write}) into the
committed summaries under profiles/: kernel-trace stats CSV, a PMC table (HBM bytes = 2*FETCH_SIZE +
WRITE_SIZE per MI355X_MICROARCH.md) and profiles/traffic.json, which bench.py reads for
roofline.traffic.   usage: summarize_profiles.py gpurun_out/prof2 r01 22"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, tag, E = sys.argv[1], sys.argv[2], int(sys.argv[3])
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ne = 2 ** E
TE, M = 128, 4   # BtdTile<4, true>: 256 threads x 2 slabs


def grid(owned):
    return ((ne + owned - 1) // owned) * 256


roles = {grid(TE - 2 * 4): "fused_down_L0", grid(((TE - 2 * 3) // 1)): "fused_up_L0",
         grid(((TE - 2 * 7) // 4) * 4): "fused_mid_L0"}
expected = {  # bytes per fine element the kernel must move
    # symmetric operator: packed inverse 80, q row 32, b 32, u 32, L rows 64; the explicit residual
    # behind the restriction (default) also reads the diagonal block 128 and the sub-diagonal column 32
    "fused_down_L0": (80 + 32 + 32 + 32 + 64 + 128 + 32, 32 + 4),
    "fused_up_L0": (80 + 32 + 32 + 32 + 64, 32),
    "fused_mid_L0": (80 + 32 + 32 + 32 + 64 + 64 + 128 + 32, 32 + 4),
}


def load(kind):
    f = glob.glob(os.path.join(src, kind, "*", "*counter_collection.csv"))[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


shutil.copy(glob.glob(os.path.join(src, "kt", "*", "*kernel_stats.csv"))[0],
            os.path.join(ROOT, "profiles", f"{tag}_bench_kernel_stats_2p{E}.csv"))
F, W = load("fetch"), load("write")
lines = [f"# rocprofv3 PMC summary {tag} -- `python bench.py --steps 5 --warmup 1` (config 3/4 hierarchy, 2^{E} fine elements, MI355X)",
         "",
         "Separate passes `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (`--output-format csv`), KiB per dispatch,",
         "mean over the pass.  gfx950: FETCH_SIZE counts half the bytes of a coalesced streaming read",
         "(MI355X_MICROARCH.md, HBM), so HBM bytes = 2*FETCH_SIZE + WRITE_SIZE; the 'expected' columns are the bytes the",
         "kernel's arrays hold per launch (known counts, halo re-reads not included) -- the calibration on this access pattern.",
         "",
         "| kernel | role (grid threads) | launches | FETCH KiB | WRITE KiB | HBM bytes 2F+W | expected read | expected write |",
         "|---|---|---|---|---|---|---|---|"]
traffic = {}
for (k, g), (f, n) in sorted(F.items(), key=lambda kv: -kv[1][0]):
    if "btd_fused" not in k and "cr_" not in k:
        continue
    w = W.get((k, g), (0.0, 0))[0]
    role = roles.get(g, "")
    hb = (2 * f + w) * 1024
    er, ew = ("", "")
    if role in expected:
        er, ew = (expected[role][0] * ne, expected[role][1] * ne)
        traffic[f"{role}_log2n{E}"] = hb
    name = k.split("(")[0].replace("void ", "")
    lines.append(f"| `{name}` | {role} ({g}) | {n} | {f:.1f} | {w:.1f} | {hb:.4g} | {er} | {ew} |")
lines += ["", f"Kernel-trace stats of the same command: profiles/{tag}_bench_kernel_stats_2p{E}.csv."]
open(os.path.join(ROOT, "profiles", f"{tag}_pmc_summary_2p{E}.md"), "w").write("\n".join(lines) + "\n")
tf = os.path.join(ROOT, "profiles", "traffic.json")
old = json.load(open(tf)) if os.path.exists(tf) else {}
old.update(traffic)
json.dump(old, open(tf, "w"), indent=1)
print("\n".join(lines))
