#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of one profiled `tools/profile_vcycle.py` run into the committed summaries
under profiles/.

    summarize_profiles.py <dir> <tag> <kind> <log2_elems>
    <dir> holds three passes of the SAME command: kt/ (--kernel-trace --stats), fetch/ (--pmc FETCH_SIZE),
    write/ (--pmc WRITE_SIZE), all with --output-format csv

The profiled command runs nothing but default-mode V-cycles, so the library's kernel dispatches repeat
with the period of one cycle; a dispatch's role is its position in that period: fused launches before the
coarsest-solve kernels are the descent of levels 1, 2, ..., those after it the ascent in reverse order.
Written: profiles/<tag>_<kind>_2p<E>_kernel_stats.csv (rocprofv3's own stats table),
profiles/<tag>_<kind>_2p<E>_roles.md (per-role duration from the kernel trace + PMC bytes), and the
per-role HBM bytes in profiles/traffic.json (HBM bytes = 2*FETCH_SIZE + WRITE_SIZE, the gfx950 correction
of MI355X_MICROARCH.md section HBM), which bench.py reads for roofline.traffic."""
import csv
import glob
import json
import os
import shutil
import statistics
import sys

src, tag, kind, E = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ne = 2 ** E


def short(name):
    return name.split("(")[0].replace("void ", "").replace("aggmg::", "")


def is_ours(name):
    return "aggmg::" in name and "dot_" not in name


def period(seq, windows=5):
    """smallest S with the last `windows` windows of length S equal (the profiled command runs at least that many
    cycles; fewer windows would take the repeated sweeps of one smoothing step for a cycle)"""
    for S in range(1, len(seq) // windows + 1):
        if all(seq[len(seq) - (w + 1) * S:len(seq) - w * S] == seq[-S:] for w in range(1, windows)):
            return S
    raise SystemExit("no periodic dispatch pattern found")


def roles_of(window):
    """window: list of short kernel names of one cycle"""
    fused = [i for i, k in enumerate(window) if "fused_kernel" in k or "btd_pair_" in k]
    coarse = [i for i, k in enumerate(window) if "cr_" in k]
    first_c = coarse[0] if coarse else len(window)
    names = []
    down = [i for i in fused if i < first_c]
    up = [i for i in fused if i > first_c]
    # a two-level launch (btd_pair_*_kernel, r04) carries its level and the next coarser one: levels are counted
    # through it
    lev, lv = {}, 0
    for i in down:
        lev[i] = lv
        lv += 2 if "btd_pair_" in window[i] else 1
    lv = 0
    for i in reversed(up):
        lev[i] = lv
        lv += 2 if "btd_pair_" in window[i] else 1
    for i, k in enumerate(window):
        if i in down:
            names.append(f"pair_down_L{lev[i]}" if "btd_pair_" in k else f"fused_down_L{lev[i]}")
        elif i in up:
            names.append(f"pair_up_L{lev[i]}" if "btd_pair_" in k else f"fused_up_L{lev[i]}")
        elif i in coarse:
            names.append(f"coarse_{coarse.index(i)}:{k.split('<')[0]}")
        else:
            names.append(f"other_{i}:{k.split('<')[0]}")
    return names


def load_trace(path):
    rows = [r for r in csv.DictReader(open(path)) if is_ours(r["Kernel_Name"])]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows


def per_role(rows, value):
    seq = [(short(r["Kernel_Name"]), r.get("Grid_Size_X", r.get("Grid_Size"))) for r in rows]
    S = period(seq)
    ncyc = 0
    while (ncyc + 1) * S <= len(seq) and seq[-(ncyc + 1) * S:len(seq) - ncyc * S] == seq[-S:]:
        ncyc += 1
    names = roles_of([k for k, _ in seq[-S:]])
    out = {}
    for pos, nm in enumerate(names):
        vals = [value(rows[len(rows) - (c + 1) * S + pos]) for c in range(ncyc)]
        out[nm] = (vals, seq[len(seq) - S + pos])
    return out, ncyc, names


def newest(pattern):
    """gpurun merges a call's outputs INTO the local directory: an earlier run's files (other pids) may still lie there"""
    return max(glob.glob(pattern), key=os.path.getmtime)


kt = newest(os.path.join(src, "kt", "*", "*kernel_trace.csv"))
stats = newest(os.path.join(src, "kt", "*", "*kernel_stats.csv"))
base = f"{tag}_{kind}_2p{E}"
shutil.copy(stats, os.path.join(ROOT, "profiles", f"{base}_kernel_stats.csv"))
dur, ncyc, names = per_role(load_trace(kt), lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
vgpr = {}
for r in load_trace(kt):
    vgpr[short(r["Kernel_Name"])] = (r["VGPR_Count"], r["LDS_Block_Size"])


def pmc(which):
    f = glob.glob(os.path.join(src, which, "*", "*counter_collection.csv"))
    if not f:
        return None
    rows = load_trace(max(f, key=os.path.getmtime))
    return per_role(rows, lambda r: float(r["Counter_Value"]))[0]


F, W = pmc("fetch"), pmc("write")

# bytes the kernel's arrays hold per launch (known counts, halo re-reads not included): the calibration
# of the PMC figures on this access pattern
if kind == "dg":
    # per fine element (m = 4): packed symmetric inverse 80, q row 32, b 32, u 32, L rows 32 (r03: the unit first
    # column of the two-mode transfer is not stored, r02: 64); the explicit residual behind the restriction also
    # reads the diagonal block 128 and the sub-diagonal column 32
    expected = {"fused_down_L0": ((80 + 32 + 32 + 32 + 32 + 128 + 32) * ne, (32 + 4) * ne),
                "fused_up_L0": ((80 + 32 + 32 + 32 + 32 + 4) * ne, 32 * ne)}
else:
    # per fine block of 4 rows: dblk 128 + subrow 32 + supcol 32, b 32, u 32, perm 16, L rows 96 (3 per row)
    expected = {"fused_down_L0": ((192 + 32 + 32 + 16 + 96) * ne, (32 + 16) * ne),
                "fused_up_L0": ((192 + 32 + 32 + 16 + 96 + 16) * ne, 32 * ne)}

lines = [f"# rocprofv3 summary {tag}: `python tools/profile_vcycle.py --kind {kind} --log2-elems {E}` on MI355X",
         "",
         f"Roles = position of a dispatch in the cycle's launch sequence ({len(names)} launches per V(3,3) cycle, "
         f"{ncyc} cycles in the trace).  Duration: `--kernel-trace` pass (End - Start, ms).  FETCH / WRITE: separate",
         "`--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes, KiB per dispatch.  gfx950: FETCH_SIZE counts half the bytes of a",
         "coalesced streaming read (MI355X_MICROARCH.md, HBM), so HBM bytes = 2*FETCH_SIZE + WRITE_SIZE.  `expected` = the",
         "bytes the launch's arrays hold (halo re-reads not included).",
         "",
         "| role | kernel (grid threads) | VGPR | LDS B | ms mean | ms median | ms min | FETCH KiB | WRITE KiB | HBM bytes 2F+W | TB/s at mean | expected read | expected write |",
         "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
traffic = {}
for nm in names:
    vals, (k, g) = dur[nm]
    f = statistics.mean(F[nm][0]) if F and nm in F else None
    w = statistics.mean(W[nm][0]) if W and nm in W else None
    hb = (2 * f + w) * 1024 if f is not None and w is not None else None
    er, ew = expected.get(nm, ("", ""))
    vg, ld = vgpr.get(k, ("", ""))
    tbs = f"{hb / (statistics.mean(vals) * 1e-3) / 1e12:.2f}" if hb else ""
    lines.append(f"| {nm} | `{k}` ({g}) | {vg} | {ld} | {statistics.mean(vals):.4f} | {statistics.median(vals):.4f} | "
                 f"{min(vals):.4f} | {f if f is None else round(f, 1)} | {w if w is None else round(w, 1)} | "
                 f"{'' if hb is None else f'{hb:.4g}'} | {tbs} | {er} | {ew} |")
    if hb is not None and nm.startswith("coarse_"):   # the coarsest solve: its launches summed under one key
        ck = f"coarse_{kind}_log2n{E}"
        traffic[ck] = traffic.get(ck, 0.0) + hb
        traffic[ck + "_ms_profile_mean"] = traffic.get(ck + "_ms_profile_mean", 0.0) + statistics.mean(vals)
    if hb is not None and (nm.startswith("fused_") or nm.startswith("pair_")):
        key = nm.replace("fused_", "chain_") if kind == "cg" else nm
        traffic[f"{key}_{kind}_log2n{E}"] = hb
        traffic[f"{key}_{kind}_log2n{E}_ms_profile_mean"] = statistics.mean(vals)   # the kernel-trace mean of the same role
tot = sum(statistics.mean(dur[nm][0]) for nm in names)
lines += ["", f"Sum of the mean kernel durations of one cycle: {tot:.4f} ms.",
          f"rocprofv3's own per-kernel stats of the same trace: profiles/{base}_kernel_stats.csv."]
open(os.path.join(ROOT, "profiles", f"{base}_roles.md"), "w").write("\n".join(lines) + "\n")
tf = os.path.join(ROOT, "profiles", "traffic.json")
old = json.load(open(tf)) if os.path.exists(tf) else {}
old = {k: v for k, v in old.items() if "_log2n" in k and ("_dg_" in k or "_cg_" in k)}   # drop pre-r02 keys
old.update(traffic)
json.dump(old, open(tf, "w"), indent=1)
print("\n".join(lines))
