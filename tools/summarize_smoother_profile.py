#!/usr/bin/env python3
"""profiles/<tag>_dg_2p20_smoother.md and the smoother entries of profiles/traffic.json from the rocprofv3 passes of
tools/profile_smoother.py (kt/: --kernel-trace --stats, fetch/: --pmc FETCH_SIZE, write/: --pmc WRITE_SIZE, csv):

    summarize_smoother_profile.py <dir> <tag> <copies> [log2_elems]

The phases of identical launches are separated by marker dispatches (copy_segments_kernel); the first three
launches of a phase are left out (first touch of each copy).  HBM bytes = 2 * FETCH_SIZE + WRITE_SIZE (gfx950:
FETCH_SIZE counts half the bytes of a coalesced streaming read, MI355X_MICROARCH.md, HBM)."""
import csv
import glob
import json
import os
import statistics
import sys

src, tag, copies = sys.argv[1], sys.argv[2], int(sys.argv[3])
E = int(sys.argv[4]) if len(sys.argv) > 4 else 20
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PHASES = ("sweeps_1_per_launch", "residual", "sweeps_4_per_launch", "sweeps_8_per_launch")
SWEEPS = {"sweeps_1_per_launch": 1, "residual": 0, "sweeps_4_per_launch": 4, "sweeps_8_per_launch": 8}


def rows(sub, pat):
    # (gpurun merges a call's outputs INTO the local directory: an earlier run's files -- other pids -- may still lie there)
    f = sorted(glob.glob(os.path.join(src, sub, "**", pat), recursive=True), key=os.path.getmtime)
    if not f:
        raise SystemExit(f"no {pat} under {src}/{sub}")
    return list(csv.DictReader(open(f[-1])))   # the newest


def phases(seq, key):
    """seq: dispatches in order, each (kernel name, value); -> {phase: [values]}"""
    out, cur = {}, -1
    for name, val in seq:
        if "copy_segments_kernel" in name:
            cur += 1
            continue
        if 0 <= cur < len(PHASES) and "aggmg::" in name:
            out.setdefault(PHASES[cur], []).append(val)
    return out


kt = sorted(rows("kt", "*kernel_trace.csv"), key=lambda r: int(r["Start_Timestamp"]))
dur = phases([(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6) for r in kt], "ms")


def pmc(sub, counter):
    rr = [r for r in rows(sub, "*counter_collection.csv") if r["Counter_Name"] == counter]
    rr.sort(key=lambda r: int(r["Dispatch_Id"]))
    return phases([(r["Kernel_Name"], float(r["Counter_Value"])) for r in rr], counter)


fetch, write = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
# algorithmic bytes of SURVEY.md 8(d) for this matrix
meta = json.loads([ln for ln in open(os.path.join(src, "kt.log")) if ln.startswith("{")][-1])
N, nnz = meta["N"], meta["nnz"]
S_bytes = 12 * nnz + 4 * (N + 1) + 8 * 4 * N + 24 * N
R_bytes = 12 * nnz + 4 * (N + 1) + 24 * N
lines = [f"# rocprofv3 summary {tag}: `python tools/profile_smoother.py --copies {copies}` on MI355X (BASELINE config 2)",
         "",
         f"DG n = 2^{E}, p = 3: N = {N}, nnz(A) = {nnz}; block-Jacobi m = 4.  {copies} independent cop{'ies' if copies > 1 else 'y'} of operator, "
         f"smoother and vectors taken in turn ({'streams from HBM: the copies together exceed the 256 MB Infinity Cache' if copies > 1 else 'the same data swept again and again: largely served by the Infinity Cache'}).",
         "Duration: `--kernel-trace` pass (End - Start).  FETCH / WRITE: separate `--pmc` passes, KiB per dispatch; HBM bytes = 2*FETCH + WRITE",
         "(gfx950 correction).  Algorithmic bytes: SURVEY.md 8(d) (CSR int32 + fp64, every sweep re-reading the operator).", "",
         "| phase | launches | ms mean | ms median | us per sweep | FETCH KiB | WRITE KiB | HBM bytes 2F+W | physical TB/s | of 8 TB/s | algorithmic bytes / launch | algorithmic frac |",
         "|---|---|---|---|---|---|---|---|---|---|---|---|"]
tfile = os.path.join(ROOT, "profiles", "traffic.json")
traffic = json.load(open(tfile)) if os.path.exists(tfile) else {}
for ph in PHASES:
    d = dur.get(ph, [])[3:]
    if not d:
        continue
    f = statistics.mean(fetch[ph][3:]) if ph in fetch else float("nan")
    w = statistics.mean(write[ph][3:]) if ph in write else float("nan")
    hbm = (2 * f + w) * 1024
    ms = statistics.mean(d)
    sw = SWEEPS[ph]
    alg = S_bytes * sw if sw else R_bytes
    lines.append(f"| {ph} | {len(d)} | {ms:.4f} | {statistics.median(d):.4f} | {1e3 * ms / max(sw, 1):.1f} | {f:.1f} | {w:.1f} | {hbm:.4g} | "
                 f"{hbm / ms / 1e9:.2f} | {hbm / ms / 1e9 / 8:.2f} | {alg:.4g} | {alg / ms / 1e9 / 8:.2f} |")
    traffic[f"smoother_{ph}_dg_log2n{E}_copies{copies}"] = {"hbm_bytes": hbm, "ms_profile_mean": ms}
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
open(os.path.join(ROOT, "profiles", f"{tag}_dg_2p{E}_smoother_copies{copies}.md"), "w").write("\n".join(lines) + "\n")
json.dump(traffic, open(tfile, "w"), indent=1)
print("\n".join(lines))
