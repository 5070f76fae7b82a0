#!/usr/bin/env python3
"""bench.py -- V-cycle hot path of AgglomerationMultigrid1D on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--log2-elems E]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one multigrid_v_cycle (V(3,3), alpha = 2/3, src/solvers.jl:19-50) over the
BASELINE.json config-3/4 hierarchy: DG p=3 fine level (2^E elements) -> AggDG pAgg=1 (4:1) ->
AggDG (2:1) -> AggDG (2:1), all operators and vectors resident in HBM before the timed region.
Default E = 24: the size BASELINE.json's north_star / BASELINE.md section 2 quote the 1-vs-8-GPU
target on ("2^24-element p=3 DG hierarchy"), so that `--gpus 1,2,4,8` is one strong-scaling
series; at N = 1 the same line also carries the literal config-3 size (2^22) as `config3_2p22`.
For N > 1 the hierarchy is partitioned by contiguous element range (config 4).  Rank 0 prints
ONE JSON line.

metric  fine-level DoF-updates/s per V-cycle = N_fine * (nPre + nPost) / t_vcycle, the whole
        cycle timed (the coarsest direct solve included; its share is reported beside it).
roofline  dominant kernel = the fused fine-level launch; achieved = algorithmic bytes per launch
        (SURVEY.md 8d byte model, from actual nnz) / its mean HIP-event duration measured inside
        the timed region on the launch stream.
extra fields  preconditioned_residual_restriction (the cheaper restriction form, for comparison only:
        it is not the default because it diverges as an iteration at 2^24),
        outer_solvers_to_1e-8 (device-resident multigrid loop and ldiv!-preconditioned CG),
        vcycles_loop (cycles fused across the fine level), config3_2p22, config5_shape_1gpu.
cpu_baseline  the plain-C single-thread restatement (oracle/aggmg_oracle_c.c, kind "port") on a
        bounded sample of the same workload, host cores of this box, rank 0, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2-elems", type=int, default=24, help="fine DG elements = 2^E (north-star scaling size 24)")
    ap.add_argument("--also-log2-elems", type=int, default=22,
                    help="N = 1 only: second, untimed-region run at this size (config 3 literal: 22); 0 = off")
    ap.add_argument("--p", type=int, default=3)
    ap.add_argument("--cpu-log2-elems", type=int, default=20, help="size of the CPU-baseline sample")
    ap.add_argument("--cpu-cycles", type=int, default=4)
    ap.add_argument("--cg-log2-elems", type=int, default=20,
                    help="N = 1 only: extra V-cycle on the CG p=4,2,1 -> DG p=0 hierarchy (config 5 shape) at 2^E "
                         "elements, generic CSR kernels; 0 = off")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-smoother-bench", action="store_true")
    return ap.parse_args()


def cpu_baseline(args, nPre, nPost, alpha):
    """Plain-C port of the reference algorithm, 1 thread, on a 2^cpu_log2_elems-element instance
    of the same hierarchy (same p, ratios, BCs).  Checker code: only timed, never shipped."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle
    from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy
    n = 2 ** args.cpu_log2_elems
    U = UniformDgAggHierarchy(n, p=args.p, pAgg=1, ratios=(4, 2, 2))
    C = c_oracle.COracleHierarchy([U.stiffness_csc(k) for k in range(U.nlevels)],
                                  [U.interpolation_csc(k) for k in range(U.nlevels - 1)],
                                  [U.levels[k]['m'] for k in range(U.nlevels - 1)])
    b = U.rhs()
    x = np.zeros(len(b))
    tot = coarse = 0.0
    for _ in range(args.cpu_cycles):
        x, dt, cs = C.vcycle(x, b, nPre, nPost, alpha)
        tot += dt
        coarse += cs
    N = len(b)
    # same cycle, OpenMP row-gather variant on the cores this box gives one GPU job
    import ctypes
    nthreads = min(16, os.cpu_count() or 1)
    try:
        omp = ctypes.CDLL("libgomp.so.1")
        omp.omp_set_num_threads(nthreads)
        C.enable_omp([U.stiffness_csc(k) for k in range(U.nlevels)], [U.interpolation_csc(k) for k in range(U.nlevels - 1)])
        xo = np.zeros(N)
        xo, _, _ = C.vcycle_omp(xo, b, nPre, nPost, alpha)
        tomp = 0.0
        for _ in range(args.cpu_cycles):
            xo, dto, _ = C.vcycle_omp(xo, b, nPre, nPost, alpha)
            tomp += dto
        omp_line = {"value": N * (nPre + nPost) * args.cpu_cycles / tomp, "unit": "DoF-updates/s", "cores": nthreads,
                    "kind": "port", "sample": "same sample, OpenMP row-gather variant of the C restatement"}
    except Exception as exc:  # no libgomp: the serial line stands alone
        omp_line = {"error": str(exc)}
    return {
        "openmp": omp_line,
        "value": N * (nPre + nPost) * args.cpu_cycles / tot,
        "unit": "DoF-updates/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{args.cpu_cycles} V(3,3) cycles, same hierarchy at 2^{args.cpu_log2_elems} fine elements "
                  f"(N_fine={N}), plain-C restatement, 1 thread of {os.cpu_count()} host cores; "
                  f"coarsest solve {1e3 * coarse / args.cpu_cycles:.1f} ms of {1e3 * tot / args.cpu_cycles:.1f} ms per cycle",
        "ms_per_cycle": 1e3 * tot / args.cpu_cycles,
    }


def smoother_bench(mg, ctx, args, alpha):
    """BASELINE config 2: DG n=2^20 p=3, 100 fused block-Jacobi sweeps and 100 residuals."""
    from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy
    from agglomerationmultigrid1d_amd import _lib
    n = 2 ** 20
    U = UniformDgAggHierarchy(n, p=args.p, pAgg=1, ratios=())
    op = mg.DeviceOperator(U.stiffness_csc(0), _lib.OP_STIFFNESS, ctx)
    S = mg.BlockJacobi(op, U.descriptor(0).mBlockInds, ctx)
    N = op.shape[0]
    b = ctx.to_device(U.rhs())
    u = ctx.to_device(np.zeros(N))
    v = ctx.alloc(N)
    r = ctx.alloc(N)
    lib = ctx.lib
    from agglomerationmultigrid1d_amd.api import _ptr
    out = {}
    nnzA = op.nnz
    S_bytes = 12 * nnzA + 4 * (N + 1) + 8 * 4 * N + 24 * N
    R_bytes = 12 * nnzA + 4 * (N + 1) + 24 * N
    for label, per_launch in (("sweeps_1_per_launch", 1), ("sweeps_4_per_launch", 4), ("sweeps_8_per_launch", 8)):
        reps = 100 // per_launch
        for _ in range(2):
            ctx.check(lib.aggmg_smooth_dev(ctx.handle, op.handle, S.handle, _ptr(u), _ptr(b), alpha, per_launch, _ptr(v)))
        ctx.synchronize()
        t0 = time.perf_counter()
        src, dst = u, v
        for _ in range(reps):
            ctx.check(lib.aggmg_smooth_dev(ctx.handle, op.handle, S.handle, _ptr(src), _ptr(b), alpha, per_launch, _ptr(dst)))
            src, dst = dst, src
        ctx.synchronize()
        dt = time.perf_counter() - t0
        sweeps = reps * per_launch
        out[label] = {"dof_updates_per_s": N * sweeps / dt, "us_per_sweep": 1e6 * dt / sweeps,
                      "algorithmic_GBs": S_bytes * sweeps / dt / 1e9,
                      "frac_of_8TBs": S_bytes * sweeps / dt / 1e9 / HBM_PEAK_GBS}
    ctx.check(lib.aggmg_residual_dev(ctx.handle, op.handle, _ptr(u), _ptr(b), _ptr(r)))
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        ctx.check(lib.aggmg_residual_dev(ctx.handle, op.handle, _ptr(u), _ptr(b), _ptr(r)))
    ctx.synchronize()
    dt = time.perf_counter() - t0
    out["residual"] = {"us_per_residual": 1e6 * dt / 100, "algorithmic_GBs": R_bytes * 100 / dt / 1e9,
                       "frac_of_8TBs": R_bytes * 100 / dt / 1e9 / HBM_PEAK_GBS}
    out["workload"] = f"config 2: DG n=2^20 p={args.p}, block-Jacobi m={args.p + 1}, N={N}, nnz(A)={nnzA}"
    # the generic CSR kernels (what CG levels and unstructured operators run) on the same matrix:
    # fused point-Jacobi sweep (K2) and CSR residual, int32 indices + fp64 values actually read
    op2 = mg.DeviceOperator(U.stiffness_csc(0), _lib.OP_STIFFNESS, ctx)
    J = mg.JacobiSmoother(op2, ctx)
    Sj_bytes = 12 * nnzA + 4 * (N + 1) + 8 * N + 24 * N
    for _ in range(2):
        ctx.check(lib.aggmg_smooth_dev(ctx.handle, op2.handle, J.handle, _ptr(u), _ptr(b), alpha, 1, _ptr(v)))
    ctx.synchronize()
    t0 = time.perf_counter()
    src, dst = u, v
    for _ in range(100):
        ctx.check(lib.aggmg_smooth_dev(ctx.handle, op2.handle, J.handle, _ptr(src), _ptr(b), alpha, 1, _ptr(dst)))
        src, dst = dst, src
    ctx.synchronize()
    dt = time.perf_counter() - t0
    out["generic_csr_point_jacobi"] = {"us_per_sweep": 1e6 * dt / 100, "algorithmic_GBs": Sj_bytes * 100 / dt / 1e9,
                                       "frac_of_8TBs": Sj_bytes * 100 / dt / 1e9 / HBM_PEAK_GBS}
    ctx.check(lib.aggmg_residual_dev(ctx.handle, op2.handle, _ptr(u), _ptr(b), _ptr(r)))
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        ctx.check(lib.aggmg_residual_dev(ctx.handle, op2.handle, _ptr(u), _ptr(b), _ptr(r)))
    ctx.synchronize()
    dt = time.perf_counter() - t0
    out["generic_csr_residual"] = {"us_per_residual": 1e6 * dt / 100, "algorithmic_GBs": R_bytes * 100 / dt / 1e9,
                                   "frac_of_8TBs": R_bytes * 100 / dt / 1e9 / HBM_PEAK_GBS}
    return out


def cg_bench(mg, ctx, args, nPre, nPost, alpha):
    """BASELINE config 5's realisable shape on ONE GPU: CG p=4 -> 2 -> 1 (point-Jacobi,
    Galerkin operators) -> DG p=0, V(3,3); every level runs the generic CSR kernels."""
    from agglomerationmultigrid1d_amd.uniform import UniformCgDgHierarchy, build_device_cg_hierarchy
    n = 2 ** args.cg_log2_elems
    t0 = time.perf_counter()
    U = UniformCgDgHierarchy(n, ps=(4, 2, 1))
    H = build_device_cg_hierarchy(U, ctx)
    N = U.A[0].shape[0]
    bm = U.algorithmic_bytes(nPre, nPost)
    b = ctx.to_device(U.rhs())
    xa, xb = ctx.to_device(np.zeros(N)), ctx.alloc(N)
    t_setup = time.perf_counter() - t0
    steps = max(5, args.steps // 2)
    for _ in range(2):
        H.vcycle_dev(xa, b, xb, nPre, nPost, alpha)
        xa, xb = xb, xa
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        H.vcycle_dev(xa, b, xb, nPre, nPost, alpha)
        xa, xb = xb, xa
    ctx.synchronize()
    dt = time.perf_counter() - t0
    vb = sum(l["vcycle"] for l in bm)
    return {"workload": f"CG n=2^{args.cg_log2_elems} p=4 -> 2 -> 1 -> DG p=0, point-Jacobi, V(3,3), N_fine={N}, "
                        f"nnz(A_1)={U.A[0].nnz}",
            "value": N * (nPre + nPost) * steps / dt, "unit": "DoF-updates/s", "ms_per_step": 1e3 * dt / steps,
            "achieved_algorithmic_GBs_vcycle": vb * steps / dt / 1e9, "frac_of_8TBs": vb * steps / dt / 1e9 / HBM_PEAK_GBS,
            "coarse_solve": H.coarse_info(), "setup_s": t_setup}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    nPre = nPost = 3
    alpha = 2.0 / 3.0
    if world != args.gpus:
        if not (world == 1 and args.gpus == 1):
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    import agglomerationmultigrid1d_amd as mg
    from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy, build_device_hierarchy

    if world > 1 or os.environ.get("AGGMG_FORCE_DIST") == "1":   # (forced: smoke of the N > 1 code path on one rank)
        from agglomerationmultigrid1d_amd import distributed as dist_mg
        return dist_mg.bench_main(args, rank, world, local_rank, nPre, nPost, alpha)

    ctx = mg.Context(local_rank)

    def run_size(E, steps, warmup, profile):
        """build the hierarchy at 2^E fine elements, run `steps` timed V-cycles and the same count
        through the multi-cycle entry point"""
        n = 2 ** E
        t_setup = time.perf_counter()
        U = UniformDgAggHierarchy(n, p=args.p, pAgg=1, ratios=(4, 2, 2))
        H = build_device_hierarchy(U, ctx)
        bytes_model = U.algorithmic_bytes(nPre, nPost)
        N = U.levels[0]['m'] * U.levels[0]['ne']
        b_host = U.rhs()
        b = ctx.to_device(b_host)
        xa = ctx.to_device(np.zeros(N))
        xb = ctx.alloc(N)
        level_sizes = [lv['m'] * lv['ne'] for lv in U.levels]
        del U
        t_setup = time.perf_counter() - t_setup
        assert all(H.structured_levels()), "fused HIP kernels were not selected"
        src, dst = xa, xb
        for _ in range(warmup):
            H.vcycle_dev(src, b, dst, nPre, nPost, alpha)
            src, dst = dst, src
        ctx.synchronize()
        if profile:
            ctx.profile_enable(2)   # the timed region carries events for the dominant kernel only
        coarse_ms = 0.0
        t0 = time.perf_counter()
        for _ in range(steps):
            H.vcycle_dev(src, b, dst, nPre, nPost, alpha)
            coarse_ms += H.last_coarse_ms()
            src, dst = dst, src
        ctx.synchronize()
        dt = time.perf_counter() - t0
        prof = None
        if profile:
            ctx.profile_enable(False)
            prof = ctx.profile_collect()
            # per-kernel table: a second, untimed pass with events around every launch (an event
            # pair costs ~7 us of stream time, which would otherwise sit inside `value`)
            ctx.profile_enable(1)
            for _ in range(steps):
                H.vcycle_dev(src, b, dst, nPre, nPost, alpha)
                src, dst = dst, src
            ctx.synchronize()
            ctx.profile_enable(False)
            prof_all = ctx.profile_collect()
            prof_all.update(prof)     # the dominant kernel keeps its in-region measurement
            prof_dom, prof = prof, prof_all
        # the same K cycles through the multi-cycle entry point (the loop body of multigrid(),
        # src/solvers.jl:124-126): consecutive cycles share one fused fine-level launch.  Reported
        # beside `value`, which stays K independent multigrid_v_cycle calls.
        H.vcycles_dev(src, b, dst, steps, nPre, nPost, alpha)
        ctx.synchronize()
        t1 = time.perf_counter()
        H.vcycles_dev(src, b, dst, steps, nPre, nPost, alpha)
        ctx.synchronize()
        dt_loop = time.perf_counter() - t1
        # the cheaper form of the restricted residual (AGGMG_RESTRICT_PRECONDITIONED), timed beside
        # the default: NOT `value` -- at 2^24 it turns the multigrid iteration divergent
        from agglomerationmultigrid1d_amd import _lib as _l
        H.set_restriction(_l.RESTRICT_PRECONDITIONED)
        for _ in range(warmup):
            H.vcycle_dev(src, b, dst, nPre, nPost, alpha)
            src, dst = dst, src
        ctx.synchronize()
        t1 = time.perf_counter()
        for _ in range(steps):
            H.vcycle_dev(src, b, dst, nPre, nPost, alpha)
            src, dst = dst, src
        ctx.synchronize()
        dt_fast = time.perf_counter() - t1
        H.set_restriction(_l.RESTRICT_EXPLICIT)
        info = H.coarse_info()
        # the device-resident outer loops, outside the timed region: multigrid()
        # (src/solvers.jl:116-139, residual check every 8 cycles) and CG preconditioned with
        # ldiv! to ||A x - b|| < 1e-8 ||b|| from a zero guess
        outer = {}
        try:
            t2 = time.perf_counter()
            _, ncyc, res = mg.multigrid_dev(H, ctx.to_device(np.zeros(N)), b, 400, 1e-8, check_every=8)
            outer["multigrid"] = {"cycles": ncyc, "ms": 1e3 * (time.perf_counter() - t2), "final_residual": res[-1]}
            t2 = time.perf_counter()
            _, nit, resp = mg.pcg(H, b_host, maxiter=100, tol=1e-8)
            outer["pcg_ldiv"] = {"iterations": nit, "ms_incl_h2d_d2h": 1e3 * (time.perf_counter() - t2),
                                 "final_residual": resp[-1]}
        except Exception as e:  # reported, never fatal for the bench line
            outer["error"] = repr(e)
        H.free()
        return dict(N=N, dt=dt, dt_loop=dt_loop, prof=prof, prof_dom=(prof_dom if profile else None),
                    bytes_model=bytes_model, level_sizes=level_sizes,
                    coarse_ms=coarse_ms, t_setup=t_setup, coarse_info=info, outer=outer, dt_fast=dt_fast)

    R = run_size(args.log2_elems, args.steps, args.warmup, True)
    N, dt, dt_loop, prof, bytes_model = R["N"], R["dt"], R["dt_loop"], R["prof"], R["bytes_model"]
    level_sizes, coarse_ms, t_setup = R["level_sizes"], R["coarse_ms"], R["t_setup"]

    ms_per_step = 1e3 * dt / args.steps
    value = N * (nPre + nPost) * args.steps / dt
    vcycle_bytes = sum(l['vcycle'] for l in bytes_model)
    # dominant kernel by total event time
    dom = list(R["prof_dom"].items())[0]   # the fine-level fused-down launch, timed inside the region
    (dkind, dlevel), (dms, dcnt) = dom
    lm = bytes_model[dlevel]
    per_launch = {"fused_down": nPre * lm['sweep'] + lm['residual'] + lm['restrict'],
                  "fused_up": nPost * lm['sweep'] + lm['prolong']}.get(dkind, lm['sweep'])
    achieved = per_launch / (dms / dcnt * 1e-3) / 1e9
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tfile):
        try:
            traffic = json.load(open(tfile)).get(f"{dkind}_L{dlevel}_log2n{args.log2_elems}")
        except Exception:
            traffic = None
    kern_ms = {f"{k}_L{l}": {"ms_per_launch": v[0] / v[1], "launches": v[1]} for (k, l), v in sorted(prof.items())}
    coarse_dev = [v["ms_per_launch"] for k, v in kern_ms.items() if k.startswith("coarse_")]
    coarse_step_ms = (coarse_dev[0] if (R["coarse_info"].get("on_device") and coarse_dev) else coarse_ms / args.steps)
    out = {
        "metric": "fine_level_dof_updates_per_s_per_vcycle",
        "value": value,
        "unit": "DoF-updates/s",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"config 3/4 hierarchy: V(3,3) cycle, alpha=2/3, DG p={args.p} n=2^{args.log2_elems} -> AggDG "
                               f"pAgg=1 4:1 -> 2:1 -> 2:1 (4 levels), uniform mesh, Neumann/Dirichlet, CDir=1000n"
                               + (" (north-star 1-vs-8-GPU size; config 3's own 2^22 in config3_2p22)"
                                  if args.log2_elems == 24 else ""),
                   "fine_dofs": N, "level_dofs": level_sizes, "nPre": nPre, "nPost": nPost,
                   "parallelism": "single GPU"},
        "achieved_algorithmic_GBs_vcycle": vcycle_bytes * args.steps / dt / 1e9,
        "vcycles_loop": {"value": N * (nPre + nPost) * args.steps / dt_loop, "unit": "DoF-updates/s",
                         "ms_per_cycle": 1e3 * dt_loop / args.steps,
                         "note": f"aggmg_vcycles_dev({args.steps} cycles): same arithmetic as {args.steps} separate "
                                 "V-cycles (bitwise), post-smoothing of cycle i and pre-smoothing of cycle i+1 in one "
                                 "fine-level launch"},
        "outer_solvers_to_1e-8": R["outer"],
        "preconditioned_residual_restriction": {
            "value": N * (nPre + nPost) * args.steps / R["dt_fast"], "unit": "DoF-updates/s",
            "ms_per_step": 1e3 * R["dt_fast"] / args.steps,
            "note": "aggmg_hier_set_restriction(AGGMG_RESTRICT_PRECONDITIONED): L'(b - A u) taken from the sweeps' "
                    "preconditioned residual instead of the operator's own entries; cheaper, but its rounding error on "
                    "the smoothest mode grows like n^2 and at 2^24 fine elements the multigrid iteration diverges "
                    "(x2.1 per cycle on that mode against x0.5 for the default and for reference-order arithmetic). "
                    "Reported for comparison only; `value` is the default (explicit) form."},
        "coarse_solve": R["coarse_info"],
        "coarse_solve_host_ms_per_step": coarse_ms / args.steps,
        # SURVEY 8d times the coarsest solve separately: on the device its duration is the `coarse`
        # entry of the per-kernel table (HIP events, untimed second pass), on the host path the host clock
        "coarse_solve_ms_per_step": coarse_step_ms,
        "value_excl_coarse_solve": N * (nPre + nPost) / max(1e-3 * (ms_per_step - coarse_step_ms), 1e-12),
        "roofline": {"bound": "hbm", "kernel": f"btd_fused_kernel<{args.p + 1},cmp> {dkind} level {dlevel + 1}",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "algorithmic_bytes_per_launch": per_launch,
                     "physical_GBs": (traffic / (dms / dcnt * 1e-3) / 1e9) if traffic else None,
                     "physical_frac": (traffic / (dms / dcnt * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                     "ms_per_launch": dms / dcnt, "launches_timed": dcnt,
                     "note": "achieved = algorithmic bytes (SURVEY 8d model: CSR int32 + fp64, every sweep re-reading "
                             "the operator) / HIP-event duration; the fused kernel reads the operator once per launch, "
                             "so achieved exceeds what the launch physically moves: traffic = PMC bytes per launch "
                             "(2*FETCH_SIZE + WRITE_SIZE, profiles/), physical_* = traffic / the same duration"},
        "kernels": kern_ms,
        "setup_s": t_setup,
    }
    if args.also_log2_elems and args.also_log2_elems != args.log2_elems:
        R2 = run_size(args.also_log2_elems, args.steps, args.warmup, False)
        out[f"config3_2p{args.also_log2_elems}"] = {
            "workload": f"config 3: same hierarchy at 2^{args.also_log2_elems} fine elements (N_fine={R2['N']})",
            "value": R2["N"] * (nPre + nPost) * args.steps / R2["dt"], "unit": "DoF-updates/s",
            "ms_per_step": 1e3 * R2["dt"] / args.steps,
            "vcycles_loop_ms_per_cycle": 1e3 * R2["dt_loop"] / args.steps, "setup_s": R2["t_setup"],
            "preconditioned_residual_restriction_ms_per_step": 1e3 * R2["dt_fast"] / args.steps,
            "outer_solvers_to_1e-8": R2["outer"]}
    if args.cg_log2_elems:
        out["config5_shape_1gpu"] = cg_bench(mg, ctx, args, nPre, nPost, alpha)
    if not args.no_smoother_bench:
        out["smoother_only"] = smoother_bench(mg, ctx, args, alpha)
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, nPre, nPost, alpha)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
