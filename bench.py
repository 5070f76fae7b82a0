#!/usr/bin/env python3
"""bench.py -- V-cycle hot path of AgglomerationMultigrid1D on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--log2-elems E]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one multigrid_v_cycle (V(3,3), alpha = 2/3, src/solvers.jl:19-50) over the
BASELINE.json config-3/4 hierarchy: DG p=3 fine level (2^E elements) -> AggDG pAgg=1 (4:1) ->
AggDG (2:1) -> AggDG (2:1), all operators and vectors resident in HBM before the timed region.
Default E = 24: the size BASELINE.json's north_star / BASELINE.md section 2 quote the 1-vs-8-GPU
target on ("2^24-element p=3 DG hierarchy"), so that `--gpus 1,2,4,8` is one strong-scaling
series; at N = 1 the same line also carries the literal config-3 size (2^22) as `config3_2p22` and
BASELINE config 5's realisable shape (CG p=4 -> 2 -> 1 -> DG p=0) at its stated size,
2^24 elements, as `config5_2p24_1gpu`.  For N > 1 the hierarchy is partitioned by contiguous element
range (config 4).  Rank 0 prints ONE JSON line.

metric  fine-level DoF-updates/s per V-cycle = N_fine * (nPre + nPost) / t_vcycle, the whole
        cycle timed (the coarsest direct solve included; its share is reported beside it).  `value`
        comes from the K-step bracket the driver contract prescribes; `median_ms_per_step` (K more
        steps, each timed on its own) rides along.
roofline  dominant kernel = the fused fine-level descent launch.  achieved / frac: COMPULSORY bytes of
        the launch -- the arrays it has to read and to write, each once, in the format the level stores
        them (aggmg_hier_launch_bytes: index-free block rows, packed symmetric inverses, transfer rows,
        vectors) -- / its mean HIP-event duration measured inside the timed region on the launch
        stream, over 8 TB/s: a fraction, <= 1 by construction.  `traffic` = PMC bytes per launch of
        exactly this launch role (2*FETCH_SIZE + WRITE_SIZE, profiles/traffic.json, written by
        tools/summarize_profiles.py from a profile that runs only default-mode cycles); physical_frac =
        traffic / duration / 8 TB/s (what the memory system moved, halo re-reads included: a few per cent
        above frac).  frac_survey_model keeps the SURVEY.md 8(d) figure (CSR int32 + fp64 byte model with
        every sweep re-reading the operator): the fused kernel reads the operator once per launch for
        3 sweeps + residual + restriction, so that one exceeds 1 and is not a bound.
cpu_baseline  the plain-C restatement (oracle/aggmg_oracle_c.c, kind "port") on the SAME hierarchy at the
        SAME size as `value` (2^24 fine elements by default): 1 thread, and the OpenMP variant on the host
        cores of this box (count stated), rank 0, N = 1 only.
"""
import argparse
import json
import os
import statistics
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
    ap.add_argument("--cpu-log2-elems", type=int, default=0,
                    help="size of the CPU-baseline run; 0 (default) = the size of `value` (--log2-elems)")
    ap.add_argument("--cpu-cycles", type=int, default=3)
    ap.add_argument("--cg-log2-elems", type=int, default=24,
                    help="N = 1 only: V-cycle on the CG p=4,2,1 -> DG p=0 hierarchy (config 5 shape) at 2^E elements; 0 = off")
    ap.add_argument("--ragged-log2-elems", type=int, default=20,
                    help="N = 1 only: V-cycle on a perturbed mesh with agglomerates of different sizes ({2..6} sub-elements) at "
                         "2^E DG p=3 elements, beside the uniform-ratio hierarchy of the same size; 0 = off")
    ap.add_argument("--dist-config", type=int, default=4, choices=(4, 5),
                    help="N > 1 only: 4 = the config-3 DG hierarchy partitioned (default, the north-star scaling series), "
                         "5 = the CG p=4,2,1 -> DG p=0 hierarchy partitioned (2^cg-log2-elems elements)")
    ap.add_argument("--dist-smoother", default="jac", choices=("jac", "addSchwarz", "hybridSchwarz", "blockGS"),
                    help="N > 1, --dist-config 5 only: the CG levels' smoother (cg_smoother's kinds; blockGS = the labelled "
                         "red-black element Gauss-Seidel extension BASELINE config 5 names)")
    ap.add_argument("--rehearse-threads", action="store_true",
                    help="N > 1 on a box with fewer GPUs (or GPU process slots) than ranks: run the N ranks as THREADS of this one "
                         "process on GPU 0 (host-staged collectives through shared memory) -- the N-rank code path end to end, "
                         "one JSON line with n_gpus N and a `rehearsal` note; not a measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-smoother-bench", action="store_true")
    return ap.parse_args()


def cpu_baseline(args, nPre, nPost, alpha):
    """Plain-C port of the reference algorithm on the config-3 hierarchy at 2^cpu_log2_elems fine elements
    (same p, ratios, BCs): 1 thread in the reference's operation order, and the OpenMP row-gather variant
    on every host core.  Checker code: only timed, never shipped."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ctypes

    import c_oracle
    from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy
    E = args.cpu_log2_elems or args.log2_elems
    n = 2 ** E
    U = UniformDgAggHierarchy(n, p=args.p, pAgg=1, ratios=(4, 2, 2))
    As = [U.stiffness_csc(k) for k in range(U.nlevels)]
    Ls = [U.interpolation_csc(k) for k in range(U.nlevels - 1)]
    C = c_oracle.COracleHierarchy(As, Ls, [U.levels[k]['m'] for k in range(U.nlevels - 1)])
    b = U.rhs()
    N = len(b)
    x = np.zeros(N)
    ts, coarse = [], 0.0
    for _ in range(args.cpu_cycles):
        x, dt, cs = C.vcycle(x, b, nPre, nPost, alpha)
        ts.append(dt)
        coarse += cs
    ncores = os.cpu_count() or 1
    try:
        omp = ctypes.CDLL("libgomp.so.1")
        C.enable_omp(As, Ls)
        # the box may grant this job fewer CPUs than the host has (a CPU quota does not show in os.cpu_count()):
        # thread counts ascending from 8, one warm cycle each, until a count is clearly slower than the best so far
        # (oversubscribed counts cost seconds per cycle at this size and are never reached)
        share = ncores
        try:
            share = min(share, len(os.sched_getaffinity(0)))
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()
            if q != "max":
                share = min(share, max(1, int(int(q) / int(per))))
        except Exception:
            pass
        cand = sorted({t for t in (4, 8, 16, 32, 64, 128, 256, share, ncores) if 1 <= t <= ncores})
        sweep = {}
        xo = np.zeros(N)
        for t in cand:
            omp.omp_set_num_threads(t)
            if not sweep:
                xo, _, _ = C.vcycle_omp(xo, b, nPre, nPost, alpha)   # (first touch of the OpenMP variant's arrays)
            xo, dto, _ = C.vcycle_omp(xo, b, nPre, nPost, alpha)
            sweep[t] = dto
            if dto > 1.3 * min(sweep.values()) and len(sweep) >= 3:
                break
        best = min(sweep, key=sweep.get)
        omp.omp_set_num_threads(best)
        xo = np.zeros(N)
        to = []
        for _ in range(2 * args.cpu_cycles):
            xo, dto, _ = C.vcycle_omp(xo, b, nPre, nPost, alpha)
            to.append(dto)
        omp_line = {"value": N * (nPre + nPost) / statistics.median(to), "unit": "DoF-updates/s", "cores": best,
                    "kind": "port", "ms_per_cycle": 1e3 * statistics.median(to), "host_cores": ncores,
                    "ms_per_cycle_by_threads": {str(t): round(1e3 * v, 1) for t, v in sweep.items()},
                    "sample": f"same hierarchy and size, OpenMP row-gather variant of the C restatement; host has {ncores} "
                              f"cores (CPU share of this job: {share}), thread counts {sorted(sweep)} tried in ascending order with "
                              f"one cycle each and the fastest ({best}) kept, median of {len(to)} cycles"}
    except Exception as exc:  # no libgomp: the serial line stands alone
        omp_line = {"error": str(exc)}
    return {
        "openmp": omp_line,
        "value": N * (nPre + nPost) / statistics.median(ts),
        "unit": "DoF-updates/s",
        "cores": 1,
        "kind": "port",
        "sample": f"median of {args.cpu_cycles} V(3,3) cycles, config-3/4 hierarchy at 2^{E} fine elements "
                  f"(N_fine={N}), plain-C restatement in the reference's operation order, 1 thread of {ncores} host cores "
                  f"(the Julia reference adds ~6 heap allocations and one LAPACK call per block per sweep and "
                  f"re-factorises the coarsest level every cycle); coarsest solve {1e3 * coarse / args.cpu_cycles:.1f} ms "
                  f"of {1e3 * statistics.median(ts):.1f} ms per cycle",
        "ms_per_cycle": 1e3 * statistics.median(ts),
    }


def _time_loop(ctx, fn, reps):
    ctx.synchronize()
    t0 = time.perf_counter()
    fn(reps)
    ctx.synchronize()
    return time.perf_counter() - t0


def smoother_bench(mg, ctx, args, alpha):
    """BASELINE config 2: DG n=2^20 p=3, 100 fused block-Jacobi sweeps and 100 residuals; the generic
    CSR kernels on the same matrix; the chain kernel against the generic kernels on a CG p=4 matrix."""
    from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy, UniformCgDgHierarchy
    from agglomerationmultigrid1d_amd import _lib
    from agglomerationmultigrid1d_amd.api import _ptr
    lib = ctx.lib
    n = 2 ** 20
    U = UniformDgAggHierarchy(n, p=args.p, pAgg=1, ratios=())
    op = mg.DeviceOperator(U.stiffness_csc(0), _lib.OP_STIFFNESS, ctx)
    S = mg.BlockJacobi(op, U.descriptor(0).mBlockInds, ctx)
    N = op.shape[0]
    b = ctx.to_device(U.rhs())
    u = ctx.to_device(np.zeros(N))
    v = ctx.alloc(N)
    r = ctx.alloc(N)
    out = {}
    nnzA = op.nnz
    S_bytes = 12 * nnzA + 4 * (N + 1) + 8 * 4 * N + 24 * N
    R_bytes = 12 * nnzA + 4 * (N + 1) + 24 * N

    def sweeps(oph, smh, per_launch, uu, vv, bb):
        def run(reps):
            src, dst = uu, vv
            for _ in range(reps):
                ctx.check(lib.aggmg_smooth_dev(ctx.handle, oph, smh, _ptr(src), _ptr(bb), alpha, per_launch, _ptr(dst)))
                src, dst = dst, src
        return run

    def resid(oph, uu, bb, rr):
        def run(reps):
            for _ in range(reps):
                ctx.check(lib.aggmg_residual_dev(ctx.handle, oph, _ptr(uu), _ptr(bb), _ptr(rr)))
        return run

    # Two ways to loop.  "same operator": one operator / smoother / vector set swept again and again -- ~250 MB, i.e.
    # largely served by the 256 MB Infinity Cache (how r01 / r02 timed it: above the HBM copy ceiling).  "rotating":
    # three independent copies taken in turn (~0.75 GB), so that every launch streams from HBM -- the figure to hold
    # against the HBM roofline.  traffic / physical_frac: PMC bytes of exactly these launches
    # (profiles/r03_dg_2p20_smoother_copies{1,3}.md via profiles/traffic.json).
    sets = [(op, S, u, v, b, r)]
    for _ in range(2):
        o2 = mg.DeviceOperator(U.stiffness_csc(0), _lib.OP_STIFFNESS, ctx)
        sets.append((o2, mg.BlockJacobi(o2, U.descriptor(0).mBlockInds, ctx), ctx.to_device(np.zeros(N)), ctx.alloc(N),
                     ctx.to_device(U.rhs()), ctx.alloc(N)))

    def rot_sweeps(per_launch, ncopies):
        def run(reps):
            for i in range(reps):
                o_, s_, uu, vv, bb, _ = sets[i % ncopies]
                ctx.check(lib.aggmg_smooth_dev(ctx.handle, o_.handle, s_.handle, _ptr(uu), _ptr(bb), alpha, per_launch, _ptr(vv)))
        return run

    def rot_resid(ncopies):
        def run(reps):
            for i in range(reps):
                o_, _, uu, _, bb, rr = sets[i % ncopies]
                ctx.check(lib.aggmg_residual_dev(ctx.handle, o_.handle, _ptr(uu), _ptr(bb), _ptr(rr)))
        return run

    def comp_frac(entry, o_, s_, what, ms_per_launch):
        """`frac`: compulsory bytes of ONE launch (every array once, as stored) / its duration / 8 TB/s"""
        rw = mg.smoother_launch_bytes(o_, s_, what)
        entry["compulsory_bytes_per_launch"] = sum(rw)
        entry["frac"] = sum(rw) / (ms_per_launch * 1e-3) / 1e9 / HBM_PEAK_GBS
        return entry

    def with_traffic(entry, key, ms_per_launch):
        comp_frac(entry, op, S if key.startswith("sweeps") else None, "sweeps" if key.startswith("sweeps") else "residual",
                  ms_per_launch)
        t = _traffic(f"smoother_{key}", "")
        if isinstance(t, dict):
            entry["traffic"] = t["hbm_bytes"]
            entry["physical_frac"] = t["hbm_bytes"] / (ms_per_launch * 1e-3) / 1e9 / HBM_PEAK_GBS
            entry["physical_frac_profile_mean"] = t["hbm_bytes"] / (t["ms_profile_mean"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        return entry

    for ncopies, suffix in ((1, ""), (3, "_rotating3")):
        for label, per_launch in (("sweeps_1_per_launch", 1), ("sweeps_4_per_launch", 4), ("sweeps_8_per_launch", 8)):
            reps = 96 // per_launch
            fn = rot_sweeps(per_launch, ncopies)
            fn(3)
            dt = _time_loop(ctx, fn, reps)
            nsw = reps * per_launch
            out[label + suffix] = with_traffic(
                {"dof_updates_per_s": N * nsw / dt, "us_per_sweep": 1e6 * dt / nsw, "algorithmic_GBs": S_bytes * nsw / dt / 1e9,
                 "frac_of_8TBs": S_bytes * nsw / dt / 1e9 / HBM_PEAK_GBS,
                 "data": "3 operator copies in turn (HBM)" if ncopies > 1 else "same operator every launch (Infinity-Cache assisted)"},
                f"{label}_dg_log2n20_copies{ncopies}", 1e3 * dt / reps)
        fn = rot_resid(ncopies)
        fn(3)
        dt = _time_loop(ctx, fn, 96)
        out["residual" + suffix] = with_traffic(
            {"us_per_residual": 1e6 * dt / 96, "algorithmic_GBs": R_bytes * 96 / dt / 1e9,
             "frac_of_8TBs": R_bytes * 96 / dt / 1e9 / HBM_PEAK_GBS,
             "data": "3 operator copies in turn (HBM)" if ncopies > 1 else "same operator every launch (Infinity-Cache assisted)"},
            f"residual_dg_log2n20_copies{ncopies}", 1e3 * dt / 96)
    for o_, s_, *_ in sets[1:]:
        del o_, s_
    del sets
    out["workload"] = f"config 2: DG n=2^20 p={args.p}, block-Jacobi m={args.p + 1}, N={N}, nnz(A)={nnzA}"
    # the generic CSR kernels (unstructured operators) on the same matrix: fused point-Jacobi sweep and CSR
    # residual, int32 indices + fp64 values actually read
    op2 = mg.DeviceOperator(U.stiffness_csc(0), _lib.OP_STIFFNESS, ctx)
    J = mg.JacobiSmoother(op2, ctx, detect=False)
    Sj_bytes = 12 * nnzA + 4 * (N + 1) + 8 * N + 24 * N
    # (the DG operator is banded: the generic path keeps the x window of a row block in LDS and runs up to four
    # sweeps per launch -- csr_band_kernel; AGGMG_CSR_BAND=0 falls back to one csr_stream_kernel launch per sweep)
    for per_launch, key in ((1, "generic_csr_point_jacobi"), (3, "generic_csr_point_jacobi_3_per_launch"),
                            (4, "generic_csr_point_jacobi_4_per_launch")):
        reps = 96 // per_launch
        fn = sweeps(op2.handle, J.handle, per_launch, u, v, b)
        fn(2)
        dt = _time_loop(ctx, fn, reps)
        nsw = reps * per_launch
        out[key] = comp_frac({"us_per_sweep": 1e6 * dt / nsw, "algorithmic_GBs": Sj_bytes * nsw / dt / 1e9,
                              "frac_of_8TBs": Sj_bytes * nsw / dt / 1e9 / HBM_PEAK_GBS}, op2, J, "sweeps", 1e3 * dt / reps)
    fn = resid(op2.handle, u, b, r)
    fn(1)
    dt = _time_loop(ctx, fn, 100)
    out["generic_csr_residual"] = comp_frac({"us_per_residual": 1e6 * dt / 100, "algorithmic_GBs": R_bytes * 100 / dt / 1e9,
                                             "frac_of_8TBs": R_bytes * 100 / dt / 1e9 / HBM_PEAK_GBS}, op2, None, "residual",
                                            1e3 * dt / 100)
    del op, op2, S, J, U
    # CG p=4, n=2^20 (config 5's fine-level operator at 1/16 size): point-Jacobi through the chain kernel
    # (element lists given) and through the generic CSR kernel (operator only), smoother + residual
    C = UniformCgDgHierarchy(n, ps=(4,))
    A = C.A[0]
    Nc, nnzc = A.shape[0], A.nnz
    Sc = 12 * nnzc + 4 * (Nc + 1) + 8 * Nc + 24 * Nc
    Rc = 12 * nnzc + 4 * (Nc + 1) + 24 * Nc
    bc, uc, vc, rc = ctx.to_device(C.rhs()), ctx.to_device(np.zeros(Nc)), ctx.alloc(Nc), ctx.alloc(Nc)
    cg = {"workload": f"CG n=2^20 p=4 point-Jacobi, N={Nc}, nnz(A)={nnzc}"}
    for label, elems in (("chain", C.element_nodes(0)), ("generic_csr", None)):
        opc = mg.DeviceOperator(A, _lib.OP_STIFFNESS, ctx)
        Jc = mg.JacobiSmoother(opc, ctx, elems, detect=(elems is not None))   # "generic_csr": pattern detection off
        for per_launch in (1, 3):
            reps = 99 // per_launch
            fn = sweeps(opc.handle, Jc.handle, per_launch, uc, vc, bc)
            fn(2)
            dt = _time_loop(ctx, fn, reps)
            nsw = reps * per_launch
            cg[f"{label}_sweeps_{per_launch}_per_launch"] = comp_frac({
                "us_per_sweep": 1e6 * dt / nsw, "algorithmic_GBs": Sc * nsw / dt / 1e9,
                "frac_of_8TBs": Sc * nsw / dt / 1e9 / HBM_PEAK_GBS}, opc, Jc, "sweeps",
                1e3 * dt / (reps if label == "chain" else nsw))
        fn = resid(opc.handle, uc, bc, rc)
        fn(1)
        dt = _time_loop(ctx, fn, 100)
        cg[f"{label}_residual"] = comp_frac({"us_per_residual": 1e6 * dt / 100, "algorithmic_GBs": Rc * 100 / dt / 1e9,
                                             "frac_of_8TBs": Rc * 100 / dt / 1e9 / HBM_PEAK_GBS}, opc, None, "residual", 1e3 * dt / 100)
        del opc, Jc
    out["cg_p4_point_jacobi"] = cg
    # the element Schwarz smoothers of cg_smoother (src/smoother.jl:104-134) on the same operator: fused chain kernel
    # (element lists in mesh order) against the generic kernels (same blocks listed in another order: residual pass,
    # batched block apply with atomics, update).  Algorithmic bytes per sweep: the residual's CSR pass + the element
    # inverses (8 (p+1)^2 per element) + gather / scatter / update of the vectors
    sw = {"workload": f"CG n=2^20 p=4 element Schwarz, N={Nc}, {n} overlapping blocks of 5"}
    Ssw = Rc + 8 * 25 * n + 32 * Nc
    el = C.element_nodes(0)
    perm = np.random.default_rng(0).permutation(el.shape[1])
    for kind, cls in (("additive", mg.AdditiveSchwarzSmoother), ("hybrid", mg.HybridSchwarzSmoother)):
        for label, lists in (("chain", el), ("generic", np.ascontiguousarray(el[:, perm]))):
            opc = mg.DeviceOperator(A, _lib.OP_STIFFNESS, ctx)
            Sw = cls(opc, lists, ctx)
            assert Sw.structured == (label == "chain")
            for per_launch in (1, 3):
                reps = 60 // per_launch
                fn = sweeps(opc.handle, Sw.handle, per_launch, uc, vc, bc)
                fn(2)
                dt = _time_loop(ctx, fn, reps)
                nsw = reps * per_launch
                sw[f"{kind}_{label}_sweeps_{per_launch}_per_launch"] = comp_frac({
                    "us_per_sweep": 1e6 * dt / nsw, "algorithmic_GBs": Ssw * nsw / dt / 1e9,
                    "frac_of_8TBs": Ssw * nsw / dt / 1e9 / HBM_PEAK_GBS}, opc, Sw, "sweeps",
                    1e3 * dt / (reps if label == "chain" else nsw))
            del opc, Sw
    out["cg_p4_element_schwarz"] = sw
    return out


def ragged_bench(mg, ctx, args, nPre, nPost, alpha):
    """the fused kernels on levels whose agglomerates differ in size (parent / first-child maps, an
    agglomerate cut by a tile boundary restricted by both tiles with atomic adds) timed at size, beside a
    uniform-ratio hierarchy with the same number of fine elements and similar coarsening (ratios 4, 4, 4: the
    ragged levels average 4 sub-elements per agglomerate)."""
    from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy, build_device_hierarchy, build_device_ragged_hierarchy
    E = args.ragged_log2_elems
    n = 2 ** E
    out = {}
    try:
        t0 = time.perf_counter()
        H, b, info = build_device_ragged_hierarchy(n, ctx, p=args.p, seed=1)
        ctx.synchronize()
        out["setup_s"] = time.perf_counter() - t0
        N = len(b)
        kinds = H.level_kinds()
        bd = ctx.to_device(b)
        st = [ctx.to_device(np.zeros(N)), ctx.alloc(N)]

        def run(reps, HH=H):
            for _ in range(reps):
                HH.vcycle_dev(st[0], bd, st[1], nPre, nPost, alpha)
                st[0], st[1] = st[1], st[0]

        run(args.warmup)
        dt = _time_loop(ctx, run, args.steps)
        out.update({"workload": f"DG p={args.p} n=2^{E} on a perturbed mesh -> 3 agglomerated levels, agglomerate sizes drawn from "
                                f"{{2,..,6}} (elements per level {info['elements']}); level kernels {kinds}",
                    "ms_per_step": 1e3 * dt / args.steps, "value": N * (nPre + nPost) * args.steps / dt, "unit": "DoF-updates/s"})
        H.free()
        U = UniformDgAggHierarchy(n, p=args.p, pAgg=1, ratios=(4, 4, 4))
        Hu = build_device_hierarchy(U, ctx)
        bd = ctx.to_device(U.rhs())
        st[0], st[1] = ctx.to_device(np.zeros(N)), ctx.alloc(N)
        run(args.warmup, Hu)
        dtu = _time_loop(ctx, lambda reps: run(reps, Hu), args.steps)
        out["uniform_ratio_4_4_4_ms_per_step"] = 1e3 * dtu / args.steps
        out["ragged_over_uniform"] = dt / dtu
        Hu.free()
    except Exception as e:
        out["error"] = repr(e)
    return out


def _roofline(kernel, comp_rw, ms_per_launch, launches, survey_bytes, traffic, prof_ms, note):
    """the bench line's roofline object: `achieved` / `frac` from the COMPULSORY bytes of the launch (every array it reads or
    writes counted once, as stored: aggmg_hier_launch_bytes) over the launch's HIP-event duration -- a fraction of the
    8 TB/s HBM roofline, <= 1 by construction; physical_frac from the PMC traffic of the same launch role (what the memory
    system moved, halo re-reads included); frac_survey_model from the SURVEY 8(d) CSR byte model (not a bound)."""
    sec = ms_per_launch * 1e-3
    comp = comp_rw[0] + comp_rw[1]
    achieved = comp / sec / 1e9
    return {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "frac_basis": "compulsory bytes (arrays of the launch, each once) / HIP-event time / peak",
            "compulsory_bytes_per_launch": comp, "compulsory_read_bytes": comp_rw[0], "compulsory_write_bytes": comp_rw[1],
            "traffic": traffic,
            "physical_GBs": (traffic / sec / 1e9) if traffic else None,
            "physical_frac": (traffic / sec / 1e9 / HBM_PEAK_GBS) if traffic else None,
            # the same bytes over the profile's own mean duration of this launch role (profiles/*_roles.md): the figure a
            # reader of profiles/ recomputes
            "physical_frac_profile_mean": (traffic / (prof_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic and prof_ms else None,
            "traffic_over_compulsory": (traffic / comp) if traffic else None,
            "frac_survey_model": survey_bytes / sec / 1e9 / HBM_PEAK_GBS, "survey_model_bytes_per_launch": survey_bytes,
            "ms_per_launch": ms_per_launch, "launches_timed": launches, "note": note}


def _coarse_roofline(tag, coarse_ms):
    """the kernel furthest below its roofline: HBM bytes of the coarsest solve's launches (PMC, profiles/)
    over its event time in this run -- a latency-bound chain of log2(n) levels, reported so that it is not hidden"""
    t = _traffic("coarse", tag)
    if not t or not coarse_ms:
        return {}
    return {"traffic": t, "physical_GBs": t / (coarse_ms * 1e-3) / 1e9,
            "physical_frac": t / (coarse_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}


def _traffic(role, tag):
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tfile):
        return None
    try:
        return json.load(open(tfile)).get(f"{role}_{tag}" if tag else role)
    except Exception:
        return None


def cg_bench(mg, ctx, args, nPre, nPost, alpha):
    """BASELINE config 5's realisable shape at its stated size on ONE GPU: CG p=4 -> 2 -> 1
    (point-Jacobi, Galerkin operators) -> DG p=0, V(3,3); the CG levels run the fused chain kernel."""
    from agglomerationmultigrid1d_amd.uniform import UniformCgDgHierarchy, build_device_cg_hierarchy
    E = args.cg_log2_elems
    n = 2 ** E
    t0 = time.perf_counter()
    U = UniformCgDgHierarchy(n, ps=(4, 2, 1))
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    H = build_device_cg_hierarchy(U, ctx)
    ctx.synchronize()
    t_lib = time.perf_counter() - t0
    kinds = H.level_kinds()
    N = U.A[0].shape[0]
    nnz0 = U.A[0].nnz
    bm = U.algorithmic_bytes(nPre, nPost)
    b = ctx.to_device(U.rhs())
    # the smoother BASELINE config 5 names ("block-GS"): red-black element Gauss-Seidel on the CG levels, a labelled
    # extension (the reference has no Gauss-Seidel smoother) -- same hierarchy, alpha = 1
    gs = {}
    try:
        t0 = time.perf_counter()
        Hg = build_device_cg_hierarchy(U, ctx, smoother="blockGS")
        ctx.synchronize()
        gs["setup_library_s"] = time.perf_counter() - t0
        assert Hg.level_kinds() == kinds
        ya, yb = ctx.to_device(np.zeros(N)), ctx.alloc(N)

        def run_gs(reps):
            nonlocal ya, yb
            for _ in range(reps):
                Hg.vcycle_dev(ya, b, yb, nPre, nPost, 1.0)
                ya, yb = yb, ya

        run_gs(args.warmup)
        dtg = _time_loop(ctx, run_gs, args.steps)
        _, ncyc, res = mg.multigrid_dev(Hg, ctx.to_device(np.zeros(N)), b, 200, 1e-8, check_every=2)
        gs.update({"ms_per_step": 1e3 * dtg / args.steps, "value": N * (nPre + nPost) * args.steps / dtg, "unit": "DoF-updates/s",
                   "multigrid_cycles_to_1e-8": ncyc, "final_residual": res[-1],
                   "note": "EXTENSION: red-black element Gauss-Seidel (even elements, then odd; post-smoothing reversed), "
                           "V(3,3), alpha = 1; checked against its own restatement only"})
        Hg.free()
        del Hg, ya, yb
    except Exception as e:
        gs["error"] = repr(e)
    del U
    xa, xb = ctx.to_device(np.zeros(N)), ctx.alloc(N)
    steps = args.steps
    state = [xa, xb]

    def run(reps):
        for _ in range(reps):
            H.vcycle_dev(state[0], b, state[1], nPre, nPost, alpha)
            state[0], state[1] = state[1], state[0]

    run(args.warmup)
    ctx.synchronize()
    ctx.profile_enable(2)
    dt = _time_loop(ctx, run, steps)
    ctx.profile_enable(False)
    dom = ctx.profile_collect()
    per = []
    for _ in range(steps):
        per.append(_time_loop(ctx, run, 1))
    ctx.profile_enable(1)
    run(steps)
    ctx.synchronize()
    ctx.profile_enable(False)
    prof = ctx.profile_collect()
    prof.update(dom)
    H.vcycles_dev(state[0], b, state[1], steps, nPre, nPost, alpha)
    dt_loop = _time_loop(ctx, lambda reps: H.vcycles_dev(state[0], b, state[1], reps, nPre, nPost, alpha), steps)
    vb = sum(l["vcycle"] for l in bm)
    kern = {f"{k}_L{l}": {"ms_per_launch": v[0] / v[1], "launches": v[1]} for (k, l), v in sorted(prof.items())}
    (dkind, dlevel), (dms, dcnt) = list(dom.items())[0]
    lm = bm[0]
    per_launch = nPre * lm['sweep'] + lm['residual'] + lm['restrict']
    comp_rw = H.launch_bytes(dlevel, "down")
    traffic = _traffic("chain_down_L0", f"cg_log2n{E}")
    prof_ms = _traffic("chain_down_L0", f"cg_log2n{E}_ms_profile_mean")
    coarse_ms = kern.get("coarse_L3", {}).get("ms_per_launch", 0.0)
    outer = {}
    try:
        z0 = ctx.alloc(N)            # the zero guess, made on the device (an upload of N zeros is not part of the loop)
        t2 = time.perf_counter()
        _, ncyc, res = mg.multigrid_dev(H, z0, b, 200, 1e-8, check_every=4)
        outer["multigrid"] = {"cycles": ncyc, "ms": 1e3 * (time.perf_counter() - t2), "final_residual": res[-1]}
        mg.multigrid_dev(H, z0, b, 2, 1e-30, check_every=1)         # (work vectors, first use of the variant)
        t2 = time.perf_counter()     # the reference's semantics: a check after every cycle, formed inside the chain kernel's launch
        _, ncyc1, res1 = mg.multigrid_dev(H, z0, b, 200, 1e-8, check_every=1)
        dt1 = 1e3 * (time.perf_counter() - t2)
        outer["multigrid_check_every_cycle"] = {"cycles": ncyc1, "ms": dt1, "ms_per_cycle": dt1 / max(ncyc1, 1),
                                                "final_residual": res1[-1]}
    except Exception as e:
        outer["error"] = repr(e)
    H.free()
    return {"workload": f"config 5 shape on 1 GPU: CG n=2^{E} p=4 -> 2 -> 1 -> DG p=0, point-Jacobi, V(3,3), alpha=2/3, "
                        f"N_fine={N}, nnz(A_1)={nnz0}; level kernels {kinds}",
            "value": N * (nPre + nPost) * steps / dt, "unit": "DoF-updates/s", "ms_per_step": 1e3 * dt / steps,
            "median_ms_per_step": 1e3 * statistics.median(per),
            "vcycles_loop_ms_per_cycle": 1e3 * dt_loop / steps,
            "achieved_algorithmic_GBs_vcycle": vb * steps / dt / 1e9, "frac_of_8TBs_vcycle": vb * steps / dt / 1e9 / HBM_PEAK_GBS,
            "roofline": _roofline("cgt_fused_kernel<4> fused_down level 1 (3 sweeps + residual + restriction)", comp_rw,
                                  dms / dcnt, dcnt, per_launch, traffic, prof_ms,
                                  "compulsory bytes: chain-form operator rows, transfer rows, vectors, each once"),
            "kernels": kern, "coarse_solve_ms_per_step": coarse_ms, "coarse_solve": _coarse_roofline(f"cg_log2n{E}", coarse_ms),
            "outer_solvers_to_1e-8": outer,
            "block_gs_extension": gs,
            "setup_s": t_gen + t_lib, "setup_generator_s": t_gen, "setup_library_s": t_lib}


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same arguments>` as a child
    process.  The parent stays off the GPU (no HIP call, no package import) so nothing is exec'ed
    from an initialised process; stdout of the child (rank 0's one JSON line) is relayed."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC: RCCL across processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // args.gpus)))
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    sys.stdout.write(proc.stdout)
    sys.stdout.flush()
    raise SystemExit(proc.returncode)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    nPre = nPost = 3
    alpha = 2.0 / 3.0
    if args.gpus > 1 and args.rehearse_threads:
        import threading
        from agglomerationmultigrid1d_amd import distributed as dist_mg
        group = dist_mg.ThreadGroup(args.gpus)
        errs = []

        def one(r):
            try:
                dist_mg.bench_main(args, r, args.gpus, 0, nPre, nPost, alpha, group=group)
            except BaseException as exc:      # a rank that dies must not leave the others in a barrier
                errs.append((r, exc))
                group.barrier.abort()

        ts = [threading.Thread(target=one, args=(r,)) for r in range(args.gpus)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if errs:
            raise SystemExit(f"rehearsal failed on ranks {[r for r, _ in errs]}: {errs[0][1]!r}")
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: start the ranks ourselves, one process per GPU, as a CHILD
        # torch.distributed.run (this process has not touched HIP and never will; no exec), relay rank 0's
        # JSON line and exit with the child's code
        return self_launch(args)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}")
    import agglomerationmultigrid1d_amd as mg
    from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy, build_device_hierarchy

    if world > 1 or os.environ.get("AGGMG_FORCE_DIST") == "1":   # (forced: smoke of the N > 1 code path on one rank)
        from agglomerationmultigrid1d_amd import distributed as dist_mg
        return dist_mg.bench_main(args, rank, world, local_rank, nPre, nPost, alpha)

    ctx = mg.Context(local_rank)

    def run_size(E, steps, warmup, profile, host_entry=False):
        """build the hierarchy at 2^E fine elements, run `steps` timed V-cycles and the same count
        through the multi-cycle entry point; host_entry: also time the host-pointer entry point
        (aggmg_vcycle: x0, b in and x out over PCIe on every call) -- the PCIe-inclusive rate"""
        n = 2 ** E
        t0 = time.perf_counter()
        U = UniformDgAggHierarchy(n, p=args.p, pAgg=1, ratios=(4, 2, 2))
        # the colptr / rowval / nzval arrays of the SparseMatrixCSC inputs, as a Julia caller would hand them over
        csc = ([U.stiffness_arrays(k) for k in range(U.nlevels)], [U.interpolation_arrays(k) for k in range(U.nlevels - 1)])
        t_gen = time.perf_counter() - t0          # the synthetic SparseMatrixCSC inputs (stands in for the reference's assembly)
        t0 = time.perf_counter()
        H = build_device_hierarchy(U, ctx, csc=csc)
        ctx.synchronize()
        t_lib = time.perf_counter() - t0          # everything the library does with them: upload + device set-up
        del csc
        bytes_model = U.algorithmic_bytes(nPre, nPost)
        N = U.levels[0]['m'] * U.levels[0]['ne']
        b_host = U.rhs()
        b = ctx.to_device(b_host)
        xa = ctx.to_device(np.zeros(N))
        xb = ctx.alloc(N)
        level_sizes = [lv['m'] * lv['ne'] for lv in U.levels]
        del U
        assert H.level_kinds() == ['fused_btd'] * (len(level_sizes) - 1) + ['coarsest'], "fused HIP kernels were not selected"
        state = [xa, xb]

        def run(reps):
            for _ in range(reps):
                H.vcycle_dev(state[0], b, state[1], nPre, nPost, alpha)
                state[0], state[1] = state[1], state[0]

        run(warmup)
        ctx.synchronize()
        if profile:
            ctx.profile_enable(2)   # the timed region carries events for the dominant kernel only
        dt = _time_loop(ctx, run, steps)
        prof = prof_dom = None
        if profile:
            ctx.profile_enable(False)
            prof_dom = ctx.profile_collect()
        # median of >= 20 repetitions, each timed on its own
        per = [_time_loop(ctx, run, 1) for _ in range(steps)]
        if profile:
            # per-kernel table: an untimed pass with events around every launch (an event pair costs
            # ~7 us of stream time, which would otherwise sit inside `value`)
            ctx.profile_enable(1)
            run(steps)
            ctx.synchronize()
            ctx.profile_enable(False)
            prof = ctx.profile_collect()
            prof.update(prof_dom)     # the dominant kernel keeps its in-region measurement
        # the same K cycles through the multi-cycle entry point (the loop body of multigrid(),
        # src/solvers.jl:124-126): consecutive cycles share one fused fine-level launch.  Reported
        # beside `value`, which stays K independent multigrid_v_cycle calls.
        H.vcycles_dev(state[0], b, state[1], steps, nPre, nPost, alpha)
        dt_loop = _time_loop(ctx, lambda reps: H.vcycles_dev(state[0], b, state[1], reps, nPre, nPost, alpha), steps)
        info = H.coarse_info()
        comp_rw = H.launch_bytes(0, "down")
        comp_all = {f"fused_{kd}_L{k}": sum(H.launch_bytes(k, kd)) for k in range(len(level_sizes) - 1) for kd in ("down", "up")}
        paired = {kd: H.paired_levels(nPre if kd == "down" else nPost, kd) for kd in ("down", "up")}
        for kd in ("down", "up"):     # a two-level launch carries the arrays of both levels (attributed to the finer one)
            for k in paired[kd]:
                comp_all[f"fused_{kd}_L{k}"] += comp_all.pop(f"fused_{kd}_L{k + 1}")
        # the device-resident outer loops, outside the timed region: multigrid()
        # (src/solvers.jl:116-139, residual check every 8 cycles) and CG preconditioned with
        # ldiv! to ||A x - b|| < 1e-8 ||b|| from a zero guess
        outer = {}
        try:
            z0 = ctx.alloc(N)        # the zero guess, made on the device (an upload of N zeros is not part of the loop)
            t2 = time.perf_counter()
            _, ncyc, res = mg.multigrid_dev(H, z0, b, 400, 1e-8, check_every=8)
            outer["multigrid"] = {"cycles": ncyc, "ms": 1e3 * (time.perf_counter() - t2), "final_residual": res[-1]}
            # the reference's own semantics -- res / err after EVERY cycle (src/solvers.jl:124-131): the norms are formed
            # inside the fine-level launch that post-smooths cycle i and pre-smooths cycle i + 1
            mg.multigrid_dev(H, z0, b, 2, 1e-30, check_every=1)     # (work vectors, first use of the variant)
            t2 = time.perf_counter()
            _, ncyc1, res1 = mg.multigrid_dev(H, z0, b, 400, 1e-8, check_every=1)
            dt1 = 1e3 * (time.perf_counter() - t2)
            outer["multigrid_check_every_cycle"] = {"cycles": ncyc1, "ms": dt1, "ms_per_cycle": dt1 / max(ncyc1, 1),
                                                    "final_residual": res1[-1]}
            t2 = time.perf_counter()
            _, nit, resp = mg.pcg(H, b_host, maxiter=100, tol=1e-8)
            outer["pcg_ldiv"] = {"iterations": nit, "ms_incl_h2d_d2h": 1e3 * (time.perf_counter() - t2),
                                 "final_residual": resp[-1]}
        except Exception as e:  # reported, never fatal for the bench line
            outer["error"] = repr(e)
        pcie = None
        if host_entry:
            try:
                xh = mg.multigrid_v_cycle(H, np.zeros(N), b_host)         # (first call: staging buffers, device vectors)
                th = []
                for _ in range(5):
                    t2 = time.perf_counter()
                    xh = mg.multigrid_v_cycle(H, xh, b_host)
                    th.append(time.perf_counter() - t2)
                tm = statistics.median(th)
                # the same call on arrays the caller page-locked once and a result array it reuses (aggmg_host_register /
                # aggmg_host_alloc): three DMA transfers, nothing staged
                xp, bp, yp = ctx.pinned_empty(N), ctx.pin(b_host.copy()), ctx.pinned_empty(N)
                xp[:] = xh
                mg.multigrid_v_cycle(H, xp, bp, out=yp)
                tp = []
                for _ in range(5):
                    t2 = time.perf_counter()
                    mg.multigrid_v_cycle(H, xp, bp, out=yp)
                    tp.append(time.perf_counter() - t2)
                    xp, yp = yp, xp
                tpm = statistics.median(tp)
                tl = []
                for _ in range(5):      # ldiv!(y, H, b): zero initial guess (x0 = NULL), two transfers
                    t2 = time.perf_counter()
                    mg.ldiv(yp, H, bp)
                    tl.append(time.perf_counter() - t2)
                tlm = statistics.median(tl)
                ctx.unpin(bp)
                pcie = {"value": N * (nPre + nPost) / tm, "unit": "DoF-updates/s", "ms_per_call": 1e3 * tm,
                        "pinned": {"value": N * (nPre + nPost) / tpm, "ms_per_call": 1e3 * tpm, "effective_GBs": 3 * 8 * N / tpm / 1e9,
                                   "note": "x0, b, x_out page-locked once by the caller (Context.pin / pinned_empty), result array reused",
                                   "ldiv_ms_per_call": 1e3 * tlm,
                                   "ldiv_note": "ldiv!(y, H, b) on the same arrays: the zero initial guess is not sent (x0 = NULL)"},
                        "host_bytes_per_call": 3 * 8 * N, "effective_GBs": 3 * 8 * N / tm / 1e9,
                        "note": "multigrid_v_cycle(H, x0, b) on host arrays (aggmg_vcycle): x0, b copied in and a NEW result "
                                "array copied out on every call, pageable memory staged through pinned chunks by 4 "
                                "threads; median of 5 calls.  Never the reported `value`: callers that loop keep the "
                                "vectors on the device (aggmg_vcycle_dev / aggmg_multigrid_dev)"}
            except Exception as e:
                pcie = {"error": repr(e)}
        H.free()
        return dict(N=N, dt=dt, per=per, dt_loop=dt_loop, prof=prof, prof_dom=prof_dom,
                    bytes_model=bytes_model, level_sizes=level_sizes, t_gen=t_gen, t_lib=t_lib,
                    coarse_info=info, outer=outer, pcie=pcie, comp_rw=comp_rw, comp_all=comp_all, paired=paired)

    R = run_size(args.log2_elems, args.steps, args.warmup, True)
    N, dt, dt_loop, prof, bytes_model = R["N"], R["dt"], R["dt_loop"], R["prof"], R["bytes_model"]
    level_sizes = R["level_sizes"]

    ms_per_step = 1e3 * dt / args.steps
    value = N * (nPre + nPost) * args.steps / dt
    vcycle_bytes = sum(l['vcycle'] for l in bytes_model)
    dom = list(R["prof_dom"].items())[0]   # the fine-level fused-down launch, timed inside the region
    (dkind, dlevel), (dms, dcnt) = dom
    lm = bytes_model[dlevel]
    per_launch = {"fused_down": nPre * lm['sweep'] + lm['residual'] + lm['restrict'],
                  "fused_up": nPost * lm['sweep'] + lm['prolong']}.get(dkind, lm['sweep'])
    traffic = _traffic(f"{dkind}_L{dlevel}", f"dg_log2n{args.log2_elems}")
    prof_ms = _traffic(f"{dkind}_L{dlevel}", f"dg_log2n{args.log2_elems}_ms_profile_mean")
    kern_ms = {f"{k}_L{l}": {"ms_per_launch": v[0] / v[1], "launches": v[1]} for (k, l), v in sorted(prof.items())}
    for k_, v_ in kern_ms.items():      # every fused launch of the cycle against the roofline (compulsory bytes, as `roofline.frac`)
        if k_ in R["comp_all"]:
            v_["compulsory_bytes"] = R["comp_all"][k_]
            v_["frac"] = R["comp_all"][k_] / (v_["ms_per_launch"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            # PMC bytes of the same launch role (profiles/traffic.json; a two-level launch is filed as pair_*)
            kd_, lv_ = k_.split("_")[1], int(k_.rsplit("L", 1)[1])
            role_ = ("pair_" if lv_ in R["paired"].get(kd_, []) else "fused_") + f"{kd_}_L{lv_}"
            t_ = _traffic(role_, f"dg_log2n{args.log2_elems}")
            if t_:
                v_["traffic"] = t_
                v_["physical_frac"] = t_ / (v_["ms_per_launch"] * 1e-3) / 1e9 / HBM_PEAK_GBS
    coarse_dev = [v["ms_per_launch"] for k, v in kern_ms.items() if k.startswith("coarse_")]
    coarse_step_ms = coarse_dev[0] if coarse_dev else 0.0
    out = {
        "metric": "fine_level_dof_updates_per_s_per_vcycle",
        "value": value,
        "unit": "DoF-updates/s",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "median_ms_per_step": 1e3 * statistics.median(R["per"]),
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
                   "parallelism": "single GPU",
                   "sizes_by_leg": {"value / roofline": f"2^{args.log2_elems} fine elements",
                                    "config3_2p22": f"2^{args.also_log2_elems}" if args.also_log2_elems else None,
                                    "cpu_baseline": None if args.no_cpu_baseline else f"2^{args.cpu_log2_elems or args.log2_elems} fine elements",
                                    "config5": f"2^{args.cg_log2_elems}" if args.cg_log2_elems else None,
                                    "smoother_only (config 2)": "2^20", "ragged": f"2^{args.ragged_log2_elems}" if args.ragged_log2_elems else None}},
        "achieved_algorithmic_GBs_vcycle": vcycle_bytes * args.steps / dt / 1e9,
        "vcycles_loop": {"value": N * (nPre + nPost) * args.steps / dt_loop, "unit": "DoF-updates/s",
                         "ms_per_cycle": 1e3 * dt_loop / args.steps,
                         "note": f"aggmg_vcycles_dev({args.steps} cycles): same arithmetic as {args.steps} separate "
                                 "V-cycles (bitwise), post-smoothing of cycle i and pre-smoothing of cycle i+1 in one "
                                 "fine-level launch"},
        "outer_solvers_to_1e-8": R["outer"],
        "coarse_solve": dict(R["coarse_info"], **_coarse_roofline(f"dg_log2n{args.log2_elems}", coarse_step_ms)),
        # SURVEY 8d times the coarsest solve separately: the `coarse` entry of the per-kernel table
        # (HIP events, untimed second pass)
        "coarse_solve_ms_per_step": coarse_step_ms,
        "value_excl_coarse_solve": N * (nPre + nPost) / max(1e-3 * (ms_per_step - coarse_step_ms), 1e-12),
        "roofline": _roofline(f"btd_fused_kernel<{args.p + 1},cmp> {dkind} level {dlevel + 1}", R["comp_rw"], dms / dcnt, dcnt,
                              per_launch, traffic, prof_ms,
                              "frac: compulsory bytes of the launch (packed symmetric block inverses, coupling rows / columns, "
                              "diagonal blocks for the residual, transfer rows, vectors; each array once) / HIP-event duration "
                              "/ 8 TB/s.  physical_frac: PMC bytes of this launch role (2*FETCH_SIZE + WRITE_SIZE, profiles/) "
                              "over the same duration.  frac_survey_model: SURVEY 8d CSR model with every sweep re-reading the "
                              "operator -- exceeds 1 because the fused kernel reads the operator once for 3 sweeps + residual "
                              "+ restriction; kept for continuity with r01-r03, not a bound"),
        "kernels": kern_ms,
        "two_level_launches": {kd: [f"levels {k + 1}+{k + 2}" for k in v] for kd, v in R["paired"].items()},
        "setup_s": R["t_gen"] + R["t_lib"], "setup_generator_s": R["t_gen"], "setup_library_s": R["t_lib"],
    }
    if args.also_log2_elems and args.also_log2_elems != args.log2_elems:
        R2 = run_size(args.also_log2_elems, args.steps, args.warmup, False, host_entry=True)
        out[f"config3_2p{args.also_log2_elems}"] = {
            "workload": f"config 3: same hierarchy at 2^{args.also_log2_elems} fine elements (N_fine={R2['N']})",
            "value": R2["N"] * (nPre + nPost) * args.steps / R2["dt"], "unit": "DoF-updates/s",
            "ms_per_step": 1e3 * R2["dt"] / args.steps, "median_ms_per_step": 1e3 * statistics.median(R2["per"]),
            "vcycles_loop_ms_per_cycle": 1e3 * R2["dt_loop"] / args.steps,
            "setup_s": R2["t_gen"] + R2["t_lib"], "setup_generator_s": R2["t_gen"], "setup_library_s": R2["t_lib"],
            "outer_solvers_to_1e-8": R2["outer"], "pcie_inclusive": R2["pcie"]}
    if args.ragged_log2_elems:
        out[f"ragged_2p{args.ragged_log2_elems}"] = ragged_bench(mg, ctx, args, nPre, nPost, alpha)
    if args.cg_log2_elems:
        out[f"config5_2p{args.cg_log2_elems}_1gpu"] = cg_bench(mg, ctx, args, nPre, nPost, alpha)
    if not args.no_smoother_bench:
        out["smoother_only"] = smoother_bench(mg, ctx, args, alpha)
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, nPre, nPost, alpha)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
