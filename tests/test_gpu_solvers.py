"""Device-resident outer solver loops against the CPU oracle:
aggmg_multigrid_dev (src/solvers.jl:116-139), aggmg_smoother_solve_dev (src/solvers.jl:189-213),
the reductions behind their stopping tests, and the ldiv!-preconditioned CG extension.

Tolerances: norms / dot products 1e-13 relative (different summation tree than numpy);
residual histories: the iterates agree to 1e-12 * ||x||, so a residual norm agrees to
1e-12 * ||A|| ||x|| absolute -- stated as atol relative to ||b|| -- and the iteration counts must
be equal."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mg():
    import agglomerationmultigrid1d_amd as mg
    mg.default_context()
    return mg


def test_dot_and_norm(oracle, mg):
    ctx = mg.default_context()
    for n in (0, 1, 7, 255, 256, 257, 4097, 1_000_003):
        x = oracle.splitmix_normal(n, 3)
        y = oracle.splitmix_normal(n, 4)
        dx, dy = ctx.to_device(x), ctx.to_device(y)
        d = mg.dot(dx, dy)
        ref = float(np.dot(x, y))
        assert abs(d - ref) <= 1e-13 * max(np.linalg.norm(x) * np.linalg.norm(y), 1e-300) + 0.0
        assert abs(mg.norm2(dx) - np.linalg.norm(x)) <= 1e-13 * max(np.linalg.norm(x), 1e-300)
        assert mg.dot(dx, dy) == d                       # reproducible run to run


def test_multigrid_device_loop(oracle, mg):
    o = oracle
    Ho, b = o.build_dg_agg_hierarchy(64, p=3, pAgg=1, nAgg=3, first=4)
    H = mg.MeshHierarchy.from_reference(Ho)
    nb = np.linalg.norm(b)
    x0 = np.zeros(len(b))
    xo, ito, reso, _ = o.multigrid(Ho, x0, b, 200, 1e-10)
    xg, itg, resg, errg = mg.multigrid(H, x0, b, 200, 1e-10, exact=False)
    assert itg == ito and errg == []
    assert np.allclose(resg, reso, rtol=1e-8, atol=1e-11 * nb)
    assert np.linalg.norm(Ho.mStiffness[0] @ (xg - xo)) <= 1e-11 * nb
    # checks every 4 cycles: every 4th entry of the same history (same arithmetic), cycle count
    # rounded up to the check that first meets the tolerance
    xg4, it4, res4, _ = mg.multigrid(H, x0, b, 200, 1e-10, exact=False, check_every=4)
    k = len(res4)
    assert it4 == 4 * k and 4 * (k - 1) < ito <= 4 * k
    full = mg.multigrid(H, x0, b, it4, 0.0, exact=False)[2]
    assert np.allclose(res4, full[3::4], rtol=1e-10, atol=1e-13 * nb)
    # random initial guess, capped iteration count
    x0r = o.splitmix_normal(len(b), 0)
    _, ito, reso, _ = o.multigrid(Ho, x0r, b, 7, 1e-30)
    _, itg, resg, _ = mg.multigrid(H, x0r, b, 7, 1e-30, exact=False)
    assert itg == ito == 7
    assert np.allclose(resg, reso, rtol=1e-8, atol=1e-11 * max(nb, reso[0]))
    # maxiter = 0: the reference returns its initial `x = zeros(length(x0))` (src/solvers.jl:119)
    ctx = H.ctx
    dx, n, res = mg.multigrid_dev(H, ctx.to_device(x0r), ctx.to_device(b), 0, 1e-10)
    assert n == 0 and res == [] and np.array_equal(dx.download(), np.zeros(len(x0r)))


def test_multigrid_device_loop_generic_levels(oracle, mg):
    """CG hierarchy (generic CSR kernels, host or device coarsest solve)"""
    o = oracle
    Ho, b = o.build_cg_hierarchy(32, ps=(4, 2, 1), nDG=1)
    H = mg.MeshHierarchy.from_reference(Ho)
    x0 = np.zeros(len(b))
    xo, ito, reso, _ = o.multigrid(Ho, x0, b, 60, 1e-9)
    xg, itg, resg, _ = mg.multigrid(H, x0, b, 60, 1e-9, exact=False)
    assert itg == ito
    assert np.allclose(resg, reso, rtol=1e-7, atol=1e-11 * np.linalg.norm(b))


def test_smoother_solve_device_loop(oracle, mg):
    o = oracle
    Ho, b = o.build_dg_agg_hierarchy(48, p=3, pAgg=1, nAgg=1, first=4)
    A, dg = Ho.mStiffness[0], Ho.mMeshes[0]
    u0 = np.zeros(len(b))
    So = o.dg_smoother(dg, A, 'blockJac')
    Sg = mg.dg_smoother(dg, A, 'blockJac')
    xo, ito, reso, _ = o.iterative_smoother_solve(A, So, u0, b, maxiter=40, tol=1e-30, alpha=2.0 / 3.0)
    xg, itg, resg, errg = mg.iterative_smoother_solve(A, Sg, u0, b, maxiter=40, tol=1e-30, alpha=2.0 / 3.0, exact=False)
    assert itg == ito == 40 and errg == []
    assert np.allclose(resg, reso, rtol=1e-10, atol=1e-12 * np.linalg.norm(b))
    assert np.linalg.norm(xg - xo) <= 1e-12 * np.linalg.norm(xo)
    # maxiter = 0: the reference returns its initial `x = zeros(length(x0))` (src/solvers.jl:192), whatever x0 holds
    xz, itz, resz, _ = mg.iterative_smoother_solve(A, Sg, np.ones(len(b)), b, maxiter=0, tol=1e-30, alpha=2.0 / 3.0, exact=False)
    assert itz == 0 and resz == [] and not xz.any()
    x5, it5, res5, _ = mg.iterative_smoother_solve(A, Sg, u0, b, maxiter=40, tol=1e-30, alpha=2.0 / 3.0, exact=False,
                                                  check_every=5)
    assert it5 == 40 and len(res5) == 8
    assert np.allclose(res5, np.asarray(resg)[4::5], rtol=1e-10)
    # point Jacobi on a CG operator (generic kernels)
    Hc, bc = o.build_cg_hierarchy(32, ps=(2, 1), nDG=1)
    Ac, cgm = Hc.mStiffness[0], Hc.mMeshes[0]
    xo, ito, reso, _ = o.iterative_smoother_solve(Ac, o.cg_smoother(cgm, Ac, 'jac'), np.zeros(len(bc)), bc,
                                                  maxiter=25, tol=1e-30, alpha=0.5)
    xg, itg, resg, _ = mg.iterative_smoother_solve(Ac, mg.cg_smoother(cgm, Ac, 'jac'), np.zeros(len(bc)), bc,
                                                   maxiter=25, tol=1e-30, alpha=0.5, exact=False)
    assert itg == ito
    assert np.allclose(resg, reso, rtol=1e-10, atol=1e-12 * np.linalg.norm(bc))


@pytest.mark.parametrize("ne", [48, 3000])    # one tile / many tiles
def test_smoother_solve_checkpoints_inside_the_sweep_launches(oracle, mg, ne):
    """iterative_smoother_solve's test after every sweep (src/solvers.jl:198-206) formed INSIDE multi-sweep launches of the
    fused block-Jacobi kernel (AGGMG_OPT_MG_CHECKPOINT, default on) against one sweep launch + one residual launch per
    iteration: the same iterates bit for bit, the same iteration counts -- also when the tolerance is met in the middle of a
    launch, which is then run again up to that sweep --, histories equal to round-off; check_every 1 / 2 / 5 / 9 with a
    maxiter that is not a multiple; with and without the error history; and against the oracle"""
    from agglomerationmultigrid1d_amd import _lib
    o = oracle
    Ho, b = o.build_dg_agg_hierarchy(ne, p=3, pAgg=1, nAgg=1, first=4)
    A, dg = Ho.mStiffness[0], Ho.mMeshes[0]
    u0 = o.splitmix_normal(len(b), 11)
    # a tolerance the iteration meets after a sweep count that is no multiple of the launch's sweeps
    _, _, r_probe, _ = o.iterative_smoother_solve(A, o.dg_smoother(dg, A, 'blockJac'), u0, b, maxiter=17, tol=1e-30, alpha=2.0 / 3.0)
    tol_mid = 0.5 * (r_probe[9] + r_probe[10]) / np.linalg.norm(b)
    cases = ((1, 23, 1e-30, True), (1, 40, tol_mid, False), (2, 23, 1e-30, False), (5, 23, tol_mid, True), (9, 40, 1e-30, False),
             (1, 1, 1e-30, True), (3, 2, 1e-30, False))
    out = {}
    for chk in (1, 0):
        ctx = mg.Context(0)
        ctx.set_option(_lib.OPT_MG_CHECKPOINT, chk)
        Sg = mg.dg_smoother(dg, A, 'blockJac', ctx=ctx)
        out[chk] = [mg.iterative_smoother_solve(A, Sg, u0, b, maxiter=mi, tol=tol, alpha=2.0 / 3.0, exact=ex, check_every=ce)
                    for ce, mi, tol, ex in cases]
    for (ce, mi, tol, ex), (xa, ia, ra, ea), (xb, ib, rb, eb) in zip(cases, out[1], out[0]):
        assert ia == ib and len(ra) == len(rb) and len(ea) == len(eb) == (len(ra) if ex else 0), (ce, mi)
        assert np.array_equal(xa, xb), (ce, mi)
        assert np.allclose(ra, rb, rtol=1e-10, atol=1e-13 * np.linalg.norm(b)) and np.allclose(ea, eb, rtol=1e-9, atol=1e-12), (ce, mi)
    assert out[1][1][1] == 11                         # stopped inside a launch: 11 sweeps, not the launch's full count
    xo, ito, reso, _ = o.iterative_smoother_solve(A, o.dg_smoother(dg, A, 'blockJac'), u0, b, maxiter=40, tol=tol_mid, alpha=2.0 / 3.0)
    xg, itg, resg, _ = out[1][1]
    assert itg == ito and np.allclose(resg, reso, rtol=1e-9, atol=1e-12 * np.linalg.norm(b))
    assert np.linalg.norm(xg - xo) <= 1e-11 * np.linalg.norm(xo)


@pytest.mark.parametrize("ne", [40, 3000])    # one tile / many tiles
def test_checkpoints_inside_the_chain_kernel_launches(oracle, mg, ne):
    """The same two loops on a CG hierarchy (CG p = 4 -> 2 -> 1 -> DG p = 0, point-Jacobi; the fine levels run the chain
    kernel): multigrid's test after every cycle inside the launch shared by consecutive cycles, iterative_smoother_solve's
    test after every sweep inside multi-sweep launches -- against the forms with separate residual launches
    (AGGMG_OPT_MG_CHECKPOINT = 0): iterates bit for bit (the checkpoint's x goes out through the level's permutation),
    equal counts, histories to round-off; and against the oracle."""
    from agglomerationmultigrid1d_amd import _lib
    o = oracle
    Ho, b = o.build_cg_hierarchy(ne, ps=(4, 2, 1), nDG=1)
    A, cgm = Ho.mStiffness[0], Ho.mMeshes[0]
    x0 = o.splitmix_normal(len(b), 3)
    _, _, rp, _ = o.iterative_smoother_solve(A, o.cg_smoother(cgm, A, 'jac'), x0, b, maxiter=12, tol=1e-30, alpha=0.5)
    tol_mid = 0.5 * (rp[8] + rp[9]) / np.linalg.norm(b)
    mg_cases = ((1, 25, 1e-7, True), (1, 6, 1e-30, False), (2, 7, 1e-30, True), (3, 25, 1e-5, False), (1, 1, 1e-30, False))
    sm_cases = ((1, 19, 1e-30, False), (1, 30, tol_mid, True), (4, 19, 1e-30, False), (3, 30, tol_mid, False), (1, 1, 1e-30, False))
    out = {}
    for chk in (1, 0):
        ctx = mg.Context(0)
        ctx.set_option(_lib.OPT_MG_CHECKPOINT, chk)
        H = mg.MeshHierarchy.from_reference(Ho, ctx=ctx)
        assert H.level_kinds()[0] == "fused_chain"
        runs = [mg.multigrid(H, x0, b, mi, tol, exact=ex, check_every=ce) for ce, mi, tol, ex in mg_cases]
        Sg = mg.cg_smoother(cgm, A, 'jac', ctx=ctx)
        runs += [mg.iterative_smoother_solve(A, Sg, x0, b, maxiter=mi, tol=tol, alpha=0.5, exact=ex, check_every=ce)
                 for ce, mi, tol, ex in sm_cases]
        out[chk] = runs
    for k, ((xa, ia, ra, ea), (xb, ib, rb, eb)) in enumerate(zip(out[1], out[0])):
        assert ia == ib and len(ra) == len(rb) and len(ea) == len(eb), k
        assert np.array_equal(xa, xb), k
        assert np.allclose(ra, rb, rtol=1e-10, atol=1e-13 * np.linalg.norm(b)) and np.allclose(ea, eb, rtol=1e-8, atol=1e-11), k
    assert out[1][len(mg_cases) + 1][1] == 10        # the smoother loop stopped inside a launch
    xo, ito, reso, _ = o.multigrid(Ho, x0, b, 25, 1e-7)
    xg, itg, resg, _ = out[1][0]
    assert itg == ito and np.allclose(resg, reso, rtol=1e-7, atol=1e-11 * np.linalg.norm(b))
    xo, ito, reso, _ = o.iterative_smoother_solve(A, o.cg_smoother(cgm, A, 'jac'), x0, b, maxiter=30, tol=tol_mid, alpha=0.5)
    xg, itg, resg, _ = out[1][len(mg_cases) + 1]
    assert itg == ito and np.allclose(resg, reso, rtol=1e-9, atol=1e-12 * np.linalg.norm(b))
    assert np.linalg.norm(xg - xo) <= 1e-11 * np.linalg.norm(xo)


@pytest.mark.parametrize("kind", ["dg", "cg"])
def test_checkpoints_with_no_pre_or_no_post_smoothing(oracle, mg, kind):
    """the checked loops at V(0, nPost), V(nPre, 0) and V(0, 0) -- the checkpoint then sits before the first or after the
    last sweep of the shared launch (or is all of it) -- and at uneven sweep counts: iterates bitwise, histories to
    round-off against the separate-launch form, on both kernel families"""
    from agglomerationmultigrid1d_amd import _lib
    o = oracle
    if kind == "dg":
        Ho, b = o.build_dg_agg_hierarchy(1536, p=3, pAgg=1, nAgg=2, first=4)
    else:
        Ho, b = o.build_cg_hierarchy(1500, ps=(4, 2, 1), nDG=1)
    x0 = o.splitmix_normal(len(b), 9)
    cases = ((0, 2), (2, 0), (0, 0), (1, 4), (4, 1))
    out = {}
    for chk in (1, 0):
        ctx = mg.Context(0)
        ctx.set_option(_lib.OPT_MG_CHECKPOINT, chk)
        H = mg.MeshHierarchy.from_reference(Ho, ctx=ctx)
        out[chk] = [mg.multigrid(H, x0, b, 5, 1e-30, exact=(k % 2 == 0), nPre=a, nPost=c, alpha=0.5) for k, (a, c) in enumerate(cases)]
    for case, (xa, ia, ra, ea), (xb, ib, rb, eb) in zip(cases, out[1], out[0]):
        assert ia == ib == 5 and len(ra) == len(rb) == 5, case
        assert np.array_equal(xa, xb), case
        assert np.allclose(ra, rb, rtol=1e-10, atol=1e-13 * np.linalg.norm(b)) and np.allclose(ea, eb, rtol=1e-8, atol=1e-11), case
    xo = x0
    for _ in range(5):
        xo = o.multigrid_v_cycle(Ho, xo, b, nPre=1, nPost=4, alpha=0.5)
    assert np.linalg.norm(Ho.mStiffness[0] @ (out[1][3][0] - xo)) < 1e-11 * np.linalg.norm(b)


def test_err_histories_are_formed_on_the_device(oracle, mg):
    """multigrid / iterative_smoother_solve return the reference's full 4-tuple (x, iter, res, err) with
    err[i] = ||x_i - A \\ b||_2 (src/solvers.jl:120,128 and :194,202) -- exact=True, the default of both mirrors -- and the
    direct solve behind it runs on the device where the fine operator is block-tridiagonal with blocks of at most 8 x 8
    (block cyclic reduction; DG p = 8 has 9 x 9 blocks: the library's host banded LU, factored once per hierarchy; the CG
    hierarchy falls back to a host sparse factorisation, once).
    Against the oracle's histories on the tests/dg_heirarchy_test.jl shape (n = 128, DG p = 8 -> 4 -> 2 -> 1,
    CDir = 1000 n): 1e-9 of the first error (the two direct solves agree to cond(A) eps ~ 1e-10 of ||u||)."""
    o = oracle
    Ho, b = o.build_dg_p_hierarchy(128, ps=(8, 4, 2, 1))
    H = mg.MeshHierarchy.from_reference(Ho)
    x0 = np.zeros(len(b))
    xo, ito, reso, erro = o.multigrid(Ho, x0, b, 100, 1e-10)
    xg, itg, resg, errg = mg.multigrid(H, x0, b, 100, 1e-10)
    assert itg == ito and len(errg) == len(erro) == ito
    assert np.allclose(errg, erro, rtol=1e-6, atol=1e-9 * erro[0])
    assert np.allclose(resg, reso, rtol=1e-6, atol=1e-11 * np.linalg.norm(b))
    assert H._direct_solver.where == "host banded LU"
    # DG p = 7 (8 x 8 blocks, the largest the cyclic reduction is instantiated for): the direct solve is the device's
    H7o, b7 = o.build_dg_p_hierarchy(64, ps=(7, 3, 1))
    H7 = mg.MeshHierarchy.from_reference(H7o)
    _, it7o, _, err7o = o.multigrid(H7o, np.zeros(len(b7)), b7, 100, 1e-10)
    _, it7, _, err7 = mg.multigrid(H7, np.zeros(len(b7)), b7, 100, 1e-10)
    assert it7 == it7o and np.allclose(err7, err7o, rtol=1e-6, atol=1e-9 * err7o[0])
    assert H7._direct_solver.where == "device"
    # vectors that stay in HBM: same histories, a DeviceVector comes back
    ctx = H.ctx
    xd, itd, resd, errd = mg.multigrid(H, ctx.to_device(x0), ctx.to_device(b), 100, 1e-10)
    assert isinstance(xd, mg.DeviceVector) and itd == itg and errd == errg and resd == resg
    assert np.array_equal(xd.download(), xg)
    # one error per check with check_every > 1
    x3, it3, res3, err3 = mg.multigrid(H, x0, b, 100, 1e-10, check_every=3)
    assert len(err3) == len(res3) == (it3 + 2) // 3
    assert np.allclose(err3[:-1], np.asarray(errg)[2::3][:len(err3) - 1], rtol=1e-9)
    # the stationary smoother loop (tests/dg_smoother_test.jl shape: n = 16, p = 2, CDir = 1000 n)
    Hs, bs = o.build_dg_p_hierarchy(16, ps=(2, 1))
    A, dg = Hs.mStiffness[0], Hs.mMeshes[0]
    u0 = np.zeros(len(bs))
    _, ito, reso, erro = o.iterative_smoother_solve(A, o.dg_smoother(dg, A, 'blockJac'), u0, bs, maxiter=60, tol=1e-30,
                                                    alpha=2.0 / 3.0)
    Sg = mg.dg_smoother(dg, A, 'blockJac')
    _, itg, resg, errg = mg.iterative_smoother_solve(A, Sg, u0, bs, maxiter=60, tol=1e-30, alpha=2.0 / 3.0)
    assert itg == ito == 60 and len(errg) == 60
    assert np.allclose(errg, erro, rtol=1e-9, atol=1e-9 * erro[0])
    assert Sg._direct_solver.where == "device"
    # CG fine level (vertices-first numbering: no band): the direct solve falls back to a host factorisation, the
    # error history is still formed on the device
    Hc, bc = o.build_cg_hierarchy(32, ps=(4, 2, 1), nDG=1)
    Hg = mg.MeshHierarchy.from_reference(Hc)
    _, ito, _, erro = o.multigrid(Hc, np.zeros(len(bc)), bc, 60, 1e-9)
    _, itg, _, errg = mg.multigrid(Hg, np.zeros(len(bc)), bc, 60, 1e-9)
    assert itg == ito and np.allclose(errg, erro, rtol=1e-6, atol=1e-9 * erro[0])


def test_pcg_with_ldiv_preconditioner(oracle, mg):
    """extension (no reference loop): device recurrence == the oracle's restatement of it"""
    o = oracle
    Ho, b = o.build_dg_agg_hierarchy(128, p=3, pAgg=1, nAgg=3, first=4)
    H = mg.MeshHierarchy.from_reference(Ho)
    nb = np.linalg.norm(b)
    xo, ito, reso = o.pcg_ldiv(Ho, b, maxiter=60, tol=1e-10)
    xg, itg, resg = mg.pcg(H, b, maxiter=60, tol=1e-10)
    assert abs(itg - ito) <= 1                      # round-off may move the last crossing by one
    k = min(itg, ito) - 1
    assert np.allclose(resg[:k], reso[:k], rtol=1e-4, atol=1e-11 * nb)
    A = Ho.mStiffness[0]
    assert np.linalg.norm(A @ xg - b) <= 2e-10 * nb
    # fewer iterations than the stationary multigrid loop to the same tolerance
    _, itm, _, _ = mg.multigrid(H, np.zeros(len(b)), b, 200, 1e-10, exact=False)
    assert itg < itm
    # warm start from a random guess
    x0 = o.splitmix_normal(len(b), 9)
    xg2, it2, res2 = mg.pcg(H, b, x0=x0, maxiter=80, tol=1e-10)
    assert np.linalg.norm(A @ xg2 - b) <= 2e-10 * nb


def test_v_cycle_on_device_vectors_equals_host_entry(oracle, mg):
    """multigrid_v_cycle(H, x0, b) on DeviceVectors (the drop-in call kept device-resident: aggmg_vcycle_dev) returns a
    DeviceVector with bit for bit the result of the host-array entry (aggmg_vcycle: same kernels, the copies staged
    through pinned chunks by worker threads) -- at a size above the staging threshold, so that path runs -- and leaves
    x0, b untouched"""
    from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy, build_device_hierarchy
    ctx = mg.default_context()
    U = UniformDgAggHierarchy(2 ** 19, p=3, pAgg=1, ratios=(4, 2, 2))      # 16.8 MB vectors
    H = build_device_hierarchy(U, ctx)
    b = U.rhs()
    N = len(b)
    x0 = np.random.default_rng(3).standard_normal(N)
    xh = mg.multigrid_v_cycle(H, x0, b)
    d0, db = ctx.to_device(x0), ctx.to_device(b)
    xd = mg.multigrid_v_cycle(H, d0, db)
    assert isinstance(xd, mg.DeviceVector)
    assert np.array_equal(xd.download(), xh)
    assert np.array_equal(d0.download(), x0) and np.array_equal(db.download(), b)
    xh2 = mg.multigrid_v_cycle(H, xh, b)                                    # the cached device vectors are reused
    assert np.array_equal(mg.multigrid_v_cycle(H, xd, db).download(), xh2)
    with pytest.raises(mg.DimensionMismatch):
        mg.multigrid_v_cycle(H, ctx.alloc(N - 1), db)
    H.free()


@pytest.mark.parametrize("ne", [256, 4096])    # one tile / many tiles (the last ascent's checkpoint needs its own halo)
def test_multigrid_checkpoint_inside_the_fine_level_launch(oracle, mg, ne):
    """multigrid's residual test after every cycle (src/solvers.jl:124-131) formed INSIDE the fine-level launch that
    post-smooths the cycle and pre-smooths the next (AGGMG_OPT_MG_CHECKPOINT, default on) against the form with a residual
    launch of its own: the same iterates bit for bit, the same cycle counts, histories equal to round-off -- with and
    without the error history, check_every 1 / 2 / 3, stopping on the tolerance and on maxiter; and against the oracle"""
    from agglomerationmultigrid1d_amd import _lib
    o = oracle
    Ho, b = o.build_dg_agg_hierarchy(ne, p=3, pAgg=1, nAgg=3, first=4)
    x0 = o.splitmix_normal(len(b), 5)
    out = {}
    for chk in (1, 0):
        ctx = mg.Context(0)
        ctx.set_option(_lib.OPT_MG_CHECKPOINT, chk)
        H = mg.MeshHierarchy.from_reference(Ho, ctx=ctx)
        assert H.level_kinds()[0] == "fused_btd"
        runs = []
        for ce, maxiter, tol, exact in ((1, 40, 1e-9, True), (1, 7, 1e-30, False), (2, 9, 1e-30, True), (3, 40, 1e-6, False), (1, 1, 1e-30, True)):
            runs.append(mg.multigrid(H, x0, b, maxiter, tol, exact=exact, check_every=ce))
        out[chk] = runs
    for (xa, ia, ra, ea), (xb, ib, rb, eb) in zip(out[1], out[0]):
        assert ia == ib and len(ra) == len(rb) and len(ea) == len(eb)
        assert np.array_equal(xa, xb)                                  # the iterates: the same arithmetic
        assert np.allclose(ra, rb, rtol=1e-10, atol=1e-13 * np.linalg.norm(b)) and np.allclose(ea, eb, rtol=1e-9, atol=1e-14)
    xo, ito, reso, erro = o.multigrid(Ho, x0, b, 40, 1e-9)
    xg, itg, resg, errg = out[1][0]
    assert itg == ito and np.allclose(resg, reso, rtol=1e-6, atol=1e-11 * np.linalg.norm(b))
    # (u_exact: the device's cyclic reduction here, a banded LU there -- they differ by cond(A) * eps, which grows as ne^2)
    assert np.allclose(errg, erro, rtol=1e-6, atol=(1e-9 if ne <= 256 else 1e-6) * erro[0])
