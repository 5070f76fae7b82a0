"""Red-black block Gauss-Seidel -- an EXTENSION (the reference has no Gauss-Seidel smoother,
SURVEY.md D1; BASELINE.json's north_star and config 5 name one).  There is no reference behaviour
to match: the HIP path is checked against the oracle's own restatement of the same definition
(oracle BlockGaussSeidelRB / smooth_once), relative 2-norm <= 1e-12 for sweeps, and for V-cycles
on the residual as for the block-Jacobi cycles (tests/test_gpu_parity.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-12


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture(scope="module")
def mg():
    import agglomerationmultigrid1d_amd as mg
    mg.default_context()
    return mg


def gs_hierarchy(o, Ho):
    """same operators, every smoothed level with the Gauss-Seidel restatement on the same blocks"""
    Ho.mSmoothers = [o.BlockGaussSeidelRB(S.mBlocks, S.mBlockInds) for S in Ho.mSmoothers]
    return Ho


@pytest.mark.parametrize("n,p", [(64, 3), (250, 3), (37, 1), (130, 2), (96, 4)])
def test_sweeps_match_the_restatement(oracle, mg, n, p):
    """fine DG level ("compressed" block-tridiagonal form) and the agglomerated levels (dense
    form), 1 .. 9 sweeps (more than one launch's halo budget: chunked), ragged element counts"""
    o = oracle
    nAgg = 2 if n % 8 == 0 else 0
    if nAgg:
        Ho, _ = o.build_dg_agg_hierarchy(n, p=p, pAgg=1, nAgg=nAgg, first=4)
        levels = list(zip(Ho.mStiffness[:-1], Ho.mSmoothers))
    else:
        Ho, _ = o.build_dg_p_hierarchy(n, ps=(p,))
        levels = [(Ho.mStiffness[0], Ho.mSmoothers[0])]
    for k, (A, Sj) in enumerate(levels):
        N = A.shape[0]
        So = o.BlockGaussSeidelRB(Sj.mBlocks, Sj.mBlockInds)
        op = mg.DeviceOperator(A)
        Sg = mg.BlockGaussSeidel(op, Sj.mBlockInds)
        assert Sg.structured
        u0, b = o.splitmix_normal(N, 40 + k), o.splitmix_normal(N, 50 + k)
        for nsw, alpha in ((1, 1.0), (3, 2.0 / 3.0), (9, 0.8)):
            ref = u0.copy()
            for _ in range(nsw):
                ref = So.sweep(A, ref, b, alpha)
            got = mg.smooth(op, Sg, u0, b, alpha=alpha, nsweeps=nsw)
            assert rel(got, ref) < TOL, (n, p, k, nsw)
        # the block-diagonal part alone is what apply_smoother gives, as for BlockJacobi
        assert rel(mg.apply_smoother(Sg, b, alpha=0.5), o.apply_smoother(So, b, alpha=0.5)) < TOL


def test_vcycle_with_gauss_seidel_smoothers(oracle, mg):
    o = oracle
    for n, kw in ((64, {}), (992, {}), (64, dict(nPre=1, nPost=2, alpha=0.9))):
        Ho, b = o.build_dg_agg_hierarchy(n, p=3, pAgg=1, nAgg=3, first=4)
        gs_hierarchy(o, Ho)
        H = mg.MeshHierarchy.from_reference(Ho)
        assert all(H.structured_levels())
        A = Ho.mStiffness[0]
        for x0 in (np.zeros(len(b)), o.splitmix_normal(len(b), 3)):
            xo = o.multigrid_v_cycle(Ho, x0, b, **kw)
            xg = mg.multigrid_v_cycle(H, x0, b, **kw)
            scale = max(np.linalg.norm(b - A @ x0), np.linalg.norm(b), np.linalg.norm(A @ xo))
            assert np.linalg.norm(A @ (xg - xo)) <= TOL * scale, (n, kw)
    # K cycles through the multi-cycle entry point (no cross-cycle fusion for Gauss-Seidel)
    ctx = H.ctx
    x0 = o.splitmix_normal(len(b), 4)
    dx, dy = ctx.to_device(x0), ctx.alloc(len(b))
    H.vcycles_dev(dx, ctx.to_device(b), dy, 3, **kw)
    xo = x0
    for _ in range(3):
        xo = o.multigrid_v_cycle(Ho, xo, b, **kw)
    assert np.linalg.norm(A @ (dy.download() - xo)) <= TOL * max(np.linalg.norm(b), np.linalg.norm(A @ xo))


def test_cycle_is_symmetric_and_contracts_faster_than_block_jacobi(oracle, mg):
    """size-independent properties: pre-smoothing (even, odd) and post-smoothing (odd, even) make
    the cycle from a zero guess a symmetric operator -- what aggmg_pcg_dev needs -- and Gauss-Seidel
    smoothing contracts the residual faster than block-Jacobi on the same hierarchy"""
    o = oracle
    n = 2048
    from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy
    from agglomerationmultigrid1d_amd import _lib
    U = UniformDgAggHierarchy(n, p=3, pAgg=1, ratios=(4, 2, 2))
    ctx = mg.default_context()
    nl = U.nlevels
    ops = [mg.DeviceOperator(U.stiffness_csc(k), _lib.OP_STIFFNESS, ctx) for k in range(nl)]
    Ls = [mg.DeviceOperator(U.interpolation_csc(k), _lib.OP_TRANSFER, ctx) for k in range(nl - 1)]
    desc = [U.descriptor(k) for k in range(nl)]
    # two hierarchies over the same operators: smoother set-up and the coarsest factorisation read
    # the operators' host copies, so only the last hierarchy may release them (keep_host=False)
    Sgs = [mg.BlockGaussSeidel(ops[k], desc[k].mBlockInds, ctx) for k in range(nl - 1)]
    Sj = [mg.BlockJacobi(ops[k], desc[k].mBlockInds, ctx) for k in range(nl - 1)]
    Hgs = mg.MeshHierarchy(desc, ops, Sgs, Ls, ctx=ctx, keep_host=True)
    Hj = mg.MeshHierarchy(desc, ops, Sj, Ls, ctx=ctx, keep_host=False)
    N = 4 * n
    r1, r2, z = o.splitmix_normal(N, 1), o.splitmix_normal(N, 2), np.zeros(N)
    v1 = mg.multigrid_v_cycle(Hgs, z, r1)
    v2 = mg.multigrid_v_cycle(Hgs, z, r2)
    assert abs(v1 @ r2 - r1 @ v2) <= 1e-10 * (np.linalg.norm(v1) * np.linalg.norm(r2))
    b = U.rhs()
    _, _, res_gs, _ = mg.multigrid(Hgs, z, b, 12, 0.0, exact=False)
    _, _, res_j, _ = mg.multigrid(Hj, z, b, 12, 0.0, exact=False)
    assert res_gs[-1] < res_j[-1]
    x, it, res = mg.pcg(Hgs, b, maxiter=60, tol=1e-10)
    assert res[-1] < 1e-10 * np.linalg.norm(b) and it < 60


def test_unsupported_operators_are_refused(oracle, mg):
    """two colours need contiguous blocks and element-to-neighbour coupling only -- or the element chain of a CG
    mesh (tests/test_gpu_chain.py); overlapping blocks that are no such chain are refused"""
    o = oracle
    Hc, _ = o.build_cg_hierarchy(16, ps=(2, 1), nDG=1)
    A, cgm = Hc.mStiffness[0], Hc.mMeshes[0]
    _, inds = o._element_blocks(cgm, A)          # overlapping vertex-sharing element blocks
    assert mg.BlockGaussSeidel(mg.DeviceOperator(A), inds).structured      # in mesh order: the chain form
    shuffled = inds[:, np.random.default_rng(0).permutation(inds.shape[1])]
    with pytest.raises(mg.UnsupportedError):
        mg.BlockGaussSeidel(mg.DeviceOperator(A), shuffled)
