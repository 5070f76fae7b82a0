"""Edge cases of the hot path on the GPU against the oracle: degenerate mesh sizes (1, 2, 3
elements; element counts that do not fill a tile; a single-level hierarchy), every DG block size the
fused kernels are instantiated for inside a full V-cycle, Schwarz smoothers inside fused sweeps,
zero sweeps, alpha = 1, multi-column smoother application."""
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


def dense_lu_solve(A, b):
    import scipy.linalg as sla
    return sla.lu_solve(sla.lu_factor(A.toarray()), b)


def vcycle_check(o, mg, Ho, b, x0=None, it_tol=1e-9, **kw):
    H = mg.MeshHierarchy.from_reference(Ho)
    x0 = np.zeros(len(b)) if x0 is None else x0
    x = mg.multigrid_v_cycle(H, x0, b, **kw)
    xr = o.multigrid_v_cycle(Ho, x0, b, coarse_solve=dense_lu_solve, **kw)
    A = Ho.mStiffness[0]
    r0 = max(np.linalg.norm(b - A @ x0), np.linalg.norm(b))
    assert np.linalg.norm(A @ (x - xr)) <= TOL * r0
    assert rel(x, xr) < it_tol
    return H


@pytest.mark.parametrize("n", [1, 2, 3, 5, 63, 249, 257])
def test_tiny_and_ragged_meshes(oracle, mg, n):
    """smoother + residual on meshes of 1..3 elements and sizes straddling the 248/256-element tiles"""
    o = oracle
    mesh, bd = o.model_problem(n)
    dg = o.DgMesh(mesh, 3)
    G, D, C = o.dg_flux_operators(dg, mesh, bd, 1000.0 * n)
    A = o.dg_stiffness(dg, G, D, C)
    So, Sg = o.dg_smoother(dg, A, 'blockJac'), mg.dg_smoother(dg, A, 'blockJac')
    assert Sg.structured
    u, b = o.splitmix_normal(A.shape[0], 1), o.splitmix_normal(A.shape[0], 2)
    for ns in (1, 4, 9):
        ref = u
        for _ in range(ns):
            ref = ref + o.apply_smoother(So, b - o.csc_matvec(A, ref), alpha=2.0 / 3.0)
        assert rel(mg.smooth(Sg.A, Sg, u, b, 2.0 / 3.0, ns), ref) < TOL
    assert rel(mg.residual(Sg.A, u, b), b - o.csc_matvec(A, u)) < TOL


def test_two_level_and_single_level_hierarchies(oracle, mg):
    o = oracle
    Ho, b = o.build_dg_agg_hierarchy(8, p=3, pAgg=1, nAgg=1, first=4)       # 2 levels, 2 coarse elements
    vcycle_check(o, mg, Ho, b)
    Ho, b = o.build_dg_agg_hierarchy(4, p=1, pAgg=0, nAgg=1, first=4)       # coarsest = one scalar
    vcycle_check(o, mg, Ho, b)
    # a hierarchy with one level: the "V-cycle" is the direct solve (src/solvers.jl:39)
    mesh, bd = o.model_problem(40)
    dg = o.DgMesh(mesh, 2)
    G, D, C = o.dg_flux_operators(dg, mesh, bd, 40000.0)
    A = o.dg_stiffness(dg, G, D, C)
    bb = o.splitmix_normal(A.shape[0], 3)
    H = mg.MeshHierarchy([dg], [A], [], [])
    x = mg.multigrid_v_cycle(H, np.zeros(len(bb)), bb)
    assert np.linalg.norm(A @ x - bb) <= 1e-9 * np.linalg.norm(bb)
    assert rel(x, o.sparse_direct_solve(A, bb)) < 1e-8


@pytest.mark.parametrize("ps", [(5, 2), (6, 3), (7, 3), (9, 4, 2)])
def test_vcycle_all_block_sizes(oracle, mg, ps):
    """dg_dg p-coarsening chains through every fused-kernel instantiation (m = 3 .. 10 -> generic)"""
    o = oracle
    Ho, b = o.build_dg_p_hierarchy(40, ps=ps)
    H = vcycle_check(o, mg, Ho, b, it_tol=1e-8)
    # p = 9 (m = 10) has no fused instantiation: the generic CSR path must take over silently
    assert H.structured_levels()[0] == (ps[0] + 1 <= 9)


def test_sweep_parameters(oracle, mg):
    o = oracle
    Ho, b = o.build_dg_agg_hierarchy(96, p=3, pAgg=1, nAgg=2, first=4)
    for kw in (dict(nPre=0, nPost=3), dict(nPre=3, nPost=0), dict(nPre=2, nPost=5, alpha=1.0),
               dict(nPre=8, nPost=8, alpha=0.3)):
        vcycle_check(o, mg, Ho, b, x0=o.splitmix_normal(len(b), 5), it_tol=1e-8, **kw)
    H = mg.MeshHierarchy.from_reference(Ho)
    with pytest.raises(mg.ArgumentError):
        mg.multigrid_v_cycle(H, np.zeros(len(b)), b, nPre=-1)
    with pytest.raises(TypeError):
        mg.multigrid_v_cycle(H, np.zeros(len(b)), b, nPre=2.5)     # nPre::Integer


def test_schwarz_smoothers_in_fused_sweeps(oracle, mg):
    """aggmg_smooth with overlapping CG element blocks (additive / hybrid Schwarz, src/smoother.jl
    :1-46): the tests/cg_smoother_test.jl iteration through the device path"""
    o = oracle
    for n, p in ((16, 4), (333, 4), (200, 1), (150, 7)):
        _schwarz_case(o, mg, n, p)


def _schwarz_case(o, mg, n, p):
    mesh = o.create_uniform_mesh(n, 0.0, 1.0)
    ue = lambda x: -0.5 * x**2 + x
    bd = o.set_boundary(mesh, 0.0, 1.0, [('dir', ue(0.0)), ('dir', ue(1.0))])
    cg = o.CgMesh(mesh, p)
    A, b = o.cg_stiffness_and_rhs(cg, mesh, lambda x: 1.0, bd)
    u0 = o.splitmix_normal(A.shape[0], 2)
    for kind, alpha in (('jac', 0.5), ('addSchwarz', 0.5), ('hybridSchwarz', 1.0)):
        So, Sg = o.cg_smoother(cg, A, kind), mg.cg_smoother(cg, A, kind)
        assert Sg.structured            # the element chain was recognised: fused chain kernel
        ref = u0
        for _ in range(5):
            ref = ref + o.apply_smoother(So, b - o.csc_matvec(A, ref), alpha=alpha)
        assert rel(mg.smooth(Sg.A, Sg, u0, b, alpha, 5), ref) < TOL, kind
        assert rel(mg.apply_smoother(Sg, b, alpha), o.apply_smoother(So, b, alpha=alpha)) < TOL     # generic apply kernel
        if n > 16:
            continue
        xo, ito, reso, _ = o.iterative_smoother_solve(A, So, np.zeros(len(b)), b, maxiter=40, alpha=alpha)
        xg, itg, resg, _ = mg.iterative_smoother_solve(A, Sg, np.zeros(len(b)), b, maxiter=40, alpha=alpha)
        assert itg == ito and rel(xg, xo) < 1e-10 and np.allclose(resg, reso, rtol=1e-8)


def test_multi_column_apply_and_empty(oracle, mg):
    o = oracle
    mesh, bd = o.model_problem(12)
    dg = o.DgMesh(mesh, 2)
    G, D, C = o.dg_flux_operators(dg, mesh, bd, 12000.0)
    A = o.dg_stiffness(dg, G, D, C)
    So, Sg = o.dg_smoother(dg, A, 'blockJac'), mg.dg_smoother(dg, A, 'blockJac')
    B = np.stack([o.splitmix_normal(A.shape[0], s) for s in range(5)], axis=1)
    assert rel(mg.apply_smoother(Sg, B, 0.7), o.apply_smoother(So, B, 0.7)) < TOL
    Y0 = mg.apply_smoother(Sg, np.zeros((A.shape[0], 0)))
    assert Y0.shape == (A.shape[0], 0)


def test_blockdiagonal_container(oracle, mg):
    """tests/blockdiagonal_test.jl through the device path: BlockDiagonal * x / * B and
    lu(BlockDiagonal) \\ x / \\ B against the dense equivalents (the reference prints five norms
    ~1e-16; the sparse-operand variants are set-up code and stay on the oracle side)."""
    o = oracle
    rng = np.random.default_rng(0)
    blocks = [rng.random((3, 3)) + 3 * np.eye(3) for _ in range(50)]
    A = mg.BlockDiagonal(blocks)
    Ao = o.BlockDiagonal([b.copy() for b in blocks])
    A2 = A.todense()
    x = rng.standard_normal(150)
    B = rng.standard_normal((150, 4))
    assert rel(A @ x, A2 @ x) < 1e-14 and rel(A @ B, A2 @ B) < 1e-14
    assert rel(A @ x, Ao.mul_dense(x)) < 1e-14
    ALU = A.lu()
    assert rel(ALU.solve(x), np.linalg.solve(A2, x)) < 1e-13
    assert rel(ALU.solve(B), Ao.lu().solve_dense(B)) < 1e-13
    assert np.array_equal(A.mBlockInds, Ao.mBlockInds) and A.shape == (150, 150)
    with pytest.raises(mg.ArgumentError):
        mg.BlockDiagonal([np.eye(2), np.eye(3)])
    with pytest.raises(mg.DimensionMismatch):
        A @ np.zeros(149)
    with pytest.raises(mg.SingularException):
        mg.BlockDiagonal([np.eye(2), np.zeros((2, 2))]).lu()


def test_nonsymmetric_operators_take_the_general_path(oracle, mg):
    """The fused kernels store the packed symmetric inverse only when the operator is symmetric to
    round-off; an operator with the same pattern but perturbed (non-symmetric) values must run the
    general form -- and both must match the oracle, which knows nothing about symmetry."""
    o = oracle
    import scipy.sparse as sp
    rng = np.random.default_rng(3)
    for p, n in ((3, 528), (1, 304), (7, 64)):
        Ho, b = o.build_dg_agg_hierarchy(n, p=p, pAgg=1, nAgg=2, first=4)
        A = Ho.mStiffness[0].copy()
        # same pattern, symmetry broken at 1e-6 (far above the 1e-13 detection threshold, small
        # enough that the fine operator still matches its coarse levels and the cycle stays sane)
        A.data = A.data * (1.0 + 1e-6 * rng.standard_normal(A.nnz))
        assert abs(A - A.T).max() > 1e-9 * abs(A).max()
        Ho.mStiffness[0] = A
        Ho.mSmoothers[0] = o.dg_smoother(Ho.mMeshes[0], A, 'blockJac')
        H = mg.MeshHierarchy.from_reference(Ho)
        assert all(H.structured_levels())
        x0 = o.splitmix_normal(len(b), 4)
        x = mg.multigrid_v_cycle(H, x0, b)
        xr = o.multigrid_v_cycle(Ho, x0, b)
        # the perturbed fine operator no longer matches its coarse levels, so the cycle amplifies
        # instead of contracting: round-off is measured against ||A x||, not the initial residual
        scale = max(np.linalg.norm(b - A @ x0), np.linalg.norm(A @ xr))
        assert np.linalg.norm(A @ (x - xr)) <= TOL * scale and rel(x, xr) < 1e-9
        S = H.mSmoothers[0]
        u = o.splitmix_normal(len(b), 5)
        ref = u
        for _ in range(4):
            ref = ref + o.apply_smoother(Ho.mSmoothers[0], b - o.csc_matvec(A, ref), alpha=2.0 / 3.0)
        assert rel(mg.smooth(S.A, S, u, b, 2.0 / 3.0, 4), ref) < TOL
        assert rel(mg.residual(S.A, u, b), b - o.csc_matvec(A, u)) < TOL
    # dense off-diagonal blocks (agglomerated level) with broken symmetry
    Ho, b = o.build_dg_agg_hierarchy(512, p=2, pAgg=1, nAgg=2, first=2)
    A1 = Ho.mStiffness[1].copy()
    A1.data = A1.data * (1.0 + 1e-3 * rng.standard_normal(A1.nnz))
    So = o.BlockJacobi(*[x for x in (lambda bl, ii: (bl, ii))(
        [o.LU(A1[np.ix_(Ho.mSmoothers[1].mBlockInds[:, k] - 1, Ho.mSmoothers[1].mBlockInds[:, k] - 1)].toarray())
         for k in range(Ho.mSmoothers[1].mBlockInds.shape[1])], Ho.mSmoothers[1].mBlockInds)])
    Sg = mg.BlockJacobi(A1, So.mBlockInds)
    assert Sg.structured
    u, bb = o.splitmix_normal(A1.shape[0], 6), o.splitmix_normal(A1.shape[0], 7)
    ref = u
    for _ in range(5):
        ref = ref + o.apply_smoother(So, bb - o.csc_matvec(A1, ref), alpha=2.0 / 3.0)
    assert rel(mg.smooth(Sg.A, Sg, u, bb, 2.0 / 3.0, 5), ref) < TOL


@pytest.mark.parametrize("seed", range(12))
def test_randomized_hierarchies(oracle, mg, seed):
    """seeded random shapes: element count, fine p, agglomerated p, ratios, depth, sweep counts,
    damping, initial guess -- one V-cycle and a three-cycle fused loop against the oracle"""
    o = oracle
    rng = np.random.default_rng(1000 + seed)
    p = int(rng.integers(1, 5))
    pAgg = int(rng.integers(0, 2))
    first = int(rng.choice([2, 4]))
    nAgg = int(rng.integers(1, 4))
    tot = first * 2 ** (nAgg - 1)
    n = tot * int(rng.integers(2, 40))
    nPre, nPost = int(rng.integers(0, 5)), int(rng.integers(0, 5))
    alpha = float(rng.uniform(0.3, 1.0))
    Ho, b = o.build_dg_agg_hierarchy(n, p=p, pAgg=pAgg, nAgg=nAgg, first=first)
    x0 = o.splitmix_normal(len(b), seed) * float(rng.choice([0.0, 1.0]))
    H = vcycle_check(o, mg, Ho, b, x0=x0, it_tol=1e-8, nPre=nPre, nPost=nPost, alpha=alpha)
    ctx = H.ctx
    dz = ctx.alloc(len(b))
    H.vcycles_dev(ctx.to_device(x0), ctx.to_device(b), dz, 3, nPre=nPre, nPost=nPost, alpha=alpha)
    xr = x0
    for _ in range(3):
        xr = o.multigrid_v_cycle(Ho, xr, b, nPre=nPre, nPost=nPost, alpha=alpha, coarse_solve=dense_lu_solve)
    A = Ho.mStiffness[0]
    # weak coarse spaces (pAgg = 0) amplify a random iterate instead of damping it: round-off is
    # then relative to ||A x||, not to the initial residual
    scale = max(np.linalg.norm(b - A @ x0), np.linalg.norm(b), np.linalg.norm(A @ xr))
    assert np.linalg.norm(A @ (dz.download() - xr)) <= 3 * TOL * scale, (n, p, pAgg, first, nAgg, nPre, nPost, alpha)


def test_split_ascent_is_bitwise_the_plain_ascent(oracle, mg):
    """aggmg_vcycle_up_split_dev (coarser levels, fine-level tiles at the two ends, the middle) against
    aggmg_vcycle_up_dev, for several end widths incl. empty and everything"""
    from agglomerationmultigrid1d_amd import _lib
    from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy, block_tridiag_to_csc
    U = UniformDgAggHierarchy(4096, p=3, pAgg=1, ratios=(4, 2, 2))
    ctx = mg.default_context()
    n = U.nlevels
    ops = [mg.DeviceOperator(U.stiffness_csc(k), _lib.OP_STIFFNESS, ctx) for k in range(n)]
    sms = [mg.BlockJacobi(ops[k], U.descriptor(k).mBlockInds, ctx) for k in range(n - 1)]
    Ls = [mg.DeviceOperator(U.interpolation_csc(k), _lib.OP_TRANSFER, ctx) for k in range(n - 1)]
    H = mg.MeshHierarchy([U.descriptor(k) for k in range(n)], ops, sms, Ls, ctx=ctx, keep_host=False,
                         coarse_mode=_lib.COARSE_EXTERNAL)
    N = 4096 * 4
    b = ctx.to_device(U.rhs())
    x0 = ctx.to_device(oracle.splitmix_normal(N, 5))
    H.vcycle_down_dev(x0, b)
    rp, sp_, nc = H.coarse_buffers()
    sol = oracle.splitmix_normal(nc, 6)
    ctx.check(ctx.lib.aggmg_memcpy_h2d(ctx.handle, sp_, sol.ctypes.data, nc * 8))
    ref = ctx.alloc(N)
    H.vcycle_up_dev(b, ref)
    ref = ref.download()
    for head, tail in ((80 + 80, 4096 - 160), (0, 4096), (1, 4095), (4096, 4096), (0, 0), (2000, 2100), (3000, 100)):
        H.vcycle_down_dev(x0, b)
        ctx.check(ctx.lib.aggmg_memcpy_h2d(ctx.handle, sp_, sol.ctypes.data, nc * 8))
        out = ctx.to_device(np.full(N, np.nan))
        H.vcycle_up_split_dev(b, out, head, tail, 0)
        H.vcycle_up_split_dev(b, out, head, tail, 2)     # the order of the two fine-level parts is free
        H.vcycle_up_split_dev(b, out, head, tail, 1)
        assert np.array_equal(out.download(), ref), (head, tail)



def test_dense_interpolation_matrix_through_the_device(oracle, mg):
    """mInterpolation::Vector{AbstractMatrix{Float64}} may hold a dense Matrix: dg_cg_interpolation(..., interpFlag = 0)
    returns `mMassMatrixLU \\ Array(N)` (src/interpolation.jl:205).  The product builder gives the same dense array, a
    DeviceOperator takes it, `L' * r` / `u + L * u_c` (src/solvers.jl:36,42) and a two-level V-cycle with it as the
    transfer match the dense NumPy arithmetic."""
    import scipy.sparse as sp
    from agglomerationmultigrid1d_amd import interpolation as ip
    from agglomerationmultigrid1d_amd import _lib
    o = oracle
    n = 24
    Ho, b = o.build_cg_hierarchy(n, ps=(2,), nDG=1, pDG=0)          # CG p=2 -> DG p=0, transfer built with flag 1
    xv = np.linspace(0.0, 1.0, n + 1)
    L = ip.dg_cg_interpolation(ip.DgMesh(xv, 0), ip.CgMesh(xv, 2), 0)
    assert isinstance(L, np.ndarray) and L.shape == Ho.mInterpolation[0].shape
    Lop = mg.DeviceOperator(L, _lib.OP_TRANSFER)
    r, uc, u = rand_like(o, L.shape[0], 1), rand_like(o, L.shape[1], 2), rand_like(o, L.shape[0], 3)
    assert rel(mg.restrict(Lop, r), L.T @ r) < 1e-13
    assert rel(mg.prolong_add(Lop, uc, u), u + L @ uc) < 1e-13
    # the hierarchy with the dense transfer: Galerkin coarse operator L' A L (dense too), V-cycle against NumPy
    A = Ho.mStiffness[0]
    Ac = sp.csc_matrix(L.T @ (A @ L))
    opA = mg.DeviceOperator(A)
    H = mg.MeshHierarchy(None, [opA, Ac], [mg.JacobiSmoother(opA)], [L])
    x = mg.multigrid_v_cycle(H, np.zeros(len(b)), b)
    Ad, d = A.toarray(), A.diagonal()
    v = np.zeros(len(b))
    for _ in range(3):
        v = v + (2.0 / 3.0) * (b - Ad @ v) / d
    v = v + L @ np.linalg.solve(Ac.toarray(), L.T @ (b - Ad @ v))
    for _ in range(3):
        v = v + (2.0 / 3.0) * (b - Ad @ v) / d
    assert np.linalg.norm(Ad @ (x - v)) <= 1e-12 * np.linalg.norm(b)


def rand_like(o, n, seed):
    return np.random.default_rng(100 + seed).standard_normal(n)


def test_page_locked_host_vectors(oracle, mg):
    """aggmg_host_register / aggmg_host_alloc: host-pointer cycles on arrays page-locked once give the bits of the staged
    pageable path; ranges can be unregistered again; an unknown range is refused"""
    o = oracle
    Ho, b = o.build_dg_agg_hierarchy(64, p=3, pAgg=1, nAgg=3, first=4)
    H = mg.MeshHierarchy.from_reference(Ho)
    ctx = H.ctx
    N = len(b)
    x0 = o.splitmix_normal(N, 3)
    ref = mg.multigrid_v_cycle(H, x0, b)                     # pageable arrays, a new result array
    xp = ctx.pinned_empty(N)
    xp[:] = x0
    bp = ctx.pin(np.array(b))
    out = ctx.pinned_empty(N)
    got = mg.multigrid_v_cycle(H, xp, bp, out=out)
    assert got is out and np.array_equal(out, ref) and np.array_equal(xp, x0)
    # mixed: pinned inputs, pageable result
    assert np.array_equal(mg.multigrid_v_cycle(H, xp, bp), ref)
    # ldiv!(y, H, b) straight into a pinned y
    y = ctx.pinned_empty(N)
    mg.ldiv(y, H, bp)
    assert np.array_equal(y, mg.multigrid_v_cycle(H, np.zeros(N), b))
    ctx.unpin(bp)
    with pytest.raises(mg.ArgumentError):
        ctx.unpin(bp)                                        # no longer registered
    assert np.array_equal(mg.multigrid_v_cycle(H, xp, bp, out=out), ref)   # bp is pageable again: staged
    with pytest.raises(mg.ArgumentError):
        mg.multigrid_v_cycle(H, xp, bp, out=xp)              # the result must not alias an input


def test_null_initial_guess_is_a_vector_of_zeros(oracle, mg):
    """aggmg_vcycle / aggmg_vcycle_dev with x0 = NULL -- ldiv!'s zero initial guess (src/solvers.jl:63-92) without a vector of
    zeros being sent or read -- gives the bits of passing zeros: fused block-tridiagonal, chain and generic fine levels, host
    and device forms; ldiv and the preconditioned CG loop go through it"""
    o = oracle
    for Ho, b, kw in ((*o.build_dg_agg_hierarchy(128, p=3, pAgg=1, nAgg=3, first=4), {}),
                      (*o.build_cg_hierarchy(64, ps=(4, 2, 1), nDG=1, pDG=0), {}),
                      (*o.build_dg_p_hierarchy(32, ps=(4, 2, 1)), {})):
        H = mg.MeshHierarchy.from_reference(Ho)
        ctx = H.ctx
        N = len(b)
        ref = mg.multigrid_v_cycle(H, np.zeros(N), b)
        assert np.array_equal(mg.multigrid_v_cycle(H, None, b), ref)
        bd = ctx.to_device(b)
        assert np.array_equal(mg.multigrid_v_cycle(H, None, bd).download(), ref)
        y = np.empty(N)
        mg.ldiv(y, H, b)
        assert np.array_equal(y, ref)
        b2 = b.copy()
        mg.ldiv(H, b2)
        assert np.array_equal(b2, ref)
        xr = o.multigrid_v_cycle(Ho, np.zeros(N), b)
        assert np.linalg.norm(Ho.mStiffness[0] @ (ref - xr)) <= 1e-12 * np.linalg.norm(b)
