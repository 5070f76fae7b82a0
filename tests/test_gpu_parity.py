"""GPU parity tests proper: the HIP path (through the C ABI, via the Python host mirror) against
the CPU oracle on the same inputs.

Tolerances (north_star / SURVEY.md 8d):
  * index maps of the transfers: bit-exact;
  * smoothed iterates, residuals, transfers: relative 2-norm <= 1e-12;
  * full V-cycle: the residual A*x - b within 1e-12 of the oracle's relative to ||b||; the
    iterate itself within 1e-12 * (1 + kappa-dependent slack) because the coarsest direct solve
    (UMFPACK in the reference, SuperLU in the oracle, banded LU here) is only determined up to
    cond(A_coarse)*eps by ANY solver -- stated per test.
"""
import math

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

TOL = 1e-12


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture(scope="module")
def mg():
    import agglomerationmultigrid1d_amd as mg
    mg.default_context()
    return mg


def rand_vec(o, n, seed):
    return o.splitmix_normal(n, seed)


# ------------------------------------------------------------------------------------------
# index maps
# ------------------------------------------------------------------------------------------
def test_transfer_index_maps_bit_exact(oracle, mg):
    o = oracle
    Ho, _ = o.build_dg_agg_hierarchy(32, p=3, pAgg=1, nAgg=3)
    Hc, _ = o.build_cg_hierarchy(16, ps=(4, 2, 1), nDG=1)
    for L in Ho.mInterpolation + Hc.mInterpolation:
        op = mg.DeviceOperator(L, kind=1)
        Lcsr = sp.csr_matrix(L)
        Lcsr.sort_indices()
        rp, ci, v = op.download(False)
        assert np.array_equal(rp, Lcsr.indptr) and np.array_equal(ci, Lcsr.indices)
        assert np.array_equal(v, Lcsr.data)
        Lcsc = sp.csc_matrix(L)
        Lcsc.sort_indices()
        rp, ci, v = op.download(True)   # CSR of L' == the CSC arrays of L
        assert np.array_equal(rp, Lcsc.indptr) and np.array_equal(ci, Lcsc.indices)
        assert np.array_equal(v, Lcsc.data)
        # the Julia 1-based Int64 triple uploads to the same maps
        cp, rv, nz = o.julia_csc(L)
        op2 = mg.DeviceOperator((L.shape[0], L.shape[1], cp, rv, nz), kind=1)
        rp2, ci2, v2 = op2.download(True)
        assert np.array_equal(rp2, rp) and np.array_equal(ci2, ci) and np.array_equal(v2, v)


# ------------------------------------------------------------------------------------------
# stand-alone ops
# ------------------------------------------------------------------------------------------
def test_residual_restrict_prolong(oracle, mg):
    o = oracle
    for H in (o.build_dg_agg_hierarchy(64, p=3, nAgg=3)[0], o.build_cg_hierarchy(16, ps=(8, 4, 2, 1), nAgg=2)[0],
              o.build_dg_p_hierarchy(8, ps=(8, 4, 2, 1))[0]):
        for k, A in enumerate(H.mStiffness[:-1]):
            N = A.shape[0]
            u, b = rand_vec(o, N, 10 + k), rand_vec(o, N, 20 + k)
            op = mg.DeviceOperator(A)
            r_ref = b - o.csc_matvec(A, u)
            assert rel(mg.residual(op, u, b), r_ref) < TOL
            L = H.mInterpolation[k]
            Lop = mg.DeviceOperator(L, kind=1)
            assert rel(mg.restrict(Lop, r_ref), o.csc_adjoint_matvec(L, r_ref)) < TOL
            uc = rand_vec(o, L.shape[1], 30 + k)
            assert rel(mg.prolong_add(Lop, uc, u), u + o.csc_matvec(L, uc)) < TOL


def test_apply_smoother_seam(oracle, mg):
    """apply_smoother(S, B; alpha) for every smoother type, vector and matrix B."""
    o = oracle
    n = 16
    mesh = o.create_uniform_mesh(n, 0.0, 1.0)
    bd = o.set_boundary(mesh, 0.0, 1.0, [('dir', 0.0), ('dir', 0.5)])
    cg = o.CgMesh(mesh, 4)
    A, _ = o.cg_stiffness_and_rhs(cg, mesh, lambda x: 1.0, bd)
    B = np.stack([rand_vec(o, A.shape[0], s) for s in (1, 2, 3)], axis=1)
    for kind in ('jac', 'addSchwarz', 'hybridSchwarz'):
        So = o.cg_smoother(cg, A, kind)
        Sg = mg.cg_smoother(cg, A, kind)
        assert Sg.structured     # the CG mesh's element lists give all three the chain form (fused sweeps)
        for alpha in (1.0, 0.5):
            assert rel(mg.apply_smoother(Sg, B[:, 0], alpha), o.apply_smoother(So, B[:, 0], alpha)) < TOL
            Y = mg.apply_smoother(Sg, B, alpha)
            assert Y.shape == B.shape and rel(Y, o.apply_smoother(So, B, alpha)) < TOL
    dg = o.DgMesh(mesh, 2)
    G, D, C = o.dg_flux_operators(dg, mesh, bd, 1000.0 * n)
    A = o.dg_stiffness(dg, G, D, C)
    B = rand_vec(o, A.shape[0], 5)
    for kind in ('jac', 'blockJac'):
        So, Sg = o.dg_smoother(dg, A, kind), mg.dg_smoother(dg, A, kind)
        assert rel(mg.apply_smoother(Sg, B, 2.0 / 3.0), o.apply_smoother(So, B, 2.0 / 3.0)) < TOL
    with pytest.raises(mg.DimensionMismatch):
        mg.apply_smoother(Sg, np.zeros(A.shape[0] + 1))
    with pytest.raises(mg.ArgumentError):
        mg.dg_smoother(dg, A, 'gs')      # no Gauss-Seidel in the reference


def oracle_sweeps(o, A, S, u, b, alpha, ns):
    for _ in range(ns):
        u = u + o.apply_smoother(S, b - o.csc_matvec(A, u), alpha=alpha)
    return u


@pytest.mark.parametrize("p", [1, 2, 3, 4, 5, 8])
def test_structured_block_jacobi_sweeps(oracle, mg, p):
    """K1: fused block-Jacobi sweeps on a nodal DG level (LDS-tiled kernel, compressed
    off-diagonal blocks), many tile boundaries, every sweep count incl. chunking (>8)."""
    o = oracle
    n = 700 if p <= 4 else 150
    mesh, bd = o.model_problem(n)
    dg = o.DgMesh(mesh, p)
    G, D, C = o.dg_flux_operators(dg, mesh, bd, 1000.0 * n)
    A = o.dg_stiffness(dg, G, D, C)
    So = o.dg_smoother(dg, A, 'blockJac')
    Sg = mg.dg_smoother(dg, A, 'blockJac')
    assert Sg.structured
    N = A.shape[0]
    u0, b = rand_vec(o, N, 0), rand_vec(o, N, 1)
    for ns in (0, 1, 3, 8, 19):
        ref = oracle_sweeps(o, A, So, u0, b, 2.0 / 3.0, ns)
        got = mg.smooth(Sg.A, Sg, u0, b, 2.0 / 3.0, ns)
        assert rel(got, ref) < TOL, (p, ns)


@pytest.mark.parametrize("pAgg,first", [(0, 2), (1, 2), (1, 4)])
def test_structured_sweeps_agglomerated_levels(oracle, mg, pAgg, first):
    """dense off-diagonal variant (m = 1, 2) on Galerkin agglomerated levels"""
    o = oracle
    H, _ = o.build_dg_agg_hierarchy(1024, p=2, pAgg=pAgg, nAgg=2, first=first)
    for k in (1, 2):
        A, So = H.mStiffness[k], H.mSmoothers[k]
        Sg = mg.BlockJacobi(A, So.mBlockInds)
        assert Sg.structured
        N = A.shape[0]
        u0, b = rand_vec(o, N, 3), rand_vec(o, N, 4)
        for ns in (1, 3, 11):
            assert rel(mg.smooth(Sg.A, Sg, u0, b, 2.0 / 3.0, ns),
                       oracle_sweeps(o, A, So, u0, b, 2.0 / 3.0, ns)) < TOL


def test_generic_point_jacobi_sweeps(oracle, mg):
    """K2: fused point-Jacobi sweep on CG levels (generic CSR row-per-lane-group kernel)."""
    o = oracle
    H, b = o.build_cg_hierarchy(64, ps=(4, 2, 1))
    for k in range(3):
        A, So = H.mStiffness[k], H.mSmoothers[k]
        Sg = mg.JacobiSmoother(A, detect=False)        # the generic CSR kernels (no lists, pattern detection off)
        assert not Sg.structured
        N = A.shape[0]
        u0, bb = rand_vec(o, N, 6), rand_vec(o, N, 7)
        for ns in (1, 2, 3, 6):
            assert rel(mg.smooth(Sg.A, Sg, u0, bb, 2.0 / 3.0, ns),
                       oracle_sweeps(o, A, So, u0, bb, 2.0 / 3.0, ns)) < TOL


@pytest.mark.parametrize("n,p", [(7, 3), (64, 3), (700, 3), (2000, 1), (1500, 4), (333, 7)])
def test_banded_generic_point_jacobi_many_sweeps_per_launch(oracle, mg, n, p):
    """dg_smoother(mesh, A, :jac) (src/smoother.jl:146-151) on DG operators through the GENERIC kernels: the operator is
    banded, so csr_band_kernel keeps the x window of a row block in LDS and runs several sweeps per launch (the operator's own limit, 1 + 32 / bw, at most 8)
    (temporal blocking with halo rows) -- 1 .. 9 sweeps (one launch, several launches, uneven splits; several row
    blocks with halos at n = 700 .. 2000; tiny systems whose halo is the whole matrix), in place and out of place,
    against the oracle's sweeps; the residual and y = A x take the window kernel too"""
    o = oracle
    mesh = o.create_uniform_mesh(n, 0.0, 1.0)
    bd = o.set_boundary(mesh, 0.0, 1.0, [('neu', 0.0), ('dir', 1.0)])
    dg = o.DgMesh(mesh, p)
    G, D, C = o.dg_flux_operators(dg, mesh, bd, 1000.0 * n)
    A = o.dg_stiffness(dg, G, D, C)
    So = o.dg_smoother(dg, A, 'jac')
    Sg = mg.JacobiSmoother(A, detect=False)        # no element lists: generic CSR kernels
    assert not Sg.structured
    N = A.shape[0]
    u0, bb = rand_vec(o, N, 16), rand_vec(o, N, 17)
    for ns in (1, 2, 3, 4, 5, 6, 7, 9):
        ref = oracle_sweeps(o, A, So, u0, bb, 2.0 / 3.0, ns)
        assert rel(mg.smooth(Sg.A, Sg, u0, bb, 2.0 / 3.0, ns), ref) < TOL, ns
    ctx = Sg.A.ctx
    bd_ = ctx.to_device(bb)
    for ns in (1, 3, 8):                           # in place on the device
        x = ctx.to_device(u0)
        ctx.check(ctx.lib.aggmg_smooth_dev(ctx.handle, Sg.A.handle, Sg.handle, x.ptr, bd_.ptr, 2.0 / 3.0, ns, x.ptr))
        assert rel(x.download(), oracle_sweeps(o, A, So, u0, bb, 2.0 / 3.0, ns)) < TOL, ns
    r = mg.residual(Sg.A, u0, bb)
    assert rel(r, bb - o.csc_matvec(A, u0)) < TOL


@pytest.mark.parametrize("n,ps", [(48, (3, 2, 1)), (900, (3, 2, 1)), (1100, (2, 1))])
def test_vcycle_on_banded_generic_levels_with_point_jacobi(oracle, mg, n, ps):
    """A p-coarsened DG hierarchy smoothed by dg_smoother(mesh, A, :jac) (src/smoother.jl:146-151; tests/dg_smoother_test.jl's
    smoother inside MeshHierarchy): its levels have no block smoother to give the block-tridiagonal form away, so they run
    the GENERIC kernels -- and being banded, the descent's sweeps and the residual for the restriction come out of ONE
    csr_band_kernel launch (tile entries in registers, nPre sweeps + a residual pass), the ascent's sweeps out of another.
    V(nPre, nPost) for several sweep counts -- more than one launch's worth too -- against the oracle's cycle; one tile and
    several tiles per level."""
    o = oracle
    Ho, b = o.build_dg_p_hierarchy(n, ps=ps)
    for k in range(len(ps) - 1):
        Ho.mSmoothers[k] = o.dg_smoother(Ho.mMeshes[k], Ho.mStiffness[k], 'jac')
    H = mg.MeshHierarchy.from_reference(Ho)
    assert H.level_kinds()[:-1] == ['generic'] * (len(ps) - 1)
    x0 = rand_vec(o, len(b), 3)
    for nPre, nPost in ((3, 3), (1, 2), (2, 0), (0, 1), (9, 10)):
        xg = mg.multigrid_v_cycle(H, x0, b, nPre=nPre, nPost=nPost, alpha=0.5)
        xo = o.multigrid_v_cycle(Ho, x0, b, nPre=nPre, nPost=nPost, alpha=0.5)
        # (the coarsest solves differ by cond(A_c) eps -- cyclic reduction here, a sparse LU there -- which the iterate shows
        # at 1e-11 for n = 900; an indexing or halo error would show at the scale of the iterate)
        assert rel(xg, xo) < (TOL if n < 100 else 1e-9), (nPre, nPost)
        assert np.linalg.norm(Ho.mStiffness[0] @ (xg - xo)) < 1e-11 * np.linalg.norm(b), (nPre, nPost)


@pytest.mark.parametrize("N,bw,fill", [(5, 1, 1.0), (3000, 1, 1.0), (2500, 6, 0.6), (4000, 13, 0.3), (3000, 30, 0.15),
                                       (3000, 32, 0.1), (1200, 32, 1.0), (2000, 40, 0.2)])
def test_band_kernel_on_synthetic_bands(mg, N, bw, fill):
    """csr_band_kernel beyond the DG operators: random diagonally dominant matrices with entries within `bw` of the diagonal
    -- tridiagonal, mid bands, the widest band the window kernels take (32: two sweeps per launch), a FULL band of 65
    entries per row (a tile could not hold its own halo: the planner declines and the stream kernel runs), a band too wide
    (40: stream kernel) -- 1 .. 11 point-Jacobi sweeps against NumPy; several launches, uneven splits, ragged rows."""
    import scipy.sparse as sp
    rng = np.random.default_rng(N * 131 + bw)
    rows, cols, vals = [], [], []
    for d in range(-bw, bw + 1):
        if d == 0:
            continue
        i = np.arange(max(0, -d), min(N, N - d))
        keep = rng.random(i.size) < fill
        if d in (-bw, bw):
            keep[:] = True                      # the band's edge is really there
        i = i[keep]
        rows.append(i); cols.append(i + d); vals.append(rng.standard_normal(i.size))
    rows, cols, vals = map(np.concatenate, (rows, cols, vals))
    A = sp.csr_matrix((vals, (rows, cols)), shape=(N, N))
    A = A + sp.diags(np.asarray(abs(A).sum(axis=1)).ravel() + 1.0)
    A = sp.csc_matrix(A)
    Sg = mg.JacobiSmoother(A, detect=False)
    assert not Sg.structured
    dg = A.diagonal()
    u0, b = rng.standard_normal(N), rng.standard_normal(N)
    for ns in (1, 2, 3, 5, 8, 11):
        x = u0.copy()
        for _ in range(ns):
            x = x + 0.7 * ((b - A @ x) / dg)
        got = mg.smooth(Sg.A, Sg, u0, b, 0.7, ns)
        assert np.linalg.norm(got - x) <= 1e-13 * np.linalg.norm(x), (ns, np.linalg.norm(got - x) / np.linalg.norm(x))


def test_iterative_smoother_solve_matches(oracle, mg):
    """tests/dg_smoother_test.jl call pattern through the product API: same iteration count,
    same iterate."""
    o = oracle
    n = 16
    mesh = o.create_uniform_mesh(n, 0.0, 1.0)
    ue = lambda x: -0.5 * x**2 + x
    bd = o.set_boundary(mesh, 0.0, 1.0, [('dir', ue(0.0)), ('dir', ue(1.0))])
    dg = o.DgMesh(mesh, 2)
    G, D, C = o.dg_flux_operators(dg, mesh, bd, 1000.0 * n)
    A = o.dg_stiffness(dg, G, D, C)
    f, r = o.dg_flux_rhs(dg, mesh, lambda x: 1.0, bd, 1000.0 * n)
    b = o.dg_rhs(dg, D, f, r)
    u0 = np.zeros(A.shape[1])
    xo, ito, reso, erro = o.iterative_smoother_solve(A, o.dg_smoother(dg, A, 'blockJac'), u0, b,
                                                     maxiter=10**4, alpha=2.0 / 3.0)
    xg, itg, resg, errg = mg.iterative_smoother_solve(A, mg.dg_smoother(dg, A, 'blockJac'), u0, b,
                                                      maxiter=10**4, alpha=2.0 / 3.0)
    assert abs(itg - ito) <= 1
    k = min(itg, ito) - 1
    assert abs(resg[k] - reso[k]) <= 1e-9 * reso[0]
    assert rel(xg, xo) < 1e-9


# ------------------------------------------------------------------------------------------
# V-cycle
# ------------------------------------------------------------------------------------------
def dense_lu_solve(A, b):
    """LAPACK getrf/getrs on the densified coarsest operator: the same partial-pivot arithmetic
    as the product's banded LU, used where a test wants the coarsest solve out of the
    comparison."""
    import scipy.linalg as sla
    return sla.lu_solve(sla.lu_factor(A.toarray()), b)


def check_vcycle(o, mg, Ho, b, x0=None, nPre=3, nPost=3, alpha=2.0 / 3.0, it_tol=1e-9,
                 coarse_solve=None):
    H = mg.MeshHierarchy.from_reference(Ho)
    x0 = np.zeros(len(b)) if x0 is None else x0
    x0c, bc = x0.copy(), b.copy()
    x = mg.multigrid_v_cycle(H, x0, b, nPre=nPre, nPost=nPost, alpha=alpha)
    assert np.array_equal(x0, x0c) and np.array_equal(b, bc)   # inputs untouched
    xr = o.multigrid_v_cycle(Ho, x0, b, nPre=nPre, nPost=nPost, alpha=alpha,
                             coarse_solve=coarse_solve)
    A = Ho.mStiffness[0]
    # residual parity (north_star: "residual within 1e-12 of reference"), relative to the
    # residual the cycle started from (= ||b|| for the zero initial guess)
    r0 = max(np.linalg.norm(b - A @ x0), np.linalg.norm(b))
    assert np.linalg.norm(A @ (x - xr)) <= TOL * r0
    assert rel(x, xr) < it_tol
    return H, x, xr


@pytest.mark.parametrize("n", [16, 64, 1000, 4096])
def test_vcycle_dg_agg_config3_shape(oracle, mg, n):
    """BASELINE config 3 shape at oracle-feasible n: DG p=3 -> Agg(4:1) -> Agg(2:1) -> Agg(2:1),
    V(3,3), alpha=2/3; every level runs the fused kernels."""
    o = oracle
    nn = n if n % 16 == 0 else 992
    Ho, b = o.build_dg_agg_hierarchy(nn, p=3, pAgg=1, nAgg=3, first=4)
    H, x, xr = check_vcycle(o, mg, Ho, b)
    assert all(H.structured_levels())
    # second cycle from a non-zero iterate and a random right-hand side
    check_vcycle(o, mg, Ho, rand_vec(o, len(b), 1), x0=rand_vec(o, len(b), 0))


def test_vcycle_same_coarse_solver_is_1e12(oracle, mg):
    """With the coarsest solve taken out of the comparison (hierarchy deep enough that the
    coarsest level is a single agglomerate, solved by the same partial-pivot LU arithmetic on
    both sides) the iterate itself agrees to 1e-12."""
    o = oracle
    Ho, b = o.build_dg_agg_hierarchy(64, p=3, pAgg=1, nAgg=5, first=4)
    assert Ho.mStiffness[-1].shape[0] == 2
    check_vcycle(o, mg, Ho, b, it_tol=TOL, coarse_solve=dense_lu_solve)
    check_vcycle(o, mg, Ho, b, nPre=1, nPost=2, alpha=0.5, it_tol=TOL, coarse_solve=dense_lu_solve)
    # no smoothing at all: x = P A_c^{-1} R b, the round-off of the restrictions goes straight
    # through cond(A_c) ~ 1e4 of the 2x2 coarsest operator
    check_vcycle(o, mg, Ho, b, nPre=0, nPost=0, it_tol=1e-11, coarse_solve=dense_lu_solve)
    check_vcycle(o, mg, Ho, b, nPre=6, nPost=7, it_tol=TOL, coarse_solve=dense_lu_solve)   # falls back to unfused chunks


def test_coarse_solver_modes(oracle, mg):
    """Coarsest `A \\ rhs` (src/solvers.jl:39): device block cyclic reduction vs host banded LU vs
    the oracle's sparse LU -- all within the residual tolerance; CR refused where not applicable."""
    o = oracle
    from agglomerationmultigrid1d_amd import _lib
    Ho, b = o.build_dg_agg_hierarchy(4096, p=3, pAgg=1, nAgg=2, first=4)   # coarsest: 512 blocks of 2
    xr = o.multigrid_v_cycle(Ho, np.zeros(len(b)), b)
    A = Ho.mStiffness[0]
    xs = {}
    for mode in (_lib.COARSE_HOST_BANDED, _lib.COARSE_DEVICE_CR, _lib.COARSE_AUTO):
        H = mg.MeshHierarchy.from_reference(Ho, coarse_mode=mode)
        info = H.coarse_info()
        assert info['on_device'] == (mode != _lib.COARSE_HOST_BANDED)
        if info['on_device']:
            assert info['block_size'] == 2 and info['cond_est'] < 1e8
        x = mg.multigrid_v_cycle(H, np.zeros(len(b)), b)
        assert np.linalg.norm(A @ (x - xr)) <= TOL * np.linalg.norm(b)
        assert rel(x, xr) < 1e-9
        xs[mode] = x
    assert rel(xs[_lib.COARSE_DEVICE_CR], xs[_lib.COARSE_HOST_BANDED]) < 1e-9
    # odd block counts / tail-only / scalar tridiagonal / chunked (coarsest > 2048 rows) operators
    for n, nAgg in ((48, 2), (16, 1), (6000, 1), (10000, 1), (16386, 1)):
        Ho, b = o.build_dg_agg_hierarchy(n, p=2, pAgg=0, nAgg=nAgg, first=2)
        H = mg.MeshHierarchy.from_reference(Ho, coarse_mode=_lib.COARSE_DEVICE_CR)
        x = mg.multigrid_v_cycle(H, np.zeros(len(b)), b)
        xr = o.multigrid_v_cycle(Ho, np.zeros(len(b)), b)
        assert np.linalg.norm(Ho.mStiffness[0] @ (x - xr)) <= TOL * np.linalg.norm(b)
    # CG p=2 coarsest level in vertices-first numbering is not banded: CR must refuse, AUTO falls back
    Hc, bc = o.build_cg_hierarchy(64, ps=(4, 2))
    with pytest.raises(mg.UnsupportedError):
        mg.MeshHierarchy.from_reference(Hc, coarse_mode=_lib.COARSE_DEVICE_CR)
    H = mg.MeshHierarchy.from_reference(Hc)
    assert not H.coarse_info()['on_device']
    x = mg.multigrid_v_cycle(H, np.zeros(len(bc)), bc)
    xr = o.multigrid_v_cycle(Hc, np.zeros(len(bc)), bc)
    assert np.linalg.norm(Hc.mStiffness[0] @ (x - xr)) <= TOL * np.linalg.norm(bc)


def test_structured_residual(oracle, mg):
    """aggmg_residual on an operator whose block-Jacobi smoother recognised the block-tridiagonal
    structure runs the index-free fused kernel (S = 0 sweeps)."""
    o = oracle
    for p, n in ((3, 1000), (1, 300), (8, 70)):
        mesh, bd = o.model_problem(n)
        dg = o.DgMesh(mesh, p)
        G, D, C = o.dg_flux_operators(dg, mesh, bd, 1000.0 * n)
        A = o.dg_stiffness(dg, G, D, C)
        S = mg.dg_smoother(dg, A, 'blockJac')
        assert S.structured
        u, b = rand_vec(o, A.shape[0], 8), rand_vec(o, A.shape[0], 9)
        assert rel(mg.residual(S.A, u, b), b - o.csc_matvec(A, u)) < TOL


def test_vcycle_dg_p_hierarchy(oracle, mg):
    """tests/dg_heirarchy_test.jl shape: DG p = 8,4,2,1 (dg_dg transfers, rho = 1)."""
    o = oracle
    Ho, b = o.build_dg_p_hierarchy(128, ps=(8, 4, 2, 1))
    H, x, xr = check_vcycle(o, mg, Ho, b)
    assert all(H.structured_levels())


def test_vcycle_config1_cg_plus_dg0(oracle, mg):
    """BASELINE config 1: CG n=1024 p=1, point-Jacobi, + DG p=0 coarse level.  With the mesh's element
    lists the CG level runs the fused chain kernel; without them (operators only) the library recognises the chain
    from the operator (AGGMG_OPT_DETECT_CHAIN) -- with that switched off, the generic CSR path."""
    o = oracle
    Ho, b = o.build_cg_hierarchy(1024, ps=(1,), nDG=1, pDG=0)
    H, x, xr = check_vcycle(o, mg, Ho, b, it_tol=1e-8)
    assert H.level_kinds() == ['fused_chain', 'coarsest']
    from agglomerationmultigrid1d_amd import _lib
    ctx = mg.default_context()
    for detect, kinds in ((1, ['fused_chain', 'coarsest']), (0, ['generic', 'coarsest'])):
        ctx.set_option(_lib.OPT_DETECT_CHAIN, detect)
        try:
            Hg = mg.MeshHierarchy(None, Ho.mStiffness, Ho.mSmoothers, Ho.mInterpolation)
        finally:
            ctx.set_option(_lib.OPT_DETECT_CHAIN, 1)
        assert Hg.level_kinds() == kinds
        xg = mg.multigrid_v_cycle(Hg, np.zeros(len(b)), b)
        assert np.linalg.norm(Ho.mStiffness[0] @ (xg - xr)) <= TOL * np.linalg.norm(b)


def test_vcycle_mixed_cg_dg_agg(oracle, mg):
    """tests/dg_cg_heirarchy_test.jl and full_heirarchy_test.jl shapes (CG chain, then DG / Agg)."""
    o = oracle
    Ho, b = o.build_cg_hierarchy(128, ps=(8, 4, 2, 1), nDG=1)
    check_vcycle(o, mg, Ho, b, it_tol=1e-8)
    Ho, b = o.build_cg_hierarchy(64, ps=(8, 4, 2, 1), nAgg=5)
    check_vcycle(o, mg, Ho, b, it_tol=1e-9)
    Ho, b = o.build_cg_hierarchy(64, ps=(4, 2, 1), nDG=1, pDG=0)   # config 5 shape
    check_vcycle(o, mg, Ho, b, it_tol=1e-8)


def test_multigrid_and_ldiv(oracle, mg):
    """multigrid(H,x0,b,maxiter,tol) -> same iteration count / histories; ldiv! semantics."""
    o = oracle
    Ho, b = o.build_dg_agg_hierarchy(64, p=3, pAgg=1, nAgg=3, first=4)
    H = mg.MeshHierarchy.from_reference(Ho)
    x0 = np.zeros(len(b))
    xo, ito, reso, erro = o.multigrid(Ho, x0, b, 200, 1e-10)
    xg, itg, resg, errg = mg.multigrid(H, x0, b, 200, 1e-10)
    assert itg == ito
    assert np.allclose(resg, reso, rtol=1e-6, atol=1e-12 * np.linalg.norm(b))
    assert np.allclose(errg, erro, rtol=1e-6, atol=1e-10)
    y = np.empty(len(b))
    mg.ldiv(y, H, b)
    bb = b.copy()
    mg.ldiv(H, bb)                       # overwrites b
    assert np.array_equal(y, bb)
    yo = o.multigrid_v_cycle(Ho, x0, b)
    assert np.linalg.norm(Ho.mStiffness[0] @ (y - yo)) <= TOL * np.linalg.norm(b)


def test_vcycles_fused_across_cycles(oracle, mg):
    """aggmg_vcycles_dev: K cycles back to back with the fine level's post- and pre-smoothing in
    one launch == K separate V-cycles (bitwise: same per-row arithmetic) == the oracle's K cycles."""
    o = oracle
    Ho, b = o.build_dg_agg_hierarchy(992, p=3, pAgg=1, nAgg=3, first=4)
    H = mg.MeshHierarchy.from_reference(Ho)
    ctx = H.ctx
    x0 = rand_vec(o, len(b), 0)
    db = ctx.to_device(b)
    for K in (1, 2, 5):
        dx, dy, dz = ctx.to_device(x0), ctx.alloc(len(b)), ctx.alloc(len(b))
        H.vcycles_dev(dx, db, dz, K)
        fused = dz.download()
        for _ in range(K):
            H.vcycle_dev(dx, db, dy)
            dx, dy = dy, dx
        sep = dx.download()
        assert np.array_equal(fused, sep), K
        xr = x0
        for _ in range(K):
            xr = o.multigrid_v_cycle(Ho, xr, b)
        A = Ho.mStiffness[0]
        assert np.linalg.norm(A @ (fused - xr)) <= TOL * np.linalg.norm(b - A @ x0)
    # other sweep counts, incl. one too deep to fuse (falls back to separate launches)
    for nPre, nPost in ((1, 2), (4, 4)):
        dx, dz = ctx.to_device(x0), ctx.alloc(len(b))
        H.vcycles_dev(dx, db, dz, 3, nPre=nPre, nPost=nPost)
        xr = x0
        for _ in range(3):
            xr = o.multigrid_v_cycle(Ho, xr, b, nPre=nPre, nPost=nPost)
        assert np.linalg.norm(Ho.mStiffness[0] @ (dz.download() - xr)) <= TOL * np.linalg.norm(b - Ho.mStiffness[0] @ x0)
    # generic (CG) hierarchy: plain sequence path
    Hc, bc = o.build_cg_hierarchy(64, ps=(4, 2, 1), nDG=1, pDG=0)
    Hg = mg.MeshHierarchy.from_reference(Hc)
    dx, dz = ctx.to_device(np.zeros(len(bc))), ctx.alloc(len(bc))
    Hg.vcycles_dev(dx, ctx.to_device(bc), dz, 3)
    xr = np.zeros(len(bc))
    for _ in range(3):
        xr = o.multigrid_v_cycle(Hc, xr, bc)
    assert np.linalg.norm(Hc.mStiffness[0] @ (dz.download() - xr)) <= TOL * np.linalg.norm(bc)
    # multigrid with a residual check every 3 cycles: same iterate after the same number of cycles
    xo, ito, reso, erro = o.multigrid(Ho, np.zeros(len(b)), b, 200, 1e-10)
    xg, itg, resg, errg = mg.multigrid(H, np.zeros(len(b)), b, 200, 1e-10, check_every=3)
    assert ito <= itg <= ito + 2 and len(resg) == (itg + 2) // 3
    assert resg[-1] < 1e-10 * np.linalg.norm(b)


def test_error_behaviour(oracle, mg):
    o = oracle
    A = sp.csc_matrix(np.array([[0.0, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 0.0], [0, 0, 1.0, 0], [0, 0, 0, 1.0]]))
    with pytest.raises(mg.SingularException):
        mg.BlockJacobi(A, np.array([[1, 3], [2, 4]]))
    with pytest.raises(mg.DimensionMismatch):
        mg.BlockJacobi(A, np.array([[1, 3], [2, 9]]))
    Ho, b = o.build_dg_agg_hierarchy(16, p=1, pAgg=1, nAgg=1)
    H = mg.MeshHierarchy.from_reference(Ho)
    with pytest.raises(mg.DimensionMismatch):
        mg.multigrid_v_cycle(H, np.zeros(3), b)
    with pytest.raises(mg.ArgumentError):
        mg.MeshHierarchy(None, Ho.mStiffness, Ho.mSmoothers, [])
    with pytest.raises(mg.DimensionMismatch):
        mg.MeshHierarchy(None, Ho.mStiffness, Ho.mSmoothers, [Ho.mInterpolation[0].T])
