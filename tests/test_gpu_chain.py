"""GPU parity of the CG chain path (cgt_kernels.hpp): continuous-Galerkin levels set up with their
element node lists run the temporally blocked fused point-Jacobi kernel in element-contiguous order
while every vector at the C ABI stays in the reference's vertices-first numbering
(src/cg_mesh.jl:37-45,59-65).  Checked against the CPU oracle (JacobiSmoother src/smoother.jl:52-58,
multigrid_v_cycle src/solvers.jl:19-50) and, for operators the reference would never assemble, against
plain SciPy arithmetic.

Tolerances: sweeps / residuals / transfers relative 2-norm <= 1e-12; V-cycle residual within 1e-12
of the oracle's relative to the starting residual, iterate within the coarsest-solver slack stated
per test (tests/test_gpu_parity.py explains why)."""
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


def jacobi_sweeps(A, u, b, alpha, ns):
    d = A.diagonal()
    for _ in range(ns):
        u = u + alpha * ((b - A @ u) / d)
    return u


def cg_level(o, n, p, bc=None):
    mesh, bd = o.model_problem(n, bc=bc)
    cg = o.CgMesh(mesh, p)
    A, b = o.cg_stiffness_and_rhs(cg, mesh, np.cos, bd)
    return cg, A, b


@pytest.mark.parametrize("p", [1, 2, 3, 4, 5, 6, 7, 8])
def test_chain_sweeps_and_residual(oracle, mg, p):
    """fused point-Jacobi sweeps on an assembled CG level: every block size, many tile boundaries,
    sweep counts incl. chunking (> 8 per launch), element counts around the tile size and tiny"""
    o = oracle
    for n in (1, 2, 3, 37, 900 if p <= 4 else 260):
        cg, A, _ = cg_level(o, n, p)
        So = o.cg_smoother(cg, A, 'jac')
        Sg = mg.cg_smoother(cg, A, 'jac')
        assert Sg.structured, (p, n)
        N = A.shape[0]
        u0, b = o.splitmix_normal(N, 10 + p), o.splitmix_normal(N, 20 + p)
        for ns in (0, 1, 3, 8, 9, 19):
            ref = u0.copy()
            for _ in range(ns):
                ref = ref + o.apply_smoother(So, b - o.csc_matvec(A, ref), alpha=2.0 / 3.0)
            got = mg.smooth(Sg.A, Sg, u0, b, 2.0 / 3.0, ns)
            assert rel(got, ref) < TOL, (p, n, ns)
        assert rel(mg.residual(Sg.A, u0, b), b - o.csc_matvec(A, u0)) < TOL
        # zero initial guess through the device entry point (u_in = NULL is the V-cycle's first sweep)
        z = mg.smooth(Sg.A, Sg, np.zeros(N), b, 0.5, 2)
        assert rel(z, jacobi_sweeps(sp.csr_matrix(A), np.zeros(N), b, 0.5, 2)) < TOL


def test_chain_boundary_conditions(oracle, mg):
    """Dirichlet / Neumann at either end change the first and last block rows"""
    o = oracle
    for bc in ([('dir', 0.3), ('dir', 0.5)], [('dir', 1.0), ('neu', 0.2)], [('neu', -0.1), ('dir', 0.0)]):
        cg, A, b = cg_level(o, 50, 3, bc=bc)
        Sg = mg.cg_smoother(cg, A, 'jac')
        assert Sg.structured
        u0 = o.splitmix_normal(A.shape[0], 3)
        assert rel(mg.smooth(Sg.A, Sg, u0, b, 2.0 / 3.0, 5), jacobi_sweeps(sp.csr_matrix(A), u0, b, 2.0 / 3.0, 5)) < TOL


def test_chain_arbitrary_values_and_numbering(oracle, mg):
    """Nothing is assumed beyond the chain pattern: random unsymmetric values on the CG pattern and a
    scrambled node numbering (elements listed through the scrambled numbers) give the SciPy result."""
    o = oracle
    rng = np.random.default_rng(7)
    for p, n in ((4, 300), (2, 11), (1, 600), (5, 90)):
        cg, A, _ = cg_level(o, n, p, bc=[('neu', 0.0), ('neu', 0.0)])
        A = sp.csr_matrix(A)
        A.data = rng.standard_normal(A.nnz)
        A = A + sp.diags(10.0 + rng.random(A.shape[0]))          # comfortably invertible diagonal
        N = A.shape[0]
        perm = rng.permutation(N)                                 # new number of old node i: perm[i]
        P = sp.csr_matrix((np.ones(N), (perm, np.arange(N))), shape=(N, N))
        As = (P @ A @ P.T).tocsc()
        elems = np.array([[perm[v - 1] + 1 for v in el.mNodesInd] for el in cg.mElements], dtype=np.int64).T
        Sg = mg.JacobiSmoother(As, None, elems)
        assert Sg.structured
        u0, b = rng.standard_normal(N), rng.standard_normal(N)
        assert rel(mg.smooth(Sg.A, Sg, u0, b, 0.7, 4), jacobi_sweeps(sp.csr_matrix(As), u0, b, 0.7, 4)) < TOL
        assert rel(mg.residual(Sg.A, u0, b), b - As @ u0) < TOL


def test_chain_fallbacks(oracle, mg):
    """element lists that do not describe the operator leave the generic path in charge, with the
    same results"""
    o = oracle
    cg, A, b = cg_level(o, 24, 3)
    N = A.shape[0]
    u0 = o.splitmix_normal(N, 1)
    ref = jacobi_sweeps(sp.csr_matrix(A), u0, b, 2.0 / 3.0, 3)
    elems = np.array([el.mNodesInd for el in cg.mElements], dtype=np.int64).T
    bad = elems.copy()
    bad[[0, 1]] = bad[[1, 0]]              # elements listed right-to-left: consecutive ones no longer chain
    for E in (bad, elems[:, ::2], elems[:3, :]):
        S = mg.JacobiSmoother(A, None, E)
        assert not S.structured
        assert rel(mg.smooth(S.A, S, u0, b, 2.0 / 3.0, 3), ref) < TOL
    # an entry outside the pattern (vertex 0 coupled to vertex 5)
    A2 = sp.lil_matrix(A)
    A2[0, 5] = 0.25
    S = mg.JacobiSmoother(sp.csc_matrix(A2), None, elems)
    assert not S.structured
    with pytest.raises(mg.DimensionMismatch):
        E = elems.copy()
        E[2, 3] = N + 7
        mg.JacobiSmoother(A, None, E)


@pytest.mark.parametrize("p", [1, 2, 3, 4, 6, 8])
def test_chain_detected_from_the_operator_alone(oracle, mg, p):
    """AGGMG_OPT_DETECT_CHAIN: dg_smoother / cg_smoother(..., :jac) WITHOUT element lists (aggmg_jacobi_setup) recognises a
    CG operator in the reference's vertices-first numbering from its pattern -- size n p + 1, every entry inside the
    element chain -- and runs the fused chain kernels, same results; whatever does not fit (a DG operator, a stray
    entry, a scrambled numbering) stays on the generic kernels, as does everything with detect=False"""
    o = oracle
    for n in (1, 5, 64, 301):
        cg, A, b = cg_level(o, n, p)
        S = mg.JacobiSmoother(A)                               # no lists
        N = A.shape[0]
        assert S.structured == (N >= 3), (p, n)                # two unknowns: nothing to fuse
        assert not mg.JacobiSmoother(A, detect=False).structured
        u0 = o.splitmix_normal(N, 31 + p)
        assert rel(mg.smooth(S.A, S, u0, b, 2.0 / 3.0, 4), jacobi_sweeps(sp.csr_matrix(A), u0, b, 2.0 / 3.0, 4)) < TOL
        assert rel(mg.residual(S.A, u0, b), b - A @ u0) < TOL
    cg, A, b = cg_level(o, 40, p)
    N = A.shape[0]
    if N > 8:
        A2 = sp.lil_matrix(A)
        A2[0, N - 2] = 0.25                                    # a coupling no element has
        assert not mg.JacobiSmoother(sp.csc_matrix(A2)).structured
        perm = np.random.default_rng(p).permutation(N)         # a scrambled numbering is not the reference's
        P = sp.csr_matrix((np.ones(N), (perm, np.arange(N))), shape=(N, N))
        if p > 1:
            assert not mg.JacobiSmoother((P @ A @ P.T).tocsc()).structured
    # a DG operator of the same size class is not a CG chain
    mesh, bd = o.model_problem(30)
    dg = o.DgMesh(mesh, 3)
    G, D, C = o.dg_flux_operators(dg, mesh, bd, 1000.0 * 30)
    assert not mg.JacobiSmoother(o.dg_stiffness(dg, G, D, C)).structured


def test_hierarchy_without_meshes_takes_the_chain_kernels(oracle, mg):
    """MeshHierarchy(nothing but operators, :jac smoothers, transfers): the CG levels of the reference's CG-fine
    hierarchy are recognised from the operators and fused with their transfers -- the path a caller without element
    lists used to get was the generic CSR one (r02: 5.15 ms per cycle at 2^22 elements against ~1 ms)"""
    o = oracle
    Ho, b = o.build_cg_hierarchy(48, ps=(4, 2, 1), nDG=1, pDG=0)
    ops = [mg.DeviceOperator(A) for A in Ho.mStiffness]
    sms = [mg.JacobiSmoother(ops[k]) for k in range(3)]
    H = mg.MeshHierarchy(None, ops, sms, Ho.mInterpolation)
    assert H.level_kinds() == ['fused_chain'] * 3 + ['coarsest'], H.level_kinds()
    x = mg.multigrid_v_cycle(H, np.zeros(len(b)), b)
    xr = o.multigrid_v_cycle(Ho, np.zeros(len(b)), b)
    assert np.linalg.norm(Ho.mStiffness[0] @ (x - xr)) <= 1e-12 * np.linalg.norm(b)


def check_vcycle(o, mg, Ho, b, x0=None, nPre=3, nPost=3, alpha=2.0 / 3.0, it_tol=1e-9, kinds=None):
    H = mg.MeshHierarchy.from_reference(Ho)
    if kinds is not None:
        assert H.level_kinds() == kinds, H.level_kinds()
    x0 = np.zeros(len(b)) if x0 is None else x0
    x0c, bc = x0.copy(), b.copy()
    x = mg.multigrid_v_cycle(H, x0, b, nPre=nPre, nPost=nPost, alpha=alpha)
    assert np.array_equal(x0, x0c) and np.array_equal(b, bc)
    xr = o.multigrid_v_cycle(Ho, x0, b, nPre=nPre, nPost=nPost, alpha=alpha)
    A = Ho.mStiffness[0]
    r0 = max(np.linalg.norm(b - A @ x0), np.linalg.norm(b))
    assert np.linalg.norm(A @ (x - xr)) <= TOL * r0
    assert rel(x, xr) < it_tol
    return H


@pytest.mark.parametrize("n", [4, 64, 130, 1000])
def test_vcycle_cg_chain_config5_shape(oracle, mg, n):
    """BASELINE config 5's realisable shape: CG p = 4, 2, 1 then DG p = 0 -- the three CG
    levels run the fused chain kernel (chain transfers between them, the agglomerating one into the DG
    level), V(3,3), alpha = 2/3, zero and random initial guesses"""
    o = oracle
    Ho, b = o.build_cg_hierarchy(n, ps=(4, 2, 1), nDG=1, pDG=0)
    kinds = ['fused_chain', 'fused_chain', 'fused_chain', 'coarsest']
    check_vcycle(o, mg, Ho, b, it_tol=1e-8, kinds=kinds)
    N = len(b)
    check_vcycle(o, mg, Ho, o.splitmix_normal(N, 1), x0=o.splitmix_normal(N, 0), it_tol=1e-8)
    check_vcycle(o, mg, Ho, b, nPre=1, nPost=2, alpha=0.5, it_tol=1e-8)
    check_vcycle(o, mg, Ho, b, nPre=0, nPost=0, it_tol=1e-8)
    check_vcycle(o, mg, Ho, b, nPre=11, nPost=9, it_tol=1e-8)      # sweeps chunked over several launches


def test_vcycle_cg_reference_shapes(oracle, mg):
    """tests/cg_heirarchy_test.jl (CG 8,4,2,1, coarsest CG p=1 stays in the reference numbering),
    dg_cg_heirarchy_test.jl (+ DG p=0) and full_heirarchy_test.jl (CG chain then agglomerated levels:
    the last CG level restricts into 4:1 agglomerates)"""
    o = oracle
    Ho, b = o.build_cg_hierarchy(128, ps=(8, 4, 2, 1))
    check_vcycle(o, mg, Ho, b, it_tol=1e-8, kinds=['fused_chain'] * 3 + ['coarsest'])
    Ho, b = o.build_cg_hierarchy(128, ps=(8, 4, 2, 1), nDG=1)
    check_vcycle(o, mg, Ho, b, it_tol=1e-8, kinds=['fused_chain'] * 4 + ['coarsest'])
    Ho, b = o.build_cg_hierarchy(64, ps=(8, 4, 2, 1), nAgg=5)
    check_vcycle(o, mg, Ho, b, it_tol=1e-9, kinds=['fused_chain'] * 4 + ['fused_btd'] * 4 + ['coarsest'])
    Ho, b = o.build_cg_hierarchy(96, ps=(6, 3), nDG=2, pDG=1)       # odd orders, DG p=1 -> p=0 below
    H = check_vcycle(o, mg, Ho, b, it_tol=1e-8)
    assert H.level_kinds()[:2] == ['fused_chain', 'fused_chain']
    Ho, b = o.build_cg_hierarchy(1024, ps=(1,), nDG=1, pDG=0)        # BASELINE config 1
    check_vcycle(o, mg, Ho, b, it_tol=1e-8, kinds=['fused_chain', 'coarsest'])


def test_chain_level_below_generic_level(oracle, mg):
    """a chain level whose finer neighbour runs the generic kernels exchanges vectors with it in the
    reference numbering (hybrid Schwarz on the finest CG level, src/smoother.jl:24-46)"""
    o = oracle
    Ho, b = o.build_cg_hierarchy(48, ps=(4, 2, 1), nDG=1, pDG=0)
    S = o.cg_smoother(Ho.mMeshes[0], Ho.mStiffness[0], 'hybridSchwarz')
    # element blocks listed in another order than the mesh's: the same smoother, but its lists are no element
    # chain any more, so the level runs the generic kernels
    order = np.random.default_rng(1).permutation(len(S.mBlocks))
    S.mBlocks = [S.mBlocks[i] for i in order]
    S.mBlockInds = S.mBlockInds[:, order]
    Ho.mSmoothers[0] = S
    check_vcycle(o, mg, Ho, b, it_tol=1e-8, kinds=['generic', 'fused_chain', 'fused_chain', 'coarsest'])


@pytest.mark.parametrize("kind", ["addSchwarz", "hybridSchwarz"])
@pytest.mark.parametrize("n,ps", [(5, (4, 2, 1)), (64, (4, 2, 1)), (700, (4, 2, 1)), (300, (8, 4)), (500, (2, 1)), (90, (3,))])
def test_schwarz_levels_run_the_chain_kernel(oracle, mg, kind, n, ps):
    """cg_smoother(mesh, A, :addSchwarz / :hybridSchwarz) (src/smoother.jl:104-134) on every CG level: the overlapping
    element blocks are the element chain, so the sweeps u += alpha [1/count] sum_e R_e' (A_e \\ R_e (b - A u)) run
    in the fused chain kernel (two LDS phases per sweep, two blocks of halo), V-cycles against the oracle"""
    o = oracle
    Ho, b = o.build_cg_hierarchy(n, ps=ps, nDG=1, pDG=0)
    for k in range(len(ps)):
        Ho.mSmoothers[k] = o.cg_smoother(Ho.mMeshes[k], Ho.mStiffness[k], kind)
    alpha = 0.5 if kind == "addSchwarz" else 1.0       # tests/cg_smoother_test.jl
    kinds = ['fused_chain'] * len(ps) + ['coarsest']
    H = check_vcycle(o, mg, Ho, b, alpha=alpha, it_tol=1e-8, kinds=kinds)
    N = len(b)
    check_vcycle(o, mg, Ho, o.splitmix_normal(N, 1), x0=o.splitmix_normal(N, 0), alpha=alpha, it_tol=1e-8)
    check_vcycle(o, mg, Ho, b, nPre=1, nPost=2, alpha=0.4, it_tol=1e-8)
    check_vcycle(o, mg, Ho, b, nPre=9, nPost=7, alpha=alpha, it_tol=1e-7)     # sweeps chunked over several launches
    # the multi-cycle entry point: post- and pre-smoothing of consecutive cycles in one launch, bitwise
    ctx = H.ctx
    db = ctx.to_device(b)
    xa, xb = ctx.to_device(np.zeros(N)), ctx.alloc(N)
    for _ in range(3):
        H.vcycle_dev(xa, db, xb, 3, 3, alpha)
        xa, xb = xb, xa
    out = ctx.alloc(N)
    H.vcycles_dev(ctx.to_device(np.zeros(N)), db, out, 3, 3, 3, alpha)
    assert np.array_equal(out.download(), xa.download())


def test_multigrid_loop_on_chain_hierarchy(oracle, mg):
    """multigrid() (src/solvers.jl:116-139) on the config-5 shape: same iteration count and residual
    history as the oracle, device-resident loop included"""
    o = oracle
    Ho, b = o.build_cg_hierarchy(64, ps=(4, 2, 1), nDG=1, pDG=0)
    H = mg.MeshHierarchy.from_reference(Ho)
    x0 = np.zeros(len(b))
    xo, ito, reso, _ = o.multigrid(Ho, x0, b, 60, 1e-10)
    xg, itg, resg, _ = mg.multigrid(H, x0, b, 60, 1e-10)
    assert itg == ito
    assert np.allclose(resg, reso, rtol=1e-6, atol=1e-12 * reso[0])
    xd, itd, resd, _ = mg.multigrid(H, x0, b, 60, 1e-10, exact=False)
    assert itd == ito and rel(xd, xo) < 1e-8


def test_vcycles_loop_fuses_across_cycles_bitwise(oracle, mg):
    """aggmg_vcycles_dev on a chain fine level: post-smoothing of cycle i and pre-smoothing of cycle i+1 share
    one launch (prolongation + 6 sweeps + residual + restriction); the result is bitwise that of separate cycles,
    also when the fused sweep count needs more than one launch"""
    o = oracle
    Ho, b = o.build_cg_hierarchy(300, ps=(4, 2, 1), nDG=1, pDG=0)
    H = mg.MeshHierarchy.from_reference(Ho)
    ctx = H.ctx
    N = len(b)
    for nPre, nPost, ncyc in ((3, 3, 5), (1, 2, 3), (6, 7, 3), (0, 0, 2)):
        x0 = o.splitmix_normal(N, 3)
        db = ctx.to_device(b)
        xa, xb = ctx.to_device(x0), ctx.alloc(N)
        for _ in range(ncyc):
            H.vcycle_dev(xa, db, xb, nPre, nPost, 0.6)
            xa, xb = xb, xa
        ref = xa.download()
        out = ctx.alloc(N)
        H.vcycles_dev(ctx.to_device(x0), db, out, ncyc, nPre, nPost, 0.6)
        assert np.array_equal(out.download(), ref), (nPre, nPost, ncyc)


@pytest.mark.parametrize("n,ps", [(6, (4, 2, 1)), (64, (4, 2, 1)), (700, (4, 2, 1)), (301, (8, 4)), (500, (1,))])
def test_red_black_element_gauss_seidel_on_cg_levels(oracle, mg, n, ps):
    """EXTENSION (the reference has no Gauss-Seidel smoother; BASELINE.json names a block-GS smoother for the
    CG-fine hierarchy of config 5): red-black ELEMENT Gauss-Seidel on every CG level -- even elements, then odd ones,
    post-smoothing in the reverse order -- in the fused chain kernel, against the oracle's own restatement
    (BlockGaussSeidelRB on the element blocks): no reference parity claim"""
    o = oracle
    Ho, b = o.build_cg_hierarchy(n, ps=ps, nDG=1, pDG=0)
    for k in range(len(ps)):
        Ho.mSmoothers[k] = o.BlockGaussSeidelRB(*o._element_blocks(Ho.mMeshes[k], Ho.mStiffness[k]))
    kinds = ['fused_chain'] * len(ps) + ['coarsest']
    H = check_vcycle(o, mg, Ho, b, alpha=1.0, it_tol=1e-8, kinds=kinds)
    N = len(b)
    check_vcycle(o, mg, Ho, o.splitmix_normal(N, 1), x0=o.splitmix_normal(N, 0), alpha=0.8, it_tol=1e-8)
    check_vcycle(o, mg, Ho, b, nPre=1, nPost=2, alpha=1.0, it_tol=1e-8)
    check_vcycle(o, mg, Ho, b, nPre=5, nPost=4, alpha=1.0, it_tol=1e-7)       # several launches per smoothing step
    # stand-alone sweeps and the smoother factory
    A, cg = Ho.mStiffness[0], Ho.mMeshes[0]
    Sg = mg.cg_smoother(cg, A, 'blockGS')
    assert Sg.structured
    u = o.splitmix_normal(N, 4)
    ref = u
    for _ in range(3):
        ref = Ho.mSmoothers[0].sweep(A, ref, b, 0.9)
    assert rel(mg.smooth(Sg.A, Sg, u, b, 0.9, 3), ref) < TOL
    # the multi-cycle entry point equals separate cycles (no cross-cycle fusion for this smoother)
    ctx = H.ctx
    db = ctx.to_device(b)
    xa, xb = ctx.to_device(np.zeros(N)), ctx.alloc(N)
    for _ in range(3):
        H.vcycle_dev(xa, db, xb, 2, 2, 1.0)
        xa, xb = xb, xa
    out = ctx.alloc(N)
    H.vcycles_dev(ctx.to_device(np.zeros(N)), db, out, 3, 2, 2, 1.0)
    assert np.array_equal(out.download(), xa.download())
