"""GPU parity of the set-up products: everything between an uploaded
SparseMatrixCSC and the first V-cycle is computed on the device (csrc/setup.hip, csrc/spops.hip) and is
compared here with the CPU oracle at small n --
  * K6, the batched block LU (dg_smoother(:blockJac), src/smoother.jl:153-165): explicit inverses against
    a line-by-line getf2 restatement (bit for bit: the set-up kernels are compiled without FMA contraction)
    and against LAPACK; SingularException with the block number;
  * BlockDiagonal * sparse and BlockDiagonalLU \\ sparse (src/block_diagonal.jl:195-264, 314-383): index maps
    bit-exact (all m rows of every touched block, zeros included), values to round-off;
  * sparse * sparse, sparse - sparse, transpose; the recurrences of the DG-fine constructor
    (src/mesh_heirarchy.jl:79-84,98-103,140-181) run on the device, level by level against the oracle's
    MeshHierarchy."""
import math

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mg():
    import agglomerationmultigrid1d_amd as mg
    mg.default_context()
    return mg


def relmax(A, B):
    A, B = sp.csc_matrix(A), sp.csc_matrix(B)
    return abs(A - B).max() / max(abs(B).max(), 1e-300)


def same_maps(A, B, drop_zeros=False):
    """drop_zeros: SparseArrays' `*` keeps entries that cancel (so does the device product), SciPy's -- the
    oracle's -- prunes exact zeros, and whether a cancellation is exact depends on the last bit of the
    factors: compare the patterns of the entries above 1e-13 of the largest one then (as
    tests/test_uniform_setup.py does for the stiffness patterns)"""
    A, B = sp.csc_matrix(A, copy=True), sp.csc_matrix(B, copy=True)
    if drop_zeros:
        thr = 1e-13 * max(abs(B).max(), 1e-300)
        for M in (A, B):
            M.data[np.abs(M.data) <= thr] = 0.0
            M.eliminate_zeros()
    A.sort_indices(), B.sort_indices()
    return A.shape == B.shape and np.array_equal(A.indptr, B.indptr) and np.array_equal(A.indices, B.indices)


def getf2_inverse(a):
    """partial-pivot LU in LAPACK getf2 order, then the inverse column by column (the operation sequence of
    lu_invert in csrc/setup_kernels.hpp), plain Python floats"""
    m = a.shape[0]
    a = a.copy()
    piv = [0] * m
    for k in range(m):
        p = k + int(np.argmax(np.abs(a[k:, k])))
        piv[k] = p
        if a[p, k] == 0.0:
            raise ZeroDivisionError
        if p != k:
            a[[k, p], :] = a[[p, k], :]
        rp = 1.0 / a[k, k]
        for i in range(k + 1, m):
            a[i, k] *= rp
        for i in range(k + 1, m):
            l = a[i, k]
            for j in range(k + 1, m):
                a[i, j] -= l * a[k, j]
    inv = np.zeros((m, m))
    for c in range(m):
        x = np.zeros(m)
        x[c] = 1.0
        for k in range(m):
            if piv[k] != k:
                x[k], x[piv[k]] = x[piv[k]], x[k]
        for i in range(1, m):
            s = x[i]
            for j in range(i):
                s -= a[i, j] * x[j]
            x[i] = s
        for i in range(m - 1, -1, -1):
            s = x[i]
            for j in range(i + 1, m):
                s -= a[i, j] * x[j]
            x[i] = s / a[i, i]
        inv[:, c] = x
    return inv


@pytest.mark.parametrize("p", [0, 1, 2, 3, 4, 7, 8])
def test_k6_block_lu_matches_getf2_restatement(oracle, mg, p):
    o = oracle
    n = 40
    mesh, bd = o.model_problem(n)
    dg = o.DgMesh(mesh, p)
    G, D, C = o.dg_flux_operators(dg, mesh, bd, 1000.0 * n)
    A = o.dg_stiffness(dg, G, D, C)
    S = mg.dg_smoother(dg, A, 'blockJac')
    inv = S.inverse_blocks()
    Ad = A.toarray()
    m = p + 1
    worst = 0.0
    for k in range(n):
        blk = Ad[k * m:(k + 1) * m, k * m:(k + 1) * m]
        ref = getf2_inverse(blk)
        assert np.allclose(inv[k], np.linalg.inv(blk), rtol=1e-11, atol=1e-14 * np.abs(ref).max())
        worst = max(worst, np.abs(inv[k] - ref).max() / np.abs(ref).max())
    assert worst == 0.0, worst          # same IEEE operations in the same order as the host LU it replaces


def test_k6_non_contiguous_blocks_and_singular_block(oracle, mg):
    o = oracle
    n = 12
    mesh = o.create_uniform_mesh(n, 0.0, 1.0)
    bd = o.set_boundary(mesh, 0.0, 1.0, [('dir', 0.0), ('dir', 0.5)])
    cg = o.CgMesh(mesh, 3)
    A, _ = o.cg_stiffness_and_rhs(cg, mesh, lambda x: 1.0, bd)
    S = mg.cg_smoother(cg, A, 'addSchwarz')            # overlapping vertex-sharing element blocks
    inv = S.inverse_blocks()
    Ad = A.toarray()
    for k, el in enumerate(cg.mElements):
        idx = np.array(el.mNodesInd) - 1
        assert np.abs(inv[k] - getf2_inverse(Ad[np.ix_(idx, idx)])).max() == 0.0
    # a singular diagonal block: SingularException naming the (1-based) block, as la.lu would raise it
    dg = o.DgMesh(mesh, 1)
    B = sp.lil_matrix((2 * n, 2 * n))
    B.setdiag(1.0)
    B[6, 6], B[6, 7], B[7, 6], B[7, 7] = 1.0, 2.0, 2.0, 4.0
    with pytest.raises(mg.SingularException, match="singular block 4"):
        mg.dg_smoother(dg, sp.csc_matrix(B), 'blockJac')
    with pytest.raises(mg.SingularException, match="singular block 2"):
        mg.BlockDiagonal([np.eye(2), np.zeros((2, 2)), np.eye(2)]).lu()


def test_block_diagonal_sparse_operands(oracle, mg):
    """tests/blockdiagonal_test.jl: BD * sparse and LU \\ sparse against the dense equivalents"""
    o = oracle
    rng = np.random.default_rng(3)
    for m, nb, ncols in ((3, 17, 9), (4, 60, 240), (1, 20, 5), (8, 9, 30)):
        blocks = [rng.standard_normal((m, m)) + 3.0 * np.eye(m) for _ in range(nb)]
        S = sp.random(m * nb, ncols, density=0.08, random_state=5, format='csc')
        S.data[::7] = 0.0                                      # stored zeros stay stored
        Ao = o.BlockDiagonal(blocks)
        Ag = mg.BlockDiagonal(blocks)
        ref = Ao.mul_sparse(S)
        got = (Ag @ S).to_scipy()
        assert same_maps(got, ref) and relmax(got, ref) < 1e-14
        ref = Ao.lu().solve_sparse(S)
        got = Ag.lu().solve(S).to_scipy()
        assert same_maps(got, ref) and relmax(got, ref) < 1e-12
        assert relmax(got, np.linalg.solve(Ao.todense(), S.toarray())) < 1e-12
    with pytest.raises(mg.DimensionMismatch):
        mg.BlockDiagonal([np.eye(2)] * 3) @ sp.identity(5, format='csc')


def test_sparse_products_and_transpose(oracle, mg):
    o = oracle
    Ho, _ = o.build_dg_agg_hierarchy(32, p=3, pAgg=1, nAgg=2)
    G, L = Ho.mGradient[0], Ho.mInterpolation[0]
    dG, dL = mg.DeviceOperator(G), mg.DeviceOperator(L, kind=1)
    Lt = dL.transpose()
    assert same_maps(Lt.to_scipy(), sp.csc_matrix(L.T)) and np.array_equal(Lt.to_scipy().toarray(), L.T.toarray())
    ref = (L.T @ G @ L).tocsc()
    got = Lt.matmul(dG).matmul(dL).to_scipy()
    assert same_maps(got, ref, drop_zeros=True) and relmax(got, ref) < 1e-14
    assert got.nnz >= ref.nnz
    C, X = Ho.mC[0], (Ho.mDivergence[0] @ Ho.mMeshes[0].mMassMatrixLU.solve_sparse(G)).tocsc()
    ref = o.sp_sub(C, X)
    got = mg.DeviceOperator(C).sub(X).to_scipy()
    assert same_maps(got, ref) and relmax(got, ref) < 1e-15
    Z = mg.DeviceOperator(G).sub(G).to_scipy()                 # everything cancels: nothing stored
    assert Z.nnz == 0
    with pytest.raises(mg.DimensionMismatch):
        dG.matmul(sp.identity(7, format='csc'))


def test_product_size_guard_is_taken_in_64_bits(mg):
    """the entry total of a set-up product is formed in 64 bits before the int32 scan: counts that add up
    to 2^31 or more are refused (UnsupportedError), they do not wrap into a small or negative total"""
    import ctypes
    ctx = mg.default_context()
    lib = ctx.lib

    def scan(counts):
        c = np.ascontiguousarray(counts, dtype=np.int32)
        tot = ctypes.c_int64(-1)
        st = lib.aggmg_debug_scan_counts(ctx.handle, c.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), len(c), ctypes.byref(tot))
        return st, tot.value

    st, tot = scan([3, 0, 5, 7])
    assert st == 0 and tot == 15
    st, tot = scan([2 ** 30] * 3 + [17])                       # int32 arithmetic: 3 * 2^30 + 17 wraps to -1073741807
    assert tot == 3 * 2 ** 30 + 17
    with pytest.raises(mg.UnsupportedError):
        ctx.check(st)
    st, tot = scan([2 ** 30, 2 ** 30 - 1])                     # 2^31 - 1 entries still fit
    assert st == 0 and tot == 2 ** 31 - 1
    st, tot = scan([2 ** 30, 2 ** 30])
    assert tot == 2 ** 31 and st != 0


@pytest.mark.parametrize("n,p,pAgg,nAgg,first", [(32, 3, 1, 3, 4), (48, 2, 0, 2, 2), (64, 1, 1, 3, 4)])
def test_dg_constructor_recurrences_on_device(oracle, mg, n, p, pAgg, nAgg, first):
    """MeshHierarchy.from_dg_operators: Galerkin products, A = C - D (M_LU \\ G) and the block smoothers of
    every level on the device, given the fine operators, the L_k and the mass matrices -- against the
    oracle's constructor (D4 extension), then one V-cycle"""
    o = oracle
    Ho, b = o.build_dg_agg_hierarchy(n, p=p, pAgg=pAgg, nAgg=nAgg, first=first)
    masses = [mg.BlockDiagonal(m.mMassMatrix.mBlocks) for m in Ho.mMeshes[1:]]
    H = mg.MeshHierarchy.from_dg_operators(Ho.mMeshes, Ho.mStiffness[0], Ho.mGradient[0], Ho.mDivergence[0], Ho.mC[0],
                                           Ho.mInterpolation, masses)
    for k in range(1, len(Ho.mMeshes)):
        for got, ref in ((H.mGradient[k], Ho.mGradient[k]), (H.mDivergence[k], Ho.mDivergence[k]), (H.mC[k], Ho.mC[k])):
            assert same_maps(got.to_scipy(), ref, drop_zeros=True) and relmax(got.to_scipy(), ref) < 1e-13, k
        # A = C - D (M \\ G): a numerical pattern (entries that cancel to round-off may or may not survive)
        Ag, Ar = H.mStiffness[k].to_scipy(), Ho.mStiffness[k]
        assert relmax(Ag, Ar) < 1e-12, k
    assert all(kind == 'fused_btd' for kind in H.level_kinds()[:-1])
    x = mg.multigrid_v_cycle(H, np.zeros(len(b)), b)
    xr = o.multigrid_v_cycle(Ho, np.zeros(len(b)), b)
    assert np.linalg.norm(Ho.mStiffness[0] @ (x - xr)) <= 1e-12 * np.linalg.norm(b)
    assert np.linalg.norm(x - xr) <= 1e-8 * np.linalg.norm(xr)


@pytest.mark.parametrize("kw", [dict(n=32, ps=(4, 2, 1)), dict(n=32, ps=(4, 2, 1), nDG=1, pDG=0), dict(n=32, ps=(2, 1), nAgg=3)])
def test_cg_constructor_recurrences_on_device(oracle, mg, kw):
    """MeshHierarchy.from_cg_operators: Galerkin operators of the CG chain and of the levels below the first
    (re-discretised) DG / agglomerated level on the device -- against the oracle's CG-fine constructor"""
    o = oracle
    Ho, b = o.build_cg_hierarchy(**kw)
    nCG = len(kw["ps"])
    n = len(Ho.mMeshes)
    dg_ops = (Ho.mGradient[0], Ho.mDivergence[0], Ho.mC[0]) if n > nCG else None
    masses = [mg.BlockDiagonal(m.mMassMatrix.mBlocks) for m in Ho.mMeshes[nCG:]]
    H = mg.MeshHierarchy.from_cg_operators(Ho.mMeshes, Ho.mStiffness[0], Ho.mInterpolation, nCG, dg_ops, masses)
    for k in range(1, n):
        assert relmax(H.mStiffness[k].to_scipy(), Ho.mStiffness[k]) < 1e-12, k
    kinds = H.level_kinds()
    assert kinds[:nCG - (1 if n == nCG else 0)] == ['fused_chain'] * (nCG - (1 if n == nCG else 0)), kinds
    x = mg.multigrid_v_cycle(H, np.zeros(len(b)), b)
    xr = o.multigrid_v_cycle(Ho, np.zeros(len(b)), b)
    assert np.linalg.norm(Ho.mStiffness[0] @ (x - xr)) <= 1e-12 * np.linalg.norm(b)


@pytest.mark.parametrize("n,sizes0", [(46, (4, 2, 3)), (1403, (4, 2, 3)), (1403, (12, 2, 11))])
def test_nonuniform_mesh_ragged_agglomerates_end_to_end(oracle, mg, n, sizes0):
    """a12 + a13 on a mesh the uniform generator cannot make: perturbed vertices, agglomerates of 4/2/3 (or 12/2/11: larger
    than the extra halo the agglomerate-aligned tile ownership allows, i.e. the two-part atomic restriction), then 2/3,
    then 2 sub-elements.  The fine-level G, D, C, A come from the assembly (the oracle's restatement: out of scope);
    every L_k and mass block from the product's builders (interpolation.py), the Galerkin recurrences, A_k and the
    block smoothers from the device -- against the oracle's constructor on the same mesh, then V-cycles."""
    from agglomerationmultigrid1d_amd import interpolation as ip
    o = oracle
    p, pAgg = 3, 1
    mesh = o.create_uniform_mesh(n, 0.0, 1.0)
    rng = np.random.default_rng(3)
    for v in mesh.mVertices[1:-1]:
        v.mX += 0.3 / n * (2 * rng.random() - 1)
    bd = o.set_boundary(mesh, 0.0, 1.0, [('neu', 0.0), ('dir', 0.54)])

    def ragged(k, sizes):
        out, a, i = [], 1, 0
        while a <= k:
            s = min(sizes[i % len(sizes)], k - a + 1)
            out.append(list(range(a, a + s)))
            a, i = a + s, i + 1
        return out

    aggs = [ragged(n, sizes0)]
    aggs.append(ragged(len(aggs[0]), (2, 3)))
    aggs.append(ragged(len(aggs[1]), (2,)))
    dg = o.DgMesh(mesh, p)
    omeshes = [dg, o.AgglomeratedDgMesh1(pAgg, aggs[0], mesh, dg)]
    for a in aggs[1:]:
        omeshes.append(o.AgglomeratedDgMeshN(pAgg, a, omeshes[-1], dg))
    CDir = 1000.0 * n
    G, D, C = o.dg_flux_operators(dg, mesh, bd, CDir)
    A = o.dg_stiffness(dg, G, D, C)
    f, r = o.dg_flux_rhs(dg, mesh, math.cos, bd, CDir)
    b = o.dg_rhs(dg, D, f, r)
    Ho = o.MeshHierarchy_dg(omeshes, [bd] * len(omeshes), A, G, D, C, nDG=1, nAgg=3)
    # product side: meshes as arrays, builders, device recurrences
    xv = np.array([v.mX for v in mesh.mVertices])
    meshes = [ip.DgMesh(xv, p)]
    for a in aggs:
        meshes.append(ip.AgglomeratedDgMesh(pAgg, a, meshes[-1]))
    Ls = [ip.aggdg_dg_interpolation(meshes[1], meshes[0])] + \
         [ip.aggdg_aggdg_interpolation(meshes[k + 1], meshes[k]) for k in range(1, 3)]
    for L, Lo in zip(Ls, Ho.mInterpolation):
        assert same_maps(L, Lo) and relmax(L, Lo) < 1e-12
    masses = [mg.BlockDiagonal(list(m.mass_blocks())) for m in meshes[1:]]
    H = mg.MeshHierarchy.from_dg_operators(meshes, A, G, D, C, Ls, masses)
    for k in range(1, len(meshes)):
        assert relmax(H.mStiffness[k].to_scipy(), Ho.mStiffness[k]) < 1e-11, k
    # agglomerates of different sizes run the fused kernels too (parent / first-child maps instead of one ratio; at
    # n = 1403 the fine level has a dozen tiles, so agglomerates are cut by tile boundaries and restricted in two parts)
    assert all(kind == 'fused_btd' for kind in H.level_kinds()[:-1]), H.level_kinds()
    x, xr = np.zeros(len(b)), np.zeros(len(b))
    for _ in range(3):
        x = mg.multigrid_v_cycle(H, x, b)
        xr = o.multigrid_v_cycle(Ho, xr, b)
    assert np.linalg.norm(A @ (x - xr)) <= 1e-12 * np.linalg.norm(b)
    assert np.linalg.norm(A @ x - b) < 0.05 * np.linalg.norm(b)
    # the multi-cycle entry point (post- and pre-smoothing of consecutive cycles in one launch) and determinism
    ctx = H.ctx
    yd = ctx.alloc(len(b))
    H.vcycles_dev(ctx.to_device(np.zeros(len(b))), ctx.to_device(b), yd, 3)
    # (bit for bit the separate cycles, as with agglomerates of one size: the tiles' owned ranges sit on agglomerate
    # boundaries, so every agglomerate is restricted by one thread over all its children in ascending order -- whatever
    # the halo depth of the launch; before r03 the two parts of a cut agglomerate were added atomically: round-off)
    y = yd.download()
    if max(sizes0) <= 9:
        assert np.array_equal(y, x)
    else:   # agglomerates cut by tile boundaries, two atomic adds each: round-off
        assert np.linalg.norm(A @ (y - x)) <= 1e-13 * np.linalg.norm(b) and np.linalg.norm(y - x) <= 1e-9 * np.linalg.norm(x)
    x2 = np.zeros(len(b))
    for _ in range(3):
        x2 = mg.multigrid_v_cycle(H, x2, b)
    assert np.array_equal(x2, x)


def test_nonuniform_cg_chain_with_dg0_end_to_end(oracle, mg):
    """CG p = 4 -> 2 -> 1 -> DG p = 0 on a perturbed mesh (the shape of BASELINE config 5): transfers from the
    product's builders (cg_cg_interpolation, dg_cg_interpolation with the lumped mass), Galerkin operators and the
    point-Jacobi chain forms from the device, the re-discretised DG level's G, D, C from the assembly -- against the
    oracle's CG-fine constructor on the same mesh"""
    from agglomerationmultigrid1d_amd import interpolation as ip
    o = oracle
    n, ps = 37, (4, 2, 1)
    mesh = o.create_uniform_mesh(n, 0.0, 1.0)
    rng = np.random.default_rng(8)
    for v in mesh.mVertices[1:-1]:
        v.mX += 0.3 / n * (2 * rng.random() - 1)
    bd = o.set_boundary(mesh, 0.0, 1.0, [('neu', -0.1), ('dir', 0.54)])
    omeshes = [o.CgMesh(mesh, p) for p in ps] + [o.DgMesh(mesh, 0)]
    A, b = o.cg_stiffness_and_rhs(omeshes[0], mesh, math.cos, bd)
    CDir = 1000.0 * n
    Ho = o.MeshHierarchy_cg(omeshes, mesh, [bd] * 4, A, nCG=3, nDG=1, nAgg=0, CDir=CDir)
    xv = np.array([v.mX for v in mesh.mVertices])
    meshes = [ip.CgMesh(xv, p) for p in ps] + [ip.DgMesh(xv, 0)]
    Ls = [ip.cg_cg_interpolation(meshes[k + 1], meshes[k]) for k in range(2)] + \
         [ip.dg_cg_interpolation(meshes[3], meshes[2], 1)]
    for L, Lo in zip(Ls, Ho.mInterpolation):
        assert same_maps(L, Lo) and relmax(L, Lo) < 1e-12
    dg_ops = (Ho.mGradient[0], Ho.mDivergence[0], Ho.mC[0])
    masses = [mg.BlockDiagonal(omeshes[3].mMassMatrix.mBlocks)]
    H = mg.MeshHierarchy.from_cg_operators(meshes, A, Ls, 3, dg_ops, masses)
    for k in range(1, 4):
        assert relmax(H.mStiffness[k].to_scipy(), Ho.mStiffness[k]) < 1e-11, k
    assert H.level_kinds()[:3] == ['fused_chain'] * 3, H.level_kinds()
    x, xr = np.zeros(len(b)), np.zeros(len(b))
    for _ in range(3):
        x = mg.multigrid_v_cycle(H, x, b)
        xr = o.multigrid_v_cycle(Ho, xr, b)
    assert np.linalg.norm(A @ (x - xr)) <= 1e-12 * np.linalg.norm(b)
