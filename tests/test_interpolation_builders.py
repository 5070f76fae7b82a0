"""the product-side interpolation builders (agglomerationmultigrid1d_amd/interpolation.py,
vectorised, meshes as arrays) against the loop-for-loop oracle (src/interpolation.jl restated) on NON-uniform
meshes with ragged agglomerates: index maps (colptr, rowval -- explicit zeros and duplicate sums included) bit
for bit, values to 1e-13."""
import numpy as np
import pytest
import scipy.sparse as sp

from agglomerationmultigrid1d_amd import interpolation as ip
from agglomerationmultigrid1d_amd._lib import ArgumentError, UnsupportedError


def nonuniform_mesh(o, n, seed):
    """the oracle's mesh with randomly perturbed interior vertices (and our array description of it)"""
    mesh = o.create_uniform_mesh(n, 0.25, 1.75)
    rng = np.random.default_rng(seed)
    h = 1.5 / n
    for v in mesh.mVertices[1:-1]:
        v.mX += 0.35 * h * (2 * rng.random() - 1)
    o.set_boundary(mesh, 0.25, 1.75, (('neu', 0.0), ('dir', 0.0)))
    xv = np.array([v.mX for v in mesh.mVertices])
    return mesh, xv


def ragged(n, sizes):
    """cover 1..n with consecutive groups whose sizes cycle through `sizes` (1-based lists, as the reference's agg)"""
    out, k, i = [], 1, 0
    while k <= n:
        s = min(sizes[i % len(sizes)], n - k + 1)
        out.append(list(range(k, k + s)))
        k += s
        i += 1
    return out


def same(A, B, tol=1e-13):
    A, B = sp.csc_matrix(A), sp.csc_matrix(B)
    B.sort_indices()
    assert A.has_sorted_indices
    assert A.shape == B.shape
    assert np.array_equal(A.indptr, B.indptr), "colptr differs"
    assert np.array_equal(A.indices, B.indices), "rowval differs"
    scale = max(np.abs(B.data).max(), 1e-300)
    assert np.abs(A.data - B.data).max() <= tol * scale


@pytest.mark.parametrize("n,plo,phi", [(7, 1, 2), (12, 2, 4), (9, 1, 4), (5, 4, 8), (6, 3, 3), (1, 1, 2)])
def test_cg_cg(oracle, n, plo, phi):
    o = oracle
    mesh, xv = nonuniform_mesh(o, n, n + plo)
    same(ip.cg_cg_interpolation(ip.CgMesh(xv, plo), ip.CgMesh(xv, phi)),
         o.cg_cg_interpolation(o.CgMesh(mesh, plo), o.CgMesh(mesh, phi)))


@pytest.mark.parametrize("n,plo,phi", [(7, 0, 1), (12, 2, 4), (9, 1, 3), (4, 4, 8)])
def test_dg_dg(oracle, n, plo, phi):
    o = oracle
    mesh, xv = nonuniform_mesh(o, n, 3 * n + phi)
    same(ip.dg_dg_interpolation(ip.DgMesh(xv, plo), ip.DgMesh(xv, phi)),
         o.dg_dg_interpolation(o.DgMesh(mesh, plo), o.DgMesh(mesh, phi)))


@pytest.mark.parametrize("n,pdg,pcg", [(8, 0, 1), (11, 1, 2), (6, 2, 4), (5, 0, 3), (1, 1, 1)])
@pytest.mark.parametrize("flag", [1, 2])
def test_dg_cg(oracle, n, pdg, pcg, flag):
    o = oracle
    mesh, xv = nonuniform_mesh(o, n, 5 * n + pcg)
    same(ip.dg_cg_interpolation(ip.DgMesh(xv, pdg), ip.CgMesh(xv, pcg), flag),
         o.dg_cg_interpolation(o.DgMesh(mesh, pdg), o.CgMesh(mesh, pcg), mesh, flag))


@pytest.mark.parametrize("n,pdg,pcg", [(8, 0, 1), (11, 1, 2), (6, 2, 4), (1, 1, 1)])
def test_dg_cg_dense_flag_0(oracle, n, pdg, pcg):
    """interpFlag = 0: `highMesh.mMassMatrixLU \\ Array(N)` (src/interpolation.jl:205) -- a dense matrix, as in the
    reference; the consistent CG mass matrix of the array mesh against the oracle's assembly"""
    o = oracle
    mesh, xv = nonuniform_mesh(o, n, 7 * n + pcg)
    cg, ocg = ip.CgMesh(xv, pcg), o.CgMesh(mesh, pcg)
    M, Mo = cg.mass_matrix(), ocg.mMassMatrix.tocsc()
    assert abs(M - Mo).max() <= 1e-15 * abs(Mo).max() and M.nnz == Mo.nnz
    L = ip.dg_cg_interpolation(ip.DgMesh(xv, pdg), cg, 0)
    Lo = o.dg_cg_interpolation(o.DgMesh(mesh, pdg), ocg, mesh, 0)
    assert isinstance(L, np.ndarray) and L.shape == Lo.shape
    assert np.abs(L - Lo).max() <= 1e-12 * np.abs(Lo).max()


@pytest.mark.parametrize("n,pcg,sizes", [(12, 2, (4, 2, 3)), (8, 1, (2,))])
def test_aggdg_cg_dense_flag_0(oracle, n, pcg, sizes):
    o = oracle
    mesh, xv = nonuniform_mesh(o, n, 11 * n + pcg)
    agg, a, i = [], 1, 0
    while a <= n:
        s_ = min(sizes[i % len(sizes)], n - a + 1)
        agg.append(list(range(a, a + s_)))
        a, i = a + s_, i + 1
    odg = o.DgMesh(mesh, max(pcg, 1))
    oagg = o.AgglomeratedDgMesh1(1, agg, mesh, odg)
    L = ip.aggdg_cg_interpolation(ip.AgglomeratedDgMesh(1, agg, ip.DgMesh(xv, max(pcg, 1))), ip.CgMesh(xv, pcg), 0)
    Lo = o.aggdg_cg_interpolation(oagg, o.CgMesh(mesh, pcg), mesh, 0)
    assert isinstance(L, np.ndarray) and np.abs(L - Lo).max() <= 1e-12 * np.abs(Lo).max()


@pytest.mark.parametrize("n,plo,phi", [(7, 1, 2), (12, 2, 4), (9, 1, 3), (4, 4, 8), (5, 3, 3)])
def test_dg_dg_interpolation2(oracle, n, plo, phi):
    """src/interpolation.jl:111-139 (outside SURVEY section 8, unused by MeshHierarchy): index maps bit for bit"""
    o = oracle
    mesh, xv = nonuniform_mesh(o, n, 13 * n + phi)
    same(ip.dg_dg_interpolation2(ip.DgMesh(xv, plo), ip.DgMesh(xv, phi)),
         o.dg_dg_interpolation2(o.DgMesh(mesh, plo), o.DgMesh(mesh, phi)))
    with pytest.raises(ArgumentError):
        ip.dg_dg_interpolation2(ip.DgMesh(xv, 0), ip.DgMesh(xv, phi))


@pytest.mark.parametrize("n,plo,phi", [(6, 1, 2), (9, 2, 4), (5, 1, 3), (1, 1, 1)])
def test_cg_cg_interpolation2(oracle, n, plo, phi):
    """src/interpolation.jl:57-85: dense consistent-mass L2 projection between CG spaces"""
    o = oracle
    mesh, xv = nonuniform_mesh(o, n, 17 * n + phi)
    L = ip.cg_cg_interpolation2(ip.CgMesh(xv, plo), ip.CgMesh(xv, phi))
    Lo = o.cg_cg_interpolation2(o.CgMesh(mesh, plo), o.CgMesh(mesh, phi))
    assert isinstance(L, np.ndarray) and L.shape == Lo.shape and np.abs(L - Lo).max() <= 1e-12 * np.abs(Lo).max()
    # an L2 projection reproduces what the coarse space holds: coarse nodal values of a polynomial of degree p_low
    clo, chi = ip.CgMesh(xv, plo), ip.CgMesh(xv, phi)
    f = lambda x: 1.0 + 2.0 * x - (x ** plo)

    def nodal(cg):
        x = np.empty(cg.mNumNodes)
        el = cg.element_nodes() - 1
        xi = cg.ref.nodes
        for i in range(cg.mP + 1):
            x[el[i]] = cg.xc + cg.h / 2.0 * xi[i]
        return f(x)

    assert np.abs(L @ nodal(clo) - nodal(chi)).max() <= 1e-11


def test_bad_arguments(oracle):
    mesh, xv = nonuniform_mesh(oracle, 6, 1)
    with pytest.raises(ValueError):
        ip.dg_cg_interpolation(ip.DgMesh(xv, 1), ip.CgMesh(xv, 2), 3)       # src/interpolation.jl:218
    with pytest.raises(ValueError):
        ip.AgglomeratedDgMesh(2, [[1, 2, 3], [4, 5, 6]], ip.DgMesh(xv, 3))   # agglomerated_dg_mesh.jl:312
    with pytest.raises(ArgumentError):
        ip.AgglomeratedDgMesh(1, [[1, 3], [2, 4, 5, 6]], ip.DgMesh(xv, 3))   # not contiguous
    with pytest.raises(ArgumentError):
        ip.AgglomeratedDgMesh(1, [[1, 2, 3], [4, 5, 6]], ip.DgMesh(xv, 0))   # base mesh needs p >= 1
    with pytest.raises(ArgumentError):
        ip.cg_cg_interpolation(ip.CgMesh(xv, 1), ip.CgMesh(xv[:-1], 2))
    with pytest.raises(ArgumentError):
        ip.DgMesh(xv[::-1], 1)


@pytest.mark.parametrize("n,p,pAgg,sizes", [(13, 3, 1, (4, 2, 3)), (16, 3, 1, (4,)), (9, 1, 0, (2, 1, 3)), (10, 2, 1, (1, 5))])
def test_aggdg_dg_both_variants(oracle, n, p, pAgg, sizes):
    o = oracle
    mesh, xv = nonuniform_mesh(o, n, 7 * n + p)
    agg = ragged(n, sizes)
    obase = o.DgMesh(mesh, p)
    oagg = o.AgglomeratedDgMesh1(pAgg, agg, mesh, obase)
    base = ip.DgMesh(xv, p)
    A = ip.AgglomeratedDgMesh(pAgg, agg, base)
    same(ip.aggdg_dg_interpolation(A, base), o.aggdg_dg_interpolation(oagg, obase))
    same(ip.aggdg_dg_interpolation2(A, base), o.aggdg_dg_interpolation2(oagg, obase), tol=1e-12)
    # the array form of the agglomeration gives the same mesh
    A2 = ip.AgglomeratedDgMesh(pAgg, A.sub_starts, base)
    same(ip.aggdg_dg_interpolation(A2, base), o.aggdg_dg_interpolation(oagg, obase))
    # mass blocks of the agglomerated mesh
    Mb = A.mass_blocks()
    for k, blk in enumerate(oagg.mMassMatrix.mBlocks):
        assert np.abs(Mb[k] - blk).max() <= 1e-13 * np.abs(blk).max()


@pytest.mark.parametrize("n,p,pAgg,s1,s2,s3", [(23, 3, 1, (4, 2, 3), (2, 3), (2,)), (32, 3, 1, (4,), (2,), (2,)),
                                              (14, 2, 0, (3, 1), (1, 2), (3, 2))])
def test_aggdg_aggdg_three_levels(oracle, n, p, pAgg, s1, s2, s3):
    o = oracle
    mesh, xv = nonuniform_mesh(o, n, 11 * n + p)
    obase = o.DgMesh(mesh, p)
    base = ip.DgMesh(xv, p)
    agg1 = ragged(n, s1)
    o1 = o.AgglomeratedDgMesh1(pAgg, agg1, mesh, obase)
    m1 = ip.AgglomeratedDgMesh(pAgg, agg1, base)
    agg2 = ragged(len(agg1), s2)
    o2 = o.AgglomeratedDgMeshN(pAgg, agg2, o1, obase)
    m2 = ip.AgglomeratedDgMesh(pAgg, agg2, m1)
    agg3 = ragged(len(agg2), s3)
    o3 = o.AgglomeratedDgMeshN(pAgg, agg3, o2, obase)
    m3 = ip.AgglomeratedDgMesh(pAgg, agg3, m2)
    same(ip.aggdg_aggdg_interpolation(m2, m1, base), o.aggdg_aggdg_interpolation(o2, o1, obase), tol=1e-12)
    same(ip.aggdg_aggdg_interpolation(m3, m2), o.aggdg_aggdg_interpolation(o3, o2, obase), tol=1e-12)
    for om, mm in ((o2, m2), (o3, m3)):
        Mb = mm.mass_blocks()
        for k, blk in enumerate(om.mMassMatrix.mBlocks):
            assert np.abs(Mb[k] - blk).max() <= 1e-13 * np.abs(blk).max()


@pytest.mark.parametrize("n,pcg,pAgg,sizes", [(13, 1, 1, (4, 2, 3)), (12, 2, 1, (3,)), (9, 4, 0, (2, 1, 3)), (7, 3, 1, (7,))])
@pytest.mark.parametrize("flag", [1, 2])
def test_aggdg_cg(oracle, n, pcg, pAgg, sizes, flag):
    o = oracle
    mesh, xv = nonuniform_mesh(o, n, 13 * n + pcg)
    agg = ragged(n, sizes)
    ocg = o.CgMesh(mesh, pcg)
    # the agglomerated mesh is built over a DG mesh on the same faces (its bounding boxes read the end nodes)
    odg = o.DgMesh(mesh, max(pcg, 1))
    oagg = o.AgglomeratedDgMesh1(pAgg, agg, mesh, odg)
    A = ip.AgglomeratedDgMesh(pAgg, agg, ip.DgMesh(xv, max(pcg, 1)))
    same(ip.aggdg_cg_interpolation(A, ip.CgMesh(xv, pcg), flag), o.aggdg_cg_interpolation(oagg, ocg, mesh, flag))


def test_uniform_generator_agrees_with_the_general_builders():
    """the O(n) uniform-mesh generator (uniform.py) and the general builders on its mesh: same maps, same values"""
    from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy
    n = 64
    U = UniformDgAggHierarchy(n, p=3, pAgg=1, ratios=(4, 2, 2), workers=1)
    xv = 0.0 + (np.arange(n + 1) / n) * 1.0
    base = ip.DgMesh(xv, 3)
    m1 = ip.AgglomeratedDgMesh(1, np.arange(0, n + 1, 4), base)
    m2 = ip.AgglomeratedDgMesh(1, np.arange(0, n // 4 + 1, 2), m1)
    m3 = ip.AgglomeratedDgMesh(1, np.arange(0, n // 8 + 1, 2), m2)
    same(ip.aggdg_dg_interpolation(m1, base), U.interpolation_csc(0))
    same(ip.aggdg_aggdg_interpolation(m2, m1), U.interpolation_csc(1), tol=1e-12)
    same(ip.aggdg_aggdg_interpolation(m3, m2), U.interpolation_csc(2), tol=1e-12)
