"""The reference's interpolation builders (src/interpolation.jl) for 1-D meshes given by
their arrays: `L_k` values AND index maps, O(n), vectorised, with the reference's names and argument order.

The reference walks its mesh objects element by element, pushes COO triples and calls
`sparse(I, J, V, m, n)` (duplicates summed, explicit zeros kept, rows ascending per column).  Here a
mesh is described by arrays -- vertex coordinates, polynomial degree, for agglomerated meshes the first
sub-element of every agglomerate -- and every builder writes its `colptr / rowval / nzval` directly:
same pattern (zeros kept), same entry order, values equal to round-off (the quadrature sums are formed
once on the reference element and scaled, instead of point by point per element).

    DgMesh(xv, p)                   src/dg_mesh.jl:16-26,58-111        nodes k(p+1)+i, local order [left, right, interior]
    CgMesh(xv, p)                   src/cg_mesh.jl:12-20,54-79         vertices first, then p-1 interior nodes per element
    AgglomeratedDgMesh(p, agg, sub) src/agglomerated_dg_mesh.jl:32-72  agg: lists of CONTIGUOUS sub-element indices
                                    (sub = DgMesh: ctor :391-495; sub = AgglomeratedDgMesh: :593-635)

    cg_cg_interpolation(low, high)                 src/interpolation.jl:5-55
    dg_dg_interpolation(low, high)                 :91-109
    cg_cg_interpolation2(low, high)                :57-85    (dense L2 projection, as in the reference)
    dg_dg_interpolation2(low, high)                :111-139
    dg_cg_interpolation(low, high, interpFlag)     :145-220   (interpFlag 0 -- dense, as in the reference --, 1, 2)
    aggdg_aggdg_interpolation(coarse, fine)        :226-264
    aggdg_dg_interpolation(agg, base)              :270-292
    aggdg_dg_interpolation2(agg, base)             :294-324
    aggdg_cg_interpolation(agg, base, interpFlag)  :330-410   (interpFlag 0 -- dense --, 1, 2)

Every builder returns a SciPy CSC matrix with sorted indices (hand it to DeviceOperator / MeshHierarchy).
Checked against the loop-for-loop oracle on non-uniform meshes with ragged agglomerates in
tests/test_interpolation_builders.py: index maps bit for bit, values to 1e-13."""
import numpy as np
import scipy.sparse as sp

from ._lib import ArgumentError, UnsupportedError
from .uniform import RefElement, gauss_quad, legendre_vandermonde

__all__ = ["DgMesh", "CgMesh", "AgglomeratedDgMesh", "cg_cg_interpolation", "dg_dg_interpolation",
           "dg_cg_interpolation", "aggdg_aggdg_interpolation", "aggdg_dg_interpolation",
           "aggdg_dg_interpolation2", "aggdg_cg_interpolation", "evaluate_local_modal_basis_fun"]


# ------------------------------------------------------------------------------------------------
# meshes as arrays
# ------------------------------------------------------------------------------------------------
class _Mesh1d:
    def __init__(self, xv, p):
        xv = np.ascontiguousarray(xv, dtype=np.float64)
        if xv.ndim != 1 or xv.size < 2 or not np.all(np.diff(xv) > 0):
            raise ArgumentError("mesh vertices must be an ascending 1-D array with at least two entries")
        self.xv = xv
        self.n = xv.size - 1
        self.mP = int(p)
        self.ref = RefElement(self.mP)
        self.h = xv[1:] - xv[:-1]                     # face.mVertices[2].mX - face.mVertices[1].mX
        self.xc = (xv[:-1] + xv[1:]) / 2.0
        self.J = self.h / 2.0                         # mJacobian

    def ref_map(self, xi):
        """mRefMap of every element at reference points xi -> (n, len(xi))"""
        return self.xc[:, None] + self.h[:, None] / 2.0 * np.asarray(xi, dtype=np.float64)[None, :]


def _contiguous_block_inds(m, n):
    """mBlockInds of a DG-numbered mesh: (m x n), 1-based, node i of element k = k m + i + 1"""
    return (np.arange(n, dtype=np.int64)[None, :] * m + np.arange(1, m + 1, dtype=np.int64)[:, None])


class DgMesh(_Mesh1d):
    """DgMesh (src/dg_mesh.jl:16-26): element k owns the nodes k(p+1) .. k(p+1)+p (0-based)."""

    def __init__(self, xv, p):
        super().__init__(xv, p)
        self.m = self.mP + 1
        self.mNumNodes = self.n * self.m

    @property
    def mBlockInds(self):
        """mMassMatrix.mBlockInds (what dg_smoother(mesh, A, :blockJac) reads, src/smoother.jl:153-165)"""
        return _contiguous_block_inds(self.m, self.n)


class CgMesh(_Mesh1d):
    """CgMesh (src/cg_mesh.jl:12-20): vertices 0..n first, then the p-1 interior nodes of every element in
    element order (src/cg_mesh.jl:37-45,59-65)."""

    def __init__(self, xv, p):
        if p < 1:
            raise ArgumentError("CgMesh needs p >= 1")
        super().__init__(xv, p)
        self.q = self.mP - 1
        self.mNumNodes = self.n * self.mP + 1

    def element_nodes(self):
        """mElements[e].mNodesInd as the (p+1) x n matrix (1-based): the hint cg_smoother passes to
        aggmg_jacobi_setup_elements"""
        n, q = self.n, self.q
        k = np.arange(n, dtype=np.int64)
        out = np.empty((self.mP + 1, n), dtype=np.int64)
        out[0], out[1] = k + 1, k + 2
        for j in range(q):
            out[2 + j] = (n + 1) + k * q + j + 1
        return out

    @property
    def mBlockInds(self):
        """the element node lists in the slot cg_smoother(mesh, A, ...) looks at (src/smoother.jl:88-139)"""
        return self.element_nodes()

    def mass_matrix(self):
        """mMassMatrix (src/cg_mesh.jl:67-75): J_e * M_ref of every element scattered to its nodes, duplicates at shared
        vertices summed (element k-1 first, the order `sparse` meets them) -> scipy CSC"""
        import scipy.sparse as sp
        el = self.element_nodes() - 1                                   # (p+1, n)
        m = self.mP + 1
        rows = np.broadcast_to(el.T[:, :, None], (self.n, m, m))
        cols = np.broadcast_to(el.T[:, None, :], (self.n, m, m))
        vals = self.J[:, None, None] * self.ref.mass[None, :, :]
        return sp.csc_matrix((vals.reshape(-1), (rows.reshape(-1), cols.reshape(-1))), shape=(self.mNumNodes, self.mNumNodes))

    def lumped_mass(self):
        """lumped[j] = sum(mMassMatrix[j, :]) (src/interpolation.jl:207-211): the rows of J_e * M_ref summed, vertex
        rows over the two elements that share the vertex (element v-1 first, as the assembly order has it)"""
        rs = self.ref.mass.sum(axis=1)
        lum = np.zeros(self.mNumNodes)
        lum[1:self.n + 1] += self.J * rs[1]
        lum[0:self.n] += self.J * rs[0]
        if self.q:
            lum[self.n + 1:] = (self.J[:, None] * rs[None, 2:]).reshape(-1)
        return lum


def evaluate_local_modal_basis_fun(p, lo, hi, x):
    """src/agglomerated_dg_mesh.jl:297-315 for boxes [lo, hi] broadcast against points x -> (..., p+1)"""
    if p not in (0, 1):
        raise ValueError("Only implemented for p = 0 and p = 1.")
    out = np.ones(np.broadcast_shapes(np.shape(x), np.shape(lo)) + (p + 1,))
    if p == 1:
        xC = (lo + hi) / 2.0
        out[..., 1] = 2 * (x - xC) / (hi - lo)
    return out


class AgglomeratedDgMesh:
    """AgglomeratedDgMesh{1} / {N} (src/agglomerated_dg_mesh.jl:32-72).  `agg`: the reference's list of lists of
    sub-element indices (1-based), every list a contiguous ascending run and the lists in mesh order -- what an
    agglomeration of a 1-D mesh is -- or the array of first sub-elements (0-based, length ne+1).  `sub`: the DgMesh
    itself (first agglomerated level) or the previous AgglomeratedDgMesh."""

    def __init__(self, p, agg, sub):
        if p not in (0, 1):
            raise ValueError("Only implemented for p = 0 and p = 1.")       # agglomerated_dg_mesh.jl:312
        self.mP = int(p)
        self.m = self.mP + 1
        self.sub = sub
        self.base = sub if isinstance(sub, DgMesh) else sub.base
        if self.base.mP < 1:
            # AgglomeratedDgElement reads baseMesh.mElements[k].mNodesX[2] (src/agglomerated_dg_mesh.jl:183-189)
            raise ArgumentError("agglomeration needs a base mesh with p >= 1")
        nsub = sub.n
        if len(agg) and np.ndim(agg[0]) == 0:
            starts = np.asarray(agg, dtype=np.int64)
        else:
            flat = np.concatenate([np.asarray(g, dtype=np.int64) for g in agg]) if len(agg) else np.zeros(0, np.int64)
            if flat.size != nsub or not np.array_equal(flat, np.arange(1, nsub + 1)):
                raise ArgumentError("agglomerates must be contiguous runs of sub-elements covering the mesh in order")
            starts = np.concatenate([[0], np.cumsum([len(g) for g in agg])]).astype(np.int64)
        if starts.size < 2 or starts[0] != 0 or starts[-1] != nsub or not np.all(np.diff(starts) > 0):
            raise ArgumentError("agglomerate boundaries must ascend from 0 to the number of sub-elements")
        self.sub_starts = starts                                  # first sub-element of every agglomerate
        self.n = starts.size - 1
        self.mNumNodes = self.n * self.m
        self.base_starts = starts if isinstance(sub, DgMesh) else sub.base_starts[starts]
        b = self.base
        # bounding boxes from the end nodes mRefMap(-1), mRefMap(+1) of the first / last base element
        # (src/agglomerated_dg_mesh.jl:183-189; level N: the union of the sub-agglomerates' boxes, :520-527)
        self.lo = (b.xc + b.h / 2.0 * (-1.0))[self.base_starts[:-1]]
        self.hi = (b.xc + b.h / 2.0 * (1.0))[self.base_starts[1:] - 1]
        self.gq, self.gw = gauss_quad(2 * self.mP)                # mGaussQuadNodes / Weights
        self.elem_of_base = np.repeat(np.arange(self.n, dtype=np.int64), np.diff(self.base_starts))

    @property
    def mBlockInds(self):
        return _contiguous_block_inds(self.m, self.n)

    def basis_at_base(self, xi):
        """modal basis of the agglomerate of every base element at its mRefMap(xi) -> (n_base, len(xi), p+1)
        (mBasisGQFunVal for xi = Gauss points)"""
        e = self.elem_of_base
        return evaluate_local_modal_basis_fun(self.mP, self.lo[e][:, None], self.hi[e][:, None], self.base.ref_map(xi))

    def mass_blocks(self):
        """mMassMatrix blocks (src/agglomerated_dg_mesh.jl:449-472): sum over base elements and Gauss points"""
        phi = self.basis_at_base(self.gq)                                         # (nb, nq, m)
        per = np.einsum('k,l,kli,klj->kij', self.base.J, self.gw, phi, phi)
        return np.add.reduceat(per, self.base_starts[:-1], axis=0)


# ------------------------------------------------------------------------------------------------
# CSC writers
# ------------------------------------------------------------------------------------------------
def _csc(colptr, rowval, nzval, shape):
    A = sp.csc_matrix((nzval, rowval, colptr), shape=shape)
    A.has_sorted_indices = True
    return A


def _dg_rows_csc(V, row_starts, mc, shape):
    """Columns (E, j), j < mc, of group E hold ALL rows [row_starts[E], row_starts[E+1]) with the values V[row, j]
    (V: (nrows, mc)): the pattern of every transfer whose fine side is DG-numbered."""
    nrows = V.shape[0]
    lens = np.diff(row_starts)
    g = np.repeat(np.arange(lens.size, dtype=np.int64), lens)          # group of every row
    off = np.arange(nrows, dtype=np.int64) - row_starts[g]             # offset of the row in its group
    colptr = np.empty(lens.size * mc + 1, dtype=np.int64)
    colptr[:-1] = (mc * row_starts[:-1])[:, None].repeat(mc, 1).reshape(-1) + \
        (np.arange(mc, dtype=np.int64)[None, :] * lens[:, None]).reshape(-1)
    colptr[-1] = mc * nrows
    pos = (mc * row_starts[g] + off)[:, None] + np.arange(mc, dtype=np.int64)[None, :] * lens[g][:, None]
    rowval = np.empty(mc * nrows, dtype=np.int64)
    nzval = np.empty(mc * nrows)
    rowval[pos] = np.arange(nrows, dtype=np.int64)[:, None]
    nzval[pos] = V
    return _csc(colptr, rowval, nzval, shape)


def _cg_rows_csc(T, cg, base_starts, mc, vertex_weight=None, row_scale=None):
    """Columns (E, j) of group E = base elements [base_starts[E], base_starts[E+1]) of the CgMesh `cg`; T[k, i, j]:
    contribution of element k, local CG node i.  Rows of a column: the group's vertices (ascending; a vertex shared
    by two elements of the group sums their contributions, element k-1 first as `sparse` meets them), then the
    interior nodes of its elements.  vertex_weight[v]: factor on every vertex contribution (interpFlag 2);
    row_scale[r]: divisor of every row (lumped mass)."""
    n, q = cg.n, cg.q
    ne = base_starts.size - 1
    nb = np.diff(base_starts)
    clen = nb + 1 + nb * q                                             # rows per column
    col0 = np.concatenate([[0], np.cumsum(clen * mc)])[:-1]            # first entry of the group's columns
    colptr = np.empty(ne * mc + 1, dtype=np.int64)
    colptr[:-1] = (col0[:, None] + np.arange(mc, dtype=np.int64)[None, :] * clen[:, None]).reshape(-1)
    colptr[-1] = int((clen * mc).sum())
    nnz = int(colptr[-1])
    rowval = np.empty(nnz, dtype=np.int64)
    nzval = np.zeros(nnz)
    g = np.repeat(np.arange(ne, dtype=np.int64), nb)                   # group of every element
    k = np.arange(n, dtype=np.int64)
    loc = k - base_starts[g]                                           # element's position in its group
    jj = np.arange(mc, dtype=np.int64)[None, :]
    base = col0[g][:, None] + jj * clen[g][:, None]                    # (n, mc): start of column (g, j)
    Tl, Tr = T[:, 0, :], T[:, 1, :]
    if vertex_weight is not None:
        Tl = Tl * vertex_weight[k][:, None]
        Tr = Tr * vertex_weight[k + 1][:, None]
    # vertex rows: element k writes its left vertex at slot loc, its right vertex at slot loc + 1
    pl, pr = base + loc[:, None], base + loc[:, None] + 1
    rowval[pl] = k[:, None]
    rowval[pr] = (k + 1)[:, None]
    np.add.at(nzval, pr.reshape(-1), Tr.reshape(-1))                   # element k-1 reaches a shared vertex first
    np.add.at(nzval, pl.reshape(-1), Tl.reshape(-1))
    if q:
        ii = np.arange(q, dtype=np.int64)
        pi = (base + (nb[g] + 1)[:, None] + (loc * q)[:, None])[:, None, :] + ii[None, :, None]   # (n, q, mc)
        rowval[pi] = ((n + 1) + k[:, None] * q + ii[None, :])[:, :, None]
        nzval[pi] = T[:, 2:, :]
    if row_scale is not None:
        nzval = nzval / row_scale[rowval]
    return _csc(colptr, rowval, nzval, (cg.mNumNodes, ne * mc))


def _consistent_mass_solve(cg, N):
    """`cg.mMassMatrixLU \\ Array(N)` (src/interpolation.jl:205,395): sparse LU of the consistent CG mass matrix (SuperLU
    here, UMFPACK in the reference: equal to round-off), dense right-hand side, dense result"""
    import scipy.sparse.linalg as spla
    if cg.mNumNodes * N.shape[1] > (1 << 27):
        raise UnsupportedError("interpFlag = 0 yields a dense matrix of %d x %d entries: meant for small meshes"
                               % (cg.mNumNodes, N.shape[1]))
    return spla.splu(cg.mass_matrix()).solve(N.toarray())


def _nodal_basis_at(ref, x):
    """evaluate_nodal_basis_fun (src/reference_element.jl:75-90): one dot product per (point, function) -> (len(x), p+1)"""
    V = legendre_vandermonde(x, ref.p)
    return np.array([[np.dot(ref.coeff[:, j], V[i]) for j in range(ref.p + 1)] for i in range(len(x))])


def _check_same_faces(a, b):
    if a.n != b.n or not np.array_equal(a.xv, b.xv):
        raise ArgumentError("the two meshes must be built on the same faces")


# ------------------------------------------------------------------------------------------------
# the builders
# ------------------------------------------------------------------------------------------------
def cg_cg_interpolation(lowMesh, highMesh):
    """src/interpolation.jl:5-55: nodal injection CG(p_low) -> CG(p_high) on the same faces.  Pattern: interior fine
    nodes x all coarse nodes of the element, plus the (fine vertex, coarse vertex) pairs; the vertex entry is written
    by every element that touches the vertex, the last one wins (:47-52)."""
    _check_same_faces(lowMesh, highMesh)
    n = lowMesh.n
    lowVal = _nodal_basis_at(lowMesh.ref, highMesh.ref.nodes)        # (p_hi+1, p_lo+1)
    qf, qc = highMesh.q, lowMesh.q
    Nf, Nc = highMesh.mNumNodes, lowMesh.mNumNodes
    v = np.arange(n + 1, dtype=np.int64)
    e = np.arange(n, dtype=np.int64)
    jf = np.arange(qf, dtype=np.int64)
    # coarse vertex column v: fine vertex v, fine interior of element v-1 (coarse local node 1), of element v (node 0)
    has_l, has_r = (v >= 1), (v <= n - 1)
    cnt_v = 1 + qf * (has_l.astype(np.int64) + has_r.astype(np.int64))
    cnt_i = np.full(n * qc, qf, dtype=np.int64)                       # coarse interior column (e, j): fine interior of e
    colptr = np.concatenate([[0], np.cumsum(np.concatenate([cnt_v, cnt_i]))]).astype(np.int64)
    rowval = np.empty(int(colptr[-1]), dtype=np.int64)
    nzval = np.empty(int(colptr[-1]))
    p0 = colptr[:n + 1]
    rowval[p0] = v
    vv = np.full(n + 1, lowVal[0, 0])
    vv[n] = lowVal[1, 1]                                              # the last element writes vertex n as its local node 1
    nzval[p0] = vv
    if qf:
        # interior of element v-1 (rows (n+1) + (v-1) qf + i) comes before the interior of element v
        pl = (p0[1:] + 1)[:, None] + jf[None, :]                      # vertices 1..n
        rowval[pl] = (n + 1) + e[:, None] * qf + jf[None, :]
        nzval[pl] = lowVal[2:, 1][None, :]
        pr = (p0[:-1] + 1 + qf * has_l[:-1])[:, None] + jf[None, :]   # vertices 0..n-1
        rowval[pr] = (n + 1) + e[:, None] * qf + jf[None, :]
        nzval[pr] = lowVal[2:, 0][None, :]
        if qc:
            pc = colptr[n + 1:-1].reshape(n, qc)[:, :, None] + jf[None, None, :]        # (n, qc, qf)
            rowval[pc] = ((n + 1) + e[:, None] * qf + jf[None, :])[:, None, :]
            nzval[pc] = lowVal[2:, 2:].T[None, :, :]
    return _csc(colptr, rowval, nzval, (Nf, Nc))


def dg_dg_interpolation(lowMesh, highMesh):
    """src/interpolation.jl:91-109: p-coarsening DG(p_low) -> DG(p_high), one dense block per element"""
    _check_same_faces(lowMesh, highMesh)
    lowVal = _nodal_basis_at(lowMesh.ref, highMesh.ref.nodes)        # (m_hi, m_lo)
    V = np.broadcast_to(lowVal, (lowMesh.n,) + lowVal.shape).reshape(-1, lowMesh.m)
    starts = np.arange(lowMesh.n + 1, dtype=np.int64) * highMesh.m
    return _dg_rows_csc(V, starts, lowMesh.m, (highMesh.mNumNodes, lowMesh.mNumNodes))


def dg_dg_interpolation2(lowMesh, highMesh):
    """src/interpolation.jl:111-139: the interior nodes of every fine element take all coarse nodes of the element
    (as dg_dg_interpolation), its two end nodes only the coarse end node of the same side.  Needs p >= 1 on both meshes
    (the reference indexes mNodesInd[1:2] of both elements)."""
    _check_same_faces(lowMesh, highMesh)
    if lowMesh.mP < 1 or highMesh.mP < 1:
        raise ArgumentError("dg_dg_interpolation2 needs p >= 1 on both meshes")
    n, mh, ml = lowMesh.n, highMesh.m, lowMesh.m
    lowVal = _nodal_basis_at(lowMesh.ref, highMesh.ref.nodes)        # (m_hi, m_lo)
    qh = mh - 2
    # column (k, j): [end node j of the fine element if j < 2] then its interior nodes, ascending
    cnt = np.where(np.arange(ml) < 2, 1 + qh, qh).astype(np.int64)
    colptr = np.concatenate([[0], np.cumsum(np.tile(cnt, n))]).astype(np.int64)
    rowval = np.empty(int(colptr[-1]), dtype=np.int64)
    nzval = np.empty(int(colptr[-1]))
    k = np.arange(n, dtype=np.int64)
    cp = colptr[:-1].reshape(n, ml)
    ii = np.arange(qh, dtype=np.int64)
    for j in range(ml):
        off = 0
        if j < 2:
            rowval[cp[:, j]] = k * mh + j
            nzval[cp[:, j]] = lowVal[j, j]
            off = 1
        if qh:
            pos = (cp[:, j] + off)[:, None] + ii[None, :]
            rowval[pos] = (k * mh + 2)[:, None] + ii[None, :]
            nzval[pos] = lowVal[2:, j][None, :]
    return _csc(colptr, rowval, nzval, (highMesh.mNumNodes, lowMesh.mNumNodes))


def cg_cg_interpolation2(lowMesh, highMesh):
    """src/interpolation.jl:57-85: the consistent-mass L2 projection CG(p_low) -> CG(p_high),
    `highMesh.mMassMatrixLU \\ Array(N)` -- a DENSE ndarray, as in the reference (small meshes)"""
    import scipy.sparse as sp
    _check_same_faces(lowMesh, highMesh)
    T = lowMesh.J[:, None, None] * _ref_l2_block(highMesh.ref, lowMesh.ref)[None, :, :]      # (n, m_hi, m_lo)
    eh, el = highMesh.element_nodes() - 1, lowMesh.element_nodes() - 1
    rows = np.broadcast_to(eh.T[:, :, None], T.shape)
    cols = np.broadcast_to(el.T[:, None, :], T.shape)
    N = sp.csc_matrix((T.reshape(-1), (rows.reshape(-1), cols.reshape(-1))), shape=(highMesh.mNumNodes, lowMesh.mNumNodes))
    return _consistent_mass_solve(highMesh, N)


def _ref_l2_block(ref_rows, ref_cols):
    """sum_l w_l phi_rows[l, i] phi_cols[l, j] with the Gauss rule of degree p_rows + p_cols (src/interpolation.jl:150-168)"""
    gq, gw = gauss_quad(ref_rows.p + ref_cols.p)
    hi, lo = _nodal_basis_at(ref_rows, gq), _nodal_basis_at(ref_cols, gq)
    return np.einsum('l,li,lj->ij', gw, hi, lo)


def dg_cg_interpolation(lowMesh, highMesh, interpFlag):
    """src/interpolation.jl:145-220: DG (coarser, `lowMesh`) -> CG (finer, `highMesh`) on the same faces.
    interpFlag 1: lumped-mass L2 projection; 2: nodal evaluation, interior vertices averaged; 0: the consistent-mass
    L2 projection `mMassMatrixLU \\ Array(N)` (:205) -- a DENSE (mNumNodes x lowMesh.mNumNodes) ndarray, as in the
    reference: O(n^2) storage, small meshes only (the hierarchy constructors hard-wire flag 1)."""
    _check_same_faces(lowMesh, highMesh)
    n = lowMesh.n
    starts = np.arange(n + 1, dtype=np.int64)
    if interpFlag in (0, 1):
        T = lowMesh.J[:, None, None] * _ref_l2_block(highMesh.ref, lowMesh.ref)[None, :, :]
        if interpFlag == 0:
            return _consistent_mass_solve(highMesh, _cg_rows_csc(T, highMesh, starts, lowMesh.m))
        return _cg_rows_csc(T, highMesh, starts, lowMesh.m, row_scale=highMesh.lumped_mass())
    if interpFlag == 2:
        lowVal = _nodal_basis_at(lowMesh.ref, highMesh.ref.nodes)
        T = np.broadcast_to(lowVal, (n,) + lowVal.shape)
        w = np.full(n + 1, 0.5)
        w[0] = w[n] = 1.0                                             # isBoundary(vertex)
        return _cg_rows_csc(T, highMesh, starts, lowMesh.m, vertex_weight=w)
    raise ValueError("Only implemented for interpFlag = 0, 1, or 2.")


def aggdg_dg_interpolation(aggMesh, baseMesh):
    """src/interpolation.jl:270-292: the agglomerate's modal basis evaluated at the base DG nodes"""
    if aggMesh.base is not baseMesh or aggMesh.sub is not baseMesh:
        raise ArgumentError("aggdg_dg_interpolation: aggMesh must agglomerate baseMesh directly")
    V = aggMesh.basis_at_base(baseMesh.ref.nodes).reshape(-1, aggMesh.m)
    return _dg_rows_csc(V, aggMesh.base_starts * baseMesh.m, aggMesh.m, (baseMesh.mNumNodes, aggMesh.mNumNodes))


def _batched_solve(M, X):
    m = M.shape[-1]
    if m == 1:
        return X / M
    return np.linalg.solve(M, X)


def aggdg_dg_interpolation2(aggMesh, baseMesh):
    """src/interpolation.jl:294-324: L2 projection, `M_base \\ N` block by block"""
    if aggMesh.base is not baseMesh or aggMesh.sub is not baseMesh:
        raise ArgumentError("aggdg_dg_interpolation2: aggMesh must agglomerate baseMesh directly")
    ref = baseMesh.ref
    aggV = aggMesh.basis_at_base(ref.gq)                              # (nb, nq, mc)
    N = np.einsum('k,l,li,klj->kij', baseMesh.J, ref.gw, ref.phi, aggV)
    Mb = baseMesh.J[:, None, None] * ref.mass[None, :, :]
    V = _batched_solve(Mb, N).reshape(-1, aggMesh.m)
    return _dg_rows_csc(V, aggMesh.base_starts * baseMesh.m, aggMesh.m, (baseMesh.mNumNodes, aggMesh.mNumNodes))


def aggdg_aggdg_interpolation(coarseMesh, fineMesh, baseMesh=None):
    """src/interpolation.jl:226-264: `M_fine \\ N`, N[i, j] = sum over the fine agglomerate's base elements and Gauss
    points of J w phi_fine_i phi_coarse_j; the result keeps all rows of every fine block a column touches"""
    if coarseMesh.mP != fineMesh.mP:
        raise ValueError("The two agglomerated meshes must have the same p.")
    if coarseMesh.sub is not fineMesh or (baseMesh is not None and baseMesh is not fineMesh.base):
        raise ArgumentError("aggdg_aggdg_interpolation: coarseMesh must agglomerate fineMesh (over the same base mesh)")
    b = fineMesh.base
    fphi = fineMesh.basis_at_base(fineMesh.gq)
    cphi = coarseMesh.basis_at_base(fineMesh.gq)
    per = np.einsum('k,l,kli,klj->kij', b.J, fineMesh.gw, fphi, cphi)
    N = np.add.reduceat(per, fineMesh.base_starts[:-1], axis=0)
    V = _batched_solve(fineMesh.mass_blocks(), N).reshape(-1, coarseMesh.m)
    return _dg_rows_csc(V, coarseMesh.sub_starts * fineMesh.m, coarseMesh.m, (fineMesh.mNumNodes, coarseMesh.mNumNodes))


def aggdg_cg_interpolation(aggMesh, baseMesh, interpFlag):
    """src/interpolation.jl:330-410: agglomerated DG -> CG base mesh.  `aggMesh` must agglomerate the DgMesh on the
    faces of `baseMesh` directly.  interpFlag as in dg_cg_interpolation."""
    if not isinstance(aggMesh.sub, DgMesh):
        raise ArgumentError("aggdg_cg_interpolation: aggMesh must be a first-level agglomeration")
    _check_same_faces(aggMesh.base, baseMesh)
    ref = baseMesh.ref
    if interpFlag in (0, 1):
        aggV = evaluate_local_modal_basis_fun(aggMesh.mP, aggMesh.lo[aggMesh.elem_of_base][:, None],
                                              aggMesh.hi[aggMesh.elem_of_base][:, None], baseMesh.ref_map(ref.gq))
        T = np.einsum('k,l,li,klj->kij', baseMesh.J, ref.gw, ref.phi, aggV)
        if interpFlag == 0:       # `mMassMatrixLU \\ Array(N)` (:395): dense, as in the reference
            return _consistent_mass_solve(baseMesh, _cg_rows_csc(T, baseMesh, aggMesh.base_starts, aggMesh.m))
        return _cg_rows_csc(T, baseMesh, aggMesh.base_starts, aggMesh.m, row_scale=baseMesh.lumped_mass())
    if interpFlag == 2:
        T = evaluate_local_modal_basis_fun(aggMesh.mP, aggMesh.lo[aggMesh.elem_of_base][:, None],
                                           aggMesh.hi[aggMesh.elem_of_base][:, None], baseMesh.ref_map(ref.nodes))
        w = np.full(baseMesh.n + 1, 0.5)
        w[0] = w[baseMesh.n] = 1.0
        return _cg_rows_csc(T, baseMesh, aggMesh.base_starts, aggMesh.m, vertex_weight=w)
    raise ValueError("Only implemented for interpFlag = 0, 1, or 2.")
