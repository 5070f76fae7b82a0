"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU restatement (NumPy/SciPy, loop-for-loop) of mheinz757/AgglomerationMultigrid1D, the
Julia reference mounted read-only at /root/reference.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module.  The product package
(agglomerationmultigrid1d_amd) never does.

PARITY STATUS: "restatement-derived, reference not executed".  Julia is not installed in the
build container (plain unavailability: `julia: command not found`), the reference commits no
golden vectors, and its tests assert nothing.  This restatement is
therefore pinned by the identities and asymptotics the reference's own test scripts print:
block-diagonal vs dense equivalence, Galerkin identities, polynomial reproduction, p+1
discretisation order, smoother convergence, V-cycle convergence (tests/test_oracle_pins.py).
Numerical parity with an *executed* reference is "parity unpinned".

Conventions kept from Julia:
  * all node / element / vertex indices stored in the objects are 1-based;
  * sparse matrices are scipy.sparse.csc_matrix (0-based inside SciPy), `julia_csc(A)` gives
    the 1-based Int64 (colptr, rowval, nzval) triple the Julia `SparseMatrixCSC` would hold;
  * `sparse(I, J, V, m, n)` sums duplicates and keeps explicit zeros (SparseArrays.sparse);
  * `A - B` on sparse matrices drops numerically-zero results (SparseArrays zero-preserving
    map), products keep structural zeros.

Every function cites the reference file:line it restates.
"""
import math

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp
import scipy.sparse.linalg as spla

EPS = np.finfo(np.float64).eps

# --------------------------------------------------------------------------------------
# SparseArrays semantics
# --------------------------------------------------------------------------------------


def sparse(I, J, V, m, n):
    """SparseArrays.sparse(I,J,V,m,n): 1-based COO -> CSC, duplicates summed, explicit zeros
    kept, rows sorted within each column."""
    I = np.asarray(I, dtype=np.int64) - 1
    J = np.asarray(J, dtype=np.int64) - 1
    V = np.asarray(V, dtype=np.float64)
    A = sp.coo_matrix((V, (I, J)), shape=(m, n)).tocsc()
    A.sum_duplicates()
    A.sort_indices()
    return A


def sp_sub(A, B):
    """`A - B` for SparseMatrixCSC: numerically-zero results are not stored
    (used at src/mesh_heirarchy.jl:71-72)."""
    C = (A - B).tocsc()
    C.eliminate_zeros()
    C.sort_indices()
    return C


def julia_csc(A):
    """1-based Int64 (colptr, rowval, nzval) of a SciPy CSC matrix."""
    A = A.tocsc()
    A.sort_indices()
    return (A.indptr.astype(np.int64) + 1, A.indices.astype(np.int64) + 1,
            A.data.astype(np.float64).copy())


def csc_matvec(A, x):
    """SparseArrays `A*x` for CSC: column scatter `y[rowval[p]] += nzval[p]*x[j]`, each y[i]
    accumulated in ascending column order.  SciPy's csc_matvec is the
    same loop."""
    return A.tocsc() @ np.asarray(x, dtype=np.float64)


def csc_adjoint_matvec(A, x):
    """`A' * x` for CSC A: one dot product per column of A in ascending row order
    (src/solvers.jl:36).  SciPy: A.T of a CSC is CSR -> csr_matvec, the same loop."""
    return A.tocsc().T @ np.asarray(x, dtype=np.float64)


# --------------------------------------------------------------------------------------
# src/meshes.jl, src/boundary_conditions.jl, tests/mesh_generator.jl
# --------------------------------------------------------------------------------------


class Vertex:
    """src/meshes.jl:11-17"""

    def __init__(self, mIndex, mX):
        self.mIndex = mIndex
        self.mX = mX
        self.mFaces = [0, 0]


class Face:
    """src/meshes.jl:29-37"""

    def __init__(self, mIndex, nV):
        self.mIndex = mIndex
        self.mVertices = [None] * nV
        self.mNeighbors = [0] * nV


class Mesh:
    """src/meshes.jl:48-51"""

    def __init__(self, vertices, faces):
        self.mVertices = vertices
        self.mFaces = faces


def isBoundary(obj):
    """src/meshes.jl:58-69 (Vertex / Face) and src/agglomerated_dg_mesh.jl:78-80."""
    if isinstance(obj, Face):
        return obj.mNeighbors[-1] == 0
    return obj.mFaces[1] < 1


class BoundaryCondition:
    """src/boundary_conditions.jl:1-6"""

    def __init__(self, mBdCond, mDirNodes, mDirVals, mNeuNodes):
        self.mBdCond = mBdCond
        self.mDirNodes = mDirNodes
        self.mDirVals = mDirVals
        self.mNeuNodes = mNeuNodes


def create_uniform_mesh(n, xin, xout):
    """tests/mesh_generator.jl:5-59"""
    faces = [None] * n
    vertices = [None] * (n + 1)
    vertices[0] = Vertex(1, xin)
    for i in range(1, n + 1):
        vertices[i] = Vertex(i + 1, xin + (i / n) * (xout - xin))
        faces[i - 1] = Face(i, 2)
        for j in range(2):
            faces[i - 1].mVertices[j] = vertices[i - 1 + j]
    for cFace in faces:
        for cVertex in cFace.mVertices:
            if cVertex.mFaces[0] == 0:
                cVertex.mFaces[0] = cFace.mIndex
            elif cVertex.mFaces[1] == 0:
                cVertex.mFaces[1] = cFace.mIndex
            else:
                raise RuntimeError("Vertex can only neighbor two faces.")
    for cFace in faces:
        adj = {}
        for fVert in cFace.mVertices:
            for f in fVert.mFaces:
                adj[f] = adj.get(f, 0) + 1
        nIndex = 0
        for f in sorted(adj):  # Dict order is unspecified in Julia; only the count matters
            if f != 0 and f != cFace.mIndex and adj[f] == 1:
                cFace.mNeighbors[nIndex] = f
                nIndex += 1
    return Mesh(vertices, faces)


def set_boundary(mesh, xin, xout, bdCond):
    """tests/mesh_generator.jl:61-93 (`set_boundary!`)"""
    dirNodes, dirVals, neuNodes = [], [], []
    for face in mesh.mFaces:
        if isBoundary(face):
            for vert in face.mVertices:
                if isBoundary(vert) and abs(vert.mX - xin) < 1e-15:
                    vert.mFaces[1] = -1
                    if bdCond[0][0] == 'dir':
                        dirNodes.append(vert.mIndex)
                        dirVals.append(bdCond[0][1])
                    elif bdCond[0][0] == 'neu':
                        neuNodes.append(vert.mIndex)
                elif isBoundary(vert) and abs(vert.mX - xout) < 1e-15:
                    vert.mFaces[1] = -2
                    if bdCond[1][0] == 'dir':
                        dirNodes.append(vert.mIndex)
                        dirVals.append(bdCond[1][1])
                    elif bdCond[1][0] == 'neu':
                        neuNodes.append(vert.mIndex)
    return BoundaryCondition(bdCond, dirNodes, dirVals, neuNodes)


# --------------------------------------------------------------------------------------
# src/legendre.jl, src/gauss_quad.jl, src/reference_element.jl
# --------------------------------------------------------------------------------------


def legendre_val(x, n):
    """src/legendre.jl:14-25"""
    if n == 0:
        return np.array([1.0])
    f = [1.0, x]
    for i in range(2, n + 1):
        f.append(((2 * i - 1) * x * f[i - 1] - (i - 1) * f[i - 2]) / i)
    return np.array(f)


def legendre_val_and_deriv(x, n):
    """src/legendre.jl:44-58"""
    if n == 0:
        return np.array([1.0]), np.array([0.0])
    f = [1.0, x]
    d = [0.0, 1.0]
    for i in range(2, n + 1):
        f.append(((2 * i - 1) * x * f[i - 1] - (i - 1) * f[i - 2]) / i)
        d.append((2 * i - 1) * f[i - 1] + d[i - 2])
    return np.array(f), np.array(d)


def gauss_quad(p):
    """src/gauss_quad.jl:6-13 (Golub-Welsch; nodes ascending, weights 2*v1^2)."""
    n = int(math.ceil((p + 1) / 2))
    b = np.arange(1, n, dtype=np.float64)
    b = b / np.sqrt(4 * b**2 - 1)
    Jm = np.diag(b, 1) + np.diag(b, -1)
    ev, evec = np.linalg.eigh(Jm)
    return ev, 2 * evec[0, :]**2


def evaluate_nodal_basis_fun(basisFunCoeff, nodes):
    """src/reference_element.jl:60-73"""
    nodes = np.atleast_1d(nodes)
    p = basisFunCoeff.shape[0] - 1
    val = np.zeros((len(nodes), p + 1))
    for l, x in enumerate(nodes):
        lv = legendre_val(x, p)
        for i in range(p + 1):
            val[l, i] = np.dot(basisFunCoeff[:, i], lv)
    return val


def evaluate_nodal_basis_fun_and_deriv(basisFunCoeff, nodes):
    """src/reference_element.jl:75-90"""
    nodes = np.atleast_1d(nodes)
    p = basisFunCoeff.shape[0] - 1
    val = np.zeros((len(nodes), p + 1))
    der = np.zeros((len(nodes), p + 1))
    for l, x in enumerate(nodes):
        lv, ld = legendre_val_and_deriv(x, p)
        for i in range(p + 1):
            val[l, i] = np.dot(basisFunCoeff[:, i], lv)
            der[l, i] = np.dot(basisFunCoeff[:, i], ld)
    return val, der


class ReferenceElement:
    """src/reference_element.jl:1-57.  Local node order: endpoints [-1, 1] first, then
    cos(pi*i/p), i=1..p-1 (descending x)."""

    def __init__(self, mP):
        self.mP = mP
        x = np.zeros(mP + 1)
        if mP >= 1:
            x[0:2] = [-1.0, 1.0]
            x[2:mP + 1] = np.cos(np.pi * np.arange(1, mP) / mP)
        else:
            x[0] = 0.0
        self.mNodesX = x
        V = np.zeros((mP + 1, mP + 1))
        for i in range(mP + 1):
            V[i, :] = legendre_val(x[i], mP)
        self.mBasisFunCoeff = np.linalg.inv(V)
        self.mGaussQuadNodes, self.mGaussQuadWeights = gauss_quad(2 * mP)
        self.mBasisGQFunVal, self.mBasisGQDerivVal = evaluate_nodal_basis_fun_and_deriv(
            self.mBasisFunCoeff, self.mGaussQuadNodes)
        M = np.zeros((mP + 1, mP + 1))
        for j in range(mP + 1):
            for i in range(j + 1):
                for l in range(len(self.mGaussQuadWeights)):
                    M[i, j] += (self.mGaussQuadWeights[l] * self.mBasisGQFunVal[l, i] *
                                self.mBasisGQFunVal[l, j])
        for j in range(mP + 1):
            for i in range(j + 1, mP + 1):
                M[i, j] = M[j, i]
        self.mMassMatrix = M


# --------------------------------------------------------------------------------------
# src/block_diagonal.jl
# --------------------------------------------------------------------------------------


class LU:
    """LinearAlgebra.lu(::Matrix) = LAPACK getrf (partial pivoting); `\\` = getrs.
    A singular block raises (Julia: SingularException) -- src/smoother.jl:160."""

    def __init__(self, A):
        A = np.array(A, dtype=np.float64)
        self.lu, self.piv = sla.lu_factor(A, check_finite=False)
        if np.any(np.diag(self.lu) == 0.0):
            raise np.linalg.LinAlgError("SingularException")

    def solve(self, b):
        return sla.lu_solve((self.lu, self.piv), b, check_finite=False)


class BlockDiagonal:
    """src/block_diagonal.jl:11-15, ctor :27-41.  mBlockInds is (mBlockSize x nblocks),
    1-based."""

    def __init__(self, mBlocks, mBlockSize=None, mBlockInds=None):
        if mBlockSize is None:
            mBlockSize = mBlocks[0].shape[0]
            mBlockInds = np.zeros((mBlockSize, len(mBlocks)), dtype=np.int64)
            for i, block in enumerate(mBlocks):
                mBlockInds[:, i] = np.arange(i * mBlockSize + 1, (i + 1) * mBlockSize + 1)
                if block.shape != (mBlockSize, mBlockSize):
                    raise ValueError("All blocks must be of the same size.")
        self.mBlocks = mBlocks
        self.mBlockSize = mBlockSize
        self.mBlockInds = np.asarray(mBlockInds, dtype=np.int64)

    @property
    def shape(self):
        n = len(self.mBlocks) * self.mBlockSize
        return (n, n)

    def todense(self):
        """Matrix(A) src/block_diagonal.jl:107-114"""
        B = np.zeros(self.shape)
        for i, block in enumerate(self.mBlocks):
            idx = self.mBlockInds[:, i] - 1
            B[np.ix_(idx, idx)] = block
        return B

    def tosparse(self):
        """sparse(A) src/block_diagonal.jl:118-129"""
        I, J, V = [], [], []
        for k, block in enumerate(self.mBlocks):
            inds = self.mBlockInds[:, k]
            for j, node2 in enumerate(inds):
                for i, node1 in enumerate(inds):
                    I.append(node1), J.append(node2), V.append(block[i, j])
        return sparse(I, J, V, *self.shape)

    def mul_dense(self, B):
        """mul!(C, A::BlockDiagonal, B) src/block_diagonal.jl:166-176 (C starts at zero)."""
        B = np.asarray(B, dtype=np.float64)
        if self.shape[1] != B.shape[0]:
            raise ValueError("DimensionMismatch")
        C = np.zeros(B.shape)
        for i, block in enumerate(self.mBlocks):
            idx = self.mBlockInds[:, i] - 1
            C[idx] += block @ B[idx]
        return C

    def mul_sparse(self, B):
        """bd_sp_matmul src/block_diagonal.jl:195-215"""
        return _bd_sp_apply(self, B, lambda k, v: self.mBlocks[k] @ v)

    def lu(self):
        return BlockDiagonalLU(self)


class BlockDiagonalLU:
    """src/block_diagonal.jl:17-21, ctor :47-58."""

    def __init__(self, A):
        self.mBlockSize = A.mBlockSize
        self.mBlockInds = A.mBlockInds
        self.mBlocksLU = [LU(b) for b in A.mBlocks]

    @property
    def shape(self):
        n = len(self.mBlocksLU) * self.mBlockSize
        return (n, n)

    def solve_dense(self, B):
        """ldiv!(C, A::BlockDiagonalLU, B) src/block_diagonal.jl:299-309"""
        B = np.asarray(B, dtype=np.float64)
        if self.shape[1] != B.shape[0]:
            raise ValueError("DimensionMismatch")
        C = np.zeros(B.shape)
        for i, blk in enumerate(self.mBlocksLU):
            idx = self.mBlockInds[:, i] - 1
            C[idx] += blk.solve(B[idx])
        return C

    def solve_sparse(self, B):
        """bd_sp_solve src/block_diagonal.jl:314-334"""
        return _bd_sp_apply(self, B, lambda k, v: self.mBlocksLU[k].solve(v))


def _bd_sp_apply(A, B, blockop):
    """bd_sp_colmul / bd_sp_colsolve, src/block_diagonal.jl:217-264 and :336-383.
    Per column of the sparse B: consecutive stored rows are grouped by block
    (`blockInd = div(row-1, bs)+1`, i.e. contiguous aligned blocks are assumed), the block is
    applied to the gathered dense sub-vector and ALL `bs` rows of the block are emitted
    (zeros included)."""
    B = B.tocsc()
    B.sort_indices()
    if A.shape[1] != B.shape[0]:
        raise ValueError("DimensionMismatch")
    bs = A.mBlockSize
    colptr = [0]
    rowval, nzval = [], []
    for col in range(B.shape[1]):
        lo, hi = B.indptr[col], B.indptr[col + 1]
        rows = B.indices[lo:hi] + 1  # 1-based
        vals = B.data[lo:hi]
        ind = 0
        cnt = 0
        while ind < len(vals):
            row = rows[ind]
            blockInd = (row - 1) // bs + 1
            minBlockRow = A.mBlockInds[0, blockInd - 1]
            maxBlockRow = A.mBlockInds[-1, blockInd - 1]
            tempVec = np.zeros(bs)
            tempVec[row - minBlockRow] = vals[ind]
            i = 1
            tempRow = rows[ind + i] if ind + i < len(vals) else row + bs
            while tempRow <= maxBlockRow:
                tempVec[tempRow - minBlockRow] = vals[ind + i]
                i += 1
                tempRow = rows[ind + i] if ind + i < len(vals) else row + bs
            ind += i
            b = blockop(blockInd - 1, tempVec)
            for ii in range(bs):
                rowval.append(minBlockRow + ii - 1)
                nzval.append(b[ii])
                cnt += 1
        colptr.append(colptr[-1] + cnt)
    return sp.csc_matrix((np.array(nzval, dtype=np.float64), np.array(rowval, dtype=np.int64),
                          np.array(colptr, dtype=np.int64)), shape=(A.shape[0], B.shape[1]))


# --------------------------------------------------------------------------------------
# src/cg_mesh.jl
# --------------------------------------------------------------------------------------


class CgElement:
    """src/cg_mesh.jl:1-10, ctor :26-48.  Vertex nodes first (global vertex index), then
    p-1 interior nodes numbered from vertCounter."""

    def __init__(self, face, mP, vertCounter, refEl):
        self.mIndex = face.mIndex
        self.mP = mP
        h = face.mVertices[1].mX - face.mVertices[0].mX
        xc = (face.mVertices[0].mX + face.mVertices[1].mX) / 2.0
        self.mJacobian = h / 2.0
        self._xc, self._h = xc, h
        self.mNodesInd = [0] * (mP + 1)
        self.mNodesX = [0.0] * (mP + 1)
        for i in range(2):
            self.mNodesInd[i] = face.mVertices[i].mIndex
            self.mNodesX[i] = face.mVertices[i].mX
        for i in range(2, mP + 1):
            self.mNodesInd[i] = vertCounter
            self.mNodesX[i] = self.mRefMap(refEl.mNodesX[i])
            vertCounter += 1

    def mRefMap(self, xi):
        return self._xc + self._h / 2.0 * xi


class CgMesh:
    """src/cg_mesh.jl:12-20, ctor :54-79 (the UMFPACK LU of the mass matrix is replaced by
    SciPy's SuperLU; it is only used by the dense interpFlag==0 transfers)."""

    def __init__(self, mesh, mP):
        if mP < 1:
            raise ValueError("CgMesh needs p >= 1 (src/cg_mesh.jl:35-40 writes mNodesInd[2])")
        self.mP = mP
        self.mRefEl = ReferenceElement(mP)
        self.mElements = []
        vertCounter = len(mesh.mVertices) + 1
        for face in mesh.mFaces:
            self.mElements.append(CgElement(face, mP, vertCounter, self.mRefEl))
            vertCounter += mP - 1
        self.mNumNodes = vertCounter - 1
        I, J, V = [], [], []
        for el in self.mElements:
            for j, node2 in enumerate(el.mNodesInd):
                for i, node1 in enumerate(el.mNodesInd):
                    I.append(node1), J.append(node2)
                    V.append(el.mJacobian * self.mRefEl.mMassMatrix[i, j])
        self.mMassMatrix = sparse(I, J, V, self.mNumNodes, self.mNumNodes)
        self._massLU = None

    @property
    def mMassMatrixLU(self):
        if self._massLU is None:
            self._massLU = spla.splu(self.mMassMatrix.tocsc())
        return self._massLU


def _cg_local_stiffness(el, refEl):
    n = len(el.mNodesInd)
    temp = np.zeros((n, n))
    for j in range(n):
        for i in range(n):
            for l in range(len(refEl.mGaussQuadNodes)):
                temp[i, j] += ((1.0 / el.mJacobian) * refEl.mGaussQuadWeights[l] *
                               refEl.mBasisGQDerivVal[l, i] * refEl.mBasisGQDerivVal[l, j])
    return temp


def _strong_dirichlet(A, dirNodes):
    """A[dir,:]=spzeros; A[:,dir]=spzeros; A[dir,dir]=I (src/cg_mesh.jl:117-119, :177-182).
    Sparse-to-sparse setindex! replaces the addressed region's structure, so the stored entries
    of the Dirichlet rows/columns are removed; every other stored entry (zeros included) stays."""
    A = A.tocoo()
    d = np.array([k - 1 for k in dirNodes], dtype=np.int64)
    keep = ~(np.isin(A.row, d) | np.isin(A.col, d))
    rows = np.concatenate([A.row[keep], d])
    cols = np.concatenate([A.col[keep], d])
    vals = np.concatenate([A.data[keep], np.ones(len(d))])
    return sparse(rows + 1, cols + 1, vals, *A.shape)


def cg_stiffness_and_rhs(cgMesh, mesh, func, bdCond):
    """src/cg_mesh.jl:125-185"""
    refEl = cgMesh.mRefEl
    I, J, V = [], [], []
    f = np.zeros(cgMesh.mNumNodes)
    for el in cgMesh.mElements:
        temp = _cg_local_stiffness(el, refEl)
        for j, node2 in enumerate(el.mNodesInd):
            for i, node1 in enumerate(el.mNodesInd):
                I.append(node1), J.append(node2), V.append(temp[i, j])
        for i, node in enumerate(el.mNodesInd):
            for l in range(len(refEl.mGaussQuadNodes)):
                f[node - 1] += (el.mJacobian * refEl.mGaussQuadWeights[l] *
                                refEl.mBasisGQFunVal[l, i] *
                                func(el.mRefMap(refEl.mGaussQuadNodes[l])))
    A = sparse(I, J, V, cgMesh.mNumNodes, cgMesh.mNumNodes)
    for k in bdCond.mNeuNodes:
        vert = mesh.mVertices[k - 1]
        face = mesh.mFaces[vert.mFaces[0] - 1]
        if vert is face.mVertices[0]:
            f[vert.mIndex - 1] += -bdCond.mBdCond[-vert.mFaces[1] - 1][1]
        else:
            f[vert.mIndex - 1] += bdCond.mBdCond[-vert.mFaces[1] - 1][1]
    d = [k - 1 for k in bdCond.mDirNodes]
    if d:
        f = f + (-(A[:, d] @ np.array(bdCond.mDirVals, dtype=np.float64)))
        f[d] = bdCond.mDirVals
    A = _strong_dirichlet(A, bdCond.mDirNodes)
    return A, f


def cg_stiffness(cgMesh, bdCond):
    """src/cg_mesh.jl:87-122"""
    refEl = cgMesh.mRefEl
    I, J, V = [], [], []
    for el in cgMesh.mElements:
        temp = _cg_local_stiffness(el, refEl)
        for j, node2 in enumerate(el.mNodesInd):
            for i, node1 in enumerate(el.mNodesInd):
                I.append(node1), J.append(node2), V.append(temp[i, j])
    A = sparse(I, J, V, cgMesh.mNumNodes, cgMesh.mNumNodes)
    return _strong_dirichlet(A, bdCond.mDirNodes)


# --------------------------------------------------------------------------------------
# src/dg_mesh.jl
# --------------------------------------------------------------------------------------


class DgElement:
    """src/dg_mesh.jl:1-14, ctor :32-52.  Nodes (k-1)(p+1)+1 .. k(p+1), contiguous."""

    def __init__(self, face, mP, refEl):
        self.mIndex = face.mIndex
        self.mP = mP
        h = face.mVertices[1].mX - face.mVertices[0].mX
        xc = (face.mVertices[0].mX + face.mVertices[1].mX) / 2.0
        self.mJacobian = h / 2.0
        self._xc, self._h = xc, h
        self.mNodesInd = [(self.mIndex - 1) * (mP + 1) + i for i in range(1, mP + 2)]
        self.mNodesX = [self.mRefMap(refEl.mNodesX[i]) for i in range(mP + 1)]

    def mRefMap(self, xi):
        return self._xc + self._h / 2.0 * xi


def _switch_vector(vertices, face_of, xs_of):
    """mSwitch, src/dg_mesh.jl:83-108 and src/agglomerated_dg_mesh.jl:364-389.  The interior
    branch reads `vert.mFaces[1]` twice (reference bug), so x1 == x2 and the switch
    is 1 on every interior vertex.  Restated as is."""
    sw = []
    for vert in vertices:
        if isBoundary(vert):
            xs = xs_of(face_of(vert.mFaces[0]))
            sw.append(1 if vert.mX > min(xs) else 2)
        else:
            x1 = max(xs_of(face_of(vert.mFaces[0])))
            x2 = max(xs_of(face_of(vert.mFaces[0])))
            sw.append(2 if x1 > x2 else 1)
    return sw


class DgMesh:
    """src/dg_mesh.jl:16-26, ctor :58-111."""

    def __init__(self, mesh, mP):
        self.mP = mP
        self.mRefEl = ReferenceElement(mP)
        self.mElements = [DgElement(face, mP, self.mRefEl) for face in mesh.mFaces]
        self.mNumNodes = len(self.mElements) * (mP + 1)
        blocks = [None] * len(self.mElements)
        inds = np.zeros((mP + 1, len(self.mElements)), dtype=np.int64)
        for el in self.mElements:
            blocks[el.mIndex - 1] = el.mJacobian * self.mRefEl.mMassMatrix
            inds[:, el.mIndex - 1] = el.mNodesInd
        self.mMassMatrix = BlockDiagonal(blocks, mP + 1, inds)
        self.mMassMatrixLU = self.mMassMatrix.lu()
        self.mSwitch = _switch_vector(
            mesh.mVertices, lambda k: mesh.mFaces[k - 1],
            lambda face: (face.mVertices[0].mX, face.mVertices[1].mX))


def dg_flux_operators_dg(dgMesh, mesh, bdCond, CDir):
    """dg_flux_operators(::DgMesh, ...) src/dg_mesh.jl:144-336 -> (G, D, C).
    LDG fluxes: uhat = u_L, qhat = q_R on interior vertices."""
    refEl = dgMesh.mRefEl
    dG, dD, dC = [], [], []
    p1 = dgMesh.mP >= 1
    # local node used for "left end" / "right end" of an element: (1, 2) for p>=1, (1, 1) p=0
    nL, nR = (0, 1) if p1 else (0, 0)
    if p1:
        for el in dgMesh.mElements:
            n = len(el.mNodesInd)
            temp = np.zeros((n, n))
            for j in range(n):
                for i in range(n):
                    for l in range(len(refEl.mGaussQuadNodes)):
                        temp[i, j] += (refEl.mGaussQuadWeights[l] *
                                       refEl.mBasisGQDerivVal[l, i] * refEl.mBasisGQFunVal[l, j])
            for j, node2 in enumerate(el.mNodesInd):
                for i, node1 in enumerate(el.mNodesInd):
                    dG.append((node1, node2, temp[i, j]))
                    dD.append((node1, node2, temp[i, j]))
    for i, vert in enumerate(mesh.mVertices):
        if isBoundary(vert):
            meshEl = mesh.mFaces[vert.mFaces[0] - 1]
            dgEl = dgMesh.mElements[vert.mFaces[0] - 1]
            if vert.mIndex in bdCond.mDirNodes:
                if vert is meshEl.mVertices[0]:
                    dD.append((dgEl.mNodesInd[nL], dgEl.mNodesInd[nL], 1.0))
                    dC.append((dgEl.mNodesInd[nL], dgEl.mNodesInd[nL], CDir))
                elif vert is meshEl.mVertices[1]:
                    dD.append((dgEl.mNodesInd[nR], dgEl.mNodesInd[nR], -1.0))
                    dC.append((dgEl.mNodesInd[nR], dgEl.mNodesInd[nR], CDir))
                else:
                    raise RuntimeError("vertex / element mismatch")
            elif vert.mIndex in bdCond.mNeuNodes:
                if vert is meshEl.mVertices[0]:
                    dG.append((dgEl.mNodesInd[nL], dgEl.mNodesInd[nL], 1.0))
                elif vert is meshEl.mVertices[1]:
                    dG.append((dgEl.mNodesInd[nR], dgEl.mNodesInd[nR], -1.0))
                else:
                    raise RuntimeError("vertex / element mismatch")
            else:
                raise RuntimeError("Boundary vertex is not included in the boundary condition.")
        else:
            S = dgMesh.mSwitch[i]
            uhatEl = dgMesh.mElements[vert.mFaces[S - 1] - 1]
            qhatEl = dgMesh.mElements[vert.mFaces[S % 2] - 1]
            for k in vert.mFaces:
                meshEl = mesh.mFaces[k - 1]
                dgEl = dgMesh.mElements[k - 1]
                if vert is meshEl.mVertices[0]:
                    dG.append((dgEl.mNodesInd[nL], uhatEl.mNodesInd[nR], 1.0))
                    dD.append((dgEl.mNodesInd[nL], qhatEl.mNodesInd[nL], 1.0))
                elif vert is meshEl.mVertices[1]:
                    dG.append((dgEl.mNodesInd[nR], uhatEl.mNodesInd[nR], -1.0))
                    dD.append((dgEl.mNodesInd[nR], qhatEl.mNodesInd[nL], -1.0))
                else:
                    raise RuntimeError("vertex / element mismatch")
    N = dgMesh.mNumNodes

    def mk(data):
        if not data:
            return sp.csc_matrix((N, N))
        I, J, V = zip(*data)
        return sparse(I, J, V, N, N)

    return mk(dG), mk(dD), mk(dC)


def dg_flux_rhs_dg(dgMesh, mesh, func, bdCond, CDir):
    """dg_flux_rhs(::DgMesh, ...) src/dg_mesh.jl:342-457 -> (f, r)."""
    f = np.zeros(dgMesh.mNumNodes)
    r = np.zeros(dgMesh.mNumNodes)
    refEl = dgMesh.mRefEl
    for el in dgMesh.mElements:
        for i, node in enumerate(el.mNodesInd):
            for l in range(len(refEl.mGaussQuadNodes)):
                f[node - 1] += (el.mJacobian * refEl.mGaussQuadWeights[l] *
                                refEl.mBasisGQFunVal[l, i] *
                                func(el.mRefMap(refEl.mGaussQuadNodes[l])))
    nL, nR = (0, 1) if dgMesh.mP >= 1 else (0, 0)
    for i, nodeIdx in enumerate(bdCond.mDirNodes):
        vert = mesh.mVertices[nodeIdx - 1]
        dirVal = bdCond.mDirVals[i]
        meshEl = mesh.mFaces[vert.mFaces[0] - 1]
        dgEl = dgMesh.mElements[vert.mFaces[0] - 1]
        if vert is meshEl.mVertices[0]:
            f[dgEl.mNodesInd[nL] - 1] += CDir * dirVal
            r[dgEl.mNodesInd[nL] - 1] += -dirVal
        elif vert is meshEl.mVertices[1]:
            f[dgEl.mNodesInd[nR] - 1] += CDir * dirVal
            r[dgEl.mNodesInd[nR] - 1] += dirVal
        else:
            raise RuntimeError("vertex / element mismatch")
    for nodeIdx in bdCond.mNeuNodes:
        vert = mesh.mVertices[nodeIdx - 1]
        meshEl = mesh.mFaces[vert.mFaces[0] - 1]
        dgEl = dgMesh.mElements[vert.mFaces[0] - 1]
        if vert is meshEl.mVertices[0]:
            f[dgEl.mNodesInd[nL] - 1] += -bdCond.mBdCond[0][1]
        elif vert is meshEl.mVertices[1]:
            f[dgEl.mNodesInd[nR] - 1] += bdCond.mBdCond[1][1]
        else:
            raise RuntimeError("vertex / element mismatch")
    return f, r


# --------------------------------------------------------------------------------------
# src/agglomerated_dg_mesh.jl
# --------------------------------------------------------------------------------------


def evaluate_local_modal_basis_fun(p, boundingBox, nodes):
    """src/agglomerated_dg_mesh.jl:297-315 (p in {0,1} only)."""
    nodes = np.atleast_1d(np.asarray(nodes, dtype=np.float64))
    val = np.zeros((len(nodes), p + 1))
    if p == 0:
        val[:, 0] = 1.0
    elif p == 1:
        xC = (boundingBox[0] + boundingBox[1]) / 2.0
        h = boundingBox[1] - boundingBox[0]
        val[:, 0] = 1.0
        val[:, 1] = 2 * (nodes - xC) / h
    else:
        raise ValueError("Only implemented for p = 0 and p = 1.")
    return val


def evaluate_local_modal_basis_deriv(p, boundingBox):
    """src/agglomerated_dg_mesh.jl:317-327"""
    if p == 0:
        return np.array([0.0])
    if p == 1:
        h = boundingBox[1] - boundingBox[0]
        return np.array([0.0, 2.0 / h])
    raise ValueError("Only implemented for p = 0 and p = 1.")


class AgglomeratedDgVertex:
    """src/agglomerated_dg_mesh.jl:1-7"""

    def __init__(self, mIndex, mX):
        self.mIndex = mIndex
        self.mX = mX
        self.mFaces = [0, 0]


class AgglomeratedDgElement1:
    """src/agglomerated_dg_mesh.jl:9-30, ctor :175-256 (the gaussQuadNodes variant)."""

    def __init__(self, mIndex, mP, mBaseElementInds, baseMesh, mesh, allVertices,
                 gaussQuadNodes):
        self.mIndex = mIndex
        self.mP = mP
        self.mNodesInd = list(range((mIndex - 1) * (mP + 1) + 1, mIndex * (mP + 1) + 1))
        self.mBaseElementInds = list(mBaseElementInds)
        self.mSubAggElementInds = self.mBaseElementInds
        min_x, max_x = math.inf, -math.inf
        for elInd in self.mBaseElementInds:
            min_x = min(min_x, baseMesh.mElements[elInd - 1].mNodesX[0])
            max_x = max(max_x, baseMesh.mElements[elInd - 1].mNodesX[1])
        self.mBoundingBox = [min_x, max_x]
        self.mBasisGQFunVal = []
        for elInd in self.mBaseElementInds:
            el = baseMesh.mElements[elInd - 1]
            elGQ = np.array([el.mRefMap(x) for x in gaussQuadNodes])
            self.mBasisGQFunVal.append(
                evaluate_local_modal_basis_fun(mP, self.mBoundingBox, elGQ))
        self.mBasisDerivVal = evaluate_local_modal_basis_deriv(mP, self.mBoundingBox)
        self.mVertices = []
        self.mVertices2 = []
        for i in self.mBaseElementInds:
            el = mesh.mFaces[i - 1]
            for vert in el.mVertices:
                isBd = not ((vert.mFaces[0] in self.mBaseElementInds) and
                            (vert.mFaces[1] in self.mBaseElementInds))
                if isBd:
                    self.mVertices2.append(vert)
                    self.mVertices.append(allVertices[vert.mIndex - 1])
        self.mBdBasisGQFunVal = [
            evaluate_local_modal_basis_fun(mP, self.mBoundingBox, v.mX)[0, :]
            for v in self.mVertices]


def _agg_mass_blocks(mP, mElements, baseMesh, gqW):
    blocks = [None] * len(mElements)
    inds = np.zeros((mP + 1, len(mElements)), dtype=np.int64)
    for el in mElements:
        n = len(el.mNodesInd)
        temp = np.zeros((n, n))
        for k, baseElInd in enumerate(el.mBaseElementInds):
            baseEl = baseMesh.mElements[baseElInd - 1]
            for j in range(n):
                for i in range(n):
                    for l in range(len(gqW)):
                        temp[i, j] += (baseEl.mJacobian * gqW[l] *
                                       el.mBasisGQFunVal[k][l, i] * el.mBasisGQFunVal[k][l, j])
        blocks[el.mIndex - 1] = temp
        inds[:, el.mIndex - 1] = el.mNodesInd
    return BlockDiagonal(blocks, mP + 1, inds)


class AgglomeratedDgMesh1:
    """src/agglomerated_dg_mesh.jl:32-47, ctor (mP, agg, mesh, baseMesh) :391-495."""

    def __init__(self, mP, agg, mesh, baseMesh):
        self.mP = mP
        self.mGaussQuadNodes, self.mGaussQuadWeights = gauss_quad(2 * mP)
        self.mAllVertices = [AgglomeratedDgVertex(v.mIndex, v.mX) for v in mesh.mVertices]
        self.mElements = [
            AgglomeratedDgElement1(k + 1, mP, baseElInds, baseMesh, mesh, self.mAllVertices,
                                   self.mGaussQuadNodes)
            for k, baseElInds in enumerate(agg)]
        self.mNumNodes = len(self.mElements) * (mP + 1)
        self.mVertices = []
        for el in self.mElements:
            for vert in el.mVertices:
                if vert.mFaces[0] == 0:
                    vert.mFaces[0] = el.mIndex
                    self.mVertices.append(vert)
                elif vert.mFaces[1] == 0:
                    vert.mFaces[1] = el.mIndex
                else:
                    raise RuntimeError("Vertex can only neighbor two elements.")
        self.mMassMatrix = _agg_mass_blocks(mP, self.mElements, baseMesh,
                                            self.mGaussQuadWeights)
        self.mMassMatrixLU = self.mMassMatrix.lu()
        self.mSwitch = _switch_vector(
            self.mVertices, lambda k: self.mElements[k - 1],
            lambda el: (el.mVertices[0].mX, el.mVertices[1].mX))


class AgglomeratedDgElementN:
    """src/agglomerated_dg_mesh.jl:49-60, ctor :501-556."""

    def __init__(self, mIndex, mP, mSubAggElementInds, subAggMesh, baseMesh):
        self.mIndex = mIndex
        self.mP = mP
        self.mNodesInd = list(range((mIndex - 1) * (mP + 1) + 1, mIndex * (mP + 1) + 1))
        self.mSubAggElementInds = list(mSubAggElementInds)
        self.mBaseElementInds = []
        for elInd in self.mSubAggElementInds:
            self.mBaseElementInds.extend(subAggMesh.mElements[elInd - 1].mBaseElementInds)
        min_x, max_x = math.inf, -math.inf
        for elInd in self.mSubAggElementInds:
            min_x = min(min_x, subAggMesh.mElements[elInd - 1].mBoundingBox[0])
            max_x = max(max_x, subAggMesh.mElements[elInd - 1].mBoundingBox[1])
        self.mBoundingBox = [min_x, max_x]
        gq, _ = gauss_quad(2 * mP)
        self.mBasisGQFunVal = []
        for elInd in self.mBaseElementInds:
            el = baseMesh.mElements[elInd - 1]
            elGQ = np.array([el.mRefMap(x) for x in gq])
            self.mBasisGQFunVal.append(
                evaluate_local_modal_basis_fun(mP, self.mBoundingBox, elGQ))
        self.mBasisDerivVal = evaluate_local_modal_basis_deriv(mP, self.mBoundingBox)


class AgglomeratedDgMeshN:
    """src/agglomerated_dg_mesh.jl:62-72, ctor (mP, agg, subAggMesh, baseMesh) :593-635."""

    def __init__(self, mP, agg, subAggMesh, baseMesh):
        self.mP = mP
        self.mGaussQuadNodes, self.mGaussQuadWeights = gauss_quad(2 * mP)
        self.mElements = [AgglomeratedDgElementN(k + 1, mP, inds, subAggMesh, baseMesh)
                          for k, inds in enumerate(agg)]
        self.mNumNodes = len(self.mElements) * (mP + 1)
        self.mMassMatrix = _agg_mass_blocks(mP, self.mElements, baseMesh,
                                            self.mGaussQuadWeights)
        self.mMassMatrixLU = self.mMassMatrix.lu()


def dg_flux_operators_agg(aggDgMesh, baseMesh, bdCond, CDir):
    """dg_flux_operators(::AgglomeratedDgMesh1, ...) src/agglomerated_dg_mesh.jl:641-870."""
    dG, dD, dC = [], [], []
    p1 = aggDgMesh.mP >= 1
    gqW = aggDgMesh.mGaussQuadWeights

    def modes(el):
        # p>=1 loops over all local modes; the p==0 branch addresses mode 1 only
        return list(enumerate(el.mNodesInd)) if p1 else [(0, el.mNodesInd[0])]

    if p1:
        for el in aggDgMesh.mElements:
            n = len(el.mNodesInd)
            temp = np.zeros((n, n))
            for k, baseElInd in enumerate(el.mBaseElementInds):
                baseEl = baseMesh.mElements[baseElInd - 1]
                for j in range(n):
                    for i in range(n):
                        for l in range(len(gqW)):
                            temp[i, j] += (baseEl.mJacobian * gqW[l] * el.mBasisDerivVal[i] *
                                           el.mBasisGQFunVal[k][l, j])
            for j, node2 in enumerate(el.mNodesInd):
                for i, node1 in enumerate(el.mNodesInd):
                    dG.append((node1, node2, temp[i, j]))
                    dD.append((node1, node2, temp[i, j]))
    for i, vert in enumerate(aggDgMesh.mVertices):
        if isBoundary(vert):
            el = aggDgMesh.mElements[vert.mFaces[0] - 1]
            if vert is el.mVertices[0]:
                sign, side = -1.0, 0
            elif vert is el.mVertices[1]:
                sign, side = 1.0, 1
            else:
                raise RuntimeError("vertex / element mismatch")
            bv = el.mBdBasisGQFunVal[side]
            if vert.mIndex in bdCond.mDirNodes:
                for j, node2 in modes(el):
                    for ii, node1 in modes(el):
                        dD.append((node1, node2, -sign * bv[ii] * bv[j]))
                        dC.append((node1, node2, CDir * bv[ii] * bv[j]))
            elif vert.mIndex in bdCond.mNeuNodes:
                for j, node2 in modes(el):
                    for ii, node1 in modes(el):
                        dG.append((node1, node2, -sign * bv[ii] * bv[j]))
            else:
                raise RuntimeError("Boundary vertex is not included in the boundary condition.")
        else:
            S = aggDgMesh.mSwitch[i]
            uhatEl = aggDgMesh.mElements[vert.mFaces[S - 1] - 1]
            qhatEl = aggDgMesh.mElements[vert.mFaces[S % 2] - 1]
            for k in vert.mFaces:
                el = aggDgMesh.mElements[k - 1]
                if vert is el.mVertices[0]:
                    sign, side = -1.0, 0
                elif vert is el.mVertices[1]:
                    sign, side = 1.0, 1
                else:
                    raise RuntimeError("vertex / element mismatch")
                bv = el.mBdBasisGQFunVal[side]
                for j, node2 in modes(uhatEl):
                    for ii, node1 in modes(el):
                        dG.append((node1, node2, -sign * bv[ii] * uhatEl.mBdBasisGQFunVal[1][j]))
                for j, node2 in modes(qhatEl):
                    for ii, node1 in modes(el):
                        dD.append((node1, node2, -sign * bv[ii] * qhatEl.mBdBasisGQFunVal[0][j]))
    N = aggDgMesh.mNumNodes

    def mk(data):
        if not data:
            return sp.csc_matrix((N, N))
        I, J, V = zip(*data)
        return sparse(I, J, V, N, N)

    return mk(dG), mk(dD), mk(dC)


def dg_flux_rhs_agg(aggDgMesh, baseMesh, func, bdCond, CDir):
    """dg_flux_rhs(::AgglomeratedDgMesh1, ...) src/agglomerated_dg_mesh.jl:872-994."""
    f = np.zeros(aggDgMesh.mNumNodes)
    r = np.zeros(aggDgMesh.mNumNodes)
    gq, gqW = aggDgMesh.mGaussQuadNodes, aggDgMesh.mGaussQuadWeights
    for el in aggDgMesh.mElements:
        for k, baseElInd in enumerate(el.mBaseElementInds):
            baseEl = baseMesh.mElements[baseElInd - 1]
            for i, node in enumerate(el.mNodesInd):
                for l in range(len(gq)):
                    f[node - 1] += (baseEl.mJacobian * gqW[l] * el.mBasisGQFunVal[k][l, i] *
                                    func(baseEl.mRefMap(gq[l])))
    p1 = aggDgMesh.mP >= 1

    def modes(el):
        return list(enumerate(el.mNodesInd)) if p1 else [(0, el.mNodesInd[0])]

    for i, nodeIdx in enumerate(bdCond.mDirNodes):
        vert = aggDgMesh.mAllVertices[nodeIdx - 1]
        dirVal = bdCond.mDirVals[i]
        el = aggDgMesh.mElements[vert.mFaces[0] - 1]
        if vert is el.mVertices[0]:
            sign, side = -1.0, 0
        elif vert is el.mVertices[1]:
            sign, side = 1.0, 1
        else:
            raise RuntimeError("vertex / element mismatch")
        for ii, node in modes(el):
            f[node - 1] += CDir * dirVal * el.mBdBasisGQFunVal[side][ii]
            r[node - 1] += sign * dirVal * el.mBdBasisGQFunVal[side][ii]
    for nodeIdx in bdCond.mNeuNodes:
        vert = aggDgMesh.mAllVertices[nodeIdx - 1]
        el = aggDgMesh.mElements[vert.mFaces[0] - 1]
        if vert is el.mVertices[0]:
            sign, side, neuVal = -1.0, 0, bdCond.mBdCond[0][1]
        elif vert is el.mVertices[1]:
            sign, side, neuVal = 1.0, 1, bdCond.mBdCond[1][1]
        else:
            raise RuntimeError("vertex / element mismatch")
        for ii, node in modes(el):
            f[node - 1] += sign * neuVal * el.mBdBasisGQFunVal[side][ii]
    return f, r


def dg_flux_operators(m, base, bdCond, CDir):
    """Dispatch of the two `dg_flux_operators` methods."""
    if isinstance(m, DgMesh):
        return dg_flux_operators_dg(m, base, bdCond, float(CDir))
    return dg_flux_operators_agg(m, base, bdCond, float(CDir))


def dg_flux_rhs(m, base, func, bdCond, CDir):
    if isinstance(m, DgMesh):
        return dg_flux_rhs_dg(m, base, func, bdCond, float(CDir))
    return dg_flux_rhs_agg(m, base, func, bdCond, float(CDir))


def dg_stiffness(m, G, D, C):
    """`A = C - D * (M_LU \\ G)` src/mesh_heirarchy.jl:71-72 / tests/dg_heirarchy_test.jl:37."""
    return sp_sub(C, (D @ m.mMassMatrixLU.solve_sparse(G)).tocsc())


def dg_rhs(m, D, f, r):
    """`b = f - D * (M_LU \\ r)` tests/dg_heirarchy_test.jl:40."""
    return f - csc_matvec(D, m.mMassMatrixLU.solve_dense(r))


# --------------------------------------------------------------------------------------
# src/interpolation.jl  (prolongations L: fine x coarse; restriction is L')
# --------------------------------------------------------------------------------------


def _csc_pos(P, r, c):
    """position of stored entry (r, c) (0-based) in a sorted CSC matrix"""
    lo, hi = P.indptr[c], P.indptr[c + 1]
    k = lo + int(np.searchsorted(P.indices[lo:hi], r))
    assert k < hi and P.indices[k] == r
    return k


def cg_cg_interpolation(lowMesh, highMesh):
    """src/interpolation.jl:5-55 (nodal injection, p-coarsening).  The pattern is fixed by the
    first `sparse(...)` of zeros; values are then written slot by slot, interior rows only
    where the slot is still <= eps."""
    lowVal = evaluate_nodal_basis_fun(lowMesh.mRefEl.mBasisFunCoeff, highMesh.mRefEl.mNodesX)
    I, J = [], []
    for k, lowEl in enumerate(lowMesh.mElements):
        highEl = highMesh.mElements[k]
        for j in range(len(lowEl.mNodesInd)):
            for i in range(2, len(highEl.mNodesInd)):
                I.append(highEl.mNodesInd[i]), J.append(lowEl.mNodesInd[j])
        for j in range(2):
            I.append(highEl.mNodesInd[j]), J.append(lowEl.mNodesInd[j])
    L = sparse(I, J, np.zeros(len(I)), highMesh.mNumNodes, lowMesh.mNumNodes)
    for k, lowEl in enumerate(lowMesh.mElements):
        highEl = highMesh.mElements[k]
        for j in range(len(lowEl.mNodesInd)):
            lowNode = lowEl.mNodesInd[j]
            for i in range(2, len(highEl.mNodesInd)):
                pos = _csc_pos(L, highEl.mNodesInd[i] - 1, lowNode - 1)
                if abs(L.data[pos]) <= EPS:
                    L.data[pos] = lowVal[i, j]
        for j in range(2):
            L.data[_csc_pos(L, highEl.mNodesInd[j] - 1, lowEl.mNodesInd[j] - 1)] = lowVal[j, j]
    return L


def cg_cg_interpolation2(lowMesh, highMesh):
    """src/interpolation.jl:57-85 (L2 projection between CG spaces; returns a DENSE matrix:
    `highMesh.mMassMatrixLU \\ Array(N)`)."""
    gq, gqW = gauss_quad(lowMesh.mP + highMesh.mP)
    hiV = evaluate_nodal_basis_fun(highMesh.mRefEl.mBasisFunCoeff, gq)
    loV = evaluate_nodal_basis_fun(lowMesh.mRefEl.mBasisFunCoeff, gq)
    I, J, V = [], [], []
    for k, lowEl in enumerate(lowMesh.mElements):
        highEl = highMesh.mElements[k]
        temp = np.zeros((len(highEl.mNodesInd), len(lowEl.mNodesInd)))
        for j in range(len(lowEl.mNodesInd)):
            for i in range(len(highEl.mNodesInd)):
                for l in range(len(gq)):
                    temp[i, j] += lowEl.mJacobian * gqW[l] * hiV[l, i] * loV[l, j]
        for j, lowNode in enumerate(lowEl.mNodesInd):
            for i, highNode in enumerate(highEl.mNodesInd):
                I.append(highNode), J.append(lowNode), V.append(temp[i, j])
    N = sparse(I, J, V, highMesh.mNumNodes, lowMesh.mNumNodes)
    return highMesh.mMassMatrixLU.solve(N.toarray())


def dg_dg_interpolation2(lowMesh, highMesh):
    """src/interpolation.jl:111-139: as dg_dg_interpolation for the interior nodes of the fine element, the two
    end nodes take the value of their own coarse end node only."""
    lowVal = evaluate_nodal_basis_fun(lowMesh.mRefEl.mBasisFunCoeff, highMesh.mRefEl.mNodesX)
    I, J, V = [], [], []
    for k, lowEl in enumerate(lowMesh.mElements):
        highEl = highMesh.mElements[k]
        for j, lowNode in enumerate(lowEl.mNodesInd):
            for i in range(2, len(highEl.mNodesInd)):
                I.append(highEl.mNodesInd[i]), J.append(lowNode), V.append(lowVal[i, j])
        for j in range(2):
            I.append(highEl.mNodesInd[j]), J.append(lowEl.mNodesInd[j]), V.append(lowVal[j, j])
    return sparse(I, J, V, highMesh.mNumNodes, lowMesh.mNumNodes)


def dg_dg_interpolation(lowMesh, highMesh):
    """src/interpolation.jl:91-109"""
    lowVal = evaluate_nodal_basis_fun(lowMesh.mRefEl.mBasisFunCoeff, highMesh.mRefEl.mNodesX)
    I, J, V = [], [], []
    for k, lowEl in enumerate(lowMesh.mElements):
        highEl = highMesh.mElements[k]
        for j, lowNode in enumerate(lowEl.mNodesInd):
            for i, highNode in enumerate(highEl.mNodesInd):
                I.append(highNode), J.append(lowNode), V.append(lowVal[i, j])
    return sparse(I, J, V, highMesh.mNumNodes, lowMesh.mNumNodes)


def _lumped_rowscale(massMatrix, N):
    """`Diagonal(lumped) \\ N` with lumped[j] = sum(M[j,:]) (src/interpolation.jl:207-216,
    :397-406): row scaling on N's pattern, by division."""
    M = massMatrix.tocsr()
    lumped = np.zeros(M.shape[0])
    for j in range(M.shape[0]):
        lumped[j] = np.sum(M.data[M.indptr[j]:M.indptr[j + 1]])
    L = N.tocsc(copy=True)
    L.data = L.data / lumped[L.indices]
    return L


def dg_cg_interpolation(lowMesh, highMesh, mesh, interpFlag):
    """src/interpolation.jl:145-220.  lowMesh: DgMesh (coarser), highMesh: CgMesh (finer)."""
    if interpFlag in (0, 1):
        gq, gqW = gauss_quad(lowMesh.mP + highMesh.mP)
        hiV = evaluate_nodal_basis_fun(highMesh.mRefEl.mBasisFunCoeff, gq)
        loV = evaluate_nodal_basis_fun(lowMesh.mRefEl.mBasisFunCoeff, gq)
        I, J, V = [], [], []
        for k, lowEl in enumerate(lowMesh.mElements):
            highEl = highMesh.mElements[k]
            temp = np.zeros((len(highEl.mNodesInd), len(lowEl.mNodesInd)))
            for j in range(len(lowEl.mNodesInd)):
                for i in range(len(highEl.mNodesInd)):
                    for l in range(len(gq)):
                        temp[i, j] += lowEl.mJacobian * gqW[l] * hiV[l, i] * loV[l, j]
            for j, lowNode in enumerate(lowEl.mNodesInd):
                for i, highNode in enumerate(highEl.mNodesInd):
                    I.append(highNode), J.append(lowNode), V.append(temp[i, j])
        N = sparse(I, J, V, highMesh.mNumNodes, lowMesh.mNumNodes)
        if interpFlag == 0:
            return highMesh.mMassMatrixLU.solve(N.toarray())  # dense (Array(N))
        return _lumped_rowscale(highMesh.mMassMatrix, N)
    if interpFlag == 2:
        lowVal = evaluate_nodal_basis_fun(lowMesh.mRefEl.mBasisFunCoeff,
                                          highMesh.mRefEl.mNodesX)
        I, J, V = [], [], []
        for k, lowEl in enumerate(lowMesh.mElements):
            highEl = highMesh.mElements[k]
            for j, lowNode in enumerate(lowEl.mNodesInd):
                for i in range(2):
                    highNode = highEl.mNodesInd[i]
                    vert = mesh.mVertices[highNode - 1]
                    w = 1.0 if isBoundary(vert) else 0.5
                    I.append(highNode), J.append(lowNode)
                    V.append(lowVal[i, j] if w == 1.0 else 0.5 * lowVal[i, j])
                for i in range(2, len(highEl.mNodesInd)):
                    I.append(highEl.mNodesInd[i]), J.append(lowNode), V.append(lowVal[i, j])
        return sparse(I, J, V, highMesh.mNumNodes, lowMesh.mNumNodes)
    raise ValueError("Only implemented for interpFlag = 0, 1, or 2.")


def aggdg_aggdg_interpolation(coarseMesh, fineMesh, baseMesh):
    """src/interpolation.jl:226-264: L = M_fine^{-1} N via BlockDiagonalLU \\ sparse."""
    if coarseMesh.mP != fineMesh.mP:
        raise ValueError("The two agglomerated meshes must have the same p.")
    p = coarseMesh.mP
    gqW = fineMesh.mGaussQuadWeights
    I, J, V = [], [], []
    for coarseEl in coarseMesh.mElements:
        count = 0
        for fineElInd in coarseEl.mSubAggElementInds:
            fineEl = fineMesh.mElements[fineElInd - 1]
            temp = np.zeros((p + 1, p + 1))
            for k, baseElInd in enumerate(fineEl.mBaseElementInds):
                baseEl = baseMesh.mElements[baseElInd - 1]
                for j in range(len(coarseEl.mNodesInd)):
                    for i in range(len(fineEl.mNodesInd)):
                        for l in range(len(fineMesh.mGaussQuadNodes)):
                            temp[i, j] += (baseEl.mJacobian * gqW[l] *
                                           fineEl.mBasisGQFunVal[k][l, i] *
                                           coarseEl.mBasisGQFunVal[count + k][l, j])
            count += len(fineEl.mBaseElementInds)
            for j, node2 in enumerate(coarseEl.mNodesInd):
                for i, node1 in enumerate(fineEl.mNodesInd):
                    I.append(node1), J.append(node2), V.append(temp[i, j])
    N = sparse(I, J, V, fineMesh.mNumNodes, coarseMesh.mNumNodes)
    return fineMesh.mMassMatrixLU.solve_sparse(N)


def aggdg_dg_interpolation(aggMesh, baseMesh):
    """src/interpolation.jl:270-292: modal basis evaluated at the base DG nodes."""
    refEl = baseMesh.mRefEl
    I, J, V = [], [], []
    for aggEl in aggMesh.mElements:
        for baseElInd in aggEl.mBaseElementInds:
            baseEl = baseMesh.mElements[baseElInd - 1]
            val = evaluate_local_modal_basis_fun(
                aggMesh.mP, aggEl.mBoundingBox, [baseEl.mRefMap(x) for x in refEl.mNodesX])
            for j, aggNode in enumerate(aggEl.mNodesInd):
                for i, baseNode in enumerate(baseEl.mNodesInd):
                    I.append(baseNode), J.append(aggNode), V.append(val[i, j])
    return sparse(I, J, V, baseMesh.mNumNodes, aggMesh.mNumNodes)


def aggdg_dg_interpolation2(aggMesh, baseMesh):
    """src/interpolation.jl:294-324 (L2 projection variant)."""
    refEl = baseMesh.mRefEl
    gq, gqW = refEl.mGaussQuadNodes, refEl.mGaussQuadWeights
    I, J, V = [], [], []
    for aggEl in aggMesh.mElements:
        for baseElInd in aggEl.mBaseElementInds:
            baseEl = baseMesh.mElements[baseElInd - 1]
            temp = np.zeros((len(baseEl.mNodesInd), len(aggEl.mNodesInd)))
            aggV = evaluate_local_modal_basis_fun(
                aggMesh.mP, aggEl.mBoundingBox, [baseEl.mRefMap(x) for x in gq])
            for j in range(len(aggEl.mNodesInd)):
                for i in range(len(baseEl.mNodesInd)):
                    for l in range(len(gq)):
                        temp[i, j] += (baseEl.mJacobian * gqW[l] * refEl.mBasisGQFunVal[l, i] *
                                       aggV[l, j])
            for j, aggNode in enumerate(aggEl.mNodesInd):
                for i, baseNode in enumerate(baseEl.mNodesInd):
                    I.append(baseNode), J.append(aggNode), V.append(temp[i, j])
    N = sparse(I, J, V, baseMesh.mNumNodes, aggMesh.mNumNodes)
    return baseMesh.mMassMatrixLU.solve_sparse(N)


def aggdg_cg_interpolation(aggMesh, baseMesh, mesh, interpFlag):
    """src/interpolation.jl:330-410.  baseMesh: CgMesh."""
    refEl = baseMesh.mRefEl
    if interpFlag in (0, 1):
        gq, gqW = refEl.mGaussQuadNodes, refEl.mGaussQuadWeights
        I, J, V = [], [], []
        for aggEl in aggMesh.mElements:
            for baseElInd in aggEl.mBaseElementInds:
                baseEl = baseMesh.mElements[baseElInd - 1]
                temp = np.zeros((len(baseEl.mNodesInd), len(aggEl.mNodesInd)))
                aggV = evaluate_local_modal_basis_fun(
                    aggMesh.mP, aggEl.mBoundingBox, [baseEl.mRefMap(x) for x in gq])
                for j in range(len(aggEl.mNodesInd)):
                    for i in range(len(baseEl.mNodesInd)):
                        for l in range(len(gq)):
                            temp[i, j] += (baseEl.mJacobian * gqW[l] *
                                           refEl.mBasisGQFunVal[l, i] * aggV[l, j])
                for j, aggNode in enumerate(aggEl.mNodesInd):
                    for i, baseNode in enumerate(baseEl.mNodesInd):
                        I.append(baseNode), J.append(aggNode), V.append(temp[i, j])
        N = sparse(I, J, V, baseMesh.mNumNodes, aggMesh.mNumNodes)
        if interpFlag == 0:
            return baseMesh.mMassMatrixLU.solve(N.toarray())
        return _lumped_rowscale(baseMesh.mMassMatrix, N)
    if interpFlag == 2:
        I, J, V = [], [], []
        for aggEl in aggMesh.mElements:
            for baseElInd in aggEl.mBaseElementInds:
                baseEl = baseMesh.mElements[baseElInd - 1]
                val = evaluate_local_modal_basis_fun(
                    aggMesh.mP, aggEl.mBoundingBox,
                    [baseEl.mRefMap(x) for x in refEl.mNodesX])
                for j, aggNode in enumerate(aggEl.mNodesInd):
                    for i in range(2):
                        baseNode = baseEl.mNodesInd[i]
                        vert = mesh.mVertices[baseNode - 1]
                        I.append(baseNode), J.append(aggNode)
                        V.append(val[i, j] if isBoundary(vert) else 0.5 * val[i, j])
                    for i in range(2, len(baseEl.mNodesInd)):
                        I.append(baseEl.mNodesInd[i]), J.append(aggNode), V.append(val[i, j])
        return sparse(I, J, V, baseMesh.mNumNodes, aggMesh.mNumNodes)
    raise ValueError("Only implemented for interpFlag = 0, 1, or 2.")


# --------------------------------------------------------------------------------------
# src/smoother.jl
# --------------------------------------------------------------------------------------


class JacobiSmoother:
    """src/smoother.jl:52-58: alpha * (Diagonal \\ B)"""

    def __init__(self, mJac):
        self.mJac = np.asarray(mJac, dtype=np.float64)

    def apply(self, B, alpha=1.0):
        B = np.asarray(B, dtype=np.float64)
        d = self.mJac if B.ndim == 1 else self.mJac[:, None]
        return alpha * (B / d)


class BlockJacobi:
    """src/smoother.jl:64-81: Y=zeros; per column, per block Y[inds] += LU_i \\ B[inds];
    return alpha*Y.  (AdditiveSchwarzSmoother :1-18 is the same loop with overlapping
    blocks.)"""

    def __init__(self, mBlocks, mBlockInds):
        self.mBlocks = mBlocks
        self.mBlockInds = np.asarray(mBlockInds, dtype=np.int64)  # (p+1) x n, 1-based

    def apply(self, B, alpha=1.0):
        B = np.asarray(B, dtype=np.float64)
        Y = np.zeros(B.shape)
        cols = [None] if B.ndim == 1 else range(B.shape[1])
        for j in cols:
            for i, block in enumerate(self.mBlocks):
                idx = self.mBlockInds[:, i] - 1
                if j is None:
                    Y[idx] += block.solve(B[idx])
                else:
                    Y[idx, j] += block.solve(B[idx, j])
        return alpha * Y


AdditiveSchwarzSmoother = BlockJacobi


class BlockGaussSeidelRB(BlockJacobi):
    """EXTENSION -- the reference has no Gauss-Seidel smoother; BASELINE.json's
    north_star / config 5 name one, so the product offers a red-black block Gauss-Seidel on
    block-tridiagonal operators and this class restates it for the tests.  No reference parity
    claim: it is checked against this restatement only.

    One sweep with damping alpha, blocks in element order, colour = element index parity
    (0-based): for colour in (0, 1) [reverse=True: (1, 0)]:
        r = b - A u;   u[inds_e] += alpha * (LU_e \ r[inds_e])   for every element e of that colour.
    apply() (inherited) is the block-diagonal part alone, as for BlockJacobi."""

    def sweep(self, A, u, b, alpha=1.0, reverse=False):
        u = np.array(u, dtype=np.float64)
        for colour in ((1, 0) if reverse else (0, 1)):
            r = b - csc_matvec(A, u)
            for e in range(colour, len(self.mBlocks), 2):
                idx = self.mBlockInds[:, e] - 1
                u[idx] += alpha * self.mBlocks[e].solve(r[idx])
        return u


def smooth_once(S, A, u, rhs, alpha, post=False):
    """one smoothing step of the V-cycle.  Reference smoothers: exactly the expression of
    src/solvers.jl:33 / :44, `u + apply_smoother(S, rhs - A*u; alpha)`.  The Gauss-Seidel extension
    sweeps its two colours, in reverse order when post-smoothing."""
    if isinstance(S, BlockGaussSeidelRB):
        return S.sweep(A, u, rhs, alpha, reverse=post)
    return u + apply_smoother(S, rhs - csc_matvec(A, u), alpha=alpha)


class HybridSchwarzSmoother:
    """src/smoother.jl:24-46"""

    def __init__(self, mBlocks, mBlockInds, mCountingMatrix):
        self.mBlocks = mBlocks
        self.mBlockInds = np.asarray(mBlockInds, dtype=np.int64)
        self.mCountingMatrix = np.asarray(mCountingMatrix, dtype=np.float64)

    def apply(self, B, alpha=1.0):
        B = np.asarray(B, dtype=np.float64)
        Y = np.zeros(B.shape)
        cols = [None] if B.ndim == 1 else range(B.shape[1])
        for j in cols:
            temp = np.zeros(B.shape[0])
            for i, block in enumerate(self.mBlocks):
                idx = self.mBlockInds[:, i] - 1
                temp[idx] += block.solve(B[idx] if j is None else B[idx, j])
            if j is None:
                Y[:] = temp / self.mCountingMatrix
            else:
                Y[:, j] = temp / self.mCountingMatrix
        return alpha * Y


def apply_smoother(S, B, alpha=1.0):
    """apply_smoother(A::AbstractSmoother, B; alpha=1.0) src/smoother.jl:6,30,56,69"""
    return S.apply(B, alpha=alpha)


def _element_blocks(mesh_, A):
    A = A.tocsc()
    n = len(mesh_.mElements)
    p = mesh_.mP
    blocks = []
    blockInds = np.zeros((p + 1, n), dtype=np.int64)
    for i, el in enumerate(mesh_.mElements):
        idx = np.array(el.mNodesInd) - 1
        blocks.append(LU(A[np.ix_(idx, idx)].toarray()))
        blockInds[:, i] = el.mNodesInd
    return blocks, blockInds


def cg_smoother(cgMesh, A, smootherType):
    """src/smoother.jl:88-139"""
    if smootherType == 'jac':
        return JacobiSmoother(A.diagonal())
    if smootherType == 'addSchwarz':
        return AdditiveSchwarzSmoother(*_element_blocks(cgMesh, A))
    if smootherType == 'hybridSchwarz':
        blocks, inds = _element_blocks(cgMesh, A)
        counting = np.zeros(A.shape[0])
        for el in cgMesh.mElements:
            for l in el.mNodesInd:
                counting[l - 1] += 1.0
        return HybridSchwarzSmoother(blocks, inds, counting)
    raise ValueError(smootherType)


def dg_smoother(dgMesh, A, smootherType):
    """src/smoother.jl:142-168"""
    if smootherType == 'jac':
        return JacobiSmoother(A.diagonal())
    if smootherType == 'blockJac':
        return BlockJacobi(*_element_blocks(dgMesh, A))
    if smootherType == 'blockGS':    # EXTENSION, see BlockGaussSeidelRB
        return BlockGaussSeidelRB(*_element_blocks(dgMesh, A))
    raise ValueError(smootherType)


# --------------------------------------------------------------------------------------
# src/mesh_heirarchy.jl
# --------------------------------------------------------------------------------------


class MeshHierarchy:
    """src/mesh_heirarchy.jl:17-28"""

    def __init__(self, mMeshes, mStiffness, mGradient, mDivergence, mC, mSmoothers,
                 mInterpolation, mBdConds):
        self.mMeshes = mMeshes
        self.mStiffness = mStiffness
        self.mGradient = mGradient
        self.mDivergence = mDivergence
        self.mC = mC
        self.mSmoothers = mSmoothers
        self.mInterpolation = mInterpolation
        self.mBdConds = mBdConds


def _galerkin(L, X):
    return (L.T @ X @ L).tocsc()


def _galerkin_level(L, Gp, Dp, Cp, m):
    """src/mesh_heirarchy.jl:79-85 (and :98-105, :122-130): Galerkin products on G, D, C
    separately, then A = C - D (M_LU \\ G) with the level's own mass matrix."""
    G, D, C = _galerkin(L, Gp), _galerkin(L, Dp), _galerkin(L, Cp)
    A = dg_stiffness(m, G, D, C)
    return G, D, C, A, dg_smoother(m, A, 'blockJac')


def MeshHierarchy_cg(mMeshes, mesh, mBdConds, A, nCG=1, nDG=0, nAgg=0, CDir=1.0):
    """MeshHierarchy(mMeshes, mesh, mBdConds, A; nCG, nDG, nAgg, CDir)
    src/mesh_heirarchy.jl:30-138 (CG-fine constructor)."""
    if nCG <= 0:
        raise ValueError("At least one CG mesh required.")
    if len(mMeshes) != nCG + nDG + nAgg:
        raise ValueError("Length of vector of meshes does not match inputed number of CG, "
                         "DG, and agglomerated meshes.")
    nl = nCG + nDG + nAgg
    St, Sm, Li = [None] * nl, [None] * nl, [None] * (nl - 1)
    Gs, Ds, Cs = [None] * (nDG + nAgg), [None] * (nDG + nAgg), [None] * (nDG + nAgg)
    St[0] = A.tocsc()
    Sm[0] = cg_smoother(mMeshes[0], St[0], 'jac')
    for i in range(nCG - 1):
        L = cg_cg_interpolation(mMeshes[i + 1], mMeshes[i])
        Li[i] = L
        St[i + 1] = _galerkin(L, St[i])
        Sm[i + 1] = cg_smoother(mMeshes[i + 1], St[i + 1], 'jac')
    if nDG >= 1:
        m = mMeshes[nCG]
        Li[nCG - 1] = dg_cg_interpolation(m, mMeshes[nCG - 1], mesh, 1)
        Gs[0], Ds[0], Cs[0] = dg_flux_operators(m, mesh, mBdConds[nCG], CDir)
        St[nCG] = dg_stiffness(m, Gs[0], Ds[0], Cs[0])
        Sm[nCG] = dg_smoother(m, St[nCG], 'blockJac')
        for i in range(1, nDG):
            L = dg_dg_interpolation(mMeshes[nCG + i], mMeshes[nCG + i - 1])
            Li[nCG + i - 1] = L
            Gs[i], Ds[i], Cs[i], St[nCG + i], Sm[nCG + i] = _galerkin_level(
                L, Gs[i - 1], Ds[i - 1], Cs[i - 1], mMeshes[nCG + i])
        for i in range(nAgg):
            if i == 0:
                L = aggdg_dg_interpolation(mMeshes[nCG + nDG], mMeshes[nCG + nDG - 1])
            else:
                L = aggdg_aggdg_interpolation(mMeshes[nCG + nDG + i], mMeshes[nCG + nDG + i - 1],
                                              mMeshes[nCG + nDG - 1])
            Li[nCG + nDG + i - 1] = L
            k = nDG + i
            Gs[k], Ds[k], Cs[k], St[nCG + k], Sm[nCG + k] = _galerkin_level(
                L, Gs[k - 1], Ds[k - 1], Cs[k - 1], mMeshes[nCG + k])
    elif nAgg >= 1:
        m = mMeshes[nCG]
        Li[nCG - 1] = aggdg_cg_interpolation(m, mMeshes[nCG - 1], mesh, 1)
        Gs[0], Ds[0], Cs[0] = dg_flux_operators(m, mMeshes[nCG - 1], mBdConds[nCG], CDir)
        St[nCG] = dg_stiffness(m, Gs[0], Ds[0], Cs[0])
        Sm[nCG] = dg_smoother(m, St[nCG], 'blockJac')
        for i in range(1, nAgg):
            L = aggdg_aggdg_interpolation(mMeshes[nCG + i], mMeshes[nCG + i - 1],
                                          mMeshes[nCG - 1])
            Li[nCG + i - 1] = L
            Gs[i], Ds[i], Cs[i], St[nCG + i], Sm[nCG + i] = _galerkin_level(
                L, Gs[i - 1], Ds[i - 1], Cs[i - 1], mMeshes[nCG + i])
    return MeshHierarchy(mMeshes, St, Gs, Ds, Cs, Sm, Li, mBdConds)


def MeshHierarchy_dg(mMeshes, mBdConds, A, G, D, C, nDG=1, nAgg=0):
    """MeshHierarchy(mMeshes, mBdConds, A, G, D, C; nDG, nAgg) src/mesh_heirarchy.jl:140-181
    (DG-fine constructor).  The reference sizes its vectors nDG+nAgg but never fills the
    agglomerated slots.  EXTENSION (labelled, no reference counterpart): the
    missing nAgg loop is added here with exactly the recurrence of the CG-fine constructor,
    src/mesh_heirarchy.jl:89-106."""
    if nDG <= 0:
        raise ValueError("At least one DG mesh required.")
    if len(mMeshes) != nDG + nAgg:
        raise ValueError("Length of vector of meshes does not match inputed number of DG and "
                         "agglomerated meshes.")
    nl = nDG + nAgg
    St, Sm, Li = [None] * nl, [None] * nl, [None] * (nl - 1)
    Gs, Ds, Cs = [None] * nl, [None] * nl, [None] * nl
    Gs[0], Ds[0], Cs[0] = G.tocsc(), D.tocsc(), C.tocsc()
    St[0] = A.tocsc()
    Sm[0] = dg_smoother(mMeshes[0], St[0], 'blockJac')
    for i in range(1, nDG):
        L = dg_dg_interpolation(mMeshes[i], mMeshes[i - 1])
        Li[i - 1] = L
        Gs[i], Ds[i], Cs[i], St[i], Sm[i] = _galerkin_level(
            L, Gs[i - 1], Ds[i - 1], Cs[i - 1], mMeshes[i])
    for i in range(nAgg):  # EXTENSION, see docstring
        if i == 0:
            L = aggdg_dg_interpolation(mMeshes[nDG], mMeshes[nDG - 1])
        else:
            L = aggdg_aggdg_interpolation(mMeshes[nDG + i], mMeshes[nDG + i - 1],
                                          mMeshes[nDG - 1])
        k = nDG + i
        Li[k - 1] = L
        Gs[k], Ds[k], Cs[k], St[k], Sm[k] = _galerkin_level(
            L, Gs[k - 1], Ds[k - 1], Cs[k - 1], mMeshes[k])
    return MeshHierarchy(mMeshes, St, Gs, Ds, Cs, Sm, Li, mBdConds)


# --------------------------------------------------------------------------------------
# src/solvers.jl  -- THE hot path
# --------------------------------------------------------------------------------------


def sparse_direct_solve(A, b):
    """`A \\ b` for SparseMatrixCSC (UMFPACK in the reference, src/solvers.jl:39,120,194).
    UMFPACK is a Julia-stdlib dependency absent here; SciPy's SuperLU sparse LU is the
    stand-in: a backward-stable direct solve, equal to UMFPACK's up to cond(A)*eps."""
    return spla.spsolve(A.tocsc(), np.asarray(b, dtype=np.float64))


def multigrid_v_cycle(H, x0, b, nPre=3, nPost=3, alpha=2.0 / 3.0, coarse_solve=None):
    """src/solvers.jl:19-50, operation for operation:
    t = A*u; r = rhs - t; y = S \\ r; z = alpha*y; u = u + z."""
    solve = coarse_solve or sparse_direct_solve
    n = len(H.mMeshes)
    u = [None] * n
    rhs = [None] * n
    u[0] = np.asarray(x0, dtype=np.float64)
    rhs[0] = np.asarray(b, dtype=np.float64)
    for k in range(n - 1):
        if k > 0:
            u[k] = np.zeros(H.mStiffness[k].shape[1])
        for _ in range(nPre):
            u[k] = smooth_once(H.mSmoothers[k], H.mStiffness[k], u[k], rhs[k], alpha)
        rhs[k + 1] = csc_adjoint_matvec(H.mInterpolation[k],
                                        rhs[k] - csc_matvec(H.mStiffness[k], u[k]))
    u[n - 1] = solve(H.mStiffness[n - 1], rhs[n - 1])
    for k in range(n - 2, -1, -1):
        u[k] = u[k] + csc_matvec(H.mInterpolation[k], u[k + 1])
        for _ in range(nPost):
            u[k] = smooth_once(H.mSmoothers[k], H.mStiffness[k], u[k], rhs[k], alpha, post=True)
    return u[0]


def ldiv(H, b, y=None):
    """ldiv!(H, b) src/solvers.jl:63-71 (overwrites b) and ldiv!(y, H, b) :84-92."""
    u0 = np.zeros(H.mStiffness[0].shape[0])
    out = multigrid_v_cycle(H, u0, b)
    if y is None:
        b[:] = out
    else:
        y[:] = out


def multigrid(H, x0, b, maxiter, tol):
    """src/solvers.jl:116-139 -> (x, iter, res, err)"""
    x = np.zeros(len(x0))
    u_exact = sparse_direct_solve(H.mStiffness[0], b)
    err, res = [], []
    for i in range(maxiter):
        x = multigrid_v_cycle(H, x0, b)
        x0 = x
        err.append(np.linalg.norm(x - u_exact, 2))
        res.append(np.linalg.norm(csc_matvec(H.mStiffness[0], x) - b, 2))
        if res[i] < tol * np.linalg.norm(b, 2):
            break
    return x, len(res), res, err


def iterative_smoother_solve(A, smoother, x0, b, maxiter=1000, tol=1e-6, alpha=1.0):
    """src/solvers.jl:189-213 -> (x, iter, res, err)"""
    x = np.zeros(len(x0))
    uExact = sparse_direct_solve(A, b)
    err, res = [], []
    for i in range(maxiter):
        x = x0 + apply_smoother(smoother, b - csc_matvec(A, x0), alpha=alpha)
        x0 = x
        err.append(np.linalg.norm(x - uExact, 2))
        res.append(np.linalg.norm(csc_matvec(A, x) - b, 2))
        if res[i] < tol * np.linalg.norm(b, 2):
            break
    return x, len(res), res, err


def pcg_ldiv(H, b, x0=None, maxiter=50, tol=1e-10, nPre=3, nPost=3, alpha=2.0 / 3.0):
    """Conjugate gradients preconditioned with ldiv!(y, H, r) (src/solvers.jl:84-92).
    EXTENSION: the reference has no Krylov loop; this restates the textbook
    recurrence the device path (aggmg_pcg_dev) follows, operation for operation, so that the
    two can be compared.  -> (x, iter, res) with res[i] = ||r_i|| of the recurrence."""
    A = H.mStiffness[0]
    N = A.shape[0]
    x = np.zeros(N) if x0 is None else np.array(x0, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    zero = np.zeros(N)
    nb = np.linalg.norm(b, 2)
    r = b - csc_matvec(A, x)
    z = multigrid_v_cycle(H, zero, r, nPre, nPost, alpha)
    p = z.copy()
    rz = float(r @ z)
    res = []
    for _ in range(maxiter):
        q = -csc_matvec(A, p)
        a = rz / (-(float(p @ q)))
        x = x + a * p
        r = r + a * q
        res.append(np.linalg.norm(r, 2))
        if res[-1] < tol * nb:
            break
        z = multigrid_v_cycle(H, zero, r, nPre, nPost, alpha)
        rz_new = float(r @ z)
        p = z + (rz_new / rz) * p
        rz = rz_new
    return x, len(res), res


# --------------------------------------------------------------------------------------
# Problem builders in the shape of the reference's test scripts (harness rows)
# --------------------------------------------------------------------------------------


def model_problem(n, xin=0.0, xout=1.0, bc=None):
    """tests/dg_heirarchy_test.jl:7-24: uniform mesh on [0,1], left Neumann -sin(xin), right
    Dirichlet cos(xout), f = cos."""
    mesh = create_uniform_mesh(n, xin, xout)
    if bc is None:
        bc = [('neu', -math.sin(xin)), ('dir', math.cos(xout))]
    bdCond = set_boundary(mesh, xin, xout, bc)
    return mesh, bdCond


def uniform_agglomerations(n, nAgg, first=4):
    """tests/full_heirarchy_test.jl:63-90: first level `first`:1 then 2:1."""
    aggs = []
    tempN = n // first
    for i in range(nAgg):
        if i == 0:
            aggs.append([list(range(first * j - first + 1, first * j + 1))
                         for j in range(1, tempN + 1)])
        else:
            aggs.append([[2 * j - 1, 2 * j] for j in range(1, tempN + 1)])
        tempN //= 2
    return aggs


def build_dg_agg_hierarchy(n, p=3, pAgg=1, nAgg=3, first=4, CDir=None, func=math.cos):
    """BASELINE config 3 shape: DgMesh(p) -> AggDgMesh1 (first:1) -> AggDgMeshN (2:1) ...
    built with the labelled D4 extension of the DG-fine constructor.  Returns (H, b)."""
    CDir = 1000.0 * n if CDir is None else CDir
    mesh, bdCond = model_problem(n)
    dg = DgMesh(mesh, p)
    meshes = [dg]
    for i, agg in enumerate(uniform_agglomerations(n, nAgg, first)):
        if i == 0:
            meshes.append(AgglomeratedDgMesh1(pAgg, agg, mesh, dg))
        else:
            meshes.append(AgglomeratedDgMeshN(pAgg, agg, meshes[-1], dg))
    G, D, C = dg_flux_operators(dg, mesh, bdCond, CDir)
    A = dg_stiffness(dg, G, D, C)
    f, r = dg_flux_rhs(dg, mesh, func, bdCond, CDir)
    b = dg_rhs(dg, D, f, r)
    H = MeshHierarchy_dg(meshes, [bdCond] * len(meshes), A, G, D, C, nDG=1, nAgg=nAgg)
    return H, b


def build_dg_p_hierarchy(n, ps=(8, 4, 2, 1), CDir=None, func=math.cos):
    """tests/dg_heirarchy_test.jl:11-50 shape (p-coarsened DG hierarchy)."""
    CDir = 1000.0 * n if CDir is None else CDir
    mesh, bdCond = model_problem(n)
    meshes = [DgMesh(mesh, p) for p in ps]
    G, D, C = dg_flux_operators(meshes[0], mesh, bdCond, CDir)
    A = dg_stiffness(meshes[0], G, D, C)
    f, r = dg_flux_rhs(meshes[0], mesh, func, bdCond, CDir)
    b = dg_rhs(meshes[0], D, f, r)
    H = MeshHierarchy_dg(meshes, [bdCond] * len(meshes), A, G, D, C, nDG=len(ps))
    return H, b


def build_cg_hierarchy(n, ps=(8, 4, 2, 1), nDG=0, pDG=None, nAgg=0, pAgg=1, first=4, CDir=None,
                       func=math.cos):
    """tests/cg_heirarchy_test.jl, dg_cg_heirarchy_test.jl:14-50 and full_heirarchy_test.jl
    :13-96 shapes: CG p-chain, optional DG levels below it (p halved per level) and optional
    agglomerated levels."""
    CDir = 1000.0 * n if CDir is None else CDir
    mesh, bdCond = model_problem(n)
    meshes = [CgMesh(mesh, p) for p in ps]
    tempP = ps[-1] // 2 if pDG is None else pDG
    for _ in range(nDG):
        meshes.append(DgMesh(mesh, tempP))
        tempP //= 2
    nCG = len(ps)
    for i, agg in enumerate(uniform_agglomerations(n, nAgg, first)):
        if i == 0:
            meshes.append(AgglomeratedDgMesh1(pAgg, agg, mesh, meshes[0]))
        else:
            meshes.append(AgglomeratedDgMeshN(pAgg, agg, meshes[-1], meshes[0]))
    A, b = cg_stiffness_and_rhs(meshes[0], mesh, func, bdCond)
    H = MeshHierarchy_cg(meshes, mesh, [bdCond] * len(meshes), A, nCG=nCG, nDG=nDG, nAgg=nAgg,
                         CDir=CDir)
    return H, b


# --------------------------------------------------------------------------------------
# Counter-based N(0,1) generator shared by oracle, product tests and bench
# --------------------------------------------------------------------------------------


def splitmix_normal(n, seed):
    """SplitMix64 -> Box-Muller; bit-identical wherever IEEE doubles and libm log/cos agree to
    the last place is NOT guaranteed, so fixtures carry the generated vectors themselves."""
    idx = np.arange(2 * n, dtype=np.uint64)
    with np.errstate(over='ignore'):
        z = (idx + np.uint64(seed) * np.uint64(0x632BE59BD9B4E019)) * np.uint64(0x9E3779B97F4A7C15) \
            + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = ((z >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)
    return np.sqrt(-2.0 * np.log(u[0::2])) * np.cos(2.0 * np.pi * u[1::2])
