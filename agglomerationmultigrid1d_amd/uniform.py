"""O(n) set-up of the operators the hot path consumes, for the uniform-mesh model problem.

The reference assembles `G, D, C`, `A = C - D*(M\\G)`, the agglomerated levels and their
Galerkin operators through a pointer-y object graph with O(n^2) traps, which
cannot build the 2^20..2^24-element configurations of BASELINE.json.  This module produces the
same matrices with vectorised NumPy on block-banded arrays:

    dg_flux_operators(::DgMesh)        src/dg_mesh.jl:144-336      -> DgLevel
    dg_flux_rhs(::DgMesh)              src/dg_mesh.jl:342-457      -> DgLevel.rhs()
    A = C - D*(M_LU \\ G)               src/mesh_heirarchy.jl:71-72 -> _stiffness_blocks
    aggdg_dg_interpolation             src/interpolation.jl:270-292
    aggdg_aggdg_interpolation          src/interpolation.jl:226-264
    Galerkin L'XL on G, D, C           src/mesh_heirarchy.jl:98-103
    agglomerated mass matrices         src/agglomerated_dg_mesh.jl:444-461,611-628

Values agree with the loop-for-loop CPU restatement at small n to round-off and the transfer
index maps (colptr / rowval of every L) agree bit for bit -- tests/test_uniform_setup.py.
Only the shapes the BASELINE configurations need are covered: a nodal DG fine level (any p >= 0)
followed by agglomerated levels (pAgg in {0,1}) with uniform ratios.  It is set-up code: the hot
path itself never runs here.
"""
import math
import os

import numpy as np
import scipy.sparse as sp


# ------------------------------------------------------------------------------------------
# reference-element numerics (src/legendre.jl, src/gauss_quad.jl, src/reference_element.jl)
# ------------------------------------------------------------------------------------------
def legendre_vandermonde(x, p, deriv=False):
    """Values (and derivatives) of P_0..P_p at points x by the three-term recurrence
    (src/legendre.jl:14-25,44-58).  -> (len(x), p+1)"""
    x = np.atleast_1d(np.asarray(x, dtype=np.float64))
    F = np.zeros((len(x), p + 1))
    Dv = np.zeros((len(x), p + 1))
    F[:, 0] = 1.0
    if p >= 1:
        F[:, 1] = x
        Dv[:, 1] = 1.0
    for i in range(2, p + 1):
        F[:, i] = ((2 * i - 1) * x * F[:, i - 1] - (i - 1) * F[:, i - 2]) / i
        Dv[:, i] = (2 * i - 1) * F[:, i - 1] + Dv[:, i - 2]
    return (F, Dv) if deriv else F


def gauss_quad(p):
    """Golub-Welsch rule exact to degree p (src/gauss_quad.jl:6-13)."""
    n = int(math.ceil((p + 1) / 2))
    k = np.arange(1, n, dtype=np.float64)
    off = k / np.sqrt(4 * k**2 - 1)
    ev, evec = np.linalg.eigh(np.diag(off, 1) + np.diag(off, -1))
    return ev, 2 * evec[0, :]**2


class RefElement:
    """src/reference_element.jl:15-57: nodes [-1, 1, cos(pi i/p)], nodal basis = inv(Vandermonde),
    Gauss rule of degree 2p, reference mass matrix."""

    def __init__(self, p):
        self.p = p
        x = np.zeros(p + 1)
        if p >= 1:
            x[:2] = [-1.0, 1.0]
            x[2:] = np.cos(np.pi * np.arange(1, p) / p)
        self.nodes = x
        self.coeff = np.linalg.inv(legendre_vandermonde(x, p))
        self.gq, self.gw = gauss_quad(2 * p)
        F, Dv = legendre_vandermonde(self.gq, p, deriv=True)
        nq = len(self.gq)
        # basis values / derivatives at the Gauss points: one dot product per (point, function),
        # as src/reference_element.jl:75-90 does
        self.phi = np.array([[np.dot(self.coeff[:, i], F[l]) for i in range(p + 1)] for l in range(nq)])
        self.dphi = np.array([[np.dot(self.coeff[:, i], Dv[l]) for i in range(p + 1)] for l in range(nq)])
        M = np.zeros((p + 1, p + 1))
        for j in range(p + 1):
            for i in range(j + 1):
                for l in range(nq):
                    M[i, j] += self.gw[l] * self.phi[l, i] * self.phi[l, j]
        self.mass = np.triu(M) + np.triu(M, 1).T

    def basis_at(self, xi):
        return legendre_vandermonde(xi, self.p) @ self.coeff


# ------------------------------------------------------------------------------------------
# element-range parallelism of the generators: NumPy releases the GIL inside its loops, so disjoint
# element ranges are built by a few threads and concatenated (bit for bit the serial result)
# ------------------------------------------------------------------------------------------
_PAR_MIN_ELEMS = 1 << 18


def _gen_workers():
    env = os.environ.get("AGGMG_GEN_WORKERS")
    if env:
        return max(1, int(env))
    try:
        return max(1, min(16, len(os.sched_getaffinity(0))))
    except AttributeError:
        return max(1, min(16, os.cpu_count() or 1))


def _pool_map(fn, items, workers):
    if workers <= 1 or len(items) <= 1:
        return [fn(x) for x in items]
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(workers) as ex:
        return list(ex.map(fn, items))


def _par_concat(parts, workers):
    """np.concatenate of 1-D pieces with the copies spread over threads"""
    if workers <= 1 or sum(q.size for q in parts) < (1 << 22):
        return np.concatenate(parts)
    out = np.empty(sum(q.size for q in parts), dtype=parts[0].dtype)
    jobs, o = [], 0
    step = 1 << 24
    for q in parts:
        for a in range(0, q.size, step):
            jobs.append((o + a, q, a, min(a + step, q.size)))
        o += q.size

    def put(j):
        out[j[0]:j[0] + j[3] - j[2]] = j[1][j[2]:j[3]]

    _pool_map(put, jobs, workers)
    return out


def _ranges(n, parts, align=1):
    """<= parts contiguous ranges of [0, n) with boundaries at multiples of align"""
    units = n // align
    parts = max(1, min(parts, units))
    cuts = [(units * i // parts) * align for i in range(parts)] + [n]
    return [(cuts[i], cuts[i + 1]) for i in range(parts) if cuts[i + 1] > cuts[i]]


# ------------------------------------------------------------------------------------------
# block-banded helpers
# ------------------------------------------------------------------------------------------
def _shift_up(X):
    """Y[k] = X[k+1], zero at the end"""
    Y = np.zeros_like(X)
    Y[:-1] = X[1:]
    return Y


def _shift_down(X):
    """Y[k] = X[k-1], zero at the start"""
    Y = np.zeros_like(X)
    Y[1:] = X[:-1]
    return Y


def _lt_x_l(Lrow, X, Lcol):
    """batched Lrow[k]' @ X[k] @ Lcol[k]"""
    return np.matmul(np.matmul(Lrow.transpose(0, 2, 1), X), Lcol)


def block_tridiag_to_csc(sub, diag, sup, keep_zeros=False, workers=None):
    """CSC (0-based int64 colptr/rowval, values) of the block-tridiagonal matrix with blocks
    sub[k] = (k, k-1), diag[k] = (k, k), sup[k] = (k, k+1), keeping the numerically non-zero
    entries (the pattern `C - D*X` has in SparseArrays) in ascending row order
    per column."""
    ne, m, _ = diag.shape
    if workers is None:
        workers = _gen_workers() if ne >= _PAR_MIN_ELEMS else 1

    def part(rng):
        e0, e1 = rng
        k = e1 - e0
        # column strip of element k: rows of elements k-1 (sup[k-1]), k (diag[k]), k+1 (sub[k+1])
        strip = np.zeros((k, m, 3 * m))
        strip[:, :, m:2 * m] = diag[e0:e1].transpose(0, 2, 1)
        lo = max(e0, 1)
        strip[lo - e0:, :, 0:m] = sup[lo - 1:e1 - 1].transpose(0, 2, 1)
        hi = min(e1, ne - 1)
        strip[:hi - e0, :, 2 * m:] = sub[e0 + 1:hi + 1].transpose(0, 2, 1)
        mask = np.ones(strip.shape, dtype=bool) if keep_zeros else (strip != 0.0)
        if e0 == 0:
            mask[0, :, 0:m] = False
        if e1 == ne:
            mask[-1, :, 2 * m:] = False
        rows = (np.arange(e0, e1, dtype=np.int64)[:, None, None] - 1) * m + np.arange(3 * m, dtype=np.int64)[None, None, :]
        rows = np.broadcast_to(rows, strip.shape)
        return mask.sum(axis=2).reshape(-1), rows[mask], strip[mask]

    rngs = _ranges(ne, 4 * workers if workers > 1 else 1)
    parts = _pool_map(part, rngs, workers)
    colptr = np.zeros(ne * m + 1, dtype=np.int64)
    if len(parts) == 1:
        np.cumsum(parts[0][0], out=colptr[1:])
        return colptr, parts[0][1], parts[0][2], ne * m
    for (e0, e1), q in zip(rngs, parts):
        colptr[e0 * m + 1:e1 * m + 1] = q[0]
    np.cumsum(colptr, out=colptr)
    nnz = int(colptr[-1])
    rowval, nzval = np.empty(nnz, dtype=np.int64), np.empty(nnz)

    def put(i):      # every range's entries straight to their place (the concatenation, in threads)
        o = int(colptr[rngs[i][0] * m])
        rowval[o:o + parts[i][1].size] = parts[i][1]
        nzval[o:o + parts[i][2].size] = parts[i][2]

    _pool_map(put, list(range(len(parts))), workers)
    return colptr, rowval, nzval, ne * m


def _csc(colptr, rowval, nzval, shape):
    return sp.csc_matrix((nzval, rowval, colptr), shape=shape)


class LevelDescriptor:
    """What the hot path reads from a reference mesh object: `mP` and the element index lists
    (`mBlockInds`, (p+1) x n, 1-based, contiguous: (k-1)(p+1)+1 .. k(p+1), src/dg_mesh.jl:44,
    src/agglomerated_dg_mesh.jl:89,505)."""

    def __init__(self, mP, ne):
        self.mP = mP
        self.ne = ne

    @property
    def mBlockInds(self):
        m = self.mP + 1
        return (np.arange(self.ne, dtype=np.int64)[None, :] * m + np.arange(1, m + 1, dtype=np.int64)[:, None])


# ------------------------------------------------------------------------------------------
# the hierarchy builder
# ------------------------------------------------------------------------------------------
class UniformDgAggHierarchy:
    """DgMesh(p) on a uniform n-element mesh of [xin, xout] followed by agglomerated-DG levels
    (first level `ratios[0]`:1 base elements, then `ratios[i]`:1 agglomerates), operators by
    Galerkin products on G, D, C with `A = C - D(M\\G)` per level -- the D4 extension of the
    DG-fine constructor (recurrence of src/mesh_heirarchy.jl:89-106).

    Boundary conditions as in tests/dg_heirarchy_test.jl: bc = ((kind, value), (kind, value)),
    kind in {'neu', 'dir'}; default left Neumann -sin(xin), right Dirichlet cos(xout), f = cos,
    CDir = 1000 n."""

    def __init__(self, n, p=3, pAgg=1, ratios=(4, 2, 2), CDir=None, xin=0.0, xout=1.0, bc=None,
                 func=np.cos, elem_range=None, workers=None, _left_pad=False):
        """elem_range=(a, b): build only the rows/columns of fine elements a..b-1 (0-based, multiples
        of prod(ratios)) and of their agglomerates -- the local operators of one rank of an
        element-partitioned run.  Couplings to elements outside the range are dropped; everything
        else equals the corresponding entries of the global operators.

        workers: large meshes are built range by range in that many threads and concatenated
        (default: the CPUs of the process, at most 16; bit for bit the serial result).
        _left_pad (internal, the ranges of such a build): one coarsest-level element of padding on
        the left as well, so that the block arrays also hold the couplings of the first agglomerates
        to their left neighbours."""
        if pAgg not in (0, 1):
            raise ValueError("Only implemented for p = 0 and p = 1.")  # agglomerated_dg_mesh.jl:312
        if p < 1 and len(ratios):
            # AgglomeratedDgElement1 reads baseMesh.mElements[k].mNodesX[2]
            # (src/agglomerated_dg_mesh.jl:183-189: "assumes the baseMesh has at least p >= 1")
            raise ValueError("agglomeration needs a base mesh with p >= 1")
        tot = int(np.prod(ratios)) if len(ratios) else 1
        if n % tot:
            raise ValueError("n must be divisible by the product of the agglomeration ratios")
        a, b = (0, n) if elem_range is None else (int(elem_range[0]), int(elem_range[1]))
        if not (0 <= a < b <= n) or a % tot or b % tot:
            raise ValueError("elem_range must be a non-empty range aligned to prod(ratios)")
        self.n_global, self.a, self.b = n, a, b
        self.at_left, self.at_right = (a == 0), (b == n)
        # one coarsest-level element of padding on the right: the last local rows need their right
        # neighbour's mass matrix / gradient blocks (A = C - D*(M\G) couples k to k+1)
        pad = 0 if self.at_right else tot
        self.padL = tot if (_left_pad and not self.at_left) else 0
        # whether the ARRAYS (owned range + padding) end at the domain boundary: the padding elements carry
        # their true blocks then (the coupling blocks of the owned elements next to them depend on that)
        self.end_left, self.end_right = (a - self.padL == 0), (b + pad == n)
        self.nloc = b - a
        self.n = self.nloc + pad + self.padL   # elements the arrays are built on (padding trimmed later)
        self.p, self.pAgg, self.ratios = p, pAgg, tuple(ratios)
        self.CDir = 1000.0 * n if CDir is None else float(CDir)
        self.bc = bc or (('neu', -math.sin(xin)), ('dir', math.cos(xout)))
        self.func = func
        self.ref = RefElement(p)
        if not _left_pad:      # (a range of such a build is built in one piece)
            workers = _gen_workers() if workers is None else int(workers)
            rngs = [(a + r0, a + r1) for r0, r1 in _ranges(b - a, 4 * workers if b - a >= (1 << 22) else workers, tot)]
            if len(rngs) > 1 and (workers > 1) and (b - a >= _PAR_MIN_ELEMS or os.environ.get("AGGMG_GEN_FORCE_PARALLEL")):
                self._build_by_ranges(rngs, workers, dict(p=p, pAgg=pAgg, ratios=ratios, CDir=self.CDir, xin=xin,
                                                          xout=xout, bc=self.bc, func=func))
                return
        a0 = a - self.padL
        i = np.arange(a0, a0 + self.n + 1, dtype=np.float64)
        self.xv = xin + (i / n) * (xout - xin)              # tests/mesh_generator.jl:11-13
        if a0 == 0:
            self.xv[0] = xin
        self.h = self.xv[1:] - self.xv[:-1]
        self.xc = (self.xv[:-1] + self.xv[1:]) / 2.0
        self.J = self.h / 2.0
        self._build_fine()
        self.levels = [dict(m=p + 1, ne=self.n, G=(self.Gl, self.Gd), D=(self.Dd, self.Du), C=self.Cd,
                            M=self.M, A=self._stiffness_blocks(self.Gl, self.Gd, self.Dd, self.Du,
                                                               self.Cd, self.M, self.Minv))]
        self.transfers = []     # Lb[k]: (ne_f, m_f, m_c) rows of L per fine element, rho
        self._build_agglomerated()
        self._rhs_full = self._rhs_padded()
        self._trim()

    def _trim(self):
        """drop the padding from every level"""
        per = 1
        ne = self.nloc
        off = self.padL
        for k, lv in enumerate(self.levels):
            if k > 0:
                per = self.ratios[k - 1]
                ne //= per
                off //= per
            sl = slice(off, off + ne)
            lv['ne'] = ne
            lv['A'] = tuple(x[sl] for x in lv['A'])
            lv['M'] = lv['M'][sl]
            lv['C'] = lv['C'][sl]
            lv['G'] = tuple(x[sl] for x in lv['G'])
            lv['D'] = tuple(x[sl] for x in lv['D'])
            if k < len(self.transfers):
                self.transfers[k]['Lb'] = self.transfers[k]['Lb'][sl]
        m = self.p + 1
        self._rhs_full = self._rhs_full[self.padL * m:(self.padL + self.nloc) * m]
        self.n = self.nloc
        self.padL = 0

    def _build_by_ranges(self, rngs, workers, kw):
        """the whole mesh range by range (threads); every worker writes its slice of the stiffness blocks,
        the interpolation rows and the right-hand side (the intermediate G, D, C, M blocks stay range-local)"""
        n, a = self.n_global, self.a        # the ranges lie in [a, b): the owned elements of this build
        m = [self.p + 1] + [self.pAgg + 1] * len(self.ratios)
        ne = [self.nloc]
        for rho in self.ratios:
            ne.append(ne[-1] // rho)
        A = [tuple(np.empty((ne[k], m[k], m[k])) for _ in range(3)) for k in range(len(m))]
        Lb = [np.empty((ne[k], m[k], m[k + 1])) for k in range(len(self.ratios))]
        rhs = np.empty(self.nloc * m[0])

        def work(r):
            q = UniformDgAggHierarchy(n, elem_range=r, _left_pad=True, **kw)
            per = 1
            for k in range(len(m)):
                if k > 0:
                    per *= self.ratios[k - 1]
                sl = slice((r[0] - a) // per, (r[1] - a) // per)
                for i in range(3):
                    A[k][i][sl] = q.levels[k]['A'][i]
                if k < len(Lb):
                    Lb[k][sl] = q.transfers[k]['Lb']
            rhs[(r[0] - a) * m[0]:(r[1] - a) * m[0]] = q._rhs_full
            return None

        _pool_map(work, rngs, workers)
        self.levels = [dict(m=m[k], ne=ne[k], G=None, D=None, C=None, M=None, A=A[k]) for k in range(len(m))]
        self.transfers = [dict(Lb=Lb[k], rho=rho, mc=self.pAgg + 1) for k, rho in enumerate(self.ratios)]
        self._rhs_full = rhs
        self.n = self.nloc
        self.padL = 0

    # ---- fine DG level -----------------------------------------------------------------------
    def _build_fine(self):
        n, p, ref = self.n, self.p, self.ref
        m = p + 1
        nL, nR = (0, 1) if p >= 1 else (0, 0)
        V = np.zeros((m, m))
        if p >= 1:  # volume term, accumulated over the Gauss points in loop order
            for l in range(len(ref.gw)):
                V += (ref.gw[l] * ref.dphi[l])[:, None] * ref.phi[l][None, :]
        Gd = np.broadcast_to(V, (n, m, m)).copy()
        Dd = np.broadcast_to(V, (n, m, m)).copy()
        Gl = np.zeros((n, m, m))
        Du = np.zeros((n, m, m))
        Cd = np.zeros((n, m, m))
        lk, rk = self.bc[0][0], self.bc[1][0]
        # interior vertices: uhat = u_L, qhat = q_R (src/dg_mesh.jl:219-249)
        Gd[:-1, nR, nR] += -1.0       # right end of element k, uhat = own value
        Gl[1:, nL, nR] += 1.0         # left end of element k, uhat from element k-1
        Dd[1:, nL, nL] += 1.0         # left end, qhat = own q
        Du[:-1, nR, nL] += -1.0       # right end, qhat from element k+1
        # first / last element of the array: domain boundary (src/dg_mesh.jl:170-218) or, for a
        # partial element range, one more interior vertex
        if not self.end_left:
            Gl[0, nL, nR] += 1.0
            Dd[0, nL, nL] += 1.0
        elif lk == 'dir':
            Dd[0, nL, nL] += 1.0
            Cd[0, nL, nL] += self.CDir
        else:
            Gd[0, nL, nL] += 1.0
        if not self.end_right:
            Gd[-1, nR, nR] += -1.0
            Du[-1, nR, nL] += -1.0
        elif rk == 'dir':
            Dd[-1, nR, nR] += -1.0
            Cd[-1, nR, nR] += self.CDir
        else:
            Gd[-1, nR, nR] += -1.0
        self.Gd, self.Gl, self.Dd, self.Du, self.Cd = Gd, Gl, Dd, Du, Cd
        self.M = self.J[:, None, None] * ref.mass[None, :, :]
        # mass blocks are J_e * M_ref (src/dg_mesh.jl:69-79): their inverses cost one small inversion
        self.Minv = (1.0 / self.J)[:, None, None] * np.linalg.inv(ref.mass)[None, :, :]

    def rhs(self):
        """b = f - D*(M\\r) with (f, r) = dg_flux_rhs (src/dg_mesh.jl:342-457;
        tests/dg_heirarchy_test.jl:39-40), rows of the element range."""
        return self._rhs_full

    def _rhs_padded(self):
        n, p, ref = self.n, self.p, self.ref
        nL, nR = (0, 1) if p >= 1 else (0, 0)
        xq = self.xc[:, None] + self.h[:, None] / 2.0 * ref.gq[None, :]
        fq = self.func(xq)
        f = np.zeros((n, p + 1))
        for l in range(len(ref.gw)):
            f += (self.J[:, None] * ref.gw[l]) * ref.phi[l][None, :] * fq[:, l][:, None]
        r = np.zeros((n, p + 1))
        (lk, lv), (rk, rv) = self.bc
        if self.end_left:
            if lk == 'dir':
                f[0, nL] += self.CDir * lv
                r[0, nL] += -lv
            else:
                f[0, nL] += -lv
        if self.end_right:
            if rk == 'dir':
                f[-1, nR] += self.CDir * rv
                r[-1, nR] += rv
            else:
                f[-1, nR] += rv
        y = np.matmul(self.Minv, r[:, :, None])[:, :, 0]
        Dy = np.einsum('kij,kj->ki', self.Dd, y) + np.einsum('kij,kj->ki', self.Du, _shift_up(y))
        return (f - Dy).reshape(-1)

    @staticmethod
    def _bsolve(M, X):
        """batched M[k] \\ X[k] for small blocks: by division (1 x 1) or the adjugate (2 x 2), LAPACK
        beyond that (equal to the reference's per-block LU solve up to round-off)"""
        m = M.shape[-1]
        if m == 1:
            return X / M
        if m == 2:
            det = M[:, 0, 0] * M[:, 1, 1] - M[:, 0, 1] * M[:, 1, 0]
            out = np.empty(np.broadcast_shapes(M.shape[:1] + X.shape[1:], X.shape))
            out[:, 0] = (M[:, 1, 1, None] * X[:, 0] - M[:, 0, 1, None] * X[:, 1]) / det[:, None]
            out[:, 1] = (M[:, 0, 0, None] * X[:, 1] - M[:, 1, 0, None] * X[:, 0]) / det[:, None]
            return out
        return np.linalg.solve(M, X)

    @classmethod
    def _stiffness_blocks(cls, Gl, Gd, Dd, Du, Cd, M, Minv=None):
        """Blocks of A = C - D*(M \\ G) for block-lower-bidiagonal G (Gl, Gd), block-upper-
        bidiagonal D (Dd, Du), block-diagonal C and M.  -> (sub, diag, sup).  Minv: explicit block
        inverses when they are cheap to come by (the nodal DG mass blocks are J_e * M_ref)."""
        if Minv is not None:
            XGd, XGl = np.matmul(Minv, Gd), np.matmul(Minv, Gl)
        else:
            XGd, XGl = cls._bsolve(M, Gd), cls._bsolve(M, Gl)
        diag = Cd - (Dd @ XGd + Du @ _shift_up(XGl))
        sub = -(Dd @ XGl)
        sup = -(Du @ _shift_up(XGd))
        return sub, diag, sup

    # ---- agglomerated levels -------------------------------------------------------------------
    def _modal(self, lo, hi, x):
        """evaluate_local_modal_basis_fun (src/agglomerated_dg_mesh.jl:297-315) for boxes
        [lo, hi] broadcast against points x -> (..., pAgg+1)"""
        out = np.ones(x.shape + (self.pAgg + 1,))
        if self.pAgg == 1:
            xC = (lo + hi) / 2.0
            hh = hi - lo
            out[..., 1] = 2 * (x - xC) / hh
        return out

    def _build_agglomerated(self):
        n, ref = self.n, self.ref
        mc = self.pAgg + 1
        gq, gw = gauss_quad(2 * self.pAgg)
        xq = self.xc[:, None] + self.h[:, None] / 2.0 * gq[None, :]          # (n, nq) base Gauss points
        prev = self.levels[0]
        base_per = 1
        prev_phi_q = None   # modal basis of the previous agglomerated level at base Gauss points
        for li, rho in enumerate(self.ratios):
            base_per *= rho
            ne = n // base_per
            # bounding boxes from the base DG elements' end nodes mRefMap(-1), mRefMap(+1)
            # (src/agglomerated_dg_mesh.jl:183-189 reads baseMesh.mElements[k].mNodesX[1:2])
            lo = (self.xc + self.h / 2.0 * (-1.0))[0:n:base_per]
            hi = (self.xc + self.h / 2.0 * (1.0))[base_per - 1::base_per]
            # modal basis of this level at the base elements' Gauss points: (ne, base_per, nq, mc)
            phi_q = self._modal(lo[:, None, None], hi[:, None, None], xq.reshape(ne, base_per, -1))
            Jb = self.J.reshape(ne, base_per)
            # mass: sum over base elements k, Gauss points l, in loop order k then l
            M = np.zeros((ne, mc, mc))
            for k in range(base_per):
                for l in range(len(gw)):
                    M += (Jb[:, k] * gw[l])[:, None, None] * phi_q[:, k, l, :, None] * phi_q[:, k, l, None, :]
            if li == 0:
                # aggdg_dg_interpolation: modal basis at the base DG nodes
                xn = self.xc[:, None] + self.h[:, None] / 2.0 * ref.nodes[None, :]   # (n, m_f)
                Lb = self._modal(np.repeat(lo, rho)[:, None], np.repeat(hi, rho)[:, None], xn)
            else:
                # aggdg_aggdg_interpolation: N then M_f \ N per fine agglomerate
                nef = prev['ne']
                bpf = base_per // rho                       # base elements per fine agglomerate
                fphi = prev_phi_q                           # (nef, bpf, nq, mc)
                cphi = phi_q.reshape(nef, bpf, len(gw), mc)  # coarse basis, grouped by fine agglomerate
                Jf = self.J.reshape(nef, bpf)
                N = np.zeros((nef, mc, mc))
                for k in range(bpf):
                    for l in range(len(gw)):
                        N += (Jf[:, k] * gw[l])[:, None, None] * fphi[:, k, l, :, None] * cphi[:, k, l, None, :]
                Lb = self._bsolve(prev['M'], N)
            self.transfers.append(dict(Lb=Lb, rho=rho, mc=mc))
            # Galerkin products on the block-bidiagonal G, D and block-diagonal C
            Gl, Gd = prev['G']
            Dd, Du = prev['D']
            Gl_c, Gd_c = self._galerkin_lower(Lb, Gl, Gd, rho)
            Du_c, Dd_c = self._galerkin_upper(Lb, Du, Dd, rho)
            Cd_c = self._galerkin_diag(Lb, prev['C'], rho)
            lvl = dict(m=mc, ne=ne, G=(Gl_c, Gd_c), D=(Dd_c, Du_c), C=Cd_c, M=M,
                       A=self._stiffness_blocks(Gl_c, Gd_c, Dd_c, Du_c, Cd_c, M))
            self.levels.append(lvl)
            prev = lvl
            prev_phi_q = phi_q

    @staticmethod
    def _galerkin_diag(Lb, Xd, rho):
        t = _lt_x_l(Lb, Xd, Lb)
        return t.reshape(-1, rho, *t.shape[1:]).sum(axis=1)

    @staticmethod
    def _galerkin_lower(Lb, Xl, Xd, rho):
        """L'XL for block-lower-bidiagonal X (Xl[k] couples k to k-1)."""
        d = _lt_x_l(Lb, Xd, Lb)
        l = _lt_x_l(Lb, Xl, _shift_down(Lb))   # rows of k, cols of k-1
        d = d.reshape(-1, rho, *d.shape[1:])
        l = l.reshape(-1, rho, *l.shape[1:])
        diag = d.sum(axis=1) + l[:, 1:].sum(axis=1)   # couplings inside the agglomerate
        sub = l[:, 0].copy()                          # first fine element couples to previous agglomerate
        return sub, diag

    @staticmethod
    def _galerkin_upper(Lb, Xu, Xd, rho):
        """L'XL for block-upper-bidiagonal X (Xu[k] couples k to k+1)."""
        d = _lt_x_l(Lb, Xd, Lb)
        u = _lt_x_l(Lb, Xu, _shift_up(Lb))
        d = d.reshape(-1, rho, *d.shape[1:])
        u = u.reshape(-1, rho, *u.shape[1:])
        diag = d.sum(axis=1) + u[:, :-1].sum(axis=1)
        sup = u[:, -1].copy()
        return sup, diag

    # ---- products for the hot path -----------------------------------------------------------
    @property
    def nlevels(self):
        return len(self.levels)

    def stiffness_arrays(self, k):
        """H.mStiffness[k] as the arrays of a SparseMatrixCSC: (m, n, colptr, rowval, nzval, one_based=0),
        int64 indices (numerical pattern of `C - D*(M\\G)`) -- what DeviceOperator takes as it is"""
        sub, diag, sup = self.levels[k]['A']
        colptr, rowval, nzval, N = block_tridiag_to_csc(sub, diag, sup)
        return (N, N, colptr, rowval, nzval, 0)

    def stiffness_csc(self, k):
        """H.mStiffness[k] as scipy CSC"""
        N, _, colptr, rowval, nzval, _ = self.stiffness_arrays(k)
        return _csc(colptr, rowval, nzval, (N, N))

    def interpolation_arrays(self, k):
        """H.mInterpolation[k] (level k+1 -> k) as (m, n, colptr, rowval, nzval, one_based=0): every column
        holds all rows of its agglomerate, zeros included."""
        t = self.transfers[k]
        Lb, rho, mc = t['Lb'], t['rho'], t['mc']
        nef, mf, _ = Lb.shape
        nec = nef // rho
        R = rho * mf
        colptr = np.arange(nec * mc + 1, dtype=np.int64) * R
        vals = np.empty(nec * mc * R)
        rowval = np.empty(nec * mc * R, dtype=np.int64)
        workers = _gen_workers() if nef >= _PAR_MIN_ELEMS else 1
        r = np.arange(R, dtype=np.int64)

        def part(rng):
            c0, c1 = rng
            vals[c0 * mc * R:c1 * mc * R] = Lb[c0 * rho:c1 * rho].reshape(c1 - c0, R, mc).transpose(0, 2, 1).reshape(-1)
            rowval[c0 * mc * R:c1 * mc * R] = (np.repeat(np.arange(c0, c1, dtype=np.int64), mc)[:, None] * R
                                                + r[None, :]).reshape(-1)

        _pool_map(part, _ranges(nec, 4 * workers if workers > 1 else 1), workers)
        return (nef * mf, nec * mc, colptr, rowval, vals, 0)

    def interpolation_csc(self, k):
        """H.mInterpolation[k] as scipy CSC"""
        m, n, colptr, rowval, vals, _ = self.interpolation_arrays(k)
        return _csc(colptr, rowval, vals, (m, n))

    def descriptor(self, k):
        lv = self.levels[k]
        return LevelDescriptor(lv['m'] - 1, lv['ne'])

    def algorithmic_bytes(self, nPre=3, nPost=3):
        """SURVEY.md 8(d) / BASELINE.md section 3 byte model, from the actual nnz:
        per-level sweep S, residual R, restriction and prolongation bytes and the V-cycle sum."""
        out = []
        n = self.nlevels
        for k in range(n - 1):
            lv = self.levels[k]
            N = lv['m'] * lv['ne']
            sub, diag, sup = lv['A']
            nnzA = int(np.count_nonzero(diag)) + int(np.count_nonzero(sub[1:])) + int(np.count_nonzero(sup[:-1]))
            Nc = self.levels[k + 1]['m'] * self.levels[k + 1]['ne']
            nnzL = self.transfers[k]['Lb'].size
            S = 12 * nnzA + 4 * (N + 1) + 8 * lv['m'] * N + 24 * N
            R = 12 * nnzA + 4 * (N + 1) + 24 * N
            Td = 12 * nnzL + 4 * (Nc + 1) + 8 * N + 8 * Nc
            Tu = 12 * nnzL + 4 * (N + 1) + 8 * Nc + 16 * N
            out.append(dict(N=N, nnzA=nnzA, nnzL=nnzL, sweep=S, residual=R, restrict=Td, prolong=Tu,
                            vcycle=(nPre + nPost) * S + R + Td + Tu))
        return out


def build_device_hierarchy(U, ctx=None, keep_host=False, smoother="blockJac", csc=None):
    """Upload a UniformDgAggHierarchy through the CSC boundary and return the product
    MeshHierarchy (block-Jacobi on every smoothed level, src/mesh_heirarchy.jl:58,73,85,104;
    smoother="blockGS": the labelled red-black block Gauss-Seidel extension).
    csc = (stiffness list, interpolation list): matrices already assembled by the caller (bench.py times
    the generator and the library set-up apart)."""
    from .api import BlockGaussSeidel, BlockJacobi, DeviceOperator, MeshHierarchy
    from . import _lib
    BlockJacobi = {"blockJac": BlockJacobi, "blockGS": BlockGaussSeidel}[smoother]
    n = U.nlevels
    ops, sms = [], []
    for k in range(n):
        op = DeviceOperator(csc[0][k] if csc else U.stiffness_csc(k), _lib.OP_STIFFNESS, ctx)
        ops.append(op)
        if k < n - 1:
            sms.append(BlockJacobi(op, U.descriptor(k).mBlockInds, ctx))
    Ls = [DeviceOperator(csc[1][k] if csc else U.interpolation_csc(k), _lib.OP_TRANSFER, ctx) for k in range(n - 1)]
    H = MeshHierarchy([U.descriptor(k) for k in range(n)], ops, sms, Ls, ctx=ctx, keep_host=keep_host)
    return H


# ------------------------------------------------------------------------------------------
# CG p-chain + DG p=0 coarsest level (the realisable shape of BASELINE config 5 / config 1,
# tests/dg_cg_heirarchy_test.jl with nCG levels and nDG = 1)
# ------------------------------------------------------------------------------------------
def build_device_ragged_hierarchy(n, ctx=None, p=3, pAgg=1, nAgg=3, sizes=(2, 3, 4, 5, 6), jitter=0.3, seed=0,
                                  keep_host=False, generic=False):
    """A hierarchy the uniform generator cannot make, at benchmark size: DG p on a mesh with PERTURBED vertices
    (every interior vertex moved by up to jitter * h), then nAgg agglomerated levels whose agglomerates have sizes
    drawn from `sizes` -- the fused kernels then run on parent / first-child maps instead of one ratio, an
    agglomerate cut by a tile boundary is restricted by both tiles (two atomic adds).  Fine-level G, D, C, A in O(n)
    (the LDG blocks of src/dg_mesh.jl:144-336 do not depend on the element sizes, the mass blocks are J_e M_ref);
    every L_k and mass matrix from the product's builders (interpolation.py), the Galerkin recurrences, A_k and the
    smoothers on the device (MeshHierarchy.from_dg_operators).  generic = True: the SAME operators with the block
    lists handed over in a scrambled order, which sends every level through the generic kernels (block-Jacobi does
    not depend on the order of its blocks) -- the unfused composition the fused path is checked against -- as
    info["generic"], sharing H's operators.
    -> (H, b_host, info)"""
    from . import _lib
    from . import interpolation as ip
    from .api import BlockDiagonal, BlockJacobi, DeviceOperator, MeshHierarchy, default_context
    ctx = ctx or default_context()
    rng = np.random.default_rng(seed)
    U = UniformDgAggHierarchy(n, p=p, pAgg=pAgg, ratios=(), workers=1)    # (one piece: keeps the LDG blocks Gl .. Cd)
    xv = np.linspace(0.0, 1.0, n + 1)
    xv[1:-1] += (jitter / n) * (2.0 * rng.random(n - 1) - 1.0)
    J = np.diff(xv) / 2.0
    M = J[:, None, None] * U.ref.mass[None, :, :]
    Minv = (1.0 / J)[:, None, None] * np.linalg.inv(U.ref.mass)[None, :, :]
    sub, diag, sup = U._stiffness_blocks(U.Gl, U.Gd, U.Dd, U.Du, U.Cd, M, Minv)
    Z = np.zeros_like(U.Gd)

    def csc(s_, d_, u_):
        colptr, rowval, nzval, N = block_tridiag_to_csc(s_, d_, u_)
        return _csc(colptr, rowval, nzval, (N, N))

    A, G, D, C = csc(sub, diag, sup), csc(U.Gl, U.Gd, Z), csc(Z, U.Dd, U.Du), csc(Z, U.Cd, Z)
    meshes = [ip.DgMesh(xv, p)]
    for _ in range(nAgg):
        nsub = meshes[-1].n
        lens = rng.choice(np.asarray(sizes, dtype=np.int64), size=nsub // int(min(sizes)) + 1)
        starts = np.concatenate([[0], np.cumsum(lens)])
        starts = starts[starts < nsub]
        starts = np.concatenate([starts, [nsub]]).astype(np.int64)
        meshes.append(ip.AgglomeratedDgMesh(pAgg, starts, meshes[-1]))
    Ls = [ip.aggdg_dg_interpolation(meshes[1], meshes[0])] + \
         [ip.aggdg_aggdg_interpolation(meshes[k + 1], meshes[k]) for k in range(1, nAgg)]
    masses = [BlockDiagonal(m_.mass_blocks(), ctx) for m_ in meshes[1:]]
    H = MeshHierarchy.from_dg_operators(meshes, A, G, D, C, Ls, masses, ctx=ctx, keep_host=keep_host)
    info = {"elements": [int(m_.n) for m_ in meshes], "mean_agglomerate": [float(meshes[k].n / meshes[k + 1].n) for k in range(nAgg)]}
    if generic:
        sms = []
        for k in range(nAgg):
            inds = np.asarray(meshes[k].mBlockInds)
            sms.append(BlockJacobi(H._ops[k], np.ascontiguousarray(inds[:, rng.permutation(inds.shape[1])]), ctx))
        info["generic"] = MeshHierarchy(None, list(H._ops), sms, list(H._Ls), ctx=ctx, keep_host=False)
    # right-hand side: the load vector of f = cos with the boundary terms of the UNIFORM mesh of the same size (entries
    # of size h, a solution of size one: the scaling of the model problem, under which the 1e-12 residual criterion
    # is meaningful; a random vector of size one has a solution of size 1 / h^2 and round-off to match).  Not the
    # exact load vector of the perturbed mesh -- the cycle's linear algebra does not care
    b = U.rhs()
    return H, b, info


class UniformCgDgHierarchy:
    """CgMesh(p) for p in `ps` (p-coarsening by nodal injection, Galerkin operators, point-Jacobi)
    followed by one re-discretised DgMesh(p=0) level reached through the lumped-mass L2 transfer
    (interpFlag = 1), as MeshHierarchy(mMeshes, mesh, bdConds, A; nCG, nDG=1, CDir) builds it
    (src/mesh_heirarchy.jl:30-73).  O(n): every matrix is written straight into its CSC arrays from
    per-element blocks, no sparse products and no COO sort --

        cg_stiffness_and_rhs       src/cg_mesh.jl:125-185      -> element matrices K_e, rhs
        cg_cg_interpolation        src/interpolation.jl:5-55   -> L (same for every element but the last)
        L'*A*L                     src/mesh_heirarchy.jl:57    -> K_e^c = L_e' K_e L_e element by element
        dg_cg_interpolation(...,1) src/interpolation.jl:145-220

    CG numbering: vertices 1..n+1, then the p-1 interior nodes of every element in element order
    (src/cg_mesh.jl:37-45,59-65); node indices here are 0-based."""

    def __init__(self, n, ps=(4, 2, 1), CDir=None, xin=0.0, xout=1.0, bc=None, func=np.cos, elem_range=None):
        """elem_range=(a, b): only the elements a..b-1 (0-based) and their nodes -- the local operators of one rank
        of an element-partitioned run, numbered vertices-first on the sub-mesh.  Contributions of elements
        outside the range are dropped (the cut vertices carry incomplete rows: ghost layers absorb that);
        everything else equals the corresponding entries of the global operators."""
        a, b = (0, n) if elem_range is None else (int(elem_range[0]), int(elem_range[1]))
        if not (0 <= a < b <= n):
            raise ValueError("elem_range must be a non-empty range of elements")
        self.n_global, self.a, self.b = n, a, b
        self.at_left, self.at_right = (a == 0), (b == n)
        self.n, self.ps = b - a, tuple(ps)          # elements the arrays are built on
        self.CDir = 1000.0 * n if CDir is None else float(CDir)
        self.bc = bc or (('neu', -math.sin(xin)), ('dir', math.cos(xout)))
        self.func = func
        i = np.arange(a, b + 1, dtype=np.float64)
        self.xv = xin + (i / n) * (xout - xin)
        if a == 0:
            self.xv[0] = xin
        self.h = self.xv[1:] - self.xv[:-1]
        self.xc = (self.xv[:-1] + self.xv[1:]) / 2.0
        self.J = self.h / 2.0
        self.refs = [RefElement(p) for p in self.ps]
        nlev = len(self.ps) + 1
        self.A, self.L = [None] * nlev, [None] * (nlev - 1)
        # the element matrices of the levels follow from each other (cheap); every assembly / transfer after that
        # is independent of the others: large meshes run them as tasks in threads (same functions, same bits)
        Ke, extra, dirv, self.b = self._cg_element_matrices_and_rhs(self.ps[0], self.refs[0])
        tasks = [lambda Ke=Ke, extra=extra: self.A.__setitem__(0, self._assemble(self.ps[0], Ke, extra, dirv))]
        for k in range(1, len(self.ps)):
            lowVal = self._low_val(self.refs[k], self.refs[k - 1])
            tasks.append(lambda k=k, lowVal=lowVal: self.L.__setitem__(k - 1, self._cg_cg(self.ps[k], self.ps[k - 1], lowVal)))
            Ke, extra = self._galerkin(Ke, extra, lowVal)
            tasks.append(lambda k=k, Ke=Ke, extra=extra: self.A.__setitem__(k, self._assemble(self.ps[k], Ke, extra, None)))

        def dg0_level():
            # DG p = 0 level: re-discretised operator, lumped-mass transfer from the last CG level
            self.dg0 = UniformDgAggHierarchy(n, p=0, pAgg=0, ratios=(), CDir=self.CDir, xin=xin, xout=xout, bc=self.bc,
                                             func=func, elem_range=elem_range, workers=1)
            self.A[nlev - 1] = self.dg0.stiffness_csc(0)

        tasks.append(dg0_level)
        tasks.append(lambda: self.L.__setitem__(nlev - 2, self._dg0_cg(self.ps[-1], self.refs[-1])))
        workers = min(len(tasks), _gen_workers()) if (self.n >= _PAR_MIN_ELEMS or os.environ.get("AGGMG_GEN_FORCE_PARALLEL")) else 1
        _pool_map(lambda t: t(), tasks, workers)

    def _workers(self):
        """threads for the element-range fills of one assembly (the assemblies themselves also run side by side)"""
        if self.n >= _PAR_MIN_ELEMS or os.environ.get("AGGMG_GEN_FORCE_PARALLEL"):
            return _gen_workers()
        return 1

    def nodes(self, p):
        """(n, p+1) 0-based node numbers per element in local order [left, right, interior...]"""
        n = self.n
        k = np.arange(n, dtype=np.int64)
        out = np.empty((n, p + 1), dtype=np.int64)
        out[:, 0], out[:, 1] = k, k + 1
        for j in range(p - 1):
            out[:, 2 + j] = (n + 1) + k * (p - 1) + j
        return out

    def element_nodes(self, k):
        """mElements[e].mNodesInd of CG level k as the (p+1) x n matrix (1-based) that
        cg_smoother(cgMesh, A, ...) has at hand -- the hint of aggmg_jacobi_setup_elements"""
        return np.ascontiguousarray(self.nodes(self.ps[k]).T) + 1

    def num_nodes(self, p):
        return self.n * p + 1

    def _dir_nodes(self):
        d = []
        if self.bc[0][0] == 'dir' and self.at_left:
            d.append((0, self.bc[0][1]))
        if self.bc[1][0] == 'dir' and self.at_right:
            d.append((self.n, self.bc[1][1]))
        return d

    def _cg_element_matrices_and_rhs(self, p, ref):
        """-> K_e (n, p+1, p+1) with the Dirichlet rows / columns zeroed, the identity entries of the
        Dirichlet vertices as a vertex vector, the Dirichlet vertex mask and the right-hand side"""
        n = self.n
        N = self.num_nodes(p)
        m = p + 1
        # temp[i,j] += (1/J) * w_l * dphi_i * dphi_j (src/cg_mesh.jl:140-150): the l-sum is formed once on
        # the reference element and scaled per element (equal to the loop order up to round-off)
        Kref = np.zeros((m, m))
        for l in range(len(ref.gw)):
            Kref += (ref.gw[l] * ref.dphi[l])[:, None] * ref.dphi[l][None, :]
        K = np.empty((n, m, m))
        fe = np.empty((n, m))
        workers = self._workers()

        def part(r):
            a, b = r
            K[a:b] = (1.0 / self.J[a:b])[:, None, None] * Kref[None, :, :]
            xq = self.xc[a:b, None] + self.h[a:b, None] / 2.0 * ref.gq[None, :]
            fq = self.func(xq)
            t = np.zeros((b - a, m))
            for l in range(len(ref.gw)):
                t += (self.J[a:b, None] * ref.gw[l]) * ref.phi[l][None, :] * fq[:, l][:, None]
            fe[a:b] = t

        _pool_map(part, _ranges(n, 4 * workers if workers > 1 else 1), workers)
        f = np.zeros(N)
        f[1:n + 1] += fe[:, 1]          # element v-1 reaches vertex v before element v does
        f[0:n] += fe[:, 0]
        if p > 1:
            f[n + 1:] = fe[:, 2:].reshape(-1)
        (lk, lv), (rk, rv) = self.bc
        if lk == 'neu' and self.at_left:
            f[0] += -lv
        if rk == 'neu' and self.at_right:
            f[n] += rv
        extra = np.zeros(n + 1)
        dirv = np.zeros(n + 1, dtype=bool)
        dirs = self._dir_nodes()
        nd = self.nodes(p)
        for node, val in dirs:                 # f += -A[:, dir] * dirVals on the unconstrained matrix
            for e, loc in ((node - 1, 1), (node, 0)):
                if 0 <= e < n:
                    f[nd[e]] += -K[e][:, loc] * val
        for node, val in dirs:
            f[node] = val
            extra[node] = 1.0
            dirv[node] = True
            for e, loc in ((node - 1, 1), (node, 0)):
                if 0 <= e < n:
                    K[e][loc, :] = 0.0
                    K[e][:, loc] = 0.0
        return K, extra, dirv, f

    def _low_val(self, ref_lo, ref_hi):
        """coarse nodal basis at the fine nodes: lowVal[i, j] (src/interpolation.jl:10)"""
        V = legendre_vandermonde(ref_hi.nodes, ref_lo.p)
        return np.array([[np.dot(ref_lo.coeff[:, j], V[i]) for j in range(ref_lo.p + 1)]
                         for i in range(ref_hi.p + 1)])

    def _vertex_l(self, lowVal):
        """value of L at (fine vertex v, coarse vertex v): the last element to touch a vertex writes it
        (src/interpolation.jl:47-52): local node 1 of the last element for vertex n, local node 0 otherwise"""
        vv = np.full(self.n + 1, lowVal[0, 0])
        if self.at_right:
            vv[self.n] = lowVal[1, 1]
        return vv

    def _galerkin(self, Ke, extra, lowVal):
        """element matrices of L'*A*L: K_e^c = L_e' K_e L_e with L_e = rows of L at the element's
        fine nodes (vertex rows carry the vertex entry of L, interior rows the coarse basis values)"""
        n = self.n
        vv = self._vertex_l(lowVal)
        Le = np.broadcast_to(lowVal, (n,) + lowVal.shape).copy()
        Le[:, 0, :] = 0.0
        Le[:, 1, :] = 0.0
        Le[:, 0, 0] = vv[:-1]
        Le[:, 1, 1] = vv[1:]
        Kc = np.matmul(Le.transpose(0, 2, 1), np.matmul(Ke, Le))
        return Kc, vv * (extra * vv)

    @staticmethod
    def _pack_columns(rows, vals, keep_rows):
        """CSC pieces of a strip array (ncols, W): column c stores rows[c, :] / vals[c, :], except the
        columns listed in keep_rows = {c: bool mask of length W}.  Runs of regular columns are taken
        with one reshape each, so no boolean pass over the whole strip is needed.
        -> (counts per column, list of row pieces, list of value pieces)"""
        nc, W = rows.shape
        counts = np.full(nc, W, dtype=np.int64)
        rparts, vparts = [], []
        a = 0
        for c in sorted(keep_rows):
            if c > a:
                rparts.append(rows[a:c].reshape(-1)), vparts.append(vals[a:c].reshape(-1))
            k = keep_rows[c]
            rparts.append(rows[c][k]), vparts.append(vals[c][k])
            counts[c] = int(k.sum())
            a = c + 1
        if a < nc:
            rparts.append(rows[a:].reshape(-1)), vparts.append(vals[a:].reshape(-1))
        return counts, rparts, vparts

    def _assemble(self, p, Ke, extra, dirv):
        """CSC of sum_e P_e' K_e P_e + diag(extra) in the vertices-first numbering.  dirv (fine level):
        Dirichlet vertices whose rows / columns are removed from the pattern, as the reference's
        sparse setindex! does (src/cg_mesh.jl:177-182); Galerkin levels keep every entry."""
        n = self.n
        q = p - 1
        N = self.num_nodes(p)
        v = np.arange(n + 1, dtype=np.int64)
        W = 3 + 2 * q
        dirs = [] if dirv is None else [int(d) for d in np.flatnonzero(dirv)]
        # ---- vertex columns: rows v-1, v, v+1, interior of element v-1, interior of element v ----
        rows = np.empty((n + 1, W), dtype=np.int64)
        vals = np.zeros((n + 1, W))
        workers = self._workers()
        jj = np.arange(q, dtype=np.int64)

        def fill_vertices(r):
            v0, v1 = r                       # vertices v0 .. v1-1; element v-1 reaches vertex v before element v does
            vv = v[v0:v1]
            lo, hi = max(v0, 1), min(v1, n)
            rows[v0:v1, 0], rows[v0:v1, 1], rows[v0:v1, 2] = vv - 1, vv, vv + 1
            vals[lo:v1, 0] = Ke[lo - 1:v1 - 1, 0, 1]
            vals[lo:v1, 1] = Ke[lo - 1:v1 - 1, 1, 1]
            vals[v0:hi, 1] += Ke[v0:hi, 0, 0]
            vals[v0:v1, 1] += extra[v0:v1]
            vals[v0:hi, 2] = Ke[v0:hi, 1, 0]
            if q:
                rows[v0:v1, 3:3 + q] = (n + 1) + (vv[:, None] - 1) * q + jj[None, :]
                rows[v0:v1, 3 + q:] = (n + 1) + vv[:, None] * q + jj[None, :]
                vals[lo:v1, 3:3 + q] = Ke[lo - 1:v1 - 1, 2:, 1]
                vals[v0:hi, 3 + q:] = Ke[v0:hi, 2:, 0]

        _pool_map(fill_vertices, _ranges(n + 1, 4 * workers if workers > 1 else 1), workers)
        irregular = {}

        def vmask(c):
            if c not in irregular:
                k = np.ones(W, dtype=bool)
                if c == 0:
                    k[0] = False
                    k[3:3 + q] = False
                if c == n:
                    k[2] = False
                    k[3 + q:] = False
                irregular[c] = k
            return irregular[c]

        vmask(0), vmask(n)
        for d in dirs:
            k = vmask(d)
            k[:] = False
            k[1] = True                                   # the identity entry
            if d + 1 <= n:
                vmask(d + 1)[0] = False                   # row v-1 is a Dirichlet row
            if d - 1 >= 0:
                vmask(d - 1)[2] = False
        counts, rparts, vparts = self._pack_columns(rows, vals, irregular)
        counts = [counts]
        # ---- interior columns (e, j): rows v_e, v_e+1, interior of element e ----
        if q:
            irows = np.empty((n, q, p + 1), dtype=np.int64)
            ivals = np.empty((n, q, p + 1))

            def fill_interiors(r):
                e0, e1 = r
                e = np.arange(e0, e1, dtype=np.int64)
                irows[e0:e1, :, 0] = e[:, None]
                irows[e0:e1, :, 1] = e[:, None] + 1
                irows[e0:e1, :, 2:] = (n + 1) + e[:, None, None] * q + jj[None, None, :]
                ivals[e0:e1] = Ke[e0:e1, :, 2:].transpose(0, 2, 1)

            _pool_map(fill_interiors, _ranges(n, 4 * workers if workers > 1 else 1), workers)
            irr = {}
            for d in dirs:
                for el, loc in ((d - 1, 1), (d, 0)):       # elements touching the Dirichlet vertex
                    if 0 <= el < n:
                        for j in range(q):
                            k = irr.setdefault(el * q + j, np.ones(p + 1, dtype=bool))
                            k[loc] = False
            c2, r2, v2 = self._pack_columns(irows.reshape(n * q, p + 1), ivals.reshape(n * q, p + 1), irr)
            counts.append(c2)
            rparts += r2
            vparts += v2
        colptr = np.zeros(N + 1, dtype=np.int64)
        np.cumsum(np.concatenate(counts), out=colptr[1:])
        A = sp.csc_matrix((_par_concat(vparts, workers), _par_concat(rparts, workers), colptr), shape=(N, N))
        A.has_sorted_indices = True
        return A

    def _cg_cg(self, p_lo, p_hi, lowVal):
        """prolongation CG(p_lo) -> CG(p_hi): interior fine nodes x all coarse nodes of the element,
        plus the vertex identity pairs"""
        n = self.n
        qf, qc = p_hi - 1, p_lo - 1
        Nf, Nc = self.num_nodes(p_hi), self.num_nodes(p_lo)
        v = np.arange(n + 1, dtype=np.int64)
        jf = np.arange(qf, dtype=np.int64)
        # coarse vertex column v: fine vertex v, fine interior of element v-1 (its local node 1), of element v (node 0)
        W = 1 + 2 * qf
        rows = np.empty((n + 1, W), dtype=np.int64)
        vals = np.empty((n + 1, W))
        rows[:, 0] = v
        vals[:, 0] = self._vertex_l(lowVal)
        rows[:, 1:1 + qf] = (n + 1) + (v[:, None] - 1) * qf + jf[None, :]
        rows[:, 1 + qf:] = (n + 1) + v[:, None] * qf + jf[None, :]
        vals[:, 1:1 + qf] = lowVal[2:, 1][None, :]
        vals[:, 1 + qf:] = lowVal[2:, 0][None, :]
        k0, kn = np.ones(W, dtype=bool), np.ones(W, dtype=bool)
        k0[1:1 + qf] = False
        kn[1 + qf:] = False
        if n == 0:
            raise ValueError("need at least one element")
        counts, rparts, vparts = self._pack_columns(rows, vals, {0: k0, n: kn})
        counts = [counts]
        if qc:   # coarse interior column (e, j): fine interior of element e
            e = np.arange(n, dtype=np.int64)
            irows = np.broadcast_to(((n + 1) + e[:, None, None] * qf + jf[None, None, :]), (n, qc, qf))
            ivals = np.broadcast_to(lowVal[2:, 2:].T[None, :, :], (n, qc, qf))
            counts.append(np.full(n * qc, qf, dtype=np.int64))
            rparts.append(irows.reshape(-1)), vparts.append(ivals.reshape(-1))
        colptr = np.zeros(Nc + 1, dtype=np.int64)
        np.cumsum(np.concatenate(counts), out=colptr[1:])
        L = sp.csc_matrix((np.concatenate(vparts), np.concatenate(rparts), colptr), shape=(Nf, Nc))
        L.has_sorted_indices = True
        return L

    def _dg0_cg(self, p_hi, ref_hi):
        """dg_cg_interpolation(DgMesh(p=0), CgMesh(p_hi), mesh, 1): N row-scaled by the lumped mass"""
        n = self.n
        nd = self.nodes(p_hi)
        m = p_hi + 1
        q = p_hi - 1
        gq, gw = gauss_quad(0 + p_hi)
        hiV = np.array([[np.dot(ref_hi.coeff[:, i], legendre_vandermonde(gq, p_hi)[l]) for i in range(m)]
                        for l in range(len(gq))])
        T = np.zeros((n, m))
        for l in range(len(gq)):               # temp[i,0] += J * w_l * hiV[l,i] * 1.0
            T += (self.J[:, None] * gw[l]) * hiV[l][None, :] * 1.0
        # lumped mass: row sums of the assembled CG mass matrix in ascending column order; the matrix
        # is symmetric, so the sums run down the columns of its vertices-first strips
        Me = self.J[:, None, None] * ref_hi.mass[None, :, :]
        lumped = np.zeros(self.num_nodes(p_hi))
        sv = np.zeros((n + 1, 3 + 2 * q))
        sv[1:, 0] = Me[:, 0, 1]
        sv[1:, 1] = Me[:, 1, 1]
        sv[:-1, 1] += Me[:, 0, 0]
        sv[:-1, 2] = Me[:, 1, 0]
        if q:
            sv[1:, 3:3 + q] = Me[:, 2:, 1]
            sv[:-1, 3 + q:] = Me[:, 2:, 0]
        acc = np.zeros(n + 1)
        for c in range(sv.shape[1]):
            acc += sv[:, c]
        lumped[:n + 1] = acc
        if q:
            si = Me[:, :, 2:].transpose(0, 2, 1)           # (n, q, m): column (e, j), rows v_e, v_e+1, interior
            acc = np.zeros((n, q))
            for c in range(m):
                acc += si[:, :, c]
            lumped[n + 1:] = acc.reshape(-1)
        # column e of N: rows v_e, v_e+1, interior of element e (ascending)
        vals = T / lumped[nd]
        colptr = np.arange(n + 1, dtype=np.int64) * m
        Nm = sp.csc_matrix((vals.reshape(-1), nd.reshape(-1), colptr), shape=(self.num_nodes(p_hi), n))
        Nm.has_sorted_indices = True
        return Nm

    @property
    def nlevels(self):
        return len(self.A)

    def rhs(self):
        return self.b

    def algorithmic_bytes(self, nPre=3, nPost=3):
        out = []
        for k in range(self.nlevels - 1):
            N, nnzA = self.A[k].shape[0], self.A[k].nnz
            Nc, nnzL = self.L[k].shape[1], self.L[k].nnz
            S = 12 * nnzA + 4 * (N + 1) + 8 * N + 24 * N          # point-Jacobi sweep
            R = 12 * nnzA + 4 * (N + 1) + 24 * N
            Td = 12 * nnzL + 4 * (Nc + 1) + 8 * N + 8 * Nc
            Tu = 12 * nnzL + 4 * (N + 1) + 8 * Nc + 16 * N
            out.append(dict(N=N, nnzA=nnzA, nnzL=nnzL, sweep=S, residual=R, restrict=Td, prolong=Tu,
                            vcycle=(nPre + nPost) * S + R + Td + Tu))
        return out


def build_device_cg_hierarchy(U, ctx=None, keep_host=False, chain=True, smoother="jac"):
    """UniformCgDgHierarchy -> product MeshHierarchy (CG levels :jac, src/mesh_heirarchy.jl:51,58; smoother =
    'addSchwarz' / 'hybridSchwarz': cg_smoother's element Schwarz smoothers; 'blockGS': the labelled red-black element
    Gauss-Seidel extension)"""
    from . import _lib
    from .api import (AdditiveSchwarzSmoother, BlockGaussSeidel, DeviceOperator, HybridSchwarzSmoother, JacobiSmoother,
                      MeshHierarchy)
    ops = [DeviceOperator(A, _lib.OP_STIFFNESS, ctx) for A in U.A]
    # cg_smoother(cgMesh, A, :jac) with the mesh's element node lists (chain form, fused kernel);
    # chain=False is the operators-only route through the generic CSR kernels
    if smoother == "jac":
        sms = [JacobiSmoother(ops[k], ctx, U.element_nodes(k) if chain else None, detect=chain) for k in range(U.nlevels - 1)]
    else:
        cls = {"addSchwarz": AdditiveSchwarzSmoother, "hybridSchwarz": HybridSchwarzSmoother, "blockGS": BlockGaussSeidel}[smoother]
        sms = [cls(ops[k], U.element_nodes(k), ctx) for k in range(U.nlevels - 1)]
    Ls = [DeviceOperator(L, _lib.OP_TRANSFER, ctx) for L in U.L]
    return MeshHierarchy(None, ops, sms, Ls, ctx=ctx, keep_host=keep_host)
