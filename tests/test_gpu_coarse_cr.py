"""The device coarsest solve on its own (csrc/cr_kernels.hpp; replaces `A_n \\ rhs_n`, src/solvers.jl:39):
a one-level hierarchy's V-cycle IS the direct solve.  Block-tridiagonal systems of every instantiated block
size and of sizes that take each path -- a single block, the single-workgroup tail alone, one chunk stage +
tail (with and without the per-chunk stack, several sub-chunks per thread), ragged ends (n not a power of two,
N not a multiple of m) -- against SciPy's sparse LU, on the residual (<= 1e-12 ||b||) and on
the solution (cond * eps).  The phase-by-phase entry points of the element-partitioned driver
(aggmg_coarse_chunk_forward_dev / _boundary_solve_dev / _chunk_backward_dev) are run for 1, 2 and 3 "ranks"
on one GPU and must reproduce the one-call solve bit for bit."""
import ctypes

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mg():
    import agglomerationmultigrid1d_amd as mg
    mg.default_context()
    return mg


def block_tridiag(nb, m, seed, ragged=0):
    """random block-tridiagonal, block-diagonally dominant (pivot blocks well conditioned), not symmetric;
    ragged: drop that many trailing rows/columns so that N is not a multiple of m"""
    rng = np.random.default_rng(seed)
    ii, jj = np.meshgrid(np.arange(m), np.arange(m), indexing="ij")
    rows, cols, vals = [], [], []
    for dr, dc, cnt, shift in ((0, 0, nb, 4.0 * m), (1, 0, nb - 1, 0.0), (0, 1, nb - 1, 0.0)):
        if cnt <= 0:
            continue
        blk = rng.standard_normal((cnt, m, m)) + shift * np.eye(m)
        e = np.arange(cnt)
        rows.append(((e + dr)[:, None, None] * m + ii).ravel())
        cols.append(((e + dc)[:, None, None] * m + jj).ravel())
        vals.append(blk.ravel())
    A = sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(nb * m, nb * m))
    N = nb * m - ragged
    return sp.csc_matrix(A[:N, :N])


def one_level(mg, A, mode=None):
    from agglomerationmultigrid1d_amd import _lib
    ctx = mg.default_context()
    op = mg.DeviceOperator(A, _lib.OP_STIFFNESS, ctx)
    H = mg.MeshHierarchy(None, [op], [], [], ctx=ctx, keep_host=False,
                         coarse_mode=_lib.COARSE_DEVICE_CR if mode is None else mode)
    return ctx, H


CASES = [
    # (blocks, m, ragged rows)
    (1, 1, 0), (1, 3, 0), (2, 2, 0), (3, 1, 0), (7, 4, 0), (9, 2, 1),
    (100, 1, 0), (513, 2, 0), (2048, 2, 0), (4096, 1, 0), (333, 5, 2), (500, 8, 3), (1000, 3, 0),   # tail only
    (4097, 1, 0), (5000, 1, 0), (2049, 2, 0), (3001, 2, 1), (1500, 3, 0), (1025, 4, 0), (900, 5, 0), (600, 7, 0),
    (513, 8, 0),                                                                                      # small stage
    (1 << 15, 1, 0), ((1 << 15) + 3, 2, 0), ((1 << 16) - 1, 1, 0), (40000, 3, 2), (33000, 4, 0),      # stage + stack
    ((1 << 17) + 5, 6, 0), (1 << 18, 2, 0), ((1 << 20) + 1, 1, 0), (1 << 21, 1, 0),                   # > 1 sub-chunk per thread
    (1 << 16, 8, 0), (1 << 20, 5, 0),     # 2^20 blocks of 5: chunks capped at 2^11 blocks by the CU's LDS (2^12 would ask for 164 KB)
]


@pytest.mark.parametrize("nb,m,ragged", CASES)
def test_cr_solve_matches_sparse_lu(mg, nb, m, ragged):
    A = block_tridiag(nb, m, seed=nb * 31 + m, ragged=ragged)
    N = A.shape[0]
    ctx, H = one_level(mg, A)
    info = H.coarse_info()
    assert info["on_device"] and 1 <= info["block_size"] <= 8   # the smallest block size that fits the band
    # the boundary system of the chunk stages (or a small system as a whole) by parallel cyclic reduction where set-up
    # offers it (block sizes 1 and 2, 2 .. 1024 blocks) AND finds it as accurate as the register-blocked form
    offered = info["block_size"] <= 2 and 2 <= info["tail_blocks"] <= 1024
    assert info["tail"] in (("parallel cyclic reduction", "cyclic reduction") if offered else ("cyclic reduction",)), info
    rng = np.random.default_rng(5)
    b = rng.standard_normal(N)
    bd, xd, z = ctx.to_device(b), ctx.alloc(N), ctx.to_device(np.zeros(N))
    H.vcycle_dev(z, bd, xd, 0, 0, 1.0)
    x = xd.download()
    assert np.linalg.norm(A @ x - b) <= 1e-12 * np.linalg.norm(b)
    if N <= 3_000_000:      # (SuperLU runs out of memory beyond: the residual above stands alone there)
        ref = spla.splu(A).solve(b)
        assert np.linalg.norm(x - ref) <= 1e-11 * np.linalg.norm(ref)
    # further solves with the same handle (ticket counter / stack reuse)
    b2 = rng.standard_normal(N)
    y = ctx.alloc(N)
    H.vcycle_dev(z, ctx.to_device(b2), y, 0, 0, 1.0)
    x2 = y.download()
    assert np.linalg.norm(A @ x2 - b2) <= 1e-12 * np.linalg.norm(b2)
    H.vcycle_dev(z, bd, xd, 0, 0, 1.0)
    assert np.array_equal(xd.download(), x)
    H.free()


@pytest.mark.parametrize("nb,m", [(7, 2), (513, 2), ((1 << 15) + 3, 2), (1500, 3), (33000, 4), (600, 7), (513, 8), (1 << 18, 2)])
def test_cr_with_row_pivoting_inside_the_blocks(mg, nb, m):
    """the equations of every block row in reverse order: the same solution, but every diagonal block (and every Schur
    complement of the reduction) now needs its rows exchanged -- the stored permutations are not the identity, which the
    diagonally dominant systems above never produce.  Exercises the permuted triangular solves of the back substitution
    and the row-vector solves  a b^-1 = ((a U^-1) L^-1) P  behind the forward multipliers (cr_even_multipliers_kernel)."""
    A0 = block_tridiag(nb, m, seed=nb * 17 + m)
    N = A0.shape[0]
    rev = (np.arange(N) // m) * m + (m - 1 - np.arange(N) % m)
    A = sp.csc_matrix(A0.tocsr()[rev, :])
    ctx, H = one_level(mg, A)
    info = H.coarse_info()
    assert info["on_device"] and info["block_size"] == m and info["probe_backward_error"] < 1e-13
    b = np.random.default_rng(11).standard_normal(N)
    xd, z = ctx.alloc(N), ctx.to_device(np.zeros(N))
    H.vcycle_dev(z, ctx.to_device(b), xd, 0, 0, 1.0)
    x = xd.download()
    assert np.linalg.norm(A @ x - b) <= 1e-12 * np.linalg.norm(b)
    if N <= 3_000_000:      # (SuperLU runs out of memory beyond: the residual above stands alone there)
        ref = spla.splu(A).solve(b)
        assert np.linalg.norm(x - ref) <= 1e-11 * np.linalg.norm(ref)
    H.free()


def test_parallel_tail_is_kept_on_evidence(mg, oracle):
    """The parallel cyclic reduction of the tail (one thread per row through all levels, no back substitution) accumulates
    like an inverse.  Set-up keeps it only where it is as accurate as the register-blocked form on a right-hand side with
    a large smooth solution: on well-conditioned systems it is (and is used), on an ill-conditioned operator taken as a
    whole -- the DG p = 0 coarsest level of the CG-fine hierarchy: scaled 1-D Laplacian, Neumann end, Dirichlet penalty,
    where its residual was measured at 5000 x the other form's -- it is not."""
    for nb, m in ((513, 2), (1000, 1), ((1 << 15) + 3, 2), (1 << 18, 2)):
        ctx, H = one_level(mg, block_tridiag(nb, m, seed=3))
        info = H.coarse_info()
        assert info["tail"] == "parallel cyclic reduction" and 2 <= info["tail_blocks"] <= 1024, (nb, m, info)
        H.free()
    n = 1000
    Ho, _ = oracle.build_cg_hierarchy(n, ps=(4, 2, 1), nDG=1, pDG=0)
    A = sp.csc_matrix(Ho.mStiffness[-1])
    ctx, H = one_level(mg, A)
    info = H.coarse_info()
    assert info["on_device"] and info["tail_blocks"] == n and info["tail"] == "cyclic reduction", info
    d = 1.0 + 0.5 * np.cos(np.arange(n) * np.pi / n)
    xd, z = ctx.alloc(n), ctx.to_device(np.zeros(n))
    H.vcycle_dev(z, ctx.to_device(d), xd, 0, 0, 1.0)
    y = xd.download()   # (a large solution: the residual is measured against ||A|| ||y||)
    assert np.linalg.norm(A @ y - d) <= 1e-17 * abs(A).sum(axis=1).max() * np.linalg.norm(y)
    H.free()


@pytest.mark.parametrize("nb,m", [(1 << 14, 1), (1 << 15, 2), (3 << 13, 1), (1 << 16, 4), (1 << 20, 1)])
def test_cr_phases_reproduce_the_one_call_solve(mg, nb, m):
    """every 'rank' eliminates / back-substitutes the chunks of its own block range; the boundary system is solved
    redundantly (here once).  Must equal the single-call solve bit for bit."""
    A = block_tridiag(nb, m, seed=7 * nb + m)
    N = A.shape[0]
    ctx, H = one_level(mg, A)
    c = ctx
    q, nq, mb, nblk = ctypes.c_int(0), ctypes.c_int64(0), ctypes.c_int(0), ctypes.c_int64(0)
    c.check(c.lib.aggmg_coarse_plan(c.handle, H.handle, ctypes.byref(q), ctypes.byref(nq), ctypes.byref(mb),
                                    ctypes.byref(nblk)))
    assert q.value > 0 and mb.value == m and nblk.value == nb
    rng = np.random.default_rng(11)
    b = rng.standard_normal(N)
    bd, xd, z = ctx.to_device(b), ctx.alloc(N), ctx.to_device(np.zeros(N))
    H.vcycle_dev(z, bd, xd, 0, 0, 1.0)
    x = xd.download()
    chunk = 1 << q.value
    nchunks = (nb + chunk - 1) // chunk
    for world in (1, 2, 3):
        if nchunks < world:
            continue
        cuts = [((nchunks * r) // world) * chunk for r in range(world)] + [nb]
        partR, partL = ctx.to_device(np.zeros((nq.value + 1) * m)), ctx.to_device(np.zeros((nq.value + 1) * m))
        xq, out = ctx.alloc((nq.value + 1) * m), ctx.to_device(np.zeros(N))
        for r in range(world):
            lo, hi = cuts[r], cuts[r + 1]
            c.check(c.lib.aggmg_coarse_chunk_forward_dev(c.handle, H.handle, bd.ptr.value + 8 * lo * m, lo, hi, partR.ptr,
                                                         partL.ptr))
        c.check(c.lib.aggmg_coarse_boundary_solve_dev(c.handle, H.handle, partR.ptr, partL.ptr, xq.ptr))
        for r in range(world):
            lo, hi = cuts[r], cuts[r + 1]
            c.check(c.lib.aggmg_coarse_chunk_backward_dev(c.handle, H.handle, bd.ptr.value + 8 * lo * m, lo, hi, xq.ptr,
                                                          out.ptr.value + 8 * lo * m))
        assert np.array_equal(out.download(), x), world
    H.free()


@pytest.mark.parametrize("nb,m,tail_rows,max_q", [(1 << 15, 1, 64, 5), ((1 << 14) + 7, 2, 32, 4), (5000, 3, 16, 3),
                                                  (1 << 16, 1, 8, 12)])
def test_cr_several_stages(mg, monkeypatch, nb, m, tail_rows, max_q):
    """systems beyond 2^24 rows chain several chunk stages (the boundary system of one stage is the input of the
    next); the plan knobs shrink the tail and the chunks so that the same plan runs at a size SciPy solves in
    seconds -- and the partitioned phases hand the boundary system to the remaining stages"""
    monkeypatch.setenv("AGGMG_CR_TAIL_ROWS", str(tail_rows))
    monkeypatch.setenv("AGGMG_CR_MAX_Q", str(max_q))
    A = block_tridiag(nb, m, seed=nb + m)
    N = A.shape[0]
    ctx, H = one_level(mg, A)
    rng = np.random.default_rng(2)
    b = rng.standard_normal(N)
    bd, xd, z = ctx.to_device(b), ctx.alloc(N), ctx.to_device(np.zeros(N))
    H.vcycle_dev(z, bd, xd, 0, 0, 1.0)
    x = xd.download()
    assert np.linalg.norm(A @ x - b) <= 1e-12 * np.linalg.norm(b)
    assert np.linalg.norm(x - spla.splu(A).solve(b)) <= 1e-11 * np.linalg.norm(x)
    c = ctx
    q, nq, mb, nblk = ctypes.c_int(0), ctypes.c_int64(0), ctypes.c_int(0), ctypes.c_int64(0)
    c.check(c.lib.aggmg_coarse_plan(c.handle, H.handle, ctypes.byref(q), ctypes.byref(nq), ctypes.byref(mb),
                                    ctypes.byref(nblk)))
    assert 0 < q.value <= max_q and nq.value * m > tail_rows          # more than one stage
    chunk = 1 << q.value
    nchunks = (nb + chunk - 1) // chunk
    cuts = [0, (nchunks // 3) * chunk, (2 * nchunks // 3) * chunk, nb]
    partR, partL = ctx.to_device(np.zeros((nq.value + 1) * m)), ctx.to_device(np.zeros((nq.value + 1) * m))
    xq, out = ctx.alloc((nq.value + 1) * m), ctx.to_device(np.zeros(N))
    for r in range(3):
        c.check(c.lib.aggmg_coarse_chunk_forward_dev(c.handle, H.handle, bd.ptr.value + 8 * cuts[r] * m, cuts[r], cuts[r + 1],
                                                     partR.ptr, partL.ptr))
    c.check(c.lib.aggmg_coarse_boundary_solve_dev(c.handle, H.handle, partR.ptr, partL.ptr, xq.ptr))
    for r in range(3):
        c.check(c.lib.aggmg_coarse_chunk_backward_dev(c.handle, H.handle, bd.ptr.value + 8 * cuts[r] * m, cuts[r], cuts[r + 1],
                                                      xq.ptr, out.ptr.value + 8 * cuts[r] * m))
    assert np.array_equal(out.download(), x)
    H.free()


def test_probe_rejects_factorisation_that_needs_pivoting_across_blocks(mg):
    """cyclic reduction pivots inside its blocks only.  tridiag(1, eps, 1) with n even is well conditioned
    (cond ~ n), its 1 x 1 pivot blocks have condition number 1 -- the per-block monitor sees nothing -- and the
    elimination divides by eps: element growth 1/eps.  The probe solve of aggmg_hier_create measures the backward
    error, rejects the device factorisation (AUTO: host banded LU with partial pivoting, as accurate as the
    reference's UMFPACK; DEVICE_CR forced: UnsupportedError), and the solve stays at round-off."""
    from agglomerationmultigrid1d_amd import _lib
    n = 4096 + 512
    eps = 1e-9
    A = sp.diags([np.ones(n - 1), np.full(n, eps), np.ones(n - 1)], [-1, 0, 1], format="csc")
    rng = np.random.default_rng(11)
    b = rng.standard_normal(n)
    ctx, H = one_level(mg, A, mode=_lib.COARSE_AUTO)
    info = H.coarse_info()
    assert not info["on_device"] and info["probe_backward_error"] > 1e-10
    bd, xd, z = ctx.to_device(b), ctx.alloc(n), ctx.to_device(np.zeros(n))
    H.vcycle_dev(z, bd, xd, 0, 0, 1.0)
    x = xd.download()
    assert np.linalg.norm(A @ x - b) <= 1e-12 * np.linalg.norm(b)
    with pytest.raises(mg.UnsupportedError):
        one_level(mg, A, mode=_lib.COARSE_DEVICE_CR)
    # a well-scaled system of the same shape passes the probe with a backward error at round-off level
    A2 = sp.diags([np.ones(n - 1), np.full(n, 4.0), np.ones(n - 1)], [-1, 0, 1], format="csc")
    _, H2 = one_level(mg, A2, mode=_lib.COARSE_AUTO)
    info2 = H2.coarse_info()
    assert info2["on_device"] and 0.0 <= info2["probe_backward_error"] < 1e-14
