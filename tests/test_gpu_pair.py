"""Two levels per launch (csrc/pair_kernels.hpp): the small agglomerated levels of a hierarchy, levels k and k + 1 of
src/solvers.jl:28-37 / :41-47 with the hand-over in LDS.  The arithmetic of every element is that of the one-level
fused kernel, so a cycle with paired launches must equal the cycle with separate ones BIT FOR BIT -- and the oracle to
the usual tolerance."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mg():
    import agglomerationmultigrid1d_amd as m
    return m


def _cycles(mg, U, pair, ncyc=2, nPre=3, nPost=3, smoother="blockJac"):
    from agglomerationmultigrid1d_amd import _lib
    from agglomerationmultigrid1d_amd.uniform import build_device_hierarchy
    ctx = mg.Context(0)
    ctx.set_option(_lib.OPT_PAIR_LEVELS, 1 if pair else 0)
    H = build_device_hierarchy(U, ctx, smoother=smoother)
    b = U.rhs()
    N = len(b)
    bd = ctx.to_device(b)
    xa, xb = ctx.to_device(np.zeros(N)), ctx.alloc(N)
    for _ in range(ncyc):
        H.vcycle_dev(xa, bd, xb, nPre, nPost, 2.0 / 3.0)
        xa, xb = xb, xa
    x = xa.download()
    paired = H.paired_levels(nPre)
    # the multi-cycle entry point takes the same launches below the finest level
    H.vcycles_dev(ctx.to_device(np.zeros(N)), bd, xb, ncyc, nPre, nPost, 2.0 / 3.0)
    xl = xb.download()
    H.free()
    return x, xl, paired


@pytest.mark.parametrize("n,ratios,sweeps", [(4096, (4, 2, 2), (3, 3)), (2**15, (4, 2, 2), (3, 3)), (4096 + 512, (4, 2, 2), (1, 2)),
                                             (2**14, (4, 4, 4), (3, 3)), (2**13, (2, 2, 2, 2), (2, 1)), (3 * 2**10, (4, 2, 4), (4, 4)),
                                             (64, (4, 2, 2), (3, 3)), (2**16, (4, 2), (3, 3))])
def test_paired_launches_equal_separate_ones_bitwise(mg, n, ratios, sweeps):
    """config 3/4 shape and others: ratios 2 and 4 in both positions, tiles cut by the domain ends, a level smaller than one
    tile, five levels (levels 1 + 2 paired, level 3 on its own), three levels (nothing to pair: level 1 is followed by the
    coarsest)"""
    from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy
    U = UniformDgAggHierarchy(n, p=3, pAgg=1, ratios=ratios)
    nPre, nPost = sweeps
    x1, xl1, paired = _cycles(mg, U, True, 2, nPre, nPost)
    x0, xl0, none = _cycles(mg, U, False, 2, nPre, nPost)
    assert none == []
    assert paired == ([1] if len(ratios) >= 3 else []), paired
    assert np.array_equal(x1, x0), float(np.max(np.abs(x1 - x0)))
    assert np.array_equal(xl1, xl0) and np.array_equal(xl1, x1)


def test_paired_cycle_against_the_oracle(oracle, mg):
    o = oracle
    Ho, b = o.build_dg_agg_hierarchy(256, p=3, pAgg=1, nAgg=3, first=4)
    H = mg.MeshHierarchy.from_reference(Ho)
    assert H.paired_levels() == [1]
    x = mg.multigrid_v_cycle(H, np.zeros(len(b)), b)
    xr = o.multigrid_v_cycle(Ho, np.zeros(len(b)), b)
    A = Ho.mStiffness[0]
    assert np.linalg.norm(A @ (x - xr)) <= 1e-12 * np.linalg.norm(b)
    assert np.linalg.norm(x - xr) <= 1e-10 * np.linalg.norm(xr)


def test_levels_the_pair_kernels_do_not_take(mg):
    """block Gauss-Seidel levels and the preconditioned restriction keep one launch per level"""
    from agglomerationmultigrid1d_amd import _lib
    from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy, build_device_hierarchy
    U = UniformDgAggHierarchy(2048, p=3, pAgg=1, ratios=(4, 2, 2))
    ctx = mg.Context(0)
    Hg = build_device_hierarchy(U, ctx, smoother="blockGS")
    assert Hg.paired_levels() == []
    Hg.free()
    H = build_device_hierarchy(U, ctx)
    assert H.paired_levels() == [1] and H.paired_levels(9) == []     # (halo budget: at most 8 sweeps per launch)
    ctx.set_option(_lib.OPT_PAIR_LEVELS, 0)
    assert H.paired_levels() == []
    H.free()
