"""Parity over the configuration space, not only at the benchmark shapes: hierarchies of randomly drawn shape -- the three
families the reference's constructors make (DG fine + agglomerated levels, DG p-chain, CG p-chain with optional DG and
agglomerated levels: src/mesh_heirarchy.jl:30-181, tests/*_heirarchy_test.jl), degrees 1 .. 8, sizes, sweep counts and
damping drawn from a seeded generator -- one V-cycle of the HIP path through the C ABI against the NumPy restatement:
||A (x - x_ref)|| <= 1e-12 ||b|| (the north-star tolerance).  Whatever kernel family a level ends up on (fused
block-tridiagonal, two levels per launch, chain, generic) is what gets checked."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mg():
    import agglomerationmultigrid1d_amd as m
    return m


def _draw(seed):
    rng = np.random.default_rng(1000 + seed)
    fam = ("dg_agg", "dg_p", "cg")[seed % 3]
    nPre, nPost = int(rng.integers(0, 5)), int(rng.integers(0, 5))
    alpha = float(rng.choice([0.5, 2.0 / 3.0, 0.8, 1.0]))
    if fam == "dg_agg":
        first = int(rng.choice([2, 4]))
        nAgg = int(rng.integers(1, 5))
        n = first * 2 ** (nAgg - 1) * int(rng.integers(2, 9))
        return fam, dict(n=n, p=int(rng.integers(1, 9)), pAgg=int(rng.choice([0, 1, 1])), nAgg=nAgg, first=first), nPre, nPost, alpha
    if fam == "dg_p":
        p0 = int(rng.integers(2, 9))
        ps = [p0]
        while ps[-1] > 1 and len(ps) < 4:
            ps.append(max(1, ps[-1] // 2))
        return fam, dict(n=int(rng.integers(4, 49)), ps=tuple(ps)), nPre, nPost, alpha
    p0 = int(rng.integers(1, 9))
    ps = [p0]
    while ps[-1] > 1 and len(ps) < 3:
        ps.append(max(1, ps[-1] // 2))
    nDG = int(rng.integers(0, 2))
    nAgg = int(rng.integers(0, 3)) if nDG else 0
    first = 2
    n = (first * 2 ** max(nAgg - 1, 0) if nAgg else 1) * int(rng.integers(3, 17))
    return fam, dict(n=n, ps=tuple(ps), nDG=nDG, pDG=(ps[-1] if nDG and rng.integers(0, 2) else None) if nDG else None,
                     nAgg=nAgg, pAgg=1, first=first), nPre, nPost, alpha


@pytest.mark.parametrize("seed", range(36))
def test_vcycle_parity_on_random_shapes(oracle, mg, seed):
    o = oracle
    fam, kw, nPre, nPost, alpha = _draw(seed)
    build = {"dg_agg": o.build_dg_agg_hierarchy, "dg_p": o.build_dg_p_hierarchy, "cg": o.build_cg_hierarchy}[fam]
    Ho, b = build(**kw)
    H = mg.MeshHierarchy.from_reference(Ho)
    x0 = o.splitmix_normal(len(b), seed)
    x = mg.multigrid_v_cycle(H, x0, b, nPre=nPre, nPost=nPost, alpha=alpha)
    xr = o.multigrid_v_cycle(Ho, x0, b, nPre=nPre, nPost=nPost, alpha=alpha)
    A = Ho.mStiffness[0]
    scale = np.linalg.norm(b) + np.linalg.norm(A @ x0)
    assert np.linalg.norm(A @ (x - xr)) <= 1e-12 * scale, (fam, kw, nPre, nPost, alpha, H.level_kinds())
    # and the device-resident form gives the bits of the host-pointer form
    ctx = H.ctx
    xd = mg.multigrid_v_cycle(H, ctx.to_device(x0), ctx.to_device(b), nPre=nPre, nPost=nPost, alpha=alpha)
    assert np.array_equal(xd.download(), x)
