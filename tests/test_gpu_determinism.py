"""Race detection by deterministic re-run (no GPU sanitizer on this pool): every kernel family of the
path runs the same cycles twice from scratch -- fresh context, fresh uploads, fresh set-up -- and the results must agree BIT
FOR BIT.  A data race, an uninitialised read or an order-dependent accumulation (atomics) shows up as a difference; since r04
no kernel on a sweep / cycle path accumulates with atomics (the generic block smoothers combine overlapping blocks in list
order), so all of them are held to it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mg():
    import agglomerationmultigrid1d_amd as m
    return m


def _run(mg, build, ncyc=3, nPre=3, nPost=3, alpha=2.0 / 3.0, solver=False):
    ctx = mg.Context(0)
    H, b = build(ctx)
    N = len(b)
    bd = ctx.to_device(b)
    xa, xb = ctx.to_device(np.zeros(N)), ctx.alloc(N)
    for _ in range(ncyc):
        H.vcycle_dev(xa, bd, xb, nPre, nPost, alpha)
        xa, xb = xb, xa
    out = [xa.download()]
    H.vcycles_dev(ctx.to_device(np.zeros(N)), bd, xb, ncyc, nPre, nPost, alpha)
    out.append(xb.download())
    if solver:
        x, it, res = mg.multigrid_dev(H, ctx.to_device(np.zeros(N)), bd, 12, 1e-30, check_every=3)
        out += [x.download(), np.asarray(res)]
    kinds = H.level_kinds()
    H.free()
    return out, kinds


def _dg(n, ratios, smoother="blockJac"):
    def build(ctx):
        from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy, build_device_hierarchy
        U = UniformDgAggHierarchy(n, p=3, pAgg=1, ratios=ratios)
        return build_device_hierarchy(U, ctx, smoother=smoother), U.rhs()
    return build


def _ragged(n):
    def build(ctx):
        from agglomerationmultigrid1d_amd.uniform import build_device_ragged_hierarchy
        H, b, _ = build_device_ragged_hierarchy(n, ctx, p=3, seed=3)
        return H, b
    return build


def _cg(n, ps, smoother="jac", chain=True, permute=False):
    def build(ctx):
        from agglomerationmultigrid1d_amd import _lib
        from agglomerationmultigrid1d_amd.uniform import UniformCgDgHierarchy, build_device_cg_hierarchy
        U = UniformCgDgHierarchy(n, ps=ps)
        if not permute:
            return build_device_cg_hierarchy(U, ctx, chain=chain, smoother=smoother), U.rhs()
        # element Schwarz blocks listed in a random order: the generic one-pass block sweep (off the chain kernel)
        cls = {"addSchwarz": mg_mod.AdditiveSchwarzSmoother, "hybridSchwarz": mg_mod.HybridSchwarzSmoother}[smoother]
        ops = [mg_mod.DeviceOperator(A, _lib.OP_STIFFNESS, ctx) for A in U.A]
        sms = []
        for k in range(U.nlevels - 1):
            el = U.element_nodes(k)
            perm = np.random.default_rng(k).permutation(el.shape[1])
            sms.append(cls(ops[k], np.ascontiguousarray(el[:, perm]), ctx))
            assert not sms[-1].structured
        Ls = [mg_mod.DeviceOperator(L, _lib.OP_TRANSFER, ctx) for L in U.L]
        return mg_mod.MeshHierarchy(None, ops, sms, Ls, ctx=ctx), U.rhs()
    import agglomerationmultigrid1d_amd as mg_mod
    return build


CASES = {
    "dg_agg_fused_and_paired": (_dg(2**15, (4, 2, 2)), ["fused_btd"] * 3 + ["coarsest"], True),
    "dg_agg_block_gs": (_dg(2**13, (4, 2, 2), "blockGS"), ["fused_btd"] * 3 + ["coarsest"], False),
    "dg_agg_ragged": (_ragged(2**13), ["fused_btd"] * 3 + ["coarsest"], False),
    "cg_chain_jacobi": (_cg(2**13, (4, 2, 1)), ["fused_chain"] * 3 + ["coarsest"], True),
    "cg_chain_hybrid_schwarz": (_cg(2**12, (4, 2, 1), "hybridSchwarz"), ["fused_chain"] * 3 + ["coarsest"], False),
    "cg_generic_csr": (_cg(2**12, (4, 2, 1), chain=False), ["generic"] * 3 + ["coarsest"], True),
    "cg_generic_additive_schwarz": (_cg(2**12, (4, 2), "addSchwarz", permute=True), ["generic"] * 2 + ["coarsest"], False),
    "cg_generic_hybrid_schwarz": (_cg(2**12, (4, 2), "hybridSchwarz", permute=True), ["generic"] * 2 + ["coarsest"], False),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_two_runs_from_scratch_agree_bitwise(mg, name):
    build, kinds_want, solver = CASES[name]
    a, kinds = _run(mg, build, solver=solver)
    b, _ = _run(mg, build, solver=solver)
    assert kinds == kinds_want, kinds
    for u, v in zip(a, b):
        assert np.all(np.isfinite(u)) and np.array_equal(u, v), (name, float(np.max(np.abs(u - v))))


@pytest.mark.parametrize("kind", ["addSchwarz", "hybridSchwarz"])
def test_apply_smoother_on_overlapping_lists_is_reproducible(mg, kind):
    """apply_smoother of the Schwarz smoothers (src/smoother.jl:6-46) adds the results of overlapping blocks per row in list
    order (r03: atomic adds into a zeroed vector): two contexts, blocks listed in different random orders -- the same bits
    (the blocks are sorted once, so the listing order does not matter either)"""
    from agglomerationmultigrid1d_amd import _lib
    from agglomerationmultigrid1d_amd.uniform import UniformCgDgHierarchy
    U = UniformCgDgHierarchy(2**12, ps=(4,))
    A, el = U.A[0], U.element_nodes(0)
    B = np.random.default_rng(0).standard_normal((A.shape[0], 3))
    outs = []
    for seed in (1, 2):
        ctx = mg.Context(0)
        op = mg.DeviceOperator(A, _lib.OP_STIFFNESS, ctx)
        perm = np.random.default_rng(seed).permutation(el.shape[1])
        cls = {"addSchwarz": mg.AdditiveSchwarzSmoother, "hybridSchwarz": mg.HybridSchwarzSmoother}[kind]
        S = cls(op, np.ascontiguousarray(el[:, perm]), ctx)
        outs.append(mg.apply_smoother(S, B, alpha=0.7))
    assert np.array_equal(outs[0], outs[1])
