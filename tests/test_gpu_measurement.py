"""The measurement aids of the C ABI: compulsory bytes of a launch (aggmg_hier_launch_bytes, aggmg_smoother_launch_bytes)
-- the numerator of bench.py's roofline.frac -- against the array sizes the layouts of DESIGN.md section 3 give."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mg():
    import agglomerationmultigrid1d_amd as m
    return m


def test_compulsory_bytes_of_the_fused_launches(mg):
    """config 3/4 hierarchy (DG p = 3 -> AggDG 4:1 -> 2:1 -> 2:1): per fine element the descent reads the packed symmetric
    inverse (80 B), q row, b, x0, the second column of L (32 B each), and for the explicit residual the diagonal block
    (128 B) and the sub-diagonal column (32 B); it writes the iterate (32 B) and a quarter of a two-mode coarse element
    (4 B).  The ascent reads inverse, q row, b, u, L column and the coarse iterate, writes the iterate."""
    from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy, build_device_hierarchy
    ne = 4096
    U = UniformDgAggHierarchy(ne, p=3, pAgg=1, ratios=(4, 2, 2))
    H = build_device_hierarchy(U)
    assert H.level_kinds() == ["fused_btd"] * 3 + ["coarsest"]
    assert H.launch_bytes(0, "down") == ((80 + 32 + 32 + 32 + 32 + 128 + 32) * ne, (32 + 4) * ne)
    assert H.launch_bytes(0, "up") == ((80 + 32 + 32 + 32 + 32 + 4) * ne, 32 * ne)
    # between two cycles of aggmg_vcycles_dev: both halves on one pass over the operator
    rd, wr = H.launch_bytes(0, "mid")
    assert (rd, wr) == ((80 + 32 + 32 + 32 + 32 + 128 + 32 + 4) * ne, (32 + 4) * ne)
    # level 1: agglomerated, m = 2, dense off-diagonal blocks, symmetric: packed inverse 24 B, super-diagonal block 32 B,
    # b 16 B per element for the sweeps; diagonal + sub-diagonal block 64 B for the residual; rows of L 32 B (16 B where
    # the first column is the stored-as-implied unit column)
    n1 = ne // 4
    assert H.launch_bytes(1, "down") in [((24 + 32 + 16 + 64 + lrow) * n1, (16 + 8) * n1) for lrow in (16, 32)]
    assert H.launch_bytes(1, "up") in [((24 + 32 + 16 + 16 + lrow + 8) * n1, 16 * n1) for lrow in (16, 32)]
    with pytest.raises(mg.ArgumentError):
        H.launch_bytes(3, "down")      # the coarsest level has no fused launch
    H.free()


def test_compulsory_bytes_of_standalone_launches(mg):
    """aggmg_smooth_dev / aggmg_residual_dev on the config-2 operator: block-tridiagonal form and generic CSR"""
    from agglomerationmultigrid1d_amd import _lib
    from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy
    ne = 2048
    U = UniformDgAggHierarchy(ne, p=3, pAgg=1, ratios=())
    op = mg.DeviceOperator(U.stiffness_csc(0), _lib.OP_STIFFNESS)
    S = mg.BlockJacobi(op, U.descriptor(0).mBlockInds)
    assert mg.smoother_launch_bytes(op, S, "sweeps") == ((80 + 32 + 32 + 32) * ne, 32 * ne)
    assert mg.smoother_launch_bytes(op, None, "residual") == ((32 + 32 + 32 + 128 + 32) * ne, 32 * ne)
    op2 = mg.DeviceOperator(U.stiffness_csc(0), _lib.OP_STIFFNESS)
    J = mg.JacobiSmoother(op2, detect=False)
    N, nnz = op2.shape[0], op2.nnz
    assert mg.smoother_launch_bytes(op2, J, "sweeps") == (12 * nnz + 4 * (N + 1) + 24 * N, 8 * N)
    assert mg.smoother_launch_bytes(op2, None, "residual") == (12 * nnz + 4 * (N + 1) + 16 * N, 8 * N)
    assert np.isfinite(nnz)
