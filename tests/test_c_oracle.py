"""The plain-C restatement (oracle/aggmg_oracle_c.c, the CPU-baseline port) against the
NumPy/SciPy restatement: same operation order, so smoothing agrees to the last bits and the
V-cycle to the coarse-solver round-off."""
import numpy as np


def test_c_oracle_matches_python_oracle(oracle):
    import c_oracle
    o = oracle
    for Ho, b in (o.build_dg_agg_hierarchy(64, p=3, pAgg=1, nAgg=3), o.build_cg_hierarchy(32, ps=(2, 1), nDG=1, pDG=0),
                  o.build_dg_p_hierarchy(16, ps=(4, 2, 1))):
        C = c_oracle.from_oracle_hierarchy(Ho)
        x0 = o.splitmix_normal(len(b), 0)
        xs, _ = C.smooth(0, x0, b, 2.0 / 3.0, 5)
        ref = x0.copy()
        for _ in range(5):
            ref = ref + o.apply_smoother(Ho.mSmoothers[0], b - o.csc_matvec(Ho.mStiffness[0], ref), alpha=2.0 / 3.0)
        assert np.linalg.norm(xs - ref) <= 1e-14 * np.linalg.norm(ref)
        x, dt, cs = C.vcycle(np.zeros(len(b)), b)
        xr = o.multigrid_v_cycle(Ho, np.zeros(len(b)), b)
        A = Ho.mStiffness[0]
        assert np.linalg.norm(A @ (x - xr)) <= 1e-12 * np.linalg.norm(b)
        assert np.linalg.norm(x - xr) <= 1e-9 * np.linalg.norm(xr)
        assert 0.0 <= cs <= dt
        C.enable_omp(Ho.mStiffness, Ho.mInterpolation)      # OpenMP row-gather variant, same cycle
        xo, _, _ = C.vcycle_omp(np.zeros(len(b)), b)
        assert np.linalg.norm(A @ (xo - x)) <= 1e-12 * np.linalg.norm(b)
