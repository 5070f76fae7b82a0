#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (oracle/aggmg_oracle.py).

RESTATEMENT-DERIVED FIXTURES: the reference (Julia) was not executed -- it cannot run in the build
image and commits no expected values of its own.  Each file holds the inputs of
one small hot-path case (operators as Julia-style 1-based CSC triples, mBlockInds, x0, b) and the
oracle's outputs (V-cycle result, smoothed iterates, residual, restricted / prolonged vectors).
Re-run with:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
import aggmg_oracle as o  # noqa: E402


def pack(H, b, x0, name, nPre=3, nPost=3, alpha=2.0 / 3.0):
    d = {"nlevels": len(H.mStiffness), "nPre": nPre, "nPost": nPost, "alpha": alpha, "x0": x0, "b": b}
    for k, A in enumerate(H.mStiffness):
        cp, rv, nz = o.julia_csc(A)
        d[f"A{k}_colptr"], d[f"A{k}_rowval"], d[f"A{k}_nzval"] = cp, rv, nz
        d[f"A{k}_shape"] = np.array(A.shape)
    for k, L in enumerate(H.mInterpolation):
        cp, rv, nz = o.julia_csc(L)
        d[f"L{k}_colptr"], d[f"L{k}_rowval"], d[f"L{k}_nzval"] = cp, rv, nz
        d[f"L{k}_shape"] = np.array(L.shape)
    for k in range(len(H.mStiffness) - 1):
        S = H.mSmoothers[k]
        d[f"S{k}_inds"] = S.mBlockInds if hasattr(S, "mBlockInds") else np.zeros((0, 0), dtype=np.int64)
    # outputs
    d["x_vcycle"] = o.multigrid_v_cycle(H, x0, b, nPre=nPre, nPost=nPost, alpha=alpha)
    A0, S0 = H.mStiffness[0], H.mSmoothers[0]
    u = x0.copy()
    for s in range(1, 4):
        u = u + o.apply_smoother(S0, b - o.csc_matvec(A0, u), alpha=alpha)
        d[f"u_after_{s}_sweeps"] = u
    r = b - o.csc_matvec(A0, x0)
    d["residual_x0"] = r
    d["restricted"] = o.csc_adjoint_matvec(H.mInterpolation[0], r)
    uc = o.splitmix_normal(H.mInterpolation[0].shape[1], 3)
    d["uc"] = uc
    d["prolonged"] = x0 + o.csc_matvec(H.mInterpolation[0], uc)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    print(name, "levels", d["nlevels"], "fine dofs", len(b))


def main():
    H, b = o.build_dg_agg_hierarchy(32, p=3, pAgg=1, nAgg=3, first=4)           # config 3 shape
    pack(H, b, o.splitmix_normal(len(b), 0), "dg_p3_agg_4level_n32")
    H, b = o.build_dg_p_hierarchy(16, ps=(4, 2, 1))                              # dg_heirarchy_test shape
    pack(H, b, np.zeros(len(b)), "dg_pchain_4_2_1_n16")
    H, b = o.build_cg_hierarchy(64, ps=(1,), nDG=1, pDG=0)                       # config 1 shape
    pack(H, b, np.zeros(len(b)), "cg_p1_dg0_n64")
    H, b = o.build_cg_hierarchy(16, ps=(4, 2, 1), nAgg=3)                        # full_heirarchy_test shape
    pack(H, b, o.splitmix_normal(len(b), 1), "cg_4_2_1_agg3_n16", nPre=2, nPost=1, alpha=0.5)


if __name__ == "__main__":
    main()
