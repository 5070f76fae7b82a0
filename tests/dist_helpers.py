"""Helpers for the multi-rank tests: a NumPy engine (oracle arithmetic, tests only) that plugs into
agglomerationmultigrid1d_amd.distributed.DistributedVCycle in place of the GPU engine."""
import numpy as np
import torch


class LocalRef:
    """container with the reference's field names for oracle.multigrid_v_cycle"""

    def __init__(self, o, U):
        n = U.nlevels
        self.mMeshes = [None] * n
        self.mStiffness = [U.stiffness_csc(k) for k in range(n)]
        self.mInterpolation = [U.interpolation_csc(k) for k in range(n - 1)]
        self.mSmoothers = []
        for k in range(n - 1):
            A = self.mStiffness[k]
            inds = U.descriptor(k).mBlockInds
            blocks = [o.LU(A[np.ix_(inds[:, i] - 1, inds[:, i] - 1)].toarray()) for i in range(inds.shape[1])]
            self.mSmoothers.append(o.BlockJacobi(blocks, inds))


class NumpyEngine:
    """The engine interface of DistributedVCycle on the CPU with the oracle's operations
    (src/solvers.jl:28-37 / :41-47 split around the coarsest solve)."""

    def __init__(self, o, H_local, A_coarse_global):
        self.o, self.H, self.Ac = o, H_local, A_coarse_global
        self.u = [None] * len(H_local.mStiffness)
        self.rhs = [None] * len(H_local.mStiffness)

    def new(self, n):
        return torch.zeros(int(n), dtype=torch.float64)

    def owned_sumsq(self, ranges, b, x=None):
        v = b.numpy()
        if x is not None:
            v = v - self.o.csc_matvec(self.H.mStiffness[0], x.numpy())
        return float(sum(np.dot(v[lo:hi], v[lo:hi]) for lo, hi in ranges))

    def down(self, x0, b, nPre, alpha):
        o, H = self.o, self.H
        n = len(H.mStiffness)
        self.u[0] = x0.numpy().copy()
        self.rhs[0] = b.numpy().copy()
        for k in range(n - 1):
            if k > 0:
                self.u[k] = np.zeros(H.mStiffness[k].shape[1])
            for _ in range(nPre):
                self.u[k] = self.u[k] + o.apply_smoother(
                    H.mSmoothers[k], self.rhs[k] - o.csc_matvec(H.mStiffness[k], self.u[k]), alpha=alpha)
            self.rhs[k + 1] = o.csc_adjoint_matvec(H.mInterpolation[k],
                                                   self.rhs[k] - o.csc_matvec(H.mStiffness[k], self.u[k]))

    def coarse_rhs(self):
        return torch.from_numpy(self.rhs[-1])

    def coarse_solve(self, rhs_global):
        return torch.from_numpy(self.o.sparse_direct_solve(self.Ac, rhs_global.numpy()))

    def set_coarse_solution(self, t):
        self.u[-1] = t.numpy().copy()

    # split ascent / second stream: the schedule's bookkeeping is exercised, the work is not split
    def can_split_up(self, nPost):
        return True

    def up_split(self, b, x_out, nPost, alpha, head, tail, part):
        if part == 1:     # everything with the "ends" call: it is the one the exchange depends on
            self.up(b, x_out, nPost, alpha)

    def mark_side_ends(self):
        pass

    def wait_side_ends(self):
        pass

    def side_stream(self):
        import contextlib
        return contextlib.nullcontext()

    def wait_side_stream(self):
        pass

    def up(self, b, x_out, nPost, alpha):
        o, H = self.o, self.H
        for k in range(len(H.mStiffness) - 2, -1, -1):
            self.u[k] = self.u[k] + o.csc_matvec(H.mInterpolation[k], self.u[k + 1])
            for _ in range(nPost):
                self.u[k] = self.u[k] + o.apply_smoother(
                    H.mSmoothers[k], self.rhs[k] - o.csc_matvec(H.mStiffness[k], self.u[k]), alpha=alpha)
        x_out.copy_(torch.from_numpy(self.u[0]))
