"""(a measurement script, not a test: it lives here because it imports the oracle, which tools/ may not)
Accuracy of the tail forms of the coarsest solve on the config-5 shaped hierarchy of tests/test_gpu_chain.py (coarsest level:
DG p = 0, n scalar rows, Neumann / Dirichlet-penalty ends): probe backward error and the V-cycle's distance to the oracle's,
parallel cyclic reduction against the register-blocked cyclic reduction (AGGMG_CR_PCR=0)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import aggmg_oracle as o
import agglomerationmultigrid1d_amd as mg

for n in (130, 500, 1000, 1024):
    Ho, b = o.build_cg_hierarchy(n, ps=(4, 2, 1), nDG=1, pDG=0)
    A = Ho.mStiffness[0]
    xr = o.multigrid_v_cycle(Ho, np.zeros(len(b)), b)
    Ac = Ho.mStiffness[-1]
    for pcr in ("1", "0"):
        os.environ["AGGMG_CR_PCR"] = pcr
        H = mg.MeshHierarchy.from_reference(Ho)
        info = H.coarse_info()
        x = mg.multigrid_v_cycle(H, np.zeros(len(b)), b)
        # the coarse solve on its own: a random right-hand side through a one-level hierarchy of the same operator
        H1 = mg.MeshHierarchy(None, [mg.DeviceOperator(Ac)], [], [])
        d = np.random.default_rng(1).standard_normal(Ac.shape[0])
        y = mg.multigrid_v_cycle(H1, np.zeros(len(d)), d, nPre=0, nPost=0, alpha=1.0)
        print(n, info["tail"], "probe %.2e" % info["probe_backward_error"],
              "| ||A(x-xr)||/||b|| %.2e" % (np.linalg.norm(A @ (x - xr)) / np.linalg.norm(b)),
              "| coarse alone: ||Ac y - d||/||d|| %.2e" % (np.linalg.norm(Ac @ y - d) / np.linalg.norm(d)),
              " ||Ac y - d||/(||Ac|| ||y||) %.2e" % (np.linalg.norm(Ac @ y - d) / (abs(Ac).sum(axis=1).max() * np.linalg.norm(y))), flush=True)
        H.free(); H1.free()

# the coarsest operator of the benchmarked DG hierarchy (config 3/4: DG p=3 -> agglomerated p=1, ratios 4, 2, 2) at 2^20 fine
# elements: 2^16 blocks of 2, chunk stages + a 256-block tail; right-hand sides: random, and constant (a large smooth solution)
from agglomerationmultigrid1d_amd import _lib
from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy
for E in (16, 20):
    U = UniformDgAggHierarchy(2 ** E, p=3, pAgg=1, ratios=(4, 2, 2))
    Ac = U.stiffness_csc(U.nlevels - 1).tocsr()
    N = Ac.shape[0]
    nA = abs(Ac).sum(axis=1).max()
    for pcr in ("1", "0"):
        os.environ["AGGMG_CR_PCR"] = pcr
        H1 = mg.MeshHierarchy(None, [mg.DeviceOperator(Ac.tocsc())], [], [])
        info = H1.coarse_info()
        out = []
        for name, d in (("random", np.random.default_rng(1).standard_normal(N)), ("ones", np.ones(N)),
                        ("smooth", np.cos(np.arange(N) * (np.pi / N)))):
            y = mg.multigrid_v_cycle(H1, np.zeros(N), d, nPre=0, nPost=0, alpha=1.0)
            r = np.linalg.norm(Ac @ y - d)
            out.append("%s: res/||d|| %.1e res/(|A||y|) %.1e" % (name, r / np.linalg.norm(d), r / (nA * np.linalg.norm(y))))
        print("2^%d fine, N_c = %d" % (E, N), info["tail"], info["tail_blocks"], "probe %.1e" % info["probe_backward_error"], " | ".join(out), flush=True)
        H1.free()
