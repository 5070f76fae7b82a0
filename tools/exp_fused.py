#!/usr/bin/env python3
"""Kernel tuning experiment: time the structured smoother launches (S sweeps per launch) and the
V-cycle's fused launches at 2^E fine elements for the library named by AGGMG_HIP_LIB."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import agglomerationmultigrid1d_amd as mg
from agglomerationmultigrid1d_amd.api import _ptr
from agglomerationmultigrid1d_amd.uniform import UniformDgAggHierarchy, build_device_hierarchy

E = int(sys.argv[1]) if len(sys.argv) > 1 else 22
ctx = mg.Context(0)
U = UniformDgAggHierarchy(2**E, p=3, pAgg=1, ratios=(4, 2, 2))
H = build_device_hierarchy(U, ctx)
N = 4 * 2**E
b = ctx.to_device(U.rhs())
u = ctx.to_device(np.zeros(N))
v = ctx.alloc(N)
op, S = H._ops[0], H.mSmoothers[0]
out = {"lib": os.environ.get("AGGMG_HIP_LIB", "default")}
for ns in (0, 1, 3, 6):
    for _ in range(3):
        ctx.check(ctx.lib.aggmg_smooth_dev(ctx.handle, op.handle, S.handle, _ptr(u), _ptr(b), 2 / 3, ns, _ptr(v)))
    ctx.synchronize()
    t0 = time.perf_counter()
    R = 30
    for _ in range(R):
        ctx.check(ctx.lib.aggmg_smooth_dev(ctx.handle, op.handle, S.handle, _ptr(u), _ptr(b), 2 / 3, ns, _ptr(v)))
    ctx.synchronize()
    out[f"smooth{ns}_us"] = round(1e6 * (time.perf_counter() - t0) / R, 1)
    if ns == 0:   # S = 0 is a plain D2D copy of the iterate: the box's copy ceiling
        out["copy_TBs"] = round(2 * 8 * N / (out["smooth0_us"] * 1e-6) / 1e12, 2)
xa, xb = u, v
for _ in range(3):
    H.vcycle_dev(xa, b, xb)
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    H.vcycle_dev(xa, b, xb)
    xa, xb = xb, xa
ctx.synchronize()
out["vcycle_noevents_us"] = round(1e6 * (time.perf_counter() - t0) / 20, 1)
ctx.profile_enable(True)
t0 = time.perf_counter()
for _ in range(20):
    H.vcycle_dev(xa, b, xb)
    xa, xb = xb, xa
ctx.synchronize()
out["vcycle_us"] = round(1e6 * (time.perf_counter() - t0) / 20, 1)
ctx.profile_enable(False)
for (k, l), (ms, c) in sorted(ctx.profile_collect().items()):
    out[f"{k}_L{l}_us"] = round(1e3 * ms / c, 1)
for _ in range(2):
    H.vcycles_dev(xa, b, xb, 20)
ctx.synchronize()
ctx.profile_enable(True)
t0 = time.perf_counter()
H.vcycles_dev(xa, b, xb, 20)
ctx.synchronize()
out["vcycles20_us_per_cycle"] = round(1e6 * (time.perf_counter() - t0) / 20, 1)
ctx.profile_enable(False)
for (k, l), (ms, c) in sorted(ctx.profile_collect().items()):
    if k == "fused_mid":
        out[f"{k}_L{l}_us"] = round(1e3 * ms / c, 1)
print(json.dumps(out))
