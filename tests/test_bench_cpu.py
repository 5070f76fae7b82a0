"""bench.py's bookkeeping that needs no GPU: the roofline object (frac from compulsory bytes, a fraction), the PMC traffic
table the default workload looks up, the argument surface the driver uses."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_roofline_object_is_a_fraction_of_the_hbm_peak():
    b = _bench()
    comp = (6174015488, 603979776)          # the fine-level descent at 2^24 elements: arrays read / written once
    r = b._roofline("k", comp, 1.17, 20, 35534143124, 7046739748.0, 1.2148, "note")
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert abs(r["achieved"] - sum(comp) / 1.17e-3 / 1e9) < 1e-6 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12
    assert 0.0 < r["frac"] <= 1.0                                   # compulsory bytes over the duration: a fraction
    assert r["frac"] <= r["physical_frac"] <= 1.1 * r["frac"]       # counter bytes: the same plus a few per cent of halo re-reads
    assert r["frac_survey_model"] > 1.0                             # the SURVEY-8d CSR model is not a bound: kept apart
    assert abs(r["traffic_over_compulsory"] - 7046739748.0 / sum(comp)) < 1e-12
    r0 = b._roofline("k", comp, 1.17, 20, 1.0, None, None, "")     # no profile: traffic stays null, frac is still there
    assert r0["traffic"] is None and r0["physical_frac"] is None and r0["frac"] == r["frac"]


def test_traffic_table_covers_the_default_workload():
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    need = ["fused_down_L0_dg_log2n24", "fused_up_L0_dg_log2n24", "pair_down_L1_dg_log2n24", "pair_up_L1_dg_log2n24",
            "chain_down_L0_cg_log2n24", "coarse_dg_log2n24", "coarse_cg_log2n24"]
    for k in need:
        assert isinstance(t.get(k), (int, float)) and t[k] > 0, k
        assert t.get(k + "_ms_profile_mean", 1.0) > 0
    for k in ("sweeps_1_per_launch", "residual", "sweeps_4_per_launch", "sweeps_8_per_launch"):
        for c in (1, 3):
            e = t.get(f"smoother_{k}_dg_log2n20_copies{c}")
            assert isinstance(e, dict) and e["hbm_bytes"] > 0 and e["ms_profile_mean"] > 0, (k, c)
    # the headline launch: counter bytes within 10 % of what its arrays hold (368 + 36 bytes per fine element)
    assert 1.0 <= t["fused_down_L0_dg_log2n24"] / (404 * 2**24) <= 1.10


def test_argument_surface_of_the_driver_contract(monkeypatch):
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = b.parse()
    assert (a.gpus, a.log2_elems) == (1, 24) and a.steps >= 1 and a.warmup >= 0 and not a.rehearse_threads
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "2"])
    a = b.parse()
    assert (a.gpus, a.steps, a.warmup) == (8, 5, 2) and a.cpu_log2_elems == 0    # cpu_baseline at the size of `value`
