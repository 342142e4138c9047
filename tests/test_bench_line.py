"""bench.py's roofline arithmetic on CPU: the fields of a bench line recompute from each other with the peaks of
MI355X_MICROARCH.md (the pipe the kernel issues on), and the recorded driver-style line of the round is self-consistent."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_pipe_peaks_by_kernel_and_mode():
    b = _bench()
    assert b.kernel_peak("sfcx_bwd_data", "split") == (2500.0 / 5, 5)
    assert b.kernel_peak("gemmx_group_nk_edge", "bf16") == (2500.0, 1)
    assert b.kernel_peak("sfcx_fwd", "split6") == (2500.0 / 6, 6)
    # exact-fp32 kernels (fp32 mode, and everything that is not on the bf16 pipe) are priced against the fp32 MFMA peak
    assert b.kernel_peak("sfc_bwd_data", "fp32") == (157.3, 0)
    assert b.kernel_peak("gemm_group_kn", "split") == (157.3, 0)


def test_roofline_object_recomputes():
    b = _bench()
    launches, ms, flops_per = 260, 44.2, 9.2e9
    prof = {"sfcx_bwd_data": dict(launches=launches, total_ms=ms, flops=flops_per * launches, bytes=1.8e8 * launches),
            "sfcx_fwd": dict(launches=launches, total_ms=40.0, flops=flops_per * launches, bytes=1.0e8 * launches),
            "attn_fwd": dict(launches=120, total_ms=90.0, flops=0.0, bytes=5e7 * 120)}  # no flops: never the roofline kernel
    r = b.roofline_of(prof, dt_s=0.22, mode="split", with_pmc=False)
    assert r["kernel"] == "sfcx_bwd_data" and r["bound"] == "mfma" and r["unit"] == "TFLOP/s"
    assert r["achieved"] == pytest.approx(flops_per / (ms / launches * 1e-3) / 1e12)
    assert r["peak"] == 500.0 and r["frac"] == pytest.approx(r["achieved"] / r["peak"])
    assert r["frac_of_fp32_peak"] == pytest.approx(r["achieved"] / 157.3)
    # matrix-pipe occupancy: 5 plane products per algorithmic product, 32 768 flops and 32 cycles per instruction, 1 024 SIMDs
    insts = flops_per * 5 / 32768.0
    assert r["mfma_busy"] == pytest.approx(insts * 32 / 1024 / 2.4e9 / (ms / launches * 1e-3))
    # the same statement as `frac` when only algorithmic products are issued (2 500 TF/s is the guide's rounding of
    # 32 768 flop / 32 cycles x 1 024 SIMDs x 2.4 GHz = 2 517)
    assert r["mfma_busy"] == pytest.approx(r["frac"], rel=1e-2)
    assert r["share_of_step"] == pytest.approx(ms / 220.0)
    assert r["traffic"] is None and r["traffic_over_algorithmic"] is None
    rb = b.roofline_of(prof, dt_s=0.22, mode="bf16", with_pmc=False)
    assert rb["peak"] == 2500.0 and rb["frac"] == pytest.approx(rb["achieved"] / 2500.0)


def test_recorded_driver_style_line_is_self_consistent():
    path = os.path.join(ROOT, "profiles", "r05", "r05_drv_bench_default.json")
    d = json.loads(open(path).read().strip().splitlines()[-1])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["unit"] == "molecules/s" and d["n_gpus"] == 1 and d["higher_is_better"] and d["scaling"] == "weak"
    assert d["metric"].split()[0] in json.dumps(base)  # BASELINE.json's metric
    assert d["value"] == pytest.approx(d["config"]["global_batch"] / (d["ms_per_step"] * 1e-3), rel=1e-9)
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f32"
    r = d["roofline"]
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"]) and r["peak"] == 500.0
    assert r["achieved"] == pytest.approx(r["flops_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e12)
    assert r["traffic_over_algorithmic"] == pytest.approx(r["traffic"] / r["algorithmic_bytes_per_launch"])
    assert 0.0 < r["mfma_busy"] < 1.0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert d["value"] / c["value"] > 100  # reported, not a target
    assert d["spread"]["values"][0] == d["value"] and len(d["spread"]["values"]) == 3
    subs = {(s["workload"], s["matrix_mode"]) for s in d["configs"]}
    assert subs == {("qm9", "bf16"), ("qm9", "fp32"), ("oc20", "split"), ("md17_l2", "split"), ("md17_l3", "split")}
    fp32 = [s for s in d["configs"] if s["matrix_mode"] == "fp32"][0]
    assert fp32["dominant"]["peak"] == 157.3 and fp32["dtype"] == "f32"  # exact-fp32 MFMA everywhere: priced against its own peak
    # the two kernels north_star names carry counter evidence of the same build (tools/gpu_profile.sh -> profiles/pmc_dominant.json)
    ns = d["north_star_kernels"]
    assert ns["scatter"]["traffic_over_algorithmic"] == pytest.approx(ns["scatter"]["traffic"] / ns["scatter"]["algorithmic_bytes_per_launch"])
    assert 0.0 < ns["radial_mlp"]["widest_layer_kernel"]["mfma_busy"] < 1.0
    assert d["config"]["deferred_weight_gradients"]["queued"] > 0
    assert [s for s in d["configs"] if s["matrix_mode"] == "bf16"][0]["dtype"] == "bf16"
