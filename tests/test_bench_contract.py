"""bench.py's contract, as far as it can be checked without a GPU: the evidence file roofline.traffic is read from holds EVERY
kernel key bench.py can ask for at the shapes it documents (a kernel renamed in the sources silently dropped out of it in
round 2), with plausible bytes per frame; the launcher parses its options; the workload names are the documented ones."""
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_traffic_file_has_every_kernel_the_bench_can_ask_for():
    b = _bench()
    tab = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    # (workload, dominant kernel, frame_len, stride) -> plausible measured bytes per unit (algorithmic .. 3x)
    asked = [("pipeline", "analyze", 1200, 480), ("config3", "pitch", 1200, 480), ("config2", "autocorr_lpc", 512, 512),
             ("config4", "burg_lags", 512, 512), ("config4", "burg_recursion", 512, 512),
             ("config4", "formant_resonances", 512, 512), ("config4", "tracker_chunked", 512, 512),
             ("pipeline", "burg_lags", 1200, 480), ("pipeline", "burg_recursion", 1200, 480),
             ("pipeline", "formant_resonances", 1200, 480), ("pipeline", "tracker_chunked", 1200, 480),
             ("frontend", "pcm16", 1200, 480)]
    for wl, dom, n, hop in asked:
        key = b.traffic_key(wl, dom, n, hop)
        assert key in tab, f"profiles/pmc_traffic.json has no entry '{key}' ({wl}: {dom})"
        e = tab[key]
        alg = b.ALG_BYTES[dom](n, hop, b.P)
        assert 0.5 * alg <= e["bytes_per_frame"] <= 3.0 * alg, (key, e["bytes_per_frame"], alg)
        assert e["commit"] and "pmc_counters.json" in e["source"]
        assert os.path.exists(os.path.join(ROOT, e["source"].split(" ")[0])), e["source"]
    # config 2 is the HBM-bound config: measured traffic == algorithmic traffic (4304 B/frame) within 3 %
    c2 = tab["autocorr_lpc_512"]["bytes_per_frame"]
    assert abs(c2 - 4304) <= 0.03 * 4304, c2
    # the vector-issue-bound kernels carry the SQ pass the headline's issue_frac is read from
    for key in ("analyze", "pitch"):
        assert 0.2 < tab[key]["valu_busy"] <= 1.0 and tab[key]["valu_insts_per_frame"] > 1000, (key, tab[key])


def test_bench_help_lists_the_workloads_and_shape_options():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, check=True).stdout
    for word in ("--gpus", "--steps", "--warmup", "--frame-len", "--hop", "--no-sub", "pipeline", "config2", "config3", "config4", "frontend"):
        assert word in out, word


def test_headline_roofline_is_the_executed_fraction():
    """The headline `frac` of the pitch / analyze line is built from the flops the kernel EXECUTES; the comparison with the
    reference's O(N^2) sums lives under its own key and cannot exceed-1 its way into `frac`."""
    b = _bench()
    prof = {"analyze": (3 * 178.0, 3), "burg_lags": (3 * 9.0, 3 * 18), "burg_recursion": (3 * 1.0, 3 * 18)}
    work = (3 * 4_500_000, 3 * 4_500_000 * 40, 3 * 4_500_000 * 24, 3 * 4_500_000 * 16_000)
    roof, hbm, kms, _ = b.roofline_for("pipeline", prof, work, 4_500_000, 1200, 480, 3)
    assert roof["kernel"] == "analyze" and roof["bound"] == "fp64_valu" and roof["unit"] == "TFLOP/s"
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12 and roof["frac"] < 0.3
    ref = roof["reference_sums_at_peak"]
    assert ref["flops_per_frame"] > 4 * roof["flops_per_frame"] and "NOT a roofline" in ref["meaning"]
    assert hbm["bound"] == "hbm" and hbm["algorithmic_bytes_per_frame"] == 480 * 8 + 16 + 104 + 104
    assert "issue_frac" in roof
