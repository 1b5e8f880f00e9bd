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


def _canned_record(b, blowup=1):
    """A full record of the shape run_rank assembles (round 5's default run: 14 pipeline shapes, speech, config 5 whole), with
    `blowup` x the per-shape tables -- the part that grew 3.9 -> 25 KB over rounds 2-5."""
    prof = {"analyze": (3 * 125.0, 3), "burg_lags": (3 * 2.7, 3), "burg_recursion": (3 * 8.0, 3), "formant_resonances": (3 * 15.3, 3),
            "tracker_chunked": (3 * 8.1, 3), "burg_direct_list": (1.1, 3), "pitch_direct_fallback": (0.02, 3)}
    work = (3 * 4_500_000, 3 * 4_500_000 * 43, 3 * 4_500_000 * 21, 3 * 4_500_000 * 14_824)
    roof, hbm, kms, _ = b.roofline_for("pipeline", prof, work, 4_500_000, 1200, 480, 3)
    shapes = [{"frame_len": n, "hop": h, "frames": 1000, "value": 1.23456789e7, "ms_per_step": 1.0, "dominant_kernel": "analyze",
               "kernels_ms": dict(kms), "beside_it": sorted(kms)} for n, h in b.PIPELINE_SHAPES] * blowup
    sub = [{"name": f"config3_kmax{k}", "workload": "x" * 200, "value": 4.2e7, "roofline": roof, "kernels_ms": kms} for k in (1, 8, 302)]
    sub += [{"name": "config2", "value": 1.15e9, "roofline": hbm}, {"name": "config4", "value": 4.6e8, "roofline": hbm},
            {"name": "pipeline_shapes", "shapes": shapes},
            {"name": "speech_44k", "shapes": [{"frame_len": n, "hop": h, "speech": {"value": 2.4e7, "burg_direct": 0.4, "kernels_ms": kms},
                                               "synthetic": {"value": 2.8e7, "burg_direct": 0.01}} for n, h in b.SPEECH_SHAPES]},
            {"name": "config5_100h_1gpu", "skipped": "RuntimeError('HIP out of memory')" * 20}]
    return {"metric": b.METRIC, "value": 35971000.123456, "unit": "frames/s", "n_gpus": 1, "steps": 3, "warmup": 1, "ms_per_step": 125.1,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "full pitch+LPC+formants+MFCC pipeline, 12.5 h/GPU synthetic 48 kHz, 25 ms / 10 ms hop", "frames_per_gpu": 4500000,
                       "frame_len": 1200, "hop": 480, "parallelism": "frame-range split x1, one process per GPU, RCCL gather of the records to rank 0",
                       "gather": "g" * 400},
            "roofline": roof, "roofline_hbm": hbm, "kernels_ms": kms, "parity": b.PARITY_NOTE, "sub_benchmarks": sub,
            "cpu_baseline": {"value": 1641.05, "unit": "frames/s", "cores": 16, "kind": "port", "sample": "s" * 600,
                             "one_core": {"value": 101.2, "unit": "frames/s", "cores": 1, "frames": 811, "seconds": 8.0}}}


def test_the_stdout_line_stays_under_the_drivers_window():
    """Round 5: the line reached 25 KB, the driver keeps the last 8 KB of stdout, BENCH_r05.parsed = null.  The line bench.py prints is
    assembled by compact_line() from the full record; it must stay under LINE_LIMIT whatever the sub-benchmarks carry, keep every
    contract key, `roofline` and `cpu_baseline`, and reduce each sub-benchmark to numbers."""
    b = _bench()
    assert b.LINE_LIMIT <= 6000
    for blowup in (1, 40):
        rec = _canned_record(b, blowup)
        assert len(json.dumps(rec)) > 3 * b.LINE_LIMIT                       # the full record is what used to be printed
        text = b.compact_line(rec, os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
        assert "\n" not in text and len(text) < b.LINE_LIMIT, len(text)
        d = json.loads(text)
        for k in b.CONTRACT_KEYS + ("config", "roofline", "cpu_baseline"):
            assert k in d, k
        assert d["metric"] == b.METRIC and d["dtype"] == "f64" and d["config"]["frame_len"] == 1200
        r = d["roofline"]
        assert r["kernel"] == "analyze" and r["bound"] == "fp64_valu" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5
        assert r["traffic"] and d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 16
        assert d["detail"] == "gpurun_out/bench_detail.json"
    d = json.loads(b.compact_line(_canned_record(b), None))
    sub = d["sub_benchmarks"]
    assert sub["config2"] == 1.15e9 and sub["pipeline_shapes"]["1200/480"] == 12345700.0 and "error" in sub["config5_100h_1gpu"]
    assert sub["speech_44k"]["1103/441"]["burg_direct"] == 0.4


def test_a_failing_sub_benchmark_becomes_a_record_not_a_lost_headline():
    b = _bench()

    def boom():
        raise RuntimeError("hipMalloc failed")
    r = b.guarded("config5_100h_1gpu", boom)
    assert r["name"] == "config5_100h_1gpu" and "hipMalloc" in r["error"]
    assert b.guarded("ok", lambda: {"name": "ok", "value": 1.0}) == {"name": "ok", "value": 1.0}


def test_live_counter_csv_parsing(tmp_path):
    """bench.py measures the headline kernel's traffic in the run from rocprofv3's counter CSVs: the one launch of the kernel is found by
    its name's prefix, other kernels' rows do not count, and zero or several launches are an error (never a silent zero)."""
    import pytest
    b = _bench()
    d = tmp_path / "FETCH_SIZE" / "runc"
    d.mkdir(parents=True)
    head = "Correlation_Id,Dispatch_Id,Agent_Id,Queue_Id,Process_Id,Thread_Id,Grid_Size,Kernel_Id,Kernel_Name,Workgroup_Size,LDS_Block_Size,Scratch_Size,VGPR_Count,Accum_VGPR_Count,SGPR_Count,Counter_Name,Counter_Value,Start_Timestamp,End_Timestamp\n"
    row = lambda disp, kern, cnt, val: f'1,{disp},0,1,1,1,64,1,"{kern}",64,0,0,168,0,104,{cnt},{val},0,1\n'
    k = "void vbx::analyze_kernel<true, true, true, 0, 3>(vbx::spectral_args_t)"
    (d / "1_counter_collection.csv").write_text(head + row(1, "vbx::synth_kernel(double*)", "FETCH_SIZE", 5.0) + row(2, k, "FETCH_SIZE", 364397.0)
                                                + row(3, "void vbx::analyze_kernel<false, false, true, 0, 3>(vbx::spectral_args_t)", "FETCH_SIZE", 7.0))
    got = b.counter_values(str(tmp_path / "FETCH_SIZE"), b.LIVE_KERNELS["analyze"], ["FETCH_SIZE"])
    assert got == {"FETCH_SIZE": 364397.0}
    assert abs(got["FETCH_SIZE"] * 1024 * 2 / 180000 - 4146.03) < 0.01                  # x1024 x2 per frame of the 0.5 h child run
    with pytest.raises(RuntimeError):
        b.counter_values(str(tmp_path / "FETCH_SIZE"), "void vbx::renamed_kernel<", ["FETCH_SIZE"])
    (d / "2_counter_collection.csv").write_text(head + row(9, k, "FETCH_SIZE", 1.0))
    with pytest.raises(RuntimeError):
        b.counter_values(str(tmp_path / "FETCH_SIZE"), b.LIVE_KERNELS["analyze"], ["FETCH_SIZE"])
