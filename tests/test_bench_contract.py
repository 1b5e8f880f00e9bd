"""bench.py's contract, as far as it can be checked without a GPU: the evidence file roofline.traffic is read from holds the
kernels of the default workloads (a kernel renamed in the sources silently drops out of it otherwise), the launcher parses its
options, and the workload names are the documented ones."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_traffic_file_has_the_kernels_the_bench_reports():
    tab = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    for key, lo, hi in (("analyze", 4000, 12000), ("pitch", 3800, 12000), ("burg_512", 4000, 6000),
                        ("formant_resonances_512", 500, 2000)):
        assert key in tab, key
        e = tab[key]
        assert lo <= e["bytes_per_frame"] <= hi, (key, e["bytes_per_frame"])
        assert e["commit"] and "pmc_counters.json" in e["source"]
        assert os.path.exists(os.path.join(ROOT, e["source"].split(" ")[0])), e["source"]


def test_bench_help_lists_the_workloads_and_shape_options():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, check=True).stdout
    for word in ("--gpus", "--steps", "--warmup", "--frame-len", "--hop", "pipeline", "config2", "config3", "config4", "frontend"):
        assert word in out, word
