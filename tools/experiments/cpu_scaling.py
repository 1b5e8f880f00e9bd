"""How the CPU oracle's pipeline rate scales with threads on this box (is there a cgroup quota behind the affinity mask?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
import importlib
o = g.load_oracle(); g.load_package()
synth = importlib.import_module(g.PKG_NAME + ".synth")
audio = synth.synth_speech(10 * 48000 + 1200, sample_offset=0)
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(p, open(p).read().strip())
    except OSError as e: print(p, "absent")
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
for nt in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    n, t = o.cpu_bench("pipeline", audio, 1200, 480, 12, 48000.0, nt, 3.0)
    print(nt, "threads:", round(n / t, 1), "frames/s", round(n / t / nt, 2), "per thread")
