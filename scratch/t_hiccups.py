"""How often does a streamed step take much longer than the others on the DEVICE, and in which stage?  (One step in ~200 showed the
lattice + pose kernel at 1.4 ms instead of 0.17.)  Runs bench.py's step loop for many steps and lists the outliers.
usage: python scratch/t_hiccups.py [STEPS] [REPS]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
tot = out = 0
for r in range(reps):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", os.environ.get("WARMUP", "5"), "--no-cpu-baseline", "--no-extra-legs"] + os.environ.get("EXTRA", "").split(), capture_output=True, text=True)
    d = json.loads(p.stdout.strip().splitlines()[-1])
    t = d["steps_trace"]
    dev = t["device_ms_all"]; med = t["device_ms_median"]
    bad = [(i, round(v, 3)) for i, v in enumerate(dev) if v > 1.15 * med]
    tot += len(dev); out += len(bad)
    print("run %d: value %.0f, median device step %.3f ms, outliers (> 1.15 x median): %s; slowest: %s" % (r, d["value"], med, bad, t.get("slowest_step")), flush=True)
print("%d outliers in %d steps" % (out, tot))
