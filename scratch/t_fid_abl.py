import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for v in ("", "NOLOAD", "NOTREE", "NOREFINE"):
    env = dict(os.environ)
    if v: env["RCC_LIBRARY"] = os.path.join(ROOT, "robot_camera_calibration_amd", "librcc_hip_abl_%s.so" % v)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--fiducials", "6x4", "--no-cpu-baseline", "--no-extra-legs", "--steps", "5", "--warmup", "2"], env=env, capture_output=True, text=True)
    import json
    try:
        j = json.loads(r.stdout.strip().splitlines()[-1]); print(v or "product", j["ms_per_step"], j["stage_ms_single_pass"], j["targets_found_in_last_step"])
    except Exception as e:
        print(v, "failed", r.stderr[-300:])
