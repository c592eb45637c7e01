#!/usr/bin/env python3
"""Turns the two rocprofv3 --pmc passes of scripts/prof_dense.py (or prof_ingest.py) into
profiles/<tag>_pmc_<kernel>.json and, for the dense pass, profiles/traffic_dense.json (read by bench.py).
FETCH_SIZE / WRITE_SIZE are calibrated on a streaming copy with a known byte count and the same access width
(k_calib_copy_x4 for the 16 B/lane band and ingest kernels, k_calib_copy_dword for the strip kernel), as
MI355X_MICROARCH.md's HBM section prescribes.  Counter unit: KiB.
usage: summarize_pmc.py FETCH.csv WRITE.csv NFRAMES TAG KERNEL_SUBSTRING ALG_BYTES_PER_FRAME [CALIB_KERNEL [SHORT_NAME]]
SHORT_NAME names the outputs (profiles/<tag>_pmc_<short>.json, profiles/traffic_<short>.json); default dense / ingest."""
import csv, json, sys, os
fetch_csv, write_csv, nframes, tag, ksub, alg_pf = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5], float(sys.argv[6])
calib = sys.argv[7] if len(sys.argv) > 7 else "k_calib_copy_x4"
px = 1920 * 1080
def mean_by_kernel(path):
    acc = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"): continue
        acc.setdefault(r["Kernel_Name"].split("(")[0], []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}
F, W = mean_by_kernel(fetch_csv), mean_by_kernel(write_csv)
kname = [k for k in F if ksub in k][0]
out = {"kernel": kname, "frames_in_profiled_launch": nframes, "FETCH_SIZE_KiB_raw": F[kname], "WRITE_SIZE_KiB_raw": W[kname]}
cf = cw = None
if calib in F and calib in W:
    copy_bytes = float(os.environ.get("CALIB_BYTES", nframes * px))
    cf = copy_bytes / (F[calib] * 1024.0); cw = copy_bytes / (W[calib] * 1024.0)
    out["calibration"] = {"kernel": calib, "bytes_each_way": copy_bytes, "FETCH_SIZE_KiB_raw": F[calib], "WRITE_SIZE_KiB_raw": W[calib],
                          "fetch_correction": cf, "write_correction": cw}
else:
    cf, cw = 2.0, 1.0      # the guide's gfx950 figures
    out["calibration"] = {"kernel": None, "fetch_correction": cf, "write_correction": cw, "note": "MI355X_MICROARCH.md: FETCH_SIZE reports 1/2 on gfx950"}
rd = F[kname] * 1024.0 * cf; wr = W[kname] * 1024.0 * cw
alg = alg_pf * nframes
out.update({"hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes_per_launch": rd + wr, "algorithmic_bytes_per_launch": alg,
            "traffic_over_algorithmic": (rd + wr) / alg, "hbm_bytes_per_frame": (rd + wr) / nframes})
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
short = sys.argv[8] if len(sys.argv) > 8 else ("dense" if "dense" in kname else ("ingest" if "ingest" in kname else kname))
json.dump(out, open(os.path.join(root, "profiles", "%s_pmc_%s.json" % (tag, short)), "w"), indent=1)
json.dump({"hbm_bytes_per_frame": out["hbm_bytes_per_frame"], "source": "%s_pmc_%s.json" % (tag, short), "kernel": kname,
           "frames_in_profiled_launch": nframes}, open(os.path.join(root, "profiles", "traffic_%s.json" % short), "w"), indent=1)
print(json.dumps(out, indent=1))
