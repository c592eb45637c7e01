#!/usr/bin/env python3
"""Turns the two rocprofv3 --pmc passes of scripts/prof_dense.py into profiles/<tag>_pmc_dense.json and
profiles/traffic_dense.json (read by bench.py).  FETCH_SIZE is calibrated on k_calib_copy_dword (known
byte count, same 4 B/lane access width) as MI355X_MICROARCH.md's HBM section prescribes; WRITE_SIZE
likewise.  Counter unit: KiB.  usage: summarize_pmc.py FETCH.csv WRITE.csv NFRAMES TAG"""
import csv, json, sys, os
fetch_csv, write_csv, nframes, tag = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
px = 1920 * 1080
def mean_by_kernel(path):
    acc = {}
    for r in csv.DictReader(open(path)):
        acc.setdefault(r["Kernel_Name"].split("(")[0], []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}
F, W = mean_by_kernel(fetch_csv), mean_by_kernel(write_csv)
copy_bytes = nframes * px
cf = copy_bytes / (F["k_calib_copy_dword"] * 1024.0)      # correction factors from the known copy
cw = copy_bytes / (W["k_calib_copy_dword"] * 1024.0)
rd = F["k_dense_march"] * 1024.0 * cf
wr = W["k_dense_march"] * 1024.0 * cw
alg = 2.0 * px * nframes
out = {"kernel": "k_dense_march", "frames_in_profiled_launch": nframes,
       "FETCH_SIZE_KiB_raw": F["k_dense_march"], "WRITE_SIZE_KiB_raw": W["k_dense_march"],
       "calibration": {"kernel": "k_calib_copy_dword", "bytes_each_way": copy_bytes, "FETCH_SIZE_KiB_raw": F["k_calib_copy_dword"],
                       "WRITE_SIZE_KiB_raw": W["k_calib_copy_dword"], "fetch_correction": cf, "write_correction": cw},
       "hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes_per_launch": rd + wr, "algorithmic_bytes_per_launch": alg,
       "traffic_over_algorithmic": (rd + wr) / alg, "hbm_bytes_per_frame": (rd + wr) / nframes}
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
json.dump(out, open(os.path.join(root, "profiles", tag + "_pmc_dense.json"), "w"), indent=1)
json.dump({"hbm_bytes_per_frame": out["hbm_bytes_per_frame"], "source": tag + "_pmc_dense.json",
           "frames_in_profiled_launch": nframes}, open(os.path.join(root, "profiles", "traffic_dense.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
