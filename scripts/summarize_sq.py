#!/usr/bin/env python3
"""SQ instruction counters of the threshold+corner kernels (rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU ..., the dense_sq1 pass of
scripts/collect_profiles.sh) -> profiles/issue_<short>.json, read by bench.py for roofline.issue.
usage: summarize_sq.py SQ_DIR NFRAMES TAG KERNEL_SUBSTRING SHORT_NAME"""
import csv, glob, json, os, sys
d, nframes, tag, ksub, short = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
acc = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if ksub in r["Kernel_Name"]:
            acc.setdefault(r["Kernel_Name"].split("(")[0], {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
kname = sorted(acc)[0]
m = {c: sum(v) / len(v) for c, v in acc[kname].items()}
out = {"kernel": kname, "frames_in_profiled_launch": nframes, "source": "%s_sq_dense.txt" % tag,
       "valu_wave_insts_per_frame": m["SQ_INSTS_VALU"] / nframes, "salu_wave_insts_per_frame": m["SQ_INSTS_SALU"] / nframes,
       "lds_wave_insts_per_frame": m.get("SQ_INSTS_LDS", 0.0) / nframes,
       "vmem_wave_insts_per_frame": (m.get("SQ_INSTS_VMEM_RD", 0.0) + m.get("SQ_INSTS_VMEM_WR", 0.0)) / nframes,
       "waves_per_frame": m.get("SQ_WAVES", 0.0) / nframes}
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
json.dump(out, open(os.path.join(root, "profiles", "issue_%s.json" % short), "w"), indent=1)
print(json.dumps(out, indent=1))
