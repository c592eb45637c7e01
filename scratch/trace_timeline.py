#!/usr/bin/env python3
"""prints a compact timeline from a rocprofv3 kernel_trace.csv: the last N dispatches, start/end relative, queue id"""
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = list(csv.DictReader(open(f)))
rows = [r for r in rows if "synth" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-N:]
t0 = int(rows[0]["Start_Timestamp"])
qs = sorted(set(r["Queue_Id"] for r in rows))
for r in rows:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print("q%d  %9.1f -> %9.1f  (%7.1f us)  %s" % (qs.index(r["Queue_Id"]), s, e, e - s, r["Kernel_Name"].split("(")[0][:40]))
