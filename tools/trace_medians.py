#!/usr/bin/env python3
"""Median / minimum duration of a kernel per group of consecutive launches in a rocprofv3 --kernel-trace CSV (A-B runs that launch
one variant after the other):   python3 tools/trace_medians.py <trace dir> [launches per group = 300] [kernel name fragment]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/p_kernel_trace.csv", recursive=True)[0]
per = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rows = [r for r in csv.DictReader(open(f)) if (sys.argv[3] if len(sys.argv) > 3 else "features_fwd") in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print(len(rows))
for g in range(len(rows)//per):
    d = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[g*per+per//6:(g+1)*per])
    print(g, "median %.2f us  min %.2f" % (d[len(d)//2]/1e3, d[0]/1e3))
