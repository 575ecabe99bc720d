#!/usr/bin/env python3
"""Prints the GPU timeline (kernels and copies) of one steady-state vfe_process_frame call from a rocprofv3 trace made by
tools/prof_ctx.sh.  Usage: python3 tools/timeline_ctx.py gpurun_out/ctxprof/trace"""
import csv
import sys

d = sys.argv[1]
ev = []
for r in csv.DictReader(open(d + "/p_kernel_trace.csv")):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]))
try:
    for r in csv.DictReader(open(d + "/p_memory_copy_trace.csv")):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", "")))
except FileNotFoundError:
    pass
ev.sort()
klt = [i for i, e in enumerate(ev) if "k_klt21" in e[2]]
i = klt[len(klt) // 2]
# back up to the first upload band of this call (the frame arrives in a few consecutive host-to-device copies)
def is_up(e):
    return e[2].startswith("COPY") and "HOST_TO_DEVICE" in e[2] and e[1] - e[0] > 8000
j = i
while j > 0 and not is_up(ev[j]):
    j -= 1
while j > 0 and is_up(ev[j - 1]):
    j -= 1
t0 = ev[j][0]
for s, e, n in ev[j:j + 28]:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  {n}")
