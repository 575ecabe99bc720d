#!/usr/bin/env python3
"""Print the headline figures of a bench.py JSON line (kernel tables omitted). Usage: show_bench.py FILE"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.0f %s  n_gpus %d  ms/step %.3f  frac_hbm %.3f" % (d["value"], d["unit"], d["n_gpus"], d["ms_per_step"], d["frontend_frac_of_hbm_roofline"]))
print("roofline", {k: v for k, v in d["roofline"].items() if k != "note"})
for k, v in d["kernels"].items():
    print("  %-18s n=%-5d avg %.4f ms share %.3f algoGB/s %s" % (k, v["launches"], v["avg_ms"], v["share"], v["algo_GBs"] and round(v["algo_GBs"])))
for leg in ("streams64", "orb_match", "reloc"):
    v = d.get(leg)
    if not v:
        continue
    print(leg, {a: b for a, b in v.items() if a not in ("kernels", "workload", "roofline")})
    print("   roofline", v.get("roofline"))
    for k, kv in v["kernels"].items():
        print("  %-18s n=%-5d avg %.4f ms share %.3f algoGB/s %s" % (k, kv["launches"], kv["avg_ms"], kv["share"], kv["algo_GBs"] and round(kv["algo_GBs"])))
for leg in ("single_stream", "cpu_baseline", "speedup_vs_cpu_all_cores", "speedup_vs_cpu_1core"):
    if leg in d:
        print(leg, d[leg])
