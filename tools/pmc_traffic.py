#!/usr/bin/env python3
"""HBM bytes per launch of every library kernel from the FETCH_SIZE / WRITE_SIZE passes of
tools/profile_round.sh.  FETCH_SIZE and WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts
128-byte read requests as 64 bytes, so it is doubled (calibrated for wide streaming reads only).
Usage: python3 tools/pmc_traffic.py DIR TAG"""
import csv, glob, json, os, sys
from collections import defaultdict


def per_kernel(root, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"].split("(")[0].replace("void ", "")
            if "at::" in name or not name.startswith("k_"):
                continue
            acc[name.split("<")[0]].append(float(row["Counter_Value"]))
    return acc


def main():
    root, tag = sys.argv[1], sys.argv[2]
    fetch = per_kernel(os.path.join(root, "pmc_fetch"), "FETCH_SIZE")
    write = per_kernel(os.path.join(root, "pmc_write"), "WRITE_SIZE")
    out = {"tag": tag, "note": "bytes per launch; read = 2 x FETCH_SIZE KiB (gfx950 correction), write = WRITE_SIZE KiB",
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        rd = 2.0 * 1024.0 * (sum(fetch[k]) / len(fetch[k])) if fetch.get(k) else None
        wr = 1024.0 * (sum(write[k]) / len(write[k])) if write.get(k) else None
        out["kernels"][k] = {"read_bytes": rd, "write_bytes": wr, "launches": len(fetch.get(k, [])),
                             "total_bytes": (rd or 0.0) + (wr or 0.0)}
    # bench.py names kernels by the library's profiling slots
    for kernel, slot in (("k_klt21", "k_klt"), ("k_pyr_down_stream", "k_pyr_down")):
        if kernel in out["kernels"]:
            out["kernels"][slot] = out["kernels"][kernel]
    # what limits the two big kernels: VALU issue = SQ_INSTS_VALU x 4 cycles / (SIMDs x kernel duration x clock)
    try:
        valu = per_kernel(os.path.join(root, "pmc_valu"), "SQ_INSTS_VALU")
        dur = {}
        for row in csv.DictReader(open(os.path.join(root, "trace", "p_kernel_stats.csv"))):
            dur[row["Name"].split("(")[0].replace("void ", "").split("<")[0]] = float(row["AverageNs"])
        out["limiter"] = {}
        for kernel, slot in (("k_klt21", "k_klt"), ("k_gftt_eig", "k_gftt_eig")):
            if valu.get(kernel) and kernel in dur:
                frac = (sum(valu[kernel]) / len(valu[kernel])) * 4.0 / (1024 * dur[kernel] * 2.4)
                out["limiter"][slot] = ("VALU issue: SQ_INSTS_VALU x 4 cycles = %.0f%% of SIMD cycles (1024 SIMDs, 2.4 GHz; "
                                        "profiles/%s_bench_pmc_valu.txt)" % (100 * frac, tag))
    except Exception:
        pass
    try:
        cfg = json.loads(open(os.path.join(root, "pmc_fetch.stdout")).read().strip().splitlines()[-1])
        out["streams_per_gpu"] = cfg["config"]["streams_per_gpu"]
    except Exception:
        pass
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
