#!/bin/bash
# Kernel-by-kernel times of the single-stream host-pointer API (rocprofv3 kernel trace).  Usage (on the GPU box): tools/prof_ctx.sh
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/ctxprof; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 280 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $out/trace -o p --output-format csv -- python3 $root/tools/bench_ctx_latency.py --frames 40 "$@" > $out/stdout 2> $out/stderr || { tail -5 $out/stderr; exit 1; }
cat $out/stdout
python3 - <<PY
import csv
for r in csv.DictReader(open("$out/trace/p_kernel_stats.csv")):
    print(f'{r["Name"][:60]:60s} {int(r["Calls"]):5d} {int(r["TotalDurationNs"])/1e3:10.1f} us  avg {float(r["AverageNs"])/1e3:8.1f} us')
PY
