#!/bin/bash
# Kernel trace of the headline leg alone, every kernel on one stream: the per-kernel averages are the ones bench.py's
# `kernels` table and `roofline.avg_launch_ms` report (its one-stream pass).  Usage (on the GPU box): tools/prof_main.sh r02
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/profile_main_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 280 rocprofv3 --kernel-trace --stats -d $out/trace -o p --output-format csv -- python3 $root/bench.py --no-cpu --legs main --overlap off --steps 40 --warmup 5 > $out/stdout 2> $out/stderr || { tail -5 $out/stderr; exit 1; }
cp $out/trace/p_kernel_stats.csv $out/${tag}_main_kernel_stats.csv
head -8 $out/${tag}_main_kernel_stats.csv | cut -c1-140
