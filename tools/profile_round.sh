#!/bin/bash
# Profiles of the default bench.py command for one round: kernel trace + stats, then separate counter
# passes (HBM traffic, VALU utilisation).  Usage (on the GPU box): tools/profile_round.sh r01 [bench args]
# Outputs under gpurun_out/profile_<tag>/ ; copy what should be kept into profiles/.
set -e -o pipefail
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/profile_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
run() { # name, rocprof args...
    local name=$1; shift
    timeout -s KILL 280 rocprofv3 "$@" -d "$out/$name" -o p --output-format csv -- python3 "$root/bench.py" --no-cpu "${BENCH_ARGS[@]}" > "$out/$name.stdout" 2> "$out/$name.stderr"
}
BENCH_ARGS=("$@")
run trace --kernel-trace --stats
cp "$out/trace/p_kernel_stats.csv" "$out/${tag}_bench_kernel_stats.csv"
BENCH_ARGS=(--steps 4 --warmup 2 "$@")
run pmc_fetch --pmc FETCH_SIZE TCC_HIT_sum
run pmc_write --pmc WRITE_SIZE TCC_MISS_sum
run pmc_valu --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES
run pmc_mem --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
for p in pmc_fetch pmc_write pmc_valu pmc_mem; do
    python3 "$root/tools/pmc_summary.py" "$out/$p" k_ > "$out/${tag}_bench_$p.txt"
done
python3 "$root/tools/pmc_traffic.py" "$out" "$tag" > "$out/${tag}_traffic.json"
cat "$out/${tag}_traffic.json"
