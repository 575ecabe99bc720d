#!/bin/bash
# Profiles of bench.py for one round: kernel trace + stats of the default command (every leg), then separate counter passes
# (HBM traffic, VALU instructions) per leg with every kernel on one stream.
# Usage (on the GPU box): tools/profile_round.sh r02
# Outputs under gpurun_out/profile_<tag>/ ; copy what should be kept into profiles/.
set -e -o pipefail
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/profile_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
run() { # name, bench args (quoted string), rocprof args...
    local name=$1; local bargs=$2; shift 2
    echo "[profile] $name: rocprofv3 $* -- bench.py $bargs"
    timeout -s KILL 280 rocprofv3 "$@" -d "$out/$name" -o p --output-format csv -- python3 "$root/bench.py" --no-cpu $bargs > "$out/$name.stdout" 2> "$out/$name.stderr"
}
run trace "--legs main,streams64,orb_match,reloc,single_stream" --kernel-trace --stats
cp "$out/trace/p_kernel_stats.csv" "$out/${tag}_bench_kernel_stats.csv"
# counter passes: one stream, a few steps, one configuration each
passes=(p256 p64 preloc)
declare -A PARGS=( [p256]="--overlap off --steps 4 --warmup 2 --legs main" \
                   [p64]="--overlap off --steps 4 --warmup 2 --batch 64 --legs main,orb_match" \
                   [preloc]="--overlap off --steps 4 --warmup 2 --batch 8 --legs reloc" )
for p in "${passes[@]}"; do
    run ${p}_fetch "${PARGS[$p]}" --pmc FETCH_SIZE TCC_HIT_sum
    run ${p}_write "${PARGS[$p]}" --pmc WRITE_SIZE TCC_MISS_sum
    run ${p}_valu "${PARGS[$p]}" --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES
    for c in fetch write valu; do
        python3 "$root/tools/pmc_summary.py" "$out/${p}_$c" k_ > "$out/${tag}_${p}_pmc_$c.txt"
    done
done
echo '{"pass": "p256", "batch": 256, "command": "bench.py --no-cpu --overlap off --steps 4 --warmup 2 --legs main"}' > "$out/main.meta.json"
echo '{"pass": "p64", "batch": 64, "command": "bench.py --no-cpu --overlap off --steps 4 --warmup 2 --batch 64 --legs main,orb_match"}' > "$out/streams64.meta.json"
echo '{"pass": "p64", "batch": 64, "command": "bench.py --no-cpu --overlap off --steps 4 --warmup 2 --batch 64 --legs main,orb_match"}' > "$out/orb_match.meta.json"
echo '{"pass": "preloc", "batch": 10000000, "command": "bench.py --no-cpu --overlap off --steps 4 --warmup 2 --batch 8 --legs reloc"}' > "$out/reloc.meta.json"
python3 "$root/tools/pmc_traffic.py" "$out" "$tag" > "$out/${tag}_traffic.json"
head -c 3000 "$out/${tag}_traffic.json"
