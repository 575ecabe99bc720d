#!/bin/bash
# One rocprofv3 counter pass over a microbench selection.
# Usage: tools/pmc_pass.sh TAG "COUNTER COUNTER ..." microbench-args...
set -e -o pipefail
tag=$1; ctrs=$2; shift 2
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/pmc_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc $ctrs -d "$out" -o r --output-format csv -- python3 "$root/tools/microbench.py" "$@" > "$out/stdout.log" 2>&1
python3 "$root/tools/pmc_summary.py" "$out" > "$out/summary.txt"
cat "$out/summary.txt"
