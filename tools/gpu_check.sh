#!/bin/bash
# Run on the GPU box: parity tests, then a short bench.  Usage: tools/gpu_check.sh <tag> [bench args]
tag=$1; shift
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/test_$tag.log 2>&1
rc=$?
tail -4 gpurun_out/test_$tag.log
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" gpurun_out/test_$tag.log | head -20; exit $rc; fi
timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu "$@" > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || { tail -5 gpurun_out/bench_$tag.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/bench_$tag.json"))
print("frames/s %.0f  ms/step %.3f" % (d["value"], d["ms_per_step"]))
for k,v in d["kernels"].items(): print("  %-14s n=%3d avg %.4f ms share %.3f algo %.0f GB/s" % (k, v["launches"], v["avg_ms"], v["share"], v["algo_GBs"] or 0))
print(d["tracks"])
PY
