set -e
export TMPDIR=/tmp
VFE_FUZZ_SCALE=10 timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "klt or track or frontend or engine or golden or cpp" > gpurun_out/wide_tests.log 2>&1 || { tail -30 gpurun_out/wide_tests.log; exit 1; }
tail -3 gpurun_out/wide_tests.log
timeout -k 10 300 python bench.py --legs main,single_stream --steps 200 --warmup 20 --no-cpu 2>/dev/null > gpurun_out/wide_main.json
python - <<PY
import json
d=json.load(open("gpurun_out/wide_main.json"))
print("main ms_per_step", d["ms_per_step"], d["value"])
print("single", {k:(v["median_ms"], v["p95_ms"]) for k,v in d.get("single_stream").items()})
PY
