set -e
export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "klt or track or frontend or fuzz" > gpurun_out/wide_tests.log 2>&1 || { tail -30 gpurun_out/wide_tests.log; exit 1; }
tail -3 gpurun_out/wide_tests.log
for m in 0 8192; do
 echo "WIDE_MAX=$m"
 VFE_KLT_WIDE_MAX=$m timeout -k 10 300 python bench.py --legs main,single_stream --batch 8 --steps 200 --warmup 20 --no-cpu 2>/dev/null > gpurun_out/wide_b8_$m.json
 python - <<PY
import json
d=json.load(open("gpurun_out/wide_b8_$m.json"))
print("b8 ms_per_step", d["ms_per_step"], {k:v for k,v in d.get("kernels_ms",{}).items()} if "kernels_ms" in d else "")
print("single", d.get("single_stream"))
PY
done
