#!/bin/bash
# Counter pass for tools/exp_klt_pmc.py; prints VALU instructions per wave for each tracker dispatch in launch order.
set -e -o pipefail
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/pmc_klt_exp
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d "$out" -o r --output-format csv -- python3 "$root/tools/exp_klt_pmc.py" > "$out/stdout.log" 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "k_klt21" not in r["Kernel_Name"]: continue
    rows.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
for k, v in rows.items():
    w = v["SQ_WAVES"]
    print(k, "waves %d  valu/wave %.0f  salu/wave %.0f  lds/wave %.0f  vmem_rd/wave %.0f" % (w, v["SQ_INSTS_VALU"] / w, v["SQ_INSTS_SALU"] / w, v["SQ_INSTS_LDS"] / w, v["SQ_INSTS_VMEM_RD"] / w))
PY
