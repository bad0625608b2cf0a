#!/bin/bash
# issue / wait counters of the ContextEncoder's convolution kernels (scripts/ctx_time.py 256 <form>), per launch average
R=${GRAFT_REPO_ROOT:-/root/repo}
FORM=${1:-winograd}
OUT=$R/gpurun_out/pmc_ctx_$FORM
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $OUT/a -o p -- python3 $R/scripts/ctx_time.py 256 $FORM > $OUT/log_a.txt 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/b -o p -- python3 $R/scripts/ctx_time.py 256 $FORM > $OUT/log_b.txt 2>&1 || exit 1
python3 - $OUT <<'PY'
import csv, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("a", "b"):
    for r in csv.DictReader(open(f"{out}/{sub}/p_counter_collection.csv")):
        k = r["Kernel_Name"].split("(")[0]
        if "wino" in k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(agg.items()):
    print(k)
    for n, v in sorted(c.items()): print(f"   {n:28s} {sum(v)/len(v):16.0f}")
PY
