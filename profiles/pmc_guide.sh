#!/bin/bash
# SQ counter passes for the guidance kernel (one formulation -- 3 = 8-agent 4x4x1, 2 = 16-agent 16x16x4 -- at 2,048 agents):
#   bash profiles/pmc_guide.sh <tag> <form> [lib]   -> gpurun_out/<tag>/p1..p4 and a per-launch summary on stdout
set -u
TAG=${1:-pg}; FORM=${2:-3}; R=$GRAFT_REPO_ROOT
[ -n "${3:-}" ] && export CLD_LIB_PATH=$R/controllable-latent-diffusion-for-traffic-simulation_amd/$3
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$R/scripts/guide_one.py 2048 $FORM 6"
P() { n=$1; shift; rm -rf $OUT/p$n; timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/p$n -o p -- python3 $ARGS > $OUT/p$n.log 2>&1; }
P 1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY && \
P 2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD && \
P 3 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU && \
P 4 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS
python3 - <<PY
import csv, glob, collections
for n in (1, 2, 3, 4):
    for f in glob.glob("$OUT/p%d/**/*counter_collection.csv" % n, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "guide" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(f"{k:32s} per launch {sum(v) / max(1, len(v)):16.0f}   ({len(v)} samples)")
PY
