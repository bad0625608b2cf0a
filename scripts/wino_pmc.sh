#!/bin/bash
# clock and MFMA-busy share of the ContextEncoder's convolution kernels: GRBM_GUI_ACTIVE (sum over 8 XCDs) and SQ_VALU_MFMA_BUSY_CYCLES
# (sum over 1,024 SIMDs) per launch of scripts/ctx_time.py 256 <form>.   usage (GPU box): bash scripts/wino_pmc.sh winograd [libname]
R=${GRAFT_REPO_ROOT:-/root/repo}
FORM=${1:-winograd}
[ -n "$2" ] && export CLD_LIB_PATH=$R/controllable-latent-diffusion-for-traffic-simulation_amd/libcld_$2.so
OUT=$R/gpurun_out/wino_pmc_$FORM${2:+_$2}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT -o p -- python3 $R/scripts/ctx_time.py 256 $FORM > $OUT/log.txt 2>&1 || exit 1
python3 - $OUT <<'PY'
import csv, sys, collections
out = sys.argv[1]
rows = list(csv.DictReader(open(out + "/p_counter_collection.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in rows:
    k = r["Kernel_Name"].split("(")[0]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if "Start_Timestamp" in r and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k, c in sorted(agg.items()):
    if "conv" not in k: continue
    g = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"]) / 8
    m = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
    d = sum(dur[k]) / len(dur[k]) if dur[k] else 0
    print(f"{k:60s} launches {len(c['GRBM_GUI_ACTIVE']):3d}  {d/1e3:8.1f} us  clock {g/d if d else 0:5.2f} GHz  MFMA busy {m/1024/g*100:5.1f} % of the SIMD cycles")
PY
