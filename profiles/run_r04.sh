#!/bin/bash
# Round-4 profile set (run on the GPU box from the repo root):  bash profiles/run_r04.sh
# (round 4 adds: the Winograd layer chains (chain_check with the chainw form, in-kernel stamps of the head chain), the MFMA / VALU co-issue
# microbenchmark, the split collision-guided step)
# Headline bench line (BASELINE configs[2]), kernel-trace stats of the same command, per-launch U-Net breakdowns, the PMC
# passes (separate runs, --kernel-trace + --pmc only) for the conv kernels at the headline's 4,096 rows per launch and for
# the stem, batch sweep, guidance-kernel timing, and a 2-rank gloo rehearsal of the strong-scaling entry.  Everything lands
# under gpurun_out/r04/; profiles/summarize_r04.py condenses it into profiles/r04/.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
HL="--steps 1 --warmup 1 --no-cpu-baseline --no-extras"
PART=${PART:-all}      # 1 = benches, traces, sweeps; 2 = PMC passes and stamps (a gpurun call is limited to 20 minutes)
if [ "$PART" != 2 ]; then
python3 $R/bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats -o bench -- python3 $R/bench.py $HL > $OUT/kstats.log 2>&1 && \
bash $R/scripts/trace_unet.sh r04 64 1024 2048 4096 && \
python3 $R/scripts/chain_check.py 64 1024 2048 4096 > $OUT/chain_check.txt 2>&1 && \
python3 $R/scripts/collision_time.py > $OUT/collision_time.txt 2>&1 && \
python3 $R/scripts/guide_time.py 1024 2048 4096 > $OUT/guide_time.txt 2>&1 && \
python3 $R/scripts/sweep_batch.py 8 64 128 256 512 768 1024 1536 2048 4096 8192 > $OUT/batch_sweep.txt 2>&1 && \
CLD_SWEEP_CONV5=direct python3 $R/scripts/sweep_batch.py 512 768 1024 2048 4096 > $OUT/batch_sweep_direct_form.txt 2>&1 && \
python3 $R/scripts/ctx_time.py 1024 winograd > $OUT/ctx_time_1024.txt 2>&1 && python3 $R/scripts/ctx_time.py 1024 winograd_f2 >> $OUT/ctx_time_1024.txt 2>&1 && python3 $R/scripts/ctx_time.py 1024 direct >> $OUT/ctx_time_1024.txt 2>&1 && \
python3 $R/bench.py --workload configs1 --no-cpu-baseline --no-extras > $OUT/bench_n1_configs1.json 2> $OUT/err1 && \
python3 $R/bench.py --workload configs3 --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_n1_configs3.json 2> $OUT/err3 && \
python3 $R/bench.py --workload configs4 --scenes 64 --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_n1_configs4_shard.json 2> $OUT/err4 && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats_ctx -o ctx -- python3 $R/scripts/ctx_time.py 256 > $OUT/kstats_ctx.log 2>&1 && \
(CLD_DIST_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 $R/bench.py --gpus 2 --scenes 64 --steps 1 --warmup 1 --no-cpu-baseline 2> $OUT/err7 | grep "^{" > $OUT/bench_n2_gloo_rehearsal.json) || { echo "bench/stat chain failed"; tail -5 $OUT/*.err $OUT/err* 2>/dev/null; exit 1; }
fi
[ "$PART" = 1 ] && { ls $OUT | head -80; exit 0; }
ARGS="$R/bench.py --steps 1 --warmup 0 --denoise-steps 10 --no-cpu-baseline --no-profile --no-extras"
CLD_LIB_PATH=$R/controllable-latent-diffusion-for-traffic-simulation_amd/libcld_stamps.so python3 $R/scripts/wino1d_stamps.py 4096 3 8 13 18 > $OUT/wino1d_stamps_4096.txt 2>&1
CLD_LIB_PATH=$R/controllable-latent-diffusion-for-traffic-simulation_amd/libcld_stamps.so python3 $R/scripts/chainw_stamps.py 4096 > $OUT/chainw_stamps_4096.txt 2>&1
$R/scripts/ubench/mfma_covalu > $OUT/mfma_covalu.txt 2>&1
$R/scripts/ubench/w44_unit 56 256 > $OUT/wino44_stamps.txt 2>&1; $R/scripts/ubench/w44_unit 28 256 >> $OUT/wino44_stamps.txt 2>&1
bash $R/scripts/wino_pmc.sh winograd > $OUT/ctx_clock_mfma_busy.txt 2>&1; bash $R/scripts/wino_pmc.sh direct >> $OUT/ctx_clock_mfma_busy.txt 2>&1
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o p -- python3 $ARGS > $OUT/pmc_sq.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_write -o p -- python3 $ARGS > $OUT/pmc_write.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_ctx_fetch -o p -- python3 $R/scripts/ctx_time.py 256 > $OUT/pmc_ctx_fetch.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_ctx_write -o p -- python3 $R/scripts/ctx_time.py 256 > $OUT/pmc_ctx_write.log 2>&1
# raw traces are large: keep the per-kernel stats and the counter tables only
find $OUT -name '*_kernel_trace.csv' -size +8M -delete
ls $OUT | head -60
