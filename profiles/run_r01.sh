#!/bin/bash
# Round-1 profile set (run on the GPU box from the repo root):  bash profiles/run_r01.sh
# Bench lines, kernel-trace stats of the default bench command and of a ContextEncoder pass, batch sweep, and the PMC
# passes (separate runs, --kernel-trace + --pmc only) for the conv kernels and the stem.  Everything lands under
# gpurun_out/r01/; profiles/summarize_r01.py condenses it into profiles/r01/.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r01
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err && \
python3 $R/bench.py --precision f16x2 --no-cpu-baseline --no-context > $OUT/bench_n1_f16x2.json 2> $OUT/bench_n1_f16x2.err && \
python3 $R/bench.py --agents 64 --cfg-w 2.0 --no-cpu-baseline --no-context > $OUT/bench_n1_configs2_cfg.json 2> $OUT/err2 && \
python3 $R/bench.py --agents 64 --cfg-w 2.0 --guide --no-cpu-baseline --no-context > $OUT/bench_n1_configs2_cfg_guide.json 2> $OUT/err3 && \
python3 $R/bench.py --closed-loop 20 --denoise-steps 50 --scenes 64 --agents 64 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_n1_closed_loop.json 2> $OUT/err4 && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats -o bench -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/kstats.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats_ctx -o ctx -- python3 $R/scripts/ctx_time.py 256 > $OUT/kstats_ctx.log 2>&1 && \
python3 $R/scripts/sweep_batch.py 8 64 128 256 512 768 1024 1536 2048 4096 8192 > $OUT/batch_sweep.txt 2>&1 && \
CLD_SWEEP_PRECISION=f16x2 python3 $R/scripts/sweep_batch.py > $OUT/batch_sweep_f16x2.txt 2>&1 && \
python3 $R/bench.py --scenes 128 --agents 64 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_n1_configs3_shard.json 2> $OUT/err5 && \
python3 $R/bench.py --closed-loop 20 --denoise-steps 50 --scenes 64 --agents 64 --steps 1 --warmup 1 --no-cpu-baseline --no-context > $OUT/bench_n1_closed_loop_nocontext.json 2> $OUT/err6 && \
(CLD_DIST_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 $R/bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-context 2> $OUT/err7 | grep "^{" > $OUT/bench_n2_gloo_rehearsal.json) || { echo "bench/stat chain failed"; exit 1; }
ARGS="$R/bench.py --steps 1 --warmup 0 --denoise-steps 10 --no-cpu-baseline --no-profile --no-context"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o p -- python3 $ARGS > $OUT/pmc_sq.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_write -o p -- python3 $ARGS > $OUT/pmc_write.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_ctx_fetch -o p -- python3 $R/scripts/ctx_time.py 256 > $OUT/pmc_ctx_fetch.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_ctx_write -o p -- python3 $R/scripts/ctx_time.py 256 > $OUT/pmc_ctx_write.log 2>&1
ls -R $OUT | head -60
