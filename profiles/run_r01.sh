#!/bin/bash
# Round-1 profile set (run on the GPU box from the repo root):  bash profiles/run_r01.sh
# Kernel-trace stats of the default bench command, the bench lines themselves, the ContextEncoder pass, and the PMC passes
# (separate runs, --kernel-trace only) for the dominant kernel.  Everything lands under gpurun_out/r01/; the summaries
# are copied into profiles/r01/ by hand afterwards.
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
python3 $R/scripts/sweep_batch.py > $OUT/batch_sweep.txt 2>&1
ls -R $OUT | head -40
