#!/bin/bash
# rocprofv3 evidence for the bench line (run on the GPU box through gpurun): kernel-trace statistics of the timed bench command and the
# two HBM counter passes (FETCH_SIZE / WRITE_SIZE collected separately, /opt/skills/guides/MI355X_MICROARCH.md §HBM), summarised per kernel
# by tools/pmc_summary.py.  Outputs under gpurun_out/prof/ ; copy what is to be judged into profiles/rNN/.
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0 --no-reference-precision --no-trained-leg --no-train-leg"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $ROOT/bench.py $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/bench_b16_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc_write.err
F=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1)
W=$(find $OUT/pmc_write -name "*counter_collection.csv" | head -1)
python3 $ROOT/tools/pmc_summary.py $F $W $OUT/pmc_hbm_per_kernel.json > /dev/null
# the same two measurements with box.roi_align in score order (RS_ROI_ORDER=0): what the visiting order is worth
export RS_ROI_ORDER=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace0 -o bench -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/trace0.err
find $OUT/trace0 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/bench_b16_kernel_stats_roi_score_order.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch0 -o f -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc_fetch0.err
F0=$(find $OUT/pmc_fetch0 -name "*counter_collection.csv" | head -1)
python3 $ROOT/tools/pmc_summary.py $F0 $W $OUT/pmc_hbm_per_kernel_roi_score_order.json > /dev/null
unset RS_ROI_ORDER
# one training step (BASELINE configs[4], batch 8, fp16 trainer) under the kernel trace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_t -o train -- python3 $ROOT/bench.py --train --train-legs b8 --steps 10 --warmup 3 > $OUT/train_b8_under_rocprof.json 2> $OUT/trace_t.err
find $OUT/trace_t -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/train_b8_kernel_stats.csv
rm -rf $OUT/trace0 $OUT/pmc_fetch0 $OUT/trace_t
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write       # raw traces are large; the summaries stay
ls -la $OUT
