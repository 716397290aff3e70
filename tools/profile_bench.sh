#!/bin/bash
# rocprofv3 evidence for the bench line (run on the GPU box through gpurun): kernel-trace statistics of the timed bench command, the two HBM
# counter passes (FETCH_SIZE / WRITE_SIZE collected separately, /opt/skills/guides/MI355X_MICROARCH.md §HBM) summarised per kernel by
# tools/pmc_summary.py, and two SQ / GRBM passes (matrix-pipe busy cycles, clock under load, wave-state split) summarised by
# tools/pmc_mfma_summary.py.  PREC=split (the headline mode, default) or fp16.  Outputs under gpurun_out/prof_$PREC/ ; copy what is to be
# judged into profiles/rNN/.  The program itself follows `--` (no shell hop: the profiler's library initialises the GPU first).
set -e
PREC=${PREC:-split}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$PREC
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--precision $PREC --steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0 --no-reference-precision --no-trained-leg --no-train-leg --no-fp16-leg --no-single-tile-leg"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $ROOT/bench.py $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/bench_b16_kernel_stats.csv
echo "[profile] kernel trace done"
# The default command runs two lanes on independent streams: launches of the two lanes overlap and the durations above are times on a shared chip
# (the line's roofline.in_timed_region).  The line's roofline itself is measured on ONE lane; the same command with --lanes 1 gives its per-kernel
# durations without overlap.
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1 -o bench -- python3 $ROOT/bench.py $ARGS --lanes 1 > $OUT/bench_under_rocprof_lanes1.json 2> $OUT/trace1.err
find $OUT/trace1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/bench_b16_kernel_stats_lanes1.csv
echo "[profile] one-lane kernel trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc_write.err
F=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1)
W=$(find $OUT/pmc_write -name "*counter_collection.csv" | head -1)
python3 $ROOT/tools/pmc_summary.py $F $W $OUT/pmc_hbm_per_kernel.json > /dev/null
echo "[profile] HBM counters done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_mfma -o m -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc_mfma.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_wave -o v -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc_wave.err
M=$(find $OUT/pmc_mfma -name "*counter_collection.csv" | head -1)
V=$(find $OUT/pmc_wave -name "*counter_collection.csv" | head -1)
python3 $ROOT/tools/pmc_mfma_summary.py $OUT/pmc_mfma_per_kernel.json $M $V > /dev/null
echo "[profile] SQ counters done"
head -3 $M > $OUT/pmc_mfma_csv_head.txt
rm -rf $OUT/trace $OUT/trace1 $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma $OUT/pmc_wave       # raw traces are large; the summaries stay
date -u +%Y-%m-%dT%H:%MZ > $OUT/measured_at.txt
ls -la $OUT
