#!/bin/bash
# rocprofv3 evidence for the bench line (run on the GPU box through gpurun): kernel-trace statistics of the timed bench command and the
# two HBM counter passes (FETCH_SIZE / WRITE_SIZE collected separately, /opt/skills/guides/MI355X_MICROARCH.md §HBM), summarised per kernel
# by tools/pmc_summary.py.  Outputs under gpurun_out/prof/ ; copy what is to be judged into profiles/rNN/.
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0 --no-fp32-mode"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $ROOT/bench.py $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/bench_b16_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc_write.err
F=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1)
W=$(find $OUT/pmc_write -name "*counter_collection.csv" | head -1)
python3 $ROOT/tools/pmc_summary.py $F $W $OUT/pmc_hbm_per_kernel.json > /dev/null
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write       # raw traces are large; the summaries stay
ls -la $OUT
