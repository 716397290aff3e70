#!/bin/bash
# Second half of tools/final_measure.sh (the first is tools/profile_bench.sh): the default bench line with every leg, the training
# line, the one-lane stage table, BASELINE configs[3] and the end-to-end CLI on both synthetic workloads.
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd $ROOT
python3 bench.py > $OUT/bench_b16.json 2> $OUT/bench_b16.err
python3 bench.py --train > $OUT/bench_train.json 2> $OUT/bench_train.err
python3 bench.py --no-cpu-baseline --no-reference-precision --no-trained-leg --no-train-leg --lanes 1 --stages > $OUT/bench_b16_lane1.json 2> $OUT/bench_b16_stage_table.txt
python3 bench.py --tile 1024 --bands 4 --batch 8 --no-cpu-baseline --no-trained-leg --no-train-leg > $OUT/bench_cfg4_1024x1024x4_b8.json 2> $OUT/bench_cfg4.err
python3 tools/cli_bench.py --weights trained --tiles 8192 > $OUT/cli_bench_trained_8192_tiles.json 2> $OUT/cli_trained.err
python3 tools/cli_bench.py --weights random --tiles 1024 > $OUT/cli_bench_random_1024_tiles.json 2> $OUT/cli_random.err
grep -h "tiles/s\|stage busy\|forward thread" $OUT/cli_trained.err $OUT/cli_random.err > $OUT/cli_bench_log_lines.txt
cat $OUT/cli_bench_log_lines.txt
tail -3 $OUT/bench_b16_stage_table.txt
