#!/bin/bash
# Round-4 measurement set on ONE box (run through gpurun; tools/profile_bench.sh holds the rocprofv3 passes): the default bench line with every leg, the
# training line, BASELINE configs[3], the batch sweep, the one-lane stage tables of both fast modes and the end-to-end CLI in the fp16 and split modes.
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/final_r04
mkdir -p $OUT
cd $ROOT
python3 bench.py > $OUT/bench_b16.json 2> $OUT/bench_b16.err
echo "[final] bench line done"
python3 bench.py --train > $OUT/bench_train.json 2> $OUT/bench_train.err
python3 bench.py --tile 1024 --bands 4 --batch 8 --no-cpu-baseline --no-trained-leg --no-train-leg > $OUT/bench_cfg4_1024x1024x4_b8.json 2> $OUT/bench_cfg4.err
for b in 8 32; do python3 bench.py --batch $b --no-cpu-baseline --no-trained-leg --no-train-leg --no-reference-precision > $OUT/bench_batch$b.json 2> $OUT/bench_batch$b.err; done
echo "[final] bench variants done"
python3 tools/ubench/ref_stages.py 16 split > $OUT/split_stage_table.txt 2>&1
python3 bench.py --precision fp16 --no-cpu-baseline --no-reference-precision --no-trained-leg --no-train-leg --no-single-tile-leg --lanes 1 --stages > $OUT/bench_fp16_lane1.json 2> $OUT/fp16_stage_table.txt
python3 tools/cli_bench.py --weights trained --tiles 8192 --precision split > $OUT/cli_bench_split_trained_8192.json 2> $OUT/cli_split.err
python3 tools/cli_bench.py --weights trained --tiles 8192 --precision fp16 > $OUT/cli_bench_fp16_trained_8192.json 2> $OUT/cli_fp16.err
grep -h "tiles/s\|stage busy\|forward thread" $OUT/cli_split.err > $OUT/cli_split_log_lines.txt
grep -h "tiles/s\|stage busy\|forward thread" $OUT/cli_fp16.err > $OUT/cli_fp16_log_lines.txt
cat $OUT/cli_split_log_lines.txt $OUT/cli_fp16_log_lines.txt
