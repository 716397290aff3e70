#!/bin/bash
# Everything DESIGN.md §5 quotes, in one GPU call: rocprofv3 kernel statistics + HBM counter passes of the bench command
# (tools/profile_bench.sh), the default bench line (with cpu_baseline and the fp32 mode), the one-lane stage table, BASELINE configs[3]
# and the end-to-end CLI on both synthetic workloads.  Outputs under gpurun_out/prof/ ; copy what is to be judged into profiles/rNN/.
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
bash $ROOT/tools/profile_bench.sh > $OUT/profile_bench.log 2>&1
cd $ROOT
cp $OUT/pmc_hbm_per_kernel.json $ROOT/profiles/pmc_latest.json          # bench.py reads the traffic of the dominant kernel from here
python3 bench.py > $OUT/bench_b16.json 2> $OUT/bench_b16.err
python3 bench.py --train > $OUT/bench_train.json 2> $OUT/bench_train.err
python3 bench.py --no-cpu-baseline --no-reference-precision --no-trained-leg --no-train-leg --lanes 1 --stages > $OUT/bench_b16_lane1.json 2> $OUT/bench_b16_stage_table.txt
python3 bench.py --tile 1024 --bands 4 --batch 8 --no-cpu-baseline --no-trained-leg --no-train-leg > $OUT/bench_cfg4_1024x1024x4_b8.json 2> $OUT/bench_cfg4.err
python3 tools/cli_bench.py --weights trained --tiles 8192 > $OUT/cli_bench_trained_8192_tiles.json 2> $OUT/cli_trained.err
python3 tools/cli_bench.py --weights random --tiles 1024 > $OUT/cli_bench_random_1024_tiles.json 2> $OUT/cli_random.err
grep -h "tiles/s\|stage busy" $OUT/cli_trained.err $OUT/cli_random.err > $OUT/cli_bench_log_lines.txt
python3 - <<PY
import json
for f in ("bench_b16", "bench_cfg4_1024x1024x4_b8"):
    d = json.loads(open("$OUT/%s.json" % f).read().strip().splitlines()[-1])
    print(f, round(d["value"], 1), round(d["sustained_tiles_per_s"] or 0, 1), round(d["pcie_inclusive_tiles_per_s"] or 0, 1), round(d["roofline"]["frac"], 3),
          (d.get("reference_precision") or {}).get("tiles_per_s"), (d.get("parity") or {}).get("matched_fw"), (d.get("cpu_baseline") or {}).get("value"))
PY
cat $OUT/cli_bench_log_lines.txt
tail -3 $OUT/bench_b16_stage_table.txt
