#!/bin/bash
# the bench line on ONE box with the 1x1 layers on conv_igemm's 128 x 256 tile (RS_CONV_WREG=0), on conv_wreg with 64-pixel tiles
# (RS_WREG_WAVES=4) and on the form that ships (32-pixel tiles, two workgroups per CU); twice, interleaved
for r in 1 2; do
for cfg in "RS_CONV_WREG=0" "RS_WREG_WAVES=4" "RS_WREG_WAVES=2"; do
  echo -n "$cfg  "
  env $cfg python3 bench.py --no-cpu-baseline --no-reference-precision --no-trained-leg --no-train-leg --steps 40 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'tiles/s', round(d['ms_per_step'],3), 'ms', round(d['roofline']['frac'],4))"
done
done
