#!/bin/bash
# the bench line on ONE box: every layer on conv_igemm's tiles (RS_CONV_WREG=0 RS_DECONV_VARIANT=14), the 1x1 layers on conv_wreg but the mask
# head's deconv + predictor still on conv_igemm (RS_DECONV_VARIANT=14), and what ships; twice, interleaved
for r in 1 2; do
for cfg in "RS_CONV_WREG=0 RS_DECONV_VARIANT=14" "RS_DECONV_VARIANT=14" "RS_CONV_WREG=1"; do
  echo -n "$cfg  "
  env $cfg python3 bench.py --no-cpu-baseline --no-reference-precision --no-trained-leg --no-train-leg --steps 40 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'tiles/s', round(d['ms_per_step'],3), 'ms', round(d['roofline']['frac'],4))"
done
done
