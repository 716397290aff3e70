#!/bin/bash
# RoIAlign backward: float atomics (rounds 1-2) against owner-computes regions, same box, one training step profile each
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for a in 1 0; do
  echo "RS_ROI_BWD_ATOMIC=$a"
  RS_ROI_BWD_ATOMIC=$a python3 tools/ubench/train_stages.py 8 2>&1 | grep -E "ms/step|bwd.*roi_align|chain"
done
mkdir -p gpurun_out/prof_roibwd
RS_ROI_BWD_ATOMIC=0 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_roibwd -o t -- python3 tools/ubench/train_stages.py 8 > /dev/null 2>&1
f=$(ls gpurun_out/prof_roibwd/*kernel_stats.csv gpurun_out/prof_roibwd/*/*kernel_stats.csv 2>/dev/null | head -1)
echo "stats: $f"
grep -E "roi_bwd|roi_align_bwd|Name" "$f" | cut -c1-200
