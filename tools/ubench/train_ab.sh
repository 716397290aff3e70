#!/bin/bash
# training step at batch 8 with the RoIAlign backward by atomics / by owner-computes regions, same box (the early anchor labels are on in both)
for a in 1 0; do
  echo "RS_ROI_BWD_ATOMIC=$a"
  RS_ROI_BWD_ATOMIC=$a python3 tools/ubench/train_stages.py 8 2>&1 | grep -E "ms/step|bwd.*roi_align|rpn.match|rpn.subsample|mask.fcn1 |chain"
done
