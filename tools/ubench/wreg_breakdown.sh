#!/bin/bash
# Where conv_wreg's iteration goes: the same launches with the stores (1) or the LDS-DMA loads (4) left out (RS_WREG_DBG bits;
# results are wrong by construction, only the time counts).
for d in 0 1 4 5; do echo "RS_WREG_DBG=$d"; RS_WREG_DBG=$d python tools/ubench/wreg_shapes.py 22 23 2>&1 | grep -v amdgpu.ids | head -3; done
