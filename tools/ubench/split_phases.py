#!/usr/bin/env python3
"""Where a wave of the split-operand conv_deep loop spends a K step (diagnostic build: make -C proj_roadsurf_amd/csrc OUT=../librs_engine_phases.so
BUILD=build_exp_phases EXTRA=-DRS_SPLIT_PHASES): shader-clock stamps at the three points of a step where no LDS read is in flight give, per wave and
step, the cycles from the barrier to the end of the step's LDS reads ("work": MFMA blocks, read bursts, LDS-DMA issue), the wait for the step's own
LDS-DMA pieces, and the wait at the barrier.  3x3 256 -> 256 on 16 x 200 x 200 (fpn_output2 / rpn.conv2)."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from proj_roadsurf_amd import engine as E

lib = E.load_library(os.environ.get("RS_LIB") or os.path.join(ROOT, "proj_roadsurf_amd", "librs_engine_phases.so"))
lib.rs_debug_set_conv_probe.argtypes = [C.c_void_p]
dev = torch.device("cuda:0")
N, H, W, Cc = 16, 200, 200, 256
x = (torch.randn(2, N, H + 2, W + 2, Cc, device=dev) * 0.5).half()
w = (torch.randn(2, Cc, 9 * Cc, device=dev) * 100).half()
sc = torch.full((Cc,), 1e-4, device=dev)
b = torch.zeros(Cc, device=dev)
o = torch.zeros(2, N, H + 2, W + 2, Cc, device=dev, dtype=torch.float16)
tiles = (N * H * W + 255) // 256
probe = torch.zeros(tiles + 256, 8, 8, dtype=torch.int64, device=dev)
lib.rs_debug_set_conv_probe(C.c_void_p(probe.data_ptr()))


def launch():
    rc = lib.rs_op_conv2d_split(C.c_void_p(x.data_ptr()), x[0].numel(), C.c_void_p(w.data_ptr()), w[0].numel(), C.c_void_p(sc.data_ptr()), C.c_void_p(b.data_ptr()),
                                C.c_void_p(o.data_ptr()), o[0].numel(), None, 0, None, 0, N, H, W, Cc, 1, 3, 3, 1, 1, Cc, 9 * Cc, 1, 1, 0, 0, 12, None)
    assert rc == 0, lib.rs_last_error()


for _ in range(20):
    launch()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    launch()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 50
p = probe.cpu().numpy().astype(np.float64)[:tiles]
ok = p[:, :, 3] > 0
steps = p[:, :, 3][ok].mean()
out = {"ms_per_launch": ms, "k_steps": steps}
for name, sel in (("all waves", slice(0, 8)), ("waves 0-3 (issue work first)", slice(0, 4)), ("waves 4-7 (last MFMA block first)", slice(4, 8))):
    q = p[:, sel, :]
    m = q[:, :, 3] > 0
    out[name] = {"work_cycles_per_step": float((q[:, :, 0][m] / q[:, :, 3][m]).mean()), "dma_wait_cycles_per_step": float((q[:, :, 1][m] / q[:, :, 3][m]).mean()),
                 "barrier_wait_cycles_per_step": float((q[:, :, 2][m] / q[:, :, 3][m]).mean())}
m = p[:, :, 3] > 0
out["per tile, cycles"] = {"prologue (entry -> first step)": float(p[:, :, 4][m].mean()), "K loop": float(p[:, :, 5][m].mean()),
                           "epilogue until the last store is issued": float(p[:, :, 6][m].mean()), "until the stores have drained": float(p[:, :, 7][m].mean())}
print(json.dumps(out, indent=1))
