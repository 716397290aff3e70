#!/usr/bin/env python3
"""In-kernel shader clock of the dominant kernel (conv_deep_kernel) under sustained load, by the method of
MI355X_MICROARCH.md "DVFS give-back" item 6: stamps of s_memtime (shader clocks) and s_memrealtime (100 MHz) around the K
loop of every workgroup, after >= 2 s of back-to-back launches on random data; clock = d(memtime)/d(memrealtime) * 100 MHz.

Needs the diagnostic library: make -C proj_roadsurf_amd/csrc OUT=../librs_engine_probe.so BUILD=build_probe EXTRA=-DRS_CLOCK_PROBE
(the production library contains no stamp code).  Prints the median clock and the MFMA peak it implies."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from proj_roadsurf_amd import engine as E

lib = E.load_library(os.path.join(ROOT, "proj_roadsurf_amd", "librs_engine_probe.so"))
lib.rs_debug_set_conv_probe.argtypes = [C.c_void_p]
dev = torch.device("cuda:0")
N, H, W, Cc = 16, 200, 200, 256                      # fpn_output2 / rpn.conv2 of the batch-16 forward
x = torch.randn(N, H + 2, W + 2, Cc, device=dev).half()
w = (torch.randn(Cc, 9 * Cc, device=dev) * 0.02).half()
b = torch.zeros(Cc, device=dev)
o = torch.zeros(N, H + 2, W + 2, Cc, device=dev, dtype=torch.float16)
tiles = (N * H * W + 255) // 256
probe = torch.zeros(tiles, 2, dtype=torch.int64, device=dev)
lib.rs_debug_set_conv_probe(C.c_void_p(probe.data_ptr()))


def launch():
    rc = lib.rs_op_conv2d(C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(o.data_ptr()), None, None,
                          N, H, W, Cc, 1, 3, 3, 1, 1, Cc, 9 * Cc, 1, 1, 0, 0, 12, 1, None)
    assert rc == 0, lib.rs_last_error()


torch.cuda.synchronize()
t0 = time.time()
n = 0
while time.time() - t0 < 3.0:                        # sustained load first
    for _ in range(50):
        launch()
    torch.cuda.synchronize()
    n += 50
dt = (time.time() - t0) / n
p = probe.cpu().numpy().astype(np.float64)
ok = p[:, 1] > 0
clk = np.median(p[ok, 0] / p[ok, 1]) * 100e6
flop = 2.0 * N * H * W * 9 * Cc * Cc
peak = 256 * 4 * 1024 * clk                           # 256 CUs x 4 SIMDs x 1024 FLOP/clk (16x16x32 f16 MFMA every 16 clocks)
print(json.dumps({"kernel": "conv_deep_kernel (3x3 256->256, 16x200x200)", "launches": n, "ms_per_launch_incl_launch_gap": dt * 1e3,
                  "tflops": flop / dt / 1e12, "in_kernel_clock_ghz": clk / 1e9, "mfma_peak_at_that_clock_tflops": peak / 1e12,
                  "fraction_of_clock_adjusted_peak": flop / dt / peak, "workgroups_sampled": int(ok.sum())}))
