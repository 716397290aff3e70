#!/usr/bin/env python3
"""Time rs_op_conv2d_wgrad_f32 (conv_wgrad_f32_kernel + the split reduce) per training-relevant layer shape (batch 8, 800x800 network
input), standalone: TFLOP/s against the 157.3 TFLOP/s fp32 matrix peak.  (The op allocates its scratch per call: times include two
hipMalloc / hipFree pairs and a memset, so the kernel itself is a little faster than printed.)"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from proj_roadsurf_amd.engine import load_library, _check

lib = load_library()
lib.rs_op_conv2d_wgrad_f32.argtypes = lib.rs_op_conv2d_wgrad.argtypes
dev = torch.device("cuda:0")
N = 8
shapes = [  # name, cin, cout, k, stride, h, w
    ("fpn_output2 / rpn.conv p2", 256, 256, 3, 1, 200, 200),
    ("fpn_output3", 256, 256, 3, 1, 100, 100),
    ("res3.x.conv2", 128, 128, 3, 1, 100, 100),
    ("res4.x.conv2", 256, 256, 3, 1, 50, 50),
    ("res4.x.conv1", 1024, 256, 1, 1, 50, 50),
    ("res4.x.conv3", 256, 1024, 1, 1, 50, 50),
    ("fpn_lateral2", 256, 256, 1, 1, 200, 200),
    ("box.fc1 (8192 x 12544 -> 1024)", 12544, 1024, 1, 1, 8192, 1),
]
for name, cin, cout, k, stride, h, w in shapes:
    pad = k // 2
    halo = 1 if k == 3 else 0
    x = torch.randn(N if w > 1 else 1, h + 2 * halo, w + 2 * halo, cin, device=dev)
    dy = torch.randn(N if w > 1 else 1, h + 2 * halo, w + 2 * halo, cout, device=dev)
    n = N if w > 1 else 1
    kpad = k * k * cin
    g = torch.empty(cout, kpad, device=dev)
    def call():
        rc = lib.rs_op_conv2d_wgrad_f32(C.c_void_p(dy.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(g.data_ptr()), None,
                                        n, h, w, cin, halo, k, k, stride, pad, cout, kpad, halo, 0, None)
        _check(lib, rc, name)
    for _ in range(2):
        call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        call()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    fl = 2.0 * n * h * w * k * k * cin * cout
    print(f"{name:34s} GFLOP {fl / 1e9:8.1f}  {dt * 1e3:8.3f} ms  {fl / dt / 1e12:6.1f} TFLOP/s  ({fl / dt / 1e12 / 157.3:.2f} of peak)", flush=True)
