#!/usr/bin/env python3
"""Launch rs_op_conv2d_wgrad once per training-relevant layer shape (batch 8, 800x800 network input) so that
`rocprofv3 --kernel-trace --stats` gives per-shape kernel durations.  Prints the FLOP count per shape."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from proj_roadsurf_amd.engine import load_library, _check

lib = load_library()
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
]
for name, cin, cout, k, stride, h, w in shapes:
    pad = k // 2
    x = torch.randn(N, h + 2, w + 2, cin, device=dev).half()
    dy = torch.randn(N, h + 2, w + 2, cout, device=dev).half()
    kpad = k * k * cin
    g = torch.empty(cout, kpad, device=dev)
    for _ in range(3):
        rc = lib.rs_op_conv2d_wgrad(C.c_void_p(dy.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(g.data_ptr()), None,
                                    N, h, w, cin, 1, k, k, stride, pad, cout, kpad, 1, 0, None)
        _check(lib, rc, name)
    torch.cuda.synchronize()
    print(f"{name:28s} GFLOP {2.0 * N * h * w * k * k * cin * cout / 1e9:8.1f}")
