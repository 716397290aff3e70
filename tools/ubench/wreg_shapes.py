#!/usr/bin/env python3
"""conv_wreg.hip (variant 22) against conv_igemm's 128x256 and 64x256 tiles (14, 10) on the layers it is written for, WITH their
epilogue operands (res4.x.conv3 adds a residual, fpn_lateral2 the up-sampled p3): kernel time by HIP events, 200 back-to-back launches
after a warm-up, and the algorithmic HBM rate (activations + epilogue operand + output, weights once).
Usage: wreg_shapes.py [variant ...]   (default 14 10 22);  BATCH=16"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from proj_roadsurf_amd import engine as E

lib = E.load_library(os.environ.get("RS_LIB") or None)
dev = torch.device("cuda:0")
variants = [int(v) for v in sys.argv[1:]] or [14, 22, 23, 25]
B = int(os.environ.get("BATCH", "16"))
shapes = [("res4.x.conv3 +res", B, 50, 50, 1024, "res"), ("fpn_lateral2 +up", B, 200, 200, 256, "up"), ("fpn_lateral2 plain", B, 200, 200, 256, ""),
          ("res4.x.conv3 b8", 8, 50, 50, 1024, "res"), ("lateral2 b8 +up", 8, 200, 200, 256, "up"), ("lateral2 b3 +up", 3, 200, 200, 256, "up"),
          ("lateral2 b2 +up", 2, 200, 200, 256, "up"), ("lateral2 b1 +up", 1, 200, 200, 256, "up"), ("res4.x.conv3 b4", 4, 50, 50, 1024, "res"),
          ("res4.x.conv3 b3", 3, 50, 50, 1024, "res"), ("res4.x.conv3 b2", 2, 50, 50, 1024, "res"), ("res4.x.conv3 b1", 1, 50, 50, 1024, "res")]
for name, N, H, W, Cout, epi in shapes:
    x = torch.randn(N, H, W, 256, device=dev).half()
    w = (torch.randn(Cout, 256, device=dev) * 0.02).half()
    b = torch.zeros(Cout, device=dev)
    o = torch.zeros(N, H + 2, W + 2, Cout, device=dev, dtype=torch.float16)
    r = torch.randn(N, H + 2, W + 2, Cout, device=dev).half() if epi == "res" else None
    u = torch.randn(N, H // 2 + 2, W // 2 + 2, Cout, device=dev).half() if epi == "up" else None
    M = N * H * W
    byts = M * 256 * 2 + M * Cout * 2 + Cout * 256 * 2 + (M * Cout * 2 if r is not None else 0) + (M // 4 * Cout * 2 if u is not None else 0)
    line = f"{name:19s} M {M:6d} N {Cout:4d}:"
    for v in variants:
        def launch():
            return lib.rs_op_conv2d(C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(o.data_ptr()),
                                    C.c_void_p(r.data_ptr()) if r is not None else None, C.c_void_p(u.data_ptr()) if u is not None else None,
                                    N, H, W, 256, 0, 1, 1, 1, 0, Cout, 256, 1, 1, 0, 0, v, 1, None)
        if launch() != 0:
            line += f"  v{v}: n/a"
            continue
        t0 = time.time()
        while time.time() - t0 < 0.7:
            for _ in range(50):
                launch()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            launch()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 200
        line += f"  v{v}: {ms * 1e3:6.1f} us {byts / ms / 1e6:5.0f} GB/s"
    print(line, flush=True)
