#!/usr/bin/env python3
"""Tile-variant sweep of the split-operand convolution (rs_op_conv2d_split) over every conv / GEMM shape of the batch-B forward: per shape the time of
each applicable tile and the one the dispatch rule picks (variant -1).  Kernel time by HIP events on the null stream, 60 back-to-back launches after
a warm-up.  Usage: split_shapes.py [batch]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from proj_roadsurf_amd import engine as E

lib = E.load_library(os.environ.get("RS_LIB") or None)
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
NAMES = {-1: "rule", 0: "128x128", 1: "256x64", 3: "256x128", 4: "256x256", 7: "64x128", 8: "128x64", 10: "64x256", 14: "128x256", 12: "deep256", 15: "deep160", 16: "deep192",
         17: "deep224"}
shapes = []


def add(name, h, cin, cout, k=1, stride=1, res=False, up=False, n=B, w=None):
    shapes.append((name, n, h, w or h, cin, cout, k, stride, res, up))


add("res2.0.conv1 64->64", 200, 64, 64); add("res2.x.conv1 256->64", 200, 256, 64); add("res2.conv2 3x3 64", 200, 64, 64, 3)
add("res2.conv3 64->256 +res", 200, 64, 256, res=True)
add("res3.0.conv1 256->128 s2", 200, 256, 128, stride=2); add("res3.x.conv1 512->128", 100, 512, 128); add("res3.conv2 3x3 128", 100, 128, 128, 3)
add("res3.conv3 128->512 +res", 100, 128, 512, res=True)
add("res4.0.conv1 512->256 s2", 100, 512, 256, stride=2); add("res4.x.conv1 1024->256", 50, 1024, 256); add("res4.conv2 3x3 256", 50, 256, 256, 3)
add("res4.conv3 256->1024 +res", 50, 256, 1024, res=True)
add("res5.0.conv1 1024->512 s2", 50, 1024, 512, stride=2); add("res5.x.conv1 2048->512", 25, 2048, 512); add("res5.conv2 3x3 512", 25, 512, 512, 3)
add("res5.conv3 512->2048 +res", 25, 512, 2048, res=True)
add("fpn_lateral5 2048->256", 25, 2048, 256); add("fpn_lateral4 1024->256 +up", 50, 1024, 256, up=True); add("fpn_lateral3 512->256 +up", 100, 512, 256, up=True)
add("fpn_lateral2 256->256 +up", 200, 256, 256, up=True)
add("3x3 256 @200 (fpn_out2/rpn2)", 200, 256, 256, 3); add("3x3 256 @100", 100, 256, 256, 3)
add("mask fcn 3x3 256 (100 rois/tile)", 14, 256, 256, 3, n=100 * B)
add("fc1 12544->1024", 1000 * B, 12544, 1024, w=1, n=1); add("fc2 1024->1024", 1000 * B, 1024, 1024, w=1, n=1)

for (name, n, h, w, cin, cout, k, stride, res, up) in shapes:
    pad = k // 2
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    ih = 1 if w > 1 else 0                                      # FC rows: no halo
    x = (torch.randn((2, n, h + 2 * ih, w + 2 * ih, cin), device=dev) * 0.5).half()
    kpad = (k * k * cin + 63) // 64 * 64
    wt = (torch.randn((2, cout, kpad), device=dev) * 100.0).half()
    sc = torch.full((cout,), 1e-4, dtype=torch.float32, device=dev)
    b = torch.zeros(cout, dtype=torch.float32, device=dev)
    out = torch.zeros((2, n, ho + 2 * ih, wo + 2 * ih, cout), dtype=torch.float16, device=dev)
    r = torch.randn((2, n, ho + 2 * ih, wo + 2 * ih, cout), device=dev).half() if res else None
    u = torch.randn((2, n, ho // 2 + 2 * ih, wo // 2 + 2 * ih, cout), device=dev).half() if up else None
    flop = 2.0 * n * ho * wo * k * k * cin * cout

    def call(v):
        return lib.rs_op_conv2d_split(C.c_void_p(x.data_ptr()), x[0].numel(), C.c_void_p(wt.data_ptr()), wt[0].numel(), C.c_void_p(sc.data_ptr()), C.c_void_p(b.data_ptr()),
                                      C.c_void_p(out.data_ptr()), out[0].numel(), C.c_void_p(r.data_ptr()) if r is not None else None, r[0].numel() if r is not None else 0,
                                      C.c_void_p(u.data_ptr()) if u is not None else None, u[0].numel() if u is not None else 0,
                                      n, h, w, cin, ih if ih >= pad else pad, k, k, stride, pad, cout, kpad, ih, 1, 0, 0, v, None)

    cand = [-1] + [v for v in (0, 7, 8, 1, 3, 14, 10, 4, 12, 15, 16, 17)
                   if (v in (0, 7, 3) and cout % 128 == 0) or (v in (4, 10, 14, 12, 15, 16, 17) and cout % 256 == 0) or (v in (1, 8) and cout % 64 == 0)]
    res_t = {}
    for v in cand:
        if call(v) != 0:
            continue
        for _ in range(8):
            call(v)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(60):
            call(v)
        e1.record()
        torch.cuda.synchronize()
        res_t[v] = e0.elapsed_time(e1) / 60 * 1e3
    best = min((v for v in res_t if v >= 0), key=res_t.get)
    print(f"{name:34s} " + " ".join(f"{NAMES[v]}={res_t[v]:6.1f}" for v in res_t) + f"  | best {NAMES[best]} {flop / res_t[best] / 1e6:5.0f} TF, rule {res_t[-1] / res_t[best]:.2f}x", flush=True)
