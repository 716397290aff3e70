#!/usr/bin/env python3
"""Offline tile-variant sweep over every conv/GEMM shape of the batch-16 forward (rs_op_conv2d, HIP-event timed).
Prints, per shape, the time of each applicable variant; used to derive the selection rule in launch_conv()."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from proj_roadsurf_amd.engine import load_library
lib = load_library()
dev = torch.device("cuda:0")
NAMES = {0: "128x128", 1: "256x64", 4: "256x256", 7: "64x128", 8: "128x64", 10: "64x256"}

def bench(n, h, w_, cin, cout, k, stride, res, up, variant, iters=20):
    pad = k // 2
    ho, wo = (h + 2 * pad - k) // stride + 1, (w_ + 2 * pad - k) // stride + 1
    x = torch.randn((n, h + 2, w_ + 2, cin), dtype=torch.float16, device=dev)
    kpad = (k * k * cin + 63) // 64 * 64
    wt = torch.randn((cout, kpad), dtype=torch.float16, device=dev) * 0.05
    b = torch.zeros(cout, dtype=torch.float32, device=dev)
    out = torch.zeros((n, ho + 2, wo + 2, cout), dtype=torch.float16, device=dev)
    r = torch.randn((n, ho + 2, wo + 2, cout), dtype=torch.float16, device=dev) if res else None
    u = torch.randn((n, ho // 2 + 2, wo // 2 + 2, cout), dtype=torch.float16, device=dev) if up else None
    def call():
        return lib.rs_op_conv2d(C.c_void_p(x.data_ptr()), C.c_void_p(wt.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(out.data_ptr()),
                                C.c_void_p(r.data_ptr()) if r is not None else None, C.c_void_p(u.data_ptr()) if u is not None else None,
                                n, h, w_, cin, 1, k, k, stride, pad, cout, kpad, 1, 1, 0, 0, variant, 1, None)
    if call() != 0:
        return None
    for _ in range(2): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

N = 16
shapes = []
def add(name, h, cin, cout, k=1, stride=1, res=False, up=False, n=N, w=None):
    shapes.append((name, n, h, w or h, cin, cout, k, stride, res, up))
add("res2.0.conv1 64->64", 200, 64, 64); add("res2.x.conv1 256->64", 200, 256, 64); add("res2.conv2 3x3 64", 200, 64, 64, 3)
add("res2.conv3 64->256 +res", 200, 64, 256, res=True); add("res2.0.shortcut 64->256", 200, 64, 256)
add("res3.0.conv1 256->128 s2", 200, 256, 128, stride=2); add("res3.x.conv1 512->128", 100, 512, 128); add("res3.conv2 3x3 128", 100, 128, 128, 3)
add("res3.conv3 128->512 +res", 100, 128, 512, res=True); add("res3.0.shortcut 256->512 s2", 200, 256, 512, stride=2)
add("res4.0.conv1 512->256 s2", 100, 512, 256, stride=2); add("res4.x.conv1 1024->256", 50, 1024, 256); add("res4.conv2 3x3 256", 50, 256, 256, 3)
add("res4.conv3 256->1024 +res", 50, 256, 1024, res=True); add("res4.0.shortcut 512->1024 s2", 100, 512, 1024, stride=2)
add("res5.0.conv1 1024->512 s2", 50, 1024, 512, stride=2); add("res5.x.conv1 2048->512", 25, 2048, 512); add("res5.conv2 3x3 512", 25, 512, 512, 3)
add("res5.conv3 512->2048 +res", 25, 512, 2048, res=True); add("res5.0.shortcut 1024->2048 s2", 50, 1024, 2048, stride=2)
add("fpn_lateral5 2048->256", 25, 2048, 256); add("fpn_lateral4 1024->256 +up", 50, 1024, 256, up=True); add("fpn_lateral3 512->256 +up", 100, 512, 256, up=True)
add("fpn_lateral2 256->256 +up", 200, 256, 256, up=True)
add("3x3 256 @200 (fpn_out2/rpn2)", 200, 256, 256, 3); add("3x3 256 @100", 100, 256, 256, 3); add("3x3 256 @50", 50, 256, 256, 3); add("3x3 256 @25", 25, 256, 256, 3)
add("3x3 256 @13 (rpn6)", 13, 256, 256, 3)
add("mask fcn 3x3 256 (1600 rois 14^2)", 14, 256, 256, 3, n=1600)
add("fc1 12544->1024 (M=16000)", 16000, 12544, 1024, w=1, n=1); add("fc2 1024->1024", 16000, 1024, 1024, w=1, n=1)
for (name, n, h, w, cin, cout, k, stride, res, up) in shapes:
    cand = [v for v in (0, 4, 7, 8, 10, 1) if (v in (0, 7) and cout % 128 == 0) or (v in (4, 10) and cout % 256 == 0) or (v in (1, 8) and cout % 64 == 0)]
    if w == 1:
        # FC: rs_op_conv2d treats (h, w) as the image; halo 1 inflates it, fine for timing
        pass
    res_t = {}
    for v in cand:
        t = bench(n, h, w, cin, cout, k, stride, res, up, v)
        if t is not None:
            res_t[v] = t
    best = min(res_t, key=res_t.get)
    print(f"{name:36s} " + "  ".join(f"{NAMES[v]}={res_t[v]:7.1f}" for v in res_t) + f"   best {NAMES[best]}", flush=True)
