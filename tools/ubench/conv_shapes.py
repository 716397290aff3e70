#!/usr/bin/env python3
"""Time rs_op_conv2d on the HBM-bound layer shapes of the backbone (batch 16) for a few tile variants."""
import ctypes as C, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from proj_roadsurf_amd.engine import load_library, _check
lib = load_library()
dev = torch.device("cuda:0")

def bench(name, n, hw, cin, cout, k, res, variant, iters=30):
    pad = k // 2
    x = torch.randn((n, hw + 2, hw + 2, cin), dtype=torch.float16, device=dev)
    kpad = (k * k * cin + 63) // 64 * 64
    w = (torch.randn((cout, kpad), dtype=torch.float16, device=dev) * 0.05)
    b = torch.zeros(cout, dtype=torch.float32, device=dev)
    out = torch.zeros((n, hw + 2, hw + 2, cout), dtype=torch.float16, device=dev)
    r = torch.randn((n, hw + 2, hw + 2, cout), dtype=torch.float16, device=dev) if res else None
    def call():
        rc = lib.rs_op_conv2d(C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(out.data_ptr()),
                              C.c_void_p(r.data_ptr()) if r is not None else None, None, n, hw, hw, cin, 1, k, k, 1, pad, cout, kpad, 1, 1, 0, 0, variant, 1, None)
        _check(lib, rc, "conv")
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    m = n * hw * hw
    by = m * (cin + cout * (2 if res else 1)) * 2
    fl = 2.0 * m * k * k * cin * cout
    print(f"{name:34s} v{variant:2d} {ms*1e3:8.1f} us  {by/ms/1e6:7.0f} GB/s  {fl/ms/1e9:7.0f} TFLOP/s", flush=True)

for v in (7, 9, 10):
    bench("res2.conv3 64->256 +res 200^2", 16, 200, 64, 256, 1, True, v)
    bench("res2.conv3 64->256 no res", 16, 200, 64, 256, 1, False, v)
    bench("res2.conv1 256->64... (as 256->256)", 16, 200, 256, 256, 1, False, v)
    bench("res3.conv3 128->512 +res 100^2", 16, 100, 128, 512, 1, True, v)
    bench("res4.conv3 256->1024 +res 50^2", 16, 50, 256, 1024, 1, True, v)
bench("res2.conv1 256->64 200^2", 16, 200, 256, 64, 1, False, 1)
bench("fpn_out2 3x3 256->256 200^2", 16, 200, 256, 256, 3, False, 0)
bench("fpn_out2 3x3 256->256 200^2", 16, 200, 256, 256, 3, False, 4)
