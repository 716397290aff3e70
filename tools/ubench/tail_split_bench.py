"""Time rs_op_bneck_tail_split (csrc/bneck_split.hip) on the shapes the engine runs at batch 16 of 800 x 800 tiles: res2 (200 x 200, width 64) and
res3 (100 x 100, width 128), with and without the next block's conv1.  Prints microseconds and TB/s of the algorithmic bytes."""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from proj_roadsurf_amd.engine import load_library, _check      # noqa: E402
from proj_roadsurf_amd.weights import split_planes              # noqa: E402


def main():
    lib = load_library()
    dev = torch.device("cuda:0")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    P = lambda t: C.c_void_p(t.data_ptr())
    for width, hw in ((64, 200), (128, 100)):
        cb, c4 = width, 4 * width
        g = torch.Generator().manual_seed(1)
        mk = lambda *s: (torch.randn(*s, generator=g).relu().half()).to(dev)
        t1 = torch.stack([mk(n, hw + 2, hw + 2, cb), mk(n, hw + 2, hw + 2, cb) * 2 ** -11])
        x = torch.stack([mk(n, hw + 2, hw + 2, c4), mk(n, hw + 2, hw + 2, c4) * 2 ** -11])
        out = torch.zeros_like(x)
        t1n = torch.zeros_like(t1)
        rng = np.random.default_rng(0)
        packs = []
        for rows, k in ((cb, 9 * cb), (c4, cb), (cb, c4)):
            ws, wsi = split_planes((rng.standard_normal((rows, k)) / np.sqrt(k)).astype(np.float32))
            packs.append((torch.from_numpy(ws).to(dev), torch.from_numpy(wsi).to(dev), torch.zeros(rows, device=dev)))
        for with_next in (True, False):
            def run():
                rc = lib.rs_op_bneck_tail_split(P(t1), t1[0].numel(), P(packs[0][0]), P(packs[0][1]), P(packs[0][2]), P(packs[1][0]), P(packs[1][1]), P(packs[1][2]),
                                                P(x), x[0].numel(), P(out), out[0].numel(),
                                                P(packs[2][0]) if with_next else None, P(packs[2][1]) if with_next else None, P(packs[2][2]) if with_next else None,
                                                P(t1n) if with_next else None, t1n[0].numel() if with_next else 0, None, 0, n, hw, hw, width, None)
                _check(lib, rc, "rs_op_bneck_tail_split")
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20
            e0.record()
            for _ in range(reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / reps * 1e3
            m = n * hw * hw
            nbytes = 4.0 * m * (cb + c4 + c4 + (cb if with_next else 0))
            flops = 2.0 * m * (9 * cb * cb + cb * c4 + (c4 * cb if with_next else 0))
            print(f"width {width:4d} {hw}x{hw} batch {n} next={int(with_next)}: {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s  {flops / us / 1e6:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
