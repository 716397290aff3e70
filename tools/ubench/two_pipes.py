"""Does the chip gain from TWO independent lane pipelines (each: two engines on one shared wide stream) fed alternately?  The lockstep of one
wide stream makes every CU reach a conv_deep tile's prologue / epilogue together; kernels of two queues decorrelate that.
usage: two_pipes.py [pipes] [lanes per pipe] [precision] [batch] [steps]"""
import sys
import time

sys.path.insert(0, ".")
import numpy as np      # noqa: E402
import torch            # noqa: E402

from proj_roadsurf_amd.engine import LanePipeline      # noqa: E402
from proj_roadsurf_amd.spec import EngineSpec          # noqa: E402
from proj_roadsurf_amd.synthetic import synthetic_tiles  # noqa: E402
from proj_roadsurf_amd.weights import synthetic_weights  # noqa: E402


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    prec = sys.argv[3] if len(sys.argv) > 3 else "split"
    B = int(sys.argv[4]) if len(sys.argv) > 4 else 16
    steps = int(sys.argv[5]) if len(sys.argv) > 5 else 40
    spec = EngineSpec(num_classes=2, precision=prec)
    W = synthetic_weights(spec, seed=0)
    tiles = synthetic_tiles(B, 512, 512, 3, seed=1234)
    pipes = [LanePipeline(spec, W, (512, 512, 3), max_batch=B, device=0, lanes=L) for _ in range(P)]
    ptrs = [[e.upload_tiles(tiles) for e in p.engines] for p in pipes]

    def run(n):
        for k in range(n):
            p = pipes[k % P]
            p.submit(ptrs[k % P][p.k % L], B)
        for p in pipes:
            p.flush()
        torch.cuda.synchronize()

    run(3 * P * L)
    t0 = time.perf_counter()
    run(steps)
    dt = time.perf_counter() - t0
    print(f"pipes {P} x lanes {L} {prec} batch {B}: {steps * B / dt:8.1f} tiles/s  ({dt / steps * 1e3:.2f} ms per batch)", flush=True)
    for p in pipes:
        p.close()


if __name__ == "__main__":
    main()
