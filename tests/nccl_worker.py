"""Worker of tests/test_gpu_trainer.py::test_rccl_allreduce_path_on_one_gpu: a ONE-rank "nccl" (= RCCL) process group on this box's
GPU, so that the device-side form of Trainer.allreduce_gradients runs for real -- zero-copy views of the flat gradient buffer
handed to torch.distributed, asynchronous all-reduces enqueued behind the per-bucket completion events, the trainer's stream
waiting for the collectives before the SGD step.  A one-rank SUM must leave the gradient bit-identical, and the following
step must equal the step of a trainer that never called it.  Prints NCCL_OK."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    from proj_roadsurf_amd.engine import Trainer
    from proj_roadsurf_amd.spec import EngineSpec
    from proj_roadsurf_amd.synthetic import synthetic_tiles
    from proj_roadsurf_amd.weights import synthetic_weights

    spec = EngineSpec(num_classes=2, min_size_test=320, max_size_test=533)
    W = synthetic_weights(spec, seed=0)
    tiles = synthetic_tiles(1, 256, 256, 3, seed=77)
    gb = [np.array([[20.0, 30.0, 120.0, 160.0], [150.0, 40.0, 300.0, 130.0]], np.float32)]
    gc = [np.array([0, 1])]
    polys = [[[np.array([b[0], b[1], b[2], b[1], b[2], b[3], b[0], b[3]], np.float64)] for b in gb[0]]]
    out = {}
    for name, reduce_ in (("plain", False), ("reduced", True)):
        tr = Trainer(spec, W, (256, 256, 3), batch=1, loss_scale=256.0)
        tr.set_targets(gb, gc)
        tr.forward_trunk(tr.upload_tiles(tiles), 1)
        tr.rpn_forward(1)
        tr.roi_step(1, 5)
        tr.mask_forward(1)
        targets, _ = tr.mask_entries(polys, 1)
        tr.mask_backward(1, targets)
        tr.rpn_step(1, 5)
        tr.backward_trunk(1)
        if reduce_:
            tr.allreduce_gradients(force=True)          # enqueued behind the backward pass: nothing has been synchronised yet
        g = tr.flat("grad")
        tr.apply_sgd(0.01, 0.9, 1e-4)
        out[name] = (g, tr.flat("master"))
        tr.close()
    # RoIAlign-backward's float atomics make two runs of the SAME step differ in the last bits of the trunk gradients; the head
    # buckets are deterministic.  So: heads bit-identical, everything close, and the divisor (world = 1) changed nothing.
    (g0, m0), (g1, m1) = out["plain"], out["reduced"]
    assert float(np.abs(g1).max()) > 0 and np.isfinite(g1).all()
    rel = float(np.linalg.norm(g1 - g0) / np.linalg.norm(g0))
    assert rel < 1e-3, rel
    relm = float(np.linalg.norm(m1 - m0) / np.linalg.norm(m0))
    assert relm < 1e-5, relm
    dist.destroy_process_group()
    print("NCCL_OK", rel, relm, flush=True)


if __name__ == "__main__":
    main()
