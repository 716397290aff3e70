"""Worker of tests/test_gpu_trainer.py::test_data_parallel_allreduce_sums_and_averages (one process per rank, gloo, every rank on
this box's one GPU; on a node the same calls run one rank per GPU over RCCL).  Checks, on every rank:
  * after ``allreduce_gradients`` the flat gradient is EXACTLY g0 + g1 (each rank's own gradient, gathered beforehand);
  * after ``apply_sgd`` the fp32 master weights are bit-identical on all ranks;
  * they are bit-identical to a single-process trainer stepping on the summed gradient with divisor 2 -- i.e. 1/world is
    applied exactly once;
  * the buckets tile the flat buffer and the bucketed path touched every value.
Prints DP_OK on success."""
import hashlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    from proj_roadsurf_amd.engine import Trainer, _check
    from proj_roadsurf_amd.spec import EngineSpec
    from proj_roadsurf_amd.synthetic import synthetic_tiles
    from proj_roadsurf_amd.weights import synthetic_weights

    spec = EngineSpec(num_classes=2, min_size_test=320, max_size_test=533)
    W = synthetic_weights(spec, seed=0)
    tiles = synthetic_tiles(world, 256, 256, 3, seed=77)
    gbs = [np.array([[20.0, 30.0, 120.0, 160.0], [150.0, 40.0, 300.0, 130.0]], np.float32), np.array([[100.0, 100.0, 260.0, 280.0]], np.float32)]
    gcs = [np.array([0, 1]), np.array([1])]
    gb, gc = [gbs[rank % 2]], [gcs[rank % 2]]
    polys = [[[np.array([b[0], b[1], b[2], b[1], b[2], b[3], b[0], b[3]], np.float64)] for b in gb[0]]]
    tr = Trainer(spec, W, (256, 256, 3), batch=1, loss_scale=256.0)
    bk = tr.buckets()
    assert [b[0] for b in bk] == ["heads", "fpn", "res5", "res4", "res3"], bk
    spans = sorted((o, o + c) for _, o, c in bk)
    assert spans[0][0] == 0 and spans[-1][1] == tr.param_count and all(a[1] == b[0] for a, b in zip(spans, spans[1:])), spans
    losses = tr.train_step(tiles[rank:rank + 1], gb, gc, polys, seed=5 + rank)
    assert all(np.isfinite(v) for v in losses.values()), losses
    g = tr.flat("grad")
    assert float(np.abs(g).max()) > 0
    every = [torch.empty(g.shape[0]) for _ in range(world)]
    dist.all_gather(every, torch.from_numpy(g))
    want = every[0].numpy().copy()
    for e in every[1:]:
        want += e.numpy()
    tr.allreduce_gradients()
    post = tr.flat("grad")
    assert np.array_equal(post, want), f"rank {rank}: reduced gradient != sum of the ranks' gradients (max |d| {np.abs(post - want).max()})"
    tr.apply_sgd(0.01, 0.9, 1e-4)
    assert not tr.overflowed()
    m = tr.flat("master")
    digest = hashlib.sha256(m.tobytes()).hexdigest()
    digests = [None] * world
    dist.all_gather_object(digests, digest)
    assert len(set(digests)) == 1, f"master weights differ across ranks: {digests}"
    tr.close()
    if rank == 0:
        ref = Trainer(spec, W, (256, 256, 3), batch=1, loss_scale=256.0)
        m0 = ref.flat("master")
        ref.write_flat_grad(want)
        _check(ref.lib, ref.lib.rs_trainer_set_grad_divisor(ref._h, float(world)), "rs_trainer_set_grad_divisor")
        ref.apply_sgd(0.01, 0.9, 1e-4)
        m1 = ref.flat("master")
        ref.close()
        assert np.array_equal(m1, m), "data-parallel step != single-process step on (g0 + g1) / world"
        assert not np.array_equal(m0, m1)
        # and the step really is the AVERAGED gradient: lr * (g0 + g1) / (world * loss_scale) + weight decay, first step (momentum 0)
        upd = m0 - m1
        expect = 0.01 * (want / (world * 256.0) + 1e-4 * m0)
        nz = np.abs(expect) > 1e-9
        rel = np.abs(upd[nz] - expect[nz]) / np.abs(expect[nz])
        assert float(np.median(rel)) < 1e-3, float(np.median(rel))
    dist.barrier()
    if rank == 0:
        print("DP_OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
