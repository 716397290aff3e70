"""Multi-process (world_size 2, gloo, CPU) test of the tile-sharding path used for N>1 GPUs: every
tile is processed exactly once, results come back on rank 0 in tile order, no data-path collective."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from proj_roadsurf_amd.shard import run_sharded, shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions():
    for n in (0, 1, 7, 16, 10001):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _fake_predict(batch):
    # stands in for Predictor.predict_batch: a deterministic function of the tile content
    return [{"sum": int(t.sum()), "n": int(t.shape[0])} for t in batch]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tiles = [np.full((4, 4, 3), i, np.uint8) for i in range(11)]     # ragged: 11 tiles over 2 ranks, batch 4
    calls = []

    def pb(b):
        calls.append(len(b))
        return _fake_predict(b)

    out = run_sharded(tiles, pb, batch=4, rank=rank, world=world)
    dist.barrier()
    q.put((rank, out, calls))
    dist.destroy_process_group()


def test_run_sharded_world2_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(2):
        r, out, calls = q.get(timeout=120)
        got[r] = (out, calls)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    out0, calls0 = got[0]
    out1, calls1 = got[1]
    assert out1 is None
    assert [o["sum"] for o in out0] == [i * 48 for i in range(11)]          # tile order preserved, each exactly once
    assert calls0 == [4, 2] and calls1 == [4, 1]                              # rank 0: tiles 0-5, rank 1: tiles 6-10


def _worker8(rank, world, port, q, n_tiles):
    """The CLI's form of the call (make_detections.main): three stages + the streaming predictor + gather on rank 0."""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    entries = [{"file_name": f"t{i}.tif", "v": i} for i in range(n_tiles)]
    seen = []

    def prepare(es):
        return [np.full((2, 2, 3), e["v"], np.uint8) for e in es]

    def stream(batches):
        for b in batches:
            seen.append(len(b))
            yield [{"sum": int(t.sum())} for t in b]

    def finish(es, raw):
        return [(e["file_name"], r["sum"]) for e, r in zip(es, raw)]

    out = run_sharded(entries, None, 4, rank, world, prepare=prepare, finish=finish, workers=2, predict_stream=stream)
    dist.barrier()
    q.put((rank, out, seen))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_tiles", [5, 0, 19])
def test_run_sharded_world8_gloo_with_empty_shards(n_tiles):
    """BASELINE configs[2] shape of the job (tiles sharded over the 8 GPUs of a node, rank 0 gathers and writes) with FEWER tiles
    than ranks: ranks 5..7 of 8 get an empty block, take part in the gather all the same, and rank 0 still sees every tile exactly
    once in tile order; also no tile at all, and a ragged 19."""
    world = 8
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q, n_tiles)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, out, seen = q.get(timeout=180)
        got[r] = (out, seen)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got[0][0] == [(f"t{i}.tif", i * 12) for i in range(n_tiles)]
    assert all(got[r][0] is None for r in range(1, world))
    per_rank = [sum(got[r][1]) for r in range(world)]
    assert per_rank == [shard_range(n_tiles, r, world)[1] - shard_range(n_tiles, r, world)[0] for r in range(world)]
    if n_tiles == 5:
        assert per_rank == [1, 1, 1, 1, 1, 0, 0, 0]


def test_run_sharded_single_process():
    tiles = [np.full((2, 2, 3), i, np.uint8) for i in range(5)]
    out = run_sharded(tiles, _fake_predict, batch=2)
    assert [o["sum"] for o in out] == [i * 12 for i in range(5)]
    with pytest.raises(RuntimeError):
        run_sharded(tiles, lambda b: [], batch=2)


def test_run_sharded_three_stage_pipeline_keeps_order_and_overlaps():
    """prepare / predict / finish: results come back in item order, prepare sees every ITEM exactly once (one pool task per item,
    up to four batches ahead), predict and finish see every batch exactly once, and prepare of batch k+1 has started before
    predict of batch k returns."""
    import threading, time
    items = list(range(23))
    log, lock = [], threading.Lock()

    def prepare(b):
        with lock:
            log.append(("prep", b[0]))
        time.sleep(0.01)
        return [x * 10 for x in b]

    def predict(p):
        with lock:
            log.append(("gpu", p[0] // 10))
        time.sleep(0.02)
        return [x + 1 for x in p]

    def finish(b, raw):
        with lock:
            log.append(("fin", b[0]))
        return [(x, r) for x, r in zip(b, raw)]

    out = run_sharded(items, predict, batch=4, prepare=prepare, finish=finish, workers=3)
    assert out == [(x, x * 10 + 1) for x in items]
    assert sorted(v for t, v in log if t == "prep") == items
    for tag in ("gpu", "fin"):
        assert sorted(v for t, v in log if t == tag) == [0, 4, 8, 12, 16, 20]
    assert log.index(("prep", 4)) < log.index(("gpu", 4)) and log.index(("prep", 8)) < log.index(("fin", 4)) + 3


def test_run_sharded_streaming_form_keeps_order_and_pulls_lazily():
    """``predict_stream``: the batches reach the predictor as ONE lazy iterator (decoded up to four batches ahead of what it has
    consumed) and the per-item results come back in item order, ragged last batch included."""
    from proj_roadsurf_amd.shard import run_sharded
    items = list(range(23))
    pulled = []

    def prepare(b):
        return [x * 10 for x in b]

    def stream(batches):
        held = []
        for b in batches:                     # a consumer that, like LanePipeline.run, yields two batches behind its input
            pulled.append(list(b))
            held.append(b)
            if len(held) > 2:
                yield [x + 1 for x in held.pop(0)]
        for b in held:
            yield [x + 1 for x in b]

    out = run_sharded(items, None, 4, prepare=prepare, finish=lambda b, raw: [(i, r) for i, r in zip(b, raw)], predict_stream=stream)
    assert out == [(i, i * 10 + 1) for i in items]
    assert [len(b) for b in pulled] == [4, 4, 4, 4, 4, 3]
