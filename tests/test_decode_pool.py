"""Process tile decoder (proj_roadsurf_amd/decode_pool.py): CPU-only."""
import os

import numpy as np
import pytest
from PIL import Image

from proj_roadsurf_amd.decode_pool import DecodePool, read_tile
from proj_roadsurf_amd.shard import run_sharded


def _write_tiles(tmp_path, n, shape=(40, 48, 3), seed=0):
    rng = np.random.default_rng(seed)
    paths = []
    for i in range(n):
        a = rng.integers(0, 256, shape, dtype=np.uint8)
        p = os.path.join(tmp_path, f"18_{i}_7.tif")
        Image.fromarray(a).save(p)
        paths.append(p)
    return paths


def test_decode_pool_batches_equal_read_tile(tmp_path):
    """Every chunk arrives as one (n, H, W, C) array equal to read_tile of its files (BGR order), ragged last chunk included, with
    more chunks than slab groups so that groups are reused."""
    paths = _write_tiles(str(tmp_path), 23)
    chunks = [paths[i:i + 3] for i in range(0, 23, 3)]
    with DecodePool(2, 3, (40, 48, 3), depth=2) as pool:
        seen = 0
        for c, b in zip(chunks, pool.batches(chunks)):
            assert b.shape == (len(c), 40, 48, 3) and b.dtype == np.uint8
            for p, im in zip(c, b):
                assert np.array_equal(im, read_tile(p))
            seen += len(c)
        assert seen == 23


def test_decode_pool_reports_a_tile_of_another_shape(tmp_path):
    paths = _write_tiles(str(tmp_path), 4)
    Image.fromarray(np.zeros((20, 20, 3), np.uint8)).save(paths[2])
    with DecodePool(2, 2, (40, 48, 3)) as pool:
        with pytest.raises(ValueError, match="tile shape"):
            list(pool.batches([paths[:2], paths[2:]]))


def test_run_sharded_with_a_prepared_source(tmp_path):
    """shard.run_sharded(prepared_source=DecodePool.batches): the stream consumer sees the decoded arrays, results keep item order."""
    paths = _write_tiles(str(tmp_path), 10, seed=3)
    with DecodePool(2, 4, (40, 48, 3)) as pool:
        def stream(batches):
            for b in batches:
                assert isinstance(b, np.ndarray) and b.ndim == 4
                yield [int(im.sum()) for im in b]
        out = run_sharded(paths, None, 4, finish=lambda items, raw: list(zip(items, raw)), predict_stream=stream,
                          prepared_source=lambda chunks: pool.batches(chunks))
    assert out == [(p, int(read_tile(p).sum())) for p in paths]


def test_register_host_buffer_without_a_device_is_a_clean_no():
    """engine.register_host_buffer pins the decode slab for direct uploads; where the runtime cannot pin (no HIP device here) it
    says False and the engine keeps staging through its own pinned buffer."""
    from proj_roadsurf_amd import engine as E
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a device is present: covered by the GPU CLI test")
    except ImportError:
        pass
    a = np.zeros((4, 8, 8, 3), np.uint8)
    assert E.register_host_buffer(a) is False and not E._REGISTERED_HOST
