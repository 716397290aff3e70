#!/usr/bin/env python3
"""Process-level drop-in for the STDL object-detector's ``make_detections.py`` as proj-roadsurf invokes it
(R:README.md:78): same argv (one YAML path), same YAML section ``make_detections.py:``
(R:config/config_obj_detec.yaml:74-90), same output files
(``<dataset>_detections_at_<thr>_threshold.gpkg`` with columns ``score``, ``det_class``, ``geometry`` --
R:config/config_obj_detec.yaml:100-103; plus a ``.geojson`` twin).

    python -m proj_roadsurf_amd.make_detections config/config_obj_detec.yaml
    python -m torch.distributed.run --nproc-per-node 8 -m proj_roadsurf_amd.make_detections config/config_obj_detec.yaml

Host Python only does what BASELINE.json:north_star leaves on the host: YAML/COCO-JSON I/O, tile decode,
weight loading, mask polygonisation and file writing.  The detector itself is ``librs_engine.so``.
With several ranks (RANK/WORLD_SIZE in the environment) the tile list of every dataset is sharded, one
GPU per rank, no collectives on the data path; rank 0 gathers the features and writes the files.
"""
from __future__ import annotations

import argparse
import json
import logging
import os
import sys
import time
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import yaml

from .decode_pool import DecodePool, TileShapeError, read_tile, tile_header_shape
from .gpkg import GpkgWriter
from .shard import run_sharded
from .spec import load_d2_yaml
from .vectorize import instances_to_features, instances_to_gpkg_rows
from .weights import infer_num_classes, load_checkpoint, synthetic_weights

SECTION = "make_detections.py"
log = logging.getLogger("make_detections")


def tile_extent(meta: Dict[str, Any], file_name: str) -> Tuple[Optional[Sequence[float]], Optional[int]]:
    """Georeference of one tile from ``img_metadata.json`` (R:config/config_obj_detec.yaml:78): returns
    ((xmin, ymin, xmax, ymax), epsg) or (None, None).  Keys are matched by path or basename; the extent may be
    stored as ``extent``/``bbox`` [xmin, ymin, xmax, ymax] or as west/south/east/north."""
    rec = meta.get(file_name) or meta.get(os.path.basename(file_name))
    if rec is None:
        for k, v in meta.items():
            if os.path.basename(k) == os.path.basename(file_name):
                rec = v
                break
    if not isinstance(rec, dict):
        return None, None
    ext = rec.get("extent") or rec.get("bbox")
    if isinstance(ext, dict):
        ext = [ext.get("xmin"), ext.get("ymin"), ext.get("xmax"), ext.get("ymax")]
    if ext is None and all(k in rec for k in ("west", "south", "east", "north")):
        ext = [rec["west"], rec["south"], rec["east"], rec["north"]]
    epsg = None
    for k in ("crs", "srs", "epsg"):
        if k in rec:
            s = str(rec[k])
            digits = "".join(ch for ch in s.split(":")[-1] if ch.isdigit())
            epsg = int(digits) if digits else None
            break
    return (list(map(float, ext)) if ext is not None else None), epsg


def _close_pools(pools: Dict[Tuple[int, ...], DecodePool]) -> None:
    """Unpin every decode slab BEFORE its shared memory goes away (a registered range that outlives its mapping would let a later
    array at a recycled address pass for pinned -- engine.Engine.upload_async), then stop the workers.  Error and normal path."""
    from .engine import unregister_host_buffer
    for pl in pools.values():
        try:
            if pl.slab is not None:
                unregister_host_buffer(pl.slab)
        finally:
            pl.close()
    pools.clear()


def thr_tag(thr: float) -> str:
    return str(thr).replace(".", "dot")


def host_pool_sizes(host_workers: Optional[int], decode_procs: Optional[int], vector_threads: Optional[int],
                    cores: Optional[int] = None, local_world: Optional[int] = None) -> Tuple[int, int, int, int]:
    """Sizes of the three host pools of one rank.  Measured on a one-GPU box (16 cores for the rank): 4 decode processes + 4 host threads + 4
    vectoriser threads feed one MI355X at 1 880 tiles/s with the forward thread waiting 0.2 ms per batch for input.
    With N ranks on one host every rank gets cores / N of them (``LOCAL_WORLD_SIZE`` from torchrun); below 16 cores per rank the
    defaults shrink proportionally (a quarter of the share each, at least 1) so that 8 ranks never oversubscribe a small host.
    Values given on the command line are kept."""
    if cores is None:
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
    if local_world is None:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1)
    share = max(1, cores // max(1, local_world))
    auto = min(4, max(1, share // 4))
    hw = auto if host_workers is None else host_workers
    return hw, (auto if decode_procs is None else decode_procs), (auto if vector_threads is None else vector_threads), share


def main(argv: Optional[Sequence[str]] = None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("config_file", help="YAML with a 'make_detections.py' section (R:config/config_obj_detec.yaml)")
    ap.add_argument("--batch", type=int, default=16, help="tiles per engine call")
    ap.add_argument("--lanes", type=int, default=2, help="engine contexts fed alternately, each on its own stream (engine.LanePipeline); 2 keeps the GPU busy while the host collects a batch")
    ap.add_argument("--synthetic-weights", action="store_true",
                    help="use seeded synthetic weights instead of model_weights.pth_file (demo / smoke tests)")
    ap.add_argument("--host-workers", type=int, default=None,
                    help="threads for tile decode / vectorisation around the GPU call (default: 4, fewer when this rank's share of the host's cores is under 16)")
    ap.add_argument("--decode-procs", type=int, default=None,
                    help="worker PROCESSES that decode tiles into shared memory (decode_pool.DecodePool); 0 = decode on the --host-workers threads (default: as --host-workers)")
    ap.add_argument("--vector-threads", type=int, default=None, help="threads inside one rs_vectorize_masks call (default: as --host-workers)")
    ap.add_argument("--precision", choices=("fp16", "split", "fp32"), default="fp16",
                    help="fp16: fp16 operands / fp32 accumulate on the matrix cores (fastest; meets the SURVEY 8d tolerance on a trained detector); "
                         "split: the reference's fp32 results on the fp16 matrix cores -- every operand as hi + lo fp16 planes, three products, about 2.7x "
                         "slower than fp16 (DESIGN.md section 3.1d); fp32: fp32 activations and weights on the fp32 matrix cores, about 7x slower (section 3.1c)")
    ap.add_argument("--geojson", action="store_true", help="also write <dataset>_detections_..._threshold.geojson (slow: Python feature dicts)")
    ap.add_argument("--tagged-samples", type=int, default=10,
                    help="tagged preview PNGs per dataset in sample_tagged_img_subfolder (0 = none)")
    ap.add_argument("--max-tiles", type=int, default=0, help="debug: only the first N tiles of every dataset")
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(levelname)s %(message)s", stream=sys.stderr)
    args.host_workers, args.decode_procs, args.vector_threads, share = host_pool_sizes(args.host_workers, args.decode_procs, args.vector_threads)
    logging.getLogger("make_detections").info("host pools of this rank: %d decode processes, %d host threads, %d vectoriser threads (%d cores for the rank)",
                                              args.decode_procs, args.host_workers, args.vector_threads, share)

    with open(args.config_file) as f:
        cfg = yaml.safe_load(f)[SECTION]
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=os.environ.get("RS_DIST_BACKEND", "gloo"), rank=rank, world_size=world)
        import torch
        local_rank = local_rank % max(1, torch.cuda.device_count())      # several ranks may share a card (tests on a one-GPU box); counting devices does not initialise the GPU
        logging.getLogger("make_detections").info("rank %d of %d ranks on device %d", rank, world, local_rank)

    os.chdir(cfg["working_directory"])
    os.makedirs(cfg.get("log_subfolder", "logs"), exist_ok=True)
    meta: Dict[str, Any] = {}
    if cfg.get("image_metadata_json") and os.path.exists(cfg["image_metadata_json"]):
        with open(cfg["image_metadata_json"]) as f:
            meta = json.load(f)
    coco = {}
    for name, path in cfg["COCO_files"].items():
        with open(path) as f:
            coco[name] = json.load(f)
    cats = sorted({c["id"] for d in coco.values() for c in d.get("categories", [])})
    thr = float(cfg.get("score_lower_threshold", 0.05))
    rdp_cfg = cfg.get("rdp_simplification", {}) or {}

    if args.synthetic_weights:
        spec = load_d2_yaml(cfg["detectron2_config_file"], num_classes=max(len(cats), 1)).replace(score_thresh_test=thr)
        W = synthetic_weights(spec, seed=0)
    else:
        W = load_checkpoint(cfg["model_weights"]["pth_file"])
        k = infer_num_classes(W)
        if cats and len(cats) != k:
            raise SystemExit(f"checkpoint has {k} classes but the COCO files define {len(cats)} categories")
        spec = load_d2_yaml(cfg["detectron2_config_file"], num_classes=k).replace(score_thresh_test=thr)

    if args.precision != spec.precision:
        spec = spec.replace(precision=args.precision)
        log.info("inference in %s", {"fp32": "reference precision (fp32 on the matrix cores)", "split": "reference-equivalent precision (hi + lo fp16 operand planes, three products)",
                                     "fp16": "fp16 operands / fp32 accumulate"}[args.precision])
    from .engine import Predictor      # fails loudly without librs_engine.so / a HIP device
    predictor = Predictor(spec, W, max_batch=args.batch, device=local_rank, lanes=args.lanes)

    # decode (PIL), GPU forward and vectorisation (C++, rs_vectorize_masks) overlap across batches (shard.run_sharded)
    busy = {"decode": 0.0, "predict": 0.0, "vectorise": 0.0}      # seconds summed over the threads that ran each stage

    def prepare(entries: Sequence[dict]) -> List[Any]:
        t = time.perf_counter()
        out = [read_tile(e["file_name"]) for e in entries]
        busy["decode"] += time.perf_counter() - t
        return out

    def predict_batch(ims: Sequence[Any]) -> List[Any]:
        t = time.perf_counter()
        out = predictor.predict_batch(ims)
        busy["predict"] += time.perf_counter() - t
        return out

    def predict_stream(batches):
        # the whole dataset through one lane pipeline: batch k+1 uploads and runs while batch k's results come back
        it = predictor.predict_stream(batches)
        while True:
            t = time.perf_counter()
            out = next(it, None)
            busy["predict"] += time.perf_counter() - t
            if out is None:
                return
            yield out

    def finish(entries: Sequence[dict], outs: List[Any]) -> List[Any]:
        # per tile: (GeoPackage rows, bbox[, GeoJSON features]) -- masks -> polygons -> RDP -> georeferenced blobs in C++
        t_fin = time.perf_counter()
        res = []
        for e, o in zip(entries, outs):
            ext, epsg_t = tile_extent(meta, e["file_name"])
            name = os.path.basename(e["file_name"])
            rdp_on, eps = bool(rdp_cfg.get("enabled", False)), float(rdp_cfg.get("epsilon", 0.75))
            rows, bbox = instances_to_gpkg_rows(o["instances"], name, ext, rdp_on, eps, srs_id=cur_srs["id"], threads=args.vector_threads)
            feats = instances_to_features(o["instances"], name, ext, rdp_on, eps, threads=args.vector_threads) if args.geojson else None
            res.append((rows, bbox, feats))
        busy["vectorise"] += time.perf_counter() - t_fin
        return res

    cur_srs = {"id": -1}
    pools: Dict[Tuple[int, ...], DecodePool] = {}       # process decoders by tile shape (spawned once, reused by every dataset of that shape)
    for dataset, d in coco.items():
        images = d.get("images", [])
        if args.max_tiles:
            images = images[: args.max_tiles]
        t0 = time.time()
        epsg = None
        for e in images:
            _, epsg = tile_extent(meta, e["file_name"])
            if epsg:
                break
        cur_srs["id"] = int(epsg) if epsg else -1
        source = None
        # The process decoder serves ONE tile shape per dataset (its slab is a fixed (n, H, W, C) array); the thread path groups runs of
        # equal shape as the reference's per-tile cv2.imread loop effectively does.  Decide before starting: COCO width/height when every
        # entry carries them, the file headers otherwise; a dataset of mixed shapes keeps the thread path.
        if args.decode_procs > 0 and images:
            shape = tuple(read_tile(images[0]["file_name"]).shape)
            if all("width" in e and "height" in e for e in images):
                uniform = all((int(e["height"]), int(e["width"])) == shape[:2] for e in images)
            else:
                uniform = all(tile_header_shape(e["file_name"]) == shape for e in images)
            if not uniform:
                log.info("%s: tiles of several shapes -- decoding on the %d host threads instead of the process decoder", dataset, args.host_workers)
            elif shape not in pools:
                from .engine import register_host_buffer
                pools[shape] = DecodePool(args.decode_procs, args.batch, shape, spare=predictor.lanes + 2)
                if not register_host_buffer(pools[shape].slab):      # pinned: batches go from the slab to the device without a staging copy
                    log.info("the decode slab could not be pinned; batches are staged through the engine's own pinned buffer")
            pool = pools.get(shape) if uniform else None
            if uniform:
                t_p = time.time()
                predictor.prepare(shape)                # engines built + kernels loaded while the decoder processes start
                log.info("%s: lane pipeline for %s tiles ready in %.2f s", dataset, shape, time.time() - t_p)

            def pool_source(chunks, pool=pool):
                it = pool.batches(chunks, key=lambda e: os.path.abspath(e["file_name"]))
                while True:
                    t = time.perf_counter()
                    b = next(it, None)
                    busy["decode"] += time.perf_counter() - t          # here: time the forward thread WAITED for decoded tiles
                    if b is None:
                        return
                    yield b
            source = pool_source if pool is not None else None
        try:
            try:
                per_tile = run_sharded(images, predict_batch, args.batch, rank, world, gather=False, prepare=prepare, finish=finish,
                                       workers=args.host_workers, predict_stream=predict_stream, prepared_source=source)
            except TileShapeError as ex:
                # the COCO sizes (or the band count of the first tile) did not hold for every file: this dataset again on the thread path
                log.warning("%s: %s -- running the dataset again with thread decoding", dataset, ex)
                predictor.close()
                predictor = Predictor(spec, W, max_batch=args.batch, device=local_rank, lanes=args.lanes)
                per_tile = run_sharded(images, predict_batch, args.batch, rank, world, gather=False, prepare=prepare, finish=finish,
                                       workers=args.host_workers, predict_stream=predict_stream, prepared_source=None)
        except BaseException:
            _close_pools(pools)
            raise
        # Every rank writes the rows of ITS block of tiles into its own GeoPackage shard (rank 0: the output file itself); rank 0 then appends the
        # other ranks' shards in rank order = tile order with SQLite ATTACH -- no row travels between processes ("each rank ...
        # its own output shard; host merges shards into one GeoPackage per dataset").  The ranks of one node share the working directory.
        base = f"{dataset}_detections_at_{thr_tag(thr)}_threshold"
        t_w = time.time()
        shard_path = base + ".gpkg" if rank == 0 else f"{base}.rank{rank}.gpkg"
        if rank != 0 and os.path.exists(shard_path):
            os.remove(shard_path)
        gw = GpkgWriter(shard_path, table=base, epsg=epsg)
        for rows, bbox, _ in per_tile:
            gw.add_rows(rows, bbox)
        if args.geojson and world > 1:
            with open(f"{base}.rank{rank}.features.json", "w") as f:
                f.write(json.dumps([ft for _, _, fs in per_tile for ft in fs]))
        if rank != 0:
            gw.close()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()                                   # every shard is complete on disk
        if rank != 0:
            continue
        for r in range(1, world):
            sp = f"{base}.rank{r}.gpkg"
            gw.append_shard(sp)
            os.remove(sp)
        n = gw.close()
        dt_write = time.time() - t_w
        if args.geojson:
            feats = [f for _, _, fs in per_tile for f in fs]
            for r in range(world if world > 1 else 0):
                fp = f"{base}.rank{r}.features.json"
                if r > 0:
                    with open(fp) as f:
                        feats += json.load(f)
                os.remove(fp)
            with open(base + ".geojson", "w") as f:
                f.write(json.dumps({"type": "FeatureCollection", "features": feats}))   # dumps() = C encoder; dump() streams through the slow Python one
        dt = time.time() - t0
        log.info("%s: %d tiles -> %d features in %.1f s (%.1f tiles/s) -> %s.gpkg", dataset, len(images), n, dt,
                 len(images) / max(dt, 1e-9), base)
        log.info("%s: stage busy time (summed over threads) decode %.2f s, predict %.2f s, vectorise %.2f s, write %.2f s",
                 dataset, busy["decode"], busy["predict"], busy["vectorise"], dt_write)
        for shp, pl in getattr(predictor, "_pipes", {}).items():
            tm = getattr(pl, "timing", None)
            if tm and tm["batches"]:
                log.info("%s: forward thread per batch of %s tiles: wait for input %.2f ms, wait for results %.2f, upload %.2f, enqueue %.2f, collect %.2f (%d batches)",
                         dataset, shp, *(1e3 * tm[k] / tm["batches"] for k in ("pull", "wait_results", "upload", "enqueue", "collect")), tm["batches"])
        busy.update({k: 0.0 for k in busy})
        sub = cfg.get("sample_tagged_img_subfolder")
        if sub and args.tagged_samples > 0:
            # the reference tags the first images of every dataset with their predictions (R:config/config_obj_detec.yaml:77)
            from .tagging import draw_instances
            os.makedirs(sub, exist_ok=True)
            names = [c["name"] for c in sorted(d.get("categories", []), key=lambda c: c["id"])]
            for e in images[: args.tagged_samples]:
                im = read_tile(e["file_name"])
                inst = predictor(im)["instances"]
                rgb = im[:, :, ::-1][:, :, :3]              # read_tile hands the file's bands over in reverse (cv2) order
                png = os.path.join(sub, f"{dataset}_det_{os.path.splitext(os.path.basename(e['file_name']))[0]}.png")
                draw_instances(rgb, inst.pred_boxes, inst.pred_classes, inst.scores, inst.pred_masks, names or None).save(png)
            log.info("%s: %d tagged sample images -> %s/", dataset, min(len(images), args.tagged_samples), sub)
    _close_pools(pools)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
