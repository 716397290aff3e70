#!/usr/bin/env python3
"""Drop-in for the STDL object-detector's ``train_model.py`` call (R:README.md:77): same argv (one YAML path), same YAML
section and keys (R:config/config_obj_detec.yaml:62-72), same artefacts -- ``<log_subfolder>/model_<iter>.pth`` and
``model_final.pth`` in detectron2's checkpoint layout ({"model": state_dict, "iteration": n}; directly consumable by
``make_detections.py`` via ``model_weights.pth_file``), ``metrics.json`` (one JSON object per logged iteration) -- with the
training step (GeneralizedRCNN forward, five losses, backward, SGD) running on the MI355X training engine
(``engine.Trainer`` -> ``rs_trainer_*``) instead of detectron2's ``DefaultTrainer``.

    python -m proj_roadsurf_amd.train_model <config.yaml> [--synthetic-weights] [--max-iter N]
    python -m torch.distributed.run --nproc-per-node 8 -m proj_roadsurf_amd.train_model <config.yaml>      # data parallel

What follows the reference / detectron2 0.6 and where it is pinned:
  * data: COCO JSON (polygons), FILTER_EMPTY_ANNOTATIONS (R:4), crowd annotations dropped, TrainingSampler (infinite seeded
    shuffle, rank-strided), ResizeShortestEdge + RandomFlip(horizontal, 0.5)  [EXT d2: data/{build,dataset_mapper,samplers}.py];
  * solver: SGD momentum 0.9, weight decay 1e-4, WarmupMultiStepLR (R:268-305), IMS_PER_BATCH 8 split over the ranks,
    checkpoint every CHECKPOINT_PERIOD  [EXT d2: solver/build.py, engine/defaults.py];
  * data parallel: one process per GPU, gradients all-reduced over RCCL/xGMI in one flat 175 MB fp32 buffer and averaged
    (DistributedDataParallel semantics).
MIN_SIZE_TRAIN's multi-scale "choice" (R:31-38) is drawn per image and the batch padded to its largest image, as DatasetMapper +
ImageList.from_tensors do (``MultiScaleTrainer.select_batch``).
Documented deviations: fp16 activations/weights with fp32 master weights and dynamic loss scaling (GradScaler's policy) instead of fp32 everywhere, the
model-zoo name of ``model_weights.model_zoo_checkpoint_url`` is resolved in detectron2's local cache layout
(``weights.resolve_zoo_checkpoint``: no download; else ``model_weights.pth_file`` or ``--synthetic-weights``).  Every TEST.EVAL_PERIOD iterations the validation loss and the COCO bbox / segm AP (coco_eval.py, a
restatement of pycocotools' COCOeval) of the `val` set are logged to metrics.json.
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

from .spec import load_d2_yaml, resize_shortest_edge_shape
from .weights import infer_num_classes, load_checkpoint, synthetic_weights

log = logging.getLogger("train_model")
SECTION = "train_model.py"


def load_solver(d2_yaml: str) -> Dict[str, Any]:
    """SOLVER / sampler keys of the detectron2 YAML with detectron2's defaults where the file is silent."""
    with open(d2_yaml) as f:
        cfg = yaml.safe_load(f) or {}
    s = cfg.get("SOLVER", {}) or {}
    m = cfg.get("MODEL", {}) or {}
    return {
        "base_lr": float(s.get("BASE_LR", 0.001)), "momentum": float(s.get("MOMENTUM", 0.9)), "weight_decay": float(s.get("WEIGHT_DECAY", 1e-4)),
        "gamma": float(s.get("GAMMA", 0.1)), "steps": tuple(int(x) for x in s.get("STEPS", (30000,))), "max_iter": int(s.get("MAX_ITER", 40000)),
        "warmup_factor": float(s.get("WARMUP_FACTOR", 0.001)), "warmup_iters": int(s.get("WARMUP_ITERS", 1000)),
        "ims_per_batch": int(s.get("IMS_PER_BATCH", 16)), "checkpoint_period": int(s.get("CHECKPOINT_PERIOD", 5000)),
        "rpn_batch": int((m.get("RPN", {}) or {}).get("BATCH_SIZE_PER_IMAGE", 256)), "rpn_pos": float((m.get("RPN", {}) or {}).get("POSITIVE_FRACTION", 0.5)),
        "roi_batch": int((m.get("ROI_HEADS", {}) or {}).get("BATCH_SIZE_PER_IMAGE", 512)), "roi_pos": float((m.get("ROI_HEADS", {}) or {}).get("POSITIVE_FRACTION", 0.25)),
        "flip": str((cfg.get("INPUT", {}) or {}).get("RANDOM_FLIP", "horizontal")),
        "eval_period": int((cfg.get("TEST", {}) or {}).get("EVAL_PERIOD", 0)),
        "min_size_train": tuple(int(x) for x in ((cfg.get("INPUT", {}) or {}).get("MIN_SIZE_TRAIN") or ())),
        "min_size_sampling": str((cfg.get("INPUT", {}) or {}).get("MIN_SIZE_TRAIN_SAMPLING", "choice")),
        # a file that is silent on MAX_SIZE_TRAIN trains at its test maximum (both default to 1333 in detectron2)
        "max_size_train": int((cfg.get("INPUT", {}) or {}).get("MAX_SIZE_TRAIN", (cfg.get("INPUT", {}) or {}).get("MAX_SIZE_TEST", 1333))),
        "max_size_test": int((cfg.get("INPUT", {}) or {}).get("MAX_SIZE_TEST", 1333)),
        "crop": bool(((cfg.get("INPUT", {}) or {}).get("CROP", {}) or {}).get("ENABLED", False)),
        "pre_nms_topk_train": int((m.get("RPN", {}) or {}).get("PRE_NMS_TOPK_TRAIN", 2000)),
        "post_nms_topk_train": int((m.get("RPN", {}) or {}).get("POST_NMS_TOPK_TRAIN", 1000)),
        "warmup_method": str(s.get("WARMUP_METHOD", "linear")), "bias_lr_factor": float(s.get("BIAS_LR_FACTOR", 1.0)),
        "weight_decay_bias": s.get("WEIGHT_DECAY_BIAS", None), "nesterov": bool(s.get("NESTEROV", False)),
        "clip_gradients": bool((s.get("CLIP_GRADIENTS", {}) or {}).get("ENABLED", False)),
        "amp": bool((s.get("AMP", {}) or {}).get("ENABLED", False)),
        "lr_scheduler": str(s.get("LR_SCHEDULER_NAME", "WarmupMultiStepLR")),
        "sampler_train": str((cfg.get("DATALOADER", {}) or {}).get("SAMPLER_TRAIN", "TrainingSampler")),
    }


# capacities of the training engine (csrc/train_engine.inc): ground-truth boxes per image, mask-head entries per image,
# RoI slots per image, RPN candidates per level
GT_CAP, MASK_ENTRIES_CAP, ROI_CAP, RPN_PRE_TOPK_CAP, RPN_POST_TOPK_CAP = 128, 256, 1024, 2048, 1024


def validate_solver(sv: Dict[str, Any]) -> None:
    """Reject, up front, every solver / sampler value the training engine does not implement -- never clamp or ignore one
    silently (the same rule as ``EngineSpec.check_supported`` for the model keys).  Values of the reference YAML
    (R:config/detectron2_config_3bands.yaml:268-305, :29, :178, :192, :248-250) all pass."""
    errs = []
    if sv["lr_scheduler"] != "WarmupMultiStepLR":
        errs.append(f"SOLVER.LR_SCHEDULER_NAME {sv['lr_scheduler']!r} (only WarmupMultiStepLR)")
    if sv["warmup_method"] != "linear":
        errs.append(f"SOLVER.WARMUP_METHOD {sv['warmup_method']!r} (only linear)")
    if sv["bias_lr_factor"] != 1.0:
        errs.append(f"SOLVER.BIAS_LR_FACTOR {sv['bias_lr_factor']} (only 1.0: one SGD launch over the flat parameter buffer)")
    if sv["weight_decay_bias"] not in (None, "None") and float(sv["weight_decay_bias"]) != sv["weight_decay"]:
        errs.append(f"SOLVER.WEIGHT_DECAY_BIAS {sv['weight_decay_bias']} != WEIGHT_DECAY {sv['weight_decay']}")
    if sv["nesterov"]:
        errs.append("SOLVER.NESTEROV true")
    if sv["clip_gradients"]:
        errs.append("SOLVER.CLIP_GRADIENTS.ENABLED true")
    if sv["crop"]:
        errs.append("INPUT.CROP.ENABLED true")
    if sv["sampler_train"] != "TrainingSampler":
        errs.append(f"DATALOADER.SAMPLER_TRAIN {sv['sampler_train']!r} (only TrainingSampler)")
    if sv["max_size_train"] != sv["max_size_test"]:
        errs.append(f"INPUT.MAX_SIZE_TRAIN {sv['max_size_train']} != MAX_SIZE_TEST {sv['max_size_test']} (one maximum size per engine geometry)")
    if sv["roi_batch"] > ROI_CAP:
        errs.append(f"ROI_HEADS.BATCH_SIZE_PER_IMAGE {sv['roi_batch']} > {ROI_CAP} RoI slots per image")
    if sv["roi_batch"] * sv["roi_pos"] > MASK_ENTRIES_CAP:
        errs.append(f"ROI_HEADS.BATCH_SIZE_PER_IMAGE * POSITIVE_FRACTION = {sv['roi_batch'] * sv['roi_pos']:g} foreground RoIs per image > "
                    f"{MASK_ENTRIES_CAP} mask-head entries per image (the mask loss would silently train on a subset)")
    if not 1 <= sv["pre_nms_topk_train"] <= RPN_PRE_TOPK_CAP:
        errs.append(f"RPN.PRE_NMS_TOPK_TRAIN {sv['pre_nms_topk_train']} outside [1, {RPN_PRE_TOPK_CAP}]")
    if not 1 <= sv["post_nms_topk_train"] <= RPN_POST_TOPK_CAP:
        errs.append(f"RPN.POST_NMS_TOPK_TRAIN {sv['post_nms_topk_train']} outside [1, {RPN_POST_TOPK_CAP}]")
    if sv["min_size_sampling"] not in ("choice", "range"):
        errs.append(f"INPUT.MIN_SIZE_TRAIN_SAMPLING {sv['min_size_sampling']!r}")
    if errs:
        raise SystemExit("unsupported training configuration:\n  " + "\n  ".join(errs))


def check_gt_capacity(recs: Sequence[Dict[str, Any]], what: str) -> None:
    """More ground-truth boxes in one image than the engine's per-image capacity would abort the run mid-training: find it
    before the first step."""
    worst = max(recs, key=lambda r: len(r["classes"]), default=None)
    if worst is not None and len(worst["classes"]) > GT_CAP:
        raise SystemExit(f"{what}: {worst['file_name']} has {len(worst['classes'])} annotations; the training engine holds at most "
                         f"{GT_CAP} ground-truth boxes per image")


def lr_at(sv: Dict[str, Any], it: int) -> float:
    """WarmupMultiStepLR ([EXT d2: solver/lr_scheduler.py]): linear warm-up factor times gamma^(milestones passed)."""
    if it < sv["warmup_iters"]:
        alpha = it / sv["warmup_iters"]
        wf = sv["warmup_factor"] * (1 - alpha) + alpha
    else:
        wf = 1.0
    return sv["base_lr"] * wf * sv["gamma"] ** sum(1 for s in sv["steps"] if s <= it)


def load_coco_training_set(path: str) -> Tuple[List[Dict[str, Any]], List[int]]:
    """COCO JSON -> per-image records {file_name, width, height, boxes (k,4) XYXY, classes (k,), polygons [k][...]} with
    contiguous 0-based classes (sorted category ids), crowd annotations dropped and empty images filtered."""
    with open(path) as f:
        d = json.load(f)
    cats = sorted(c["id"] for c in d.get("categories", []))
    cid = {c: i for i, c in enumerate(cats)}
    per: Dict[int, List[dict]] = {}
    for a in d.get("annotations", []):
        if a.get("iscrowd", 0):
            continue
        per.setdefault(a["image_id"], []).append(a)
    recs = []
    for im in d.get("images", []):
        anns = per.get(im["id"], [])
        if not anns:
            continue                                        # FILTER_EMPTY_ANNOTATIONS
        boxes, classes, polys = [], [], []
        for a in anns:
            seg = a.get("segmentation")
            if not isinstance(seg, list):
                raise SystemExit("RLE segmentations are not supported (INPUT.MASK_FORMAT is polygon)")
            p = [np.asarray(s, np.float64) for s in seg if len(s) >= 6 and len(s) % 2 == 0]
            if not p:
                continue
            x, y, w, h = a["bbox"]
            boxes.append([x, y, x + w, y + h]); classes.append(cid[a["category_id"]]); polys.append(p)
        if boxes:
            recs.append({"file_name": im["file_name"], "width": im["width"], "height": im["height"], "boxes": np.asarray(boxes, np.float64),
                         "classes": np.asarray(classes, np.int64), "polygons": polys})
    return recs, cats


def map_record(rec: Dict[str, Any], tile: np.ndarray, net_hw: Tuple[int, int], flip: bool):
    """DatasetMapper for one image: optional horizontal flip, then the annotations scaled to the network input (the tile
    itself is resized on the GPU by the engine's Pillow-exact preprocess)."""
    h, w = tile.shape[:2]
    boxes = rec["boxes"].copy()
    polys = [[p.copy() for p in inst] for inst in rec["polygons"]]
    if flip:
        tile = tile[:, ::-1]
        boxes[:, [0, 2]] = w - boxes[:, [2, 0]]
        for inst in polys:
            for p in inst:
                p[0::2] = w - p[0::2]
    sy, sx = net_hw[0] / h, net_hw[1] / w
    boxes[:, [0, 2]] *= sx
    boxes[:, [1, 3]] *= sy
    boxes[:, [0, 2]] = np.clip(boxes[:, [0, 2]], 0, net_hw[1])
    boxes[:, [1, 3]] = np.clip(boxes[:, [1, 3]], 0, net_hw[0])
    for inst in polys:
        for p in inst:
            p[0::2] *= sx
            p[1::2] *= sy
    keep = ((boxes[:, 2] - boxes[:, 0]) > 1e-5) & ((boxes[:, 3] - boxes[:, 1]) > 1e-5)       # filter_empty_instances
    idx = np.nonzero(keep)[0]
    return np.ascontiguousarray(tile), boxes[idx].astype(np.float32), rec["classes"][idx], [polys[i] for i in idx]


def training_sampler(n: int, seed: int, rank: int, world: int):
    """TrainingSampler: an infinite stream of shuffled indices, every rank taking each world-th one."""
    rng = np.random.default_rng(seed)
    k = 0
    while True:
        for i in rng.permutation(n).tolist():
            if k % world == rank:
                yield i
            k += 1


def main(argv: Optional[Sequence[str]] = None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("config_file", help="YAML with a 'train_model.py' section (R:config/config_obj_detec.yaml)")
    ap.add_argument("--synthetic-weights", action="store_true", help="start from seeded synthetic weights (no checkpoint available offline)")
    ap.add_argument("--max-iter", type=int, default=0, help="override SOLVER.MAX_ITER (smoke runs)")
    ap.add_argument("--precision", choices=["auto", "fp32", "fp16"], default="auto",
                    help="arithmetic of the training step.  auto (default) follows the detectron2 YAML as the reference does: SOLVER.AMP.ENABLED "
                         "false / absent (R:config/detectron2_config_3bands.yaml has no AMP key) = fp32 activations, gradients and operands on the "
                         "fp32 matrix cores; true = fp16 operands with fp32 master weights and dynamic loss scaling (~9x faster per step)")
    ap.add_argument("--loss-scale", type=float, default=1024.0, help="initial fp16 loss scale (halved when a step overflows); fp32 runs use 1")
    ap.add_argument("--scale-window", type=int, default=2000, help="clean steps after which the loss scale doubles (0 = never)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--log-period", type=int, default=20)
    ap.add_argument("--tagged-samples", type=int, default=3, help="tagged ground-truth PNGs per dataset in sample_tagged_img_subfolder (0 = none)")
    ap.add_argument("--val-max-images", type=int, default=0, help="cap on the validation images per evaluation (0 = all)")
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(levelname)s %(message)s", stream=sys.stderr)
    with open(args.config_file) as f:
        cfg = yaml.safe_load(f)[SECTION]
    rank, world, local_rank = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("RS_DIST_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        import datetime
        # rank 0 merges the sharded validation results and runs the (pure-Python) COCO evaluation while the others wait at
        # the barrier behind it: give the collectives more room than the default watchdog's 10 minutes
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(minutes=60))
    if world > 1:
        import torch
        ndev = max(1, torch.cuda.device_count())            # counting devices does not initialise the GPU
        local_rank = local_rank % ndev                      # several ranks may share a card (tests on a one-GPU box, gloo)
    os.chdir(cfg["working_directory"])
    log_dir = cfg.get("log_subfolder", "logs")
    os.makedirs(log_dir, exist_ok=True)
    recs, cats = load_coco_training_set(cfg["COCO_files"]["trn"])
    if not recs:
        raise SystemExit("no annotated training images")
    sub = cfg.get("sample_tagged_img_subfolder")
    if sub and rank == 0 and args.tagged_samples > 0:
        # the reference saves a few images of every registered dataset with their ground truth drawn in
        # (R:config/config_obj_detec.yaml:65, detectron2's Visualizer.draw_dataset_dict there; plain PIL here)
        from .make_detections import read_tile as _read
        from .tagging import draw_annotations
        os.makedirs(sub, exist_ok=True)
        pick = np.random.default_rng(args.seed)
        for dataset, path in cfg["COCO_files"].items():
            if not os.path.exists(path):
                continue
            drecs, dcats = (recs, cats) if dataset == "trn" else load_coco_training_set(path)
            with open(path) as f:
                names = [c["name"] for c in sorted(json.load(f).get("categories", []), key=lambda c: c["id"])]
            for i in sorted(pick.choice(len(drecs), size=min(args.tagged_samples, len(drecs)), replace=False).tolist()) if drecs else []:
                r = drecs[i]
                anns = [{"bbox": [b[0], b[1], b[2] - b[0], b[3] - b[1]], "category_id": int(k), "segmentation": [q.tolist() for q in ps]}
                        for b, k, ps in zip(r["boxes"].tolist(), r["classes"].tolist(), r["polygons"])]
                rgb = _read(r["file_name"])[:, :, ::-1][:, :, :3]
                png = os.path.join(sub, f"{dataset}_tagged_{os.path.splitext(os.path.basename(r['file_name']))[0]}.png")
                draw_annotations(rgb, anns, {k: k for k in range(len(dcats))}, names or None).save(png)
        log.info("tagged sample training images -> %s/", sub)
    sv = load_solver(cfg["detectron2_config_file"])
    validate_solver(sv)
    check_gt_capacity(recs, "training set")
    max_iter = args.max_iter or sv["max_iter"]
    mw = cfg.get("model_weights", {}) or {}
    if args.synthetic_weights:
        spec = load_d2_yaml(cfg["detectron2_config_file"], num_classes=max(len(cats), 1))
        W = synthetic_weights(spec, seed=0)
    elif mw.get("pth_file"):
        W = load_checkpoint(mw["pth_file"])
        spec = load_d2_yaml(cfg["detectron2_config_file"], num_classes=infer_num_classes(W))
    elif mw.get("model_zoo_checkpoint_url"):
        # the reference's unchanged YAML (R:config/config_obj_detec.yaml:71-72): no download here, but a machine that has run the
        # reference (or whose iopath cache was filled by hand) already holds the file -- weights.resolve_zoo_checkpoint
        from .weights import adapt_num_classes, resolve_zoo_checkpoint, zoo_cache_roots
        zoo = str(mw["model_zoo_checkpoint_url"])
        path = resolve_zoo_checkpoint(zoo)
        if path is None:
            raise SystemExit(f"model_weights.model_zoo_checkpoint_url {zoo!r}: no network access and no cached copy under "
                             f"{[os.path.join(r, 'detectron2') for r in zoo_cache_roots()]} (detectron2's own cache layout, $FVCORE_CACHE "
                             "to relocate); give model_weights.pth_file or --synthetic-weights")
        W = load_checkpoint(path)
        k = max(len(cats), 1)
        W, redone = adapt_num_classes(W, k, seed=args.seed)
        log.info("initial weights: %s (model-zoo cache)%s", path,
                 f"; re-initialised for {k} classes (shape mismatch, as DetectionCheckpointer skips them): {redone}" if redone else "")
        spec = load_d2_yaml(cfg["detectron2_config_file"], num_classes=k)
    else:
        raise SystemExit("model_weights needs pth_file or model_zoo_checkpoint_url (or --synthetic-weights)")

    precision = args.precision if args.precision != "auto" else ("fp16" if sv["amp"] else "fp32")
    spec = spec.replace(precision=precision)
    if precision == "fp32":
        args.loss_scale, args.scale_window = 1.0, 0     # nothing to protect from underflow; the overflow check stays on (it also catches a diverged run)
    log.info("precision: %s (%s)", precision, "--precision" if args.precision != "auto" else f"SOLVER.AMP.ENABLED {sv['amp']}")
    from .engine import MultiScaleTrainer      # fails loudly without librs_engine.so / a HIP device
    from .make_detections import read_tile
    first = read_tile(recs[0]["file_name"])
    per_rank = max(1, sv["ims_per_batch"] // world)
    # INPUT.MIN_SIZE_TRAIN (R:31-38): "choice" draws one of the listed sizes, "range" any size in [lo, hi] (here: the listed
    # end points and every 32nd size between them, one engine geometry each); no list = the test size
    sizes = list(sv["min_size_train"]) or [spec.min_size_test]
    if sv["min_size_sampling"] == "range" and len(sizes) == 2:
        sizes = list(range(sizes[0], sizes[1] + 1, 32))
    ms = MultiScaleTrainer(spec, W, first.shape, sizes, batch=per_rank, device=local_rank, loss_scale=args.loss_scale)
    ms.set_sampling(sv["rpn_batch"], sv["rpn_pos"], sv["roi_batch"], sv["roi_pos"])
    ms.set_rpn_topk(sv["pre_nms_topk_train"], sv["post_nms_topk_train"])
    size_rng = np.random.default_rng(args.seed * 104729 + rank)
    trainer = ms.select(sizes[-1])
    log.info("training: %d images, %d classes, batch %d x %d ranks, %d iterations, shortest-edge sizes %s, %.1f M trainable values",
             len(recs), len(cats), per_rank, world, max_iter, sizes, trainer.param_count / 1e6)
    sampler = training_sampler(len(recs), args.seed, rank, world)
    flips = np.random.default_rng(args.seed * 7919 + rank)
    metrics = open(os.path.join(log_dir, "metrics.json"), "a") if rank == 0 else None

    def save(name: str, it: int) -> None:
        if rank != 0:
            return
        import torch
        state = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in ms.current.export_weights(W).items()}
        torch.save({"model": state, "iteration": it}, os.path.join(log_dir, name))
        with open(os.path.join(log_dir, "last_checkpoint"), "w") as f:
            f.write(name)

    # validation loss every TEST.EVAL_PERIOD iterations (R:322): the five training losses on the `val` set with the training
    # forward at the test size, no flip, no update -- what the object-detector's loss-evaluation hook logs as "validation_loss"
    val_recs: List[Dict[str, Any]] = []
    if sv["eval_period"] > 0 and cfg["COCO_files"].get("val") and os.path.exists(cfg["COCO_files"]["val"]):
        val_recs, _ = load_coco_training_set(cfg["COCO_files"]["val"])
        if args.val_max_images:
            val_recs = val_recs[: args.val_max_images]
        check_gt_capacity(val_recs, "validation set")

    def validation_loss(it: int) -> Optional[float]:
        if not val_recs:
            return None
        vt = ms.select(sizes[-1])
        hw = ms.net_shape(sizes[-1])
        mine = val_recs[rank::world]
        tot, cnt = 0.0, 0
        for k in range(0, len(mine), per_rank):
            chunk = mine[k:k + per_rank]
            mapped = [map_record(r, read_tile(r["file_name"]), hw, False) for r in chunk]
            l = vt.train_step(np.stack([m[0] for m in mapped]), [m[1] for m in mapped], [m[2] for m in mapped], [m[3] for m in mapped],
                              seed=777 + k)                      # gradients are discarded: the next training step overwrites them
            tot += sum(l.values()) * len(chunk)
            cnt += len(chunk)
        if world > 1:
            import torch
            import torch.distributed as dist
            t = torch.tensor([tot, float(cnt)], dtype=torch.float64)
            if dist.get_backend() == "nccl":
                t = t.cuda()
            dist.all_reduce(t)
            tot, cnt = float(t[0]), int(t[1])
        return tot / max(cnt, 1)

    def validation_ap() -> Dict[str, float]:
        """COCOEvaluator on the val set (bbox + segm AP; coco_eval.py) with the CURRENT weights.  As detectron2's
        ``inference_on_dataset`` + ``COCOEvaluator`` do, the images are sharded over the ranks (rank r takes val_recs[r::world]):
        every rank runs inference through its trainer's own forward engine at the test size and matches detections with
        ground truth per image (mask IoUs included); only the per-image match records are gathered on rank 0, which
        accumulates them in image order.  Every rank calls this; a barrier follows, so no rank runs ahead into the next
        iteration's all-reduce while rank 0 is still evaluating."""
        if not val_recs:
            return {}
        from .coco_eval import accumulate, match_images
        from .train_targets import rasterize_polygons_within_box
        vt = ms.select(sizes[-1])
        eng = vt.inference_engine()
        mine = list(range(rank, len(val_recs), world))
        gts, dts = [], []
        for k in range(0, len(mine), per_rank):
            chunk = [val_recs[i] for i in mine[k:k + per_rank]]
            tiles = np.stack([read_tile(r["file_name"]) for r in chunk])
            for r, inst in zip(chunk, eng.infer(tiles)):
                h, w = tiles.shape[1:3]
                g = {"boxes": r["boxes"], "classes": r["classes"]}
                d = {"boxes": inst.pred_boxes, "classes": inst.pred_classes, "scores": inst.scores}
                if spec.mask_on and h == w:
                    g["masks"] = np.stack([rasterize_polygons_within_box(p, np.array([0.0, 0.0, w, h]), h) for p in r["polygons"]])
                    d["masks"] = inst.pred_masks
                gts.append(g); dts.append(d)
        segm = spec.mask_on and all("masks" in g for g in gts)
        local = {"idx": mine, "bbox": match_images(gts, dts, spec.num_classes, "bbox", spec.detections_per_image),
                 "segm": match_images(gts, dts, spec.num_classes, "segm", spec.detections_per_image) if segm else None}
        parts = [local]
        if world > 1:
            import torch.distributed as dist
            gathered = [None] * world if rank == 0 else None
            dist.gather_object(local, gathered, dst=0)
            parts = gathered if rank == 0 else []
            dist.barrier()
        if rank != 0:
            return {}
        out: Dict[str, float] = {}
        for kind in ("bbox", "segm"):
            if any(p[kind] is None for p in parts):
                continue
            by_image = sorted(((i, r) for p in parts for i, r in zip(p["idx"], p[kind])), key=lambda t: t[0])
            out.update({f"{kind}/{k}": v for k, v in accumulate([r for _, r in by_image], spec.num_classes).items()})
        return {k: (None if v != v else v) for k, v in out.items()}           # NaN -> null in metrics.json

    t0 = time.time()
    last_trainer, skipped_steps, clean_steps, scale_just_cut = None, 0, 0, False
    def load_batch(_it: int):
        """One batch of the DatasetMapper (size draw, sampler, decode, flip, scale): runs on a loader thread one iteration ahead
        of the GPU step.  Only this function touches the sampler and the two RNGs, and batches are requested in order, so the
        stream of batches is the same as without the thread."""
        tiles, boxes, classes, polys, drawn = [], [], [], [], []
        for _ in range(per_rank):
            # ResizeShortestEdge draws per record ([EXT d2: data/transforms/augmentation_impl.py]); ImageList.from_tensors then pads
            # the batch to its largest image -- MultiScaleTrainer.select_batch
            size = int(sizes[int(size_rng.integers(len(sizes)))])
            rec = recs[next(sampler)]
            tile = read_tile(rec["file_name"])
            if tile.shape != first.shape:
                raise SystemExit(f"{rec['file_name']}: tile shape {tile.shape} != {first.shape} (one shape per run)")
            t, b, c, p = map_record(rec, tile, ms.net_shape(size), sv["flip"] == "horizontal" and flips.random() < 0.5)
            tiles.append(t); boxes.append(b); classes.append(c); polys.append(p); drawn.append(size)
        return drawn, np.stack(tiles), boxes, classes, polys

    from concurrent.futures import ThreadPoolExecutor
    loader = ThreadPoolExecutor(max_workers=1)
    pending = loader.submit(load_batch, 0)
    for it in range(max_iter):
        drawn, tile_batch, boxes, classes, polys = pending.result()
        if it + 1 < max_iter:
            pending = loader.submit(load_batch, it + 1)      # decoded while this iteration's step runs on the GPU
        trainer = ms.select_batch(drawn)
        losses = trainer.train_step(tile_batch, boxes, classes, polys, seed=args.seed * 1000003 + it * world + rank,
                                    allreduce=world > 1)      # bucketed all-reduce enqueued behind the backward pass
        # dynamic fp16 loss scale (GradScaler's policy): the previous step's overflow flag is read here, after this step's own
        # synchronisation, so it costs no extra stall; a skipped step halves the scale, `--scale-window` clean steps double it.
        # The flag is taken after the all-reduce, so every rank sees the same value and the scales stay in step.
        new_scale = None
        if last_trainer is not None and last_trainer.overflowed():
            skipped_steps += 1
            clean_steps = 0
            if not scale_just_cut:
                # The flag is read one iteration late: this iteration's gradient was computed with the scale that overflowed,
                # and will most likely overflow (and be skipped on the device) too.  Halve ONCE per overflow episode: the flag
                # of the step right after a cut still belongs to the old scale and is not counted again.
                new_scale = max(ms.loss_scale / 2.0, 1.0)
                log.warning("iteration %d: gradient overflow, step skipped; loss scale %g -> %g", it - 1, ms.loss_scale, new_scale)
            scale_just_cut = new_scale is not None
        else:
            scale_just_cut = False
            clean_steps += 1
            if args.scale_window > 0 and clean_steps >= args.scale_window and ms.loss_scale < 65536.0:
                clean_steps = 0
                new_scale = ms.loss_scale * 2.0
        lr = lr_at(sv, it)
        trainer.apply_sgd(lr, sv["momentum"], sv["weight_decay"])
        last_trainer = trainer
        if new_scale is not None:
            ms.set_loss_scale(new_scale)          # after the step: the gradient buffer carried the old scale
        bad = not all(np.isfinite(v) for v in losses.values())
        if world > 1:
            # the decision to stop must be collective: a rank that exits alone leaves the others hung in the next all-reduce
            import torch
            import torch.distributed as dist
            flag = torch.tensor([1.0 if bad else 0.0])
            if dist.get_backend() == "nccl":
                flag = flag.cuda()
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            bad_any = bool(flag.item() > 0)
        else:
            bad_any = bad
        if bad_any:
            if bad:
                log.error("iteration %d: non-finite loss %s on rank %d (lower --loss-scale)", it, losses, rank)
            if world > 1:
                import torch.distributed as dist
                dist.destroy_process_group()
            raise SystemExit(f"iteration {it}: non-finite loss on {'this' if bad else 'another'} rank; every rank stops")
        do_eval = sv["eval_period"] > 0 and ((it + 1) % sv["eval_period"] == 0 or it == max_iter - 1)
        vloss = validation_loss(it) if do_eval else None
        vap = validation_ap() if (do_eval and vloss is not None) else {}
        if rank == 0 and ((it + 1) % args.log_period == 0 or it == max_iter - 1 or vloss is not None):
            rec = {"iteration": it, "total_loss": float(sum(losses.values())), "lr": lr, "time": (time.time() - t0) / (it + 1), **losses,
                   "loss_scale": ms.loss_scale, "skipped_steps": skipped_steps}
            if vloss is not None:
                rec["validation_loss"] = vloss
                rec.update(vap)
            metrics.write(json.dumps(rec) + "\n")
            metrics.flush()
            log.info("iter %d  total_loss %.4f  %s  lr %.6f  %.3f s/iter", it, rec["total_loss"],
                     "  ".join(f"{k} {v:.4f}" for k, v in losses.items()), lr, rec["time"])
        if (it + 1) % sv["checkpoint_period"] == 0 and it + 1 < max_iter:
            save(f"model_{it:07d}.pth", it)
    save("model_final.pth", max_iter - 1)
    loader.shutdown(wait=True)
    ms.close()
    if metrics:
        metrics.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
