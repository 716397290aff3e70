#!/usr/bin/env python3
"""End-to-end rate of the training CLI (TIFF tiles + COCO JSON on disk -> checkpoints) on synthetic data: what a user of
train_model.py sees per iteration, data loading, mask-target rasterisation, logging and the periodic checkpoint included.

    python tools/train_cli_bench.py [--tiles 64] [--batch 8] [--iters 60]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=64)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=60)
    args = ap.parse_args()
    import yaml
    from PIL import Image
    from proj_roadsurf_amd import train_model
    from proj_roadsurf_amd.synthetic import synthetic_tiles

    with tempfile.TemporaryDirectory() as td:
        wd = os.path.join(td, "obj_detector")
        os.makedirs(os.path.join(wd, "trn-images"))
        base = synthetic_tiles(16, 512, 512, 3, seed=99)
        rng = np.random.default_rng(5)
        images, anns = [], []
        for i in range(args.tiles):
            fn = f"trn-images/18_{2000 + i}_3000.tif"
            Image.fromarray(base[i % 16][:, :, ::-1]).save(os.path.join(wd, fn))
            images.append({"id": i, "file_name": fn, "width": 512, "height": 512})
            for k in range(int(rng.integers(3, 9))):
                x, y = rng.uniform(10, 380, 2)
                w, h = rng.uniform(30, 120, 2)
                anns.append({"id": len(anns), "image_id": i, "category_id": 1 + k % 2, "bbox": [x, y, w, h], "iscrowd": 0, "area": w * h,
                             "segmentation": [[x, y, x + w, y, x + w, y + h, x, y + h]]})
        json.dump({"images": images, "annotations": anns, "categories": [{"id": 1, "name": "artificial"}, {"id": 2, "name": "natural"}]},
                  open(os.path.join(wd, "COCO_trn.json"), "w"))
        d2 = {"INPUT": {"FORMAT": "RGB"}, "SOLVER": {"IMS_PER_BATCH": args.batch, "MAX_ITER": args.iters, "CHECKPOINT_PERIOD": 1000000},
              "TEST": {"EVAL_PERIOD": 0}}
        yaml.safe_dump(d2, open(os.path.join(td, "d2.yaml"), "w"))             # everything else: the reference's defaults
        cfg = {"train_model.py": {"working_directory": wd, "log_subfolder": "logs", "COCO_files": {"trn": "COCO_trn.json"},
                                  "detectron2_config_file": os.path.join(td, "d2.yaml"), "model_weights": {}}}
        yaml.safe_dump(cfg, open(os.path.join(td, "config.yaml"), "w"))
        cwd = os.getcwd()
        t0 = time.time()
        rc = train_model.main([os.path.join(td, "config.yaml"), "--synthetic-weights", "--log-period", "20"])
        dt = time.time() - t0
        os.chdir(cwd)
        lines = [json.loads(l) for l in open(os.path.join(wd, "logs", "metrics.json"))]
    print(json.dumps({"rc": rc, "tiles_on_disk": args.tiles, "batch": args.batch, "iters": args.iters, "seconds_total_incl_engine_build": dt,
                      "s_per_iter_logged": lines[-1]["time"], "images_per_s": args.batch / lines[-1]["time"],
                      "final_total_loss": lines[-1]["total_loss"], "loss_scale": lines[-1].get("loss_scale"), "skipped_steps": lines[-1].get("skipped_steps")}))


if __name__ == "__main__":
    main()
