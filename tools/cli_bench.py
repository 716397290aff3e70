#!/usr/bin/env python3
"""End-to-end rate of the drop-in CLI (tile files -> GeoPackage) on synthetic data: N 512x512x3 TIFF tiles on disk,
seeded synthetic weights, the reference's own YAML (R:config/detectron2_config_3bands.yaml defaults via EngineSpec).
Measures what a user of make_detections.py sees: decode + H2D + forward + D2H + vectorise + write.

    python tools/cli_bench.py [--tiles 256] [--batch 16] [--host-workers 4] [--vector-threads 4]
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=256)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--host-workers", type=int, default=4)
    ap.add_argument("--vector-threads", type=int, default=4)
    ap.add_argument("--decode-procs", type=int, default=4)
    ap.add_argument("--weights", choices=["random", "trained"], default="random",
                    help="random: synthetic_weights (speckle masks, ~1900 polygons per tile); trained: synthetic.train_trained_like on "
                         "synthetic scenes (a handful of clean objects per tile, like a trained detector on real tiles)")
    ap.add_argument("--train-steps", type=int, default=300)
    ap.add_argument("--precision", choices=["fp16", "split", "fp32"], default="fp16", help="make_detections --precision")
    ap.add_argument("--lanes", type=int, default=2)
    args = ap.parse_args()
    import yaml
    from PIL import Image
    from proj_roadsurf_amd import make_detections
    from proj_roadsurf_amd.synthetic import synthetic_tiles

    with tempfile.TemporaryDirectory() as td:
        wd = os.path.join(td, "obj_detector")
        os.makedirs(os.path.join(wd, "oth-images"))
        base = synthetic_tiles(16, 512, 512, 3, seed=1234)
        extra = ["--synthetic-weights"]
        if args.weights == "trained":
            import numpy as np
            import torch
            from proj_roadsurf_amd.spec import EngineSpec
            from proj_roadsurf_amd.synthetic import synthetic_scenes, train_trained_like
            W, _ = train_trained_like(EngineSpec(num_classes=2), 512, steps=args.train_steps)
            os.makedirs(os.path.join(wd, "logs"), exist_ok=True)
            torch.save({"model": {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in W.items()}, "iteration": args.train_steps},
                       os.path.join(wd, "logs", "model_0005999.pth"))
            base = synthetic_scenes(64, 512, 512, 3, seed=555)[0]
            extra = []
        images, meta = [], {}
        for i in range(args.tiles):
            fn = f"oth-images/18_{1000 + i}_2000.tif"
            Image.fromarray(base[i % len(base)][:, :, ::-1]).save(os.path.join(wd, fn))
            images.append({"id": i, "file_name": fn, "width": 512, "height": 512})
            meta[fn] = {"extent": [100.0 * i, 0.0, 100.0 * i + 104.6, 104.6], "crs": "EPSG:3857"}
        cats = [{"id": 1, "name": "artificial"}, {"id": 2, "name": "natural"}]
        json.dump({"images": images, "annotations": [], "categories": cats}, open(os.path.join(wd, "COCO_oth.json"), "w"))
        json.dump(meta, open(os.path.join(wd, "img_metadata.json"), "w"))
        yaml.safe_dump({"INPUT": {"FORMAT": "RGB"}}, open(os.path.join(td, "d2.yaml"), "w"))     # everything else: reference defaults
        cfg = {"make_detections.py": {"working_directory": wd, "log_subfolder": "logs", "image_metadata_json": "img_metadata.json",
                                      "COCO_files": {"oth": "COCO_oth.json"}, "detectron2_config_file": os.path.join(td, "d2.yaml"),
                                      "model_weights": {"pth_file": "logs/model_0005999.pth"},
                                      "rdp_simplification": {"enabled": True, "epsilon": 0.75}, "score_lower_threshold": 0.05}}
        yaml.safe_dump(cfg, open(os.path.join(td, "config.yaml"), "w"))
        cwd = os.getcwd()
        t0 = time.time()
        rc = make_detections.main([os.path.join(td, "config.yaml"), *extra, "--batch", str(args.batch),
                                   "--host-workers", str(args.host_workers), "--vector-threads", str(args.vector_threads),
                                   "--decode-procs", str(args.decode_procs), "--precision", args.precision, "--lanes", str(args.lanes)])
        dt = time.time() - t0
        os.chdir(cwd)
        size = os.path.getsize(os.path.join(wd, "oth_detections_at_0dot05_threshold.gpkg"))
    print(json.dumps({"cli_tiles": args.tiles, "weights": args.weights, "precision": args.precision, "seconds_total_incl_engine_build": dt, "rc": rc, "gpkg_bytes": size}))


if __name__ == "__main__":
    main()
