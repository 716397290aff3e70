import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from proj_roadsurf_amd.engine import Engine
from proj_roadsurf_amd.spec import EngineSpec
from proj_roadsurf_amd.weights import synthetic_weights
from proj_roadsurf_amd.synthetic import synthetic_tiles
from tests.util import match_detections
import oracle.maskrcnn_oracle as O
for std in (1.0, 57.375, 1024.0, 1.0/64):
    spec = EngineSpec(num_classes=2, min_size_test=320, max_size_test=533, rpn_pre_nms_topk_test=300, rpn_post_nms_topk_test=300, pixel_std=(std, std, std), precision="split")
    W = synthetic_weights(spec, seed=0)
    tiles = synthetic_tiles(2, 256, 256, 3, seed=77)
    ref = O.OracleModel(spec, W)([tiles[0], tiles[1]], keep=True)
    for prec in ("split", "fp32"):
        eng = Engine(spec.replace(precision=prec), W, (256, 256, 3), max_batch=2)
        try:
            dets = eng.infer(tiles)
            rels = {}
            for name in ["res2", "res5", "p2", "p6"]:
                got = torch.from_numpy(eng.tensor(name, n=2).astype(np.float32)).permute(0, 3, 1, 2)
                want = torch.stack([ref[i]["inter"]["feats"][name] for i in range(2)])
                rels[name] = float((got - want).norm() / want.norm())
            out = []
            for i in range(2):
                r = {"boxes": ref[i]["boxes"].numpy(), "scores": ref[i]["scores"].numpy(), "classes": ref[i]["classes"].numpy(), "masks": ref[i]["masks"].numpy()}
                g = {"boxes": dets[i].pred_boxes, "scores": dets[i].scores, "classes": dets[i].pred_classes, "masks": dets[i].pred_masks}
                fw = match_detections(r, g, min_score=0.05, iou_thr=0.99)
                out.append((fw["n_ref"], fw["n_matched"], float(fw["max_dscore"]), float(fw["max_dbox"])))
            amax = float(np.abs(eng.tensor("res2", n=2)).max()), float(np.abs(eng.tensor("net_input", n=2)).max())
        finally:
            eng.close()
        print(f"std {std:8.4f} {prec:5s} maps rel L2 {{{', '.join(f'{k} {v:.1e}' for k, v in rels.items())}}} dets {out} |res2|max {amax[0]:.3g} |input|max {amax[1]:.3g}", flush=True)
