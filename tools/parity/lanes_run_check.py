"""The streaming host interface at production size: LanePipeline.run over many batches of 16 tiles of 512 x 512, every tile result against one engine run alone.
usage: lanes_run_check.py [precision] [batches]"""
import sys

sys.path.insert(0, ".")
import numpy as np      # noqa: E402

from proj_roadsurf_amd.engine import Engine, LanePipeline     # noqa: E402
from proj_roadsurf_amd.spec import EngineSpec                 # noqa: E402
from proj_roadsurf_amd.synthetic import synthetic_tiles       # noqa: E402
from proj_roadsurf_amd.weights import synthetic_weights       # noqa: E402


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "split"
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    spec = EngineSpec(num_classes=2, precision=prec)
    W = synthetic_weights(spec, seed=0)
    pool = [synthetic_tiles(16, 512, 512, 3, seed=4000 + k) for k in range(5)]
    solo = Engine(spec, W, (512, 512, 3), max_batch=16)
    want = [solo.infer(b) for b in pool]
    solo.close()
    pipe = LanePipeline(spec, W, (512, 512, 3), max_batch=16, lanes=2)
    bad = 0
    n = 0
    for k, res in enumerate(pipe.run(pool[i % len(pool)] for i in range(nb))):
        for a, b in zip(want[k % len(pool)], res):
            n += 1
            ok = (len(a) == len(b) and np.array_equal(a.pred_boxes, b.pred_boxes) and np.array_equal(a.scores, b.scores) and np.array_equal(a.pred_classes, b.pred_classes)
                  and np.array_equal(a.pred_masks, b.pred_masks))
            bad += not ok
    pipe.close()
    print(f"{prec}: {n} tile results through LanePipeline.run (two independent lanes, batch 16 of 512 x 512), {bad} differ from the single engine", flush=True)


if __name__ == "__main__":
    main()
