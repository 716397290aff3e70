"""Two (or more) engines on independent streams, fed alternately without waiting: every batch's detections against the same engine run alone.
Reports per field how many batches differ and by how much.  usage: lanes_stress.py [precision] [lanes] [rounds] [tile]"""
import sys

sys.path.insert(0, ".")
import numpy as np      # noqa: E402

from proj_roadsurf_amd.engine import Engine, LanePipeline     # noqa: E402
from proj_roadsurf_amd.spec import EngineSpec                 # noqa: E402
from proj_roadsurf_amd.synthetic import synthetic_tiles       # noqa: E402
from proj_roadsurf_amd.weights import synthetic_weights       # noqa: E402


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    T = int(sys.argv[4]) if len(sys.argv) > 4 else 256
    B = 3
    spec = EngineSpec(num_classes=2, precision=prec)
    W = synthetic_weights(spec, seed=0)
    batches = [synthetic_tiles(B, T, T, 3, seed=700 + k) for k in range(6)]
    solo = Engine(spec, W, (T, T, 3), max_batch=4)
    want = []
    for b in batches:
        solo.infer_device(solo.upload_tiles(b), B)
        want.append(solo.fetch(B, want_probs=True))
    import os
    pipe = LanePipeline(spec, W, (T, T, 3), max_batch=4, lanes=L, shared_stream=os.environ.get('STRESS_SHARED', '0') == '1')
    bad = {"boxes": 0, "scores": 0, "classes": 0, "masks": 0, "count": 0}
    worst = 0
    n = 0
    for r in range(rounds):
        order = [(r + i) % len(batches) for i in range(2 * L)]
        lanes = []
        for i, bi in enumerate(order):
            e = pipe.engines[i % L]
            if i >= L:                                   # the lane's previous batch first
                got = e.fetch(B, want_probs=True)
                lanes.append((order[i - L], got))
                if any(not np.array_equal(x._packed, y._packed) for x, y in zip(want[order[i - L]], got)):
                    again = e.fetch(B, want_probs=True)
                    same_as_first = all(np.array_equal(x._packed, y._packed) for x, y in zip(got, again))
                    same_as_want = all(np.array_equal(x._packed, y._packed) for x, y in zip(want[order[i - L]], again))
                    dev = e.tensor("masks", n=B)
                    dev_eq_want = all(np.array_equal(dev[t][: len(want[order[i - L]][t])], want[order[i - L]][t]._packed) for t in range(B))
                    import time
                    time.sleep(0.2)
                    dev2 = e.tensor("masks", n=B)
                    dev2_eq_want = all(np.array_equal(dev2[t][: len(want[order[i - L]][t])], want[order[i - L]][t]._packed) for t in range(B))
                    print(f"  refetch: equals first fetch {same_as_first}, equals solo {same_as_want}; device buffer via rs_engine_tensor equals solo {dev_eq_want}; "
                          f"0.2 s later {dev2_eq_want}", flush=True)
            e.infer_device(e.upload_tiles(batches[bi]), B)
        for i in range(L):
            e = pipe.engines[(2 * L - L + i) % L]
            lanes.append((order[L + i], e.fetch(B, want_probs=True)))
        for bi, got in lanes:
            for a, b in zip(want[bi], got):
                n += 1
                if len(a) != len(b):
                    bad["count"] += 1
                    continue
                bad["boxes"] += not np.array_equal(a.pred_boxes, b.pred_boxes)
                bad["scores"] += not np.array_equal(a.scores, b.scores)
                bad["classes"] += not np.array_equal(a.pred_classes, b.pred_classes)
                if not np.array_equal(a._packed, b._packed):
                    bad["masks"] += 1
                    d = np.unpackbits(a._packed ^ b._packed).sum()
                    worst = max(worst, int(d))
                pa, pb = getattr(a, "mask_probs", None), getattr(b, "mask_probs", None)
                if pa is None or pb is None:
                    bad["no_probs"] = bad.get("no_probs", 0) + 1
                if not np.array_equal(a._packed, b._packed) and bad["masks"] <= 3:
                    x = np.unpackbits(a._packed ^ b._packed, axis=-1)
                    idx = np.argwhere(x)
                    dets = sorted(set(int(i[0]) for i in idx))
                    xa = np.unpackbits(a._packed, axis=-1)
                    print("  bits set only in solo:", int((x & xa).sum()), " only in lane:", int((x & (1 - xa)).sum()), " lane", "?", "batch", bi, flush=True)
                    print("  mask diff: detections", dets[:10], "rows", int(idx[:, 1].min()), "-", int(idx[:, 1].max()), "cols", int(idx[:, 2].min()), "-", int(idx[:, 2].max()),
                          "probs equal" if (pa is not None and np.array_equal(pa, pb)) else "probs differ", "boxes", a.pred_boxes[dets[0]], flush=True)
                if pa is not None and pb is not None and not np.array_equal(pa, pb):
                    bad["probs"] = bad.get("probs", 0) + 1
                    bad["probs_maxdiff"] = max(bad.get("probs_maxdiff", 0.0), float(np.abs(pa.astype(np.float64) - pb.astype(np.float64)).max()))
    for li, e in enumerate(pipe.engines):
        if "paste_dbg" in e.tensor_names():
            print(f"  lane {li} paste_dbg [last generation whose successor ran, threads that ran after their successor] =", e.tensor("paste_dbg")[:2].tolist(), flush=True)
    if "paste_dbg" in solo.tensor_names():
        print("  solo paste_dbg =", solo.tensor("paste_dbg")[:2].tolist(), flush=True)
    print(f"{prec} lanes {L} tile {T}: {n} tile results, differing: {bad}, most differing mask bits in one tile {worst}", flush=True)
    pipe.close()
    solo.close()


if __name__ == "__main__":
    main()
