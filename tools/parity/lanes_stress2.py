"""Follow-up of lanes_stress.py: for every tile result whose packed masks differ from the solo engine's, decide on the host which side is wrong (numpy restatement of
paste_masks_kernel on the fetched probabilities and boxes) and whether the wrong words equal what the lane's buffer held before this forward (= words not rewritten)."""
import os
import sys

sys.path.insert(0, ".")
import numpy as np      # noqa: E402

from proj_roadsurf_amd.engine import Engine, LanePipeline     # noqa: E402
from proj_roadsurf_amd.spec import EngineSpec                 # noqa: E402
from proj_roadsurf_amd.synthetic import synthetic_tiles       # noqa: E402
from proj_roadsurf_amd.weights import synthetic_weights       # noqa: E402


def paste_row_words(probs, box, T, thr=0.5):
    """packed (T, T/8) mask of one detection, float32 arithmetic in the kernel's order"""
    S = probs.shape[0]
    x0, y0, x1, y1 = [np.float32(v) for v in box]
    out = np.zeros((T, T), np.uint8)
    ys = np.arange(T, dtype=np.float32)
    gy = (ys + np.float32(0.5) - y0) / (y1 - y0) * np.float32(2) - np.float32(1)
    iy = ((gy + np.float32(1)) * np.float32(S) - np.float32(1)) / np.float32(2)
    fy = np.floor(iy)
    gx = (ys + np.float32(0.5) - x0) / (x1 - x0) * np.float32(2) - np.float32(1)
    ix = ((gx + np.float32(1)) * np.float32(S) - np.float32(1)) / np.float32(2)
    fx = np.floor(ix)
    for y in range(T):
        iy0 = int(fy[y]); iy1 = iy0 + 1
        if iy1 < 0 or iy0 >= S:
            continue
        wy1 = np.float32(iy[y] - fy[y]); wy0 = np.float32(1) - wy1
        for x in range(T):
            ix0 = int(fx[x]); ix1 = ix0 + 1
            if ix1 < 0 or ix0 >= S:
                continue
            wx1 = np.float32(ix[x] - fx[x]); wx0 = np.float32(1) - wx1
            v = np.float32(0)
            if iy0 >= 0 and ix0 >= 0: v = np.float32(v + probs[iy0, ix0] * np.float32(wx0 * wy0))
            if iy0 >= 0 and ix1 < S: v = np.float32(v + probs[iy0, ix1] * np.float32(wx1 * wy0))
            if iy1 < S and ix0 >= 0: v = np.float32(v + probs[iy1, ix0] * np.float32(wx0 * wy1))
            if iy1 < S and ix1 < S: v = np.float32(v + probs[iy1, ix1] * np.float32(wx1 * wy1))
            out[y, x] = v >= np.float32(thr)
    return out


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    T, B, L = 256, 3, 2
    spec = EngineSpec(num_classes=2, precision=prec)
    W = synthetic_weights(spec, seed=0)
    batches = [synthetic_tiles(B, T, T, 3, seed=700 + k) for k in range(6)]
    solo = Engine(spec, W, (T, T, 3), max_batch=4)
    want = []
    for b in batches:
        solo.infer_device(solo.upload_tiles(b), B)
        want.append(solo.fetch(B, want_probs=True))
    pipe = LanePipeline(spec, W, (T, T, 3), max_batch=4, lanes=L, shared_stream=os.environ.get("STRESS_SHARED", "0") == "1")
    prev_dev = [None] * L
    found = 0
    for r in range(rounds):
        for i in range(2 * L):
            e = pipe.engines[i % L]
            bi = (r + i) % len(batches)
            prev_dev[i % L] = e.tensor("masks", n=B).copy()       # what the lane's buffer holds before this forward
            e.infer_device(e.upload_tiles(batches[bi]), B)
            if i % L == L - 1:                                     # both lanes loaded: now collect both
                for j in range(L):
                    ee = pipe.engines[j]
                    bj = (r + i - (L - 1) + j) % len(batches)
                    got = ee.fetch(B, want_probs=True)
                    for t, (a, b) in enumerate(zip(want[bj], got)):
                        if np.array_equal(a._packed, b._packed):
                            continue
                        found += 1
                        x = np.unpackbits(a._packed ^ b._packed, axis=-1)
                        for d in sorted(set(int(q[0]) for q in np.argwhere(x))):
                            ref = paste_row_words(b.mask_probs[d].reshape(28, 28), b.pred_boxes[d], T)
                            sa, sb = np.unpackbits(a._packed[d], axis=-1), np.unpackbits(b._packed[d], axis=-1)
                            rows = sorted(set(int(q[0]) for q in np.argwhere(sa != sb)))
                            bef = np.unpackbits(prev_dev[j][t][d], axis=-1) if prev_dev[j] is not None else None
                            wrong = sb if not np.array_equal(sb, ref) else sa
                            who = "lane" if not np.array_equal(sb, ref) else ("solo" if not np.array_equal(sa, ref) else "neither?")
                            stale = bef is not None and all(np.array_equal(wrong[y], bef[y]) for y in rows)
                            if not stale and bef is not None:
                                # word by word: does every wrong 32-bit word equal the previous content's word?
                                w_wrong, w_ref, w_bef = (np.packbits(v, axis=-1).view(np.uint32) for v in (wrong, ref, bef))
                                bw = np.argwhere(w_wrong != w_ref)
                                stale = f"{sum(int(w_wrong[y, x] == w_bef[y, x]) for y, x in bw)} of {len(bw)} wrong words equal the previous content's word"

                            print(f"round {r} lane {j} batch {bj} tile {t} det {d}: rows {rows[:6]}..{rows[-1]} ({len(rows)} rows), host paste says the wrong side is {who}; "
                                  f"its wrong rows equal the buffer's previous content: {stale}; box {np.round(b.pred_boxes[d], 1)}", flush=True)
    print(f"{prec}: {found} differing tile results in {rounds * 2 * L * B}", flush=True)
    pipe.close(); solo.close()


if __name__ == "__main__":
    main()
