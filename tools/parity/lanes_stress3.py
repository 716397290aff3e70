"""Which of lane B's kernels have to run beside lane A's forward for A's packed masks to pick up stale 64-byte pieces?  Lane A runs whole forwards (each result
against the solo engine); lane B repeats ONE phase (0 backbone + FPN + RPN, 1 box head, 2 mask head) of a forward it ran once.  usage: lanes_stress3.py phase [rounds] [precision]"""
import sys

sys.path.insert(0, ".")
import numpy as np      # noqa: E402

from proj_roadsurf_amd.engine import Engine           # noqa: E402
from proj_roadsurf_amd.spec import EngineSpec         # noqa: E402
from proj_roadsurf_amd.synthetic import synthetic_tiles   # noqa: E402
from proj_roadsurf_amd.weights import synthetic_weights   # noqa: E402


def main():
    phase = int(sys.argv[1])
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    prec = sys.argv[3] if len(sys.argv) > 3 else "fp16"
    only = sys.argv[4] if len(sys.argv) > 4 else ""
    T, B = 256, 3
    spec = EngineSpec(num_classes=2, precision=prec)
    W = synthetic_weights(spec, seed=0)
    batches = [synthetic_tiles(B, T, T, 3, seed=700 + k) for k in range(6)]
    a = Engine(spec, W, (T, T, 3), max_batch=4)
    b = Engine(spec, W, (T, T, 3), max_batch=4)
    want = [a.infer(x) for x in batches]
    pb = b.upload_tiles(batches[0])
    b.infer_device(pb, B)
    b.sync()
    bad = 0
    for r in range(rounds):
        bi = r % len(batches)
        pa = a.upload_tiles(batches[bi])
        reps = {-1: 1, 0: 2, 1: 12, 2: 6}[phase]
        if only:
            for _ in range(int(sys.argv[5]) if len(sys.argv) > 5 else 12):
                rc = b.lib.rs_debug_run_stages_matching(b._h, only.encode(), B)
                assert rc == 0
        elif phase >= 0:
            for _ in range(reps):                      # about a forward's worth of lane-B work, enqueued ahead
                b.infer_phase(pb, B, phase)
        else:
            b.infer_device(pb, B)
        a.infer_device(pa, B)
        got = a.fetch(B)
        b.sync()
        for x, y in zip(want[bi], got):
            assert np.array_equal(x.pred_boxes, y.pred_boxes) and np.array_equal(x.scores, y.scores)
            bad += not np.array_equal(x._packed, y._packed)
    print(f"lane B repeats {('stages *' + only + '*') if only else ('phase ' + str(phase))} ({prec}): {bad} of {rounds * B} tile results of lane A with differing masks", flush=True)
    a.close(); b.close()


if __name__ == "__main__":
    main()
