"""Latency of one tile per call (batch 1, tile resident in HBM, one engine): eager launches against hipGraph replay (RS_USE_GRAPH=1).  usage: single_tile_latency.py [precision] [reps] [tiles per call]"""
import sys
import time

sys.path.insert(0, ".")
from proj_roadsurf_amd.engine import Engine                # noqa: E402
from proj_roadsurf_amd.spec import EngineSpec              # noqa: E402
from proj_roadsurf_amd.synthetic import synthetic_tiles    # noqa: E402
from proj_roadsurf_amd.weights import synthetic_weights    # noqa: E402


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    nb = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    spec = EngineSpec(num_classes=2, precision=prec)
    e = Engine(spec, synthetic_weights(spec, seed=0), (512, 512, 3), max_batch=nb)
    ptr = e.upload_tiles(synthetic_tiles(nb, 512, 512, 3, seed=3))
    for _ in range(10):
        e.infer_device(ptr, nb)
        e.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        e.infer_device(ptr, nb)
        e.sync()
    dt = (time.perf_counter() - t0) / reps
    print(f"{prec}: {dt * 1e3:.3f} ms per call of {nb} tile(s), one call at a time", flush=True)
    e.close()


if __name__ == "__main__":
    main()
