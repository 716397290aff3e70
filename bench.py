#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): tiles/sec of the Mask R-CNN R50-FPN forward on synthetic
512x512 3-band tiles, batch 16 per GPU (BASELINE configs[1]).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU; tiles shard across ranks with no data-path collective (each rank runs its
own batch -> "weak" scaling); torch.distributed (RCCL) is used only for the barrier and the
max-over-ranks of the timed region.  A step = one pass of the whole hot path (resize+normalise,
backbone, FPN, RPN, box head, NMS, mask head, mask paste) over one batch of 16 tiles that is
already resident in HBM; results stay in HBM (PCIe-inclusive rate: DESIGN.md).

Prints ONE JSON line on rank 0, with `roofline` (dominant kernel = the conv kernel symbol with the largest share
of the step, HIP-event timed on the engine's stream during the timed steps) and `cpu_baseline` (the
oracle = CPU restatement of detectron2's path, on the host cores, bounded sample).  `--gpus N` without a
launcher (WORLD_SIZE unset) starts the N ranks itself before touching the GPU.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0   # dense fp16/bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def cpu_baseline(spec, W, tiles, max_seconds=25.0):
    """Oracle (kind "port": our CPU restatement of detectron2 0.6's path, oneDNN convolutions, batched RoIAlign) on the host
    cores.  Protocol of BASELINE.md section 3: 3 warm-up tiles, then >= 10 timed tiles at batch 1 (what DefaultPredictor does;
    this is `value`), then one timed batch of 16 (`value_batch16`).  Bounded: the batch-1 leg stops after `max_seconds`."""
    import torch
    from oracle.maskrcnn_oracle import OracleModel

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, int(os.environ.get("RS_CPU_BASELINE_THREADS", "16"))))   # a 1-GPU box's CPU share is 16
    torch.set_num_threads(cores)
    m = OracleModel(spec, W)
    print(f"[bench] cpu_baseline: oracle on {cores} threads ...", file=sys.stderr, flush=True)
    n_warm = 3
    for i in range(n_warm):
        m([tiles[i % len(tiles)]])
    t0 = time.time()
    n = 0
    while n < 10 or (time.time() - t0) < 0.0:
        m([tiles[(n_warm + n) % len(tiles)]])
        n += 1
        if time.time() - t0 > max_seconds:
            break
    dt = time.time() - t0
    print(f"[bench] cpu_baseline: batch 1: {n} tiles in {dt:.1f} s", file=sys.stderr, flush=True)
    out = {"value": n / dt, "unit": "tiles/s", "cores": cores, "kind": "port",
           "sample": f"{n_warm} warm-up + {n} timed synthetic {tiles[0].shape[0]}x{tiles[0].shape[1]}x{tiles[0].shape[2]} tile(s), "
                     f"batch 1 as DefaultPredictor does, torch CPU fp32 (oneDNN), {dt:.1f} s"}
    if len(tiles) >= 16 and n / dt >= 0.4:            # one batch of 16 would otherwise take > 40 s: keep the default run short
        t1 = time.time()
        m([tiles[i] for i in range(16)])
        d16 = time.time() - t1
        out["value_batch16"] = 16 / d16
        out["sample"] += f"; then one batch of 16 in {d16:.1f} s"
        print(f"[bench] cpu_baseline: batch 16: {d16:.1f} s", file=sys.stderr, flush=True)
    return out


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, RCCL rendezvous on
    127.0.0.1) BEFORE this process touches the GPU, pass rank 0's JSON line through and exit with the launcher's code."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--tile", type=int, default=512)
    ap.add_argument("--bands", type=int, default=3, help="3 = BASELINE configs[1] (headline); 4 = RGB+NIR tiles (configs[3], with --tile 1024 --batch 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stages", action="store_true", help="print the per-stage table to stderr")
    ap.add_argument("--lanes", type=int, default=2,
                    help="independent engine contexts (own stream + activation buffers); consecutive batches alternate lanes so "
                         "one batch's latency-bound detection glue overlaps the next batch's convolutions")
    ap.add_argument("--profile-mode", type=int, default=3,
                    help="HIP-event stage timing during the timed steps: 3 = every 4th step (default), 2 = every step, 0 = off")
    ap.add_argument("--no-fp32-mode", action="store_true", help="skip the reference-precision (fp32 MFMA) measurement")
    ap.add_argument("--sustain-seconds", type=float, default=5.0,
                    help="after the K timed steps, keep stepping for this long and report `sustained_tiles_per_s` (clocks settle "
                         "after a few seconds of load); 0 = skip (profiler runs)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world)

    from proj_roadsurf_amd.engine import LanePipeline
    from proj_roadsurf_amd.spec import EngineSpec
    from proj_roadsurf_amd.weights import synthetic_weights
    from proj_roadsurf_amd.synthetic import synthetic_tiles

    spec = EngineSpec(num_classes=2)      # R:config/detectron2_config_3bands.yaml defaults, 2 classes (artificial/natural)
    if args.bands == 4:                   # no 4-band YAML exists in the reference: PIXEL_MEAN/STD extended by a NIR entry
        spec = spec.replace(pixel_mean=spec.pixel_mean + (110.0,), pixel_std=spec.pixel_std + (1.0,))
    W = synthetic_weights(spec, seed=0)
    B, T = args.batch, args.tile
    # rank r owns tiles r*B .. r*B+B-1 of the synthetic tileset (seed = 1234 + tile id)
    C_in = args.bands
    tiles = synthetic_tiles(B, T, T, C_in, seed=1234 + rank * B)
    L = max(1, args.lanes)
    pipe = LanePipeline(spec, W, (T, T, C_in), max_batch=B, device=local_rank, lanes=L)
    engs = pipe.engines
    ptrs = [e.upload_tiles(tiles) for e in engs]
    eng = engs[0]

    def barrier():
        if world > 1:
            dist.barrier()

    for k in range(args.warmup):
        pipe.submit(ptrs[k % L], B)
    pipe.sync()
    for e in engs:
        e.set_profiling(args.profile_mode)   # HIP events around the launches, on each lane's stream, no host wait
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):              # step k = one batch of B tiles, on lane k mod L
        pipe.submit(ptrs[pipe.k % L], B)
    pipe.sync()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    # Sustained rate: the K-step region above is what the contract times (`value`); a 10k-tile job runs for seconds, by which
    # time the chip has settled at its clock under load.  Same loop, no stage events, >= --sustain-seconds, max over ranks.
    stages = eng.stage_times()               # the stage events of the timed region (set_profiling(0) would clear them)
    for e in engs[1:]:                       # same stage list on every lane: pool the HIP-event totals
        for a, b in zip(stages, e.stage_times()):
            a["ms_total"] += b["ms_total"]
            a["calls"] += b["calls"]
    sustained = None
    if args.sustain_seconds > 0:
        for e in engs:
            e.set_profiling(0)
        barrier()
        torch.cuda.synchronize()
        ts = time.perf_counter()
        n_sus = 0
        while True:
            for _ in range(32):
                pipe.submit(ptrs[pipe.k % L], B)
            n_sus += 32
            pipe.sync()
            stop = torch.tensor([1.0 if time.perf_counter() - ts >= args.sustain_seconds else 0.0], device="cuda")
            if world > 1:
                dist.all_reduce(stop, op=dist.ReduceOp.MAX)    # every rank leaves the loop after the same number of steps
            if float(stop.item()) > 0:
                break
        torch.cuda.synchronize()
        barrier()
        dts = time.perf_counter() - ts
        if world > 1:
            tm = torch.tensor([dts], dtype=torch.float64, device="cuda")
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            dts = float(tm.item())
        sustained = {"tiles_per_s": world * B * n_sus / dts, "steps": n_sus, "seconds": dts}
    for e in engs:
        e.set_profiling(0)
    # PCIe-inclusive rate of the streaming host interface (pinned H2D of the tiles + forward + D2H of boxes/scores/packed
    # masks + host-side collection into Instances, LanePipeline.run); reported beside the headline, never as `value`
    nb = 8
    for _ in pipe.run(tiles for _ in range(2)):
        pass
    t1 = time.perf_counter()
    for res in pipe.run(tiles for _ in range(nb)):
        pass
    pcie_tiles_per_s = nb * B / (time.perf_counter() - t1)
    eng.infer_device(ptrs[0], B)
    dets = eng.fetch(B)
    nprop = eng.tensor("proposal_count", n=B)
    ndet = [len(d) for d in dets]

    # Reference-precision mode (every conv / linear layer in fp32 on v_mfma_f32_16x16x4_f32, csrc/ref_f32.hip): the like-for-like
    # number against the reference's fp32 arithmetic; rank 0 only, a few steps (a step is ~16x the fp16 step's matrix work).
    fp32_mode = None
    if rank == 0 and not args.no_fp32_mode:
        from proj_roadsurf_amd.engine import Engine
        pipe.sync()
        e32 = Engine(spec.replace(precision="fp32"), W, (T, T, C_in), max_batch=B, device=local_rank)
        try:
            p32 = e32.upload_tiles(tiles)
            e32.infer_device(p32, B)
            e32.sync()
            e32.set_profiling(2)
            n32 = 3
            torch.cuda.synchronize()
            t32 = time.perf_counter()
            for _ in range(n32):
                e32.infer_device(p32, B)
            e32.sync()
            d32 = (time.perf_counter() - t32) / n32
            st32 = [s for s in e32.stage_times() if s["flops"] > 0 and s["calls"] > 0]
            fl32 = sum(s["flops"] for s in st32)
            ms32 = sum(s["ms_total"] / s["calls"] for s in st32)
            fp32_mode = {"tiles_per_s": B / d32, "ms_per_step": d32 * 1e3, "whole_path_tflops": fl32 / d32 / 1e12,
                         "matrix_stages_tflops": fl32 / (ms32 * 1e-3) / 1e12 if ms32 else None, "peak_tflops": 157.3,
                         "frac_of_fp32_matrix_peak": (fl32 / (ms32 * 1e-3) / 1e12 / 157.3) if ms32 else None,
                         "kernel": "conv_f32_mfma_kernel (v_mfma_f32_16x16x4_f32, exact fp32)", "steps": n32}
        finally:
            e32.close()
    if rank == 0:
        value = world * B * args.steps / dt
        # dominant kernel: conv_igemm 128x128 variant = every conv stage with Cout % 128 == 0
        conv = [s for s in stages if s["flops"] > 0 and s["calls"] > 0]
        by_time = sorted(stages, key=lambda s: -s["ms_total"])
        tot_ms = sum(s["ms_total"] for s in stages)
        # group the GEMM stages by the kernel symbol (tile variant) they launched; the dominant kernel is the
        # one with the largest share of the step
        groups = {}
        for s in conv:
            g = groups.setdefault(s["kernel"] or "?", {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0})
            g["ms"] += s["ms_total"]
            g["flops"] += s["flops"] * s["calls"]
            g["bytes"] += s["bytes"] * s["calls"]
            g["launches"] += s["calls"]
        if not groups:                           # --profile-mode 0: no stage timing, no roofline
            groups = {"(stage timing off)": {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0}}
        dom = max(groups, key=lambda k: groups[k]["ms"])
        G = groups[dom]
        achieved = G["flops"] / (G["ms"] * 1e-3) / 1e12 if G["ms"] > 0 else 0.0
        total_flops_step = sum(s["flops"] for s in conv)
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")     # written by tools/pmc_summary.py from two --pmc passes
        if os.path.exists(pmc) and B == 16 and T == 512:
            sym = dom.split(" ")[0]                                   # "conv_deep_kernel" or "conv_igemm_kernel<2,4,4,8>"
            want = sym.replace("conv_igemm_kernel<", "").replace(">", "").replace(",", ", ")
            for k in json.load(open(pmc)):
                if sym.startswith("conv_igemm"):
                    hit = f"conv_igemm_kernel<{want}, false, true" in k["kernel"]
                elif sym == "conv_deep_kernel":                         # the 256-pixel inference instantiation: conv_deep_kernel<0, false, 8> (or <0, false> before the tile height became a parameter)
                    hit = "conv_deep_kernel<0, false, 8>" in k["kernel"] or "conv_deep_kernel<0, false>" in k["kernel"]
                else:
                    hit = sym in k["kernel"]
                if hit:
                    traffic = k["hbm_bytes_per_launch_corrected"]
                    traffic_src = "profiles/pmc_latest.json: rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE passes of this command, (2*FETCH+WRITE)*1024 per launch"
                    break
        roofline = {"bound": "mfma", "kernel": dom + " (fp16 MFMA 16x16x32, fp32 accumulate)",
                    "achieved": achieved, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_PEAK_TFLOPS,
                    "traffic": traffic, "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch_avg": G["bytes"] / max(G["launches"], 1), "launches": G["launches"],
                    "avg_launch_ms": G["ms"] / max(G["launches"], 1), "flops_per_launch_avg": G["flops"] / max(G["launches"], 1),
                    "share_of_step_time": G["ms"] / tot_ms if tot_ms else None,
                    "whole_path_tflops": total_flops_step * args.steps * world / dt / 1e12,
                    "other_kernels": {k: {"tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] else 0.0,
                                          "share_of_step_time": v["ms"] / tot_ms if tot_ms else None, "launches": v["launches"]}
                                      for k, v in groups.items() if k != dom}}
        if args.stages:
            print(f"{'stage':28s} {'ms/call':>9s} {'TFLOP/s':>9s} {'GB/s':>9s}", file=sys.stderr)
            for s in stages:
                if s["calls"]:
                    mc = s["ms_total"] / s["calls"]
                    print(f"{s['name']:28s} {mc:9.4f} {s['flops'] / mc / 1e9 if mc else 0:9.1f} {s['bytes'] / mc / 1e6 if mc else 0:9.1f}", file=sys.stderr)
            print(f"sum of stage times {tot_ms / max(stages[0]['calls'], 1):.3f} ms/batch; wall {dt / args.steps * 1e3:.3f} ms/step", file=sys.stderr)
        out = {
            "metric": f"tiles_per_sec_{T}x{T}x{C_in}", "value": value, "unit": "tiles/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"Mask R-CNN R50-FPN inference, batch {B} of {T}x{T} {C_in}-band tiles per GPU "
                                   f"(BASELINE configs[{1 if C_in == 3 else 3}]), 800x800 network input",
                       "batch_per_gpu": B, "lanes": L, "tile": [T, T, C_in], "num_classes": 2,
                       "proposals_per_tile": float(np.mean(nprop)), "detections_per_tile": float(np.mean(ndet)),
                       "sharding": "tiles across ranks, no data-path collective"},
            "roofline": roofline,
            "value_is": f"the {args.steps} timed steps after {args.warmup} warm-up steps (driver contract); sustained_tiles_per_s = the same loop "
                        "run for >= --sustain-seconds right after it",
            "sustained_tiles_per_s": sustained["tiles_per_s"] if sustained else None,
            "sustained": sustained,
            "rccl_world_size": (dist.get_world_size() if world > 1 else 1),
            "fp32_mode": fp32_mode,
            "fp32_mode_tiles_per_s": fp32_mode["tiles_per_s"] if fp32_mode else None,
            "pcie_inclusive_tiles_per_s": pcie_tiles_per_s,
            "top_stages": [{"name": s["name"], "ms_per_step": s["ms_total"] / max(s["calls"], 1)} for s in by_time[:6]],
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(spec, W, tiles)
        print(json.dumps(out))
    pipe.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
