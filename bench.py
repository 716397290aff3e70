#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): tiles/sec of the Mask R-CNN R50-FPN forward on synthetic
512x512 3-band tiles, batch 16 per GPU (BASELINE configs[1]).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --train                      # training step of BASELINE configs[4] as its own JSON line

One process per GPU; tiles shard across ranks with no data-path collective (each rank runs its
own batch -> "weak" scaling); torch.distributed (RCCL) is used only for the barrier and the
max-over-ranks of the timed region.  A step = one pass of the whole hot path (resize+normalise,
backbone, FPN, RPN, box head, NMS, mask head, mask paste) over one batch of 16 tiles that is
already resident in HBM; results stay in HBM (PCIe-inclusive rate: `pcie_inclusive_tiles_per_s`).

`value` is measured in the SPLIT-OPERAND precision mode (`--precision split`, the default): the reference computes in fp32
(R:config/detectron2_config_3bands.yaml:268-305 has no SOLVER.AMP key) and this mode reproduces fp32 results on the fp16 matrix
cores -- every GEMM operand as hi + lo fp16 planes (22 significand bits), three MFMA products per real product, fp32 accumulate
(csrc/common.h ConvParams::split).  It meets the stated tolerance on the benched (saturated, chaotic) workload; the plain fp16 mode,
about 2.7x faster, meets it on a trained detector only and is reported beside it (`fp16_mode`).

Prints ONE JSON line on rank 0.  Beside the contract's keys it carries
  * `roofline`            dominant kernel (the conv kernel symbol with the largest share of the step), HIP-event timed on the
                          engine's stream during the timed steps;
  * `reference_precision` the SAME K-step region run by the reference-precision engine (every matrix stage in exact fp32 on
                          v_mfma_f32_16x16x4_f32): the like-for-like figure against the reference's fp32 arithmetic;
  * `parity`              detections of the headline engine matched against the reference-precision engine's on the benched batch
                          (GPU vs GPU: no oracle in any timed path), SURVEY 8d matching;
  * `fp16_mode`           the same K-step region with plain fp16 operands (the round 1-3 headline) and ITS parity on the benched batch;
  * `trained_like`        the same two measurements on a detector trained here for a few hundred steps (scores separate, a handful
                          of detections per tile -- what a deployed model looks like; the headline workload is the saturated
                          random-weight worst case);
  * `training`            one data-parallel training step (BASELINE configs[4]) at batch 8 and at 1 image per GPU;
  * `cpu_baseline`        the oracle (CPU restatement of detectron2's path) on the host cores, bounded sample.
`--gpus N` without a launcher (WORLD_SIZE unset) starts the N ranks itself before touching the GPU.
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
F32_MFMA_PEAK_TFLOPS = 157.3
HBM_PEAK_GBS = 8000.0


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


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
    log(f"cpu_baseline: oracle on {cores} threads ...")
    n_warm = 3
    for i in range(n_warm):
        m([tiles[i % len(tiles)]])
    t0 = time.time()
    n = 0
    while n < 10:
        m([tiles[(n_warm + n) % len(tiles)]])
        n += 1
        if time.time() - t0 > max_seconds:
            break
    dt = time.time() - t0
    log(f"cpu_baseline: batch 1: {n} tiles in {dt:.1f} s")
    out = {"value": n / dt, "unit": "tiles/s", "cores": cores, "kind": "port",
           "sample": f"{n_warm} warm-up + {n} timed synthetic {tiles[0].shape[0]}x{tiles[0].shape[1]}x{tiles[0].shape[2]} tile(s), "
                     f"batch 1 as DefaultPredictor does, torch CPU fp32 (oneDNN), {dt:.1f} s"}
    if len(tiles) >= 16 and n / dt >= 0.4:            # one batch of 16 would otherwise take > 40 s: keep the default run short
        t1 = time.time()
        m([tiles[i] for i in range(16)])
        d16 = time.time() - t1
        out["value_batch16"] = 16 / d16
        out["sample"] += f"; then one batch of 16 in {d16:.1f} s"
        log(f"cpu_baseline: batch 16: {d16:.1f} s")
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


class stdout_to_stderr:
    """RCCL prints a version banner on STDOUT when its first communicator comes up; the contract is ONE JSON line there.  While this
    is active, file descriptor 1 points to stderr (library prints included)."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def as_sets(dets):
    return [{"boxes": d.pred_boxes, "scores": d.scores, "classes": d.pred_classes, "masks": d.pred_masks} for d in dets]


def parity_object(got, ref, what):
    """SURVEY 8d matching (same class, box IoU >= 0.95, |dscore| <= 0.02, mask IoU >= 0.95; reference score >= 0.1) of two detection lists of the same tiles, pooled over
    the tiles, both directions, with the 95 % Wilson lower bound of each matched fraction."""
    from proj_roadsurf_amd.matching import match_detections, wilson_lower
    tot = {"fw_n": 0, "fw_m": 0, "bw_n": 0, "bw_m": 0, "fw_b": 0, "bw_b": 0}
    dscore, miou = 0.0, 1.0
    agg = []
    for g, r in zip(as_sets(got), as_sets(ref)):
        fw, bw = match_detections(r, g), match_detections(g, r)
        tot["fw_n"] += fw["n_ref"]; tot["fw_m"] += fw["n_full"]; tot["fw_b"] += fw["n_matched"]
        tot["bw_n"] += bw["n_ref"]; tot["bw_m"] += bw["n_full"]; tot["bw_b"] += bw["n_matched"]
        dscore = max(dscore, fw["max_dscore"])
        miou = min(miou, float(fw["min_mask_iou"]))
        agg.append(float(fw["agg_mask_iou"]))
    return {"of": what, "criterion": "a detection with score >= 0.1 is matched when a detection of the same class on the other side has box IoU >= 0.95, |dscore| <= 0.02 AND mask IoU >= 0.95 (SURVEY 8d); stated tolerance: >= 0.98 matched both ways",
            "matched_fw": tot["fw_m"] / max(tot["fw_n"], 1), "matched_bw": tot["bw_m"] / max(tot["bw_n"], 1),
            # class + box IoU alone: on the random-weight workload the mask logits sit at 0 +- noise, so a pair of the same box rarely has
            # mask IoU >= 0.95 -- there the gap between these two and the ones above is the masks', not the boxes' 
            "matched_class_and_box_fw": tot["fw_b"] / max(tot["fw_n"], 1), "matched_class_and_box_bw": tot["bw_b"] / max(tot["bw_n"], 1),
            "n_fw": tot["fw_n"], "n_bw": tot["bw_n"],
            "wilson95_lower_fw": wilson_lower(tot["fw_m"], tot["fw_n"]), "wilson95_lower_bw": wilson_lower(tot["bw_m"], tot["bw_n"]),
            "max_dscore": dscore, "agg_mask_iou": float(np.mean(agg)) if agg else 1.0, "min_mask_iou": miou}


def timed_steps(pipe, ptrs, B, steps, warmup, barrier, sync):
    """W untimed + exactly K timed submissions, bracketed by barrier + device sync on both sides; returns seconds of the K steps."""
    L = len(pipe.engines)
    for _ in range(warmup):
        pipe.submit(ptrs[pipe.k % L], B)
    pipe.sync()
    barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        pipe.submit(ptrs[pipe.k % L], B)
    pipe.sync()
    sync()
    barrier()
    return time.perf_counter() - t0


def reference_precision_leg(spec, W, tiles, B, steps, warmup, device):
    """The K-step region on the reference-precision engine (rs_spec.precision = 1: fp32 activations and weights, every conv / linear
    layer on v_mfma_f32_16x16x4_f32, csrc/ref_f32.hip), same tiles resident in HBM.  Returns (json object, detections)."""
    import torch
    from proj_roadsurf_amd.engine import Engine
    T, C_in = tiles.shape[1], tiles.shape[3]
    e32 = Engine(spec.replace(precision="fp32"), W, (T, T, C_in), max_batch=B, device=device)
    try:
        p32 = e32.upload_tiles(tiles)
        for _ in range(max(1, warmup)):
            e32.infer_device(p32, B)
        e32.sync()
        e32.set_profiling(2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            e32.infer_device(p32, B)
        e32.sync()
        torch.cuda.synchronize()
        d32 = (time.perf_counter() - t0) / steps
        st32 = [s for s in e32.stage_times() if s["flops"] > 0 and s["calls"] > 0]
        fl32 = sum(s["flops"] for s in st32)
        ms32 = sum(s["ms_total"] / s["calls"] for s in st32)
        e32.set_profiling(0)
        dets = e32.fetch(B)
        # PCIe-inclusive: pinned H2D of the tiles + forward + D2H of every result field (the reference's per-tile call includes both copies)
        t1 = time.perf_counter()
        for _ in range(2):
            e32.infer(tiles)
        pcie = 2 * B / (time.perf_counter() - t1)
        mt = fl32 / (ms32 * 1e-3) / 1e12 if ms32 else None
        return {"tiles_per_s": B / d32, "ms_per_step": d32 * 1e3, "steps": steps, "warmup": max(1, warmup), "dtype": "f32",
                "whole_path_tflops": fl32 / d32 / 1e12, "matrix_stages_tflops": mt, "peak_tflops": F32_MFMA_PEAK_TFLOPS,
                "frac": (mt / F32_MFMA_PEAK_TFLOPS) if mt else None,
                "frac_is": "matrix stages only (conv / linear launches by HIP events); frac_whole_path divides the whole step's FLOP by the wall time of a step",
                "frac_whole_path": fl32 / d32 / 1e12 / F32_MFMA_PEAK_TFLOPS, "pcie_inclusive_tiles_per_s": pcie,
                "kernel": "conv_f32_mfma_kernel (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulate)"}, dets
    finally:
        e32.close()


def stage_groups_of(tr, spec, B, run4):
    """Per-stage HIP events of four training steps grouped into weight gradients / input gradients / forward GEMMs / RoIAlign / other,
    with the mask head's GEMM FLOP scaled from its 256-entries-per-image capacity to the entries of the last step."""
    tr.set_profiling(True)
    run4()
    st = [x for x in tr.stage_times() if x["calls"]]
    tr.set_profiling(False)
    entries = int(tr.tensor("mask_total")[0]) if spec.mask_on else 0
    mask_fill = entries / float(B * 256)
    groups = {}
    for x in st:
        nm = x["name"]
        if nm.startswith("mask.") or nm.startswith("bwd.mask."):
            x = dict(x, flops=x["flops"] * mask_fill)
        g = ("weight gradients (conv_wgrad kernels, side stream)" if nm.endswith(".w") else
             "input gradients (the forward conv kernels on the transposed weights)" if nm.endswith(".x") else
             "forward GEMMs" if x["flops"] > 0 else "RoIAlign forward / backward" if "roi_align" in nm else "other (losses, sampling, NMS, bias gradients, pooling)")
        a = groups.setdefault(g, {"ms_per_step": 0.0, "flops_per_step": 0.0})
        a["ms_per_step"] += x["ms_total"] / x["calls"]
        a["flops_per_step"] += x["flops"]
    return groups, entries


def training_leg(spec, W, device, steps=12, warmup=3, legs=("b8", "b1", "fp32")):
    """One training step (BASELINE configs[4]: 2-class fine-tune, YAML samplers: 256 anchors / 1024 RoIs per image, 2000/1000 train
    proposals) on 1 GPU: forward + five losses + backward + SGD + refold, host-side mask-target rasterisation included; batch 8
    (the YAML's IMS_PER_BATCH on one GPU) and batch 1 (its per-GPU share on 8 GPUs).  Per-stage HIP events give the share and rate of
    the weight-gradient / input-gradient GEMMs; in a one-rank RCCL group the bucketed all-reduce runs for real (what is exposed of
    it on one GPU is its launch + wait cost, the xGMI time is not)."""
    import torch
    import torch.distributed as dist
    from proj_roadsurf_amd.engine import Trainer
    from proj_roadsurf_amd.synthetic import synthetic_scenes
    T = 512
    out = {"workload": "Mask R-CNN R50-FPN training step, 512x512x3 tiles -> 800x800, FREEZE_AT 2, fp16 operands / fp32 master weights "
                       "(BASELINE configs[4] on one GPU)", "steps": steps, "warmup": warmup}
    own_pg = False
    if not dist.is_initialized():
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        try:
            dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
            own_pg = True
        except Exception as ex:                                  # the figures without a collective are still valid
            log(f"training leg: no RCCL group ({ex})")
    for B in [b for b in (8, 1) if f"b{b}" in legs]:
        tiles, boxes, classes, polys = synthetic_scenes(B, T, T, 3, seed=4321)
        s = 800.0 / T
        nb = [b * np.float32(s) for b in boxes]
        npoly = [[[p * s for p in inst] for inst in img] for img in polys]
        tr = Trainer(spec, W, (T, T, 3), batch=B, device=device, loss_scale=1024.0)
        try:
            def run(n, allreduce):
                for it in range(n):
                    tr.train_step(tiles, nb, classes, npoly, seed=100 + it, allreduce=allreduce)
                    tr.apply_sgd(1e-5, 0.9, 1e-4)
                tr.sync()
                torch.cuda.synchronize()
            run(warmup, False)
            t0 = time.perf_counter(); run(steps, False); dt = (time.perf_counter() - t0) / steps
            rec = {"batch": B, "ms_per_step": dt * 1e3, "images_per_s": B / dt, "trainable_values_M": tr.param_count / 1e6}
            if dist.is_initialized() and dist.get_backend() == "nccl":
                _orig = tr.allreduce_gradients
                tr.allreduce_gradients = lambda force=False: _orig(force=True)      # a one-rank group: run the collectives anyway
                run(2, True)
                t0 = time.perf_counter(); run(steps, True); dta = (time.perf_counter() - t0) / steps
                rec["ms_per_step_with_bucketed_allreduce_1rank"] = dta * 1e3
                rec["exposed_allreduce_ms_1rank"] = (dta - dt) * 1e3
                tr.allreduce_gradients = _orig
            groups, entries = stage_groups_of(tr, spec, B, lambda: run(4, False))
            rec["mask_head_entries_last_step"] = entries
            for g, a in groups.items():
                a["tflops"] = a["flops_per_step"] / (a["ms_per_step"] * 1e-3) / 1e12 if a["ms_per_step"] and a["flops_per_step"] else None
                a["frac_of_mfma_peak"] = a["tflops"] / MFMA_PEAK_TFLOPS if a["tflops"] else None
            rec["stage_groups"] = groups
            rec["note"] = "stage times are HIP events on the stream each stage runs on; the weight-gradient side stream overlaps the chain, so the groups add up to more than ms_per_step"
            tot_fl = sum(a["flops_per_step"] for a in groups.values())
            rec["whole_step_tflops"] = tot_fl / dt / 1e12
            out[f"batch{B}"] = rec
        finally:
            tr.close()
    if "fp32" not in legs or "batch8" not in out:
        if own_pg:
            dist.destroy_process_group()
        return out
    # the reference's arithmetic (fp32, no AMP key in its YAML): the same step on the reference-precision trainer
    tiles, boxes, classes, polys = synthetic_scenes(8, T, T, 3, seed=4321)
    s = 800.0 / T
    nb = [b * np.float32(s) for b in boxes]
    npoly = [[[p * s for p in inst] for inst in img] for img in polys]
    tr = Trainer(spec.replace(precision="fp32"), W, (T, T, 3), batch=8, device=device, loss_scale=1.0)
    try:
        n32 = max(3, steps // 3)
        for it in range(2):
            tr.train_step(tiles, nb, classes, npoly, seed=100 + it); tr.apply_sgd(1e-5, 0.9, 1e-4)
        tr.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(n32):
            tr.train_step(tiles, nb, classes, npoly, seed=200 + it); tr.apply_sgd(1e-5, 0.9, 1e-4)
        tr.sync(); torch.cuda.synchronize()
        d32 = (time.perf_counter() - t0) / n32
        def run4():
            for it in range(4):
                tr.train_step(tiles, nb, classes, npoly, seed=300 + it); tr.apply_sgd(1e-5, 0.9, 1e-4)
            tr.sync()
        g32, _ = stage_groups_of(tr, spec, 8, run4)
        for g, a in g32.items():
            a["tflops"] = a["flops_per_step"] / (a["ms_per_step"] * 1e-3) / 1e12 if a["ms_per_step"] and a["flops_per_step"] else None
            a["frac_of_fp32_matrix_peak"] = a["tflops"] / F32_MFMA_PEAK_TFLOPS if a["tflops"] else None
        fl = out["batch8"]["whole_step_tflops"] * out["batch8"]["ms_per_step"] * 1e-3          # TFLOP of one batch-8 step (same layers)
        out["reference_precision_batch8"] = {"batch": 8, "steps": n32, "ms_per_step": d32 * 1e3, "images_per_s": 8 / d32, "dtype": "f32",
                                             "whole_step_tflops": fl / d32, "frac_of_fp32_matrix_peak": fl / d32 / F32_MFMA_PEAK_TFLOPS,
                                             "kernels": "conv_f32_mfma_kernel (forward, input gradients), conv_wgrad_f32_kernel; v_mfma_f32_16x16x4_f32",
                                             "stage_groups": g32}
    finally:
        tr.close()
    if own_pg:
        dist.destroy_process_group()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--tile", type=int, default=512)
    ap.add_argument("--bands", type=int, default=3, help="3 = BASELINE configs[1] (headline); 4 = RGB+NIR tiles (configs[3], with --tile 1024 --batch 8)")
    ap.add_argument("--weights", choices=["random", "trained"], default="random",
                    help="workload of the headline `value`: random = seeded synthetic weights, 1000 proposals / 100 detections per tile "
                         "(the saturated worst case BASELINE's FLOP count is quoted on); trained = a detector trained here for --train-steps "
                         "steps on synthetic scenes (a handful of detections per tile)")
    ap.add_argument("--precision", choices=["split", "fp16", "fp32"], default="split",
                    help="arithmetic of the headline `value`: split = reference-equivalent (hi + lo fp16 operand planes, three MFMA products, fp32 "
                         "accumulate; meets the stated tolerance on the benched workload), fp16 = plain fp16 operands (2.7x faster; meets it on a "
                         "trained detector, not on the saturated random-weight workload), fp32 = the fp32 matrix cores")
    ap.add_argument("--no-fp16-leg", action="store_true", help="skip the `fp16_mode` object of the default run")
    ap.add_argument("--no-single-tile-leg", action="store_true", help="skip the `single_tile_latency` object (profiler runs: its batch-1 launches would enter the per-kernel averages)")
    ap.add_argument("--train-steps", type=int, default=600)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stages", action="store_true", help="print the per-stage table to stderr")
    ap.add_argument("--lanes", type=int, default=2,
                    help="independent engine contexts (own stream + activation buffers); consecutive batches alternate lanes, the hardware interleaves "
                         "the lanes' kernels (one lane's latency-bound glue and HBM-bound tile prologues / epilogues run beside the other's matrix work)")
    ap.add_argument("--shared-stream", action="store_true",
                    help="rounds 2-4 form of the lanes: convolutions of all lanes serialised on one stream, only the detection glue on side streams")
    ap.add_argument("--profile-mode", type=int, default=3,
                    help="HIP-event stage timing during the timed steps: 3 = every 4th step (default), 2 = every step, 0 = off")
    ap.add_argument("--no-fp32-mode", "--no-reference-precision", dest="no_ref", action="store_true",
                    help="skip the reference-precision (fp32 MFMA) leg and the parity object")
    ap.add_argument("--no-trained-leg", action="store_true", help="skip the trained-like leg of the default run")
    ap.add_argument("--no-train-leg", action="store_true", help="skip the training-step leg of the default run")
    ap.add_argument("--train", action="store_true", help="ONLY the training-step leg, as its own JSON line (BASELINE configs[4])")
    ap.add_argument("--train-legs", default="b8,b1,fp32", help="with --train: which trainers to time (b8 = fp16 batch 8, b1 = fp16 one image, fp32 = reference precision batch 8); profiler runs take b8 alone")
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
    # One rank per GPU is the contract.  With fewer devices than local ranks (a rehearsal of the N > 1 path on a one-GPU box) the ranks share
    # devices round-robin and rendezvous over gloo (RCCL refuses two ranks on one device); the line then says so and is not a scaling point.
    ndev = torch.cuda.device_count()
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    shared = local_world > ndev
    dev = local_rank % ndev
    rdev = "cpu" if shared else "cuda"           # where the barrier / max-over-ranks tensors live
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with stdout_to_stderr():
            dist.init_process_group(backend="gloo" if shared else "nccl", rank=rank, world_size=world)
            dist.barrier()                       # the communicator (and RCCL's banner) comes up here, not inside the timed region

    from proj_roadsurf_amd.engine import LanePipeline
    from proj_roadsurf_amd.spec import EngineSpec
    from proj_roadsurf_amd.weights import synthetic_weights
    from proj_roadsurf_amd.synthetic import synthetic_scenes, synthetic_tiles, train_trained_like

    spec = EngineSpec(num_classes=2)      # R:config/detectron2_config_3bands.yaml defaults, 2 classes (artificial/natural)
    if args.bands == 4:                   # no 4-band YAML exists in the reference: PIXEL_MEAN/STD extended by a NIR entry
        spec = spec.replace(pixel_mean=spec.pixel_mean + (110.0,), pixel_std=spec.pixel_std + (1.0,))
    B, T, C_in = args.batch, args.tile, args.bands

    if args.train:
        if world != 1:
            raise SystemExit("--train measures one GPU (the 8-GPU data-parallel run is the driver's)")
        with stdout_to_stderr():
            tl = training_leg(spec, synthetic_weights(spec, seed=0), dev, steps=args.steps, warmup=args.warmup,
                              legs=tuple(["b8"] + [x for x in args.train_legs.split(",") if x]))
        b8 = tl["batch8"]
        print(json.dumps({"metric": "train_images_per_sec_512x512x3", "value": b8["images_per_s"], "unit": "images/s", "n_gpus": 1,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": b8["ms_per_step"], "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "f16 operands / f32 accumulate, f32 master weights", "data": "synthetic",
                          "config": {"workload": tl["workload"], "batch_per_gpu": 8}, "training": tl}))
        return

    def barrier():
        if world > 1:
            dist.barrier()

    def measure(W, tiles, want_stage_events, precision=None):
        """headline-style measurement of one (weights, tiles) workload: K-step region on the lane pipeline + detections of batch 0"""
        L = max(1, args.lanes)
        pipe = LanePipeline(spec.replace(precision=precision or args.precision), W, (T, T, C_in), max_batch=B, device=dev, lanes=L, shared_stream=args.shared_stream)
        try:
            engs = pipe.engines
            ptrs = [e.upload_tiles(tiles) for e in engs]
            for k in range(args.warmup):
                pipe.submit(ptrs[k % L], B)
            pipe.sync()
            if want_stage_events:
                for e in engs:
                    e.set_profiling(args.profile_mode)   # HIP events around the launches, on each lane's stream, no host wait
            dt = timed_steps(pipe, ptrs, B, args.steps, 0, barrier, torch.cuda.synchronize)
            if world > 1:
                tmax = torch.tensor([dt], dtype=torch.float64, device=rdev)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                dt = float(tmax.item())
            stages = None
            if want_stage_events:
                stages = engs[0].stage_times()           # the stage events of the timed region (set_profiling(0) would clear them)
                for e in engs[1:]:                       # same stage list on every lane: pool the HIP-event totals
                    for a, b in zip(stages, e.stage_times()):
                        a["ms_total"] += b["ms_total"]
                        a["calls"] += b["calls"]
            # Sustained rate: the K-step region above is what the contract times (`value`); a 10k-tile job runs for seconds, by which
            # time the chip has settled at its clock under load.  Same loop, no stage events, >= --sustain-seconds, max over ranks.
            sustained = None
            if want_stage_events and args.sustain_seconds > 0:
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
                    stop = torch.tensor([1.0 if time.perf_counter() - ts >= args.sustain_seconds else 0.0], device=rdev)
                    if world > 1:
                        dist.all_reduce(stop, op=dist.ReduceOp.MAX)    # every rank leaves the loop after the same number of steps
                    if float(stop.item()) > 0:
                        break
                torch.cuda.synchronize()
                barrier()
                dts = time.perf_counter() - ts
                if world > 1:
                    tm = torch.tensor([dts], dtype=torch.float64, device=rdev)
                    dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                    dts = float(tm.item())
                sustained = {"tiles_per_s": world * B * n_sus / dts, "steps": n_sus, "seconds": dts}
            for e in engs:
                e.set_profiling(0)
            # Per-kernel durations need launches that do not overlap: with independent lanes two kernels share the chip in the timed region and a
            # launch's HIP-event duration there is its time on a shared chip.  So the same K steps run once more on ONE lane, stage events on its stream.
            stages_serial = None
            if want_stage_events and L > 1 and not pipe.shared:
                e0 = engs[0]
                for _ in range(2):
                    e0.infer_device(ptrs[0], B)
                e0.sync()
                e0.set_profiling(2)
                for _ in range(args.steps):
                    e0.infer_device(ptrs[0], B)
                e0.sync()
                stages_serial = e0.stage_times()
                e0.set_profiling(0)
            # PCIe-inclusive rate of the streaming host interface (pinned H2D of the tiles + forward + D2H of boxes/scores/packed
            # mask crops + host-side collection into Instances, LanePipeline.run); reported beside the headline, never as `value`
            nb = 24                                 # long enough that filling and draining the two lanes is a small part of it
            for _ in pipe.run(tiles for _ in range(2)):
                pass
            t1 = time.perf_counter()
            for _res in pipe.run(tiles for _ in range(nb)):
                pass
            pcie = nb * B / (time.perf_counter() - t1)
            eng = engs[0]
            eng.infer_device(ptrs[0], B)
            dets = eng.fetch(B)
            nprop = eng.tensor("proposal_count", n=B)
            return {"dt": dt, "stages": stages, "stages_serial": stages_serial, "sustained": sustained, "pcie": pcie, "dets": dets, "nprop": float(np.mean(nprop)),
                    "ndet": float(np.mean([len(d) for d in dets])), "lanes": L, "shared_stream": pipe.shared}
        finally:
            pipe.close()

    W_rand = synthetic_weights(spec, seed=0)
    # rank r owns tiles r*B .. r*B+B-1 of the synthetic tileset (seed = 1234 + tile id)
    tiles_rand = synthetic_tiles(B, T, T, C_in, seed=1234 + rank * B)
    W_tr = tiles_tr = None
    need_trained = args.weights == "trained" or (rank == 0 and world == 1 and not args.no_trained_leg and C_in == 3)
    trained_error = None
    if need_trained:
        t0 = time.time()
        try:
            W_tr, curve = train_trained_like(spec, T, steps=args.train_steps, seed=rank)
            tiles_tr = synthetic_scenes(B, T, T, C_in, seed=555000 + rank, objects=(4, 12))[0]
            log(f"trained-like detector: {args.train_steps} steps in {time.time() - t0:.1f} s, loss {curve[0]:.2f} -> {np.mean(curve[-20:]):.2f}")
        except RuntimeError as ex:               # a diverged training run must not cost the headline line (it is reported in it)
            if args.weights == "trained":
                raise
            W_tr, trained_error = None, str(ex)
            log(f"trained-like leg skipped: {ex}")
    W, tiles = (W_tr, tiles_tr) if args.weights == "trained" else (W_rand, tiles_rand)
    H = measure(W, tiles, True)
    dt, stages = H["dt"], H["stages"]

    ref = par = fp16_mode = None
    PNAME = {"split": "split-operand (hi + lo fp16 planes, 3 MFMA products)", "fp16": "fp16-operand", "fp32": "fp32-MFMA"}
    if rank == 0 and not args.no_ref:
        ref, dets32 = reference_precision_leg(spec, W, tiles, B, args.steps, args.warmup, dev)
        par = parity_object(H["dets"], dets32, f"{PNAME[args.precision]} engine (the engine `value` is measured on) vs reference-precision (fp32 MFMA) engine, the benched batch of this line, GPU vs GPU")
        log(f"reference precision: {ref['tiles_per_s']:.0f} tiles/s; parity of the {args.precision} headline {par['matched_fw']:.3f} / {par['matched_bw']:.3f} of {par['n_fw']}")
    if rank == 0 and world == 1 and args.precision != "fp16" and not args.no_fp16_leg:
        try:
            H16 = measure(W, tiles, False, "fp16")
            fp16_mode = {"tiles_per_s": B * args.steps / H16["dt"], "ms_per_step": H16["dt"] / args.steps * 1e3, "steps": args.steps, "warmup": args.warmup,
                         "dtype": "f16 operands / f32 accumulate", "pcie_inclusive_tiles_per_s": H16["pcie"],
                         "note": "the round 1-3 headline mode: meets the stated tolerance on a trained detector (trained_like.parity_fp16), not on this saturated random-weight workload"}
            if not args.no_ref:
                fp16_mode["parity"] = parity_object(H16["dets"], dets32, "fp16-operand engine vs reference-precision (fp32 MFMA) engine, the benched batch of this line, GPU vs GPU")
            log(f"fp16 mode: {fp16_mode['tiles_per_s']:.0f} tiles/s")
        except Exception as ex:
            fp16_mode = {"error": f"{type(ex).__name__}: {ex}"}
    # one tile at a time, as the reference's `predictor(im)` loop submits them ([EXT od] make_detections.py): device-resident latency of a batch of 1
    single = None
    if rank == 0 and world == 1 and not args.no_single_tile_leg:
        try:
            from proj_roadsurf_amd.engine import Engine
            single = {"what": "ms per forward of ONE tile (batch 1, tile resident in HBM, one engine, no overlap between tiles): the latency of a `predictor(im)` call without its PCIe copies"}
            for prec in ([args.precision] + (["fp16"] if args.precision != "fp16" else [])):
                e1 = Engine(spec.replace(precision=prec), W, (T, T, C_in), max_batch=1, device=dev)
                try:
                    p1 = e1.upload_tiles(tiles[:1])
                    for _ in range(5):
                        e1.infer_device(p1, 1)
                    e1.sync()
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(40):
                        e1.infer_device(p1, 1)
                    e1.sync()
                    torch.cuda.synchronize()
                    single[prec + "_ms"] = (time.perf_counter() - t1) / 40 * 1e3
                finally:
                    e1.close()
        except Exception as ex:
            single = {"error": f"{type(ex).__name__}: {ex}"}
    trained = None
    if rank == 0 and world == 1 and args.weights == "random" and W_tr is not None:
        try:
            Ht = measure(W_tr, tiles_tr, False)
            Ht16 = measure(W_tr, tiles_tr, False, "fp16") if args.precision != "fp16" else Ht
            trained = {"workload": f"detector trained here ({args.train_steps} SGD steps on synthetic scenes, two classes), batch {B} of fresh {T}x{T}x{C_in} scenes with 4-12 objects",
                       "tiles_per_s": B * args.steps / Ht["dt"], "ms_per_step": Ht["dt"] / args.steps * 1e3, "steps": args.steps,
                       "proposals_per_tile": Ht["nprop"], "detections_per_tile": Ht["ndet"], "pcie_inclusive_tiles_per_s": Ht["pcie"],
                       "precision": args.precision, "fp16_tiles_per_s": B * args.steps / Ht16["dt"], "fp16_pcie_inclusive_tiles_per_s": Ht16["pcie"]}
            if not args.no_ref:
                rt, d32t = reference_precision_leg(spec, W_tr, tiles_tr, B, max(3, args.steps // 4), 1, dev)
                trained["reference_precision_tiles_per_s"] = rt["tiles_per_s"]
                # parity over a pool that makes the >= 0.98 bar decidable: 12 batches of fresh scenes (~1500 detections), both engines on the GPU
                from proj_roadsurf_amd.engine import Engine
                got, got16, want = list(Ht["dets"]), list(Ht16["dets"]), list(d32t)
                eh = Engine(spec.replace(precision=args.precision), W_tr, (T, T, C_in), max_batch=B, device=dev)
                e16 = Engine(spec.replace(precision="fp16"), W_tr, (T, T, C_in), max_batch=B, device=dev) if args.precision != "fp16" else eh
                e32 = Engine(spec.replace(precision="fp32"), W_tr, (T, T, C_in), max_batch=B, device=dev)
                try:
                    for k in range(1, 12):
                        more = synthetic_scenes(B, T, T, C_in, seed=555000 + 7919 * k, objects=(4, 12))[0]
                        got += eh.infer(more)
                        if e16 is not eh:
                            got16 += e16.infer(more)
                        want += e32.infer(more)
                finally:
                    eh.close(); e32.close()
                    if e16 is not eh:
                        e16.close()
                trained["parity"] = parity_object(got, want, f"{PNAME[args.precision]} engine vs reference-precision engine on {len(got)} trained-like scenes, GPU vs GPU")
                if e16 is not eh:
                    trained["parity_fp16"] = parity_object(got16, want, f"fp16-operand engine vs reference-precision engine on {len(got16)} trained-like scenes, GPU vs GPU")
                log(f"trained-like: {trained['tiles_per_s']:.0f} tiles/s; parity {trained['parity']['matched_fw']:.4f} / {trained['parity']['matched_bw']:.4f} of {trained['parity']['n_fw']}, "
                    f"Wilson lower {trained['parity']['wilson95_lower_fw']:.4f}")
        except Exception as ex:
            trained = {"error": f"{type(ex).__name__}: {ex}"}
            log(f"trained-like leg failed: {ex}")
    training = None
    if rank == 0 and world == 1 and not args.no_train_leg and C_in == 3 and T == 512:
        try:                                     # an optional leg never costs the headline: its failure is reported inside the line
            with stdout_to_stderr():
                training = training_leg(spec, W_rand, dev)
        except Exception as ex:
            training = {"error": f"{type(ex).__name__}: {ex}"}
            log(f"training leg failed: {ex}")

    if rank == 0:
        value = world * B * args.steps / dt
        # the mask head's stages are sized for DETECTIONS_PER_IMAGE entries per tile but run on the detections there are (device-side row
        # count): their algorithmic FLOP / bytes count the rows that exist (all 100 on the saturated random-weight workload, ~8 on a trained one)
        mask_fill = min(1.0, float(H["ndet"]) / float(spec.detections_per_image)) if spec.detections_per_image else 1.0
        def fill(st):
            return [dict(s, flops=s["flops"] * mask_fill, bytes=s["bytes"] * mask_fill) if s["name"].startswith("mask.") else s for s in st]

        def kernel_groups(st):
            """group the GEMM stages by the kernel symbol (tile variant) they launched"""
            gr = {}
            for s in st:
                if s["flops"] > 0 and s["calls"] > 0:
                    g = gr.setdefault(s["kernel"] or "?", {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0})
                    g["ms"] += s["ms_total"]
                    g["flops"] += s["flops"] * s["calls"]
                    g["bytes"] += s["bytes"] * s["calls"]
                    g["launches"] += s["calls"]
            return gr
        stages_region = fill(stages)                       # stage events of the timed region (independent lanes: launches of the two lanes overlap)
        serial = H.get("stages_serial") is not None
        stages = fill(H["stages_serial"]) if serial else stages_region
        conv = [s for s in stages if s["flops"] > 0 and s["calls"] > 0]
        by_time = sorted(stages, key=lambda s: -s["ms_total"])
        tot_ms = sum(s["ms_total"] for s in stages)
        groups = kernel_groups(stages)
        if not groups:                           # --profile-mode 0: no stage timing, no roofline
            groups = {"(stage timing off)": {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0}}
        dom = max(groups, key=lambda k: groups[k]["ms"])        # the dominant kernel is the one with the largest share of the step
        G = groups[dom]
        achieved = G["flops"] / (G["ms"] * 1e-3) / 1e12 if G["ms"] > 0 else 0.0
        total_flops_step = sum(s["flops"] for s in conv)
        GR = kernel_groups(stages_region).get(dom) if serial else None
        traffic, traffic_src = None, None
        # split-operand mode: three fp16 MFMA products per real product, so the roof of ALGORITHMIC FLOP/s is a third of the matrix peak
        split = args.precision == "split"
        peak = {"split": MFMA_PEAK_TFLOPS / 3.0, "fp16": MFMA_PEAK_TFLOPS, "fp32": F32_MFMA_PEAK_TFLOPS}[args.precision]
        pmc = os.path.join(ROOT, "profiles", "pmc_latest_split.json" if split else "pmc_latest.json")     # written by tools/pmc_summary.py from two --pmc passes
        if os.path.exists(pmc) and B == 16 and T == 512 and args.weights == "random" and args.precision != "fp32":
            sym = dom.split(" ")[0]                                   # "conv_deep_kernel" or "conv_igemm_kernel<2,4,4,8>"
            want = sym.replace("conv_igemm_kernel<", "").replace(">", "").replace(",", ", ")
            for k in json.load(open(pmc)):
                if sym.startswith("conv_igemm"):
                    hit = f"conv_igemm_kernel<{want}, false, true" in k["kernel"]
                elif sym == "conv_deep_kernel" and split:
                    hit = "conv_deep_kernel<0, false, 8, true>" in k["kernel"]
                elif sym == "conv_deep_kernel":                         # the 256-pixel inference instantiation: conv_deep_kernel<0, false, 8> (or <0, false> before the tile height became a parameter)
                    hit = ("conv_deep_kernel<0, false, 8>" in k["kernel"] or "conv_deep_kernel<0, false>" in k["kernel"] or "conv_deep_kernel<0, false, 8, false>" in k["kernel"])
                else:
                    hit = sym in k["kernel"]
                if hit:
                    traffic = k["hbm_bytes_per_launch_corrected"]
                    traffic_src = f"{os.path.relpath(pmc, ROOT)}: rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE passes of this command, (2*FETCH+WRITE)*1024 per launch"
                    break
        # matrix-pipe busy fraction, clock under load and wave-state split of the same kernel from the SQ / GRBM counter passes of tools/profile_bench.sh
        counters = None
        pmf = os.path.join(ROOT, "profiles", "pmc_mfma_latest_split.json" if split else "pmc_mfma_latest.json")
        if traffic is not None and os.path.exists(pmf):
            key = "conv_deep_kernel<0, false, 8, true>" if split else "conv_deep_kernel<0, false, 8, false>"
            for k in json.load(open(pmf)):
                if key in k["kernel"] and dom.startswith("conv_deep_kernel 256x256"):
                    counters = {kk: k.get(kk) for kk in ("mfma_busy_frac", "clock_ghz_under_load", "wave_parked_frac", "issue_stall_frac", "issuing_frac", "lds_issue_stall_frac",
                                                         "avg_duration_us_under_pmc")}
                    counters["source"] = f"{os.path.relpath(pmf, ROOT)} (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs); clock = GRBM_GUI_ACTIVE / 8 / kernel time)"
                    counters["measured_in_this_run"] = False
                    counters["measured"] = json.load(open(pmf + ".meta")) if os.path.exists(pmf + ".meta") else None
                    break
        roofline = {"bound": "mfma", "kernel": dom + (" (fp16 MFMA 16x16x32, fp32 accumulate)" if args.precision != "fp32" else " (fp32 MFMA 16x16x4)"),
                    "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                    "peak_is": ("2500 TFLOP/s dense fp16 MFMA / 3: the split-operand mode spends three matrix-core products (hi.hi, hi.lo, lo.hi) per real product; "
                                "`achieved` counts ALGORITHMIC FLOP (2 per real multiply-add), matrix_core_tflops = 3 x achieved is what the MFMA pipe delivers"
                                if split else "dense MFMA peak of the operand type (MI355X_MICROARCH.md)"),
                    "matrix_core_tflops": achieved * (3.0 if split else 1.0),
                    "counters": counters,
                    "traffic": traffic, "traffic_source": traffic_src,
                    # the PMC passes need rocprofv3 and run separately (tools/profile_bench.sh); the file is that run's summary, not this run's
                    "traffic_measured_in_this_run": False if traffic is not None else None,
                    "traffic_measured": (json.load(open(pmc + ".meta")) if traffic is not None and os.path.exists(pmc + ".meta") else None),
                    "algorithmic_bytes_per_launch_avg": G["bytes"] / max(G["launches"], 1), "launches": G["launches"],
                    "avg_launch_ms": G["ms"] / max(G["launches"], 1), "flops_per_launch_avg": G["flops"] / max(G["launches"], 1),
                    "share_of_step_time": G["ms"] / tot_ms if tot_ms else None,
                    "measured_on": ("the same K steps on ONE lane right after the timed region, HIP events on its stream (launches do not overlap there): in the timed "
                                    "region the lanes run on independent streams, two kernels share the chip and a launch's duration is not the kernel's own "
                                    "-- those numbers are in `in_timed_region`" if serial else "HIP events around every launch of the timed region, on the stream it runs on"),
                    "in_timed_region": ({"avg_launch_ms": GR["ms"] / max(GR["launches"], 1), "launches": GR["launches"],
                                         "achieved_on_a_shared_chip": GR["flops"] / (GR["ms"] * 1e-3) / 1e12 if GR["ms"] else 0.0,
                                         "sum_of_stage_ms_per_step": sum(s["ms_total"] / s["calls"] for s in stages_region if s["calls"]),
                                         "note": "stage durations of the two lanes overlap: their sum exceeds ms_per_step"} if GR else None),
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
        wl = ("seeded synthetic weights, saturated: 1000 proposals and 100 detections per tile" if args.weights == "random"
              else f"detector trained here for {args.train_steps} steps, synthetic scenes with 4-12 objects")
        out = {
            "metric": f"tiles_per_sec_{T}x{T}x{C_in}", "value": value, "unit": "tiles/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"split": "f16 hi + lo operand planes (22 significand bits, 3 MFMA products per product) / f32 accumulate", "fp16": "f16 operands / f32 accumulate",
                      "fp32": "f32 operands (v_mfma_f32_16x16x4_f32) / f32 accumulate"}[args.precision],
            "precision_mode": args.precision, "data": "synthetic",
            "config": {"workload": f"Mask R-CNN R50-FPN inference, batch {B} of {T}x{T} {C_in}-band tiles per GPU "
                                   f"(BASELINE configs[{1 if C_in == 3 else 3}]), 800x800 network input; {wl}",
                       "batch_per_gpu": B, "lanes": H["lanes"], "lane_streams": "one shared stream" if H["shared_stream"] else "independent", "tile": [T, T, C_in], "num_classes": 2, "weights": args.weights,
                       "proposals_per_tile": H["nprop"], "detections_per_tile": H["ndet"],
                       "sharding": "tiles across ranks, no data-path collective"},
            "roofline": roofline,
            "reference_precision": ref,
            "parity": par,
            "fp16_mode": fp16_mode,
            "single_tile_latency": single,
            "trained_like": trained if trained is not None else ({"skipped": trained_error} if trained_error else None),
            "training": training,
            "value_is": f"the {args.steps} timed steps after {args.warmup} warm-up steps (driver contract); sustained_tiles_per_s = the same loop "
                        "run for >= --sustain-seconds right after it; reference_precision = the same K-step region on the fp32-MFMA engine",
            "sustained_tiles_per_s": H["sustained"]["tiles_per_s"] if H["sustained"] else None,
            "sustained": H["sustained"],
            "rccl_world_size": (dist.get_world_size() if world > 1 else 1),
            **({"rehearsal": f"{local_world} ranks share {ndev} device(s): gloo rendezvous instead of RCCL, `value` is NOT a scaling point"} if shared else {}),
            "pcie_inclusive_tiles_per_s": H["pcie"],
            "top_stages": [{"name": s["name"], "ms_per_step": s["ms_total"] / max(s["calls"], 1)} for s in by_time[:6]],
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(spec, W, tiles)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()            # rank 0 runs its extra legs after the timed region: nobody tears the group down under it
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
