"""Weight I/O for the engine (host Python only, as BASELINE.json:north_star allows).

* ``synthetic_weights``  -- seeded random weights in detectron2's checkpoint key layout
  (SURVEY.md §8c key list).  No trained weights exist offline (R:config/config_obj_detec.yaml:86
  is a training artefact, R:config/detectron2_config_3bands.yaml:265 a URL), so benches and parity
  tests use these.
* ``load_checkpoint``    -- reads a detectron2 ``.pth`` (``torch.save({'model': state_dict, ...})``)
  with ``weights_only=True``; the model-zoo ``.pkl`` through ``load_zoo_pkl``, a restricted unpickler
  that admits numpy array reconstruction only and refuses every other global (nothing from the file runs).
* ``pack_weights``       -- folds FrozenBN into conv scale/bias, converts to the engine's
  kernel-ready layout (fp16 ``[Cout][KH][KW][Cin]``, K padded to 64, fp32 bias) and serialises to
  the flat blob ``rs_engine_create`` consumes (format: include/rs_engine.h "weight blob").
"""
from __future__ import annotations

import math
import os
import struct
from typing import Dict, List, Optional, Tuple

import numpy as np

from .spec import EngineSpec

BLOB_MAGIC = 0x52534557  # 'RSEW'
BLOB_VERSION = 1
DT_F16, DT_F32, DT_I32 = 1, 2, 3
ALIGN = 256
K_ALIGN = 64          # GEMM K padding (elements) -- one LDS K-step of the conv kernel
STEM_CIN_PAD = 4      # stem input channels padded to 4 (8 bytes per pixel)
STEM_KW_PAD = 8       # stem tap rows padded 7 -> 8 so that a 16-byte chunk = two adjacent taps (the 8th has zero weights)


# ----------------------------------------------------------------------------- topology
def conv_layers(spec: EngineSpec) -> List[Tuple[str, int, int, int, bool]]:
    """(name, cin, cout, k, has_norm) for every conv of the model in execution order."""
    L: List[Tuple[str, int, int, int, bool]] = []
    p = "backbone.bottom_up."
    L.append((p + "stem.conv1", spec.in_channels, spec.stem_out_channels, 7, True))
    cin = spec.stem_out_channels
    bott = spec.num_groups * spec.width_per_group
    cout = spec.res2_out_channels
    for si, nb in enumerate(spec.res_blocks):
        for bi in range(nb):
            n = f"{p}res{si + 2}.{bi}"
            if cin != cout:
                L.append((n + ".shortcut", cin, cout, 1, True))
            L.append((n + ".conv1", cin, bott, 1, True))
            L.append((n + ".conv2", bott, bott, 3, True))
            L.append((n + ".conv3", bott, cout, 1, True))
            cin = cout
        bott *= 2
        cout *= 2
    res_c = [spec.res2_out_channels * (2 ** i) for i in range(4)]
    for l, c in zip((2, 3, 4, 5), res_c):
        L.append((f"backbone.fpn_lateral{l}", c, spec.fpn_out_channels, 1, False))
        L.append((f"backbone.fpn_output{l}", spec.fpn_out_channels, spec.fpn_out_channels, 3, False))
    A = spec.num_anchors
    f = spec.fpn_out_channels
    L.append(("proposal_generator.rpn_head.conv", f, f, 3, False))
    L.append(("proposal_generator.rpn_head.objectness_logits", f, A, 1, False))
    L.append(("proposal_generator.rpn_head.anchor_deltas", f, 4 * A, 1, False))
    if spec.mask_on:
        c = f
        for i in range(spec.mask_num_conv):
            L.append((f"roi_heads.mask_head.mask_fcn{i + 1}", c, spec.mask_conv_dim, 3, False))
            c = spec.mask_conv_dim
    return L


# ----------------------------------------------------------------------------- synthetic
def synthetic_weights(spec: EngineSpec, seed: int = 0) -> Dict[str, np.ndarray]:
    """Seeded random weights, detectron2 key names, fp32 numpy.

    The init is chosen so that activations stay O(1) through all 16 bottlenecks (fp16-safe, like a
    trained net) and so that objectness / class / mask logits are well spread (std ~1.5-2): with
    detectron2's own tiny head inits (std 0.01/0.001) every score would tie at 1/(K+1) and any
    fp difference would reshuffle NMS, which tests nothing.  Documented in DESIGN.md "Synthetic
    workload"."""
    rng = np.random.default_rng(seed)
    W: Dict[str, np.ndarray] = {}

    def conv(name: str, cin: int, cout: int, k: int, gain: float, bias: bool) -> None:
        std = gain / math.sqrt(cin * k * k)
        W[name + ".weight"] = (rng.standard_normal((cout, cin, k, k)) * std).astype(np.float32)
        if bias:
            W[name + ".bias"] = (rng.standard_normal(cout) * 0.05).astype(np.float32)

    def bn(name: str, c: int, scale: float) -> None:
        W[name + ".weight"] = (rng.uniform(0.7, 1.3, c) * scale).astype(np.float32)
        W[name + ".bias"] = (rng.standard_normal(c) * 0.1).astype(np.float32)
        W[name + ".running_mean"] = (rng.standard_normal(c) * 0.1).astype(np.float32)
        W[name + ".running_var"] = rng.uniform(0.6, 1.4, c).astype(np.float32)

    s2 = math.sqrt(2.0)
    for name, cin, cout, k, has_norm in conv_layers(spec):
        if has_norm:
            if name.endswith("stem.conv1"):
                conv(name, cin, cout, k, 1.0, False)
                bn(name + ".norm", cout, 1.0 / 64.0)          # pixels are O(100) -> O(1)
            elif name.endswith(".conv3"):
                conv(name, cin, cout, k, s2, False)
                bn(name + ".norm", cout, 0.35)                # small residual branch
            elif name.endswith(".shortcut"):
                conv(name, cin, cout, k, 1.0, False)
                bn(name + ".norm", cout, 1.0)
            else:
                conv(name, cin, cout, k, s2, False)
                bn(name + ".norm", cout, 1.0)
        elif "fpn_lateral" in name:
            conv(name, cin, cout, k, 0.35, True)
        elif "fpn_output" in name:
            conv(name, cin, cout, k, 0.6, True)
        elif name.endswith("rpn_head.conv"):
            conv(name, cin, cout, k, s2, True)
        elif name.endswith("objectness_logits"):
            conv(name, cin, cout, k, 1.0, True)
        elif name.endswith("anchor_deltas"):
            conv(name, cin, cout, k, 0.35, True)
        else:                                                  # mask_fcn*
            conv(name, cin, cout, k, s2, True)

    def linear(name: str, cin: int, cout: int, gain: float) -> None:
        W[name + ".weight"] = (rng.standard_normal((cout, cin)) * (gain / math.sqrt(cin))).astype(np.float32)
        W[name + ".bias"] = (rng.standard_normal(cout) * 0.05).astype(np.float32)

    r = spec.box_pooler_resolution
    K = spec.num_classes
    linear("roi_heads.box_head.fc1", spec.fpn_out_channels * r * r, spec.box_fc_dim, s2)
    linear("roi_heads.box_head.fc2", spec.box_fc_dim, spec.box_fc_dim, s2)
    linear("roi_heads.box_predictor.cls_score", spec.box_fc_dim, K + 1, 1.6)
    linear("roi_heads.box_predictor.bbox_pred", spec.box_fc_dim, 4 * K, 0.5)
    if spec.mask_on:
        d = spec.mask_conv_dim
        # ConvTranspose2d weight is (Cin, Cout, 2, 2)
        W["roi_heads.mask_head.deconv.weight"] = (rng.standard_normal((d, d, 2, 2)) * (s2 / math.sqrt(d))).astype(np.float32)
        W["roi_heads.mask_head.deconv.bias"] = (rng.standard_normal(d) * 0.05).astype(np.float32)
        W["roi_heads.mask_head.predictor.weight"] = (rng.standard_normal((K, d, 1, 1)) * (6.0 / math.sqrt(d))).astype(np.float32)   # crisp masks: logits std ~4, like a trained head
        W["roi_heads.mask_head.predictor.bias"] = (rng.standard_normal(K) * 0.05).astype(np.float32)
    return W


def load_checkpoint(path: str) -> Dict[str, np.ndarray]:
    """Read a detectron2-style checkpoint: ``.pth`` (``{'model': state_dict}``) via
    ``torch.load(weights_only=True)`` or ``.npz``.  Returns fp32 numpy arrays by key."""
    if path.endswith(".npz"):
        with np.load(path, allow_pickle=False) as z:
            return {k: np.asarray(z[k], dtype=np.float32) for k in z.files}
    if path.endswith(".pkl"):
        return load_zoo_pkl(path)
    import torch

    ck = torch.load(path, map_location="cpu", weights_only=True)
    sd = ck["model"] if isinstance(ck, dict) and "model" in ck else ck
    return {k: v.detach().float().numpy() for k, v in sd.items() if hasattr(v, "detach")}


# globals a dict-of-ndarrays pickle needs, and nothing else: numpy's array / dtype / scalar reconstructors, the dict
# type detectron2's zoo files use, and the str->bytes helper protocol-2 pickles written by Python 2 carry
_PKL_ALLOWED = {
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
    ("numpy", "ndarray"), ("numpy", "dtype"), ("collections", "OrderedDict"), ("_codecs", "encode"),
}


def load_zoo_pkl(path: str) -> Dict[str, np.ndarray]:
    """Code-free reader for detectron2 model-zoo ``.pkl`` checkpoints (R:config/detectron2_config_3bands.yaml:265,
    R:config/config_obj_detec.yaml:71-72: ``{"model": {name: ndarray}, "__author__": ...}``).  A ``pickle.Unpickler``
    whose ``find_class`` admits ONLY the symbols in ``_PKL_ALLOWED`` (numpy array / dtype / scalar reconstruction,
    ``OrderedDict``, ``_codecs.encode``); any other global -- i.e. any pickle that would import or call something --
    raises ``pickle.UnpicklingError`` before anything of it runs.  Object-dtype arrays are rejected as well.
    Keys must already be detectron2's names (zoo files of detectron2-trained models are; Caffe2-converted files with
    ``matching_heuristics`` are not supported)."""
    import importlib
    import pickle

    class _Restricted(pickle.Unpickler):
        def find_class(self, module: str, name: str):
            if (module, name) not in _PKL_ALLOWED:
                raise pickle.UnpicklingError(f"{path}: pickle references {module}.{name}; only numpy array data is accepted")
            return getattr(importlib.import_module(module), name)

    with open(path, "rb") as f:
        obj = _Restricted(f, encoding="latin1").load()
    model = obj["model"] if isinstance(obj, dict) and "model" in obj else obj
    if not isinstance(model, dict):
        raise ValueError(f"{path}: expected a dict of arrays, got {type(model).__name__}")
    out: Dict[str, np.ndarray] = {}
    for k, v in model.items():
        a = np.asarray(v)
        if a.dtype == object or not isinstance(k, str):
            raise ValueError(f"{path}: entry {k!r} is not numeric array data")
        out[k] = a.astype(np.float32)
    if not any(k.startswith("backbone.bottom_up.") for k in out):
        raise ValueError(f"{path}: keys are not detectron2 names (Caffe2-style checkpoints need detectron2's c2_model_loading "
                         "renaming, which is not restated here)")
    return out


# detectron2's model-zoo table for the entries the reference's YAMLs can name ([EXT d2: model_zoo/model_zoo.py]
# ``_ModelZooUrls.CONFIG_PATH_TO_URL_SUFFIX``; R:config/config_obj_detec.yaml:71-72, R:config/detectron2_config_3bands.yaml:265)
ZOO_URL_SUFFIX = {
    "COCO-InstanceSegmentation/mask_rcnn_R_50_FPN_1x": "137260431/model_final_a54504.pkl",
    "COCO-InstanceSegmentation/mask_rcnn_R_50_FPN_3x": "137849600/model_final_f10217.pkl",
}
ZOO_S3_PREFIX = "https://dl.fbaipublicfiles.com/detectron2/"


def zoo_cache_roots() -> List[str]:
    """Where iopath / fvcore keep downloaded files ([EXT iopath: common/file_io.py get_cache_dir]): ``$FVCORE_CACHE``, else
    ``~/.torch/iopath_cache`` (older fvcore: ``~/.torch/fvcore_cache``)."""
    roots = []
    if os.environ.get("FVCORE_CACHE"):
        roots.append(os.path.expanduser(os.environ["FVCORE_CACHE"]))
    roots += [os.path.expanduser("~/.torch/iopath_cache"), os.path.expanduser("~/.torch/fvcore_cache")]
    return roots


def resolve_zoo_checkpoint(name_or_url: str) -> Optional[str]:
    """Local file of a model-zoo checkpoint WITHOUT network access: ``model_zoo.get_checkpoint_url(name)`` would download
    ``https://dl.fbaipublicfiles.com/detectron2/<name>/<id>/model_final_<hash>.pkl`` once and iopath's HTTP handler keeps it at
    ``<cache root>/detectron2/<name>/<id>/model_final_<hash>.pkl`` (the URL's path under the cache root).  A machine that ran
    the reference once -- or whose cache was populated by hand -- therefore already holds the file.  ``name_or_url`` is the YAML's
    zoo config name (``COCO-InstanceSegmentation/mask_rcnn_R_50_FPN_1x.yaml``) or the full https URL (``MODEL.WEIGHTS``).
    Returns the path or None."""
    import glob

    s = name_or_url.strip()
    if s.startswith("http://") or s.startswith("https://"):
        rel = s.split("://", 1)[1].split("/", 1)[1] if "/" in s.split("://", 1)[1] else ""
        cands = [rel]
    else:
        name = s[:-5] if s.endswith(".yaml") else s
        cands = []
        if name in ZOO_URL_SUFFIX:
            cands.append(f"detectron2/{name}/{ZOO_URL_SUFFIX[name]}")
        cands.append(f"detectron2/{name}/*/model_final_*.pkl")
    for root in zoo_cache_roots():
        for c in cands:
            hits = sorted(glob.glob(os.path.join(root, c)))
            if len(hits) == 1 or (hits and "*" not in c):
                return hits[0]
            if len(hits) > 1:
                raise ValueError(f"several cached checkpoints match {name_or_url!r} under {root}: {hits}")
    return None


# layers whose shape follows ROI_HEADS.NUM_CLASSES, with detectron2's own initial scale
# ([EXT d2: modeling/roi_heads/fast_rcnn.py FastRCNNOutputLayers.__init__, mask_head.py MaskRCNNConvUpsampleHead.__init__])
_CLASS_SHAPED = (("roi_heads.box_predictor.cls_score", 0.01, lambda k: k + 1), ("roi_heads.box_predictor.bbox_pred", 0.001, lambda k: 4 * k),
                 ("roi_heads.mask_head.predictor", 0.001, lambda k: k))


def adapt_num_classes(W: Dict[str, np.ndarray], num_classes: int, seed: int = 0) -> Tuple[Dict[str, np.ndarray], List[str]]:
    """What ``DetectionCheckpointer.load`` does with a COCO (80-class) zoo checkpoint and ``ROI_HEADS.NUM_CLASSES = K``
    ([EXT fvcore: common/checkpoint.py ``_load_model``: "Skip loading parameter ... due to incompatible shapes"]): tensors whose
    shape differs from the model's are NOT loaded and keep the model's fresh initialisation -- normal(0, 0.01) class scores,
    normal(0, 0.001) box deltas and mask predictor (``nn.init.normal_``; here numpy's generator, seeded), zero biases.
    Returns (weights, names of the re-initialised layers)."""
    out = dict(W)
    redone: List[str] = []
    rng = np.random.default_rng(seed + 4242)
    for name, std, rows in _CLASS_SHAPED:
        w = out.get(name + ".weight")
        if w is None or w.shape[0] == rows(num_classes):
            continue
        out[name + ".weight"] = (rng.standard_normal((rows(num_classes),) + tuple(w.shape[1:])) * std).astype(np.float32)
        out[name + ".bias"] = np.zeros(rows(num_classes), np.float32)
        redone.append(name)
    return out, redone


def infer_num_classes(W: Dict[str, np.ndarray]) -> int:
    return int(W["roi_heads.box_predictor.cls_score.weight"].shape[0]) - 1


# ----------------------------------------------------------------------------- packing
def _fold_bn(W: Dict[str, np.ndarray], name: str, eps: float) -> Tuple[np.ndarray, np.ndarray]:
    w = W[name + ".weight"].astype(np.float32)
    g = W[name + ".norm.weight"].astype(np.float32)
    b = W[name + ".norm.bias"].astype(np.float32)
    m = W[name + ".norm.running_mean"].astype(np.float32)
    v = W[name + ".norm.running_var"].astype(np.float32)
    scale = g / np.sqrt(v + np.float32(eps))
    return w * scale[:, None, None, None], b - m * scale


def _ohwi(w: np.ndarray, cin_pad: int, dtype=np.float16) -> np.ndarray:
    """(Cout,Cin,KH,KW) fp32 -> (Cout, KH*KW*cin_pad) with K padded to K_ALIGN, fp16."""
    cout, cin, kh, kw = w.shape
    t = np.zeros((cout, kh, kw, cin_pad), np.float32)
    t[..., :cin] = w.transpose(0, 2, 3, 1)
    t = t.reshape(cout, kh * kw * cin_pad)
    kpad = (t.shape[1] + K_ALIGN - 1) // K_ALIGN * K_ALIGN
    out = np.zeros((cout, kpad), np.float32)
    out[:, : t.shape[1]] = t
    return out.astype(dtype)


def _perm_k64(w: np.ndarray, group: int = 64) -> np.ndarray:
    """K columns of a [Cout][K] GEMM weight (K % group == 0) in the order csrc/bneck_fused.hip chains GEMMs through registers:
    inside every group of ``group`` input channels (64, or 128 for the conv3 of the 128-wide stage, whose producer leaves
    group/4 consecutive channels in each lane), logical column 32 s + 8 q + j holds channel (group/4) q + 8 s + j
    (s < group/32, q < 4, j < 8) -- the channels a lane's accumulators 2s, 2s+1 of the PREVIOUS GEMM hold, in the MFMA
    B-operand's k order."""
    cout, k = w.shape
    assert k % group == 0 and group % 32 == 0
    kappa = np.arange(group)
    s, q, j = kappa // 32, (kappa // 8) % 4, kappa % 8
    src = (group // 4) * q + 8 * s + j
    return np.ascontiguousarray(w.reshape(cout, k // group, group)[:, :, src].reshape(cout, k))


def _pad_rows(w: np.ndarray, b: np.ndarray, mult: int) -> Tuple[np.ndarray, np.ndarray]:
    cout = w.shape[0]
    cp = (cout + mult - 1) // mult * mult
    if cp == cout:
        return w, b
    w2 = np.zeros((cp,) + w.shape[1:], w.dtype)
    w2[:cout] = w
    b2 = np.zeros(cp, b.dtype)
    b2[:cout] = b
    return w2, b2


def engine_tensors(spec: EngineSpec, W: Dict[str, np.ndarray], w_dtype=np.float16, fold_bn: bool = True) -> Dict[str, np.ndarray]:
    """Kernel-ready tensors keyed ``<layer>.w`` (fp16 [Cout_pad][Kpad]) / ``<layer>.b`` (fp32).
    ``fold_bn=False`` keeps the raw (unfolded) convolution weight in ``.w`` -- the layout of the training master
    weights -- while ``.b`` is still the folded FrozenBN bias."""
    T: Dict[str, np.ndarray] = {}
    for name, cin, cout, k, has_norm in conv_layers(spec):
        if name.endswith("objectness_logits") or name.endswith("anchor_deltas"):
            continue
        if has_norm:
            w, b = _fold_bn(W, name, spec.bn_eps)
            if not fold_bn:
                w = W[name + ".weight"].astype(np.float32)
        else:
            w, b = W[name + ".weight"].astype(np.float32), W[name + ".bias"].astype(np.float32)
        if name.endswith("stem.conv1"):
            wpad = np.zeros(w.shape[:3] + (STEM_KW_PAD,), np.float32)
            wpad[..., : w.shape[3]] = w
            T[name + ".w"] = _ohwi(wpad, STEM_CIN_PAD, w_dtype)
            if w_dtype == np.float16 and T[name + ".w"].shape[0] == 64 and w.shape[2] == 7:
                # the same matrix in MFMA A-fragment order for the fused stem (csrc/stem_fused.hip): fragment (kh, mi) = 64 lanes x 8
                # values, lane l = row mi*16 + (l & 15), k = kh*32 + (l >> 4)*8 .. +7 -- a linear 28 KB copy into LDS
                m = T[name + ".w"]
                lane = np.arange(64)
                frag = np.empty((7, 4, 64, 8), m.dtype)
                for kh in range(7):
                    for mi in range(4):
                        frag[kh, mi] = m[(mi * 16 + (lane & 15))[:, None], (kh * 32 + (lane >> 4) * 8)[:, None] + np.arange(8)[None, :]]
                T[name[: -len("conv1")] + "conv1f.w"] = np.ascontiguousarray(frag.reshape(7 * 4 * 64, 8))
        else:
            T[name + ".w"] = _ohwi(w, cin, w_dtype)
        T[name + ".b"] = b.astype(np.float32)
    # first block of every res stage: conv3 and the projection shortcut as one GEMM over K = [conv2 out | block input]
    # (fp16 path only; [EXT d2: modeling/backbone/resnet.py BottleneckBlock.forward] adds the two conv outputs)
    if w_dtype == np.float16:
        for name in [n for n in list(T) if n.endswith(".shortcut.w")]:
            blk = name[: -len(".shortcut.w")]
            w3, _ = _fold_bn(W, blk + ".conv3", spec.bn_eps)
            wsc, _ = _fold_bn(W, blk + ".shortcut", spec.bn_eps)
            if w3.shape[1] % 64 or wsc.shape[1] % 64:
                continue
            T[blk + ".conv3sc.w"] = np.concatenate([_ohwi(w3, w3.shape[1], w_dtype), _ohwi(wsc, wsc.shape[1], w_dtype)], 1)
            T[blk + ".conv3sc.b"] = (T[blk + ".conv3.b"] + T[blk + ".shortcut.b"]).astype(np.float32)
    # fused bottleneck tails (fp16 path, 64-wide stage = res2): identity blocks run conv2 + conv3 + the next block's conv1 in one
    # launch with the later GEMMs' K columns in the register-chaining order (csrc/bneck_fused.hip)
    if w_dtype == np.float16 and spec.num_groups * spec.width_per_group == 64 and spec.res2_out_channels == 256:
        for si, width in ((0, 64), (1, 128)):          # res2 (64 -> 256) and res3 (128 -> 512)
            for bi in range(spec.res_blocks[si]):
                blk = f"backbone.bottom_up.res{si + 2}.{bi}"
                if bi >= 1 or si == 0:                   # res2.0: beside its projection shortcut (.shortcut.w, natural K order)
                    T[blk + ".conv3p.w"] = _perm_k64(T[blk + ".conv3.w"], width)
                if bi >= (1 if si == 0 else 2):
                    T[blk + ".conv1p.w"] = _perm_k64(T[blk + ".conv1.w"], 64)
    # RPN heads fused into one 1x1 conv: rows [0,A) objectness, [A,5A) deltas (a*4+d), padded to 16
    p = "proposal_generator.rpn_head."
    w = np.concatenate([W[p + "objectness_logits.weight"], W[p + "anchor_deltas.weight"]], 0).astype(np.float32)
    b = np.concatenate([W[p + "objectness_logits.bias"], W[p + "anchor_deltas.bias"]], 0).astype(np.float32)
    wp, bp = _pad_rows(_ohwi(w, w.shape[1], w_dtype), b, 16)
    T[p + "heads.w"], T[p + "heads.b"] = wp, bp
    if w_dtype == np.float16 and wp.shape == (16, 256):
        # the same 16 x 256 matrix with its K columns in the register-chaining order: the head runs inside the epilogue of the RPN's
        # 3x3 convolution (csrc/conv_deep.hip, ConvParams::head_w)
        T[p + "headsp.w"] = _perm_k64(wp, 64)
    # box head: fc1 consumes RoI features laid out [7][7][C] (channels fastest) on the device,
    # detectron2 flattens (C,7,7): permute the K axis once here.
    r = spec.box_pooler_resolution
    c = spec.fpn_out_channels
    fc1 = W["roi_heads.box_head.fc1.weight"].astype(np.float32).reshape(-1, c, r, r).transpose(0, 2, 3, 1).reshape(-1, r * r * c)
    T["roi_heads.box_head.fc1.w"] = _ohwi(fc1[:, :, None, None], fc1.shape[1], w_dtype)
    T["roi_heads.box_head.fc1.b"] = W["roi_heads.box_head.fc1.bias"].astype(np.float32)
    fc2 = W["roi_heads.box_head.fc2.weight"].astype(np.float32)
    T["roi_heads.box_head.fc2.w"] = _ohwi(fc2[:, :, None, None], fc2.shape[1], w_dtype)
    T["roi_heads.box_head.fc2.b"] = W["roi_heads.box_head.fc2.bias"].astype(np.float32)
    # predictor: rows [0,K+1) class logits, [K+1, K+1+4K) box deltas, padded to 16
    w = np.concatenate([W["roi_heads.box_predictor.cls_score.weight"], W["roi_heads.box_predictor.bbox_pred.weight"]], 0).astype(np.float32)
    b = np.concatenate([W["roi_heads.box_predictor.cls_score.bias"], W["roi_heads.box_predictor.bbox_pred.bias"]], 0).astype(np.float32)
    wp, bp = _pad_rows(_ohwi(w[:, :, None, None], w.shape[1], w_dtype), b, 16)
    T["roi_heads.box_predictor.w"], T["roi_heads.box_predictor.b"] = wp, bp
    if spec.mask_on:
        # deconv 2x2 s2 == 4 independent 1x1 convs: row (g*Cout + co), g = dy*2+dx
        dw = W["roi_heads.mask_head.deconv.weight"].astype(np.float32)          # (Cin, Cout, 2, 2)
        g = dw.transpose(2, 3, 1, 0).reshape(4 * dw.shape[1], dw.shape[0])      # (dy,dx,co) x ci
        T["roi_heads.mask_head.deconv.w"] = _ohwi(g[:, :, None, None], g.shape[1], w_dtype)
        T["roi_heads.mask_head.deconv.b"] = np.tile(W["roi_heads.mask_head.deconv.bias"].astype(np.float32), 4)
        pw = W["roi_heads.mask_head.predictor.weight"].astype(np.float32)[:, :, 0, 0]      # (K, C)
        T["roi_heads.mask_head.predictor.w"] = pw.astype(np.float32)           # fp32: tiny, used by a VALU dot
        T["roi_heads.mask_head.predictor.b"] = W["roi_heads.mask_head.predictor.bias"].astype(np.float32)
    return T


def serialize(tensors: Dict[str, np.ndarray]) -> bytes:
    """Flat blob: header (magic,u32 version,u32 n,u32 data_offset) + n entries
    {char name[96]; u32 dtype; u32 ndim; u64 dims[4]; u64 offset; u64 nbytes} + aligned data."""
    names = sorted(tensors)
    ent_size = 96 + 4 + 4 + 32 + 8 + 8
    hdr = 16
    data_off = (hdr + ent_size * len(names) + ALIGN - 1) // ALIGN * ALIGN
    entries = []
    chunks = []
    off = data_off
    for n in names:
        a = np.ascontiguousarray(tensors[n])
        dt = {np.dtype(np.float16): DT_F16, np.dtype(np.float32): DT_F32, np.dtype(np.int32): DT_I32}[a.dtype]
        dims = list(a.shape) + [1] * (4 - a.ndim)
        nb = a.nbytes
        entries.append(struct.pack("<96sII4QQQ", n.encode(), dt, a.ndim, *dims, off, nb))
        chunks.append((off, a.tobytes()))
        off = (off + nb + ALIGN - 1) // ALIGN * ALIGN
    buf = bytearray(off)
    buf[0:16] = struct.pack("<IIII", BLOB_MAGIC, BLOB_VERSION, len(names), data_off)
    for i, e in enumerate(entries):
        buf[hdr + i * ent_size: hdr + (i + 1) * ent_size] = e
    for o, b in chunks:
        buf[o: o + len(b)] = b
    return bytes(buf)


def bn_scale(W: Dict[str, np.ndarray], name: str, eps: float) -> np.ndarray:
    """FrozenBatchNorm2d scale = weight * rsqrt(running_var + eps) of layer ``name`` (fp32)."""
    return (W[name + ".norm.weight"].astype(np.float32) / np.sqrt(W[name + ".norm.running_var"].astype(np.float32) + np.float32(eps))).astype(np.float32)


def trainable_layers(spec: EngineSpec, freeze_at: int = 2) -> List[str]:
    """Engine layer names that receive gradients (FREEZE_AT 2: stem and res2 frozen, R:58), fused heads included."""
    L = []
    for name, _, _, _, _ in conv_layers(spec):
        if name.endswith("objectness_logits") or name.endswith("anchor_deltas"):
            continue
        if ".stem." in name and freeze_at >= 1:
            continue
        if ".res2." in name and freeze_at >= 2:
            continue
        L.append(name)
    L += ["proposal_generator.rpn_head.heads", "roi_heads.box_head.fc1", "roi_heads.box_head.fc2", "roi_heads.box_predictor"]
    if spec.mask_on:
        L += ["roi_heads.mask_head.deconv", "roi_heads.mask_head.predictor16"]
    return L


def train_tensors(spec: EngineSpec, W: Dict[str, np.ndarray], freeze_at: int = 2) -> Dict[str, np.ndarray]:
    """Extra blob entries of the training engine: per trainable layer the fp32 master weight ``<layer>.m32`` in the
    forward GEMM layout (UNFOLDED: FrozenBN has no parameters, the trainable tensor is the raw convolution weight) and,
    for FrozenBN layers, the per-channel scale ``<layer>.s`` that the forward fold and the weight gradient apply."""
    raw = engine_tensors(spec, W, w_dtype=np.float32, fold_bn=False)
    T: Dict[str, np.ndarray] = {}
    has_norm = {n: hn for n, _, _, _, hn in conv_layers(spec)}
    for name in trainable_layers(spec, freeze_at):
        if name == "roi_heads.mask_head.predictor16":
            # the mask predictor as a 16-row GEMM operand (forward uses a VALU dot on the predicted class only; training
            # needs all classes' logits and the layer's gradients): rows [0,K) = classes, padded to 16
            pw = W["roi_heads.mask_head.predictor.weight"].astype(np.float32)[:, :, 0, 0]
            wp, bp = _pad_rows(_ohwi(pw[:, :, None, None], pw.shape[1], np.float32), W["roi_heads.mask_head.predictor.bias"].astype(np.float32), 16)
            T[name + ".m32"], T[name + ".w"], T[name + ".b"] = wp, wp.astype(np.float16), bp
            continue
        T[name + ".m32"] = raw[name + ".w"].astype(np.float32)
        if name == "roi_heads.mask_head.deconv":
            # trainable layout [Cin][(dy,dx,co)] = transpose of the forward GEMM weight: the weight-gradient kernel produces this
            # orientation directly and the fold writes the forward operand as its transposed copy
            T[name + ".m32T"] = np.ascontiguousarray(T[name + ".m32"].T)
            T[name + ".b256"] = W["roi_heads.mask_head.deconv.bias"].astype(np.float32)
        if has_norm.get(name, False):
            T[name + ".s"] = bn_scale(W, name, spec.bn_eps)
    return T


def master_to_d2(spec: EngineSpec, base: Dict[str, np.ndarray], fetch) -> Dict[str, np.ndarray]:
    """Inverse of ``train_tensors``: ``fetch("m:<layer>.w" | "m:<layer>.b")`` -> detectron2-keyed tensors (the layout
    ``DetectionCheckpointer`` saves and ``load_checkpoint`` reads), on top of a copy of ``base`` for the frozen tensors."""
    out = {k: np.array(v, copy=True) for k, v in base.items()}
    shapes = {n: (cin, cout, k) for n, cin, cout, k, _ in conv_layers(spec)}
    has_norm = {n: hn for n, _, _, _, hn in conv_layers(spec)}
    A, K = spec.num_anchors, spec.num_classes
    for name in trainable_layers(spec):
        m = fetch(f"m:{name}.w")
        if name in shapes:
            cin, cout, k = shapes[name]
            out[name + ".weight"] = m[:cout, : k * k * cin].reshape(cout, k, k, cin).transpose(0, 3, 1, 2).astype(np.float32)
            if not has_norm[name]:
                out[name + ".bias"] = fetch(f"m:{name}.b")[:cout].astype(np.float32)
        elif name == "proposal_generator.rpn_head.heads":
            b = fetch(f"m:{name}.b")
            p = "proposal_generator.rpn_head."
            c = m.shape[1]
            out[p + "objectness_logits.weight"] = m[:A].reshape(A, c, 1, 1).astype(np.float32)
            out[p + "anchor_deltas.weight"] = m[A:5 * A].reshape(4 * A, c, 1, 1).astype(np.float32)
            out[p + "objectness_logits.bias"], out[p + "anchor_deltas.bias"] = b[:A].astype(np.float32), b[A:5 * A].astype(np.float32)
        elif name == "roi_heads.box_head.fc1":
            r, c = spec.box_pooler_resolution, spec.fpn_out_channels
            out[name + ".weight"] = m.reshape(-1, r, r, c).transpose(0, 3, 1, 2).reshape(m.shape[0], -1).astype(np.float32)
            out[name + ".bias"] = fetch(f"m:{name}.b").astype(np.float32)
        elif name == "roi_heads.box_head.fc2":
            out[name + ".weight"], out[name + ".bias"] = m.astype(np.float32), fetch(f"m:{name}.b").astype(np.float32)
        elif name == "roi_heads.box_predictor":
            b = fetch(f"m:{name}.b")
            out[name + ".cls_score.weight"], out[name + ".bbox_pred.weight"] = m[:K + 1].astype(np.float32), m[K + 1:5 * K + 1].astype(np.float32)
            out[name + ".cls_score.bias"], out[name + ".bbox_pred.bias"] = b[:K + 1].astype(np.float32), b[K + 1:5 * K + 1].astype(np.float32)
        elif name == "roi_heads.mask_head.deconv":
            cin = m.shape[0]
            out[name + ".weight"] = m.reshape(cin, 2, 2, -1).transpose(0, 3, 1, 2).astype(np.float32)      # [ci][(dy,dx,co)] -> (Cin,Cout,2,2)
            out[name + ".bias"] = fetch(f"m:{name}.b").astype(np.float32)
        elif name == "roi_heads.mask_head.predictor16":
            b = fetch(f"m:{name}.b")
            out["roi_heads.mask_head.predictor.weight"] = m[:K].reshape(K, -1, 1, 1).astype(np.float32)
            out["roi_heads.mask_head.predictor.bias"] = b[:K].astype(np.float32)
    return out


def pack_weights(spec: EngineSpec, W: Dict[str, np.ndarray], train: bool = False) -> bytes:
    return serialize(packed_tensors(spec, W, train))


def packed_tensors(spec: EngineSpec, W: Dict[str, np.ndarray], train: bool = False) -> Dict[str, np.ndarray]:
    """Every tensor of the blob ``pack_weights`` serialises, by name (the precision modes add theirs to the fp16 set)."""
    T = engine_tensors(spec, W)
    if train:
        T.update(train_tensors(spec, W))
    if spec.precision == "fp32":
        # fp32 validation mode: same layout, GEMM weights additionally kept in fp32 ("<layer>.w32")
        T32 = engine_tensors(spec, W, w_dtype=np.float32)
        for k, v in T32.items():
            if k.endswith(".w") and k != "roi_heads.mask_head.predictor.w":
                T[k + "32"] = v
        if train and "roi_heads.mask_head.predictor16.m32" in T:      # the reference-precision trainer's 16-row mask predictor operand
            T["roi_heads.mask_head.predictor16.w32"] = T["roi_heads.mask_head.predictor16.m32"]
    if spec.precision == "split":
        # split-operand mode (csrc/common.h ConvParams::split): every GEMM weight as hi + lo fp16 planes of the row-scaled fp32 weight
        T32 = engine_tensors(spec, W, w_dtype=np.float32)
        for k, v in T32.items():
            if k.endswith(".w") and k != "roi_heads.mask_head.predictor.w":
                T[k + "s"], T[k + "si"] = split_planes(v)
        # the stem matrix in MFMA A-fragment order for the fused stem (csrc/stem_fused.hip stem_pool_split_kernel): hi plane [7][4][64][8], then lo plane;
        # the row scales are those of ".stem.conv1.wsi"
        sn = "backbone.bottom_up.stem.conv1"
        if sn + ".ws" in T and T[sn + ".ws"].shape == (128, 256):
            lane = np.arange(64)
            frag = np.empty((2, 7, 4, 64, 8), np.float16)
            for pl in range(2):
                m = T[sn + ".ws"][pl * 64:(pl + 1) * 64]
                for kh in range(7):
                    for mi in range(4):
                        frag[pl, kh, mi] = m[(mi * 16 + (lane & 15))[:, None], (kh * 32 + (lane >> 4) * 8)[:, None] + np.arange(8)[None, :]]
            T["backbone.bottom_up.stem.conv1f.ws"] = np.ascontiguousarray(frag.reshape(2 * 7 * 4 * 64, 8))
        # the RPN's 16 x 256 head with its K columns in the register-chaining order: it runs inside the epilogue of the RPN's 3x3 convolution
        hp = "proposal_generator.rpn_head."
        if T32[hp + "heads.w"].shape == (16, 256):
            T[hp + "headsp.ws"], T[hp + "headsp.wsi"] = split_planes(_perm_k64(T32[hp + "heads.w"], 64))
        # first block of every res stage: conv3 and the projection shortcut as one GEMM over K = [conv2 out | block input] (engine_tensors
        # builds this operand for the fp16 path only)
        for name in [n for n in T32 if n.endswith(".shortcut.w")]:
            blk = name[: -len(".shortcut.w")]
            w3, wsc = T32[blk + ".conv3.w"], T32[blk + ".shortcut.w"]
            if w3.shape[1] % 64 == 0 and wsc.shape[1] % 64 == 0 and blk + ".conv3sc.b" in T:
                T[blk + ".conv3sc.ws"], T[blk + ".conv3sc.wsi"] = split_planes(np.concatenate([w3, wsc], 1))
        # fused bottleneck tails of the identity-shortcut blocks of res2 / res3 (csrc/bneck_split.hip): conv3 and the next block's conv1 with their
        # K columns in the register-chaining order.  The row scale does not depend on the column order, so ".conv3p.wsi" equals ".conv3.wsi".
        if spec.num_groups * spec.width_per_group == 64 and spec.res2_out_channels == 256:
            for si, width in ((0, 64), (1, 128)):
                for bi in range(1, spec.res_blocks[si]):
                    blk = f"backbone.bottom_up.res{si + 2}.{bi}"
                    T[blk + ".conv3p.ws"], T[blk + ".conv3p.wsi"] = split_planes(_perm_k64(T32[blk + ".conv3.w"], width))
                    if bi >= (1 if si == 0 else 2):
                        T[blk + ".conv1p.ws"], T[blk + ".conv1p.wsi"] = split_planes(_perm_k64(T32[blk + ".conv1.w"], 64))
            # res2.0: conv3 (chained K order) and its projection shortcut from the 64-channel stem output (natural K order) as ONE operand
            # [256][128] under one scale per row -- the projection form of the same kernel (bias: ".conv3sc.b")
            blk = "backbone.bottom_up.res2.0"
            if T32[blk + ".shortcut.w"].shape == (256, 64) and blk + ".conv3sc.b" in T:
                T[blk + ".conv3scp.ws"], T[blk + ".conv3scp.wsi"] = split_planes(
                    np.concatenate([_perm_k64(T32[blk + ".conv3.w"], 64), T32[blk + ".shortcut.w"]], 1))
    return T


def split_planes(w32: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """fp32 GEMM weight [rows][K] -> (fp16 [2 * rows][K]: hi rows then lo rows of ``w * 2**e(row)``, fp32 [rows]: ``2**-e(row)``).
    The power of two puts the row's largest magnitude in [2**13, 2**14): the lo parts (<= 2**-11 of their hi) then stay normal fp16
    numbers for every element down to 2**-17 of the row maximum, and hi + lo carries 22 significand bits of the scaled weight."""
    w32 = np.ascontiguousarray(w32, np.float32)
    mx = np.abs(w32).max(axis=1)
    e = np.where(mx > 0, 13 - np.floor(np.log2(np.maximum(mx, np.float32(1e-38)))), 0.0)
    e = np.clip(e, -100, 100)
    scale = np.exp2(e).astype(np.float32)
    ws = (w32.astype(np.float64) * scale[:, None].astype(np.float64)).astype(np.float32)        # exact: a power of two
    hi = ws.astype(np.float16)
    lo = (ws - hi.astype(np.float32)).astype(np.float16)
    return np.concatenate([hi, lo], 0), (1.0 / scale.astype(np.float64)).astype(np.float32)
