"""CPU-side checks of the product's host code and of the C ABI surface (no GPU compute)."""
import ctypes as C
import os
import re
import struct

import numpy as np
import pytest

from oracle import maskrcnn_oracle as O
from proj_roadsurf_amd import weights as Wt
from proj_roadsurf_amd.engine import LIB_PATH, RsSpec, cell_anchor_table, load_library, make_rs_spec
from proj_roadsurf_amd.spec import EngineSpec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return load_library()


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "rs_engine.h")).read()
    names = set(re.findall(r"\b(rs_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 20
    for n in sorted(names):
        assert hasattr(lib, n), f"librs_engine.so does not export {n}"
    assert lib.rs_abi_version() == 1


def test_rs_spec_layout_matches_header(lib):
    # the C side rejects a struct_size mismatch; here we at least pin the Python mirror's size
    s = make_rs_spec(EngineSpec(num_classes=2))
    assert s.struct_size == C.sizeof(RsSpec)
    assert s.num_anchors == 3 and s.num_levels == 5 and s.flip_channels == 1
    assert abs(s.pixel_mean[0] - 103.53) < 1e-5 and abs(s.scale_clamp - 4.1351666) < 1e-6


def test_cell_anchor_table_equals_oracle():
    spec = EngineSpec()
    t = cell_anchor_table(spec)
    for l in range(5):
        assert np.array_equal(t[l], O.cell_anchors(spec.anchor_sizes[l], spec.anchor_aspect_ratios).numpy())


def test_resize_shape_c_equals_python(lib):
    from proj_roadsurf_amd.spec import resize_shortest_edge_shape
    for h, w in [(512, 512), (1024, 1024), (256, 256), (600, 900), (480, 1000), (333, 777), (1000, 480)]:
        a, b = C.c_int32(), C.c_int32()
        lib.rs_resize_shape(h, w, 800, 1333, C.byref(a), C.byref(b))
        assert (a.value, b.value) == resize_shortest_edge_shape(h, w, 800, 1333)


@pytest.mark.parametrize("insz,outsz", [(512, 800), (256, 800), (1024, 800), (96, 128), (600, 800), (37, 91)])
def test_resize_coeffs_c_equals_oracle(lib, insz, outsz):
    ks = lib.rs_resize_coeffs(insz, outsz, None, None)
    b = np.zeros((outsz, 2), np.int32)
    k = np.zeros((outsz, ks), np.int32)
    assert lib.rs_resize_coeffs(insz, outsz, b.ctypes.data_as(C.POINTER(C.c_int32)), k.ctypes.data_as(C.POINTER(C.c_int32))) == ks
    ob, ok, oks = O._pil_bilinear_coeffs(insz, outsz)
    assert oks == ks and np.array_equal(b, ob) and np.array_equal(k, ok)


@pytest.mark.parametrize("shape,new", [((64, 48), (100, 75)), ((128, 128), (200, 200)), ((160, 200), (128, 160)), ((50, 50), (50, 80))])
def test_restated_pil_resize_is_bit_exact(shape, new):
    """The fixed-point 2-pass algorithm (what the HIP preprocess kernel runs) == Pillow, bit for bit,
    for up-scaling and (antialiased) down-scaling."""
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, shape + (3,), dtype=np.uint8)
    ref = O.pil_resize(img, new[0], new[1])
    got = O.pil_resize_restated(img, new[0], new[1])
    assert np.array_equal(ref, got)


def test_conv_layer_list_and_param_count():
    spec = EngineSpec(num_classes=2)
    W = Wt.synthetic_weights(spec, 0)
    layers = Wt.conv_layers(spec)
    assert len([l for l in layers if "bottom_up" in l[0]]) == 53          # 53 backbone convs
    n_params = sum(v.size for k, v in W.items() if "running" not in k and ".norm." not in k)
    assert abs(n_params - 43.9e6) < 0.2e6                                  # SURVEY §8a parameter count (K=2)
    # detectron2 key layout
    for k in ["backbone.bottom_up.stem.conv1.weight", "backbone.bottom_up.res2.0.shortcut.norm.running_var", "backbone.fpn_lateral5.bias",
              "proposal_generator.rpn_head.anchor_deltas.weight", "roi_heads.box_head.fc1.weight", "roi_heads.box_predictor.bbox_pred.bias",
              "roi_heads.mask_head.deconv.weight", "roi_heads.mask_head.predictor.weight"]:
        assert k in W


def test_pack_weights_blob_roundtrip_and_bn_fold():
    spec = EngineSpec(num_classes=2)
    W = Wt.synthetic_weights(spec, 0)
    T = Wt.engine_tensors(spec, W)
    blob = Wt.serialize(T)
    magic, ver, n, off = struct.unpack("<IIII", blob[:16])
    assert magic == Wt.BLOB_MAGIC and ver == 1 and n == len(T) and off % 256 == 0
    # parse back one entry
    ent = 96 + 4 + 4 + 32 + 8 + 8
    names = sorted(T)
    i = names.index("backbone.bottom_up.res2.0.conv2.w")
    e = blob[16 + ent * i: 16 + ent * (i + 1)]
    name = e[:96].rstrip(b"\0").decode()
    dt, nd, d0, d1, d2, d3, o, nb = struct.unpack("<II4QQQ", e[96:])
    assert name == names[i] and dt == Wt.DT_F16 and (d0, d1) == (64, 576) and nb == 64 * 576 * 2 and o % 256 == 0
    back = np.frombuffer(blob[o:o + nb], np.float16).reshape(64, 576)
    # BN fold: w' = w * gamma / sqrt(var + eps), layout (Cout, kh, kw, cin)
    p = "backbone.bottom_up.res2.0.conv2"
    scale = W[p + ".norm.weight"] / np.sqrt(W[p + ".norm.running_var"] + np.float32(1e-5))
    want = (W[p + ".weight"] * scale[:, None, None, None]).transpose(0, 2, 3, 1).reshape(64, 576).astype(np.float16)
    assert np.array_equal(back, want)
    bias = T[p + ".b"]
    assert np.allclose(bias, W[p + ".norm.bias"] - W[p + ".norm.running_mean"] * scale)
    # stem: Cin padded 3 -> 4, tap rows 7 -> 8 (two taps per 16-byte chunk): K = 7*8*4 = 224 -> 256
    assert T["backbone.bottom_up.stem.conv1.w"].shape == (64, 256)
    st = T["backbone.bottom_up.stem.conv1.w"].astype(np.float32).reshape(64, 8, 8, 4)[:, :7]
    assert not st[:, :, 7].any() and not st[..., 3].any()
    # fc1 K-axis permuted to (h, w, c)
    fc1 = T["roi_heads.box_head.fc1.w"].astype(np.float32)
    src = W["roi_heads.box_head.fc1.weight"].reshape(1024, 256, 7, 7)
    assert np.allclose(fc1[5, (3 * 7 + 2) * 256 + 17], src[5, 17, 3, 2], atol=1e-3)
    # fused heads padded to 16 rows
    assert T["proposal_generator.rpn_head.heads.w"].shape == (16, 256)
    assert T["roi_heads.box_predictor.w"].shape == (16, 1024)
    assert T["roi_heads.mask_head.deconv.w"].shape == (1024, 256)


def test_split_mode_weight_planes():
    """The split-operand mode's operands (weights.packed_tensors, precision "split"): hi + lo planes of the row-scaled fp32 weight carry it to 2^-21 of the
    row maximum; the chained-order and fragment-order copies the fused kernels read (csrc/bneck_split.hip, stem_fused.hip) are permutations of the plain ones
    under the SAME row scales."""
    spec = EngineSpec(num_classes=2, precision="split")
    W = Wt.synthetic_weights(spec, 0)
    T = Wt.packed_tensors(spec, W)
    T32 = Wt.engine_tensors(spec, W, w_dtype=np.float32)
    n = 0
    for k, w32 in T32.items():
        if not k.endswith(".w") or k + "s" not in T:
            continue
        ws, wsi = T[k + "s"], T[k + "si"]
        rows = w32.shape[0]
        assert ws.dtype == np.float16 and ws.shape == (2 * rows, w32.shape[1]) and wsi.shape == (rows,)
        assert np.all(np.exp2(np.round(np.log2(wsi))) == wsi), k                      # powers of two: the epilogue's multiply is exact
        back = (ws[:rows].astype(np.float64) + ws[rows:].astype(np.float64)) * wsi[:, None].astype(np.float64)
        bound = 2.0 ** -21 * np.abs(w32).max(axis=1, keepdims=True) + 1e-30
        assert np.all(np.abs(back - w32) <= bound), k
        hi_max = np.abs(ws[:rows].astype(np.float32)).max(axis=1)
        assert np.all((hi_max == 0) | ((hi_max >= 2.0 ** 13) & (hi_max < 2.0 ** 14 + 8))), k      # the scale puts the row maximum in [2^13, 2^14)
        n += 1
    assert n >= 60
    bu = "backbone.bottom_up."
    # chained K order = a column permutation, row by row; the row scale does not see the order
    for blk, group in ((bu + "res2.1", 64), (bu + "res2.2", 64), (bu + "res3.1", 128), (bu + "res3.3", 128)):
        assert np.array_equal(T[blk + ".conv3p.ws"], Wt._perm_k64(T[blk + ".conv3.ws"], group))
        assert np.array_equal(T[blk + ".conv3p.wsi"], T[blk + ".conv3.wsi"])
    for blk in (bu + "res2.1", bu + "res2.2", bu + "res3.2", bu + "res3.3"):
        assert np.array_equal(T[blk + ".conv1p.ws"], Wt._perm_k64(T[blk + ".conv1.ws"], 64))
        assert np.array_equal(T[blk + ".conv1p.wsi"], T[blk + ".conv1.wsi"])
    assert bu + "res3.1.conv1p.ws" not in T and bu + "res3.0.conv3p.ws" not in T          # res3.0 has no fused tail, so nothing chains into res3.1's conv1
    # projection form of res2.0: [conv3 chained | shortcut natural] under the scales of the dual-source operand
    sc, scp = T[bu + "res2.0.conv3sc.ws"], T[bu + "res2.0.conv3scp.ws"]
    assert scp.shape == (512, 128) and np.array_equal(scp[:, 64:], sc[:, 64:]) and np.array_equal(scp[:, :64], Wt._perm_k64(sc[:, :64], 64))
    assert np.array_equal(T[bu + "res2.0.conv3scp.wsi"], T[bu + "res2.0.conv3sc.wsi"])
    # fused stem: fragment (plane, kh, block) = 64 lanes x 8 halfs, lane l = row block*16 + (l & 15), k = kh*32 + (l >> 4)*8 ..
    sf, sw = T[bu + "stem.conv1f.ws"].reshape(2, 7, 4, 64, 8), T[bu + "stem.conv1.ws"]
    for pl, kh, mi, lane in ((0, 0, 0, 0), (1, 6, 3, 63), (0, 3, 2, 37), (1, 2, 1, 20)):
        assert np.array_equal(sf[pl, kh, mi, lane], sw[pl * 64 + mi * 16 + (lane & 15), kh * 32 + (lane >> 4) * 8: kh * 32 + (lane >> 4) * 8 + 8])


def test_no_compiler_division_sequence_in_device_code(tmp_path):
    """No code object of librs_engine.so contains v_div_scale_f32 / v_div_fmas_f32 / v_div_fixup_f32: the compiler's expansion of fp32 `/` returns wrong
    quotients in a wave that shares its SIMD with another kernel's MFMA waves (tools/ubench/coexec_probe.hip), so device code divides through
    csrc/common.h rs_fdiv.  A `/` on floats slipping back into a kernel shows up here, on the CPU, before it shows up as a wrong mask word once in 500 tiles."""
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "proj_roadsurf_amd", "librs_engine.so")
    if not os.path.exists(objdump) or not os.path.exists(lib):
        pytest.skip("llvm-objdump or the built library is not here")
    shutil.copy(lib, tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", "lib.so"], cwd=tmp_path, check=True, capture_output=True)
    cos = [f for f in os.listdir(tmp_path) if "gfx950" in f]
    assert len(cos) >= 10, cos
    found = {}
    n_mfma = 0
    for f in cos:
        asm = subprocess.run([objdump, "-d", f], cwd=tmp_path, check=True, capture_output=True, text=True).stdout
        n_mfma += asm.count("v_mfma_")
        k = sum(asm.count(ins) for ins in ("v_div_scale_f32", "v_div_fmas_f32", "v_div_fixup_f32"))
        if k:
            found[f] = k
    assert n_mfma > 1000            # the disassembly is that of the kernels
    assert not found, f"compiler fp32 division sequence in device code (use rs_fdiv): {found}"


def test_checkpoint_loader_pth_weights_only(tmp_path):
    import torch
    W = {"roi_heads.box_predictor.cls_score.weight": torch.zeros(3, 1024)}
    p = tmp_path / "m.pth"
    torch.save({"model": W, "iteration": 5}, str(p))
    back = Wt.load_checkpoint(str(p))
    assert Wt.infer_num_classes(back) == 2


def test_engine_fails_loudly_without_gpu(lib):
    """No CPU fallback: on a box without a HIP device engine creation must error, not degrade."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from proj_roadsurf_amd.engine import Engine, RsError
    spec = EngineSpec(num_classes=2)
    with pytest.raises(RsError):
        Engine(spec, Wt.synthetic_weights(spec, 0), (64, 64, 3), max_batch=1)


def test_zoo_pkl_reader_accepts_arrays_only(tmp_path):
    """R:config/detectron2_config_3bands.yaml:265 (model-zoo .pkl): a dict-of-ndarrays pickle loads; a pickle that references
    any other global is refused before anything of it runs."""
    import pickle

    from proj_roadsurf_amd.weights import load_checkpoint, load_zoo_pkl

    good = {"model": {"backbone.bottom_up.stem.conv1.weight": np.arange(24, dtype=np.float32).reshape(2, 3, 2, 2),
                      "roi_heads.box_predictor.cls_score.weight": np.ones((3, 4), np.float64)},
            "__author__": "Detectron2 Model Zoo"}
    for proto in (2, 4):
        p = tmp_path / f"good{proto}.pkl"
        p.write_bytes(pickle.dumps(good, protocol=proto))
        W = load_checkpoint(str(p))
        assert set(W) == set(good["model"]) and all(v.dtype == np.float32 for v in W.values())
        assert np.array_equal(W["backbone.bottom_up.stem.conv1.weight"], good["model"]["backbone.bottom_up.stem.conv1.weight"])

    marker = tmp_path / "executed"

    class Evil:
        def __reduce__(self):
            import os
            return (os.system, (f"touch {marker}",))

    for payload in ({"model": {"backbone.bottom_up.x": Evil()}}, Evil(), {"model": {"backbone.bottom_up.x": np.array([Evil()], dtype=object)}}):
        p = tmp_path / "evil.pkl"
        p.write_bytes(pickle.dumps(payload))
        with pytest.raises(pickle.UnpicklingError):
            load_zoo_pkl(str(p))
        assert not marker.exists()
    # not detectron2 key names
    p = tmp_path / "c2.pkl"
    p.write_bytes(pickle.dumps({"blobs": {"conv1_w": np.zeros(3, np.float32)}}))
    with pytest.raises(ValueError):
        load_zoo_pkl(str(p))


def test_zoo_checkpoint_resolves_in_detectron2_cache_layout(tmp_path, monkeypatch):
    """The reference's unchanged YAML names a model-zoo config (R:config/config_obj_detec.yaml:71-72); detectron2 would download
    https://dl.fbaipublicfiles.com/detectron2/<name>/<id>/model_final_<hash>.pkl into iopath's cache, at the URL's path under the
    cache root.  Offline the file is looked up there ($FVCORE_CACHE or ~/.torch/iopath_cache), read by the restricted unpickler, and
    the class-shaped layers of the 80-class COCO heads are re-initialised for K classes as DetectionCheckpointer leaves them."""
    import pickle

    from proj_roadsurf_amd.weights import adapt_num_classes, load_checkpoint, resolve_zoo_checkpoint

    name = "COCO-InstanceSegmentation/mask_rcnn_R_50_FPN_1x.yaml"
    monkeypatch.setenv("FVCORE_CACHE", str(tmp_path / "cache"))
    monkeypatch.setenv("HOME", str(tmp_path / "home"))
    assert resolve_zoo_checkpoint(name) is None
    d = tmp_path / "cache" / "detectron2" / "COCO-InstanceSegmentation" / "mask_rcnn_R_50_FPN_1x" / "137260431"
    d.mkdir(parents=True)
    model = {"backbone.bottom_up.stem.conv1.weight": np.ones((64, 3, 7, 7), np.float32),
             "roi_heads.box_predictor.cls_score.weight": np.ones((81, 1024), np.float32), "roi_heads.box_predictor.cls_score.bias": np.ones(81, np.float32),
             "roi_heads.box_predictor.bbox_pred.weight": np.ones((320, 1024), np.float32), "roi_heads.box_predictor.bbox_pred.bias": np.ones(320, np.float32),
             "roi_heads.mask_head.predictor.weight": np.ones((80, 256, 1, 1), np.float32), "roi_heads.mask_head.predictor.bias": np.ones(80, np.float32)}
    (d / "model_final_a54504.pkl").write_bytes(pickle.dumps({"model": model, "__author__": "Detectron2 Model Zoo"}, protocol=2))
    path = resolve_zoo_checkpoint(name)
    assert path == str(d / "model_final_a54504.pkl")
    assert resolve_zoo_checkpoint("https://dl.fbaipublicfiles.com/detectron2/COCO-InstanceSegmentation/mask_rcnn_R_50_FPN_1x/137260431/model_final_a54504.pkl") == path
    # the older default location, and a zoo entry that is not in the table (found by its directory)
    monkeypatch.delenv("FVCORE_CACHE")
    d2 = tmp_path / "home" / ".torch" / "iopath_cache" / "detectron2" / "COCO-Detection" / "faster_rcnn_R_50_FPN_1x" / "137257794"
    d2.mkdir(parents=True)
    (d2 / "model_final_b275ba.pkl").write_bytes(b"x")
    assert resolve_zoo_checkpoint("COCO-Detection/faster_rcnn_R_50_FPN_1x.yaml") == str(d2 / "model_final_b275ba.pkl")
    W, redone = adapt_num_classes(load_checkpoint(path), 2, seed=0)
    assert sorted(redone) == ["roi_heads.box_predictor.bbox_pred", "roi_heads.box_predictor.cls_score", "roi_heads.mask_head.predictor"]
    assert W["roi_heads.box_predictor.cls_score.weight"].shape == (3, 1024) and W["roi_heads.box_predictor.bbox_pred.weight"].shape == (8, 1024)
    assert W["roi_heads.mask_head.predictor.weight"].shape == (2, 256, 1, 1) and not W["roi_heads.mask_head.predictor.bias"].any()
    assert 0.008 < W["roi_heads.box_predictor.cls_score.weight"].std() < 0.012 and 0.0008 < W["roi_heads.box_predictor.bbox_pred.weight"].std() < 0.0012
    assert np.array_equal(W["backbone.bottom_up.stem.conv1.weight"], model["backbone.bottom_up.stem.conv1.weight"])
    same, none = adapt_num_classes(W, 2)
    assert none == [] and same["roi_heads.box_predictor.cls_score.weight"] is W["roi_heads.box_predictor.cls_score.weight"]


def test_conv_variant_table(lib):
    """The conv tile dispatch depends on M = batch * pixels, i.e. on the batch size (csrc/conv_igemm.hip
    conv_choose_variant).  Enumerate it per layer at batch 1 / 3 / 8 / 16 of the 800x800 network input and compare with
    the committed table, so that a change of the rule -- and which benchmarked layers it moves -- is visible in review.
    The GPU side (tests/test_gpu_engine.py::test_engine_launches_the_tabled_variants) checks that the engine really
    launches these."""
    import json

    from tests.util import conv_stage_shapes
    lib.rs_op_conv_variant.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_int)]
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "conv_variants.json")))
    shapes = conv_stage_shapes(EngineSpec(num_classes=2), *gold["net_input"])
    assert [s[0] for s in shapes] == list(gold["variants"])
    for name, mpi, cin, k, cout, cin2, forced in shapes:
        got = [forced if forced is not None else lib.rs_op_conv_variant(b * mpi, cin, k, cout, cin2, 0, 0, None) for b in gold["batches"]]
        assert got == gold["variants"][name], f"{name}: dispatch {got} != committed {gold['variants'][name]}"
    # the benchmarked batch (16) runs the deep 256x256 kernel on exactly these layers
    deep = [n for n, v in gold["variants"].items() if v[gold["batches"].index(16)] == 12]
    assert deep == ["res5.1.conv3", "res5.2.conv3", "fpn_lateral3", "fpn_output2-5", "rpn.conv+heads2-6",
                    "box.fc1", "box.fc2", "mask.fcn1", "mask.fcn2", "mask.fcn3", "mask.fcn4"]


def test_bench_spawns_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` without a launcher starts N ranks through torch.distributed.run on 127.0.0.1 before it
    touches the GPU."""
    import subprocess
    import sys

    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7"])
    assert bench.spawn_ranks(4) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "7"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_matching_counts_a_pair_only_when_box_score_and_mask_all_agree():
    """SURVEY section 8d's matching as the parity tests and bench.py count it (proj_roadsurf_amd/matching.py): a reference detection
    is matched when a detection of its class has box IoU >= 0.95, |dscore| <= 0.02 and mask IoU >= 0.95; a pair that fails the score
    or the mask condition still pairs up (n_matched) but is not counted (n_full); detections under score 0.1 are not asked for."""
    from proj_roadsurf_amd.matching import match_detections, wilson_lower
    def mask(x0, x1):
        m = np.zeros((4, 32, 32), bool)[0]
        m[8:24, x0:x1] = True
        return m
    boxes = np.array([[0, 0, 100, 100], [200, 0, 300, 100], [0, 200, 100, 300], [200, 200, 300, 300], [400, 400, 420, 420]], np.float32)
    ref = {"boxes": boxes, "scores": np.array([0.9, 0.8, 0.7, 0.6, 0.05], np.float32), "classes": np.array([0, 0, 1, 1, 0]),
           "masks": np.stack([mask(4, 24)] * 5)}
    got = {"boxes": boxes + np.array([[0, 0, 1, 1]], np.float32), "scores": np.array([0.9, 0.77, 0.7, 0.6, 0.05], np.float32),
           "classes": np.array([0, 0, 1, 0, 0]), "masks": np.stack([mask(4, 24), mask(4, 24), mask(4, 22), mask(4, 24), mask(4, 24)])}
    r = match_detections(ref, got)
    # 4 asked for; #0 full; #1 pairs, score off by 0.03; #2 pairs, mask IoU 0.9; #3 other class: no pair; #4 under the score bar
    assert (r["n_ref"], r["n_matched"], r["n_full"]) == (4, 3, 1)
    assert r["frac_matched"] == 0.75 and abs(r["max_dscore"] - 0.03) < 1e-6 and abs(r["min_mask_iou"] - 0.9) < 1e-9
    assert match_detections(ref, ref)["n_full"] == 4
    assert 0.979 < wilson_lower(1825, 1833) < 1825 / 1833 and wilson_lower(0, 0) == 0.0


def test_host_pools_shrink_with_the_ranks_share_of_the_cores():
    """make_detections sizes its decode processes / host threads / vectoriser threads per RANK: 4 each where the rank has >= 16 cores
    (what one MI355X was measured to need), a quarter of the share below that, never under 1; explicit values are kept."""
    from proj_roadsurf_amd.make_detections import host_pool_sizes
    assert host_pool_sizes(None, None, None, cores=128, local_world=8) == (4, 4, 4, 16)
    assert host_pool_sizes(None, None, None, cores=64, local_world=8) == (2, 2, 2, 8)
    assert host_pool_sizes(None, None, None, cores=16, local_world=8) == (1, 1, 1, 2)
    assert host_pool_sizes(None, None, None, cores=8, local_world=1) == (2, 2, 2, 8)
    assert host_pool_sizes(6, 0, None, cores=8, local_world=2) == (6, 0, 1, 4)
