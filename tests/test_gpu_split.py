"""GPU parity of the split-operand precision mode (``EngineSpec(precision="split")`` -> ``rs_spec.precision = 2``; csrc/common.h
``ConvParams::split``): every GEMM operand as hi + lo fp16 planes, three MFMA products into one fp32
accumulator.  The reference computes in fp32 (no ``SOLVER.AMP`` key, R:config/detectron2_config_3bands.yaml:268-305); this mode is
held to the SAME bounds as the fp32-MFMA mode: operator outputs against float64 to a few fp32 ulps of the terms' magnitude, the
engine end to end at SURVEY 8d's "fp32 validation" bar (tests/test_gpu_engine.py::_strict_compare).  Everything goes through the C ABI."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from proj_roadsurf_amd.engine import Engine, load_library, _check
from proj_roadsurf_amd.spec import EngineSpec
from proj_roadsurf_amd.weights import _ohwi, split_planes, synthetic_weights
from tests.util import synthetic_tiles

pytestmark = pytest.mark.gpu


def _planes(x32: torch.Tensor) -> torch.Tensor:
    """fp32 tensor -> [2, ...] fp16 planes (hi, lo)."""
    hi = x32.half()
    lo = (x32 - hi.float()).half()
    return torch.stack([hi, lo]).contiguous()


def _halo(x_nhwc: torch.Tensor, pad: int) -> torch.Tensor:
    n, h, w, c = x_nhwc.shape
    out = torch.zeros((n, h + 2 * pad, w + 2 * pad, c), dtype=x_nhwc.dtype)
    out[:, pad:pad + h, pad:pad + w] = x_nhwc
    return out


def run_split_conv(x, w, b, *, stride=1, pad=0, relu=False, res=None, up=None, in_halo=None, out_halo=1, out_f32=False, deconv=False,
                   variant=-1, cin_pad=None):
    """x (N,C,H,W) fp32, w (Cout,Cin,kh,kw) fp32 (ConvTranspose2d layout with deconv) -> (N,Cout,Ho,Wo) fp32 = hi + lo of the GPU's output."""
    lib = load_library()
    dev = torch.device("cuda:0")
    n, cin, hi, wi = x.shape
    cin_p = cin_pad or cin
    in_halo = pad if in_halo is None else in_halo
    xn = torch.zeros((n, hi, wi, cin_p), dtype=torch.float32)
    xn[..., :cin] = x.permute(0, 2, 3, 1)
    xd = _planes(_halo(xn, in_halo)).to(dev)
    if deconv:
        g = w.permute(2, 3, 1, 0).reshape(4 * w.shape[1], w.shape[0]).numpy()
        w32 = _ohwi(g[:, :, None, None], g.shape[1], np.float32)
        bias = np.tile(b.numpy().astype(np.float32), 4)
        cout = w.shape[1]
        kh = kw = 1
    else:
        cout, _, kh, kw = w.shape
        w32 = _ohwi(w.numpy().astype(np.float32), cin_p, np.float32)
        bias = b.numpy().astype(np.float32)
    rows = w32.shape[0]
    rows_pad = (rows + 15) // 16 * 16
    if rows_pad != rows:
        w32 = np.concatenate([w32, np.zeros((rows_pad - rows, w32.shape[1]), np.float32)])
        bias = np.concatenate([bias, np.zeros(rows_pad - rows, np.float32)])
    ws, wsi = split_planes(w32)
    cout_store = rows_pad if out_f32 else cout
    wd, sd, bd = torch.from_numpy(ws).to(dev), torch.from_numpy(wsi).to(dev), torch.from_numpy(bias).to(dev)
    ho = (hi + 2 * pad - kh) // stride + 1
    wo = (wi + 2 * pad - kw) // stride + 1
    oh, ow = (2 * ho, 2 * wo) if deconv else (ho, wo)
    oshape = (n, oh + 2 * out_halo, ow + 2 * out_halo, cout_store)
    od = torch.zeros(oshape, dtype=torch.float32, device=dev) if out_f32 else torch.zeros((2,) + oshape, dtype=torch.float16, device=dev)
    rd = ud = None
    if res is not None:
        rd = _planes(_halo(res.permute(0, 2, 3, 1).contiguous(), out_halo)).to(dev)
    if up is not None:
        ud = _planes(_halo(up.permute(0, 2, 3, 1).contiguous(), out_halo)).to(dev)
    torch.cuda.synchronize()
    rc = lib.rs_op_conv2d_split(C.c_void_p(xd.data_ptr()), xd[0].numel(), C.c_void_p(wd.data_ptr()), rows_pad * w32.shape[1], C.c_void_p(sd.data_ptr()),
                                C.c_void_p(bd.data_ptr()), C.c_void_p(od.data_ptr()), 0 if out_f32 else od[0].numel(),
                                C.c_void_p(rd.data_ptr()) if rd is not None else None, rd[0].numel() if rd is not None else 0,
                                C.c_void_p(ud.data_ptr()) if ud is not None else None, ud[0].numel() if ud is not None else 0,
                                n, hi, wi, cin_p, in_halo, kh, kw, stride, pad, cout_store, w32.shape[1], out_halo, int(relu), int(out_f32),
                                int(deconv), variant, None)
    _check(lib, rc, "rs_op_conv2d_split")
    torch.cuda.synchronize()
    o = od.cpu()
    o = o if out_f32 else o[0].float() + o[1].float()
    if out_halo:
        inner = o[:, out_halo:-out_halo, out_halo:-out_halo]
        assert float(o.abs().sum()) == pytest.approx(float(inner.abs().sum()), rel=1e-6), "kernel wrote into the halo"
        o = inner
    return o[..., :cout].permute(0, 3, 1, 2).contiguous()


def _err(got, ref64, scale64):
    """max |got - ref| in units of the L2 norm of the output element's terms -- what the rounding noise of a dot product scales with
    (relative error eps per term gives eps * sqrt(sum t^2) in the sum)."""
    e = float(((got.double() - ref64).abs() / scale64).max())
    print(f"    err / |terms|_2 = {e:.3e}")
    return e


# In these units one fp32 rounding per term is 2^-24 = 6e-8 and a K-term fp32 chain accumulates about sqrt(K / 2) * 2^-24 (rms; the maximum over
# ~1e6 outputs is 4-5x that).  Measured on MI355X (profiles/r04/split_operator_errors.txt): 3x3 256 -> 256 (K = 2304) split 5.2e-6, the fp32-MFMA mode
# (csrc/ref_f32.hip) 8.0e-6 on the same operands; K = 64: 5.7e-7; K = 256: 0.9-1.4e-6; K = 12544: 1.2e-5.  The fp16 mode sits at 1e-3.
def TOL(k):
    return 2e-7 * np.sqrt(k) + 1e-6


def _ref(x, w, b, **kw):
    ref = F.conv2d(x.double(), w.double(), b.double(), **kw)
    scale = F.conv2d(x.double() ** 2, w.double() ** 2, None, **kw).sqrt() + b.double().abs().view(1, -1, 1, 1) + 1e-30
    return ref, scale


@pytest.mark.parametrize("variant", [0, 4, 12, 15, 16, 17, 7, 14])
def test_split_conv3x3_256(gpu_required, variant):
    """3x3 256 -> 256 on every tile that takes deep K (conv_igemm 128x128 / 256x256 / 64x128 / 128x256, conv_deep 256 / 160 / 192 / 224 pixels; 12
    tiles, so conv_deep's last round runs as 128-pixel halves), ragged M, residual + ReLU, activations spread over five decades."""
    g = torch.Generator().manual_seed(10)
    x = torch.randn(3, 256, 33, 29, generator=g) * torch.exp(2.0 * torch.randn(3, 256, 1, 1, generator=g))
    w = torch.randn(256, 256, 3, 3, generator=g) * 0.03 * torch.exp(torch.randn(256, 1, 1, 1, generator=g))
    b = torch.randn(256, generator=g)
    res = torch.randn(3, 256, 33, 29, generator=g)
    ref, scale = _ref(x, w, b, padding=1)
    ref = F.relu(ref + res.double())
    got = run_split_conv(x, w, b, pad=1, relu=True, res=res, variant=variant)
    e = _err(got, ref, scale + res.double().abs())
    if variant == 0:
        from tests.test_gpu_conv import run_conv_f32
        e32 = _err(run_conv_f32(x, w, b, pad=1, relu=True, res=res), ref, scale + res.double().abs())
        print(f"    fp32-MFMA mode on the same operands: {e32:.3e}")
        assert e <= 1.5 * e32, "the split mode is noisier than 1.5x the fp32 mode"
    assert e <= TOL(2304)
    # and bit-identical across tiles: every variant accumulates an output element in the same order
    if variant != 0:
        base = run_split_conv(x, w, b, pad=1, relu=True, res=res, variant=0)
        assert torch.equal(got, base), f"variant {variant} differs from the 128x128 tile"


def test_split_conv1x1_shallow_k_residual_and_upsample_add(gpu_required):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, 64, 30, 34, generator=g).abs()
    w = torch.randn(256, 64, 1, 1, generator=g) * 0.1
    b = torch.randn(256, generator=g)
    res = torch.randn(3, 256, 30, 34, generator=g)
    up = torch.randn(3, 256, 15, 17, generator=g)
    ref, scale = _ref(x, w, b)
    ref = F.relu(ref + res.double() + F.interpolate(up.double(), scale_factor=2, mode="nearest"))
    got = run_split_conv(x, w, b, res=res, up=up, relu=True, in_halo=1)
    assert _err(got, ref, scale + 1.0) <= TOL(64)


def test_split_conv1x1_stride2_and_cout64(gpu_required):
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 256, 50, 50, generator=g)
    w = torch.randn(64, 256, 1, 1, generator=g) * 0.06
    b = torch.randn(64, generator=g)
    ref, scale = _ref(x, w, b, stride=2)
    got = run_split_conv(x, w, b, stride=2, in_halo=1)
    assert _err(got, ref, scale) <= TOL(256)


def test_split_stem_7x7_s2_cin8(gpu_required):
    """Small-Cin path (pass outermost): 7x7 stride 2 on 8 padded channels (3 real)."""
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(2, 3, 64, 72, generator=g) * 255.0 - 110.0)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.05
    b = torch.randn(64, generator=g)
    ref, scale = _ref(x, w, b, stride=2, padding=3)
    ref = F.relu(ref)
    xp = torch.zeros(2, 8, 64, 72)
    xp[:, :3] = x
    wp = torch.zeros(64, 8, 7, 7)
    wp[:, :3] = w
    got = run_split_conv(xp, wp, b, stride=2, pad=3, relu=True, variant=1)
    assert _err(got, ref, scale) <= TOL(392)


def test_split_small_head_fp32_out(gpu_required):
    """16-row head with fp32 output (RPN objectness + deltas, box predictor): one plane out."""
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 256, 25, 31, generator=g).abs()
    w = torch.randn(15, 256, 1, 1, generator=g) * 0.05
    b = torch.randn(15, generator=g)
    ref, scale = _ref(x, w, b)
    got = run_split_conv(x, w, b, in_halo=1, out_halo=0, out_f32=True, variant=2)
    assert _err(got, ref, scale) <= TOL(256)


def test_split_deconv2x2(gpu_required):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(5, 256, 14, 14, generator=g).abs()
    w = torch.randn(256, 256, 2, 2, generator=g) * 0.05          # ConvTranspose2d layout (Cin, Cout, 2, 2)
    b = torch.randn(256, generator=g)
    ref = F.relu(F.conv_transpose2d(x.double(), w.double(), b.double(), stride=2))
    scale = F.conv_transpose2d(x.double() ** 2, w.double() ** 2, None, stride=2).sqrt() + 1.0
    got = run_split_conv(x, w, b, relu=True, in_halo=1, out_halo=0, deconv=True)
    assert _err(got, ref, scale) <= TOL(256)


def test_split_fc_gemm_k12544(gpu_required):
    """fc1's shape: K = 12544, rows as a (M x 1) image."""
    g = torch.Generator().manual_seed(6)
    x = torch.randn(1, 12544, 300, 1, generator=g).abs()
    w = torch.randn(1024, 12544, 1, 1, generator=g) * 0.01
    b = torch.randn(1024, generator=g)
    ref, scale = _ref(x, w, b)
    got = run_split_conv(x, w, b, relu=False, in_halo=0, out_halo=0)
    assert _err(got, ref, scale) <= TOL(12544)


# ---------------------------------------------------------------------------------------------------------------- engine
# End to end the mode is held to the fp32 mode's strict bounds by the tests of tests/test_gpu_engine.py and tests/test_golden.py that are parametrised over
# REF_MODES = ["fp32", "split"]; here: what is specific to it.
def test_split_mode_gives_each_tile_the_bits_it_gets_alone(gpu_required):
    """The tile dispatch depends on the batch size (conv_choose_variant over M = batch x pixels; multi-map launches; conv_deep's tile heights and split last
    round): every tile variant accumulates an output element in the same order (32-channel slice outer, taps inner; W_hi.X_lo, W_hi.X_hi, W_lo.X_hi per
    step), so batches of 2 .. 16 give each tile the detections it gets alone, bit for bit."""
    from tests.test_gpu_engine import _same_instances
    spec = EngineSpec(num_classes=2, precision="split")
    W = synthetic_weights(spec, seed=0)
    tiles = synthetic_tiles(16, 512, 512, 3, seed=2468)
    eng = Engine(spec, W, (512, 512, 3), max_batch=16)
    try:
        alone = [eng.infer(tiles[i:i + 1])[0] for i in range(16)]
        assert all(len(d) > 0 for d in alone)
        for b in (2, 3, 5, 8, 11, 16):
            got = eng.infer(tiles[:b])
            for i in range(b):
                assert _same_instances(alone[i], got[i]), f"batch {b}: tile {i} differs from the tile run alone"
    finally:
        eng.close()


def test_split_activations_are_two_planes_and_the_lane_pipeline_equals_one_engine(gpu_required):
    """Activations are exposed as hi + lo (dtype 5 of rs_engine_tensor): the lo plane holds what fp16 rounding of the value left behind, and two lanes
    on one stream give the detections of a single engine."""
    from proj_roadsurf_amd.engine import LanePipeline
    from tests.test_gpu_engine import _same_instances
    spec = EngineSpec(num_classes=2, min_size_test=320, max_size_test=533, rpn_pre_nms_topk_test=300, rpn_post_nms_topk_test=300, precision="split")
    W = synthetic_weights(spec, seed=0)
    tiles = synthetic_tiles(4, 256, 256, 3, seed=91)
    eng = Engine(spec, W, (256, 256, 3), max_batch=4)
    try:
        single = eng.infer(tiles)
        assert eng.tensor_is_split("p3") and not eng.tensor_is_split("box_pred")
        ptr, _, shape, _ = eng.tensor_ptr("p3")
        planes = np.empty((2,) + shape, np.float16)
        _check(eng.lib, eng.lib.rs_memcpy_d2h(planes.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), planes.nbytes), "rs_memcpy_d2h")
        hi, lo = planes[0].astype(np.float32), planes[1].astype(np.float32)
        assert np.abs(hi).max() > 0 and np.abs(lo).max() > 0
        assert np.all(np.abs(lo) <= 2.0 ** -11 * np.abs(hi) + 2.0 ** -24), "the lo plane exceeds half an ulp of the hi plane"
        # hi is the fp16 rounding of the value (a lo part rounded to exactly half an ulp can tip a tie the other way: a handful of elements in millions)
        big = np.abs(hi) >= 2.0 ** -10        # below, the lo part is a subnormal on the 2^-24 grid and may carry a whole ulp of hi
        assert np.mean(((hi + lo).astype(np.float16).astype(np.float32) != hi)[big]) <= 5e-4, "hi is not the fp16 rounding of hi + lo"
    finally:
        eng.close()
    pipe = LanePipeline(spec, W, (256, 256, 3), max_batch=4, lanes=2)
    try:
        out = [r for r in pipe.run(tiles for _ in range(3))]
        for res in out:
            assert all(_same_instances(a, b) for a, b in zip(single, res))
    finally:
        pipe.close()


# ---------------------------------------------------------------------------------------------
# Fused bottleneck tail in split mode (csrc/bneck_split.hip): conv2 + conv3 + identity residual + ReLU (+ the next block's conv1) chained through
# registers on hi + lo fragments, against the same chain in float64 on the values the planes hold.  The chain stores nothing in between, so the
# reference rounds nothing in between either; the kernel's intermediate hi + lo splits (22 bits) are part of what the bound covers.
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,hw,with_next,width,proj", [(2, (20, 28), True, 64, False), (1, (17, 23), False, 64, False), (3, (8, 8), True, 64, False),
                                                       (2, (18, 22), True, 128, False), (1, (25, 13), False, 128, False),
                                                       (2, (19, 27), True, 64, True), (1, (33, 16), False, 64, True)])
def test_split_bneck_tail_vs_float64(gpu_required, n, hw, with_next, width, proj):
    """proj: the first block of res2 -- no identity residual; the shortcut's 1x1 from the 64-channel block input rides in conv3's GEMM as 64 more K columns"""
    from proj_roadsurf_amd.weights import _perm_k64
    lib = load_library()
    dev = torch.device("cuda:0")
    torch.manual_seed(n * 100 + hw[0] + width)
    h, w = hw
    cb, c4 = width, 4 * width
    val = lambda pl: pl[0].double() + pl[1].double()
    t1p = _planes(_halo(torch.relu(torch.randn(n, h, w, cb)), 1))
    xp = _planes(_halo(torch.relu(torch.randn(n, h, w, 64 if proj else c4)), 1))
    wsc = torch.randn(c4, 64, 1, 1) / 8.0
    w2 = torch.randn(cb, cb, 3, 3) / (3.0 * cb ** 0.5) * torch.exp2(torch.randint(-6, 7, (cb, 1, 1, 1)).float())     # rows of very different scale:
    w3 = torch.randn(c4, cb, 1, 1) / cb ** 0.5 * torch.exp2(torch.randint(-6, 7, (1, cb, 1, 1)).float()) / 8.0        # the per-row weight scale matters
    w1 = torch.randn(cb, c4, 1, 1) / c4 ** 0.5
    b2, b3, b1 = torch.randn(cb) * 0.1, torch.randn(c4) * 0.1, torch.randn(cb) * 0.1
    packs = {}
    for name, wt, kin, perm in (("w2", w2, cb, None), ("w3", w3, cb, cb), ("w1", w1, c4, 64)):
        m = _ohwi(wt.numpy().astype(np.float32), kin, np.float32)
        if perm:
            m = _perm_k64(m, perm)
        if proj and name == "w3":
            m = np.concatenate([m, _ohwi(wsc.numpy().astype(np.float32), 64, np.float32)], 1)
        ws, wsi = split_planes(m)
        packs[name] = (torch.from_numpy(ws).to(dev), torch.from_numpy(wsi).to(dev))
        # the value the kernel multiplies with: (hi + lo) / scale, in the natural K order
        rows = m.shape[0]
        eff = (ws[:rows].astype(np.float64) + ws[rows:].astype(np.float64)) * wsi[:, None].astype(np.float64)
        assert np.abs(eff - m).max() <= 2.0 ** -21 * np.abs(m).max(axis=1).max()
    t1v, xv = val(t1p)[:, 1:-1, 1:-1].permute(0, 3, 1, 2), val(xp)[:, 1:-1, 1:-1].permute(0, 3, 1, 2)
    t2 = torch.relu(F.conv2d(t1v, w2.double(), b2.double(), padding=1))
    pre3 = F.conv2d(t2, w3.double(), b3.double()) + (F.conv2d(xv, wsc.double()) if proj else xv)
    out = torch.relu(pre3)
    t1n = torch.relu(F.conv2d(out, w1.double(), b1.double()))
    # scale of the rounding noise: conv3's terms, the residual and what conv2's noise contributes through |W3|
    s2 = F.conv2d(t1v ** 2, w2.double() ** 2, None, padding=1).sqrt() + b2.double().abs().view(1, -1, 1, 1)
    s3 = (F.conv2d(t2 ** 2, w3.double() ** 2).sqrt() + F.conv2d(s2 ** 2, w3.double() ** 2).sqrt() + b3.double().abs().view(1, -1, 1, 1) + 1e-30 +
          (F.conv2d(xv ** 2, wsc.double() ** 2).sqrt() if proj else xv.abs()))
    s1 = F.conv2d(out ** 2, w1.double() ** 2).sqrt() + F.conv2d(s3 ** 2, w1.double() ** 2).sqrt() + b1.double().abs().view(1, -1, 1, 1) + 1e-30
    t1d, xd = t1p.to(dev), xp.to(dev)
    outd = torch.zeros((2, n, h + 2, w + 2, c4), dtype=torch.float16, device=dev)
    t1nd = torch.zeros((2, n, h + 2, w + 2, cb), dtype=torch.float16, device=dev)
    b2d, b3d, b1d = b2.to(dev), b3.to(dev), b1.to(dev)
    torch.cuda.synchronize()
    P = lambda t: C.c_void_p(t.data_ptr())
    rc = lib.rs_op_bneck_tail_split(P(t1d), t1d[0].numel(), P(packs["w2"][0]), P(packs["w2"][1]), P(b2d), P(packs["w3"][0]), P(packs["w3"][1]), P(b3d),
                                    None if proj else P(xd), xd[0].numel(), P(outd), outd[0].numel(),
                                    P(packs["w1"][0]) if with_next else None, P(packs["w1"][1]) if with_next else None, P(b1d) if with_next else None,
                                    P(t1nd) if with_next else None, t1nd[0].numel() if with_next else 0,
                                    P(xd) if proj else None, xd[0].numel() if proj else 0, n, h, w, width, None)
    _check(lib, rc, "rs_op_bneck_tail_split")
    torch.cuda.synchronize()
    go = outd.cpu()
    gv = go[0].double() + go[1].double()
    assert float(gv.abs().sum()) == pytest.approx(float(gv[:, 1:-1, 1:-1].abs().sum()), rel=1e-9), "kernel wrote into the halo"
    hi = go[0][:, 1:-1, 1:-1]
    same = (hi == (go[0].float() + go[1].float())[:, 1:-1, 1:-1].half())[hi.abs() >= 2.0 ** -10]      # below, lo sits on the 2^-24 grid (see the two-plane test)
    assert float((~same).float().mean()) <= 5e-4, "hi plane is not fp16(value)"
    assert _err(gv[:, 1:-1, 1:-1].permute(0, 3, 1, 2), out, s3) <= TOL(9 * cb) + TOL(cb + (64 if proj else 0))
    if with_next:
        gt = t1nd.cpu()
        tv = gt[0].double() + gt[1].double()
        assert float(tv.abs().sum()) == pytest.approx(float(tv[:, 1:-1, 1:-1].abs().sum()), rel=1e-9)
        assert _err(tv[:, 1:-1, 1:-1].permute(0, 3, 1, 2), t1n, s1) <= TOL(9 * cb) + TOL(cb) + TOL(c4)
    else:
        assert float(t1nd.abs().sum()) == 0.0
