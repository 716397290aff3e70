"""GPU parity of the implicit-GEMM conv kernel (through the C ABI ``rs_op_conv2d``) against
``torch.nn.functional`` fp32 on the same fp16-representable operands.  Tolerance: the kernel
accumulates fp16 products in fp32 (MFMA) and rounds the result to fp16 once, so
|err| <= 2^-10 * |ref| + accumulation noise; we assert max|err| <= 2e-3 * max(1, max|ref|)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from proj_roadsurf_amd.engine import load_library, _check
from proj_roadsurf_amd.weights import _ohwi

pytestmark = pytest.mark.gpu


def _halo(x_nhwc: torch.Tensor, pad: int) -> torch.Tensor:
    n, h, w, c = x_nhwc.shape
    out = torch.zeros((n, h + 2 * pad, w + 2 * pad, c), dtype=x_nhwc.dtype)
    out[:, pad:pad + h, pad:pad + w] = x_nhwc
    return out


def run_conv(x, w, b, *, stride=1, pad=0, relu=False, res=None, up=None, in_halo=None, out_halo=1, out_f32=False,
             deconv=False, variant=-1, glds=1, cin_pad=None):
    """x (N,C,H,W) fp32 (fp16-representable), w (Cout,Cin,kh,kw) fp32 -> (N,Cout,Ho,Wo) fp32 from the GPU."""
    lib = load_library()
    dev = torch.device("cuda:0")
    n, cin, hi, wi = x.shape
    cout, _, kh, kw = w.shape
    cin_p = cin_pad or cin
    in_halo = pad if in_halo is None else in_halo
    xn = torch.zeros((n, hi, wi, cin_p), dtype=torch.float16)
    xn[..., :cin] = x.permute(0, 2, 3, 1).half()
    xd = _halo(xn, in_halo).to(dev)
    if deconv:
        # w is ConvTranspose2d weight (Cin, Cout, 2, 2) -> rows (dy,dx,co) x ci
        g = w.permute(2, 3, 1, 0).reshape(4 * w.shape[1], w.shape[0]).numpy()
        wp = _ohwi(g[:, :, None, None], g.shape[1])
        bias = np.tile(b.numpy().astype(np.float32), 4)
        cout = w.shape[1]
        kh = kw = 1
    else:
        wp = _ohwi(w.numpy().astype(np.float32), cin_p)
        bias = b.numpy().astype(np.float32)
    rows = wp.shape[0]
    rows_pad = (rows + 15) // 16 * 16
    if rows_pad != rows:
        wp = np.concatenate([wp, np.zeros((rows_pad - rows, wp.shape[1]), np.float16)])
        bias = np.concatenate([bias, np.zeros(rows_pad - rows, np.float32)])
    cout_store = rows_pad if out_f32 else cout
    wd = torch.from_numpy(wp).to(dev)
    bd = torch.from_numpy(bias).to(dev)
    ho = (hi + 2 * pad - kh) // stride + 1
    wo = (wi + 2 * pad - kw) // stride + 1
    oh, ow = (2 * ho, 2 * wo) if deconv else (ho, wo)
    od = torch.zeros((n, oh + 2 * out_halo, ow + 2 * out_halo, cout_store), dtype=torch.float32 if out_f32 else torch.float16, device=dev)
    rd = ud = None
    if res is not None:
        rd = _halo(res.permute(0, 2, 3, 1).half().contiguous(), out_halo).to(dev)
    if up is not None:
        ud = _halo(up.permute(0, 2, 3, 1).half().contiguous(), out_halo).to(dev)
    torch.cuda.synchronize()
    rc = lib.rs_op_conv2d(C.c_void_p(xd.data_ptr()), C.c_void_p(wd.data_ptr()), C.c_void_p(bd.data_ptr()), C.c_void_p(od.data_ptr()),
                          C.c_void_p(rd.data_ptr()) if rd is not None else None, C.c_void_p(ud.data_ptr()) if ud is not None else None,
                          n, hi, wi, cin_p, in_halo, kh, kw, stride, pad, cout_store, wp.shape[1], out_halo, int(relu), int(out_f32),
                          int(deconv), variant, glds, None)
    _check(lib, rc, "rs_op_conv2d")
    torch.cuda.synchronize()
    o = od.cpu().float()
    if out_halo:
        # the halo must stay untouched (zero)
        inner = o[:, out_halo:-out_halo, out_halo:-out_halo]
        assert float(o.abs().sum()) == pytest.approx(float(inner.abs().sum()), rel=1e-6), "kernel wrote into the halo"
        o = inner
    return o[..., :cout].permute(0, 3, 1, 2).contiguous()


def _r16(t):
    return t.half().float()


def _check_close(got, ref, tol=2e-3):
    scale = max(1.0, float(ref.abs().max()))
    err = float((got - ref).abs().max())
    assert err <= tol * scale, f"max err {err} > {tol * scale}"


@pytest.mark.parametrize("glds", [1, 0])
@pytest.mark.parametrize("variant", [0, 1])
def test_conv3x3_256(gpu_required, glds, variant):
    g = torch.Generator().manual_seed(0)
    x = _r16(torch.randn(2, 256, 37, 41, generator=g))
    w = _r16(torch.randn(256, 256, 3, 3, generator=g) * 0.03)
    b = torch.randn(256, generator=g)
    ref = F.relu(F.conv2d(x, w, b, padding=1))
    got = run_conv(x, w, b, pad=1, relu=True, variant=variant, glds=glds)
    _check_close(got, ref)


@pytest.mark.parametrize("variant", [4, 12, 15, 16, 17, 18, 19, 20])
def test_conv3x3_256_big_tiles(gpu_required, variant):
    """256x256 workgroup tiles (variant 4 = conv_igemm<2,4,4,8>, variant 12 = conv_deep, production for deep-K layers; 12 tiles
    here, so every tile of the conv_deep launch runs as two 128-pixel halves -- the split last round) and conv_deep with tiles of
    160 / 192 / 224 (15 / 16 / 17: odd numbers of 16-pixel blocks per wave stage a half-used pass) and 64 / 96 / 128 pixels (18 / 19 /
    20): ragged M (not a multiple of any tile height), residual + ReLU epilogue."""
    g = torch.Generator().manual_seed(10)
    x = _r16(torch.randn(3, 256, 33, 29, generator=g))
    w = _r16(torch.randn(256, 256, 3, 3, generator=g) * 0.03)
    b = torch.randn(256, generator=g)
    res = _r16(torch.randn(3, 256, 33, 29, generator=g))
    ref = F.relu(F.conv2d(x, w, b, padding=1) + res)
    got = run_conv(x, w, b, pad=1, relu=True, res=res, variant=variant)
    _check_close(got, ref)


@pytest.mark.parametrize("glds", [1, 0])
def test_conv1x1_residual_relu(gpu_required, glds):
    g = torch.Generator().manual_seed(1)
    x = _r16(torch.randn(3, 64, 29, 31, generator=g))
    w = _r16(torch.randn(256, 64, 1, 1, generator=g) * 0.1)
    b = torch.randn(256, generator=g)
    res = _r16(torch.randn(3, 256, 29, 31, generator=g))
    ref = F.relu(F.conv2d(x, w, b) + res)
    got = run_conv(x, w, b, res=res, relu=True, in_halo=1, glds=glds)
    _check_close(got, ref)


def test_conv1x1_to64_variant1(gpu_required):
    g = torch.Generator().manual_seed(2)
    x = _r16(torch.randn(2, 256, 50, 50, generator=g))
    w = _r16(torch.randn(64, 256, 1, 1, generator=g) * 0.06)
    b = torch.randn(64, generator=g)
    ref = F.relu(F.conv2d(x, w, b))
    got = run_conv(x, w, b, relu=True, in_halo=1)
    _check_close(got, ref)


def test_conv1x1_stride2(gpu_required):
    g = torch.Generator().manual_seed(3)
    x = _r16(torch.randn(2, 256, 40, 36, generator=g))
    w = _r16(torch.randn(512, 256, 1, 1, generator=g) * 0.06)
    b = torch.randn(512, generator=g)
    ref = F.conv2d(x, w, b, stride=2)
    got = run_conv(x, w, b, stride=2, in_halo=1)
    _check_close(got, ref)


@pytest.mark.parametrize("glds", [1, 0])
def test_stem_7x7_s2_cin3(gpu_required, glds):
    g = torch.Generator().manual_seed(4)
    x = _r16(torch.randn(2, 3, 64, 96, generator=g) * 50)
    w = _r16(torch.randn(64, 3, 7, 7, generator=g) * 0.01)
    b = torch.randn(64, generator=g)
    ref = F.relu(F.conv2d(x, w, b, stride=2, padding=3))
    got = run_conv(x, w, b, stride=2, pad=3, relu=True, cin_pad=8, variant=1, glds=glds)
    _check_close(got, ref)


def test_fpn_lateral_upsample_add(gpu_required):
    g = torch.Generator().manual_seed(5)
    x = _r16(torch.randn(2, 512, 26, 30, generator=g))
    w = _r16(torch.randn(256, 512, 1, 1, generator=g) * 0.04)
    b = torch.randn(256, generator=g)
    top = _r16(torch.randn(2, 256, 13, 15, generator=g))
    ref = F.conv2d(x, w, b) + F.interpolate(top, scale_factor=2.0, mode="nearest")
    got = run_conv(x, w, b, up=top, in_halo=1)
    _check_close(got, ref)


def test_small_head_fp32_out(gpu_required):
    g = torch.Generator().manual_seed(6)
    x = _r16(torch.randn(2, 256, 25, 27, generator=g))
    w = _r16(torch.randn(15, 256, 1, 1, generator=g) * 0.06)
    b = torch.randn(15, generator=g)
    ref = F.conv2d(x, w, b)
    got = run_conv(x, w, b, out_f32=True, out_halo=0, variant=2)
    assert float((got - ref).abs().max()) <= 2e-4 * max(1.0, float(ref.abs().max()))


def test_deconv2x2_pixel_shuffle(gpu_required):
    g = torch.Generator().manual_seed(7)
    x = _r16(torch.randn(5, 256, 14, 14, generator=g))
    w = _r16(torch.randn(256, 256, 2, 2, generator=g) * 0.06)    # (Cin, Cout, 2, 2)
    b = torch.randn(256, generator=g)
    ref = F.relu(F.conv_transpose2d(x, w, b, stride=2))
    got = run_conv(x, w, b, relu=True, deconv=True, in_halo=1, out_halo=0)
    _check_close(got, ref)


def test_fc_gemm(gpu_required):
    g = torch.Generator().manual_seed(8)
    m, k, nout = 333, 12544, 1024
    a = _r16(torch.randn(m, k, generator=g))
    w = _r16(torch.randn(nout, k, generator=g) * 0.01)
    b = torch.randn(nout, generator=g)
    ref = F.relu(F.linear(a, w, b))
    x = a.t().reshape(1, k, 1, m).permute(0, 1, 3, 2).contiguous()     # (1, K, M, 1): "image" of M x 1 pixels
    got = run_conv(x, w.view(nout, k, 1, 1), b, relu=True, out_halo=0)
    got = got[0, :, :, 0].t()
    _check_close(got, ref, tol=3e-3)


@pytest.mark.parametrize("cin,cin2,cout,stride2,hw,variant", [
    (64, 64, 256, 1, (37, 41), -1),      # res2.0: conv3 64->256 + shortcut 64->256
    (64, 64, 256, 1, (37, 41), 0),
    (128, 256, 512, 2, (26, 30), -1),    # res3.0: conv3 128->512 + stride-2 shortcut 256->512
    (128, 256, 512, 2, (26, 30), 7),
    (128, 256, 512, 2, (26, 30), 10),
    (256, 512, 1024, 2, (19, 17), 4),    # res4.0 on the 256x256 tile
])
def test_bottleneck_out_dual_source(gpu_required, cin, cin2, cout, stride2, hw, variant):
    """conv3 + projection shortcut as one GEMM over two K sources (rs_op_conv2d_dual) == relu(conv1x1(a) +
    conv1x1_stride(b) + bias) in torch fp32."""
    lib = load_library()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(20 + cin)
    h, w = hw
    n = 3
    a = _r16(torch.randn(n, cin, h, w, generator=g))
    b = _r16(torch.randn(n, cin2, h * stride2, w * stride2, generator=g))
    w3 = _r16(torch.randn(cout, cin, 1, 1, generator=g) * 0.08)
    wsc = _r16(torch.randn(cout, cin2, 1, 1, generator=g) * 0.05)
    bias = torch.randn(cout, generator=g)
    ref = F.relu(F.conv2d(a, w3) + F.conv2d(b, wsc, stride=stride2) + bias.view(1, -1, 1, 1))
    ad = _halo(a.permute(0, 2, 3, 1).half().contiguous(), 1).to(dev)
    bd = _halo(b.permute(0, 2, 3, 1).half().contiguous(), 1).to(dev)
    wcat = np.concatenate([_ohwi(w3.numpy(), cin), _ohwi(wsc.numpy(), cin2)], 1)
    wd = torch.from_numpy(wcat).to(dev)
    bsd = bias.to(dev)
    od = torch.zeros((n, h + 2, w + 2, cout), dtype=torch.float16, device=dev)
    torch.cuda.synchronize()
    rc = lib.rs_op_conv2d_dual(C.c_void_p(ad.data_ptr()), C.c_void_p(bd.data_ptr()), C.c_void_p(wd.data_ptr()), C.c_void_p(bsd.data_ptr()),
                               C.c_void_p(od.data_ptr()), n, h, w, cin, 1, 1, 1, 1, 0, h * stride2, w * stride2, cin2, 1, stride2,
                               cout, wcat.shape[1], 1, 1, variant, None)
    _check(lib, rc, "rs_op_conv2d_dual")
    torch.cuda.synchronize()
    o = od.cpu().float()
    inner = o[:, 1:-1, 1:-1]
    assert float(o.abs().sum()) == pytest.approx(float(inner.abs().sum()), rel=1e-6), "kernel wrote into the halo"
    _check_close(inner.permute(0, 3, 1, 2), ref)


def test_rejects_bad_shapes(gpu_required):
    lib = load_library()
    rc = lib.rs_op_conv2d(C.c_void_p(1), C.c_void_p(1), C.c_void_p(1), C.c_void_p(1), None, None,
                          1, 8, 8, 48, 1, 3, 3, 1, 1, 64, 448, 1, 0, 0, 0, -1, 1, None)
    assert rc != 0 and b"Cin" in lib.rs_last_error()


# ---------------------------------------------------------------------------------------------
# Reference-precision kernels (csrc/ref_f32.hip): fp32 activations / weights on v_mfma_f32_16x16x4_f32 (use_glds = -1) and
# the VALU cross-check (use_glds = -2), against torch fp32 on the CPU.  fp32 sums of K <= 2304 products in different
# orders: 2e-5 relative to the output scale.
# ---------------------------------------------------------------------------------------------
def run_conv_f32(x, w, b, *, stride=1, pad=0, relu=False, res=None, up=None, out_halo=1, deconv=False, valu=False, tile=-1):
    lib = load_library()
    dev = torch.device("cuda:0")
    n, cin, hi, wi = x.shape
    if deconv:
        g = w.permute(2, 3, 1, 0).reshape(4 * w.shape[1], w.shape[0]).numpy()
        wp = _ohwi(g[:, :, None, None], g.shape[1], np.float32)
        bias = np.tile(b.numpy().astype(np.float32), 4)
        cout, kh, kw = w.shape[1], 1, 1
    else:
        cout, _, kh, kw = w.shape
        wp = _ohwi(w.numpy().astype(np.float32), cin, np.float32)
        bias = b.numpy().astype(np.float32)
    rows = wp.shape[0]
    rows_pad = (rows + 15) // 16 * 16
    if rows_pad != rows:
        wp = np.concatenate([wp, np.zeros((rows_pad - rows, wp.shape[1]), np.float32)])
        bias = np.concatenate([bias, np.zeros(rows_pad - rows, np.float32)])
    cout_store = cout if deconv else rows_pad
    xd = _halo(x.permute(0, 2, 3, 1).contiguous().float(), pad).to(dev)
    wd, bd = torch.from_numpy(wp).to(dev), torch.from_numpy(bias).to(dev)
    ho, wo = (hi + 2 * pad - kh) // stride + 1, (wi + 2 * pad - kw) // stride + 1
    oh, ow = (2 * ho, 2 * wo) if deconv else (ho, wo)
    od = torch.zeros((n, oh + 2 * out_halo, ow + 2 * out_halo, cout_store), dtype=torch.float32, device=dev)
    rd = _halo(res.permute(0, 2, 3, 1).contiguous().float(), out_halo).to(dev) if res is not None else None
    ud = _halo(up.permute(0, 2, 3, 1).contiguous().float(), out_halo).to(dev) if up is not None else None
    torch.cuda.synchronize()
    rc = lib.rs_op_conv2d(C.c_void_p(xd.data_ptr()), C.c_void_p(wd.data_ptr()), C.c_void_p(bd.data_ptr()), C.c_void_p(od.data_ptr()),
                          C.c_void_p(rd.data_ptr()) if rd is not None else None, C.c_void_p(ud.data_ptr()) if ud is not None else None,
                          n, hi, wi, cin, pad, kh, kw, stride, pad, cout_store, wp.shape[1], out_halo, int(relu), 1,
                          int(deconv), tile, -2 if valu else -1, None)
    _check(lib, rc, "rs_op_conv2d(fp32)")
    torch.cuda.synchronize()
    o = od.cpu()
    if out_halo:
        inner = o[:, out_halo:-out_halo, out_halo:-out_halo]
        assert float(o.abs().sum()) == pytest.approx(float(inner.abs().sum()), rel=1e-6), "kernel wrote into the halo"
        o = inner
    return o[..., :cout].permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("cin,cout,k,stride,hw,n", [
    (64, 128, 3, 1, (23, 31), 2),      # 128x128 tile, ragged M
    (256, 64, 1, 2, (24, 40), 1),      # 128x64 tile, stride-2 1x1 (res3.0.conv1 shape class)
    (64, 256, 1, 1, (20, 20), 2),      # expansion with residual + ReLU
    (256, 15, 1, 1, (13, 17), 2),      # 16-row head tile (Cout padded 15 -> 16)
    (32, 64, 3, 1, (9, 9), 1),         # a single 32-float K step per tap
])
def test_conv_f32_mfma_vs_torch(gpu_required, cin, cout, k, stride, hw, n):
    torch.manual_seed(cin + cout + k)
    x = torch.randn(n, cin, *hw)
    w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
    b = torch.randn(cout)
    pad = k // 2
    ref = F.conv2d(x, w, b, stride=stride, padding=pad)
    res = torch.randn_like(ref) if cout == 256 else None
    want = ref + res if res is not None else ref
    if res is not None:
        want = torch.relu(want)
    got = run_conv_f32(x, w, b, stride=stride, pad=pad, relu=res is not None, res=res)
    valu = run_conv_f32(x, w, b, stride=stride, pad=pad, relu=res is not None, res=res, valu=True)
    _check_close(got, want, tol=2e-5)
    _check_close(valu, want, tol=2e-5)
    # the dispatch picks the tile by the number of tiles (csrc/ref_f32.hip launch_conv_f32: 128 x 64 when few, 128 x 128 when many); forced
    # here (30 = 128 x 128, 31 = 128 x 64), both give the SAME bits as the dispatched one: the K order of an output element is the tile's
    for tile in ([30, 31] if cout % 128 == 0 else [31] if cout % 64 == 0 else []):
        forced = run_conv_f32(x, w, b, stride=stride, pad=pad, relu=res is not None, res=res, tile=tile)
        assert torch.equal(forced, got), f"tile {tile}: {int((forced != got).sum())} elements differ"


def test_conv_f32_mfma_upsample_add_and_deconv(gpu_required):
    torch.manual_seed(7)
    x = torch.randn(2, 128, 12, 16)
    w = torch.randn(256, 128, 1, 1) / 128 ** 0.5
    b = torch.randn(256)
    up = torch.randn(2, 256, 6, 8)
    want = F.conv2d(x, w, b) + F.interpolate(up, scale_factor=2, mode="nearest")
    _check_close(run_conv_f32(x, w, b, up=up), want, tol=2e-5)
    # 2x2 stride-2 transposed conv + ReLU (mask head deconv)
    xd = torch.randn(3, 256, 14, 14)
    wt = torch.randn(256, 256, 2, 2) / 16.0
    bt = torch.randn(256)
    want = torch.relu(F.conv_transpose2d(xd, wt, bt, stride=2))
    _check_close(run_conv_f32(xd, wt, bt, relu=True, deconv=True, out_halo=0), want, tol=2e-5)


# ---------------------------------------------------------------------------------------------
# Fused bottleneck tail (csrc/bneck_fused.hip): conv2 + conv3 + residual + ReLU (+ the next block's conv1) chained through
# registers, against the same chain in torch fp32 with the intermediate maps rounded to fp16 where the engine stores or
# forwards them as fp16 (t2, out).
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,hw,with_next,proj,width", [(2, (20, 28), True, False, 64), (1, (17, 23), False, False, 64), (3, (8, 8), True, False, 64),
                                                        (2, (19, 27), True, True, 64), (1, (33, 16), False, True, 64),
                                                        (2, (18, 22), True, False, 128), (1, (25, 13), False, False, 128)])
def test_bneck_tail_fused_vs_torch(gpu_required, n, hw, with_next, proj, width):
    from proj_roadsurf_amd.weights import _perm_k64
    lib = load_library()
    dev = torch.device("cuda:0")
    torch.manual_seed(n * 100 + hw[0])
    h, w = hw
    cb, c4 = width, 4 * width
    t1 = _r16(torch.relu(torch.randn(n, cb, h, w)))
    x = _r16(torch.relu(torch.randn(n, 64 if proj else c4, h, w)))      # proj: the 64-channel input of a projection shortcut
    wsc = _r16(torch.randn(c4, 64, 1, 1) / 8.0)
    w2 = _r16(torch.randn(cb, cb, 3, 3) / (3.0 * cb ** 0.5))
    w3 = _r16(torch.randn(c4, cb, 1, 1) / cb ** 0.5)
    w1 = _r16(torch.randn(cb, c4, 1, 1) / c4 ** 0.5)
    b2, b3, b1 = torch.randn(cb) * 0.1, torch.randn(c4) * 0.1, torch.randn(cb) * 0.1
    t2 = _r16(torch.relu(F.conv2d(t1, w2, b2, padding=1)))
    out = _r16(torch.relu(F.conv2d(t2, w3, b3) + (F.conv2d(x, wsc) if proj else x)))
    t1n = torch.relu(F.conv2d(out, w1, b1))
    t1d = _halo(t1.permute(0, 2, 3, 1).half().contiguous(), 1).to(dev)
    xd = _halo(x.permute(0, 2, 3, 1).half().contiguous(), 1).to(dev)
    outd = torch.zeros((n, h + 2, w + 2, c4), dtype=torch.float16, device=dev)
    t1nd = torch.zeros((n, h + 2, w + 2, cb), dtype=torch.float16, device=dev)
    w2d = torch.from_numpy(_ohwi(w2.numpy(), cb)).to(dev)
    w3d = torch.from_numpy(_perm_k64(_ohwi(w3.numpy(), cb), cb)).to(dev)
    w1d = torch.from_numpy(_perm_k64(_ohwi(w1.numpy(), c4), 64)).to(dev)
    wscd = torch.from_numpy(_ohwi(wsc.numpy(), 64)).to(dev)
    b2d, b3d, b1d = b2.to(dev), b3.to(dev), b1.to(dev)
    torch.cuda.synchronize()
    P = lambda t: C.c_void_p(t.data_ptr())
    rc = lib.rs_op_bneck_tail(P(t1d), P(w2d), P(b2d), P(w3d), P(b3d), None if proj else P(xd), P(outd), P(w1d) if with_next else None,
                              P(b1d) if with_next else None, P(t1nd) if with_next else None, P(xd) if proj else None, P(wscd) if proj else None,
                              n, h, w, width, None)
    _check(lib, rc, "rs_op_bneck_tail")
    torch.cuda.synchronize()
    go = outd.cpu().float()
    assert float(go.abs().sum()) == pytest.approx(float(go[:, 1:-1, 1:-1].abs().sum()), rel=1e-6), "kernel wrote into the halo"
    _check_close(go[:, 1:-1, 1:-1].permute(0, 3, 1, 2), out, tol=2e-3)
    if with_next:
        gt = t1nd.cpu().float()
        assert float(gt.abs().sum()) == pytest.approx(float(gt[:, 1:-1, 1:-1].abs().sum()), rel=1e-6)
        _check_close(gt[:, 1:-1, 1:-1].permute(0, 3, 1, 2), t1n, tol=3e-3)
    else:
        assert float(t1nd.abs().sum()) == 0.0


@pytest.mark.parametrize("epi,n,hw,cout,relu", [("none", 2, (37, 41), 256, True), ("res", 3, (50, 50), 1024, True), ("up", 2, (52, 60), 256, False),
                                                ("res", 1, (7, 9), 512, False), ("none", 1, (3, 5), 256, False)])
def test_conv1x1_register_weights_is_bit_identical_to_the_tiled_kernel(gpu_required, epi, n, hw, cout, relu):
    """conv_wreg.hip (variant 22: persistent workgroups, weights in registers, activation and residual / top-down tiles staged by
    LDS-DMA under one counted wait per tile) against conv_igemm's 128x256 tile (variant 14) on the same operands: the same bits, on maps
    with a ragged last 64-pixel tile, more tiles than workgroups x 2 (so the two-tile ring wraps), fewer tiles than workgroups, and a
    map smaller than one tile; and both against torch fp32."""
    g = torch.Generator().manual_seed(31)
    h, w_ = hw
    x = _r16(torch.randn(n, 256, h, w_, generator=g))
    w = _r16(torch.randn(cout, 256, 1, 1, generator=g) * 0.06)
    b = torch.randn(cout, generator=g)
    res = _r16(torch.randn(n, cout, h, w_, generator=g)) if epi == "res" else None
    up = _r16(torch.randn(n, cout, h // 2, w_ // 2, generator=g)) if epi == "up" else None
    ref = F.conv2d(x, w, b)
    if res is not None:
        ref = ref + res
    if up is not None:
        ref = ref + F.interpolate(up, scale_factor=2.0, mode="nearest")
    if relu:
        ref = F.relu(ref)
    a = run_conv(x, w, b, relu=relu, res=res, up=up, variant=14)
    _check_close(a, ref)
    for variant in (22, 23, 25, 26):    # the shipped form / 64-pixel tiles with eight waves / 32-pixel tiles, two workgroups per CU / 64-pixel tiles, four waves
        c = run_conv(x, w, b, relu=relu, res=res, up=up, variant=variant)
        assert torch.equal(a, c), f"variant {variant}: {int((a != c).sum())} of {a.numel()} elements differ, max {float((a - c).abs().max())}"


def test_device_division_equals_ieee_division(gpu_required):
    """csrc/common.h rs_fdiv -- what every kernel divides with since the compiler's fp32 division sequence turned out to return wrong quotients beside another
    kernel's MFMA waves -- against IEEE division (numpy float32, correctly rounded) on 8 million operand pairs: bit-identical over 24 decades of
    magnitude on either side (1e-12 .. 1e12, quotients 1e-24 .. 1e24: far beyond box coordinates, probabilities and losses), on exact cases, and on the special
    values (zero / infinite divisor, infinite / NaN dividend)."""
    lib = load_library()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(7)
    n = 1 << 22
    mant = lambda: (1.0 + rng.random(n)).astype(np.float32) * rng.choice(np.array([-1.0, 1.0], np.float32), n)
    a = mant() * np.exp2(rng.integers(-40, 41, n)).astype(np.float32)
    b = mant() * np.exp2(rng.integers(-40, 41, n)).astype(np.float32)
    # the magnitudes the kernels see: pixel coordinates over box sizes, probabilities, small integers
    a2 = np.concatenate([rng.uniform(-2000, 2000, n // 2), rng.integers(-1000, 1000, n // 2)]).astype(np.float32)
    b2 = np.concatenate([rng.uniform(1e-3, 2000, n // 2), rng.integers(1, 1000, n // 2)]).astype(np.float32)
    sa = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 3.0, 0.0, np.inf, 1e30, 1.0], np.float32)
    sb = np.array([1.0, 2.0, 0.0, -0.0, 2.0, np.inf, 1.0, np.nan, 0.0, 0.0, np.inf, -np.inf], np.float32)
    for x, y in ((a, b), (a2, b2), (sa, sb)):
        xd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
        od = torch.empty_like(xd)
        torch.cuda.synchronize()
        _check(lib, lib.rs_op_fdiv(C.c_void_p(xd.data_ptr()), C.c_void_p(yd.data_ptr()), C.c_void_p(od.data_ptr()), x.size, None), "rs_op_fdiv")
        torch.cuda.synchronize()
        got = od.cpu().numpy()
        with np.errstate(all="ignore"):
            want = (x / y).astype(np.float32)
        same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
        assert bool(same.all()), f"{int((~same).sum())} of {x.size} quotients differ, first: {x[~same][:3]} / {y[~same][:3]} -> {got[~same][:3]} against {want[~same][:3]}"
