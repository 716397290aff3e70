"""GPU parity of the training-path operators, through the C ABI, against torch autograd
(fp32) on the same fp16-representable operands.  rs_op_conv2d_wgrad accumulates fp16 products in fp32 on MFMA over
up to ~1e5 pixels; tolerance: max|err| <= 2e-3 * max|ref| (+ fp32 summation-order noise)."""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from proj_roadsurf_amd.engine import load_library, _check

pytestmark = pytest.mark.gpu


def _halo(x_nhwc: torch.Tensor, pad: int) -> torch.Tensor:
    n, h, w, c = x_nhwc.shape
    out = torch.zeros((n, h + 2 * pad, w + 2 * pad, c), dtype=x_nhwc.dtype)
    out[:, pad:pad + h, pad:pad + w] = x_nhwc
    return out


def _r16(t):
    return t.half().float()


def run_wgrad(x, dy, k, stride, pad, scale=None, splits=0, in_halo=None, dy_halo=1, f32=False):
    """x (N,Cin,H,W), dy (N,Cout,Ho,Wo) fp32 (fp16-representable unless f32) -> dW (Cout,Cin,k,k) fp32 from the GPU.
    f32: operands stay fp32 and the reference-precision kernel runs (rs_op_conv2d_wgrad_f32)."""
    lib = load_library()
    dev = torch.device("cuda:0")
    n, cin, hi, wi = x.shape
    cout = dy.shape[1]
    in_halo = pad if in_halo is None else in_halo
    dt = torch.float32 if f32 else torch.float16
    xd = _halo(x.permute(0, 2, 3, 1).to(dt).contiguous(), in_halo).to(dev)
    dyd = _halo(dy.permute(0, 2, 3, 1).to(dt).contiguous(), dy_halo).to(dev)
    if f32:
        kpad = k * k * cin
        gd = torch.full((cout, kpad), float("nan"), dtype=torch.float32, device=dev)
        sd = scale.to(dev) if scale is not None else None
        lib.rs_op_conv2d_wgrad_f32.argtypes = lib.rs_op_conv2d_wgrad.argtypes
        torch.cuda.synchronize()
        rc = lib.rs_op_conv2d_wgrad_f32(C.c_void_p(dyd.data_ptr()), C.c_void_p(xd.data_ptr()), C.c_void_p(gd.data_ptr()),
                                        C.c_void_p(sd.data_ptr()) if sd is not None else None,
                                        n, hi, wi, cin, in_halo, k, k, stride, pad, cout, kpad, dy_halo, splits, None)
        _check(lib, rc, "rs_op_conv2d_wgrad_f32")
        torch.cuda.synchronize()
        return gd.cpu().reshape(cout, k, k, cin).permute(0, 3, 1, 2).contiguous()
    kpad = (k * k * cin + 63) // 64 * 64
    gd = torch.full((cout, kpad), float("nan"), dtype=torch.float32, device=dev)
    sd = scale.to(dev) if scale is not None else None
    torch.cuda.synchronize()
    rc = lib.rs_op_conv2d_wgrad(C.c_void_p(dyd.data_ptr()), C.c_void_p(xd.data_ptr()), C.c_void_p(gd.data_ptr()),
                                C.c_void_p(sd.data_ptr()) if sd is not None else None,
                                n, hi, wi, cin, in_halo, k, k, stride, pad, cout, kpad, dy_halo, splits, None)
    _check(lib, rc, "rs_op_conv2d_wgrad")
    torch.cuda.synchronize()
    g = gd.cpu()[:, : k * k * cin].reshape(cout, k, k, cin).permute(0, 3, 1, 2).contiguous()
    return g


def ref_wgrad(x, dy, k, stride, pad):
    w = torch.zeros(dy.shape[1], x.shape[1], k, k, requires_grad=True)
    y = F.conv2d(x, w, stride=stride, padding=pad)
    assert y.shape == dy.shape, (y.shape, dy.shape)
    y.backward(dy)
    return w.grad.detach()


@pytest.mark.parametrize("cin,cout,k,stride,hw,n,splits", [
    (256, 256, 3, 1, (25, 27), 2, 0),      # FPN output / RPN / mask-head 3x3
    (256, 256, 3, 1, (25, 27), 2, 1),      # single split: every pixel through one workgroup column
    (64, 128, 1, 1, (19, 23), 3, 0),       # 1x1, one K unit (second half of the workgroup tile unused)
    (256, 128, 1, 2, (26, 30), 2, 0),      # strided 1x1 (res3.0.conv1 shape)
    (128, 128, 3, 1, (14, 14), 5, 3),      # ragged pixel count vs the 64-pixel K step
    (512, 256, 1, 1, (13, 13), 2, 0),      # FPN lateral
])
def test_conv_wgrad_matches_autograd(gpu_required, cin, cout, k, stride, hw, n, splits):
    g = torch.Generator().manual_seed(cin + cout + k)
    h, w = hw
    pad = k // 2
    x = _r16(torch.randn(n, cin, h, w, generator=g))
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    dy = _r16(torch.randn(n, cout, ho, wo, generator=g) * 0.1)
    ref = ref_wgrad(x, dy, k, stride, pad)
    got = run_wgrad(x, dy, k, stride, pad, splits=splits, in_halo=max(pad, 1))
    err = float((got - ref).abs().max())
    assert err <= 2e-3 * max(1.0, float(ref.abs().max())), f"max err {err}, ref max {float(ref.abs().max())}"


@pytest.mark.parametrize("cin,cout,k,stride,hw,n,splits", [
    (256, 256, 3, 1, (25, 27), 2, 0),      # FPN output / RPN / mask-head 3x3
    (64, 128, 1, 1, (19, 23), 3, 1),       # one K unit: three of the workgroup's four waves idle; single split
    (256, 128, 1, 2, (26, 30), 2, 0),      # strided 1x1 (res3.0.conv1 shape)
    (128, 128, 3, 1, (14, 14), 5, 3),      # pixel count not a multiple of the 4-pixel step
    (512, 16, 1, 1, (13, 13), 2, 0),       # 16 output rows (the fused heads): three quarters of the channel tile masked
])
def test_conv_wgrad_f32_matches_autograd(gpu_required, cin, cout, k, stride, hw, n, splits):
    """Reference-precision weight gradient (conv_wgrad_f32_kernel: fp32 operands on v_mfma_f32_16x16x4_f32) against fp32 autograd:
    1e-5 of the largest entry (summation order only)."""
    g = torch.Generator().manual_seed(cin + cout + k + 1)
    h, w = hw
    pad = k // 2
    x = torch.randn(n, cin, h, w, generator=g)
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    dy = torch.randn(n, cout, ho, wo, generator=g) * 0.1
    scale = torch.rand(cout, generator=g) + 0.5 if k == 3 else None
    ref = ref_wgrad(x, dy, k, stride, pad)
    if scale is not None:
        ref = ref * scale[:, None, None, None]
    got = run_wgrad(x, dy, k, stride, pad, scale=scale, splits=splits, in_halo=max(pad, 1), f32=True)
    err = float((got - ref).abs().max())
    assert err <= 1e-5 * max(1.0, float(ref.abs().max())), f"max err {err}, ref max {float(ref.abs().max())}"


def test_conv_wgrad_scale_and_fc_shape(gpu_required):
    """Linear layer as a 1x1 conv over an (M x 1) image, FrozenBN-style per-channel scale applied to the gradient."""
    g = torch.Generator().manual_seed(3)
    m, kin, nout = 333, 1024, 256
    a = _r16(torch.randn(m, kin, generator=g))
    dy = _r16(torch.randn(m, nout, generator=g) * 0.05)
    scale = torch.rand(nout, generator=g) + 0.5
    ref = (dy.t() @ a) * scale[:, None]
    x4 = a.t().reshape(1, kin, m, 1)
    dy4 = dy.t().reshape(1, nout, m, 1)
    got = run_wgrad(x4, dy4, 1, 1, 0, scale=scale, in_halo=0, dy_halo=0)[:, :, 0, 0]
    err = float((got - ref).abs().max())
    assert err <= 2e-3 * max(1.0, float(ref.abs().max())), err


def _wt_flipped(w):
    """(Cout,Cin,k,k) -> transposed, tap-flipped GEMM weight [Cin][(k-1-i, k-1-j, co)] fp16, K padded to 64."""
    cout, cin, k, _ = w.shape
    t = w.flip(2, 3).permute(1, 2, 3, 0).reshape(cin, k * k * cout)      # [ci][(i', j', co)]
    kpad = (t.shape[1] + 63) // 64 * 64
    out = torch.zeros(cin, kpad)
    out[:, : t.shape[1]] = t
    return out.half().contiguous(), kpad


def run_dgrad(dy, w, hi, wi, stride, pad, res=None, res32=None, mask=None, down=None, variant=-1):
    lib = load_library()
    dev = torch.device("cuda:0")
    n, cout, ho, wo = dy.shape
    cin, k = w.shape[1], w.shape[2]
    halo = 1
    dyd = _halo(dy.permute(0, 2, 3, 1).half().contiguous(), halo).to(dev)
    wt, kpad = _wt_flipped(w)
    wtd = wt.to(dev)
    dxd = torch.zeros((n, hi + 2, wi + 2, cin), dtype=torch.float16, device=dev)

    def nhwc(t, dt):
        return _halo(t.permute(0, 2, 3, 1).to(dt).contiguous(), halo).to(dev) if t is not None else None
    rd, r32d, md, dd = nhwc(res, torch.float16), nhwc(res32, torch.float32), nhwc(mask, torch.float16), nhwc(down, torch.float16)
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    torch.cuda.synchronize()
    rc = lib.rs_op_conv2d_dgrad(ptr(dyd), ptr(wtd), ptr(dxd), ptr(rd), ptr(r32d), ptr(md), ptr(dd), n, hi, wi, cin, ho, wo, cout,
                                k, k, stride, pad, kpad, halo, variant, None)
    _check(lib, rc, "rs_op_conv2d_dgrad")
    torch.cuda.synchronize()
    o = dxd.cpu().float()
    inner = o[:, 1:-1, 1:-1]
    assert float(o.abs().sum()) == pytest.approx(float(inner.abs().sum()), rel=1e-6), "kernel wrote into the halo"
    return inner.permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("cin,cout,k,stride,hw,variant", [
    (256, 256, 3, 1, (25, 27), -1),     # 3x3 (FPN output, RPN conv, mask fcn, bottleneck conv2)
    (256, 256, 3, 1, (25, 27), 4),      # 256x256 tile of conv_igemm
    (256, 256, 3, 1, (25, 27), 12),     # conv_deep with the backward epilogue (the big 3x3 input gradients of a real step)
    (512, 128, 1, 1, (26, 30), -1),     # bottleneck conv1
    (128, 512, 1, 1, (26, 30), -1),     # bottleneck conv3
    (512, 256, 1, 2, (26, 30), -1),     # stride-2 1x1 (res4.0.conv1 / shortcut): scattered store
])
def test_conv_dgrad_with_fused_backward_epilogue(gpu_required, cin, cout, k, stride, hw, variant):
    """dx = relu'(y_prev) * (conv_transpose(dy) + identity-path gradient + fp32 RoIAlign scatter), vs autograd of
    relu(x) -> conv (+ the two extra consumers of x)."""
    g = torch.Generator().manual_seed(cin * 3 + cout + k + stride)
    hi, wi = hw
    n = 2
    pad = k // 2
    ho, wo = (hi + 2 * pad - k) // stride + 1, (wi + 2 * pad - k) // stride + 1
    w = _r16(torch.randn(cout, cin, k, k, generator=g) * 0.05)
    dy = _r16(torch.randn(n, cout, ho, wo, generator=g))
    pre = torch.randn(n, cin, hi, wi, generator=g)                     # pre-activation of the previous layer
    y_prev = _r16(F.relu(pre))                                         # its saved (post-ReLU) output = the mask
    res = _r16(torch.randn(n, cin, hi, wi, generator=g))               # gradient arriving over the identity shortcut
    res32 = torch.randn(n, cin, hi, wi, generator=g)                   # fp32 scatter target (RoIAlign backward)
    if stride > 1:                                                      # strided store: the caller owns the other positions
        keep = torch.zeros(1, 1, hi, wi)
        keep[:, :, ::stride, ::stride] = 1
        res, res32 = res * keep, res32 * keep
    # autograd reference: x = relu(pre); L = <conv(x), dy> + <x, res> + <x, res32>
    pre_r = pre.clone().requires_grad_(True)
    x = F.relu(pre_r)
    L = (F.conv2d(x, w, stride=stride, padding=pad) * dy).sum() + (x * res).sum() + (x * res32).sum()
    L.backward()
    ref = pre_r.grad * (y_prev > 0)          # fp16 rounding can flush a tiny positive activation to 0: use the stored mask
    got = run_dgrad(dy, w, hi, wi, stride, pad, res=res, res32=res32, mask=y_prev, variant=variant)
    if stride > 1:
        assert float((got * (1 - keep)).abs().max()) == 0.0
    err = float((got - ref).abs().max())
    assert err <= 3e-3 * max(1.0, float(ref.abs().max())), f"max err {err}, ref max {float(ref.abs().max())}"


def test_conv_dgrad_fpn_topdown_backward(gpu_required):
    """d(inner_l) = conv_transpose3x3(dP_l) + 2x2-sum of d(inner_{l-1}): backward of `lateral + nearest-upsample(top)`."""
    g = torch.Generator().manual_seed(11)
    n, c, h, w = 2, 256, 13, 15
    wt = _r16(torch.randn(c, c, 3, 3, generator=g) * 0.03)
    dP = _r16(torch.randn(n, c, h, w, generator=g))
    d_finer = _r16(torch.randn(n, c, 2 * h, 2 * w, generator=g))
    inner = torch.randn(n, c, h, w, generator=g).requires_grad_(True)
    L = (F.conv2d(inner, wt, padding=1) * dP).sum() + (F.interpolate(inner, scale_factor=2.0, mode="nearest") * d_finer).sum()
    L.backward()
    for variant in (-1, 12):                 # automatic tile choice; conv_deep's backward epilogue
        got = run_dgrad(dP, wt, h, w, 1, 1, down=d_finer, variant=variant)
        err = float((got - inner.grad).abs().max())
        assert err <= 3e-3 * max(1.0, float(inner.grad.abs().max())), (variant, err)


@pytest.mark.parametrize("P", [7, 14])
def test_roi_align_backward_is_the_adjoint_of_forward(gpu_required, P):
    """RoIAlign is linear in the feature maps, so its backward is pinned by <roi_align(F), G> == <F, roi_align_bwd(G)>
    for random F, G (forward kernel: parity-tested against the oracle in test_gpu_engine.py).  Boxes cover all four
    FPN levels, image borders (clamped / skipped samples) and an elongated one (per-sample fallback path)."""
    lib = load_library()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(P)
    n_img, rpi = 2, 12
    sizes = [(48, 56), (24, 28), (12, 14), (6, 7)]
    scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32]
    feats = [(torch.randn(n_img, h, w, 256, generator=g)).half() for h, w in sizes]
    fd = [_halo(f, 1).to(dev) for f in feats]
    W_img, H_img = 224.0, 192.0
    cx = torch.rand(n_img * rpi, generator=g) * W_img
    cy = torch.rand(n_img * rpi, generator=g) * H_img
    side = torch.tensor([20.0, 60.0, 130.0, 250.0, 500.0, 33.0] * (n_img * rpi // 6))
    bw, bh = side.clone(), side.clone()
    bw[3], bh[3] = 900.0, 6.0                 # elongated: window larger than the LDS table
    rois = torch.stack([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], 1).clamp(-20, 260).float().contiguous()
    rd = rois.to(dev)
    out = torch.zeros(n_img * rpi, P, P, 256, dtype=torch.float16, device=dev)
    G = (torch.randn(n_img * rpi, P, P, 256, generator=g) * 0.5).half()
    Gd = G.to(dev)
    dfd = [torch.zeros(n_img, h + 2, w + 2, 256, dtype=torch.float32, device=dev) for h, w in sizes]
    vp4 = C.c_void_p * 4
    hs = (C.c_int32 * 4)(*[h for h, _ in sizes])
    ws = (C.c_int32 * 4)(*[w for _, w in sizes])
    sc = (C.c_float * 4)(*scales)
    torch.cuda.synchronize()
    _check(lib, lib.rs_op_roi_align(vp4(*[f.data_ptr() for f in fd]), hs, ws, sc, 4, C.c_void_p(rd.data_ptr()), n_img * rpi, rpi, P, 0,
                                    C.c_void_p(out.data_ptr()), None, None), "rs_op_roi_align")
    _check(lib, lib.rs_op_roi_align_bwd(vp4(*[f.data_ptr() for f in dfd]), hs, ws, sc, 4, C.c_void_p(rd.data_ptr()), n_img * rpi, rpi, P, 0,
                                        C.c_void_p(Gd.data_ptr()), None), "rs_op_roi_align_bwd")
    torch.cuda.synchronize()
    lhs = float((out.double().cpu() * G.double()).sum())
    rhs = 0.0
    for f, d in zip(feats, dfd):
        dd = d.cpu().double()
        assert float(dd.abs().sum()) == pytest.approx(float(dd[:, 1:-1, 1:-1].abs().sum()), rel=1e-9), "gradient written into the halo"
        rhs += float((f.double() * dd[:, 1:-1, 1:-1]).sum())
    scale = float((out.double().cpu().abs() * G.double().abs()).sum())
    # forward output is rounded to fp16 (rel 5e-4 per element): the two inner products agree to that
    assert abs(lhs - rhs) <= 1e-3 * scale, (lhs, rhs, scale)
    assert sum(float(d.abs().sum()) for d in dfd) > 0


def test_roi_align_backward_owner_computes_equals_atomics_and_is_reproducible(gpu_required, monkeypatch):
    """The owner-computes backward (one workgroup per 8 x 8-cell region of a map walks the RoIs that reach it in entry order; plain
    stores) against the float-atomic kernel of rounds 1-2 (RS_ROI_BWD_ATOMIC=1) on 2 x 512 clustered, overlapping RoIs over all four
    levels, image borders and one elongated box (left to the atomic kernel in both): the same gradient maps up to the fp32 summation
    order (a RoI's contribution to a cell is computed the same way in both), nothing in the halo, the maps ACCUMULATE (a second call
    doubles them), and two runs of the new form agree BIT for bit -- the atomics did not."""
    lib = load_library()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(77)
    n_img, rpi, P = 2, 512, 7
    sizes = [(64, 72), (32, 36), (16, 18), (8, 9)]
    scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32]
    W_img, H_img = 288.0, 256.0
    centres = torch.rand(12, 2, generator=g) * torch.tensor([W_img, H_img])
    c = centres[torch.randint(0, 12, (n_img * rpi,), generator=g)] + torch.randn(n_img * rpi, 2, generator=g) * 6.0
    side = torch.tensor([16.0, 40.0, 90.0, 170.0, 300.0, 24.0, 60.0, 560.0])[torch.randint(0, 8, (n_img * rpi,), generator=g)]
    bw = side * (0.6 + 0.8 * torch.rand(n_img * rpi, generator=g))
    bh = side * (0.6 + 0.8 * torch.rand(n_img * rpi, generator=g))
    bw[5], bh[5] = 900.0, 5.0                   # elongated: window larger than the tables
    rois = torch.stack([c[:, 0] - bw / 2, c[:, 1] - bh / 2, c[:, 0] + bw / 2, c[:, 1] + bh / 2], 1).clamp(-30, 630).float().contiguous()
    rd = rois.to(dev)
    Gd = (torch.randn(n_img * rpi, P, P, 256, generator=g) * 0.5).half().to(dev)
    vp4 = C.c_void_p * 4
    hs = (C.c_int32 * 4)(*[h for h, _ in sizes])
    ws = (C.c_int32 * 4)(*[w for _, w in sizes])
    sc = (C.c_float * 4)(*scales)

    def run(calls=1, boxes=rd):
        dfd = [torch.zeros(n_img, h + 2, w + 2, 256, dtype=torch.float32, device=dev) for h, w in sizes]
        torch.cuda.synchronize()
        for _ in range(calls):
            _check(lib, lib.rs_op_roi_align_bwd(vp4(*[f.data_ptr() for f in dfd]), hs, ws, sc, 4, C.c_void_p(boxes.data_ptr()), n_img * rpi, rpi, P, 0,
                                                C.c_void_p(Gd.data_ptr()), None), "rs_op_roi_align_bwd")
        torch.cuda.synchronize()
        return [d.cpu() for d in dfd]

    monkeypatch.setenv("RS_ROI_BWD_ATOMIC", "1")
    ref = run()
    monkeypatch.setenv("RS_ROI_BWD_ATOMIC", "0")
    got = run()
    for l, (r, x) in enumerate(zip(ref, got)):
        assert float(r.abs().max()) > 0
        xd = x.double()
        assert float(xd.abs().sum()) == pytest.approx(float(xd[:, 1:-1, 1:-1].abs().sum()), rel=1e-12), "gradient written into the halo"
        err = float((x - r).abs().max())
        assert err <= 2e-5 * max(1.0, float(r.abs().max())), (l, err, float(r.abs().max()))
    # reproducibility and accumulation without the elongated box (its bins' windows overlap, and the atomic kernel it is left to adds them
    # in whatever order they arrive)
    plain = rois.clone()
    plain[5] = plain[6]
    pd_ = plain.to(dev)
    a, b, twice = run(boxes=pd_), run(boxes=pd_), run(2, boxes=pd_)
    for l, (x, y, t) in enumerate(zip(a, b, twice)):
        assert torch.equal(x, y), f"level {l}: two runs of the owner-computes form differ"
        assert float((t - 2 * x).abs().max()) <= 4e-5 * max(1.0, float(x.abs().max())), l


def test_deconv2x2_backward_through_conv_ops(gpu_required):
    """ConvTranspose2d(k=2, s=2) of the mask head: its input gradient is a 2x2 stride-2 convolution of dY (forward
    kernel, rs_op_conv2d) and its weight gradient is rs_op_conv2d_wgrad with the operands swapped (X as the 'output
    gradient', dY as the strided 'input').  Both against autograd of F.conv_transpose2d."""
    from tests.test_gpu_conv import run_conv
    g = torch.Generator().manual_seed(21)
    n, cin, cout, h, w = 5, 256, 256, 14, 14
    wt = _r16(torch.randn(cin, cout, 2, 2, generator=g) * 0.05)            # ConvTranspose2d weight (Cin, Cout, 2, 2)
    x = _r16(torch.randn(n, cin, h, w, generator=g))
    dy = _r16(torch.randn(n, cout, 2 * h, 2 * w, generator=g) * 0.2)
    xr = x.clone().requires_grad_(True)
    wr = wt.clone().requires_grad_(True)
    (F.conv_transpose2d(xr, wr, stride=2) * dy).sum().backward()
    # dgrad: conv2d(dy, W as (out=Cin, in=Cout, 2, 2), stride 2)
    got_dx = run_conv(dy, wt, torch.zeros(cin), stride=2, pad=0, in_halo=1, out_halo=1)
    err = float((got_dx - xr.grad).abs().max())
    assert err <= 3e-3 * max(1.0, float(xr.grad.abs().max())), err
    # wgrad: dW[ci][co][dy][dx] = sum X[ci](y,x) * dY[co](2y+dy, 2x+dx)
    got_dw = run_wgrad(dy, x, 2, 2, 0, in_halo=1, dy_halo=1)              # (cout'=Cin, cin'=Cout, 2, 2)
    err = float((got_dw - wr.grad).abs().max())
    assert err <= 3e-3 * max(1.0, float(wr.grad.abs().max())), err


def _dev(t):
    return t.contiguous().to(torch.device("cuda:0"))


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _get_deltas(src, tgt, w):
    """Box2BoxTransform.get_deltas ([EXT d2: modeling/box_regression.py])."""
    sw, sh = src[:, 2] - src[:, 0], src[:, 3] - src[:, 1]
    scx, scy = src[:, 0] + 0.5 * sw, src[:, 1] + 0.5 * sh
    tw, th = tgt[:, 2] - tgt[:, 0], tgt[:, 3] - tgt[:, 1]
    tcx, tcy = tgt[:, 0] + 0.5 * tw, tgt[:, 1] + 0.5 * th
    return torch.stack([w[0] * (tcx - scx) / sw, w[1] * (tcy - scy) / sh, w[2] * torch.log(tw / sw), w[3] * torch.log(th / sh)], 1)


def _rand_boxes(n, g, lo=0.0, hi=300.0):
    xy = torch.rand(n, 2, generator=g) * (hi - lo) + lo
    wh = torch.rand(n, 2, generator=g) * 80 + 4
    return torch.cat([xy, xy + wh], 1)


def test_rpn_loss_and_gradient(gpu_required):
    """RPN.losses restated with torch autograd: BCE-with-logits (sum) over sampled anchors + L1 (SMOOTH_L1_BETA 0) over
    positive anchors' deltas, both / (BATCH_SIZE_PER_IMAGE * N); two levels sharing one label/anchor table."""
    lib = load_library()
    g = torch.Generator().manual_seed(31)
    N, A, cs = 2, 3, 16
    hws = [35, 12]
    total = sum(hw * A for hw in hws)
    anchors = _rand_boxes(total, g)
    matched = _rand_boxes(N * total, g).view(N, total, 4)
    labels = torch.randint(-1, 2, (N, total), generator=g, dtype=torch.int32)
    heads = [torch.randn(N, hw, cs, generator=g) for hw in hws]
    normalizer = 256.0 * N
    scale = 1024.0
    # reference
    hr = [h.clone().requires_grad_(True) for h in heads]
    logits = torch.cat([h[:, :, :A].reshape(N, -1) for h in hr], 1)
    deltas = torch.cat([h[:, :, A:5 * A].reshape(N, -1, 4) for h in hr], 1)
    valid, pos = labels >= 0, labels == 1
    l_cls = F.binary_cross_entropy_with_logits(logits[valid], labels[valid].float(), reduction="sum") / normalizer
    tgt = torch.stack([_get_deltas(anchors, matched[i], (1, 1, 1, 1)) for i in range(N)])
    l_loc = (deltas[pos] - tgt[pos]).abs().sum() / normalizer
    (l_cls + l_loc).backward()
    # engine
    loss = _dev(torch.zeros(2))
    ad, md, ld = _dev(anchors), _dev(matched), _dev(labels)
    off = 0
    for h, hgrad, hw in zip(heads, hr, hws):
        hd = _dev(h)
        dh = torch.full((N, hw, cs), float("nan"), dtype=torch.float16, device=hd.device)
        _check(lib, lib.rs_op_rpn_loss(_ptr(hd), _ptr(dh), _ptr(ld), _ptr(ad), _ptr(md), _ptr(loss), N, hw, A, cs, off, total,
                                       normalizer, scale, None), "rs_op_rpn_loss")
        torch.cuda.synchronize()
        got = dh.cpu().float() / scale
        ref = hgrad.grad
        assert float(got[:, :, 5 * A:].abs().max()) == 0.0                      # padding columns get zero gradient
        assert float((got[:, :, :5 * A] - ref[:, :, :5 * A]).abs().max()) <= 2e-3 * float(ref.abs().max())
        off += hw * A
    lo = loss.cpu()
    assert abs(float(lo[0]) - float(l_cls)) <= 1e-5 * max(1.0, float(l_cls)) and abs(float(lo[1]) - float(l_loc)) <= 1e-5 * max(1.0, float(l_loc))


def test_box_loss_and_gradient(gpu_required):
    """FastRCNNOutputLayers.losses: cross_entropy(mean) + class-specific L1 on foreground rows / number of sampled RoIs."""
    lib = load_library()
    g = torch.Generator().manual_seed(32)
    R, K, cs = 700, 2, 16
    pred = torch.randn(R, cs, generator=g)
    cls = torch.randint(0, K + 1, (R,), generator=g, dtype=torch.int32)
    cls[::50] = -1                                                   # empty slots of a fixed-capacity buffer
    props, gts = _rand_boxes(R, g), _rand_boxes(R, g)
    w = (10.0, 10.0, 5.0, 5.0)
    valid = cls >= 0
    n_valid = float(valid.sum())
    pr = pred.clone().requires_grad_(True)
    l_cls = F.cross_entropy(pr[valid][:, :K + 1], cls[valid].long(), reduction="mean")
    fg = valid & (cls < K)
    fg_idx = torch.nonzero(fg)[:, 0]
    d = pr[:, K + 1:K + 1 + 4 * K].view(R, K, 4)[fg_idx, cls[fg_idx].long()]
    l_reg = (d - _get_deltas(props[fg_idx], gts[fg_idx], w)).abs().sum() / n_valid
    (l_cls + l_reg).backward()
    loss = _dev(torch.zeros(2))
    pd = _dev(pred)
    dp = torch.full((R, cs), float("nan"), dtype=torch.float16, device=pd.device)
    scale = 512.0
    cd, prd, gd = _dev(cls), _dev(props), _dev(gts)            # keep the device buffers alive across the launch
    _check(lib, lib.rs_op_box_loss(_ptr(pd), _ptr(dp), _ptr(cd), _ptr(prd), _ptr(gd), _ptr(loss), R, K, cs, n_valid,
                                   (C.c_float * 4)(*w), scale, None), "rs_op_box_loss")
    torch.cuda.synchronize()
    got = dp.cpu().float() / scale
    assert float((got - pr.grad).abs().max()) <= 2e-3 * float(pr.grad.abs().max())
    lo = loss.cpu()
    assert abs(float(lo[0]) - float(l_cls)) <= 1e-5 * float(l_cls) + 1e-6 and abs(float(lo[1]) - float(l_reg)) <= 1e-5 * float(l_reg) + 1e-6


def test_mask_loss_and_gradient(gpu_required):
    lib = load_library()
    g = torch.Generator().manual_seed(33)
    M, S, cs, K = 37, 28, 16, 2
    logits = torch.randn(M, S * S, cs, generator=g) * 2
    tgt = (torch.rand(M, S * S, generator=g) > 0.5).to(torch.uint8)
    cls = torch.randint(0, K, (M,), generator=g, dtype=torch.int32)
    lr = logits.clone().requires_grad_(True)
    sel = lr[torch.arange(M), :, cls.long()]
    l = F.binary_cross_entropy_with_logits(sel, tgt.float(), reduction="mean")
    l.backward()
    loss = _dev(torch.zeros(1))
    ld = _dev(logits)
    dl = torch.full((M, S * S, cs), float("nan"), dtype=torch.float16, device=ld.device)
    scale = 4096.0
    td, cd = _dev(tgt), _dev(cls)
    _check(lib, lib.rs_op_mask_loss(_ptr(ld), _ptr(dl), _ptr(td), _ptr(cd), _ptr(loss), M, S, cs, scale, None), "rs_op_mask_loss")
    torch.cuda.synchronize()
    got = dl.cpu().float() / scale
    assert float((got - lr.grad).abs().max()) <= 2e-3 * float(lr.grad.abs().max())
    assert abs(float(loss.cpu()[0]) - float(l)) <= 1e-5 * float(l)


def test_sgd_momentum_equals_torch_optim(gpu_required):
    """Three steps of torch.optim.SGD(lr, momentum 0.9, weight_decay 1e-4) on a flat tensor, gradients carrying a loss scale."""
    lib = load_library()
    g = torch.Generator().manual_seed(34)
    n = 10007
    w0 = torch.randn(n, generator=g)
    grads = [torch.randn(n, generator=g) * 0.01 for _ in range(3)]
    wt = w0.clone().requires_grad_(True)
    opt = torch.optim.SGD([wt], lr=0.01, momentum=0.9, weight_decay=1e-4)
    wd, buf = _dev(w0), _dev(torch.zeros(n))
    scale = 256.0
    for i, gr in enumerate(grads):
        wt.grad = gr.clone()
        opt.step()
        gd = _dev(gr * scale)
        _check(lib, lib.rs_op_sgd_momentum(_ptr(wd), _ptr(buf), _ptr(gd), n, 0.01, 0.9, 1e-4, 1.0 / scale, int(i == 0), None), "sgd")
        torch.cuda.synchronize()
    assert float((wd.cpu() - wt.detach()).abs().max()) <= 1e-6


def test_fold_weights_layouts(gpu_required):
    """fp32 master [Cout][(kh,kw,ci)] -> fp16 forward weight with the FrozenBN scale folded and the transposed, tap-flipped
    copy rs_op_conv2d_dgrad consumes."""
    lib = load_library()
    g = torch.Generator().manual_seed(35)
    cout, cin, k = 128, 64, 3
    w = torch.randn(cout, cin, k, k, generator=g) * 0.1
    scale = torch.rand(cout, generator=g) + 0.5
    master = w.permute(0, 2, 3, 1).reshape(cout, k * k * cin).contiguous()
    kpad, kpad_t = k * k * cin, k * k * cout
    md = _dev(master)
    fwd = torch.zeros(cout, kpad, dtype=torch.float16, device=md.device)
    bwd = torch.zeros(cin, kpad_t, dtype=torch.float16, device=md.device)
    sd = _dev(scale)
    _check(lib, lib.rs_op_fold_weights(_ptr(md), _ptr(sd), _ptr(fwd), _ptr(bwd), cout, cin, k, k, kpad, kpad_t, None), "fold")
    torch.cuda.synchronize()
    folded = w * scale[:, None, None, None]
    want_f = folded.permute(0, 2, 3, 1).reshape(cout, kpad).half()
    want_b, _ = _wt_flipped(folded)
    df = (fwd.cpu().float() - want_f.float()).abs()
    db = (bwd.cpu().float() - want_b[:, :kpad_t].float()).abs()
    assert float(df.max()) == 0.0 and float(db.max()) == 0.0, (float(df.max()), int((df > 0).sum()), float(db.max()), int((db > 0).sum()))


def test_matcher_equals_oracle(gpu_required):
    """rs_op_match vs oracle/train_oracle.py (Matcher on pairwise_iou): RPN form (0.3/0.7, labels 0/-1/1, low-quality
    matches, anchors shared by the images) and ROI-heads form (0.5, per-image proposal lists with counts) -- exact."""
    from oracle import train_oracle as T
    from oracle import maskrcnn_oracle as O
    from proj_roadsurf_amd.spec import EngineSpec
    lib = load_library()
    g = torch.Generator().manual_seed(41)
    spec = EngineSpec()
    anchors = torch.cat([O.grid_anchors(spec, l, hw, hw) for l, hw in enumerate([40, 20, 10, 5, 3])])
    A = anchors.shape[0]
    N, cap = 3, 16
    gts = [_rand_boxes(k, g, 0, 120) for k in (5, 0, 11)]
    gt = torch.zeros(N, cap, 4)
    for i, b in enumerate(gts):
        gt[i, : b.shape[0]] = b
    cnt = torch.tensor([b.shape[0] for b in gts], dtype=torch.int32)
    ad, gd, cd = _dev(anchors), _dev(gt), _dev(cnt)
    m = torch.empty(N, A, dtype=torch.int32, device=ad.device)
    l = torch.empty(N, A, dtype=torch.int32, device=ad.device)
    _check(lib, lib.rs_op_match(_ptr(ad), 0, None, _ptr(gd), _ptr(cd), _ptr(m), _ptr(l), None, N, A, cap, 0.3, 0.7, 0, -1, 1, 1, None), "match")
    torch.cuda.synchronize()
    for i in range(N):
        wm, wl = T.matcher(T.pairwise_iou(gts[i], anchors), [0.3, 0.7], [0, -1, 1], True)
        assert torch.equal(l[i].cpu(), wl.to(torch.int32)), f"image {i}: labels"
        assert torch.equal(m[i].cpu(), wm.to(torch.int32)), f"image {i}: matched gt"
    assert int((l[0] == 1).sum()) >= 5                      # every gt has at least its low-quality match
    # ROI heads: per-image proposals (+ appended gt), threshold 0.5, no low-quality matches
    P = 300
    props = torch.stack([torch.cat([_rand_boxes(P - gts[i].shape[0], g, 0, 120), gts[i]]) for i in range(N)])
    pc = torch.tensor([P, P - 40, P], dtype=torch.int32)
    pd, pcd = _dev(props), _dev(pc)
    m2 = torch.empty(N, P, dtype=torch.int32, device=ad.device)
    l2 = torch.empty(N, P, dtype=torch.int32, device=ad.device)
    _check(lib, lib.rs_op_match(_ptr(pd), 1, _ptr(pcd), _ptr(gd), _ptr(cd), _ptr(m2), _ptr(l2), None, N, P, cap, 0.5, 0.5, 0, 1, 1, 0, None), "match")
    torch.cuda.synchronize()
    for i in range(N):
        k = int(pc[i])
        wm, wl = T.matcher(T.pairwise_iou(gts[i], props[i, :k]), [0.5], [0, 1], False)
        assert torch.equal(l2[i, :k].cpu(), wl.to(torch.int32)) and torch.equal(m2[i, :k].cpu(), wm.to(torch.int32))
        assert bool((l2[i, k:] == -1).all())


def test_subsample_is_a_valid_uniform_sample(gpu_required):
    """rs_op_subsample: quotas of subsample_labels (R:223,246 RPN; R:178,192 ROI heads), members drawn only from the right
    group, deterministic per seed, different per seed, and every candidate equally likely (chi-square-ish bound)."""
    lib = load_library()
    g = torch.Generator().manual_seed(42)
    N, n = 2, 20000
    base = torch.zeros(N, n, dtype=torch.int32)
    base[0, torch.randperm(n, generator=g)[:400]] = 1
    base[0, torch.randperm(n, generator=g)[:3000]] = -1
    base[1, torch.randperm(n, generator=g)[:30]] = 1              # fewer positives than the quota
    dev = torch.device("cuda:0")
    outs = []
    for seed in (1, 1, 2):
        lab = base.clone().to(dev)
        cnt = torch.zeros(N, 2, dtype=torch.int32, device=dev)
        _check(lib, lib.rs_op_subsample(_ptr(lab), None, _ptr(cnt), N, n, 256, 0.5, 0, 1, seed, None), "subsample")
        torch.cuda.synchronize()
        outs.append((lab.cpu(), cnt.cpu()))
    (a, ca), (b, _), (c, _) = outs
    assert torch.equal(a, b) and not torch.equal(a, c)
    for i, (want_pos, want_neg) in enumerate([(128, 128), (int((base[1] == 1).sum()), 256 - int((base[1] == 1).sum()))]):
        assert int((a[i] == 1).sum()) == want_pos == int(ca[i, 0]) and int((a[i] == 0).sum()) == want_neg == int(ca[i, 1])
        assert bool((base[i][a[i] == 1] == 1).all()) and bool((base[i][a[i] == 0] == 0).all())
    # ROI mode: index list, foreground first, ascending inside each group; class labels (bg = 2)
    n2 = 1100
    cls = torch.full((1, n2), 2, dtype=torch.int32)
    cls[0, torch.randperm(n2, generator=g)[:90]] = torch.randint(0, 2, (90,), generator=g, dtype=torch.int32)
    cls[0, torch.randperm(n2, generator=g)[:50]] = -1
    hits = torch.zeros(n2)
    trials = 200
    for seed in range(trials):
        lab = cls.clone().to(dev)
        samp = torch.empty(1, 512, dtype=torch.int32, device=dev)
        cnt = torch.zeros(1, 2, dtype=torch.int32, device=dev)
        _check(lib, lib.rs_op_subsample(_ptr(lab), _ptr(samp), _ptr(cnt), 1, n2, 512, 0.25, 2, 0, 1000 + seed, None), "subsample")
        torch.cuda.synchronize()
        s_, c_ = samp.cpu()[0], cnt.cpu()[0]
        nfg = int(((cls[0] != 2) & (cls[0] != -1)).sum())
        npos, nneg = int(c_[0]), int(c_[1])
        assert npos == min(nfg, 128) and nneg == min(int((cls[0] == 2).sum()), 512 - npos)
        fg, bg = s_[:npos], s_[npos:npos + nneg]
        assert bool((cls[0][fg.long()] < 2).all()) and bool((cls[0][fg.long()] >= 0).all()) and bool((cls[0][bg.long()] == 2).all())
        assert bool((fg[1:] > fg[:-1]).all()) and bool((bg[1:] > bg[:-1]).all()) and bool((s_[npos + nneg:] == -1).all())
        assert torch.equal(lab.cpu(), cls)                          # roi mode leaves the labels alone
        hits[bg.long()] += 1
    bgmask = cls[0] == 2
    p_sel = (512 - min(nfg, 128)) / int(bgmask.sum())
    z = (hits[bgmask] - trials * p_sel) / math.sqrt(trials * p_sel * (1 - p_sel))
    assert float(z.abs().max()) < 5.0 and abs(float(z.mean())) < 0.2


@pytest.mark.parametrize("cap", [1024, 2048])
def test_nms_operator_both_capacities(gpu_required, cap):
    """rs_op_nms == the oracle's greedy NMS (torchvision semantics, IoU > thr suppresses), exactly: capacity 1024 (inference, mask in
    LDS) and 2048 (training, PRE_NMS_TOPK_TRAIN 2000, mask in global scratch); full, ragged, tiny and empty segments, a validity
    mask, heavily overlapping boxes."""
    from oracle import maskrcnn_oracle as O
    lib = load_library()
    g = torch.Generator().manual_seed(cap)
    counts = [cap, cap - 37, 700, 65, 1, 0]
    S = len(counts)
    boxes = torch.zeros(S, cap, 4)
    valid = torch.ones(S, cap, dtype=torch.uint8)
    for s_, c in enumerate(counts):
        ctr = torch.rand(c, 2, generator=g) * 300                 # dense: many overlaps around the threshold
        wh = torch.rand(c, 2, generator=g) * 80 + 20
        boxes[s_, :c] = torch.cat([ctr - wh / 2, ctr + wh / 2], 1)
    valid[1, ::7] = 0
    bd, cd, vd = _dev(boxes), _dev(torch.tensor(counts, dtype=torch.int32)), _dev(valid)
    keep = torch.full((S, cap), 7, dtype=torch.uint8, device=bd.device)
    _check(lib, lib.rs_op_nms(_ptr(bd), _ptr(cd), _ptr(vd), _ptr(keep), S, cap, 0.7, None), "rs_op_nms")
    torch.cuda.synchronize()
    k = keep.cpu().numpy()
    for s_, c in enumerate(counts):
        b = boxes[s_, :c].numpy()
        v = valid[s_, :c].numpy().astype(bool)
        want = np.zeros(c, bool)
        want[np.nonzero(v)[0][O.nms_sorted_np(b[v], 0.7)]] = True
        assert np.array_equal(k[s_, :c].astype(bool), want), (cap, s_, c)
        assert (k[s_, c:] == 0).all()
