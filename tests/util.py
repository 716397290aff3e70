"""Shared helpers for the parity tests (synthetic tiles, detection matching)."""
import numpy as np


from proj_roadsurf_amd.synthetic import synthetic_scenes, synthetic_tiles  # noqa: F401  (one definition, in the package)


from proj_roadsurf_amd.matching import box_iou, match_detections, wilson_lower  # noqa: F401  (one definition, in the package)


def conv_stage_shapes(spec, net_h=800, net_w=800):
    """(stage name, output pixels per tile, cin, k, cout, cin2, forced variant or None) of every GEMM stage of the fp16
    inference engine, in execution order -- the host-side mirror of csrc/engine.hip rs_engine::build() that the
    tile-dispatch tests enumerate.  Output pixels per tile times the batch size is the GEMM's M."""
    out = []
    h2, w2, h4, w4 = net_h // 2, net_w // 2, net_h // 4, net_w // 4
    out.append(("stem.conv1+maxpool", h2 * w2, 4, 7, spec.stem_out_channels, 0, 21))     # csrc/stem_fused.hip
    cur_c, bott, cout, ch, cw = spec.stem_out_channels, 64, spec.res2_out_channels, h4, w4
    have_t1 = False
    for si, nb in enumerate(spec.res_blocks):
        for bi in range(nb):
            nm = f"res{si + 2}.{bi}"
            stride = 2 if (bi == 0 and si > 0) else 1
            s1 = stride if spec.stride_in_1x1 else 1
            oh, ow = ch // stride, cw // stride
            proj = cur_c != cout
            tail = stride == 1 and ((bott == 64 and cout == 256) or (bott == 128 and cout == 512 and bi > 0))    # fused tails (csrc/bneck_fused.hip), variant 13
            if not have_t1:
                out.append((nm + ".conv1", (ch // s1) * (cw // s1), cur_c, 1, bott, 0, None))
            have_t1 = False
            if tail:
                have_t1 = bi + 1 < nb
                out.append((nm + (".conv2+conv3+next.conv1" if have_t1 else ".conv2+conv3"), oh * ow, bott, 3, cout, 0, 13))
            else:
                out.append((nm + ".conv2", oh * ow, bott, 3, bott, 0, None))
                out.append((nm + ".conv3", oh * ow, bott, 1, cout, cur_c if proj else 0, None))
            cur_c, ch, cw = cout, oh, ow
        bott *= 2
        cout *= 2
    sizes = [(h4 >> l, w4 >> l) for l in range(4)]
    res_c = [spec.res2_out_channels * (2 ** i) for i in range(4)]
    for l in (3, 2, 1, 0):
        out.append((f"fpn_lateral{l + 2}", sizes[l][0] * sizes[l][1], res_c[l], 1, 256, 0, None))
    # the 3x3 output convolutions of the four levels, and the shared RPN 3x3 over the five, run as ONE multi-map conv_deep launch
    # each (launch_conv_deep_multi; variant 12 at every batch size)
    out.append(("fpn_output2-5", sum(a * b for a, b in sizes), 256, 3, 256, 0, 12))
    p6 = ((sizes[3][0] - 1) // 2 + 1, (sizes[3][1] - 1) // 2 + 1)
    # ... with the 16-row objectness + delta head inside its epilogue (ConvParams::head_w): no rpn.heads stages
    out.append(("rpn.conv+heads2-6", sum(a * b for a, b in sizes + [p6]), 256, 3, 256, 0, 12))
    pr, fc = spec.box_pooler_resolution, spec.box_fc_dim
    out.append(("box.fc1", 1024, pr * pr * 256, 1, fc, 0, None))
    out.append(("box.fc2", 1024, fc, 1, fc, 0, None))
    out.append(("box.predictor", 1024, fc, 1, 16, 0, 2))
    if spec.mask_on:
        mr, d = spec.mask_pooler_resolution, spec.detections_per_image
        for i in range(spec.mask_num_conv):
            out.append((f"mask.fcn{i + 1}", d * mr * mr, 256, 3, 256, 0, None))
        out.append(("mask.deconv_predict", d * mr * mr, 256, 1, 256, 0, 22))     # csrc/conv_wreg.hip, EPI 3
    return out
