"""Shared helpers for the parity tests (synthetic tiles, detection matching)."""
import numpy as np


from proj_roadsurf_amd.synthetic import synthetic_scenes, synthetic_tiles  # noqa: F401  (one definition, in the package)


def box_iou(a, b):
    """IoU matrix (len(a), len(b)) of XYXY boxes."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    if len(a) == 0 or len(b) == 0:
        return np.zeros((len(a), len(b)))
    ix = np.maximum(0, np.minimum(a[:, None, 2], b[None, :, 2]) - np.maximum(a[:, None, 0], b[None, :, 0]))
    iy = np.maximum(0, np.minimum(a[:, None, 3], b[None, :, 3]) - np.maximum(a[:, None, 1], b[None, :, 1]))
    inter = ix * iy
    aa = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]); ab = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / np.maximum(aa[:, None] + ab[None, :] - inter, 1e-12)


def match_detections(ref, got, min_score=0.1, iou_thr=0.95):
    """Greedy one-to-one matching (same class, box IoU >= thr) of reference detections with score >= min_score.
    ref/got: dicts with boxes (n,4), scores (n,), classes (n,), optional masks (n,H,W) bool.
    Returns dict(frac_matched, max_dscore, min_mask_iou (masks >= 100 px), agg_mask_iou (sum inter / sum union), n_ref)."""
    rb, rs, rc = np.asarray(ref["boxes"]), np.asarray(ref["scores"]), np.asarray(ref["classes"])
    gb, gs, gc = np.asarray(got["boxes"]), np.asarray(got["scores"]), np.asarray(got["classes"])
    sel = np.where(rs >= min_score)[0]
    iou = box_iou(rb, gb)
    used = set()
    matched, dscore, miou = 0, 0.0, 1.0
    inter_sum, union_sum = 0, 0
    dbox = 0.0
    for i in sel:
        cand = [(iou[i, j], j) for j in range(len(gb)) if j not in used and gc[j] == rc[i] and iou[i, j] >= iou_thr]
        if not cand:
            continue
        _, j = max(cand)
        used.add(j)
        matched += 1
        dscore = max(dscore, abs(float(rs[i]) - float(gs[j])))
        dbox = max(dbox, float(np.abs(rb[i] - gb[j]).max()))
        if "masks" in ref and "masks" in got:
            a, b = np.asarray(ref["masks"][i], bool), np.asarray(got["masks"][j], bool)
            u = np.logical_or(a, b).sum()
            it = np.logical_and(a, b).sum()
            inter_sum += it
            union_sum += u
            if u >= 100:     # a 1-pixel flip on a 3-pixel mask is not a meaningful IoU
                miou = min(miou, it / u)
    n = len(sel)
    return {"frac_matched": matched / n if n else 1.0, "max_dscore": dscore, "min_mask_iou": miou, "n_ref": n,
            "agg_mask_iou": (inter_sum / union_sum) if union_sum else 1.0, "max_dbox": dbox}


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
        out.append(("mask.deconv_predict", d * mr * mr, 256, 1, 256, 0, 14))
    return out
