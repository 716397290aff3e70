"""GPU parity of the training engine's assembled backward (rs_trainer_*, through the C ABI) against torch autograd on
the training oracle's differentiable forward (oracle/train_oracle.py + oracle/maskrcnn_oracle.py).

Round 1 scope: the trunk.  Random gradients are injected at p2..p6, the engine runs FPN + res5..res3 backward (57
weight-gradient GEMMs, 50 input-gradient convolutions with fused ReLU masks / identity adds / strided scatters / 2x2
down-sums) and every trainable trunk tensor's gradient is compared with autograd's (tolerances and measured values next
to the asserts)."""
import numpy as np
import pytest
import torch

from proj_roadsurf_amd.engine import Trainer
from proj_roadsurf_amd.spec import EngineSpec
from proj_roadsurf_amd.weights import conv_layers, synthetic_weights, trainable_layers
from tests.util import synthetic_tiles

pytestmark = pytest.mark.gpu


def _ohwi32(w):
    """(Cout,Cin,kh,kw) torch -> engine GEMM layout (Cout, kh*kw*Cin) fp32."""
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).numpy()


@pytest.fixture(scope="module")
def trunk(gpu_required):
    from oracle import maskrcnn_oracle as O
    from oracle import train_oracle as T
    spec = EngineSpec(num_classes=2, min_size_test=320, max_size_test=533, rpn_pre_nms_topk_test=300, rpn_post_nms_topk_test=300)
    Wn = synthetic_weights(spec, seed=0)
    tiles = synthetic_tiles(2, 256, 256, 3, seed=123)
    tr = Trainer(spec, Wn, (256, 256, 3), batch=2, loss_scale=1.0)
    tr.forward_trunk(tr.upload_tiles(tiles), 2)
    g = torch.Generator().manual_seed(5)
    dP = {}
    for l, hw in zip(range(2, 7), (80, 40, 20, 10, 5)):
        d = (torch.randn(2, hw, hw, 256, generator=g) * 0.02).half()
        dP[l] = d
        tr.set_tensor(f"d:p{l}", d.numpy())
    tr.backward_trunk(2)
    tr.sync()
    # oracle: same network input, autograd through resnet + FPN, L = sum_l <p_l, dP_l>
    W = {k: torch.as_tensor(np.asarray(v), dtype=torch.float32) for k, v in Wn.items()}
    keys = [k for k in T.trainable_keys(W) if k.startswith("backbone.")]
    for k in keys:
        W[k].requires_grad_(True)
    x = torch.from_numpy(tr.tensor("net_input", engine=True)[..., :3].astype(np.float32)).permute(0, 3, 1, 2)
    feats = O.resnet_forward(spec, W, x)
    feats.update(O.fpn_forward(spec, W, feats))
    L = sum((feats[f"p{l}"] * dP[l].float().permute(0, 3, 1, 2)).sum() for l in range(2, 7))
    L.backward()
    yield spec, tr, W, feats
    tr.close()


def test_trunk_forward_matches_oracle(trunk):
    spec, tr, W, feats = trunk
    for name in ["res3", "res5", "p2", "p5", "p6"]:
        got = torch.from_numpy(tr.tensor(name, engine=True).astype(np.float32)).permute(0, 3, 1, 2)
        rel = float((got - feats[name].detach()).norm() / feats[name].detach().norm())
        assert rel <= 1.5e-2, (name, rel)


def test_trunk_weight_gradients_match_autograd(trunk):
    spec, tr, W, feats = trunk
    worst = {}
    names = [n for n in trainable_layers(spec) if n.startswith("backbone.")]
    assert len(names) == 4 * 3 + 1 + 6 * 3 + 1 + 3 * 3 + 1 + 8            # res3..res5 convs + shortcuts, 8 FPN convs
    for n in names:
        got = tr.tensor(f"g:{n}.w")
        ref = _ohwi32(W[n + ".weight"].grad)
        assert got.shape == ref.shape, (n, got.shape, ref.shape)
        rel = float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-20))
        cos = float((got * ref).sum() / max(np.linalg.norm(got) * np.linalg.norm(ref), 1e-30))
        worst[n] = (rel, cos)
        assert np.isfinite(got).all()
    for n, (rel, cos) in worst.items():
        print(f"{n:45s} rel L2 err {rel:.4f}  cosine {cos:.5f}")
    rels = np.array([v[0] for v in worst.values()])
    # Measured on MI355X (round 1): the 8 FPN layers 0.06-0.11 % (their inputs are one or two fp16 roundings away from
    # the oracle's), res5 2-4.5 %, res4 3-6 %, res3 5-6.5 %, cosine >= 0.9979 everywhere.  The growth follows the
    # FORWARD activation error of the fp16 engine vs the fp32 oracle (<= 1.5 % rel. L2 per map, test above), which enters
    # every weight gradient through X and every ReLU mask; there is no jump at any layer boundary.
    assert float(np.median(rels)) <= 6e-2 and float(rels.max()) <= 8e-2, (float(np.median(rels)), float(rels.max()))
    assert min(v[1] for v in worst.values()) >= 0.997
    fpn = [v[0] for n, v in worst.items() if ".fpn_" in n]
    assert max(fpn) <= 3e-3


def test_trunk_bias_gradients_match_autograd(trunk):
    spec, tr, W, feats = trunk
    for l in range(2, 6):
        for kind in ("lateral", "output"):
            n = f"backbone.fpn_{kind}{l}"
            got = tr.tensor(f"g:{n}.b")
            ref = W[n + ".bias"].grad.numpy()
            rel = float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-20))
            assert rel <= 2e-2, (n, rel)


def test_frozen_layers_have_no_gradient_tensors(trunk):
    spec, tr, W, feats = trunk
    names = tr.tensor_names()
    assert not any(".stem." in n or ".res2." in n for n in names)
    assert "g:backbone.bottom_up.res3.0.conv1.w" in names and "d:res2" not in names
    assert tr.param_count >= sum(int(np.prod(W[k].shape)) for k in W if k.startswith("backbone.") and W[k].requires_grad)


def test_sgd_step_updates_master_and_forward_weights(trunk):
    """One torch.optim.SGD step (momentum 0.9, wd 1e-4) on the trunk parameters: master weights move exactly as
    torch's, and the refolded fp16 forward weight equals half(master * bn_scale)."""
    spec, tr, W, feats = trunk
    n = "backbone.bottom_up.res4.1.conv2"
    g = tr.tensor(f"g:{n}.w").copy()
    m0 = tr.tensor(f"m:{n}.w").copy()
    lr = 1e-7                                 # the injected p-gradients are not a real loss: keep the step tiny
    tr.apply_sgd(lr, 0.9, 1e-4)
    tr.sync()
    m1 = tr.tensor(f"m:{n}.w")
    want = m0 - np.float32(lr) * (g + np.float32(1e-4) * m0)
    assert np.allclose(m1, want, rtol=1e-6, atol=1e-9) and float(np.abs(m1 - m0).max()) > 0
    scale = (W[n + ".norm.weight"] / torch.sqrt(W[n + ".norm.running_var"] + spec.bn_eps)).detach().numpy()
    tiles = synthetic_tiles(2, 256, 256, 3, seed=123)
    tr.forward_trunk(tr.upload_tiles(tiles), 2)
    tr.sync()
    p2 = tr.tensor("p2", engine=True).astype(np.float32)
    assert np.isfinite(p2).all()
    ref = feats["p2"].detach().permute(0, 2, 3, 1).numpy()
    assert 0 < float(np.abs(p2 - ref).max())                       # the step changed the network
    assert float(np.linalg.norm(p2 - ref) / np.linalg.norm(ref)) < 0.1 and scale.shape[0] == m1.shape[0]


def test_rpn_losses_and_full_backward_match_autograd(gpu_required):
    """RPN training step end to end: engine-side Matcher + sampler pick the anchors, then loss_rpn_cls / loss_rpn_loc and
    the gradients of the RPN head AND (through d:p2..d:p6 and the trunk backward) of the backbone are compared with
    autograd of the oracle evaluated on the SAME sampled anchors (loss scale 256 on the engine side)."""
    from oracle import maskrcnn_oracle as O
    from oracle import train_oracle as T
    spec = EngineSpec(num_classes=2, min_size_test=320, max_size_test=533, rpn_pre_nms_topk_test=300, rpn_post_nms_topk_test=300)
    Wn = synthetic_weights(spec, seed=0)
    tiles = synthetic_tiles(2, 256, 256, 3, seed=321)
    scale = 256.0
    tr = Trainer(spec, Wn, (256, 256, 3), batch=2, loss_scale=scale)
    try:
        gt_boxes = [np.array([[20.0, 30.0, 120.0, 160.0], [150.0, 40.0, 300.0, 130.0], [60.0, 200.0, 110.0, 260.0]], np.float32),
                    np.array([[100.0, 100.0, 260.0, 280.0]], np.float32)]
        gt_classes = [np.array([0, 1, 0]), np.array([1])]
        tr.set_targets(gt_boxes, gt_classes)
        tr.forward_trunk(tr.upload_tiles(tiles), 2)
        tr.rpn_step(2, seed=7)
        tr.backward_trunk(2)
        tr.sync()
        labels = torch.from_numpy(tr.tensor("rpn_labels").astype(np.int64))
        matched = torch.from_numpy(tr.tensor("rpn_matched").astype(np.int64))
        losses = tr.tensor("losses")
        # engine-side targets are a valid sample of the oracle's Matcher output
        anchors = torch.cat([O.grid_anchors(spec, l, hw, hw) for l, hw in enumerate((80, 40, 20, 10, 5))])
        assert np.allclose(tr.tensor("anchors"), anchors.numpy())
        for i in range(2):
            gt = torch.from_numpy(gt_boxes[i])
            m, lab = T.matcher(T.pairwise_iou(gt, anchors), [0.3, 0.7], [0, -1, 1], True)
            assert torch.equal(m, matched[i])
            assert bool((lab[labels[i] == 1] == 1).all()) and bool((lab[labels[i] == 0] == 0).all())
            npos = int((labels[i] == 1).sum())
            assert npos == min(int((lab == 1).sum()), 128) and int((labels[i] == 0).sum()) == 256 - npos
        # oracle on the same sample
        W = {k: torch.as_tensor(np.asarray(v), dtype=torch.float32) for k, v in Wn.items()}
        keys = [k for k in T.trainable_keys(W) if k.startswith("backbone.") or k.startswith("proposal_generator.")]
        for k in keys:
            W[k].requires_grad_(True)
        x = torch.from_numpy(tr.tensor("net_input", engine=True)[..., :3].astype(np.float32)).permute(0, 3, 1, 2)
        feats = O.resnet_forward(spec, W, x)
        feats.update(O.fpn_forward(spec, W, feats))
        logits, deltas = O.rpn_head(W, [feats[n] for n in spec.rpn_in_features])
        lg = torch.cat([t.permute(0, 2, 3, 1).reshape(2, -1) for t in logits], 1)
        dl = torch.cat([t.view(2, -1, 4, t.shape[2], t.shape[3]).permute(0, 3, 4, 1, 2).reshape(2, -1, 4) for t in deltas], 1)
        mgt = [torch.from_numpy(gt_boxes[i])[matched[i]] for i in range(2)]
        ref = T.rpn_losses(anchors, lg, dl, [labels[0].to(torch.int8), labels[1].to(torch.int8)], mgt, T.TrainSpec())
        (ref["loss_rpn_cls"] + ref["loss_rpn_loc"]).backward()
        assert abs(float(losses[0]) - float(ref["loss_rpn_cls"])) <= 1e-2 * float(ref["loss_rpn_cls"])
        assert abs(float(losses[1]) - float(ref["loss_rpn_loc"])) <= 1e-2 * float(ref["loss_rpn_loc"]) + 1e-6
        p = "proposal_generator.rpn_head."
        # fused head: rows [0,A) objectness, [A,5A) deltas
        got = tr.tensor("g:" + p + "heads.w") / scale
        want = np.concatenate([_ohwi32(W[p + "objectness_logits.weight"].grad), _ohwi32(W[p + "anchor_deltas.weight"].grad)], 0)
        assert np.linalg.norm(got[:15] - want) / np.linalg.norm(want) <= 2e-2 and float(np.abs(got[15:]).max()) == 0.0
        gb = tr.tensor("g:" + p + "heads.b") / scale
        wb = np.concatenate([W[p + "objectness_logits.bias"].grad.numpy(), W[p + "anchor_deltas.bias"].grad.numpy()])
        assert np.linalg.norm(gb[:15] - wb) / np.linalg.norm(wb) <= 2e-2
        got = tr.tensor("g:" + p + "conv.w") / scale
        want = _ohwi32(W[p + "conv.weight"].grad)
        assert np.linalg.norm(got - want) / np.linalg.norm(want) <= 3e-2
        got = tr.tensor("g:" + p + "conv.b") / scale
        assert np.linalg.norm(got - W[p + "conv.bias"].grad.numpy()) / np.linalg.norm(W[p + "conv.bias"].grad.numpy()) <= 3e-2
        # ... and through the FPN / ResNet
        for n, tol in [("backbone.fpn_output2", 3e-2), ("backbone.fpn_lateral4", 3e-2), ("backbone.bottom_up.res5.2.conv3", 6e-2),
                       ("backbone.bottom_up.res4.0.shortcut", 8e-2), ("backbone.bottom_up.res3.0.conv1", 1e-1)]:
            got = tr.tensor(f"g:{n}.w") / scale
            want = _ohwi32(W[n + ".weight"].grad)
            rel = float(np.linalg.norm(got - want) / np.linalg.norm(want))
            assert rel <= tol, (n, rel)
    finally:
        tr.close()


def test_box_head_training_step_matches_autograd(gpu_required):
    """RPN + RoI box head training step: proposals + gt -> Matcher(0.5) -> sampler on the engine, then loss_cls /
    loss_box_reg and all four losses' gradients (box head, RPN, FPN/ResNet incl. the RoIAlign-backward path) against
    autograd of the oracle evaluated on the engine's own sampled anchors and RoIs."""
    from oracle import maskrcnn_oracle as O
    from oracle import train_oracle as T
    spec = EngineSpec(num_classes=2, min_size_test=320, max_size_test=533, rpn_pre_nms_topk_test=300, rpn_post_nms_topk_test=300)
    Wn = synthetic_weights(spec, seed=0)
    tiles = synthetic_tiles(2, 256, 256, 3, seed=654)
    scale = 128.0
    tr = Trainer(spec, Wn, (256, 256, 3), batch=2, loss_scale=scale)
    try:
        tr.set_sampling(256, 0.5, 96, 0.25)
        gt_boxes = [np.array([[20.0, 30.0, 120.0, 160.0], [150.0, 40.0, 300.0, 130.0], [60.0, 200.0, 110.0, 260.0]], np.float32),
                    np.array([[100.0, 100.0, 260.0, 280.0], [10.0, 10.0, 60.0, 50.0]], np.float32)]
        gt_classes = [np.array([0, 1, 0]), np.array([1, 1])]
        tr.set_targets(gt_boxes, gt_classes)
        tr.forward_trunk(tr.upload_tiles(tiles), 2)
        tr.rpn_forward(2)
        tr.roi_step(2, seed=11)
        tr.rpn_step(2, seed=11)
        tr.backward_trunk(2)
        tr.sync()
        cnt = tr.tensor("roi_sampled_count")
        boxes, cls, gtb = tr.tensor("roi_boxes"), tr.tensor("roi_classes"), tr.tensor("roi_gt_boxes")
        gti = tr.tensor("roi_gt_index")
        losses = tr.tensor("losses")
        K = 2
        sb, sc, sg, si = [], [], [], []
        for i in range(2):
            nf, nb = int(cnt[i, 0]), int(cnt[i, 1])
            k = nf + nb
            assert k == 96 and nf <= 24 and nf >= len(gt_boxes[i])          # the gt boxes themselves are foreground candidates
            c = cls[i, :k]
            assert (c[:nf] < K).all() and (c[:nf] >= 0).all() and (c[nf:] == K).all() and (cls[i, k:] == -1).all()
            # every sampled row carries the oracle Matcher's verdict for its box
            iou = T.pairwise_iou(torch.from_numpy(gt_boxes[i]), torch.from_numpy(boxes[i, :k]))
            m, lab = T.matcher(iou, [0.5], [0, 1], False)
            want = np.where(lab.numpy() == 1, gt_classes[i][m.numpy()], K)
            assert np.array_equal(c, want) and np.array_equal(gti[i, :nf], m.numpy()[:nf])
            assert np.allclose(gtb[i, :nf], gt_boxes[i][m.numpy()[:nf]])
            sb.append(torch.from_numpy(boxes[i, :k])); sc.append(torch.from_numpy(c.astype(np.int64)))
            sg.append(torch.from_numpy(np.where((c < K)[:, None], gtb[i, :k], boxes[i, :k])))
            si.append(torch.full((k,), i, dtype=torch.int64))
        # oracle on the same samples
        W = {k2: torch.as_tensor(np.asarray(v), dtype=torch.float32) for k2, v in Wn.items()}
        keys = [k2 for k2 in T.trainable_keys(W) if not k2.startswith("roi_heads.mask_head")]
        for k2 in keys:
            W[k2].requires_grad_(True)
        x = torch.from_numpy(tr.tensor("net_input", engine=True)[..., :3].astype(np.float32)).permute(0, 3, 1, 2)
        feats = O.resnet_forward(spec, W, x)
        feats.update(O.fpn_forward(spec, W, feats))
        logits, deltas = O.rpn_head(W, [feats[n] for n in spec.rpn_in_features])
        lg = torch.cat([t.permute(0, 2, 3, 1).reshape(2, -1) for t in logits], 1)
        dl = torch.cat([t.view(2, -1, 4, t.shape[2], t.shape[3]).permute(0, 3, 4, 1, 2).reshape(2, -1, 4) for t in deltas], 1)
        anchors = torch.cat([O.grid_anchors(spec, l, hw, hw) for l, hw in enumerate((80, 40, 20, 10, 5))])
        labels = torch.from_numpy(tr.tensor("rpn_labels").astype(np.int64))
        matched = torch.from_numpy(tr.tensor("rpn_matched").astype(np.int64))
        mgt = [torch.from_numpy(gt_boxes[i])[matched[i]] for i in range(2)]
        ts = T.TrainSpec()
        ref = T.rpn_losses(anchors, lg, dl, [labels[0].to(torch.int8), labels[1].to(torch.int8)], mgt, ts)
        rb, rc, rg, ri = torch.cat(sb), torch.cat(sc), torch.cat(sg), torch.cat(si)
        roi_feats = [feats[n] for n in spec.roi_in_features]
        pooled = T.roi_pooler_diff(roi_feats, [1 / 4, 1 / 8, 1 / 16, 1 / 32], rb, ri, 7)
        _, scores, reg = O.box_head(W, pooled)
        ref.update(T.fast_rcnn_losses(scores, reg, rb, rc, rg, K, spec.box_reg_weights, ts))
        sum(ref[n] for n in ("loss_rpn_cls", "loss_rpn_loc", "loss_cls", "loss_box_reg")).backward()
        for i, n in enumerate(("loss_rpn_cls", "loss_rpn_loc", "loss_cls", "loss_box_reg")):
            assert abs(float(losses[i]) - float(ref[n].detach())) <= 1.5e-2 * abs(float(ref[n].detach())) + 1e-6, (n, float(losses[i]), float(ref[n].detach()))

        def rel(name, want):
            got = tr.tensor(name) / scale
            assert got.shape == want.shape, (name, got.shape, want.shape)
            return float(np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30))
        p = "roi_heads."
        w_pred = np.concatenate([W[p + "box_predictor.cls_score.weight"].grad.numpy(), W[p + "box_predictor.bbox_pred.weight"].grad.numpy()], 0)
        got = tr.tensor("g:" + p + "box_predictor.w") / scale
        assert np.linalg.norm(got[:11] - w_pred) / np.linalg.norm(w_pred) <= 3e-2 and float(np.abs(got[11:]).max()) == 0.0
        assert rel("g:" + p + "box_head.fc2.w", W[p + "box_head.fc2.weight"].grad.numpy()) <= 3e-2
        g1 = W[p + "box_head.fc1.weight"].grad.reshape(1024, 256, 7, 7).permute(0, 2, 3, 1).reshape(1024, -1).numpy()
        assert rel("g:" + p + "box_head.fc1.w", g1) <= 3e-2
        assert rel("g:" + p + "box_head.fc1.b", W[p + "box_head.fc1.bias"].grad.numpy()) <= 3e-2
        for n, tol in [("backbone.fpn_output2", 3e-2), ("backbone.fpn_output4", 3e-2), ("backbone.fpn_lateral3", 3e-2),
                       ("proposal_generator.rpn_head.conv", 3e-2), ("backbone.bottom_up.res5.2.conv3", 6e-2), ("backbone.bottom_up.res3.1.conv2", 1e-1)]:
            r = rel(f"g:{n}.w", _ohwi32(W[n + ".weight"].grad))
            assert r <= tol, (n, r)
    finally:
        tr.close()


def _oracle_losses_on_engine_samples(tr, spec, W, gt_boxes, polys, targets, where, n=2):
    """The five training losses of the oracle (autograd-ready: ``W`` holds torch tensors, the trainable ones with requires_grad) on
    the ENGINE's own network input, sampled anchors (rpn_labels / rpn_matched) and sampled RoIs (roi_* tensors) of the step it has
    just run -- sampling is random in detectron2 (torch.randperm), so parity of a step is defined given the samples."""
    from oracle import maskrcnn_oracle as O
    from oracle import train_oracle as T
    losses = tr.tensor("losses")
    cnt = tr.tensor("roi_sampled_count")
    assert int(tr.tensor("mask_total")[0]) == len(where) == int(cnt[:, 0].sum()) and len(where) >= 3
    boxes, cls, gtb, gti = tr.tensor("roi_boxes"), tr.tensor("roi_classes"), tr.tensor("roi_gt_boxes"), tr.tensor("roi_gt_index")
    K = spec.num_classes
    x = torch.from_numpy(tr.tensor("net_input", engine=True)[..., :3].astype(np.float32)).permute(0, 3, 1, 2)
    feats = O.resnet_forward(spec, W, x)
    feats.update(O.fpn_forward(spec, W, feats))
    logits, deltas = O.rpn_head(W, [feats[n] for n in spec.rpn_in_features])
    lg = torch.cat([t.permute(0, 2, 3, 1).reshape(2, -1) for t in logits], 1)
    dl = torch.cat([t.view(2, -1, 4, t.shape[2], t.shape[3]).permute(0, 3, 4, 1, 2).reshape(2, -1, 4) for t in deltas], 1)
    anchors = torch.cat([O.grid_anchors(spec, l, hw, hw) for l, hw in enumerate((80, 40, 20, 10, 5))])
    labels = torch.from_numpy(tr.tensor("rpn_labels").astype(np.int64))
    matched = torch.from_numpy(tr.tensor("rpn_matched").astype(np.int64))
    ts = T.TrainSpec()
    ref = T.rpn_losses(anchors, lg, dl, [labels[0].to(torch.int8), labels[1].to(torch.int8)],
                       [torch.from_numpy(gt_boxes[i])[matched[i]] for i in range(2)], ts)
    rb, rc, rg, ri = [], [], [], []
    for i in range(2):
        k = int(cnt[i].sum())
        c = cls[i, :k]
        rb.append(torch.from_numpy(boxes[i, :k])); rc.append(torch.from_numpy(c.astype(np.int64)))
        rg.append(torch.from_numpy(np.where((c < K)[:, None], gtb[i, :k], boxes[i, :k]))); ri.append(torch.full((k,), i, dtype=torch.int64))
    rb, rc, rg, ri = torch.cat(rb), torch.cat(rc), torch.cat(rg), torch.cat(ri)
    roi_feats = [feats[n] for n in spec.roi_in_features]
    scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32]
    _, scores, reg = O.box_head(W, T.roi_pooler_diff(roi_feats, scales, rb, ri, 7))
    ref.update(T.fast_rcnn_losses(scores, reg, rb, rc, rg, K, spec.box_reg_weights, ts))
    mb = torch.from_numpy(np.stack([boxes[i, j] for i, j in where]))
    mi = torch.tensor([i for i, _ in where])
    mc = torch.tensor([int(cls[i, j]) for i, j in where])
    mlog, _ = O.mask_head(spec, W, T.roi_pooler_diff(roi_feats, scales, mb, mi, 14), mc)
    gm = torch.from_numpy(np.stack([T.rasterize_polygons_within_box(polys[i][int(gti[i, j])], boxes[i, j], 28) for i, j in where]))
    assert np.array_equal(gm.numpy(), targets) and 0.05 < float(gm.float().mean()) < 0.95
    ref["loss_mask"] = T.mask_rcnn_loss(mlog, mc, gm)
    return losses, ref


def test_full_training_step_five_losses_match_autograd(gpu_required):
    """The whole GeneralizedRCNN training forward + backward on the engine: trunk, RPN, proposals + gt -> sampled RoIs, box
    head, mask head on the sampled foreground with host-rasterised polygon targets -- all five losses and the gradients of
    every branch against autograd of the oracle on the engine's own samples, then one SGD step."""
    from oracle import maskrcnn_oracle as O
    from oracle import train_oracle as T
    spec = EngineSpec(num_classes=2, min_size_test=320, max_size_test=533, rpn_pre_nms_topk_test=300, rpn_post_nms_topk_test=300)
    Wn = synthetic_weights(spec, seed=0)
    tiles = synthetic_tiles(2, 256, 256, 3, seed=777)
    scale = 128.0
    tr = Trainer(spec, Wn, (256, 256, 3), batch=2, loss_scale=scale)
    try:
        tr.set_sampling(256, 0.5, 64, 0.25)
        gt_boxes = [np.array([[20.0, 30.0, 120.0, 160.0], [150.0, 40.0, 300.0, 130.0]], np.float32), np.array([[100.0, 100.0, 260.0, 280.0]], np.float32)]
        gt_classes = [np.array([0, 1]), np.array([1])]

        def blob(b, k):            # a k-gon inscribed in the box (not the box itself: partial masks inside jittered proposals)
            cx, cy, rx, ry = (b[0] + b[2]) / 2, (b[1] + b[3]) / 2, (b[2] - b[0]) / 2, (b[3] - b[1]) / 2
            th = np.linspace(0, 2 * np.pi, k, endpoint=False)
            return [np.stack([cx + rx * np.cos(th), cy + ry * np.sin(th)], 1).reshape(-1)]
        polys = [[blob(b, 7 + i) for i, b in enumerate(bs)] for bs in gt_boxes]
        tr.set_targets(gt_boxes, gt_classes)
        tr.forward_trunk(tr.upload_tiles(tiles), 2)
        tr.rpn_forward(2)
        tr.roi_step(2, seed=5)
        tr.mask_forward(2)
        targets, where = tr.mask_entries(polys, 2)
        tr.mask_backward(2, targets)
        tr.rpn_step(2, seed=5)
        tr.backward_trunk(2)
        tr.sync()
        W = {k2: torch.as_tensor(np.asarray(v), dtype=torch.float32) for k2, v in Wn.items()}
        for k2 in T.trainable_keys(W):
            W[k2].requires_grad_(True)
        K = 2
        ts = T.TrainSpec()
        losses, ref = _oracle_losses_on_engine_samples(tr, spec, W, gt_boxes, polys, targets, where)
        names = ("loss_rpn_cls", "loss_rpn_loc", "loss_cls", "loss_box_reg", "loss_mask")
        sum(ref[n] for n in names).backward()
        for i, n in enumerate(names):
            r = float(ref[n].detach())
            assert abs(float(losses[i]) - r) <= 1.5e-2 * abs(r) + 1e-6, (n, float(losses[i]), r)

        def rel(name, want):
            got = tr.tensor(name) / scale
            assert got.shape == want.shape, (name, got.shape, want.shape)
            return float(np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30))
        p = "roi_heads.mask_head."
        worst = {}
        for i in range(1, 5):
            worst[f"fcn{i}.w"] = rel(f"g:{p}mask_fcn{i}.w", _ohwi32(W[f"{p}mask_fcn{i}.weight"].grad))
            worst[f"fcn{i}.b"] = rel(f"g:{p}mask_fcn{i}.b", W[f"{p}mask_fcn{i}.bias"].grad.numpy())
        worst["deconv.w"] = rel(f"g:{p}deconv.w", W[p + "deconv.weight"].grad.permute(0, 2, 3, 1).reshape(256, 1024).numpy())
        worst["deconv.b"] = rel(f"g:{p}deconv.b", W[p + "deconv.bias"].grad.numpy())
        gp = tr.tensor(f"g:{p}predictor16.w") / scale
        wp = W[p + "predictor.weight"].grad[:, :, 0, 0].numpy()
        worst["predictor.w"] = float(np.linalg.norm(gp[:K] - wp) / np.linalg.norm(wp))
        assert float(np.abs(gp[K:]).max()) == 0.0
        gb = tr.tensor(f"g:{p}predictor16.b") / scale
        worst["predictor.b"] = float(np.linalg.norm(gb[:K] - W[p + "predictor.bias"].grad.numpy()) / np.linalg.norm(W[p + "predictor.bias"].grad.numpy()))
        for n in ("roi_heads.box_head.fc2", "backbone.fpn_output2", "backbone.fpn_output3", "proposal_generator.rpn_head.conv"):
            worst[n] = rel(f"g:{n}.w", _ohwi32(W[n + ".weight"].grad) if W[n + ".weight"].grad.dim() == 4 else W[n + ".weight"].grad.numpy())
        worst["res4.2.conv2"] = rel("g:backbone.bottom_up.res4.2.conv2.w", _ohwi32(W["backbone.bottom_up.res4.2.conv2.weight"].grad))
        print({k2: round(v, 4) for k2, v in worst.items()})
        assert max(v for k2, v in worst.items() if not k2.startswith("res")) <= 4e-2, worst
        assert worst["res4.2.conv2"] <= 8e-2
        # one optimiser step with the YAML's hyper-parameters at iteration 0 (warm-up: lr = 0.01 * 0.001)
        m0 = tr.tensor(f"m:{p}deconv.w").copy()
        g0 = tr.tensor(f"g:{p}deconv.w") / scale
        tr.apply_sgd(T.lr_at(ts, 0), ts.momentum, ts.weight_decay)
        tr.sync()
        m1 = tr.tensor(f"m:{p}deconv.w")
        assert np.allclose(m1, m0 - np.float32(T.lr_at(ts, 0)) * (g0 + np.float32(ts.weight_decay) * m0), rtol=1e-5, atol=1e-9)
    finally:
        tr.close()


def _two_image_problem():
    gt_boxes = [np.array([[20.0, 30.0, 120.0, 160.0], [150.0, 40.0, 300.0, 130.0]], np.float32), np.array([[100.0, 100.0, 260.0, 280.0]], np.float32)]
    gt_classes = [np.array([0, 1]), np.array([1])]

    def blob(b, k):
        cx, cy, rx, ry = (b[0] + b[2]) / 2, (b[1] + b[3]) / 2, (b[2] - b[0]) / 2, (b[3] - b[1]) / 2
        th = np.linspace(0, 2 * np.pi, k, endpoint=False)
        return [np.stack([cx + rx * np.cos(th), cy + ry * np.sin(th)], 1).reshape(-1)]
    polys = [[blob(b, 7 + i) for i, b in enumerate(bs)] for bs in gt_boxes]
    return gt_boxes, gt_classes, polys


def _engine_step(tr, tiles, gt_boxes, gt_classes, polys, seed):
    """One training forward + backward through the C ABI, stage by stage (engine.Trainer.train_step without the loss read-back)."""
    tr.set_targets(gt_boxes, gt_classes)
    tr.forward_trunk(tr.upload_tiles(tiles), 2)
    tr.rpn_forward(2)
    tr.roi_step(2, seed=seed)
    tr.mask_forward(2)
    targets, where = tr.mask_entries(polys, 2)
    tr.mask_backward(2, targets)
    tr.rpn_step(2, seed=seed)
    tr.backward_trunk(2)
    tr.sync()
    return targets, where


def _d2_grad(W, layer):
    """autograd's gradient of engine layer ``layer`` in the engine's GEMM layout (the layout of the flat gradient buffer)."""
    A, K = 3, 2
    if layer == "proposal_generator.rpn_head.heads":
        p = "proposal_generator.rpn_head."
        g = torch.cat([W[p + "objectness_logits.weight"].grad, W[p + "anchor_deltas.weight"].grad], 0)[:, :, 0, 0].numpy()
        out = np.zeros((16, g.shape[1]), np.float32); out[:5 * A] = g
        return out
    if layer == "roi_heads.box_predictor":
        g = torch.cat([W[layer + ".cls_score.weight"].grad, W[layer + ".bbox_pred.weight"].grad], 0).numpy()
        out = np.zeros((16, g.shape[1]), np.float32); out[:5 * K + 1] = g
        return out
    if layer == "roi_heads.box_head.fc1":
        g = W[layer + ".weight"].grad
        return g.reshape(-1, 256, 7, 7).permute(0, 2, 3, 1).reshape(g.shape[0], -1).numpy()
    if layer == "roi_heads.mask_head.deconv":
        return W[layer + ".weight"].grad.permute(0, 2, 3, 1).reshape(256, 1024).numpy()
    if layer == "roi_heads.mask_head.predictor16":
        g = W["roi_heads.mask_head.predictor.weight"].grad[:, :, 0, 0].numpy()
        out = np.zeros((16, g.shape[1]), np.float32); out[:K] = g
        return out
    g = W[layer + ".weight"].grad
    return _ohwi32(g) if g.dim() == 4 else g.numpy()


def test_reference_precision_training_step_matches_autograd(gpu_required):
    """The reference trains in fp32 (R:config/detectron2_config_3bands.yaml:268-305: no AMP key).  ``Trainer(spec.replace(precision=
    "fp32"))`` runs the whole step -- forward, the five losses, every input / weight / bias gradient -- with fp32 activations, gradients
    and operands on the fp32 matrix cores (csrc/ref_f32.hip with the backward epilogue, conv_wgrad_f32_kernel).  Against fp32
    autograd of the oracle on the engine's own samples: the losses to 1e-4, the gradient of EVERY trainable weight tensor to 1e-3
    relative L2 (measured values printed; the fp16 trainer's bar is 4e-2 / 8e-2)."""
    from oracle import train_oracle as T
    spec = EngineSpec(num_classes=2, min_size_test=320, max_size_test=533, rpn_pre_nms_topk_test=300, rpn_post_nms_topk_test=300, precision="fp32")
    Wn = synthetic_weights(spec, seed=0)
    tiles = synthetic_tiles(2, 256, 256, 3, seed=777)
    gt_boxes, gt_classes, polys = _two_image_problem()
    tr = Trainer(spec, Wn, (256, 256, 3), batch=2, loss_scale=1.0)
    try:
        tr.set_sampling(256, 0.5, 64, 0.25)
        targets, where = _engine_step(tr, tiles, gt_boxes, gt_classes, polys, seed=5)
        W = {k2: torch.as_tensor(np.asarray(v), dtype=torch.float32) for k2, v in Wn.items()}
        for k2 in T.trainable_keys(W):
            W[k2].requires_grad_(True)
        losses, ref = _oracle_losses_on_engine_samples(tr, spec, W, gt_boxes, polys, targets, where)
        names = ("loss_rpn_cls", "loss_rpn_loc", "loss_cls", "loss_box_reg", "loss_mask")
        sum(ref[n] for n in names).backward()
        for i, n in enumerate(names):
            r = float(ref[n].detach())
            assert abs(float(losses[i]) - r) <= 1e-4 * abs(r) + 1e-7, (n, float(losses[i]), r)
        worst = {}
        for layer in trainable_layers(spec):
            want = _d2_grad(W, layer)
            got = tr.tensor(f"g:{layer}.w")
            assert got.shape == want.shape, (layer, got.shape, want.shape)
            worst[layer] = float(np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30))
        top = sorted(worst.items(), key=lambda kv: -kv[1])[:6]
        print("fp32 trainer, worst weight-gradient rel L2:", [(k2.split(".", 2)[-1], f"{v:.2e}") for k2, v in top])
        assert max(worst.values()) <= 1e-3, top          # measured: 4.6e-4 (res5.2.conv1); ReLU masks of near-zero activations flip
    finally:
        tr.close()


def test_twenty_sgd_steps_side_by_side_with_the_oracle(gpu_required):
    """BASELINE.md row 5 ("loss-curve match"): 20 SGD steps (YAML solver: momentum 0.9, weight decay 1e-4, WarmupMultiStepLR from
    0.001 x BASE_LR) on one fixed batch, in the reference-precision trainer and in the oracle (torch autograd + torch.optim.SGD on
    detectron2's parameter set), the oracle taking the engine's samples of every step.  After 20 steps every trainable tensor agrees
    to 1e-4 relative L2 (1e-3 for the two tensors initialised at std 0.001) and -- the stricter statement, since 20 warm-up steps move
    a weight by ~1e-4 of its norm -- the accumulated UPDATE W20 - W0 of every tensor agrees to 2e-2; the loss curves agree to 1e-3.  The fp16 production trainer, run on the same
    batch with the same seeds, stays within 12 % + 0.05 of the reference-precision total loss at every step and within 5 % of its mean."""
    from oracle import train_oracle as T
    from proj_roadsurf_amd.synthetic import detectron2_head_init
    spec32 = EngineSpec(num_classes=2, min_size_test=320, max_size_test=533, rpn_pre_nms_topk_test=300, rpn_post_nms_topk_test=300, precision="fp32")
    # detectron2's own initial scale of the prediction layers (std 0.01 / 0.001): from the spread-logit heads of synthetic_weights SGD
    # climbs instead of descending, and a run that is blowing up amplifies every rounding difference
    Wn = detectron2_head_init(spec32, synthetic_weights(spec32, seed=0), seed=0)
    tiles = synthetic_tiles(2, 256, 256, 3, seed=777)
    gt_boxes, gt_classes, polys = _two_image_problem()
    ts = T.TrainSpec()
    steps = 20
    names = ("loss_rpn_cls", "loss_rpn_loc", "loss_cls", "loss_box_reg", "loss_mask")
    W = {k2: torch.as_tensor(np.asarray(v), dtype=torch.float32).clone() for k2, v in Wn.items()}
    keys = T.trainable_keys(W)
    for k2 in keys:
        W[k2].requires_grad_(True)
    opt = torch.optim.SGD([W[k2] for k2 in keys], lr=ts.base_lr, momentum=ts.momentum, weight_decay=ts.weight_decay)
    curve32, curve_o = [], []
    tr = Trainer(spec32, Wn, (256, 256, 3), batch=2, loss_scale=1.0)
    try:
        tr.set_sampling(256, 0.5, 64, 0.25)
        for it in range(steps):
            targets, where = _engine_step(tr, tiles, gt_boxes, gt_classes, polys, seed=100 + it)
            losses, ref = _oracle_losses_on_engine_samples(tr, spec32, W, gt_boxes, polys, targets, where)
            opt.zero_grad()
            tot = sum(ref[n] for n in names)
            tot.backward()
            for g in opt.param_groups:
                g["lr"] = T.lr_at(ts, it)
            opt.step()
            tr.apply_sgd(T.lr_at(ts, it), ts.momentum, ts.weight_decay)
            curve32.append(float(np.sum(losses[:5])))
            curve_o.append(float(tot.detach()))
        tr.sync()
        got = tr.export_weights(Wn)
    finally:
        tr.close()
    print("loss curves (engine fp32 / oracle):", [round(v, 4) for v in curve32[::4]], [round(v, 4) for v in curve_o[::4]])
    assert np.allclose(curve32, curve_o, rtol=1e-3, atol=1e-5), (curve32, curve_o)
    worst_w, worst_d = {}, {}
    for k2 in keys:
        w0, wo, we = np.asarray(Wn[k2], np.float32), W[k2].detach().numpy(), np.asarray(got[k2], np.float32)
        worst_w[k2] = float(np.linalg.norm(we - wo) / max(np.linalg.norm(wo), 1e-30))
        d = np.linalg.norm(wo - w0)
        if d > 0:
            worst_d[k2] = float(np.linalg.norm((we - w0) - (wo - w0)) / d)
    tw = sorted(worst_w.items(), key=lambda kv: -kv[1])[:3]
    td = sorted(worst_d.items(), key=lambda kv: -kv[1])[:5]
    print("after 20 steps: worst weight rel L2", [(k2, f"{v:.1e}") for k2, v in tw], "worst update rel L2", [(k2, f"{v:.1e}") for k2, v in td])
    # measured (round 3): updates <= 8.2e-3 (fpn_output3), weights <= 2e-5 except the two tensors detectron2 initialises at std 0.001
    # (bbox_pred, mask predictor: 3.3e-4 -- their 20-step update is as large as the tensor itself, so the update's error shows)
    tiny = ("roi_heads.box_predictor.bbox_pred.weight", "roi_heads.mask_head.predictor.weight")
    assert max(v for k2, v in worst_w.items() if k2 not in tiny) <= 1e-4, tw
    assert max(worst_w[k2] for k2 in tiny) <= 1e-3, tw
    assert max(worst_d.values()) <= 2e-2, td
    # the fp16 production trainer on the same batch, same sampling seeds
    spec16 = spec32.replace(precision="fp16")
    tr16 = Trainer(spec16, Wn, (256, 256, 3), batch=2, loss_scale=128.0)
    curve16 = []
    try:
        tr16.set_sampling(256, 0.5, 64, 0.25)
        for it in range(steps):
            _engine_step(tr16, tiles, gt_boxes, gt_classes, polys, seed=100 + it)
            curve16.append(float(np.sum(tr16.tensor("losses")[:5])))
            tr16.apply_sgd(T.lr_at(ts, it), ts.momentum, ts.weight_decay)
        tr16.sync()
    finally:
        tr16.close()
    print("loss curve fp16 trainer:", [round(v, 4) for v in curve16[::4]])
    # The two trainers draw their own samples: proposals differ in the last bits, the sampled anchors / RoIs then differ as sets, and
    # a step's loss depends on its sample (two RUNS of the fp32 trainer differ by up to 2 % at a step for the same reason -- its RoIAlign
    # backward uses float atomics).  Per step 12 % + 0.05, over the 20 steps 5 % of the mean (measured: <= 6.2 % / 1.1 %).
    for a, b in zip(curve16, curve32):
        assert abs(a - b) <= 0.12 * abs(b) + 0.05, (curve16, curve32)
    assert abs(np.mean(curve16) - np.mean(curve32)) <= 0.05 * np.mean(curve32), (np.mean(curve16), np.mean(curve32))


def test_trainer_stage_profiling(gpu_required):
    """rs_trainer_set_profiling / rs_trainer_stage_info (what bench.py --train groups): every stage of a step is timed once per step,
    GEMM stages carry their algorithmic FLOP, weight gradients are flagged as side-stream work, and nothing accumulates when off."""
    spec = EngineSpec(num_classes=2, min_size_test=320, max_size_test=533, rpn_pre_nms_topk_test=300, rpn_post_nms_topk_test=300)
    Wn = synthetic_weights(spec, seed=0)
    tiles = synthetic_tiles(2, 256, 256, 3, seed=777)
    gt_boxes, gt_classes, polys = _two_image_problem()
    tr = Trainer(spec, Wn, (256, 256, 3), batch=2, loss_scale=128.0)
    try:
        tr.set_sampling(256, 0.5, 64, 0.25)
        _engine_step(tr, tiles, gt_boxes, gt_classes, polys, seed=1)          # not profiled
        tr.set_profiling(True)
        for it in range(3):
            _engine_step(tr, tiles, gt_boxes, gt_classes, polys, seed=2 + it)
        st = {x["name"]: x for x in tr.stage_times()}
        tr.set_profiling(False)
        assert st["preprocess"]["calls"] == 3 and st["bwd.res4.2.conv2.w"]["calls"] == 3 and st["box.loss"]["calls"] == 3
        assert all(x["ms_total"] > 0 for x in st.values() if x["calls"])
        assert st["bwd.res4.2.conv2.w"]["side"] and not st["bwd.res4.2.conv2.x"]["side"]
        want = 2.0 * 2 * 20 * 20 * 9 * 256 * 256                              # res4 conv2 at 320x320: 20x20 pixels, batch 2
        assert st["bwd.res4.2.conv2.w"]["flops"] == want and st["bwd.res4.2.conv2.x"]["flops"] == want and st["res4.2.conv2"]["flops"] == want
        assert st["rpn.loss2"]["flops"] == 0
        _engine_step(tr, tiles, gt_boxes, gt_classes, polys, seed=9)          # off again: nothing accumulates
        assert {x["name"]: x["calls"] for x in tr.stage_times()}["preprocess"] == 3
    finally:
        tr.close()


def _tiny_training_workdir(tmp_path):
    """Four 128x128 synthetic tiles with two boxes each, COCO JSON, a small detectron2 YAML and the reference's two YAML sections
    (R:config/config_obj_detec.yaml:62-90) -- train_model.py section UNCHANGED in its model_weights key (zoo name only)."""
    import json
    import yaml
    from PIL import Image

    wd = tmp_path / "outputs" / "obj_detector"
    (wd / "trn-images").mkdir(parents=True)
    tiles = synthetic_tiles(4, 128, 128, 3, seed=19)
    images, anns = [], []
    rng = np.random.default_rng(0)
    for i in range(4):
        fn = f"trn-images/18_{100 + i}_200.tif"
        Image.fromarray(tiles[i][:, :, ::-1]).save(str(wd / fn))
        images.append({"id": i, "file_name": fn, "width": 128, "height": 128})
        for j in range(2):
            x, y = rng.integers(5, 60, 2)
            w, h = rng.integers(20, 60, 2)
            anns.append({"id": len(anns), "image_id": i, "category_id": 1 + (j % 2), "bbox": [int(x), int(y), int(w), int(h)], "iscrowd": 0,
                         "segmentation": [[int(x), int(y), int(x + w), int(y), int(x + w), int(y + h), int(x), int(y + h)]], "area": int(w * h)})
    cats = [{"id": 1, "name": "artificial"}, {"id": 2, "name": "natural"}]
    json.dump({"images": images, "annotations": anns, "categories": cats}, open(wd / "COCO_trn.json", "w"))
    d2 = {"INPUT": {"FORMAT": "RGB", "MIN_SIZE_TEST": 192, "MAX_SIZE_TEST": 320, "RANDOM_FLIP": "horizontal",
                    "MIN_SIZE_TRAIN": [160, 192], "MIN_SIZE_TRAIN_SAMPLING": "choice"},          # multi-scale: two engine geometries
          "MODEL": {"RPN": {"PRE_NMS_TOPK_TEST": 200, "POST_NMS_TOPK_TEST": 200, "BATCH_SIZE_PER_IMAGE": 64}, "ROI_HEADS": {"NUM_CLASSES": 2, "BATCH_SIZE_PER_IMAGE": 64}},
          "SOLVER": {"BASE_LR": 0.002, "IMS_PER_BATCH": 2, "MAX_ITER": 6, "WARMUP_ITERS": 2, "STEPS": [4], "GAMMA": 0.5, "CHECKPOINT_PERIOD": 3},
          "TEST": {"DETECTIONS_PER_IMAGE": 20, "EVAL_PERIOD": 3}}
    yaml.safe_dump(d2, open(tmp_path / "d2.yaml", "w"))
    cfg = {"train_model.py": {"working_directory": str(wd), "log_subfolder": "logs", "sample_tagged_img_subfolder": "sample_training_images",
                              "COCO_files": {"trn": "COCO_trn.json", "val": "COCO_trn.json", "tst": "COCO_trn.json"},
                              "detectron2_config_file": str(tmp_path / "d2.yaml"),
                              "model_weights": {"model_zoo_checkpoint_url": "COCO-InstanceSegmentation/mask_rcnn_R_50_FPN_1x.yaml"}},
           "make_detections.py": {"working_directory": str(wd), "log_subfolder": "logs", "image_metadata_json": "img_metadata.json",
                                  "COCO_files": {"trn": "COCO_trn.json"}, "detectron2_config_file": str(tmp_path / "d2.yaml"),
                                  "model_weights": {"pth_file": "logs/model_final.pth"}, "rdp_simplification": {"enabled": True, "epsilon": 0.75},
                                  "score_lower_threshold": 0.05}}
    yaml.safe_dump(cfg, open(tmp_path / "config.yaml", "w"))
    return wd


def test_train_model_cli_then_make_detections(gpu_required, tmp_path):
    """train_model.py drop-in: same YAML section/keys as the reference (R:config/config_obj_detec.yaml:62-72); a few iterations on
    a tiny synthetic COCO set write metrics.json and model_final.pth in detectron2's checkpoint layout, which make_detections.py
    then consumes through model_weights.pth_file; the weights moved and the losses are finite."""
    import json
    import os
    import yaml
    from PIL import Image
    from proj_roadsurf_amd import make_detections, train_model
    from proj_roadsurf_amd.weights import load_checkpoint

    wd = _tiny_training_workdir(tmp_path)
    cwd = os.getcwd()
    try:
        assert train_model.main([str(tmp_path / "config.yaml"), "--synthetic-weights", "--log-period", "1", "--loss-scale", "256", "--precision", "fp16"]) == 0
        os.chdir(cwd)
        lines = [json.loads(l) for l in open(wd / "logs" / "metrics.json")]
        assert [l["iteration"] for l in lines] == list(range(6))
        assert all(np.isfinite(l["total_loss"]) and l["total_loss"] > 0 for l in lines)
        assert lines[0]["lr"] == pytest.approx(0.002 * 0.001) and lines[3]["lr"] == pytest.approx(0.002) and lines[5]["lr"] == pytest.approx(0.001)
        assert [("validation_loss" in l) for l in lines] == [False, False, True, False, False, True]       # TEST.EVAL_PERIOD 3
        assert all(np.isfinite(l["validation_loss"]) and l["validation_loss"] > 0 for l in lines if "validation_loss" in l)
        ev = lines[-1]
        assert "bbox/AP" in ev and "segm/AP50" in ev and all(ev[k] is None or 0.0 <= ev[k] <= 100.0 for k in ev if k.startswith(("bbox/", "segm/")))
        assert (wd / "logs" / "model_0000002.pth").exists() and open(wd / "logs" / "last_checkpoint").read() == "model_final.pth"
        tagged = sorted(os.listdir(wd / "sample_training_images"))          # ground-truth previews (R:config/config_obj_detec.yaml:65)
        assert tagged and all(t.endswith(".png") and t.split("_")[0] in ("trn", "val", "tst") for t in tagged)
        W1 = load_checkpoint(str(wd / "logs" / "model_final.pth"))
        spec = EngineSpec(num_classes=2)
        W0 = synthetic_weights(spec, seed=0)
        assert set(W1) == set(W0)
        moved = float(np.abs(W1["roi_heads.box_head.fc2.weight"] - W0["roi_heads.box_head.fc2.weight"]).max())
        assert 0 < moved < 0.1
        assert np.array_equal(W1["backbone.bottom_up.res2.0.conv1.weight"], W0["backbone.bottom_up.res2.0.conv1.weight"])     # FREEZE_AT 2
        json.dump({}, open(wd / "img_metadata.json", "w"))
        assert make_detections.main([str(tmp_path / "config.yaml"), "--batch", "2"]) == 0
        os.chdir(cwd)
        assert (wd / "trn_detections_at_0dot05_threshold.gpkg").exists()
    finally:
        os.chdir(cwd)


def test_train_model_starts_from_the_unchanged_yaml_with_a_cached_zoo_checkpoint(gpu_required, tmp_path, monkeypatch):
    """R:config/config_obj_detec.yaml:71-72 names only ``model_zoo_checkpoint_url``.  With the 80-class COCO checkpoint present in
    detectron2's cache layout ($FVCORE_CACHE/detectron2/<name>/<id>/model_final_<hash>.pkl) the CLI starts WITHOUT an edited YAML
    and without --synthetic-weights: backbone / FPN / RPN / heads loaded, class-shaped layers re-initialised for the two classes."""
    import json
    import os
    import pickle
    from proj_roadsurf_amd import train_model
    from proj_roadsurf_amd.weights import load_checkpoint

    wd = _tiny_training_workdir(tmp_path)
    zoo = synthetic_weights(EngineSpec(num_classes=80), seed=3)
    d = tmp_path / "cache" / "detectron2" / "COCO-InstanceSegmentation" / "mask_rcnn_R_50_FPN_1x" / "137260431"
    d.mkdir(parents=True)
    with open(d / "model_final_a54504.pkl", "wb") as f:
        pickle.dump({"model": zoo, "__author__": "Detectron2 Model Zoo"}, f, protocol=2)
    monkeypatch.setenv("FVCORE_CACHE", str(tmp_path / "cache"))
    cwd = os.getcwd()
    try:
        assert train_model.main([str(tmp_path / "config.yaml"), "--max-iter", "2", "--log-period", "1", "--loss-scale", "256", "--tagged-samples", "0"]) == 0
    finally:
        os.chdir(cwd)
    lines = [json.loads(l) for l in open(wd / "logs" / "metrics.json")]
    assert [l["iteration"] for l in lines] == [0, 1] and all(np.isfinite(l["total_loss"]) for l in lines)
    W1 = load_checkpoint(str(wd / "logs" / "model_final.pth"))
    assert W1["roi_heads.box_predictor.cls_score.weight"].shape == (3, 1024) and W1["roi_heads.mask_head.predictor.weight"].shape[0] == 2
    assert np.array_equal(W1["backbone.bottom_up.res2.0.conv1.weight"], zoo["backbone.bottom_up.res2.0.conv1.weight"])       # loaded, frozen
    monkeypatch.setenv("FVCORE_CACHE", str(tmp_path / "empty"))
    monkeypatch.setenv("HOME", str(tmp_path / "nohome"))
    try:
        with pytest.raises(SystemExit) as ex:
            train_model.main([str(tmp_path / "config.yaml"), "--max-iter", "1"])
        assert "no cached copy" in str(ex.value)
    finally:
        os.chdir(cwd)


def test_train_model_two_ranks_data_parallel(gpu_required, tmp_path):
    """Data-parallel path end to end with two ranks (both on this box's one GPU, gloo all-reduce through host copies; on a node
    the same code runs one rank per GPU over RCCL): TrainingSampler strides, flat-gradient all-reduce, averaged SGD step, rank-0
    checkpoint."""
    import json
    import os
    import subprocess
    import sys
    import yaml
    from PIL import Image

    wd = tmp_path / "w"
    (wd / "trn-images").mkdir(parents=True)
    tiles = synthetic_tiles(4, 128, 128, 3, seed=23)
    images, anns = [], []
    for i in range(4):
        fn = f"trn-images/18_{i}_0.tif"
        Image.fromarray(tiles[i][:, :, ::-1]).save(str(wd / fn))
        images.append({"id": i, "file_name": fn, "width": 128, "height": 128})
        anns.append({"id": i, "image_id": i, "category_id": 1 + i % 2, "bbox": [20 + i, 30, 60, 50], "iscrowd": 0,
                     "segmentation": [[20 + i, 30, 80 + i, 30, 80 + i, 80, 20 + i, 80]]})
    json.dump({"images": images, "annotations": anns, "categories": [{"id": 1, "name": "a"}, {"id": 2, "name": "b"}]}, open(wd / "COCO_trn.json", "w"))
    d2 = {"INPUT": {"FORMAT": "RGB", "MIN_SIZE_TEST": 192, "MAX_SIZE_TEST": 320},
          "MODEL": {"RPN": {"PRE_NMS_TOPK_TEST": 200, "POST_NMS_TOPK_TEST": 200, "BATCH_SIZE_PER_IMAGE": 64}, "ROI_HEADS": {"NUM_CLASSES": 2, "BATCH_SIZE_PER_IMAGE": 64}},
          "SOLVER": {"BASE_LR": 0.002, "IMS_PER_BATCH": 2, "MAX_ITER": 3, "WARMUP_ITERS": 1, "CHECKPOINT_PERIOD": 100}}
    yaml.safe_dump(d2, open(tmp_path / "d2.yaml", "w"))
    cfg = {"train_model.py": {"working_directory": str(wd), "log_subfolder": "logs", "COCO_files": {"trn": "COCO_trn.json"},
                              "detectron2_config_file": str(tmp_path / "d2.yaml"), "model_weights": {}}}
    yaml.safe_dump(cfg, open(tmp_path / "config.yaml", "w"))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RS_DIST_BACKEND="gloo", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", "-m", "proj_roadsurf_amd.train_model", str(tmp_path / "config.yaml"), "--synthetic-weights",
                        "--log-period", "1", "--loss-scale", "256"], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(l) for l in open(wd / "logs" / "metrics.json")]
    assert [l["iteration"] for l in lines] == [0, 1, 2] and all(np.isfinite(l["total_loss"]) for l in lines)
    assert (wd / "logs" / "model_final.pth").exists()
    assert "batch 1 x 2 ranks" in r.stderr


def test_data_parallel_allreduce_sums_and_averages(gpu_required):
    """The data-parallel gradient exchange (engine.Trainer.allreduce_gradients: bucketed, behind per-bucket completion events),
    two ranks on this box's GPU over gloo -- tests/dp_worker.py asserts: reduced buffer == g0 + g1 exactly, master weights
    bit-identical across ranks after the SGD step and equal to a single-process step on the summed gradient with divisor 2."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29537", os.path.join(root, "tests", "dp_worker.py")], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DP_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_rccl_allreduce_path_on_one_gpu(gpu_required):
    """The device-side form of the gradient exchange (backend "nccl" = RCCL: zero-copy bucket views, asynchronous all-reduces
    behind the bucket events, stream hand-back) in a one-rank process group on this box's GPU -- tests/nccl_worker.py."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""), MASTER_ADDR="127.0.0.1", MASTER_PORT="29541",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "nccl_worker.py")], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "NCCL_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_multi_scale_state_handover(gpu_required):
    """MIN_SIZE_TRAIN "choice": the optimiser state follows the batch from the trainer of one input size to the next -- after a
    step at 160 px, the 192 px trainer holds bit-identical master weights and momentum and a refolded forward."""
    from proj_roadsurf_amd.engine import MultiScaleTrainer
    spec = EngineSpec(num_classes=2, min_size_test=192, max_size_test=320, rpn_pre_nms_topk_test=200, rpn_post_nms_topk_test=200)
    Wn = synthetic_weights(spec, seed=0)
    tiles = synthetic_tiles(1, 128, 128, 3, seed=31)
    ms = MultiScaleTrainer(spec, Wn, (128, 128, 3), [160, 192], batch=1, loss_scale=64.0)
    try:
        ms.set_sampling(64, 0.5, 64, 0.25)
        a = ms.select(160)
        assert ms.net_shape(160) == (160, 160) and ms.net_shape(192) == (192, 192)
        s = 160 / 128
        gb = [np.array([[20.0, 30.0, 90.0, 100.0]], np.float32) * s]
        polys = [[[np.array([20.0, 30, 90, 30, 90, 100, 20, 100]) * s]]]
        l0 = a.train_step(tiles, gb, [np.array([1])], polys, seed=3)
        a.apply_sgd(1e-3, 0.9, 1e-4)
        n = "roi_heads.box_head.fc2"
        m_a = a.tensor(f"m:{n}.w").copy()
        assert float(np.abs(m_a - Wn[n + ".weight"]).max()) > 0 and all(np.isfinite(v) for v in l0.values())
        b = ms.select(192)
        assert b is not a and np.array_equal(b.tensor(f"m:{n}.w"), m_a)
        assert np.array_equal(b.tensor("m:backbone.bottom_up.res4.3.conv2.w"), a.tensor("m:backbone.bottom_up.res4.3.conv2.w"))
        s2 = 192 / 128
        l1 = b.train_step(tiles, [gb[0] / s * s2], [np.array([1])], [[[polys[0][0][0] / s * s2]]], seed=4)
        b.apply_sgd(1e-3, 0.9, 1e-4)                   # second step: momentum carried over, not re-initialised
        assert all(np.isfinite(v) for v in l1.values())
        assert float(np.abs(b.tensor(f"m:{n}.w") - m_a).max()) > 0
    finally:
        ms.close()


def test_per_image_sizes_in_one_batch(gpu_required):
    """MIN_SIZE_TRAIN drawn per image (R:31-38): a batch of two images at shortest edges 320 and 288 in one 320 canvas -- the
    network input of image 1 is Pillow's 256 -> 288 resize (the 288-px trainer's own input) with zeros beyond, its training
    proposals are clipped to 288 (oracle's find_top_rpn_proposals with image_sizes per image), a whole step runs, and the
    uniform batch comes back with ``set_image_sizes(None)``."""
    from oracle import maskrcnn_oracle as O
    from proj_roadsurf_amd.engine import MultiScaleTrainer
    spec = EngineSpec(num_classes=2, min_size_test=320, max_size_test=533, rpn_pre_nms_topk_test=300, rpn_post_nms_topk_test=300)
    Wn = synthetic_weights(spec, seed=0)
    tiles = synthetic_tiles(2, 256, 256, 3, seed=4242)
    ms = MultiScaleTrainer(spec, Wn, (256, 256, 3), [288, 320], batch=2, loss_scale=64.0)
    try:
        ms.set_sampling(256, 0.5, 128, 0.25)
        small = ms.select(288)
        small.forward_trunk(small.upload_tiles(tiles), 2)
        small.sync()
        want1 = small.tensor("net_input", engine=True)[1].copy()                       # (288, 288, 4)
        tr = ms.select_batch([320, 288])
        assert tr is ms.select(320) and tr is not small
        tr = ms.select_batch([320, 288])
        hw = tr.set_image_sizes([320, 288])
        assert hw == [(320, 320), (288, 288)]
        gt_boxes = [np.array([[20.0, 30.0, 120.0, 160.0]], np.float32), np.array([[100.0, 100.0, 260.0, 280.0], [10.0, 10.0, 60.0, 50.0]], np.float32)]
        polys = [[[np.array([b[0], b[1], b[2], b[1], b[2], b[3], b[0], b[3]], np.float64)] for b in bs] for bs in gt_boxes]
        tr.set_targets(gt_boxes, [np.array([0]), np.array([1, 1])])
        tr.forward_trunk(tr.upload_tiles(tiles), 2)
        tr.rpn_forward(2)
        tr.roi_step(2, seed=1)
        tr.sync()
        x = tr.tensor("net_input", engine=True)
        assert x.shape[1:3] == (320, 320)
        uniform = Trainer(spec, Wn, (256, 256, 3), batch=2, loss_scale=64.0)
        try:
            uniform.forward_trunk(uniform.upload_tiles(tiles), 2)
            uniform.sync()
            assert np.array_equal(x[0], uniform.tensor("net_input", engine=True)[0])
        finally:
            uniform.close()
        assert np.array_equal(x[1, :288, :288], want1) and float(np.abs(want1).max()) > 0
        assert float(np.abs(x[1, 288:]).max()) == 0.0 and float(np.abs(x[1, :, 288:]).max()) == 0.0
        cand, cc = tr.tensor("roi_candidates"), tr.tensor("roi_candidate_count")
        A = spec.num_anchors
        logits, deltas = [], []
        for l in range(2, 7):
            h = torch.from_numpy(tr.tensor(f"rpn_head{l}", engine=True))
            logits.append(h[..., :A].permute(0, 3, 1, 2).contiguous())
            deltas.append(h[..., A:5 * A].permute(0, 3, 1, 2).contiguous())
        train_spec = spec.replace(rpn_pre_nms_topk_test=2000, rpn_post_nms_topk_test=1000)
        props = O.rpn_proposals(train_spec, logits, deltas, [(320, 320), (288, 288)], nms_trick=False)
        for i in range(2):
            pb = props[i]["boxes"].numpy()
            k = pb.shape[0]
            assert int(cc[i]) == k + gt_boxes[i].shape[0]
            assert float(np.abs(cand[i, :k] - pb).max()) <= 1e-3
        assert float(cand[1, :int(cc[1]) - 2].max()) <= 288.0 and float(cand[0, :int(cc[0]) - 1].max()) > 300.0
        losses = tr.train_step(tiles, gt_boxes, [np.array([0]), np.array([1, 1])], polys, seed=9, sizes=[320, 288])
        assert all(np.isfinite(v) and v > 0 for v in losses.values()), losses
        tr.apply_sgd(1e-4, 0.9, 1e-4)
        assert not tr.overflowed()
        # back to one size for the batch: image 1 fills the canvas again
        tr.train_step(tiles, gt_boxes, [np.array([0]), np.array([1, 1])], polys, seed=10, sizes=[320, 320])
        x2 = tr.tensor("net_input", engine=True)
        assert float(np.abs(x2[1, 288:]).max()) > 0
        with pytest.raises(Exception):
            tr.set_image_sizes([352, 320])                                             # larger than the canvas
    finally:
        ms.close()


def test_training_mode_proposals_match_oracle(gpu_required):
    """RPN proposals of the TRAINING forward (PRE_NMS_TOPK_TRAIN 2000 per level, NMS 0.7, POST_NMS_TOPK_TRAIN 1000 per image, R:245-250:
    the 2048-capacity select / NMS / merge kernels) against the oracle's find_top_rpn_proposals on the engine's own head outputs,
    followed by the appended ground-truth boxes (PROPOSAL_APPEND_GT)."""
    from oracle import maskrcnn_oracle as O
    spec = EngineSpec(num_classes=2, min_size_test=320, max_size_test=533, rpn_pre_nms_topk_test=300, rpn_post_nms_topk_test=300)
    Wn = synthetic_weights(spec, seed=0)
    tiles = synthetic_tiles(2, 256, 256, 3, seed=888)
    tr = Trainer(spec, Wn, (256, 256, 3), batch=2, loss_scale=64.0)
    try:
        gt_boxes = [np.array([[20.0, 30.0, 120.0, 160.0]], np.float32), np.array([[100.0, 100.0, 260.0, 280.0], [10.0, 10.0, 60.0, 50.0]], np.float32)]
        tr.set_targets(gt_boxes, [np.array([0]), np.array([1, 1])])
        tr.set_sampling(256, 0.5, 128, 0.25)
        tr.forward_trunk(tr.upload_tiles(tiles), 2)
        tr.rpn_forward(2)
        tr.roi_step(2, seed=1)
        tr.sync()
        cand, cc = tr.tensor("roi_candidates"), tr.tensor("roi_candidate_count")
        A = spec.num_anchors
        logits, deltas = [], []
        for l in range(2, 7):
            h = torch.from_numpy(tr.tensor(f"rpn_head{l}", engine=True))            # (2, H, W, 16)
            logits.append(h[..., :A].permute(0, 3, 1, 2).contiguous())
            deltas.append(h[..., A:5 * A].permute(0, 3, 1, 2).contiguous())
        train_spec = spec.replace(rpn_pre_nms_topk_test=2000, rpn_post_nms_topk_test=1000)
        props = O.rpn_proposals(train_spec, logits, deltas, [(320, 320)] * 2, nms_trick=False)
        for i in range(2):
            pb = props[i]["boxes"].numpy()
            k = pb.shape[0]
            assert k > 300                                                             # more than the inference capacity of this config
            assert int(cc[i]) == k + gt_boxes[i].shape[0]
            assert float(np.abs(cand[i, :k] - pb).max()) <= 1e-3
            assert np.array_equal(cand[i, k:k + gt_boxes[i].shape[0]], gt_boxes[i])
    finally:
        tr.close()


def test_trainer_inference_engine_tracks_the_weights(gpu_required):
    """After SGD steps the trainer's forward engine, used for validation inference, carries the CURRENT weights in every inference
    operand (fp16 folds, fused heads, the fp32 mask predictor of the fused deconv kernel): its detections agree with a fresh engine
    built from the exported checkpoint (which fuses the projection shortcuts, so activations differ by fp16 rounding)."""
    from proj_roadsurf_amd.engine import Engine
    from tests.util import match_detections
    spec = EngineSpec(num_classes=2, min_size_test=320, max_size_test=533, rpn_pre_nms_topk_test=300, rpn_post_nms_topk_test=300)
    Wn = synthetic_weights(spec, seed=0)
    tiles = synthetic_tiles(2, 256, 256, 3, seed=55)
    tr = Trainer(spec, Wn, (256, 256, 3), batch=2, loss_scale=128.0)
    try:
        tr.set_sampling(256, 0.5, 64, 0.25)
        gb = [np.array([[20.0, 30.0, 120.0, 160.0]], np.float32), np.array([[100.0, 100.0, 260.0, 280.0]], np.float32)]
        polys = [[[np.array([b[0], b[1], b[2], b[1], b[2], b[3], b[0], b[3]], np.float64)] for b in bs] for bs in gb]
        for it in range(3):
            tr.train_step(tiles, gb, [np.array([0]), np.array([1])], polys, seed=it)
            tr.apply_sgd(2e-3, 0.9, 1e-4)
        W1 = tr.export_weights(Wn)
        assert float(np.abs(W1["roi_heads.mask_head.predictor.weight"] - Wn["roi_heads.mask_head.predictor.weight"]).max()) > 0
        got = tr.inference_engine().infer(tiles)
        ref_eng = Engine(spec, W1, (256, 256, 3), max_batch=2)
        try:
            want = ref_eng.infer(tiles)
        finally:
            ref_eng.close()
        old_eng = Engine(spec, Wn, (256, 256, 3), max_batch=2)
        try:
            old = old_eng.infer(tiles)
        finally:
            old_eng.close()
        for a, b, o in zip(want, got, old):
            assert len(a) > 0 and len(b) > 0
            m = match_detections({"boxes": a.pred_boxes, "scores": a.scores, "classes": a.pred_classes, "masks": a.pred_masks},
                                 {"boxes": b.pred_boxes, "scores": b.scores, "classes": b.pred_classes, "masks": b.pred_masks}, iou_thr=0.9)
            # the two engines differ by fp16 rounding (fused vs separate projection shortcut), which flips some NMS decisions among
            # the near-duplicate low-score boxes of a 3-step model; measured over 24 runs: frac_matched 0.83-1.0 (the training
            # itself varies in the last bits from run to run: float atomics in ROIAlign backward), score differences <= 2e-3
            assert m["frac_matched"] >= 0.7 and m["max_dscore"] <= 0.03 and m["agg_mask_iou"] >= 0.9, m
            # ... and they are NOT the detections of the initial weights
            m0 = match_detections({"boxes": o.pred_boxes, "scores": o.scores, "classes": o.pred_classes, "masks": o.pred_masks},
                                  {"boxes": b.pred_boxes, "scores": b.scores, "classes": b.pred_classes, "masks": b.pred_masks}, iou_thr=0.9)
            assert m0["frac_matched"] < m["frac_matched"] or m0["max_dscore"] > m["max_dscore"]
    finally:
        tr.close()


def test_side_stream_gradients_equal_single_stream(gpu_required, monkeypatch):
    """Weight / bias gradients run on the trainer's side stream next to the input-gradient chain (rs_trainer::run_list), and
    RoIAlign backward may run there too (RS_TRAIN_ROI_SIDE).  The gradients of the same step (same tiles, targets, sampling
    seed) are compared PER TENSOR with the single-stream run: the head layers (box / mask / RPN: nothing of theirs is downstream
    of RoIAlign-backward's float atomics) must be bit-identical; FPN and ResNet tensors see the atomics' last-bit noise through
    d:p2..p5 (measured: 1.5e-5 relative L2 on the whole buffer, up to 2.5e-4 on single deep tensors) and must agree to 1e-3 each -- a race that corrupts one
    small layer cannot hide in a global norm."""
    spec = EngineSpec(num_classes=2, min_size_test=320, max_size_test=533)
    Wn = synthetic_weights(spec, seed=0)
    tiles = synthetic_tiles(2, 256, 256, 3, seed=77)
    gb = [np.array([[20.0, 30.0, 120.0, 160.0], [150.0, 40.0, 300.0, 130.0]], np.float32), np.array([[100.0, 100.0, 260.0, 280.0]], np.float32)]
    gc = [np.array([0, 1]), np.array([1])]
    polys = [[[np.array([b[0], b[1], b[2], b[1], b[2], b[3], b[0], b[3]], np.float64)] for b in bs] for bs in gb]

    def grads(side, roi_side):
        monkeypatch.setenv("RS_TRAIN_SIDE", str(side))
        monkeypatch.setenv("RS_TRAIN_ROI_SIDE", str(roi_side))
        tr = Trainer(spec, Wn, (256, 256, 3), batch=2, loss_scale=256.0)
        out = []
        try:
            names = [n for n in tr.tensor_names() if n.startswith("g:")]
            for _ in range(2):
                tr.train_step(tiles, gb, gc, polys, seed=5)
                out.append({n: tr.tensor(n) for n in names})
        finally:
            tr.close()
        return out

    ref = grads(0, 0)[0]
    assert len(ref) >= 60 and all(float(np.abs(v).max()) > 0 for k, v in ref.items() if k.endswith(".w"))
    worst = 0.0
    for side, roi_side in ((1, 0), (1, 1)):
        for g in grads(side, roi_side):
            for name, want in ref.items():
                got = g[name]
                if "backbone." in name:
                    rel = float(np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30))
                    worst = max(worst, rel)
                    assert rel <= 1e-3, (side, roi_side, name, rel)      # measured up to 2.5e-4 (res3.0.conv1.w, the deepest tensor)
                else:
                    assert np.array_equal(got, want), (side, roi_side, name, float(np.abs(got - want).max()))
    print("side-stream per-tensor worst rel L2 (backbone tensors):", worst)


def test_overflowing_gradient_skips_the_step(gpu_required):
    """fp16 loss scaling: with an absurd scale the activation gradients overflow, the flat gradient holds inf / nan, and
    rs_trainer_apply_sgd must leave weights and momentum untouched and raise "grad_overflow" (GradScaler semantics); after
    rs_trainer_set_loss_scale the same batch steps normally."""
    spec = EngineSpec(num_classes=2, min_size_test=320, max_size_test=533)
    Wn = synthetic_weights(spec, seed=0)
    tiles = synthetic_tiles(2, 256, 256, 3, seed=91)
    gb = [np.array([[20.0, 30.0, 120.0, 160.0]], np.float32), np.array([[100.0, 100.0, 260.0, 280.0]], np.float32)]
    gc = [np.array([0]), np.array([1])]
    polys = [[[np.array([b[0], b[1], b[2], b[1], b[2], b[3], b[0], b[3]], np.float64)] for b in bs] for bs in gb]
    tr = Trainer(spec, Wn, (256, 256, 3), batch=2, loss_scale=1e9)
    try:
        name = "roi_heads.box_head.fc2.weight"
        losses = tr.train_step(tiles, gb, gc, polys, seed=1)
        assert all(np.isfinite(v) for v in losses.values())           # the loss VALUES are fp32 and unscaled
        tr.apply_sgd(1e-3, 0.9, 1e-4)
        assert tr.overflowed()
        W1 = tr.export_weights(Wn)
        assert np.array_equal(W1[name], Wn[name])                      # step skipped
        tr.set_loss_scale(256.0)
        tr.train_step(tiles, gb, gc, polys, seed=1)
        tr.apply_sgd(1e-3, 0.9, 1e-4)
        assert not tr.overflowed()
        W2 = tr.export_weights(Wn)
        d = float(np.abs(W2[name] - Wn[name]).max())
        assert 0 < d < 0.1
    finally:
        tr.close()


def test_batch_with_an_image_without_ground_truth(gpu_required):
    """An image whose annotations are all gone (detectron2 keeps such images when they survive FILTER_EMPTY_ANNOTATIONS, e.g.
    validation tiles): every anchor / proposal of it is background, no mask entries; the losses stay finite and the step runs."""
    spec = EngineSpec(num_classes=2, min_size_test=320, max_size_test=533, rpn_pre_nms_topk_test=300, rpn_post_nms_topk_test=300)
    Wn = synthetic_weights(spec, seed=0)
    tiles = synthetic_tiles(2, 256, 256, 3, seed=12)
    gb = [np.zeros((0, 4), np.float32), np.array([[100.0, 100.0, 260.0, 280.0]], np.float32)]
    gc = [np.zeros(0, np.int64), np.array([1])]
    polys = [[], [[np.array([100.0, 100.0, 260.0, 100.0, 260.0, 280.0, 100.0, 280.0])]]]
    tr = Trainer(spec, Wn, (256, 256, 3), batch=2, loss_scale=256.0)
    try:
        tr.set_sampling(256, 0.5, 64, 0.25)
        losses = tr.train_step(tiles, gb, gc, polys, seed=3)
        assert all(np.isfinite(v) and v >= 0 for v in losses.values()), losses
        cnt = tr.tensor("roi_sampled_count")
        assert int(cnt[0, 0]) == 0 and int(cnt[1, 0]) >= 1            # no foreground RoI on the empty image
        tr.apply_sgd(1e-3, 0.9, 1e-4)
        assert not tr.overflowed()
        # both images empty: only the background terms remain
        losses = tr.train_step(tiles, [gb[0], gb[0]], [gc[0], gc[0]], [[], []], seed=4)
        assert all(np.isfinite(v) for v in losses.values()), losses
        assert losses["loss_mask"] == 0.0 and losses["loss_box_reg"] == 0.0 and losses["loss_rpn_loc"] == 0.0
        tr.apply_sgd(1e-3, 0.9, 1e-4)
        assert not tr.overflowed()
    finally:
        tr.close()


def test_two_training_runs_give_the_same_bits(gpu_required):
    """Gradients never pass through float atomics any more (round 3: the RoIAlign backward is owner-computes, csrc/detect_kernels.hip
    roi_bwd_gather_kernel; weight gradients, bias gradients and the split-K reductions always summed in a fixed order): two runs of the
    same 12 SGD steps (batch 2, 256 x 256 scenes, box + mask heads, both trainer precisions) end in BIT-identical weights.  Only the
    logged loss values still go through a float atomic (one per wave), so the loss curves agree to rounding, not bit for bit."""
    from proj_roadsurf_amd.synthetic import train_trained_like
    for precision in ("fp16", "fp32"):
        spec = EngineSpec(num_classes=2, min_size_test=256, max_size_test=426, precision=precision)
        ls = 256.0 if precision == "fp16" else 1.0
        Wa, ca = train_trained_like(spec, 256, steps=12, batch=2, seed=3, warmup=6, pool=8, loss_scale=ls)
        Wb, cb = train_trained_like(spec, 256, steps=12, batch=2, seed=3, warmup=6, pool=8, loss_scale=ls)
        assert np.allclose(ca, cb, rtol=1e-5, atol=1e-6)
        for k in Wa:
            assert np.array_equal(np.asarray(Wa[k]), np.asarray(Wb[k])), f"{precision}: {k} differs between two identical runs"
        W0 = synthetic_weights(spec, seed=0)
        moved = [k for k in Wa if k in W0 and np.asarray(Wa[k]).shape == np.asarray(W0[k]).shape and not np.array_equal(np.asarray(Wa[k]), np.asarray(W0[k]))]
        assert len(moved) > 50, "the runs did not train"
