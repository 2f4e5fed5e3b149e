#!/usr/bin/env python
"""Generates the committed golden vectors by IMPORTING THE REFERENCE (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Needs /root/reference (read-only) — which does not exist on the GPU box, so nothing in tests/,
bench.py or smoke() ever runs this file; they only read the .npz files it wrote.  Inputs and weights
are NOT stored: both sides regenerate them from flowtrack.pytorch_amd.synth (seed + names); only the
reference's OUTPUTS are stored, plus the one real-data fixture (a 256x256 crop of the reference's
samples/img0.ppm, img1.ppm FlowNet demo pair, tools/flownet/demo.py:24-25).

While generating, the script also pins the oracle: oracle.pose_ref / oracle.flow_ref (FlowNet2S) /
oracle.keypoints_ref must reproduce the imported reference on the same inputs, else it aborts.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from flowtrack.pytorch_amd import synth  # noqa: E402
from oracle import flow_ref, keypoints_ref, pose_ref  # noqa: E402

SEED = 20260928


def import_reference_pose():
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))  # transforms.py imports cv2; only transform_image uses it
    sys.path.insert(0, os.path.join(REF, "lib", "pose"))
    import models as ref_pose_models  # noqa
    from utils import evaluation as ref_eval  # noqa
    sys.path.pop(0)
    for name in [m for m in sys.modules if m == "models" or m.startswith("models.") or m == "utils" or m.startswith("utils.")]:
        sys.modules["refpose_" + name] = sys.modules.pop(name)
    return ref_pose_models, ref_eval


def import_reference_flow():
    sys.path[:0] = [os.path.join(REF, "lib", "flownet"), os.path.join(REF, "lib", "flownet", "networks")]
    for name, attr in [("networks.resample2d_package._ext", "resample2d"),
                       ("networks.channelnorm_package._ext", "channelnorm"),
                       ("correlation_package._ext", "correlation")]:
        m = types.ModuleType(name)
        m.__path__ = []
        setattr(m, attr, types.SimpleNamespace())
        sys.modules[name] = m
    from model import models as ref_flow_models  # noqa
    return ref_flow_models


def inject_restated_cuda_ops():
    """Fills the three stubbed `_ext` FFI modules with the C restatement (oracle/flow_ops_ref.c via
    oracle/ops_ref.py), at exactly the boundary the reference binds (`Correlation_forward_cuda`
    correlation_package/functions/correlation.py:33, `Resample2d_cuda_forward` resample2d.py:21,
    `ChannelNorm_cuda_forward` channelnorm.py:14).  Everything above that line — the autograd Functions,
    the Modules, FlowNetC.forward (FlowNetC.py:71-128) and the stacked models (models.py:108-178,
    180-246, 346-498) — then runs as the REFERENCE's own Python, so the committed flows pin the graphs
    (concat orders, div_flow handling, which frame is warped, bilinear vs nearest x4) even though the
    three operators themselves stay 'reference CUDA-only, restated'."""
    from oracle import ops_ref

    def corr_fwd(in1, in2, rbot1, rbot2, output, pad, k, md, s1, s2, mult):
        assert mult == 1
        out = ops_ref.correlation_c(in1.detach().numpy(), in2.detach().numpy(), pad, k, md, s1, s2)
        output.resize_(out.shape).copy_(torch.from_numpy(out))
        return 1

    def resample_fwd(in1, flow, output, kernel_size):
        assert kernel_size == 1
        output.copy_(torch.from_numpy(ops_ref.resample2d_c(in1.detach().numpy(), flow.detach().numpy())))
        return 1

    def chnorm_fwd(in1, output, norm_deg):
        assert norm_deg == 2
        output.copy_(torch.from_numpy(ops_ref.channelnorm_c(in1.detach().numpy())))
        return 1

    sys.modules["correlation_package._ext"].correlation.Correlation_forward_cuda = corr_fwd
    sys.modules["networks.resample2d_package._ext"].resample2d.Resample2d_cuda_forward = resample_fwd
    sys.modules["networks.channelnorm_package._ext"].channelnorm.ChannelNorm_cuda_forward = chnorm_fwd


def flow_c_family(ref_flow, fout):
    """FlowNet2C / CS / CSS / FlowNet2 of the imported reference with the restated ops injected (F3, N4)."""
    inject_restated_cuda_ops()
    args = types.SimpleNamespace(rgb_max=255.0, fp16=False, grads={})
    pair = synth.frame_pairs(SEED, 1, 128, 192)
    for tag, cls, orc, off in (("c", "FlowNet2C", flow_ref.flownet2c_forward, 10),
                               ("cs", "FlowNet2CS", flow_ref.flownet2cs_forward, 11),
                               ("css", "FlowNet2CSS", flow_ref.flownet2css_forward, 12),
                               ("full", "FlowNet2", flow_ref.flownet2_forward, 13)):
        net = getattr(ref_flow, cls)(args).eval()
        sd = synth.fill_flow_state_dict(net.state_dict(), SEED + off)
        net.load_state_dict(sd)
        if cls == "FlowNet2":
            # models.py:146-176 registers gradient hooks on intermediates (`if not t.volatile`): they need
            # tensors that require grad, so this one runs without no_grad (values are the same)
            f_ref = net(pair).detach()
        else:
            with torch.no_grad():
                f_ref = net(pair)
        f_orc = orc(sd, pair)
        err = (f_ref - f_orc).abs().max().item()
        print(f"{cls} (ops injected): flow range [{f_ref.min():.3f}, {f_ref.max():.3f}]; oracle-vs-reference {err:.3e}")
        assert err <= 1e-4, f"oracle/flow_ref.py does not reproduce the imported {cls} graph"
        fout[f"synth_flow_{tag}"] = f_ref.numpy()
        fout[f"synth_flow_{tag}_seed"] = np.array(SEED + off)


def top2_margin(hm: np.ndarray) -> np.ndarray:
    flat = np.sort(hm.reshape(hm.shape[0], hm.shape[1], -1), axis=-1)
    return (flat[..., -1] - flat[..., -2]).astype(np.float32)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    if "--c-family-only" in sys.argv:           # adds the F3 / N4 vectors to the committed file, nothing else re-made
        path = os.path.join(HERE, "flow_golden.npz")
        fout = dict(np.load(path, allow_pickle=False))
        flow_c_family(import_reference_flow(), fout)
        np.savez_compressed(path, **fout)
        return
    ref_pose, ref_eval = import_reference_pose()

    # ---------------- pose: R50 (config C1: batch 4 of 256x192) --------------------------------
    out = {}
    for depth, B, H, W in ((50, 4, 256, 192), (101, 1, 256, 192)):
        net = ref_pose.deconv("resnet%d" % depth, num_classes=17, pretrained=False).eval()
        sd = synth.fill_pose_state_dict(net.state_dict(), SEED)
        net.load_state_dict(sd)
        x = synth.pose_crops(SEED, B, H, W)
        with torch.no_grad():
            hm_ref = net(x)
        hm_orc = pose_ref.pose_forward(sd, x, depth=depth)
        err = (hm_ref - hm_orc).abs().max().item()
        print(f"pose r{depth}: heatmap range [{hm_ref.min():.3f}, {hm_ref.max():.3f}] std {hm_ref.std():.3f}; "
              f"oracle-vs-reference max abs {err:.3e}")
        assert err <= 1e-5, "oracle/pose_ref.py does not reproduce the imported reference"
        hm = hm_ref.numpy()
        # reference max_preds / final_preds (torch 2.x drift: y = idx / w true-divides)
        center = np.stack([np.array([96.0 + 7 * i, 128.0 + 5 * i]) for i in range(B)])
        scale = np.array([256.0 * (1.0 + 0.1 * i) for i in range(B)])
        r_coords, r_scores = ref_eval.max_preds(hm_ref)
        o_coords, o_scores, o_idx = keypoints_ref.max_preds_ref(hm)
        assert np.array_equal(r_scores, o_scores)
        assert np.array_equal(r_coords[..., 0], o_coords[..., 0])
        # drift relation: ref_y = idx / w = y + x / w  (both zeroed where score <= 0)
        xs = o_idx % hm.shape[3]
        frac = (xs / hm.shape[3]).astype(np.float32) * (o_scores[..., 0] > 0)
        assert np.allclose(r_coords[..., 1] - frac, o_coords[..., 1], atol=1e-4), "max_preds drift relation broken"
        for adjust in (False, True):
            rf_coords, rf_scores = ref_eval.final_preds(hm_ref, center, scale, adjust_coords=adjust)
            of_coords, of_scores, _, of_pre = keypoints_ref.final_preds_ref(hm, center, scale, adjust_coords=adjust)
            # x is drift-free; y differs by exactly the (x/w) * inverse-affine gain
            assert np.allclose(rf_coords[..., 0], of_coords[..., 0], atol=1e-3)
            gain = scale / hm.shape[2]
            assert np.allclose(rf_coords[..., 1] - frac * gain[:, None], of_coords[..., 1], atol=1e-3)
            out[f"r{depth}_final_coords_adjust{int(adjust)}"] = of_coords.astype(np.float32)
            out[f"r{depth}_pre_coords_adjust{int(adjust)}"] = of_pre.astype(np.float32)
        out[f"r{depth}_idx"] = o_idx.astype(np.int32)
        out[f"r{depth}_scores"] = o_scores
        out[f"r{depth}_margin"] = top2_margin(hm)
        out[f"r{depth}_center"] = center
        out[f"r{depth}_scale"] = scale
        if depth == 50:
            out["r50_heatmaps_b2"] = hm[:2].copy()  # full fp32 maps of the first two crops
        out[f"r{depth}_heatmap_sum"] = hm.astype(np.float64).sum(axis=(2, 3)).astype(np.float32)
        out[f"r{depth}_shape"] = np.array([B, H, W])
        # OKS of the reference against itself is 1: metric sanity vs ref compute_oks
        pred = np.concatenate((of_coords, of_scores), axis=2)
        oks = ref_eval.compute_oks(pred, pred.copy(), scale * scale, _coco_delta())
        assert np.allclose(oks, 1.0)
    out["seed"] = np.array(SEED)
    np.savez_compressed(os.path.join(HERE, "pose_golden.npz"), **out)

    # a perturbed-prediction OKS/mAP vector from the reference metric code (pins the host metric port)
    rng = np.random.RandomState(7)
    anno = np.concatenate((rng.uniform(0, 200, (6, 17, 2)), np.ones((6, 17, 1))), axis=2)
    anno[1, 3:6, 2] = 0
    pred = anno.copy()
    pred[..., :2] += rng.normal(0, 3.0, (6, 17, 2))
    ref_scale = rng.uniform(3000, 9000, 6)
    oks = ref_eval.compute_oks(pred, anno, ref_scale, _coco_delta())
    ap = ref_eval.eval_mAP([pred[:3], pred[3:]], [anno[:3], anno[3:]], [ref_scale[:3], ref_scale[3:]], _coco_delta())
    preds_list = [{"score": float(s), "joints": pred[i], "area": float(ref_scale[i])} for i, s in
                  enumerate([0.9, 0.8, 0.95, 0.5, 0.7, 0.6])]
    preds_list[3]["joints"] = pred[0] + 0.5  # near-duplicate of #0 -> suppressed
    keep = ref_eval.nms_oks(preds_list, 0.9, _coco_delta())
    np.savez_compressed(os.path.join(HERE, "oks_golden.npz"), anno=anno, pred=pred, ref_scale=ref_scale, oks=oks,
                        ap=np.array(ap), nms_joints=np.array([p["joints"] for p in preds_list]),
                        nms_scores=np.array([p["score"] for p in preds_list]),
                        nms_areas=np.array([p["area"] for p in preds_list]), nms_keep=np.array(keep))
    print("oks golden:", oks.round(4), "AP", np.round(ap, 3), "keep", keep)

    # ---------------- flow: FlowNet2S ---------------------------------------------------------------
    ref_flow = import_reference_flow()
    args = types.SimpleNamespace(rgb_max=255.0, fp16=False, grads={})
    net = ref_flow.FlowNet2S(args).eval()
    sd = synth.fill_flow_state_dict(net.state_dict(), SEED)
    net.load_state_dict(sd)
    fout = {}
    # (a) synthetic pair, 128x192
    pair = synth.frame_pairs(SEED, 1, 128, 192)
    with torch.no_grad():
        f_ref = net(pair)
    f_orc = flow_ref.flownet2s_forward(sd, pair)
    err = (f_ref - f_orc).abs().max().item()
    print(f"FlowNet2S synthetic: flow range [{f_ref.min():.3f}, {f_ref.max():.3f}]; oracle-vs-reference {err:.3e}")
    assert err <= 1e-4
    fout["synth_flow"] = f_ref.numpy()
    fout["synth_shape"] = np.array([1, 128, 192])
    # (b) the reference's real demo pair, centre 256x256 crop
    from PIL import Image
    im0 = np.asarray(Image.open(os.path.join(REF, "samples", "img0.ppm")).convert("RGB"))
    im1 = np.asarray(Image.open(os.path.join(REF, "samples", "img1.ppm")).convert("RGB"))
    y0, x0 = (im0.shape[0] - 256) // 2, (im0.shape[1] - 256) // 2
    c0, c1 = im0[y0:y0 + 256, x0:x0 + 256], im1[y0:y0 + 256, x0:x0 + 256]
    ims = np.array([[c0, c1]]).transpose((0, 4, 1, 2, 3)).astype(np.float32)  # demo.py:86
    with torch.no_grad():
        f_ref = net(torch.from_numpy(ims))
    f_orc = flow_ref.flownet2s_forward(sd, torch.from_numpy(ims))
    err = (f_ref - f_orc).abs().max().item()
    print(f"FlowNet2S sample pair: flow range [{f_ref.min():.3f}, {f_ref.max():.3f}]; oracle-vs-reference {err:.3e}")
    assert err <= 1e-4
    fout["sample_pair_u8"] = np.stack((c0, c1))
    fout["sample_flow"] = f_ref.numpy()
    # (c) batchNorm=True variant (submodules.py:8-13) on the synthetic pair
    netbn = ref_flow.FlowNet2S(args, batchNorm=True).eval()
    sdbn = synth.fill_flow_state_dict(netbn.state_dict(), SEED + 1)
    netbn.load_state_dict(sdbn)
    with torch.no_grad():
        f_ref = netbn(pair)
    f_orc = flow_ref.flownet2s_forward(sdbn, pair)
    err = (f_ref - f_orc).abs().max().item()
    print(f"FlowNet2S batchNorm: flow range [{f_ref.min():.3f}, {f_ref.max():.3f}]; oracle-vs-reference {err:.3e}")
    assert err <= 1e-4
    fout["synth_flow_bn"] = f_ref.numpy()
    # (d) FlowNet2SD (models.py:294-344; trunk FlowNetSD.py) — plain and batchNorm, synthetic pair
    for bn in (False, True):
        netsd = ref_flow.FlowNet2SD(args, batchNorm=bn).eval()
        sdsd = synth.fill_flow_state_dict(netsd.state_dict(), SEED + 2 + int(bn))
        netsd.load_state_dict(sdsd)
        with torch.no_grad():
            f_ref = netsd(pair)
        f_orc = flow_ref.flownet2sd_forward(sdsd, pair)
        err = (f_ref - f_orc).abs().max().item()
        print(f"FlowNet2SD bn={bn}: flow range [{f_ref.min():.3f}, {f_ref.max():.3f}]; oracle-vs-reference {err:.3e}")
        assert err <= 1e-4
        fout["synth_flow_sd_bn" if bn else "synth_flow_sd"] = f_ref.numpy()
    flow_c_family(ref_flow, fout)
    fout["seed"] = np.array(SEED)
    # state_dict contracts (names + shapes) of the models the reference can build here
    for cls in ("FlowNet2S", "FlowNet2C", "FlowNet2CS", "FlowNet2SD", "FlowNet2CSS", "FlowNet2"):
        try:
            m = getattr(ref_flow, cls)(args)
            fout[f"keys_{cls}"] = np.array([f"{k}:{tuple(v.shape)}" for k, v in m.state_dict().items()])
        except Exception as e:  # FlowNet2C/CS constructors touch the stubbed CUDA ops lazily only
            print("could not build", cls, e)
    np.savez_compressed(os.path.join(HERE, "flow_golden.npz"), **fout)

    pnet = ref_pose.deconv("resnet50", num_classes=17, pretrained=False)
    np.savez_compressed(os.path.join(HERE, "state_dict_keys.npz"),
                        pose_r50=np.array([f"{k}:{tuple(v.shape)}" for k, v in pnet.state_dict().items()]),
                        **{k: v for k, v in fout.items() if k.startswith("keys_")})
    print("golden vectors written to", HERE)


def _coco_delta():
    return 2 * np.array([.26, .25, .25, .35, .35, .79, .79, .72, .72, .62, .62, 1.07, 1.07, .87, .87, .89, .89]) / 10.0


if __name__ == "__main__":
    main()
