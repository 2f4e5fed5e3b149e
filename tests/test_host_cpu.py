"""CPU suite: host-side logic (metrics port, C-ABI surface, weight packing geometry, error behaviour)."""
import ctypes
import os
import re
import types

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT
from flowtrack.pytorch_amd import _lib
from flowtrack.pytorch_amd._lib import ConvDesc, ConvGeometry, FlowtrackHipError
from flowtrack.pytorch_amd.pose import evaluation


def test_library_exports_every_declared_symbol(hip_lib):
    header = open(os.path.join(ROOT, "include", "flowtrack_hip.h")).read()
    # the drop-in boundary = everything outside `#ifdef FT_EXPERIMENTAL`; the experimental block holds measured alternatives
    a, b = header.index("#ifdef FT_EXPERIMENTAL"), header.index("#endif /* FT_EXPERIMENTAL */")
    stable_text, exp_text = header[:a] + header[b:], header[a:b]
    pat = r"^(?:int|long long|double|size_t|const char\*)\s+(ft_[a-z0-9_]+)\s*\("
    declared = set(re.findall(pat, stable_text, flags=re.M))
    experimental = set(re.findall(pat, exp_text, flags=re.M))
    assert declared and experimental, "no declarations parsed"
    for name in declared | experimental:
        assert hasattr(hip_lib, name), f"{name} declared in flowtrack_hip.h but not exported"
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    assert experimental == set(_lib.EXPERIMENTAL_SYMBOLS), experimental ^ set(_lib.EXPERIMENTAL_SYMBOLS)
    # the library exports the C ABI and nothing else (csrc/exports.map): no mangled C++ helper, no kernel stub
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    assert exported == declared | experimental, sorted(exported ^ (declared | experimental))[:10]
    assert hip_lib.ft_version() >= 100
    assert hip_lib.ft_status_string(1) == b"invalid argument"


def test_conv_desc_struct_matches_header():
    header = open(os.path.join(ROOT, "include", "flowtrack_hip.h")).read()
    body = header[header.index("typedef struct ft_conv_desc {"):header.index("} ft_conv_desc;")]
    fields = []
    for line in body.splitlines()[1:]:
        line = line.split("/*")[0].strip()
        m = re.match(r"(int|float)\s+(.*);", line)
        if m:
            fields += [(n.strip(), m.group(1)) for n in m.group(2).split(",")]
    assert [f[0] for f in fields] == [f[0] for f in ConvDesc._fields_]
    assert all((c is ctypes.c_float) == (t == "float") for (_, t), (_, c) in zip(fields, ConvDesc._fields_))


def _desc(**kw):
    d = ConvDesc()
    base = dict(dtype=0, N=1, Hi=16, Wi=16, Cin=64, x_cstride=64, x_coff=0, Cout=64, kh=3, kw=3, stride=1, pad=1, transposed=0,
                Ho=16, Wo=16, y_cstride=64, y_coff=0, out_layout=0, act=0)
    base.update(kw)
    for k, v in base.items():
        setattr(d, k, v)
    return d


def _geom(hip_lib, d):
    g = ConvGeometry()
    return hip_lib.ft_conv_pack_geometry(ctypes.byref(d), ctypes.byref(g)), g


def test_pack_geometry_and_tap_table(hip_lib):
    st, g = _geom(hip_lib, _desc())                                  # channel-aligned 3x3: direct-to-LDS layout
    assert st == 0 and (g.nphases, g.ntaps, g.cin_pad, g.cout_pad, g.kpad, g.run_taps) == (1, 9, 64, 64, 9 * 64, 1)
    st, g = _geom(hip_lib, _desc(Cin=3, x_cstride=8, kh=7, kw=7, stride=2, pad=3, Ho=8, Wo=8))   # plain stem: generic layout
    assert st == 0 and (g.ntaps, g.cin_pad) == (49, 8) and g.kpad % 32 == 0 and g.kpad >= 49 * 8
    st, g = _geom(hip_lib, _desc(Cin=1026, x_cstride=1032, Cout=256, kh=4, kw=4, stride=2, pad=1, transposed=1, Ho=32, Wo=32, y_cstride=256))
    assert st == 0 and (g.nphases, g.ntaps, g.cin_pad, g.cout_pad) == (4, 4, 1032, 256)      # narrow stride: generic
    st, g = _geom(hip_lib, _desc(Cin=1026, x_cstride=1056, Cout=256, kh=4, kw=4, stride=2, pad=1, transposed=1, Ho=32, Wo=32, y_cstride=256))
    assert st == 0 and g.cin_pad == 1056 and g.kpad == 4 * 1056                               # act_stride: K-runs of 32
    # row-packed stem: one K-run per kernel ROW = 7 taps x 4 channels padded to 32 fp16 elements
    rp = _desc(Cin=3, x_cstride=4, kh=7, kw=7, stride=2, pad=3, Ho=8, Wo=8, x_lpad=3, x_wpitch=22)
    st, g = _geom(hip_lib, rp)
    assert st == 0 and (g.ntaps, g.cin_pad, g.kpad, g.run_taps, g.run_cpad) == (7, 32, 7 * 32, 7, 4)
    ky, kx = ctypes.c_int(), ctypes.c_int()
    assert hip_lib.ft_conv_tap_source(ctypes.byref(rp), 0, 5, 6, ctypes.byref(ky), ctypes.byref(kx)) == 0
    assert (ky.value, kx.value) == (5, 6)
    assert hip_lib.ft_conv_tap_source(ctypes.byref(rp), 0, 5, 7, ctypes.byref(ky), ctypes.byref(kx)) == 1
    assert _geom(hip_lib, _desc(Cin=3, x_cstride=4, kh=7, kw=7, stride=2, pad=3, Ho=8, Wo=8, x_lpad=2, x_wpitch=22))[0] == 1  # lpad < pad
    # transposed tap table: out[2q+p] <- in[q + p - t] * W[k], k = 1+2t (p=0) | 2t (p=1); each (ky,kx) exactly once
    d = _desc(kh=4, kw=4, stride=2, pad=1, transposed=1, Ho=32, Wo=32)
    seen = set()
    for ph in range(4):
        for t in range(4):
            assert hip_lib.ft_conv_tap_source(ctypes.byref(d), ph, t, 0, ctypes.byref(ky), ctypes.byref(kx)) == 0
            assert (ky.value + 1) % 2 == ph >> 1 and (kx.value + 1) % 2 == ph & 1
            seen.add((ky.value, kx.value))
    assert len(seen) == 16
    assert hip_lib.ft_conv_flops(ctypes.byref(_desc())) == 2.0 * 16 * 16 * 64 * 64 * 9


def test_bad_arguments_return_status_not_abort(hip_lib):
    assert _geom(hip_lib, _desc(Ho=15))[0] == 1          # inconsistent output size
    assert _geom(hip_lib, _desc(x_cstride=60))[0] == 1   # unaligned channel stride
    assert _geom(hip_lib, _desc(kh=3, kw=3, transposed=1, Ho=32, Wo=32))[0] == 2
    assert hip_lib.ft_conv2d_fwd(ctypes.byref(_desc()), None, None, None, None, None, None, None) == 1
    assert hip_lib.ft_correlation_fwd(None, None, None, 1, 1, 1, 1, 0, 1, 0, 1, 1, 1, None) == 1
    assert hip_lib.ft_resample2d_fwd(None, None, None, 1, 1, 1, 1, None) == 1
    oc = ctypes.c_int(); oh = ctypes.c_int(); ow = ctypes.c_int()
    assert hip_lib.ft_correlation_out_shape(256, 48, 64, 20, 1, 20, 1, 2, ctypes.byref(oc), ctypes.byref(oh), ctypes.byref(ow)) == 0
    assert (oc.value, oh.value, ow.value) == (441, 48, 64)


def test_product_path_fails_loudly_without_gpu(hip_lib):
    """no CPU fallback: a CPU model / tensor raises instead of silently computing somewhere else."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from flowtrack.pytorch_amd.pose import models
    m = models.deconv("resnet50", 17, False).eval()
    with pytest.raises(FlowtrackHipError):
        m(torch.zeros(1, 3, 256, 192))
    with pytest.raises(FlowtrackHipError):
        evaluation.max_preds(torch.zeros(1, 17, 64, 48))


def test_training_mode_is_rejected():
    from flowtrack.pytorch_amd.pose import models
    m = models.deconv("resnet50", 17, False)
    with pytest.raises(FlowtrackHipError):
        m(torch.zeros(1, 3, 256, 192))


def test_oks_metric_port_matches_reference_vectors():
    G = np.load(os.path.join(GOLDEN, "oks_golden.npz"))
    d = evaluation.COCO_DELTA
    assert np.allclose(evaluation.compute_oks(G["pred"], G["anno"], G["ref_scale"], d), G["oks"], atol=1e-12)
    ap = evaluation.eval_mAP([G["pred"][:3], G["pred"][3:]], [G["anno"][:3], G["anno"][3:]], [G["ref_scale"][:3], G["ref_scale"][3:]], d)
    assert np.allclose(ap, G["ap"])
    preds = [{"score": float(s), "joints": j, "area": float(a)} for s, j, a in zip(G["nms_scores"], G["nms_joints"], G["nms_areas"])]
    assert list(evaluation.nms_oks(preds, 0.9, d)) == list(G["nms_keep"])


def test_transform_preds_is_inverse_of_crop_affine():
    c, s, res = np.array([[100.0, 140.0]]), np.array([220.0]), (64, 48)
    t = evaluation.get_transform(c[0], s[0], res)
    pts = np.array([[[10.0, 20.0], [0.0, 0.0], [47.0, 63.0]]])
    img = evaluation.transform_preds(pts.copy(), c, s, res)
    back = (t @ np.concatenate((img[0], np.ones((3, 1))), 1).T)[:2].T
    assert np.allclose(back, pts[0])
    assert np.allclose(evaluation.transform_preds(np.array([[[24.0, 32.0]]]), c, s, res), c)   # heatmap centre -> box centre


def test_flow_io_and_arg_reflection(tmp_path):
    import argparse
    from flowtrack.pytorch_amd.flownet import models, tools
    f = np.arange(5 * 7 * 2, dtype=np.float32).reshape(5, 7, 2)
    path = str(tmp_path / "x.flo")
    tools.write_flow(f, path)
    raw = open(path, "rb").read()
    assert len(raw) == 12 + f.nbytes and np.frombuffer(raw[:4], np.float32)[0] == np.float32(202021.25)
    assert tuple(np.frombuffer(raw[4:12], np.int32)) == (7, 5)           # width, height (flowlib.py:139-148)
    assert np.array_equal(tools.read_flow(path), f)
    img = tools.flow_to_image(np.stack(np.meshgrid(np.linspace(-3, 3, 9), np.linspace(-2, 2, 6)), -1).astype(np.float32))
    assert img.shape == (6, 9, 3) and img.dtype == np.uint8
    parser = argparse.ArgumentParser()
    import sys
    old = sys.argv
    sys.argv = ["demo", "--model", "FlowNet2CS"]
    try:
        tools.add_arguments_for_module(parser, models, "model", default="FlowNet2S", choices=["FlowNet2S", "FlowNet2C", "FlowNet2CS"])
        args = parser.parse_args(["--model", "FlowNet2CS", "--model_div_flow", "10.0"])
    finally:
        sys.argv = old
    assert tools.kwargs_from_args(args, "model") == {"batchNorm": False, "div_flow": 10.0}
    assert tools.module_to_dict(models)["FlowNet2CS"] is models.FlowNet2CS


def test_pose_config_parse_warns_on_unknown():
    import warnings
    from tools.pose.config import DefaultConfig
    o = DefaultConfig()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        o.parse({"backbone": "resnet101", "not_an_option": 1})
    assert o.backbone == "resnet101" and any("not_an_option" in str(x.message) for x in w)


def _hints(hip_lib, d):
    buf = (ctypes.c_int * 64)()
    n = hip_lib.ft_conv_tile_candidates(ctypes.byref(d), buf, 64)
    assert n >= 0
    return [int(v) for v in buf[:n]]


def _decode(h):
    return dict(bp=h & 0xfff, bc=(h >> 12) & 0x1ff, sk=(1, 2, 4, 8, 3, 5, 6, 7)[(h >> 21) & 7], ks=(h >> 24) & 0xf, wide=(h >> 28) & 3, halo=(h >> 30) & 1)


def test_tile_candidates_and_fusion_descriptors(hip_lib):
    """Host-side contract of the tile benchmark (ft_conv_tile_candidates / tile_hint), the K-concat second input, the
    fused tail and the split-K workspace — pure host code, no launch."""
    # a big fp16 3x3 (layer1-like at batch 64): halo variants offered, no split-K (plenty of tiles), no workspace
    big = _desc(dtype=_lib.FT_F16, N=64, Hi=64, Wi=48, Ho=64, Wo=48)
    hs = [_decode(h) for h in _hints(hip_lib, big)]
    assert any(h["halo"] for h in hs) and all(h["sk"] == 1 for h in hs) and hip_lib.ft_conv_workspace_bytes(ctypes.byref(big)) == 0
    assert all(h["bc"] == 64 for h in hs), "Cout = 64 offers 64-wide channel tiles only"
    # few pixels + long K (FlowNet conv6_1-like): split-K variants + a workspace of max_sk x M x Cout_pad fp32
    deep = _desc(dtype=_lib.FT_F16, N=16, Hi=6, Wi=8, Ho=6, Wo=8, Cin=1024, x_cstride=1024, Cout=1024, y_cstride=1024)
    hs = [_decode(h) for h in _hints(hip_lib, deep)]
    sks = sorted({h["sk"] for h in hs})
    assert sks[-1] >= 4 and all(h["ks"] == 1 and not h["halo"] for h in hs if h["sk"] > 1)
    assert hip_lib.ft_conv_workspace_bytes(ctypes.byref(deep)) == sks[-1] * 16 * 6 * 8 * 1024 * 4
    assert any(h["bc"] == 256 for h in hs), "Cout % 256 == 0 offers the all-256-channel tiles"
    # fp32 (parity mode): no wide-K, no halo, no in-workgroup split-K
    f32 = _desc(dtype=_lib.FT_F32, N=8, Hi=32, Wi=24, Ho=32, Wo=24, Cin=128, x_cstride=128, Cout=128, y_cstride=128)
    assert all(h["wide"] == 0 and h["halo"] == 0 and h["ks"] == 1 for h in map(_decode, _hints(hip_lib, f32)))
    # K-concat second input: 1x1 only, geometry = [cin_pad | cin2_pad], FLOPs of both GEMMs, no split-K / halo variants
    kc = _desc(dtype=_lib.FT_F16, N=2, Hi=8, Wi=6, Ho=8, Wo=6, Cin=128, x_cstride=128, Cout=512, y_cstride=512, kh=1, kw=1, pad=0,
               x2_cin=256, x2_hi=16, x2_wi=12, x2_cstride=256, x2_coff=0, x2_stride=2)
    st, g = _geom(hip_lib, kc)
    assert st == 0 and (g.cin_pad, g.cin2_pad, g.kpad) == (128, 256, 384)
    assert hip_lib.ft_conv_flops(ctypes.byref(kc)) == 2.0 * 2 * 8 * 6 * 512 * (128 + 256)
    assert all(h["ks"] == 1 and h["sk"] == 1 and not h["halo"] for h in map(_decode, _hints(hip_lib, kc)))
    kc.kh = kc.kw = 3; kc.pad = 1
    assert _geom(hip_lib, kc)[0] == 2                    # a second input needs a 1x1 main conv: unsupported
    kc.kh = kc.kw = 1; kc.pad = 0; kc.x2_hi = 18
    assert _geom(hip_lib, kc)[0] == 1                    # second input does not line up with the output grid
    # fused tail: fp16, Cout in {64,128,256}, <= 32 tail outputs; FLOPs include the tail GEMM.  256 channels with Cin % 64 == 0:
    # two variants (the 128-pixel 8-wave tile and the 256 x 256 8-phase tile, wide level 3); otherwise one, nothing to pick
    tl = _desc(dtype=_lib.FT_F16, N=2, Hi=8, Wi=6, Cin=256, x_cstride=256, Cout=256, kh=4, kw=4, stride=2, pad=1, transposed=1, Ho=16, Wo=12,
               out_layout=1, y_cstride=0, tail_cout=17)
    assert _geom(hip_lib, tl)[0] == 0
    assert [(h["bp"], h["bc"], h["wide"], h["sk"]) for h in map(_decode, _hints(hip_lib, tl))] == [(128, 256, 0, 1), (256, 256, 3, 1)]
    tl128 = _desc(dtype=_lib.FT_F16, N=2, Hi=8, Wi=6, Cin=64, x_cstride=64, Cout=128, kh=4, kw=4, stride=2, pad=1, transposed=1, Ho=16, Wo=12,
                  out_layout=1, y_cstride=0, tail_cout=17)
    assert _geom(hip_lib, tl128)[0] == 0 and _hints(hip_lib, tl128) == []
    # the 8-phase tile is offered for fp16 layers with Cin % 64 == 0 and Cout % 256 == 0 only, with split-K forms when the
    # layer has fewer 256 x 256 tiles than CUs (deconv.0 of the pose head at batch 64: 48 tiles)
    d0 = _desc(dtype=_lib.FT_F16, N=64, Hi=8, Wi=6, Cin=2048, x_cstride=2048, Cout=256, y_cstride=256, kh=4, kw=4, stride=2, pad=1,
               transposed=1, Ho=16, Wo=12)
    h8 = [h for h in map(_decode, _hints(hip_lib, d0)) if h["wide"] == 3]
    assert sorted(h["sk"] for h in h8) == [1, 2, 4, 5, 8] and all((h["bp"], h["bc"], h["ks"], h["halo"]) == (256, 256, 1, 0) for h in h8)
    for bad in (dict(Cin=96, x_cstride=96), dict(Cout=128, y_cstride=128), dict(dtype=_lib.FT_F32)):
        kw = dict(dtype=_lib.FT_F16, N=8, Hi=32, Wi=24, Ho=32, Wo=24, Cin=128, x_cstride=128, Cout=256, y_cstride=256)
        kw.update(bad)
        assert not any(h["wide"] == 3 for h in map(_decode, _hints(hip_lib, _desc(**kw)))), bad
    assert hip_lib.ft_conv_flops(ctypes.byref(tl)) == 2.0 * 2 * 16 * 12 * 256 * (256 * 4 + 17)
    tl.tail_cout = 33
    assert _geom(hip_lib, tl)[0] == 1
    tl.tail_cout = 17; tl.dtype = _lib.FT_F32
    assert _geom(hip_lib, tl)[0] == 2                    # the fused tail is an fp16-mode feature


def test_bottleneck_desc_matches_header_and_argument_checks(hip_lib):
    """ft_bottleneck_desc mirror == header; supported / flops / argument errors need no GPU."""
    from flowtrack.pytorch_amd._lib import BottleneckDesc
    header = open(os.path.join(ROOT, "include", "flowtrack_hip.h")).read()
    body = header[header.index("typedef struct ft_bottleneck_desc {"):header.index("} ft_bottleneck_desc;")]
    fields = []
    for line in body.splitlines()[1:]:
        m = re.match(r"int\s+([^;]*);", line.split("/*")[0].strip())
        if m:
            fields += [n.strip() for n in m.group(1).split(",")]
    assert fields == [f[0] for f in BottleneckDesc._fields_]

    def desc(**kw):
        d = BottleneckDesc()
        base = dict(dtype=0, N=2, H=64, W=48, C=256, P=64, x_cstride=256, x_coff=0, y_cstride=256, y_coff=0, head_only=0)
        base.update(kw)
        for k, v in base.items():
            setattr(d, k, v)
        return d
    ok, unsupported, invalid = 0, 2, 1
    assert hip_lib.ft_bottleneck_supported(ctypes.byref(desc())) == ok
    assert hip_lib.ft_bottleneck_supported(ctypes.byref(desc(C=64, x_cstride=64, y_cstride=64, head_only=1))) == ok
    assert hip_lib.ft_bottleneck_supported(ctypes.byref(desc(C=64, x_cstride=64))) == unsupported          # full block needs C = 256
    assert hip_lib.ft_bottleneck_supported(ctypes.byref(desc(P=128))) == unsupported
    assert hip_lib.ft_bottleneck_supported(ctypes.byref(desc(dtype=1))) == unsupported                     # fp32: separate launches
    assert hip_lib.ft_bottleneck_supported(ctypes.byref(desc(x_coff=4))) == unsupported                    # 16-byte alignment
    assert hip_lib.ft_bottleneck_supported(ctypes.byref(desc(y_cstride=128))) == invalid
    assert hip_lib.ft_bottleneck_supported(ctypes.byref(desc(N=0))) == invalid
    macs = 2 * 64 * 48 * (256 * 64 + 9 * 64 * 64 + 64 * 256)
    assert hip_lib.ft_bottleneck_flops(ctypes.byref(desc())) == 2.0 * macs
    # NULL operands are an argument error, not a crash (no launch happens)
    assert hip_lib.ft_bottleneck_fwd(ctypes.byref(desc()), None, None, None, None, None, None, None) == invalid


def test_bottleneck_rstat_plan_and_weight_buffer(hip_lib, monkeypatch):
    """ft_bottleneck_rstat_* without a GPU: which shapes the strip form takes (cost rule, FT_BNK_RSTAT switch, map width), and the
    weight buffer hip_ops builds for it: BatchNorm scale folded into the fp16 weights, the shift as a (hi, lo) fp16 pair in 16
    extra K columns, in the layout include/flowtrack_hip.h states."""
    import torch
    from flowtrack.pytorch_amd import hip_ops, synth
    from flowtrack.pytorch_amd._lib import BottleneckDesc

    def desc(**kw):
        d = BottleneckDesc()
        base = dict(dtype=0, N=64, H=64, W=48, C=256, P=64, x_cstride=256, x_coff=0, y_cstride=256, y_coff=0, head_only=0)
        base.update(kw)
        for k, v in base.items():
            setattr(d, k, v)
        return d
    ok, unsupported, invalid = 0, 2, 1
    monkeypatch.delenv("FT_BNK_RSTAT", raising=False)
    monkeypatch.delenv("FT_BNR_SR", raising=False)
    sup = lambda **kw: hip_lib.ft_bottleneck_rstat_supported(ctypes.byref(desc(**kw)))
    assert sup() == ok                                  # 64 images x 4 strips of 16 rows = one strip per CU
    assert sup(N=2) == unsupported                      # one-row strips: the patch kernel's case
    assert sup(W=72, H=96) == unsupported               # T1 ring: W <= 62
    assert sup(P=128, C=512, x_cstride=512, y_cstride=512) == unsupported
    assert sup(head_only=1) == unsupported and sup(projection=1) == unsupported and sup(dtype=1) == unsupported
    assert sup(x_coff=4) == unsupported and sup(y_cstride=128) == invalid and sup(N=0) == invalid
    monkeypatch.setenv("FT_BNK_RSTAT", "2")
    assert sup(N=2) == ok and sup(W=72, H=96) == unsupported
    monkeypatch.setenv("FT_BNK_RSTAT", "0")
    assert sup() == unsupported
    monkeypatch.setenv("FT_BNK_RSTAT", "2")
    assert hip_lib.ft_bottleneck_rstat_fwd(ctypes.byref(desc()), None, None, None, None) == invalid      # NULL operands: no launch

    bn = lambda name, c: {"weight": synth.uniform(3, name + "g", (c,), 0.5, 1.5), "bias": synth.normal(3, name + "b", (c,), 0.1),
                          "running_mean": synth.normal(3, name + "m", (c,), 0.1), "running_var": synth.uniform(3, name + "v", (c,), 0.5, 1.5),
                          "eps": 1e-5}
    w1, w2, w3 = synth.normal(3, "w1", (64, 256, 1, 1), 0.08), synth.normal(3, "w2", (64, 64, 3, 3), 0.06), synth.normal(3, "w3", (256, 64, 1, 1), 0.17)
    b1, b2, b3 = bn("1", 64), bn("2", 64), bn("3", 256)
    # (FusedConv itself needs a GPU; the packer reads these fields of it)
    layer = lambda w, b: types.SimpleNamespace(cout=w.shape[0], _weight=w.float(), _bias=None, _bn=b, _packed={})
    c1, c2, c3 = layer(w1, b1), layer(w2, b2), layer(w3, b3)
    buf = hip_ops._bottleneck_rstat_weights(c1, c2, c3, torch.device("cpu"))
    assert buf.dtype == torch.float16 and buf.numel() * 2 == hip_lib.ft_bottleneck_rstat_weight_bytes() == 2 * (64 * 272 + 64 * 592 + 256 * 80)
    assert hip_ops._bottleneck_rstat_weights(c1, c2, c3, torch.device("cpu")) is buf        # built once per weight set
    pos = 0
    for w, b, k in ((w1, b1, 256), (w2, b2, 576), (w3, b3, 64)):
        co = w.shape[0]
        blk = buf[pos:pos + co * (k + 16)].reshape(co, k + 16).float()
        pos += co * (k + 16)
        scale = b["weight"].double() / torch.sqrt(b["running_var"].double() + 1e-5)
        shift = b["bias"].double() - b["running_mean"].double() * scale
        want = (w.permute(0, 2, 3, 1).reshape(co, k).double() * scale[:, None]).float()       # k = (ky * 3 + kx) * cin + ci
        # w * scale rounded ONCE to fp16 (the packer multiplies in fp32: a product on a rounding boundary may fall either way)
        assert ((blk[:, :k] - want).abs() <= want.abs() * 2.0 ** -10 + 1e-7).all(), "w * scale, rounded once"
        assert (blk[:, :k] == want.half().float()).float().mean().item() > 0.999
        assert (blk[:, k] + blk[:, k + 1] - shift.float()).abs().max().item() <= 2e-7 * max(1.0, shift.abs().max().item()), "hi + lo = shift"
        assert torch.all(blk[:, k + 2:] == 0)


def test_flip_test_pairs_follow_the_dataset():
    """tools/pose/main.py flip test: the left/right table is the dataset's (get_pairs, lib/pose/utils/transforms.py:60-95),
    not COCO's for everything; a table that does not fit the head raises instead of being skipped."""
    from tools.pose import main as pose_main
    assert pose_main.get_flip_pairs("mpii") == [(0, 5), (1, 4), (2, 3), (10, 15), (11, 14), (12, 13)]
    assert pose_main.get_flip_pairs("aic") == [(0, 3), (1, 4), (2, 5), (6, 9), (7, 10), (8, 11)]
    assert sorted(map(sorted, pose_main.get_flip_pairs("coco"))) == [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]
    with pytest.raises(ValueError):
        pose_main.get_flip_pairs("lsp")
    hm = torch.arange(2 * 16 * 3 * 4, dtype=torch.float32).reshape(2, 16, 3, 4)
    back = pose_main._flip_back(hm, pose_main.get_flip_pairs("mpii"))
    perm = list(range(16))
    for a, b in [(0, 5), (1, 4), (2, 3), (10, 15), (11, 14), (12, 13)]:
        perm[a], perm[b] = perm[b], perm[a]
    assert torch.equal(back, torch.flip(hm, dims=[3])[:, perm])
    with pytest.raises(ValueError):
        pose_main._flip_back(hm[:, :14], pose_main.get_flip_pairs("mpii"))


def test_program_choice_groups_bookkeeping(hip_lib, monkeypatch):
    """Program.begin_choice / option / end_choice: every option is recorded, resolve keeps one (the first, or the cached
    benchmark pick), nested groups resolve inside-out, flops count once and the per-launch records are re-indexed."""
    from flowtrack.pytorch_amd import hip_ops
    monkeypatch.setattr(hip_ops, "_TILE_CACHE", {})
    monkeypatch.setattr(hip_ops, "_TILE_CACHE_LOADED", True)

    def build():
        prog = hip_ops.Program(None)
        prog.add("a")
        prog.flops += 1.0
        prog.begin_choice("blk")
        prog.option("fused")                            # option 0: one fused launch
        prog.flops += 10.0
        prog.fused_records.append(("blk.fused", len(prog.calls), 10.0))
        prog.add("fused")
        prog.option("convs")                            # option 1: three launches, the middle one itself a choice
        prog.flops += 3.0
        prog.conv_records.append(("c1", len(prog.calls), 3.0, None))
        prog.add("c1")
        prog.begin_choice("c2")
        prog.option("direct")
        prog.flops += 4.0
        prog.conv_records.append(("c2.direct", len(prog.calls), 4.0, None))
        prog.add("c2_direct")
        prog.option("igemm")
        prog.flops += 4.0
        prog.conv_records.append(("c2.igemm", len(prog.calls), 4.0, None))
        prog.add("c2_igemm")
        prog.end_choice()
        prog.flops += 3.0
        prog.conv_records.append(("c3", len(prog.calls), 3.0, None))
        prog.add("c3")
        prog.end_choice()
        prog.add("z")
        return prog

    prog = build()
    assert prog.flops == 11.0                           # 1 + the block once
    prog.resolve_choices()
    assert [c[0] for c in prog.calls] == ["a", "fused", "z"] and prog.lanes == [0, 0, 0]
    assert prog.fused_records == [("blk.fused", 1, 10.0)] and prog.conv_records == []

    hip_ops._TILE_CACHE.update({"choice|blk": "convs", "choice|c2": "igemm"})      # picks are cached by option NAME
    prog = build()
    prog.resolve_choices()
    assert [c[0] for c in prog.calls] == ["a", "c1", "c2_igemm", "c3", "z"]
    assert [(r[0], r[1]) for r in prog.conv_records] == [("c1", 1), ("c2.igemm", 2), ("c3", 3)] and prog.fused_records == []

    hip_ops._TILE_CACHE.update({"choice|c2": "direct"})
    prog = build()
    prog.resolve_choices()
    assert [c[0] for c in prog.calls] == ["a", "c1", "c2_direct", "c3", "z"]

    # a stale cache — a bare index from a file written before picks were named, or a form this recording does not offer —
    # is ignored: cached_only leaves the group for the benchmark, a plain resolve falls back to the recorder's first choice
    hip_ops._TILE_CACHE.update({"choice|blk": 1, "choice|c2": "weight-stationary"})
    prog = build()
    prog.resolve_choices(cached_only=True)
    assert sum(1 for c in prog.calls if c[0] == "__choice__") == 2
    prog.resolve_choices()
    assert [c[0] for c in prog.calls] == ["a", "fused", "z"]


def test_flownet_demo_pads_by_edge_replication():
    """tools/flownet/demo.py packs a pair with the tracking glue's helper (net_utils.pad_pairs_to_64): sizes that are not
    multiples of 64 are filled by replicating the last row / column, never with zeros — the network's per-pair rgb_mean
    (lib/flownet/model/models.py:255) of a padded 1080-row frame stays the frame's own."""
    import numpy as np
    from tools.flownet import demo
    rng = np.random.default_rng(0)
    im1 = rng.integers(60, 200, (70, 100, 3)).astype(np.uint8)
    im2 = rng.integers(60, 200, (70, 100, 3)).astype(np.uint8)
    ims = demo.pack_pair(im1, im2)
    assert tuple(ims.shape) == (1, 3, 2, 128, 128) and ims.dtype == torch.float32
    assert torch.equal(ims[0, :, 0, :70, :100], torch.from_numpy(im1.astype(np.float32)).permute(2, 0, 1))
    assert torch.equal(ims[0, :, 1, :70, :100], torch.from_numpy(im2.astype(np.float32)).permute(2, 0, 1))
    assert torch.equal(ims[0, :, :, 70:, :100], ims[0, :, :, 69:70, :100].expand(-1, -1, 58, -1))      # last row repeated
    assert torch.equal(ims[0, :, :, :, 100:], ims[0, :, :, :, 99:100].expand(-1, -1, -1, 28))           # last column repeated
    assert ims.min().item() >= 60.0, "zero padding leaked in"
    same = demo.pack_pair(np.zeros((64, 128, 3), np.uint8), np.ones((64, 128, 3), np.uint8))
    assert tuple(same.shape) == (1, 3, 2, 64, 128)


def test_mean_fold_algebra_and_support_query(hip_lib):
    """The identity behind ft_flow_pack_pair_sums / ft_flow_mean_fold (FlowNet2S's rgb mean folded into conv1,
    lib/flownet/model/models.py:117-121 + FlowNetS.py:20): conv_zero-pad((x - m) / r) == conv_m-pad(x / r) - (m / r) . sum of the
    kernel, here in torch on the CPU with the fp16 roundings the device path makes; and which descriptors
    ft_conv_shift_nstride_supported accepts (no compute call)."""
    g = torch.Generator().manual_seed(3)
    B, H, W, r = 2, 20, 24, 255.0
    x = torch.rand((B, 3, 2, H, W), generator=g) * r
    w = (torch.randn((8, 6, 7, 7), generator=g) * 0.05).half().float()
    bias = torch.randn(8, generator=g)
    mean = x.view(B, 3, -1).mean(-1)
    ref_in = ((x - mean.view(B, 3, 1, 1, 1)) / r)
    ref_in = torch.cat((ref_in[:, :, 0], ref_in[:, :, 1]), 1)                       # [B, 6, H, W]
    want = torch.nn.functional.conv2d(ref_in, w, bias, stride=2, padding=3)
    m16 = (mean / r).half().float()                                                  # what the padding pixels and the shift carry
    xin = torch.cat((x[:, :, 0], x[:, :, 1]), 1) / r
    xin = xin.half().float()
    padded = m16.repeat(1, 2).view(B, 6, 1, 1).expand(B, 6, H + 6, W + 6).clone()
    padded[:, :, 3:3 + H, 3:3 + W] = xin
    wsum = w.sum(dim=(2, 3))                                                         # [Cout, 6]
    shift_n = bias[None] - m16.repeat(1, 2) @ wsum.t()                               # [B, Cout]
    got = torch.nn.functional.conv2d(padded, w, None, stride=2, padding=0) + shift_n[:, :, None, None]
    assert got.shape == want.shape
    assert (got - want).abs().max().item() <= 2e-3                                   # input rounding to fp16, nothing structural

    def fold_desc(N, H, W, **kw):
        d = _desc(N=N, Hi=H + 6, Wi=W + 6, Cin=6, x_cstride=8, Cout=64, kh=7, kw=7, stride=2, pad=0, Ho=H // 2, Wo=W // 2,
                  y_cstride=64, act=_lib.FT_ACT_LEAKY)
        d.x_wpitch = W + 6
        d.slope = 0.1
        d.shift_nstride = 64
        for k, v in kw.items():
            setattr(d, k, v)
        return d
    assert hip_lib.ft_conv_shift_nstride_supported(ctypes.byref(fold_desc(16, 384, 512))) == 0
    assert hip_lib.ft_conv_shift_nstride_supported(ctypes.byref(fold_desc(4, 256, 256))) == 0
    assert hip_lib.ft_conv_shift_nstride_supported(ctypes.byref(fold_desc(1, 64, 64))) == _lib.FT_ERR_UNSUPPORTED      # too few tiles for the persistent stem
    assert hip_lib.ft_conv_shift_nstride_supported(ctypes.byref(fold_desc(16, 384, 512, dtype=_lib.FT_F32))) == _lib.FT_ERR_UNSUPPORTED
    assert hip_lib.ft_conv_shift_nstride_supported(ctypes.byref(fold_desc(16, 384, 512, shift_nstride=62))) == _lib.FT_ERR_UNSUPPORTED
    assert hip_lib.ft_conv_shift_nstride_supported(ctypes.byref(fold_desc(16, 384, 512, pool=1))) == _lib.FT_ERR_UNSUPPORTED
    assert int(hip_lib.ft_flow_pack_pair_sums_chunks(384)) == 96 and int(hip_lib.ft_flow_pack_pair_sums_chunks(50)) == 13


def test_bench_line_summary_is_flat_and_last():
    """bench.py's `summary` (round 5): a flat object of the sub-records' headline numbers, added as the LAST key of the line so that
    the driver's stdout tail holds them; missing sub-records simply do not appear."""
    import json
    import bench
    out = {"value": 56000.0, "roofline": {"frac": 0.245}, "same_batch_replay": {"value": 55900.0},
           "flow": {"value": 19000.0, "roofline": {"frac": 0.25}, "cpu_baseline": {"value": 30.0},
                    "roofline_ops": [{"kernel": "ft_channelnorm_fwd", "frac": 0.55}, {"kernel": "ft_resample2d_fwd_smooth_flow", "frac": 0.32}]},
           "c3_per_gpu": {"value": 11300.0, "roofline": {"frac": 0.185}},
           "parity": {"fp16": {"argmax_identical_frac": 0.99, "heatmap_max_abs_err": 0.0026, "mAP_at_OKS": 0.99}, "fp32": {"argmax_identical_frac": 1.0}},
           "fp16_exact_argmax": {"value": 8500.0, "no_rerun_path": {"value": 52000.0}, "no_rerun_path_in_graph": {"value": 55000.0}},
           "rccl": {"world": 8, "ranks_verified": 8}}
    s = bench.summary_of(out)
    assert s["pose_crops_s"] == 56000.0 and s["flow_frac"] == 0.25 and s["c3_frac"] == 0.185 and s["rccl_ranks_verified"] == 8
    assert s["channelnorm_frac"] == 0.55 and s["resample2d_smooth_flow_frac"] == 0.32 and s["exact_no_rerun_in_graph_crops_s"] == 55000.0
    assert "clip_frames_s" not in s and all(not isinstance(v, (dict, list)) for v in s.values())
    out["summary"] = s
    line = json.dumps(out, separators=(",", ":"))
    assert line.rstrip("}").rsplit('"summary":', 1)[1].startswith("{") and len(line) < 4000


def test_folded_bottleneck_operands_reproduce_batchnorm():
    """hip_ops._folded_kmajor / shift_pairs (round 6: the operands of ft_bottleneck_stream_fwd with d.folded = 1), host side only: the
    fp16 weights carry conv * scale with ONE rounding from the fp32 weights, in the K-major order the pack kernel expects
    ([co][(ky * 3 + kx) * cin + ci]), and the (hi, lo) fp16 pair restores the fp32 shift to ~2^-21 relative: a conv with the folded
    weights plus hi + lo equals eval-mode BatchNorm(conv) up to the fp16 rounding of the weights."""
    import torch.nn.functional as F
    from flowtrack.pytorch_amd import hip_ops, synth
    co, ci = 32, 48
    w = synth.normal(5, "fold.w", (co, ci, 3, 3), std=0.1)
    bn = {"weight": synth.uniform(5, "fold.g", (co,), -1.5, 1.5), "bias": synth.normal(5, "fold.b", (co,), 40.0),
          "running_mean": synth.normal(5, "fold.m", (co,), 0.3), "running_var": synth.uniform(5, "fold.v", (co,), 0.5, 1.5), "eps": 1e-5}
    conv = types.SimpleNamespace(cout=co, _weight=w, _bias=None, _bn=bn)
    wf, shift = hip_ops._folded_kmajor(conv)
    assert wf.dtype == torch.float16 and tuple(wf.shape) == (co, 9 * ci)
    scale = bn["weight"].double() / torch.sqrt(bn["running_var"].double() + 1e-5)
    want_w = (w.double() * scale[:, None, None, None]).permute(0, 2, 3, 1).reshape(co, -1)
    assert torch.equal(wf, want_w.float().half()), "one rounding from the fp32 product, K-major (tap, channel) order"
    pairs = hip_ops.shift_pairs(shift)
    assert pairs.dtype == torch.int32 and pairs.shape == (co,)
    halves = pairs.view(torch.float16).reshape(co, 2).float()
    assert torch.equal(halves[:, 0], shift.half().float())                       # hi in the low 16 bits (little endian), lo above it
    rel = ((halves[:, 0].double() + halves[:, 1].double()) - shift.double()).abs() / shift.double().abs().clamp_min(1e-6)
    assert rel.max().item() < 2.0 ** -20, rel.max().item()
    x = synth.normal(5, "fold.x", (2, ci, 9, 7)).half().float()
    ref = F.batch_norm(F.conv2d(x, w, padding=1), bn["running_mean"], bn["running_var"], bn["weight"], bn["bias"], training=False, eps=1e-5)
    w_back = wf.float().reshape(co, 3, 3, ci).permute(0, 3, 1, 2).contiguous()
    got = F.conv2d(x, w_back, padding=1) + (halves[:, 0] + halves[:, 1])[None, :, None, None]
    assert (got - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()
    # the guard that sends unrepresentable operands to the table form
    bn_big = dict(bn, bias=bn["bias"].clone())
    bn_big["bias"][3] = 1.0e5
    assert hip_ops._fold_representable((conv,)) and not hip_ops._fold_representable((types.SimpleNamespace(cout=co, _weight=w, _bias=None, _bn=bn_big),))
