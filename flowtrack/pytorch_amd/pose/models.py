"""Pose network of FlowTrack on MI355X: ResNet-50/101/152 trunk + 3x ConvTranspose(4,2,1) head +
1x1 heatmap conv, executed as fused HIP launches (libflowtrack_hip.so).

Drop-in surface (reference: lib/pose/models/pose_deconv.py:12-63,166-179, resnet.py:15-55,
blocks.py:83-120):
    models.deconv(backbone: str, num_classes: int, pretrained: bool) -> module
    module.forward(x[B,3,H,W]) -> heatmaps [B,K,H/4,W/4] (fp32)
    module.state_dict() / load_state_dict() / init_net(path) / .cuda() / .half() / .eval()
with the reference's parameter names and shapes, so its checkpoints
(`ckpt['state_dict']`, tools/pose/main.py:176-181) and torchvision-keyed ImageNet backbones
(`data/pretrained/<backbone>.pth`, loaded strict=False) load unchanged.

What differs by design: eval-mode BatchNorm and ReLU / the residual add are folded into each
conv's epilogue; activations live in NHWC (fp16 or fp32) HBM buffers allocated once per input
shape; the ~60 launches of one forward are replayed as a HIP graph.
"""
from __future__ import annotations

import math
import os
from typing import List

import torch
import torch.nn as nn

from ..engine import HipModule
from ..hip_ops import (ActView, FlowtrackHipError, FusedConv, FusedShortcutConv, Program, bottleneck_fusable,
                       bottleneck_entry_fusable, bottleneck_head_fusable, bottleneck_cluster_supported, bottleneck_strips_supported, bottleneck_head_stream_fusable, bottleneck_prefers_fused, new_act, new_rowpacked_act, record_bottleneck, record_bottleneck_entry, record_bottleneck_head, record_bottleneck_head_stream,
                       record_maxpool, record_pack_input)
from ..params import ActMarker, BatchNormParams, ConvParams, ConvTransposeParams

# depth -> blocks per stage; only Bottleneck nets are valid because the head hard-codes 2048
# input channels (pose_deconv.py:20), see SURVEY §8 P1.
resnet_dict = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}


class Bottleneck(nn.Module):
    """Parameter layout of blocks.py:83-103 (stride on the 3x3, torchvision-compatible)."""
    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int = 1):
        super().__init__()
        self.stride = stride
        self.conv1 = ConvParams(inplanes, planes, 1, bias=False)
        self.bn1 = BatchNormParams(planes)
        self.conv2 = ConvParams(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = BatchNormParams(planes)
        self.conv3 = ConvParams(planes, planes * 4, 1, bias=False)
        self.bn3 = BatchNormParams(planes * 4)
        self.relu = ActMarker("relu")
        if stride != 1 or inplanes != planes * 4:
            self.downsample = nn.Sequential(ConvParams(inplanes, planes * 4, 1, stride=stride, bias=False),
                                            BatchNormParams(planes * 4))
        else:
            self.downsample = nn.Sequential()


class _PosePlan:
    def __init__(self, prog, x_static, heatmaps):
        self.prog, self.x_static, self.heatmaps = prog, x_static, heatmaps
        self.runs = 0


class DeconvResnet(HipModule):
    #: fold each block-0 projection shortcut into its conv3 launch (FusedShortcutConv); FT_FUSE_SHORTCUT=0 keeps them apart
    fuse_shortcut: bool = os.environ.get("FT_FUSE_SHORTCUT", "1") != "0"
    #: identity-shortcut blocks of the 256-wide stage as one launch (ft_bottleneck_fwd); FT_FUSE_BOTTLENECK=0 keeps 3 convs
    fuse_bottleneck: bool = os.environ.get("FT_FUSE_BOTTLENECK", "1") != "0"
    #: the stem's max-pool inside the stem conv launch (ft_conv_desc.pool); FT_FUSE_STEM_POOL=0 keeps the two launches
    fuse_stem_pool: bool = os.environ.get("FT_FUSE_STEM_POOL", "1") != "0"
    #: the fused stem reads the NCHW fp32 input itself (ft_conv_desc.x_nchw_f32): no pack launch, no packed copy, bit-identical
    #: results; FT_FUSE_STEM_PACK=0 keeps ft_pack_nchw_to_nhwc in front of it.  Batch 64 x 256x192, same box: pack 18.7 us + stem 38.5
    #: us -> stem 46.2 us, 59.52 / 59.31 -> 59.71 / 59.61 k crops/s (the first version, predicated global loads instead of
    #: out-of-range buffer loads, ran the stem at 66 us: every load in its own divergent region behind a full wait)
    fuse_stem_pack: bool = os.environ.get("FT_FUSE_STEM_PACK", "1") != "0"
    #: None: plans end at the heatmaps (the reference's forward).  True / False: plans also run max_preds on the heatmaps
    #: (with / without the adjust_coords nudge, lib/pose/utils/evaluation.py:11-35) and forward_keypoints() returns them
    keypoints_in_plan = None
    #: record the exact-arg-max mode's screen + flagged-crop gather inside fp16 plans (needs keypoints_in_plan); see exact_submit_plan
    exact_in_plan: bool = False
    #: dev: layer1 + layer2 as two half-batch lanes (parallel graph branches); FT_SPLIT_LANES=1
    split_lanes4: bool = os.environ.get("FT_SPLIT_LANES4", "0") == "1"       # dev: layer4 as two half-batch lanes
    split_lanes: int = int(os.environ.get("FT_SPLIT_LANES", "0") or 0)     # 1: lanes start together; 2: the second lane starts with its half of the stem
    #: run the 1x1 heatmap conv as the fused tail of the last deconv (fp16 mode); FT_FUSE_HEATMAP=0 keeps it a launch
    fuse_heatmap: bool = os.environ.get("FT_FUSE_HEATMAP", "1") != "0"
    #: forward_keypoint_rows_exact(): a crop is re-run in the fp32 parity arithmetic when ft_heatmap_argmax_screen flags it:
    #: E = exact_argmax_rel_bound x (range of the crop's fp16 heat maps) is taken as the bound of the fp16 mode's heat-map error,
    #: and a crop is flagged when two values of a map are closer than 2 E (an error of at most E per element cannot reorder
    #: values further apart), when a map's maximum is within E of 0 (the `score > 0` coordinate mask of max_preds could flip), or
    #: when anything is non-finite.  The bound is RELATIVE because the error follows the maps' range (the tests bound it as a
    #: fraction of the range); 1.6e-3 = 2 x the largest measured ratio (4.07e-3 on a range of 5.0 = 0.81e-3)
    exact_argmax_rel_bound: float = 1.6e-3

    def __init__(self, layers: List[int], num_classes: int):
        super().__init__()
        self.layers_cfg = list(layers)
        self.num_classes = num_classes
        self.inplanes = 64
        # trunk (resnet.py:19-27)
        self.conv1 = ConvParams(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = BatchNormParams(64)
        self.relu = ActMarker("relu")
        self.maxpool = ActMarker("maxpool3x3s2")
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1], stride=2)
        self.layer3 = self._make_layer(256, layers[2], stride=2)
        self.layer4 = self._make_layer(512, layers[3], stride=2)
        # head (pose_deconv.py:16-30)
        self.deconv_bias = False
        feats = 256
        self.deconv = nn.Sequential(
            ConvTransposeParams(2048, feats, bias=False), BatchNormParams(feats), ActMarker("relu"),
            ConvTransposeParams(feats, feats, bias=False), BatchNormParams(feats), ActMarker("relu"),
            ConvTransposeParams(feats, feats, bias=False), BatchNormParams(feats), ActMarker("relu"))
        self.heatmap = ConvParams(feats, num_classes, 1, bias=True)

    def _make_layer(self, planes: int, blocks: int, stride: int = 1) -> nn.Sequential:
        mods = [Bottleneck(self.inplanes, planes, stride)]
        self.inplanes = planes * 4
        mods += [Bottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*mods)

    # -- initialisation / loading (resnet.py:38-49, pose_deconv.py:48-63) ---------------------
    def init_weights(self, pretrained: str = "") -> None:
        if os.path.isfile(pretrained):
            print("=> loading pretrained model {}".format(pretrained))
            self.load_state_dict(torch.load(pretrained, map_location="cpu"), strict=False)
            return
        for m in self.modules():
            if isinstance(m, ConvParams):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, BatchNormParams):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self._invalidate()

    def init_net(self, pretrained: str = "") -> None:
        self.init_weights(pretrained)
        for m in self.deconv:
            if isinstance(m, ConvTransposeParams):
                nn.init.normal_(m.weight, std=0.001)
            elif isinstance(m, BatchNormParams):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        nn.init.normal_(self.heatmap.weight, std=0.001)
        nn.init.constant_(self.heatmap.bias, 0)
        self._invalidate()

    # -- plan construction ----------------------------------------------------------------------
    def _build_plan(self, B: int, H: int, W: int, device, dtype) -> _PosePlan:
        if H % 32 or W % 32:
            raise FlowtrackHipError(f"input {H}x{W}: height and width must be multiples of 32")
        prog = Program(self._side_stream(device))
        mk = dict(dtype=dtype, device=device)
        x_static = torch.empty((B, 3, H, W), dtype=torch.float32, device=device)
        # 7x7/s2/p3 stem: one kernel row = one K-run.  fp16: conv1 + bn1 + relu + maxpool (resnet.py:19-23) are ONE launch
        # (ft_conv_desc.pool; its patch starts one stem column further left: 5 physical pad columns instead of 3) and the
        # [B, H/2, W/2, 64] stem map never exists
        pool_in_stem = self.fuse_stem_pool and dtype == torch.float16
        planar_in = pool_in_stem and self.fuse_stem_pack      # the stem gathers its patches from x_static: a_in is geometry only
        a_in = new_rowpacked_act(B, H, W, 3, 5 if pool_in_stem else 3, dtype, "meta" if planar_in else device)
        if not planar_in:
            record_pack_input(prog, x_static, a_in)
        xs = (lambda lo, hi: x_static[lo:hi]) if planar_in else (lambda lo, hi: None)

        stem = self.fused("conv1" + ("+maxpool" if pool_in_stem else ""), self.conv1.weight, stride=2, pad=3, bn=self.bn1.as_dict(),
                          act="relu", **mk)
        cur = new_act(B, H // 4, W // 4, 64, dtype, device)
        split = self.split_lanes if (dtype == torch.float16 and B % 2 == 0 and B >= 32 and pool_in_stem) else 0
        if split == 2:
            stem.record(prog, a_in.batch_slice(0, B // 2), cur.batch_slice(0, B // 2), pool=True, x_nchw=xs(0, B // 2))     # (the other half: on the second lane)
        elif pool_in_stem:
            stem.record(prog, a_in, cur, pool=True, x_nchw=xs(0, B))
        else:
            a1 = new_act(B, H // 2, W // 2, 64, dtype, device)
            stem.record(prog, a_in, a1)
            record_maxpool(prog, a1, cur)

        # FT_SPLIT_LANES=1 (dev, measured in profiles/README.md): layer1 + layer2 as TWO half-batch lanes (parallel graph branches).
        # Their fused blocks are phase-locked across the chip — every workgroup reads its input, then multiplies, then writes, at
        # the same time, so HBM idles while the matrix pipe runs and vice versa; two lanes half a block apart overlap them.
        stages = {"stem": cur}         # activation views behind each stage (diagnostics: tests/error_budget.py)
        for li, layer in enumerate((self.layer1, self.layer2, self.layer3, self.layer4), start=1):
            if li > 1:
                stages[f"layer{li - 1}"] = cur
            if split and li == 1:
                chain = []
                hh, ww = cur.H, cur.W
                for lj, lay in ((1, self.layer1), (2, self.layer2)):
                    for bi, blk in enumerate(lay):
                        hh, ww = hh // blk.stride, ww // blk.stride
                        chain.append((f"layer{lj}.{bi}", blk, new_act(B, hh, ww, blk.conv1.cout * 4, dtype, device)))
                prog.fork()
                for half, (lo, hi) in enumerate(((0, B // 2), (B // 2, B))):
                    c = cur.batch_slice(lo, hi)
                    if half == 1:
                        with prog.side():
                            if split == 2:      # half a block of useful delay: the lanes' memory and matrix phases interleave
                                stem.record(prog, a_in.batch_slice(lo, hi), cur.batch_slice(lo, hi), pool=True, x_nchw=xs(lo, hi))
                            for name, blk, out_full in chain:
                                o = out_full.batch_slice(lo, hi)
                                self._record_block(prog, name, blk, c, o, dtype, device)
                                c = o
                    else:
                        for name, blk, out_full in chain:
                            o = out_full.batch_slice(lo, hi)
                            self._record_block(prog, name, blk, c, o, dtype, device)
                            c = o
                prog.join()
                cur = chain[-1][2]
                continue
            if split and li == 2:
                continue
            if self.split_lanes4 and li == 4 and dtype == torch.float16 and B % 2 == 0 and B >= 32:
                # FT_SPLIT_LANES4=1 (dev): layer4 as two half-batch lanes.  Its nine launches are small (15-30 us each, of which ~6 us
                # are the launch's ramp with the chip mostly idle: cd_phases.py shows 10-us workgroup lifetimes in 16-us kernels); two
                # lanes fill each other's ramps, and halving the pixels halves the workgroups, not the weight bytes per workgroup.
                # MEASURED (two interleaved bench runs): 56.37 / 55.82 -> 53.75 / 53.79 k crops/s: off.
                chain = []
                hh, ww = cur.H, cur.W
                for bi, blk in enumerate(layer):
                    hh, ww = hh // blk.stride, ww // blk.stride
                    chain.append((f"layer4.{bi}", blk, new_act(B, hh, ww, blk.conv1.cout * 4, dtype, device)))
                prog.fork()
                for half, (lo, hi) in enumerate(((0, B // 2), (B // 2, B))):
                    c = cur.batch_slice(lo, hi)

                    def lane(c=c, lo=lo, hi=hi):
                        for name, blk, out_full in chain:
                            o = out_full.batch_slice(lo, hi)
                            self._record_block(prog, name, blk, c, o, dtype, device)
                            c = o
                    if half == 1:
                        with prog.side():
                            lane()
                    else:
                        lane()
                prog.join()
                cur = chain[-1][2]
                continue
            for bi, blk in enumerate(layer):
                name = f"layer{li}.{bi}"
                out = new_act(B, cur.H // blk.stride, cur.W // blk.stride, blk.conv1.cout * 4, dtype, device)
                self._record_block(prog, name, blk, cur, out, dtype, device)
                cur = out

        # head: 3 x (ConvTranspose 4/2/1 + bn + relu), then the 1x1 heatmap conv (pose_deconv.py:43-45) — fused behind
        # the last deconv as its tail in fp16 mode: the [B, 64, 48, 256] map, the largest tensor of the head, never
        # reaches HBM
        stages["layer4"] = cur
        fuse_tail = self.fuse_heatmap and dtype == torch.float16 and self.num_classes <= 32 and self.deconv[6].cout in (64, 128, 256)
        heatmaps = None
        for i in (0, 3, 6):
            tail = dict(tail_weight=self.heatmap.weight, tail_bias=self.heatmap.bias) if (i == 6 and fuse_tail) else {}
            dc = self.fused(f"deconv.{i}" + ("+heatmap" if tail else ""), self.deconv[i].weight, transposed=True, stride=2, pad=1,
                            bias=self.deconv[i].bias, bn=self.deconv[i + 1].as_dict(), act="relu", **tail, **mk)
            if tail:
                heatmaps = torch.empty((B, self.num_classes, cur.H * 2, cur.W * 2), dtype=torch.float32, device=device)
                dc.record(prog, cur, heatmaps)
            else:
                nxt = new_act(B, cur.H * 2, cur.W * 2, dc.cout, dtype, device)
                dc.record(prog, cur, nxt)
                cur = nxt
                stages[f"deconv.{i}"] = cur
        if heatmaps is None:
            hm = self.fused("heatmap", self.heatmap.weight, bias=self.heatmap.bias, act=None, **mk)
            heatmaps = torch.empty((B, self.num_classes, cur.H, cur.W), dtype=torch.float32, device=device)
            hm.record(prog, cur, heatmaps)
        plan = _PosePlan(prog, x_static, heatmaps)
        plan.stages = stages
        if self.keypoints_in_plan is not None:
            # max_preds (+ the 0.25 px nudge of final_preds) as the last launch of the plan: no extra stream work per call
            K, hh, hw = self.num_classes, heatmaps.shape[2], heatmaps.shape[3]
            # written as (x, y, score) rows — what the tracking glue and the multi-GPU gather consume; scores / coords are views
            plan.kp_idx = torch.empty((B, K), dtype=torch.int32, device=device)
            plan.kp_rows = torch.empty((B, K, 3), dtype=torch.float32, device=device)
            plan.kp_score, plan.kp_coords = plan.kp_rows[:, :, 2:], plan.kp_rows[:, :, :2]
            prog.add("ft_heatmap_keypoint_rows", heatmaps.data_ptr(), B, K, hh, hw, int(bool(self.keypoints_in_plan)),
                     plan.kp_idx.data_ptr(), plan.kp_rows.data_ptr(), keep=(plan.kp_idx, plan.kp_rows))
            if self.exact_in_plan and dtype == torch.float16 and B <= 1024:
                # the exact-arg-max mode's device side INSIDE the plan's graph (round 5): screen -> flag compaction + gather of the
                # flagged crops' inputs -> header {count, indices} to pinned memory.  As graph nodes they cost their kernel time
                # (~10 us); as five eager launches behind the replay they cost ~70 us of gaps per step.  exact_rel_bound is frozen
                # into the plan (DeconvResnet.exact_submit_plan / exact_finish_plan).
                import ctypes
                ex = {"flags": torch.empty(B, dtype=torch.int32, device=device), "stats": torch.empty((B, 4), dtype=torch.float32, device=device),
                      "stage": torch.empty((B, 3, H, W), dtype=torch.float32, device=device),
                      "header": torch.zeros(1 + B, dtype=torch.int32, device=device), "header_host": torch.zeros(1 + B, dtype=torch.int32).pin_memory(),
                      "rel_bound": float(self.exact_argmax_rel_bound), "event": None, "busy": False}
                prog.add("ft_heatmap_argmax_screen", heatmaps.data_ptr(), B, K, hh, hw, ctypes.c_float(ex["rel_bound"]), ex["flags"].data_ptr(),
                         ex["stats"].data_ptr(), keep=(ex["flags"], ex["stats"]))
                prog.add("ft_gather_flagged_rows", ex["flags"].data_ptr(), B, x_static.data_ptr(), ctypes.c_longlong(3 * H * W * 4), ex["stage"].data_ptr(),
                         ex["header"].data_ptr(), keep=(ex["stage"], ex["header"]))
                prog.add("ft_memcpy_async", ex["header_host"].data_ptr(), ex["header"].data_ptr(), ctypes.c_size_t((1 + B) * 4), keep=(ex["header_host"],))
                plan.exact = ex
        return plan

    def _record_block(self, prog, name: str, blk, cur, out, dtype, device) -> None:
        """One Bottleneck (blocks.py:83-120) from view `cur` into view `out` (both may be batch slices of larger buffers): the
        fused / unfused forms are recorded as alternatives where both exist, the first-call benchmark keeps one."""
        B = cur.N
        mk = dict(dtype=dtype, device=device)
        planes = blk.conv1.cout
        s = blk.stride
        Ho, Wo = cur.H // s, cur.W // s
        c1 = self.fused(name + ".conv1", blk.conv1.weight, bn=blk.bn1.as_dict(), act="relu", **mk)
        c2 = self.fused(name + ".conv2", blk.conv2.weight, stride=s, pad=1, bn=blk.bn2.as_dict(), act="relu", **mk)
        c3 = self.fused(name + ".conv3", blk.conv3.weight, bn=blk.bn3.as_dict(), act="relu", **mk)
        if self.fuse_bottleneck and not len(blk.downsample) and s == 1 and bottleneck_fusable(c1, c2, c3, cur, out):
            # the whole block in one launch (t1 / t2 never leave LDS) or as three conv launches: which one is faster
            # depends on how many workgroups the stage's pixels make (each streams the block's whole weight set),
            # so both are recorded and the first-call benchmark keeps one (Program.tune_choices)
            t1 = new_act(B, cur.H, cur.W, planes, dtype, device)
            t2 = new_act(B, Ho, Wo, planes, dtype, device)

            def fused_form():
                record_bottleneck(prog, c1, c2, c3, cur, out, name + ".fused", form="patch")

            def conv_form():
                c1.record(prog, cur, t1)
                c2.record(prog, t1, t2)
                c3.record(prog, t2, out, residual=cur)
            forms = (("fused", fused_form), ("convs", conv_form))
            if not bottleneck_prefers_fused(cur, planes):
                forms = forms[::-1]
            if self.cluster_kernels and bottleneck_cluster_supported(cur, out, planes):
                # round 5: the same block shared by a cluster of four workgroups per image (t1 / t2 exchanged inside the
                # launch): a third recorded form, kept where the first-call benchmark measures it faster
                forms = forms + (("cluster", lambda: record_bottleneck(prog, c1, c2, c3, cur, out, name + ".cluster", cluster=True)),)
            if self.strip_kernels and planes == 64 and bottleneck_strips_supported(cur, out, planes):
                # round 5: the 64-plane block on register-stationary strips (csrc/bottleneck_rstat.hip): 54 against 57.5 us alone
                # and with warm caches, 59-63 against ~55 us behind a cold L2 (its prologue waits for 139 KB of weights before the
                # first x load): recorded only with FT_STRIP_KERNELS=1 — the first-call benchmark times a group's later options
                # behind warmer caches than its first, a bias larger than this form's margin
                strips = (("strips", lambda: record_bottleneck(prog, c1, c2, c3, cur, out, name + ".strips", form="strips")),)
                forms = strips + forms if os.environ.get("FT_STRIP_KERNELS") == "2" else forms + strips      # (2: the first choice, dev)
            prog.begin_choice(f"bottleneck|{B},{cur.H},{cur.W},{cur.C},{planes},{cur.cstride},{out.cstride}")
            for form_name, form in forms:
                prog.option(form_name)
                form()
            prog.end_choice()
            return
        t2 = new_act(B, Ho, Wo, planes, dtype, device)
        fused = None
        if len(blk.downsample) and self.fuse_shortcut:
            key = (name + ".conv3+downsample", dtype, str(device))
            fused = self._layers.get(key)
            if fused is None:
                fused = self._layers[key] = FusedShortcutConv(
                    blk.conv3.weight, blk.bn3.as_dict(), blk.downsample[0].weight, blk.downsample[1].as_dict(), s,
                    dtype=dtype, device=device, act="relu", label=name + ".conv3+downsample")
        whole = self.fuse_bottleneck and fused is not None and bottleneck_entry_fusable(c1, c2, fused, cur, out)
        if whole:
            # the 64-wide entry block whole (t1, t2 and the shortcut never leave the CU) or as head + K-concatenated
            # conv3: both recorded, the first-call benchmark keeps one
            prog.begin_choice(f"entry|{B},{cur.H},{cur.W},{cur.C},{planes},{cur.cstride},{out.cstride}")
            prog.option("fused")
            record_bottleneck_entry(prog, c1, c2, fused, cur, out, name + ".fused")
            prog.option("head+exit")
        if self.fuse_bottleneck and s == 1 and bottleneck_head_fusable(c1, c2, cur, t2):
            record_bottleneck_head(prog, c1, c2, cur, t2, name + ".conv1+conv2")   # the 64-wide entry block: t1 in LDS only
        else:
            head2 = self.fuse_bottleneck and s == 2 and bottleneck_head_stream_fusable(c1, c2, cur, t2)
            if head2:
                # the 256-plane entry block's conv1 + stride-2 conv2 as one streamed-weights launch, or as two launches:
                # both recorded, the first-call benchmark keeps one (the recorder's first choice: two launches)
                prog.begin_choice(f"head2|{B},{cur.H},{cur.W},{cur.C},{planes},{cur.cstride},{t2.cstride}")
                prog.option("convs")
            t1 = new_act(B, cur.H, cur.W, planes, dtype, device)
            c1.record(prog, cur, t1)
            c2.record(prog, t1, t2)
            if head2:
                prog.option("fused")
                record_bottleneck_head_stream(prog, c1, c2, cur, t2, name + ".conv1+conv2")
                prog.end_choice()
        if fused is not None:
            # conv3 + bn3 and the projection shortcut as one GEMM over K = [t2 | block input] (blocks.py:104-119):
            # the shortcut tensor never exists in HBM
            fused.record(prog, t2, cur, out)
        elif len(blk.downsample):
            ds = self.fused(name + ".downsample", blk.downsample[0].weight, stride=s, bn=blk.downsample[1].as_dict(),
                            act=None, **mk)
            res = new_act(B, Ho, Wo, planes * 4, dtype, device)
            ds.record(prog, cur, res)
            c3.record(prog, t2, out, residual=res)
        else:
            c3.record(prog, t2, out, residual=cur)  # relu(bn3(conv3) + residual), blocks.py:114-119
        if whole:
            prog.end_choice()


    #: offer the cluster form of the 256-plane identity blocks (ft_bottleneck_cluster_fwd) to the first-call benchmark
    #: (FT_CLUSTER_KERNELS=1).  OFF by default: measured in situ at R50 batch 64 it takes 50-52 us per block against the strip
    #: form's 48 (each of its two in-launch hand-offs costs 4-5 us: profiles/README.md, round 5), and its workgroups wait for
    #: each other inside the launch (bounded spins: nothing deadlocks, but a timed-out hand-off gives a wrong block).
    cluster_kernels: bool = os.environ.get("FT_CLUSTER_KERNELS", "0") == "1"
    strip_kernels: bool = os.environ.get("FT_STRIP_KERNELS", "0") in ("1", "2")      # the 64-plane block's register-stationary form as an option

    def plan_for(self, B: int, H: int, W: int, replica: int = 0) -> _PosePlan:
        """The plan (launch list + activation buffers + graph) of one input shape.  `replica` > 0 gives an independent copy
        with its own buffers — same packed weights, same tile picks — for callers that keep several batches of one shape
        in flight on different streams (tracking.PoseRunner per clip, tools/tracking/demo.run_clips)."""
        device, dtype = self._resolve()
        key = (B, H, W, device, dtype, self.keypoints_in_plan) + (("exact", float(self.exact_argmax_rel_bound)) if self.exact_in_plan else ()) + \
            ((replica,) if replica else ())
        plan = self._plans.get(key)
        if plan is None:
            with torch.no_grad():
                plan = self._build_plan(B, H, W, device, dtype)
            self._plans[key] = plan
        return plan

    def static_input(self, B: int, H: int, W: int) -> torch.Tensor:
        """The plan's own input buffer [B,3,H,W] fp32 (the fixed address its graph reads).  A producer that writes its
        batch here and passes this tensor to forward() skips the 38 MB staging copy per call (zero-copy binding)."""
        return self.plan_for(B, H, W).x_static

    @torch.no_grad()
    def forward(self, x: torch.Tensor, copy_output: bool = True) -> torch.Tensor:
        """x: [B,3,H,W] on the model's GPU -> heatmaps [B,K,H/4,W/4] fp32 (pose_deconv.py:32-46)."""
        self._check_eval()
        if x.dim() != 4 or x.shape[1] != 3:
            raise FlowtrackHipError(f"expected [B,3,H,W], got {tuple(x.shape)}")
        B, _, H, W = x.shape
        plan = self.plan_for(B, H, W)
        if x.device != plan.x_static.device:
            raise FlowtrackHipError("input and model are on different devices")
        if x.data_ptr() != plan.x_static.data_ptr():   # zero-copy when the caller filled static_input() in place
            plan.x_static.copy_(x)  # dtype cast + staging into the graph's fixed input address
        self._run_plan(plan.prog, first=plan.runs == 0)
        plan.runs += 1
        self._last_plan = plan
        return plan.heatmaps.clone() if copy_output else plan.heatmaps

    @torch.no_grad()
    def replay(self, plan: "_PosePlan") -> "_PosePlan":
        """Run a plan whose static input the caller filled in place (plan_for(...).x_static): forward() without the second
        plan look-up and the shape checks — the per-call host cost matters to the tracking pass, one small batch per frame."""
        self._check_eval()
        self._run_plan(plan.prog, first=plan.runs == 0)
        plan.runs += 1
        self._last_plan = plan
        return plan

    @torch.no_grad()
    def forward_keypoints(self, x: torch.Tensor):
        """forward() with max_preds inside the plan (set `keypoints_in_plan` first): returns the plan's static buffers
        (heatmaps [B,K,h,w], idx int32 [B,K], scores [B,K,1], coords [B,K,2] in heatmap pixels), valid until the next call."""
        if self.keypoints_in_plan is None:
            raise FlowtrackHipError("set model.keypoints_in_plan = True / False (adjust_coords) before forward_keypoints()")
        hm = self.forward(x, copy_output=False)
        plan = self._last_plan
        return hm, plan.kp_idx, plan.kp_score, plan.kp_coords

    @torch.no_grad()
    def forward_keypoint_rows(self, x: torch.Tensor) -> torch.Tensor:
        """forward_keypoints() returning only the plan's [B,K,3] (x, y, score) rows in heatmap pixels (static buffer, valid
        until the next call): the form `lib/tracking/net_utils.py` concatenates preds and maxvals into."""
        self.forward_keypoints(x)
        return self._last_plan.kp_rows


    # ---- the exact-arg-max mode without a host read inside the step (round 5) -------------------------------------------------
    # forward_keypoint_rows_exact() reads the B flags on the host between the fp16 plan and the re-run: with nothing flagged that
    # read (a stream synchronisation per step) cost 10 % of the step.  Here the step ends on the device: fp16 plan -> screen ->
    # ft_gather_flagged_rows (flag compaction + copy of the flagged crops' INPUTS into a staging slot, all on the device) -> an
    # asynchronous copy of the small header {count, indices} to pinned memory -> event.  exact_finish(), called one step later
    # (or whenever the caller needs the rows), waits for that event — long past — and only if the count is non-zero runs the fp32
    # plan on the staged crops and patches their rows.  The caller's x may be overwritten as soon as exact_submit() returns.
    def _exact_slots(self, B: int, H: int, W: int, device):
        key = (B, H, W, str(device))
        st = self.__dict__.setdefault("_exact_state", {})
        ent = st.get(key)
        if ent is None:
            import ctypes
            from .. import _lib
            lib = _lib.load()
            slots = []
            for _ in range(2):
                ev = ctypes.c_void_p()
                if lib.ft_event_create(ctypes.byref(ev)) != 0:
                    raise FlowtrackHipError("ft_event_create failed")
                slots.append({"stage": torch.empty((B, 3, H, W), dtype=torch.float32, device=device),
                              "flags": torch.empty(B, dtype=torch.int32, device=device), "stats": torch.empty((B, 4), dtype=torch.float32, device=device),
                              "header": torch.zeros(1 + B, dtype=torch.int32, device=device),
                              "header_host": torch.zeros(1 + B, dtype=torch.int32).pin_memory(), "event": ev, "busy": False})
            ent = st[key] = {"slots": slots, "next": 0}
        return ent

    def release_exact_state(self) -> None:
        """Destroy the events and drop the pinned headers / staging buffers of the asynchronous exact-arg-max paths (exact_submit
        slots and the per-plan state of exact_submit_plan).  Called when the plans are invalidated (weights, dtype or device changed)
        and by close(); a step that is still in flight is abandoned (its handle must not be finished afterwards)."""
        from .. import _lib
        lib = _lib.load() if (self.__dict__.get("_exact_state") or any(getattr(pl, "exact", None) for pl in self.__dict__.get("_plans", {}).values())) else None
        for ent in self.__dict__.get("_exact_state", {}).values():
            for sl in ent["slots"]:
                if sl.get("event") is not None:
                    lib.ft_event_destroy(sl["event"])
                    sl["event"] = None
        self.__dict__["_exact_state"] = {}
        for pl in self.__dict__.get("_plans", {}).values():
            ex = getattr(pl, "exact", None)
            if ex and ex.get("event") is not None:
                lib.ft_event_destroy(ex["event"])
                ex["event"], ex["busy"] = None, False

    def _invalidate(self):
        self.release_exact_state()
        super()._invalidate()

    def close(self) -> None:
        """Release what the module holds outside torch's allocator (events of the exact-arg-max paths) and its captured plans."""
        self._invalidate()

    @torch.no_grad()
    def exact_submit(self, x: torch.Tensor) -> "_ExactHandle":
        """First half of the exact-arg-max step (see above): everything is queued on the current stream, nothing is waited for."""
        if self.keypoints_in_plan is None:
            raise FlowtrackHipError("set model.keypoints_in_plan = True / False (adjust_coords) before exact_submit()")
        import ctypes
        from .. import _lib
        from ..hip_ops import check, current_stream_handle
        lib = _lib.load()
        B, _, H, W = x.shape
        if B > 1024:
            raise FlowtrackHipError("exact_submit: at most 1024 crops per call")
        ent = self._exact_slots(B, H, W, x.device)
        sl = ent["slots"][ent["next"]]
        if sl["busy"]:
            raise FlowtrackHipError("exact_submit: both staging slots are in flight: call exact_finish() on the older handle first")
        want = self.compute_dtype
        self.compute_dtype = torch.float16
        try:
            rows = self.forward_keypoint_rows(x).clone()
            hm = self._last_plan.heatmaps
        finally:
            self.compute_dtype = want
        sh = current_stream_handle(x.device)
        check(lib.ft_heatmap_argmax_screen(hm.data_ptr(), B, hm.shape[1], hm.shape[2], hm.shape[3], ctypes.c_float(self.exact_argmax_rel_bound),
                                           sl["flags"].data_ptr(), sl["stats"].data_ptr(), sh), "ft_heatmap_argmax_screen")
        xin = self._last_plan.x_static          # (fp32 NCHW: the plan's own copy of the batch)
        check(lib.ft_gather_flagged_rows(sl["flags"].data_ptr(), B, xin.data_ptr(), 3 * H * W * 4, sl["stage"].data_ptr(), sl["header"].data_ptr(), sh),
              "ft_gather_flagged_rows")
        check(lib.ft_memcpy_async(sl["header_host"].data_ptr(), sl["header"].data_ptr(), (1 + B) * 4, sh), "ft_memcpy_async")
        check(lib.ft_event_record(sl["event"], sh), "ft_event_record")
        sl["busy"] = True
        self._last_screen_stats = sl["stats"]
        h = _ExactHandle()
        h.rows, h.slot, h.B, h.H, h.W, h.done = rows, ent["next"], B, H, W, False
        ent["next"] ^= 1
        return h

    @torch.no_grad()
    def exact_submit_plan(self, plan: "_PosePlan") -> "_PosePlan":
        """exact_submit() for a caller that owns the plan (plan_for(..., replica) with `exact_in_plan` set, its x_static filled in
        place): ONE graph replay — network, key-point rows, screen, gather, header copy — and an event.  The plan's buffers are the
        step's state, so a plan must be finished (exact_finish_plan) before it is submitted again: callers that keep several steps
        in flight rotate plan replicas."""
        ex = getattr(plan, "exact", None)
        if ex is None:
            raise FlowtrackHipError("exact_submit_plan: the plan was built without exact_in_plan (fp16, keypoints_in_plan, B <= 1024)")
        if ex["busy"]:
            raise FlowtrackHipError("exact_submit_plan: this plan's previous step was not finished")
        import ctypes
        from .. import _lib
        from ..hip_ops import check, current_stream_handle
        lib = _lib.load()
        if ex["event"] is None:
            ev = ctypes.c_void_p()
            check(lib.ft_event_create(ctypes.byref(ev)), "ft_event_create")
            ex["event"] = ev
        self.replay(plan)
        check(lib.ft_event_record(ex["event"], current_stream_handle(plan.x_static.device)), "ft_event_record")
        ex["busy"] = True
        return plan

    @torch.no_grad()
    def exact_finish_plan(self, plan: "_PosePlan"):
        """(plan.kp_rows with the flagged crops' rows replaced by the fp32 parity mode's, number of crops re-run)."""
        from .. import _lib
        from ..hip_ops import check
        ex = plan.exact
        if not ex["busy"]:
            raise FlowtrackHipError("exact_finish_plan: nothing submitted")
        try:
            check(_lib.load().ft_event_synchronize(ex["event"]), "ft_event_synchronize")
            n = int(ex["header_host"][0])
            if n:
                B, _, H, W = plan.x_static.shape
                self._rerun_flagged(ex["header_host"], ex["stage"], plan.kp_rows, n, B, H, W)
        finally:
            ex["busy"] = False
        return plan.kp_rows, n

    def _rerun_flagged(self, hdr, stage, rows, n, B, H, W):
        """The n flagged crops (inputs staged in `stage`, indices in hdr[1:]) through the fp32 plan; their rows patched into `rows`."""
        dev = rows.device
        idx = hdr[1:1 + n].to(torch.int64).to(dev)
        want, ex_flag = self.compute_dtype, self.exact_in_plan
        self.compute_dtype, self.exact_in_plan = torch.float32, False
        try:
            lo = 0
            while lo < n:
                m = min(n - lo, B)
                bucket = next(b for b in (8, 16, 32, 64, 128, 256, 1 << 30) if b >= m or b >= B)
                bucket = min(bucket, max(B, 8))
                xb = self.static_input(bucket, H, W)
                take = min(m, bucket)
                xb[:take].copy_(stage[lo:lo + take])
                if take < bucket:
                    xb[take:].zero_()
                r32 = self.forward_keypoint_rows(xb)
                rows.index_copy_(0, idx[lo:lo + take], r32[:take])
                lo += take
        finally:
            self.compute_dtype, self.exact_in_plan = want, ex_flag

    @torch.no_grad()
    def exact_finish(self, h: "_ExactHandle"):
        """Second half: (rows [B,K,3] with the parity mode's arg-max, number of crops re-run in fp32)."""
        if h.done:
            raise FlowtrackHipError("exact_finish: handle already finished")
        from .. import _lib
        from ..hip_ops import check
        lib = _lib.load()
        dev = h.rows.device
        ent = self._exact_slots(h.B, h.H, h.W, dev)
        sl = ent["slots"][h.slot]
        check(lib.ft_event_synchronize(sl["event"]), "ft_event_synchronize")
        hdr = sl["header_host"]
        n = int(hdr[0])
        h.done = True
        try:
            if n:
                self._rerun_flagged(hdr, sl["stage"], h.rows, n, h.B, h.H, h.W)
        finally:
            sl["busy"] = False       # an exception inside the fp32 re-run must not pin the slot
        return h.rows, n

    @torch.no_grad()
    def forward_keypoint_rows_exact(self, x: torch.Tensor):
        """Key-point rows [B,K,3] with the arg-max of the fp32 parity mode at (mostly) fp16 speed — north_star's "keypoint
        argmax bit-exact vs CPU reference" for the fast mode (max_preds, lib/pose/utils/evaluation.py:11-20):
          1. the fp16 plan (heat maps + rows, as forward_keypoint_rows);
          2. ft_heatmap_argmax_screen: per crop a flag (see exact_argmax_rel_bound) and its statistics;
          3. flagged crops go through the fp32 plan (buckets of 8 .. B crops), their rows replace the fp16 ones.
        A crop that is not flagged has every top-1 / top-2 margin above twice the error bound and every maximum further than the
        bound from 0, so neither its arg-max nor its coordinate mask can differ from fp32's (its score and the +-0.25 px nudge
        come from the fp16 map).  What this costs depends on the heat maps: single-peak maps (a trained net) flag next to
        nothing, noise-like maps (random weights) nearly every crop — then the mode runs at fp32 speed.  Returns (rows [B,K,3]
        on the device — a buffer of this call, not the plan's —, number of crops re-run).  One device -> host read of B flags
        per call sits between 2 and 3."""
        if self.keypoints_in_plan is None:
            raise FlowtrackHipError("set model.keypoints_in_plan = True / False (adjust_coords) before forward_keypoint_rows_exact()")
        want = self.compute_dtype
        if want not in (None, torch.float16) and next(self.parameters()).dtype != torch.float16:
            raise FlowtrackHipError("forward_keypoint_rows_exact() is the fp16 mode's exact-arg-max path: compute_dtype must be fp16")
        import ctypes
        from .. import _lib
        from ..hip_ops import check, current_stream_handle
        B = x.shape[0]
        self.compute_dtype = torch.float16
        try:
            rows = self.forward_keypoint_rows(x).clone()
            plan = self._last_plan
            hm = plan.heatmaps
            flags = torch.empty(B, dtype=torch.int32, device=x.device)
            stats = torch.empty((B, 4), dtype=torch.float32, device=x.device)
            check(_lib.load().ft_heatmap_argmax_screen(hm.data_ptr(), B, hm.shape[1], hm.shape[2], hm.shape[3],
                                                       ctypes.c_float(self.exact_argmax_rel_bound), flags.data_ptr(), stats.data_ptr(),
                                                       current_stream_handle(x.device)), "ft_heatmap_argmax_screen")
            self._last_screen_stats = stats
            todo = torch.nonzero(flags.cpu()).flatten()
            n = int(todo.numel())
            if n:
                self.compute_dtype = torch.float32
                todo_dev = todo.to(x.device)
                lo = 0
                while lo < n:
                    m = min(n - lo, B)
                    bucket = next(b for b in (8, 16, 32, 64, 128, 256, 1 << 30) if b >= m or b >= B)
                    bucket = min(bucket, max(B, 8))
                    xb = self.static_input(bucket, x.shape[2], x.shape[3])
                    sel = todo_dev[lo:lo + min(m, bucket)]
                    xb[:len(sel)].copy_(x.index_select(0, sel))
                    if len(sel) < bucket:
                        xb[len(sel):].zero_()
                    r32 = self.forward_keypoint_rows(xb)
                    rows.index_copy_(0, sel, r32[:len(sel)])
                    lo += len(sel)
        finally:
            self.compute_dtype = want
        return rows, n


class _ExactHandle:
    __slots__ = ("rows", "slot", "B", "H", "W", "done")


def deconv(backbone: str, num_classes: int, pretrained: bool) -> DeconvResnet:
    """Factory with the reference's signature (pose_deconv.py:166-179): backbone 'resnet50' |
    'resnet101' | 'resnet152'; pretrained=True loads data/pretrained/<backbone>.pth if present."""
    if not backbone.startswith("resnet"):
        raise FlowtrackHipError(f"backbone '{backbone}': only the ResNet-50/101/152 + deconv head is on the HIP path")
    depth = int(backbone[len("resnet"):])
    if depth not in resnet_dict:
        raise FlowtrackHipError(f"resnet{depth}: the deconv head needs a Bottleneck trunk (50/101/152)")
    model = DeconvResnet(resnet_dict[depth], num_classes=num_classes)
    path = os.path.join("data", "pretrained", "{}.pth".format(backbone)) if pretrained else ""
    model.init_net(path)
    return model
