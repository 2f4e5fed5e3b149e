#!/usr/bin/env python
"""Benchmark of the FlowTrack hot path on MI355X (contract: see the task brief / DESIGN.md §Measurement).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the pose hot path over one batch that is already resident in HBM:
NCHW fp32 crops -> NHWC pack -> ResNet-50 + 3x deconv + heatmap conv (HIP graph) -> per-map arg-max /
sub-pixel nudge (ft_heatmap_max_preds) [-> RCCL all-gather of the keypoint rows when N > 1].
Workload = BASELINE.json configs[1]: ResNet-50 pose head, fp16, batch 64 x 256x192 per GPU (weak scaling).
BASELINE.json's metric is "pose crops/sec + flow frame-pairs/sec ...; mAP@OKS vs CPU ref": the headline `value` is the
pose half; the default run adds, in the SAME JSON line, the sub-records
  "flow"             FlowNet2S fp16 on configs[3]'s 16 x 512x384 pairs per GPU, timed with the same protocol (its own
                     roofline, cpu_baseline and `roofline_ops` = the correlation / warp / channelnorm kernels at C4 shapes);
  "fp32_parity_mode" the pose step in the arithmetic whose arg-max is bit-exact vs the CPU reference;
  "parity"           heat-map error / identical-arg-max fraction / mAP@OKS of both modes on the benchmarked batch vs the
                     CPU oracle (N = 1 only, outside every timed region).
`--workload flow` makes FlowNet2S the headline instead; `--no-extras` prints the headline alone.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from flowtrack.pytorch_amd import parallel, synth  # noqa: E402
from flowtrack.pytorch_amd.hip_ops import is_conv_call  # noqa: E402

MFMA_PEAK_TFLOPS = {"fp16": 2500.0, "fp32": 157.3}  # dense, /opt/skills/guides/MI355X_MICROARCH.md


def build_pose(device, dtype, seed=1234, backbone="resnet50"):
    from flowtrack.pytorch_amd.pose import models
    m = models.deconv(backbone, num_classes=17, pretrained=False)
    m.load_state_dict(synth.fill_pose_state_dict(m.state_dict(), seed))
    m = m.to(device).eval()
    m.compute_dtype = dtype
    return m


def build_flow(device, dtype, seed=1234, name="FlowNet2S"):
    from flowtrack.pytorch_amd.flownet import models
    m = getattr(models, name)(types.SimpleNamespace(rgb_max=255.0, fp16=dtype == torch.float16))
    m.load_state_dict(synth.fill_flow_state_dict(m.state_dict(), seed))
    m = m.to(device).eval()
    m.compute_dtype = dtype
    return m


def pmc_traffic(workload):
    """HBM bytes per conv launch from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 gfx950 correction +
    WRITE_SIZE; tools/dev/prof_traffic.sh + pmc_traffic.py on this same command). None if not measured."""
    for tag in ("r06", "r05", "r04", "r03", "r02", "r01"):       # the latest committed round
        name = f"{tag}_{workload}_hbm_traffic_pmc.json"
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                ck = json.load(f)["conv_kernels"]
            # per launch of THAT profile's launch list (the fused-block choice can differ by one launch between boxes): the bytes
            # per forward are the comparable figure, both are given with their source
            return round(ck["hbm_bytes_per_launch_avg"], 1), {
                "file": "profiles/" + name, "launches_per_forward_in_profile": round(ck["launches_per_forward"], 2),
                "hbm_MB_per_forward": round(ck["hbm_read_MB_per_forward"] + ck["hbm_write_MB_per_forward"], 1)}
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def graph_step_ms(prog, reps=3, min_s=0.25):
    """Average time of one replay of the plan's captured graph, hipEvents on the plan's stream around >= min_s of back-to-back
    replays (median of `reps` such blocks)."""
    import ctypes
    from flowtrack.pytorch_amd._lib import check
    lib, sh = prog.lib, prog.stream_handle
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    check(lib.ft_event_create(ctypes.byref(e0))); check(lib.ft_event_create(ctypes.byref(e1)))
    prog.run(); prog.stream.synchronize()
    n, out = 8, []
    while True:                                           # size the block
        check(lib.ft_event_record(e0, sh))
        for _ in range(n):
            prog.run()
        check(lib.ft_event_record(e1, sh)); check(lib.ft_event_synchronize(e1))
        ms = ctypes.c_float(); check(lib.ft_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
        if ms.value >= 1e3 * min_s or n >= 1 << 16:
            break
        n *= 2
    for _ in range(reps):
        check(lib.ft_event_record(e0, sh))
        for _ in range(n):
            prog.run()
        check(lib.ft_event_record(e1, sh)); check(lib.ft_event_synchronize(e1))
        ms = ctypes.c_float(); check(lib.ft_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
        out.append(ms.value / n)
    lib.ft_event_destroy(e0); lib.ft_event_destroy(e1)
    return sorted(out)[len(out) // 2]


def conv_roofline(prog, dtype_name, iters=5):
    """Live hipEvent timing of the plan's launches on its own stream.
    `achieved` = conv FLOPs / conv kernel time AS THE STEP RUNS THEM: the replay time of the captured graph (>= 0.25 s of
    back-to-back replays between two events) minus the few non-conv launches (pack, arg-max: timed eagerly at the boundaries
    of their runs, a device-side spin ahead so no host latency is inside).  The inter-kernel gaps of the graph stay in the
    conv time (conservative), and conv_ms_per_step <= the step time by construction.  The eager passes (an event at the
    boundaries of each conv run / after every launch: a few us of gap per launch that the graph does not have) are kept as
    cross-checks and feed the --layers table."""
    times = prog.time_calls(iters=iters)
    per_launch_conv_ms = sum(ms for name, ms in times if is_conv_call(name))
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:               # spin-up: the passes below run at the clocks of the timed region
        for _ in range(5):
            prog.run()
        prog.stream.synchronize()
    eager_conv_ms, other_ms = prog.time_conv_runs(iters=max(iters, 50))
    if prog.graph_exec is not None:
        step_ms = graph_step_ms(prog)
        conv_ms = step_ms - other_ms
    else:
        step_ms, conv_ms = eager_conv_ms + other_ms, eager_conv_ms
    n_conv = sum(1 for name, _ in times if is_conv_call(name))
    flops = prog.flops
    # `achieved` / `frac` are over the WHOLE device-side step (graph replay incl. the non-conv launches and every gap): the lower
    # bound, and the figure anyone recomputes from ms_per_step.  The conv-launches-only figure (step minus the eagerly timed
    # non-conv launches, whose eager gaps can only inflate it: ADVICE r03) is kept beside it as `frac_conv_kernels_only`.
    achieved = flops / (step_ms * 1e-3) / 1e12 if step_ms > 0 else 0.0
    achieved_conv = flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    peak = MFMA_PEAK_TFLOPS[dtype_name]
    per_layer = []
    for label, call_idx, fl in sorted([r[:3] for r in prog.conv_records] + list(prog.fused_records), key=lambda r: r[1]):
        ms = times[call_idx][1]
        per_layer.append((label, fl, ms))
    return {
        "bound": "mfma", "kernel": "all conv launches of the step (DESIGN.md 3)", "achieved": round(achieved, 2), "peak": peak,
        "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": None,
        "achieved_conv_kernels_only": round(achieved_conv, 2), "frac_conv_kernels_only": round(achieved_conv / peak, 4),
        "launches_per_step": n_conv, "flop_per_launch_avg": flops / max(n_conv, 1),
        "avg_launch_us": round(conv_ms * 1e3 / max(n_conv, 1), 2),
        "conv_ms_per_step": round(conv_ms, 4), "graph_replay_ms_per_step": round(step_ms, 4), "other_kernels_ms_per_step": round(other_ms, 4),
        "conv_ms_per_step_eager_run_events": round(eager_conv_ms, 4), "conv_ms_per_step_event_per_launch": round(per_launch_conv_ms, 4),
    }, per_layer


def physical_cores():
    """Physical cores of this host (unique (physical id, core id) pairs of /proc/cpuinfo; SURVEY §8(d) asks for the core count
    next to the CPU number).  None if the file does not say."""
    try:
        pairs, phys, core = set(), None, None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":")[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        pairs.add((phys, core))
                    phys = core = None
        if phys is not None and core is not None:
            pairs.add((phys, core))
        return len(pairs) or None
    except OSError:
        return None


def _pick_threads(fn, budget_s=12.0):
    """Host core count the container may actually use is unknown (cgroup quotas hide behind nproc):
    try a few thread counts on one forward each and keep the fastest."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cands = sorted({c for c in (8, 16, 32, 64, 128, avail) if c <= avail}) or [1]
    best, best_t, t_start = cands[0], float("inf"), time.perf_counter()
    for c in cands:
        torch.set_num_threads(c)
        fn()  # warm the thread pool
        dt = float("inf")
        for _ in range(3):
            t0 = time.perf_counter()
            fn()
            dt = min(dt, time.perf_counter() - t0)
        if dt < best_t:
            best, best_t = c, dt
        if time.perf_counter() - t_start > budget_s:
            break
    torch.set_num_threads(best)
    return best, avail


def cpu_baseline_pose(seconds=15.0):
    """The CPU oracle (the reference's own torch-CPU arithmetic, oracle/pose_ref.py) on this host:
    ResNet-50 head, batch 4 x 256x192 fp32 (BASELINE configs[0]), all host cores."""
    from flowtrack.pytorch_amd.pose import models
    from oracle import pose_ref
    m = models.deconv("resnet50", 17, False)
    sd = synth.fill_pose_state_dict(m.state_dict(), 1234)
    x = synth.pose_crops(1, 4)
    cores, avail = _pick_threads(lambda: pose_ref.pose_forward(sd, x))
    t0, n = time.perf_counter(), 0
    while True:
        pose_ref.pose_forward(sd, x)
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds or n >= 200:
            break
    return {"value": round(4 * n / el, 2), "unit": "crops/s", "cores": cores, "physical_cores": physical_cores(), "logical_cpus": avail,
            "kind": "port", **_other_thread_counts(lambda: pose_ref.pose_forward(sd, x), 4, cores, avail),
            "sample": f"{n} fwd of 4x3x256x192 fp32 (configs[0]) in {el:.1f} s, torch CPU, `cores` = fastest thread count tried"}


def _other_thread_counts(fn, units, best, avail, budget_s=6.0):
    """the CPU figure at ALL physical cores and at 1 thread beside the fastest count (`value`)."""
    out = {}
    for key, c in (("value_all_physical_cores", min(physical_cores() or avail, avail)), ("value_1_thread", 1)):
        torch.set_num_threads(c)
        fn()
        t0, n = time.perf_counter(), 0
        while n < 1 or (time.perf_counter() - t0 < budget_s / 2 and n < 20):
            fn()
            n += 1
        out[key] = round(units * n / (time.perf_counter() - t0), 2)
        out[key + "_threads"] = c
    torch.set_num_threads(best)
    return out


def cpu_baseline_flow(seconds=15.0):
    from flowtrack.pytorch_amd.flownet import models
    from oracle import flow_ref
    m = models.FlowNet2S(types.SimpleNamespace(rgb_max=255.0, fp16=False))
    sd = synth.fill_flow_state_dict(m.state_dict(), 1234)
    x = synth.frame_pairs(1, 1, 384, 512)
    cores, avail = _pick_threads(lambda: flow_ref.flownet2s_forward(sd, x))
    t0, n = time.perf_counter(), 0
    while True:
        flow_ref.flownet2s_forward(sd, x)
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds or n >= 100:
            break
    return {"value": round(n / el, 2), "unit": "pairs/s", "cores": cores, "physical_cores": physical_cores(), "logical_cpus": avail,
            "kind": "port", **_other_thread_counts(lambda: flow_ref.flownet2s_forward(sd, x), 1, cores, avail, budget_s=4.0),
            "sample": f"{n} FlowNet2S fwd of 1x3x2x384x512 fp32 in {el:.1f} s, torch CPU, `cores` = fastest thread count tried"}


HBM_PEAK_TBS = 8.0   # HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md (6.3 TB/s achievable)


def timed_region(step, steps, device):
    """Barrier + synchronize on both sides of exactly `steps` steps; returns the MAX over ranks of the elapsed seconds."""
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    getattr(step, "drain", lambda: None)()      # the last step's exchange is consumed inside the timed region
    torch.cuda.synchronize()
    parallel.barrier()
    torch.cuda.synchronize()
    return parallel.max_over_ranks(time.perf_counter() - t0, device=device)


def spin_up(step, device, seconds=0.5):
    """Untimed: keep stepping until the GPU has been busy for `seconds`, so a timed region does not start on the idle
    clocks of a freshly woken GPU.  Every rank runs the same number of steps (step() may hold a collective): the ranks
    agree on "keep going" through a MAX all-reduce of their own verdict."""
    torch.cuda.synchronize()
    t_warm = time.perf_counter()
    while True:
        more = 1.0 if time.perf_counter() - t_warm < seconds else 0.0
        if parallel.max_over_ranks(more, device=device) < 0.5:
            break
        for _ in range(10):
            step()
        torch.cuda.synchronize()


def flow_op_rooflines(device, B=16, H=384, W=512, iters=20):
    """The three custom operators of the FLOW path at BASELINE configs[3] shapes (SURVEY §8 "Reading of C4"), each
    against the HBM roof: algorithmic bytes (in the kernel's own storage type) / hipEvent time."""
    import ctypes
    from flowtrack.pytorch_amd import _lib
    from flowtrack.pytorch_amd.hip_ops import ActView, Program, new_rowpacked_act
    prog = Program(torch.cuda.Stream(device))
    f16 = _lib.dtype_code(torch.float16)
    h8, w8 = H // 8, W // 8
    c3 = torch.randn((2 * B, h8, w8, 256), device=device).half()
    cin31 = torch.zeros((B, h8, w8, 480), dtype=torch.float16, device=device)
    prog.add("ft_correlation_nhwc_fwd", c3[:B].data_ptr(), c3[B:].data_ptr(), cin31.data_ptr(), B, 256, h8, w8, 20, 2, 256, 480, 32,
             _lib.FT_ACT_LEAKY, ctypes.c_float(0.1), f16, keep=(c3, cin31))
    x6 = new_rowpacked_act(B, H, W, 6, 3, torch.float16, device)
    x6.t.normal_()
    # two flow fields: per-pixel independent noise (sigma 4 px: every lane's taps sit in their own cache lines, the WORST case of
    # a gather; what rounds 2-4 reported) and a SMOOTH field of the kind a flow network emits (a 12 x 16 grid of sigma-6-px
    # vectors, bilinearly upsampled: neighbouring pixels sample neighbouring texels)
    flow = torch.randn((B, 2, H, W), device=device) * 4
    smooth = torch.nn.functional.interpolate(torch.randn((B, 2, 12, 16), device=device) * 6, size=(H, W), mode="bilinear", align_corners=True).contiguous()
    cat1 = new_rowpacked_act(B, H, W, 12, 3, torch.float16, device)
    img = torch.randn((B, 3, H, W), device=device)
    warped = torch.empty_like(img)
    norm = torch.empty((B, 1, H, W), device=device)
    for fl in (flow, smooth):
        prog.add("ft_flow_warp_concat", x6.t.data_ptr(), fl.data_ptr(), ctypes.c_float(20.0), cat1.t.data_ptr(), B, H, W, x6.lpad,
                 x6.wpitch, cat1.lpad, cat1.wpitch, f16, keep=(x6.t, fl, cat1.t))
        prog.add("ft_resample2d_fwd", img.data_ptr(), fl.data_ptr(), warped.data_ptr(), B, 3, H, W, keep=(img, fl, warped))
    prog.add("ft_channelnorm_fwd", img.data_ptr(), norm.data_ptr(), B, 3, H, W, keep=(img, norm))
    torch.cuda.synchronize()
    prog.run_eager()
    prog.stream.synchronize()
    times = prog.time_calls(iters=iters, median=True)
    px = B * H * W
    algo = {
        # fp16 NHWC: two [B,256,48,64] feature maps in, 441 channels out (5.86 MB / pair)
        "ft_correlation_nhwc_fwd": (2 * B * 256 * h8 * w8 * 2 + B * 441 * h8 * w8 * 2, 2.0 * B * 441 * h8 * w8 * 256),
        # x6 (8 fp16 ch) + flow (2 fp32) in, 16 fp16 channels out
        "ft_flow_warp_concat": (px * (16 + 8 + 32), 0.0),
        "ft_resample2d_fwd": (px * 4 * (3 + 2 + 3), 0.0),       # 6.29 MB / pair (fp32 NCHW, the reference's own form)
        "ft_channelnorm_fwd": (px * 4 * (3 + 1), 0.0),          # (reads 3 channels, writes 1)
    }
    out = []
    seen = {}
    for name, ms in times:
        nbytes, flops = algo[name]
        tbs = nbytes / (ms * 1e-3) / 1e12
        seen[name] = seen.get(name, 0) + 1
        gather = name in ("ft_flow_warp_concat", "ft_resample2d_fwd")
        row = {"kernel": name + ("_smooth_flow" if gather and seen[name] == 2 else ""),
               "shape": f"[{B},256,{h8},{w8}] x2 -> 441 ch" if "corr" in name else f"[{B},*,{H},{W}]",
               "bound": "hbm", "us": round(ms * 1e3, 1), "algorithmic_bytes": nbytes, "achieved": round(tbs, 3), "peak": HBM_PEAK_TBS,
               "unit": "TB/s", "frac": round(tbs / HBM_PEAK_TBS, 4)}
        if flops:
            row["tflops"] = round(flops / (ms * 1e-3) / 1e12, 1)
        out.append(row)
    return out


def pose_parity(models_by_mode, x_cpu, seed, H, W, oracle_out=None):
    """The "mAP@OKS vs CPU ref" half of the metric on the benchmarked batch: GPU heat maps / key points of each mode vs
    the CPU oracle (its key points are the annotations).  Also the margin of every arg-max flip in the
    oracle's own heat map: oracle[top-1] - oracle[GPU's pick] (a flip is only possible inside twice the heat-map error)."""
    import numpy as np
    from flowtrack.pytorch_amd.pose import evaluation
    from oracle import keypoints_ref, pose_ref
    m0 = next(iter(models_by_mode.values()))
    sd = {k: v.detach().float().cpu() for k, v in m0.state_dict().items()}
    B = x_cpu.shape[0]
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    want = torch.cat([pose_ref.pose_forward(sd, x_cpu[i:i + 4]) for i in range(0, B, 4)])
    center = np.stack((np.full(B, W / 2.0), np.full(B, H / 2.0)), 1)
    scale = np.full(B, H / 200.0)
    ref_coords, ref_scores, ref_idx, _ = keypoints_ref.final_preds_ref(want.numpy(), center, scale, adjust_coords=True)
    anno = np.concatenate((ref_coords, np.ones_like(ref_scores)), axis=2)
    out = {"crops": B, "annotations": "CPU-oracle key points (final_preds, adjust_coords) of the benchmarked batch",
           "heatmap_range": round(float(want.max() - want.min()), 4)}
    wflat = want.flatten(2).numpy()
    for mode, m in models_by_mode.items():
        hm = m(x_cpu.to(next(m.parameters()).device))
        coords, scores = evaluation.final_preds(hm, center, scale, adjust_coords=True)
        _, _, idx = keypoints_ref.max_preds_ref(hm.float().cpu().numpy())
        pred = np.concatenate((coords, scores), axis=2)
        aps = evaluation.eval_mAP([pred], [anno], [scale * scale], evaluation.COCO_DELTA)
        err = float((hm.float().cpu() - want).abs().max())
        flips = idx != ref_idx
        margin = np.take_along_axis(wflat, ref_idx[..., None], 2)[..., 0] - np.take_along_axis(wflat, idx[..., None], 2)[..., 0]
        out[mode] = {"heatmap_max_abs_err": err, "argmax_identical_frac": round(float(1.0 - flips.mean()), 6),
                     "argmax_flips": int(flips.sum()), "max_flip_margin_in_oracle_heatmap": float(margin[flips].max()) if flips.any() else 0.0,
                     "keypoint_max_abs_err_px": float(np.abs(coords - ref_coords).max()),
                     "mAP_at_OKS": round(float(np.mean(aps)), 4)}
    return out


def trained_like_record(model16, x16, device, steps, world):
    """What the exact-arg-max mode costs on heat maps shaped like a TRAINED net's.  No trained weights exist
    here, and a closed-form fit of the 256 -> 17 heat-map conv on the random trunk's features cannot make such maps (tests/
    peaky_maps.py: the ridge fit peaks at 0.03 of the target, 22 map pixels off — the 8 x 6 bottleneck of a random trunk does not
    carry positions), so the two halves are measured separately:
      * the SCREEN on 1024 x 17 synthetic single-peak maps (synth.peaked_heatmaps: the training target's Gaussian, sigma 2, at
        uniform sub-pixel centres) -> the fraction of crops it flags at the shipped bound and at 1/2, 1/4 of it;
      * the MODE on the benchmarked batch with the bound tuned (bisection) until the screen flags that fraction of its crops
        -> crops/s at that re-run fraction (device-side decision, finished one step late: exact_submit / exact_finish)."""
    import ctypes
    from flowtrack.pytorch_amd import _lib
    from flowtrack.pytorch_amd.hip_ops import check, current_stream_handle
    lib = _lib.load()
    B = x16.shape[0]
    maps = synth.peaked_heatmaps(77, 1024).to(device).contiguous()
    flags = torch.empty(1024, dtype=torch.int32, device=device)
    stats = torch.empty((1024, 4), dtype=torch.float32, device=device)
    frac = {}
    for name, mul in (("shipped_bound", 1.0), ("half_bound", 0.5), ("quarter_bound", 0.25)):
        check(lib.ft_heatmap_argmax_screen(maps.data_ptr(), 1024, 17, 64, 48, ctypes.c_float(model16.exact_argmax_rel_bound * mul),
                                           flags.data_ptr(), stats.data_ptr(), current_stream_handle(device)), "ft_heatmap_argmax_screen")
        frac[name] = round(float(flags.float().mean().item()), 4)
    target = int(round(frac["shipped_bound"] * B))
    keep = model16.exact_argmax_rel_bound
    lo, hi, got = 0.0, keep, None
    try:
        for _ in range(14):                       # the flagged count is monotone in the bound
            mid = 0.5 * (lo + hi)
            model16.exact_argmax_rel_bound = mid
            _, n = model16.forward_keypoint_rows_exact(x16)
            got = (mid, n)
            if n == target:
                break
            if n < target:
                lo = mid
            else:
                hi = mid
        state = {"h": None, "n": 0, "calls": 0}

        def lagged():
            h = model16.exact_submit(x16)
            if state["h"] is not None:
                state["n"] += model16.exact_finish(state["h"])[1]
                state["calls"] += 1
            state["h"] = h

        def drain():
            if state["h"] is not None:
                state["n"] += model16.exact_finish(state["h"])[1]
                state["calls"] += 1
                state["h"] = None
        lagged.drain = drain
        el, rep, tot = measure(lagged, steps, 3, device)
        drain()
    finally:
        model16.exact_argmax_rel_bound = keep
    return {"screen_flagged_frac_1024_crops": frac, "maps": "17 Gaussian peaks per crop, sigma 2 map px, uniform sub-pixel centres, amplitude 0.7-1.0, "
            "noise 1e-3 (synth.peaked_heatmaps; lib/pose/utils/heatmap.py:19-60)",
            "mode_at_that_rerun_frac": {"value": round(B * world * steps / el, 2), "unit": "crops/s", "ms_per_step": round(1e3 * el / steps, 4),
                                        "rerun_frac": round(state["n"] / max(state["calls"] * B, 1), 4), "tuned_rel_bound": got[0], "repeats": rep,
                                        "timed_region_s": round(tot, 4)},
            "ridge_fit_of_the_heatmap_conv": "infeasible on a random trunk: tests/peaky_maps.py (validation peak 0.03 of the target's 1.0, "
                                             "median arg-max 22 map pixels from the key point)"}


def exact_argmax_record(model16, model32, x16, x32, device, steps, world):
    """north_star's "keypoint argmax bit-exact" for the FAST mode: DeconvResnet.forward_keypoint_rows_exact (fp16 pass, margin
    screen on the device, fp32 re-run of the crops whose top-1 / top-2 margin is inside twice the fp16 error bound).  Timed with
    the protocol of the headline on the benchmarked batch; arg-max identity against the fp32 parity mode (itself identical
    to the CPU reference: `parity.fp32`, tests) on that batch and on 1024 further synthetic crops."""
    import numpy as np
    rerun = {"n": 0, "calls": 0}

    def step():
        _, n = model16.forward_keypoint_rows_exact(x16)
        rerun["n"] += n
        rerun["calls"] += 1
    el, rep, tot = measure(step, steps, 3, device)
    B = x16.shape[0]
    # the same path when the screen flags nothing (bound 0: what single-peak heat maps of a trained net cost — fp16 plan + screen
    # launch + one device -> host read of B flags; no trained weights exist here, so identity in that regime rests on the bound,
    # shown on synthetic single-peak maps in tests/test_pose_gpu.py::test_argmax_screen_is_selective)
    keep_bound = model16.exact_argmax_rel_bound
    model16.exact_argmax_rel_bound = 0.0
    try:
        el0, rep0, tot0 = measure(lambda: model16.forward_keypoint_rows_exact(x16), steps, 3, device)
        # the same regime WITHOUT a host read inside the step (round 5: exact_submit / exact_finish, finished one step late)
        state = {"h": None}

        def lagged():
            h = model16.exact_submit(x16)
            if state["h"] is not None:
                model16.exact_finish(state["h"])
            state["h"] = h

        def drain():
            if state["h"] is not None:
                model16.exact_finish(state["h"])
                state["h"] = None
        lagged.drain = drain
        el0d, rep0d, tot0d = measure(lagged, steps, 3, device)
        drain()
        # ... and with the screen / gather / header copy recorded INSIDE the plan's graph (exact_in_plan): one replay + one event per
        # step, the host looks at the header one step late; four plan replicas in rotation as the headline
        model16.exact_in_plan = True
        try:
            eplans = [model16.plan_for(B, x16.shape[2], x16.shape[3], r) for r in range(4)]
            for r, pl in enumerate(eplans):
                pl.x_static.copy_(synth.pose_crops(100 + 1000 * r, B, x16.shape[2], x16.shape[3]))
            st2 = {"i": 0, "prev": None}

            def in_graph():
                pl = eplans[st2["i"]]
                st2["i"] = (st2["i"] + 1) % len(eplans)
                model16.exact_submit_plan(pl)
                if st2["prev"] is not None:
                    model16.exact_finish_plan(st2["prev"])
                st2["prev"] = pl

            def drain2():
                if st2["prev"] is not None:
                    model16.exact_finish_plan(st2["prev"])
                    st2["prev"] = None
            in_graph.drain = drain2
            el0g, rep0g, tot0g = measure(in_graph, steps, 3, device)
            drain2()
        finally:
            model16.exact_in_plan = False
    finally:
        model16.exact_argmax_rel_bound = keep_bound

    def idx_of(rows):
        r = rows.float().cpu().numpy()
        return (np.floor(r[..., 1] + 0.5) * 4096 + np.floor(r[..., 0] + 0.5)).astype(np.int64)
    import ctypes
    from flowtrack.pytorch_amd import _lib
    from flowtrack.pytorch_amd.hip_ops import check, current_stream_handle
    same = total = flagged = 0
    same_b = None
    margins = []
    mbuf = torch.empty(B, dtype=torch.float32, device=device)
    for k in range(17):                                   # k = 0: the benchmarked batch, then 16 x 64 other crops
        crops = synth.pose_crops(100, B, x16.shape[2], x16.shape[3]) if k == 0 else synth.pose_crops(5000 + k, B, x16.shape[2], x16.shape[3])
        x16.copy_(crops)
        x32.copy_(crops)
        rows16, n = model16.forward_keypoint_rows_exact(x16)
        rows32 = model32.forward_keypoint_rows(x32)
        eq = idx_of(rows16) == idx_of(rows32)
        if k == 0:
            same_b = float(eq.mean())
        else:
            same += int(eq.sum()); total += eq.size; flagged += n
            model16.forward_keypoint_rows(x16)             # the fp16 heat maps of these crops once more, for the margin statistics
            hm = model16._last_plan.heatmaps
            check(_lib.load().ft_heatmap_min_margin(hm.data_ptr(), B, hm.shape[1], hm.shape[2], hm.shape[3], mbuf.data_ptr(),
                                                    current_stream_handle(device)), "ft_heatmap_min_margin")
            margins.append(mbuf.cpu().numpy().copy())
    x16.copy_(synth.pose_crops(100, B, x16.shape[2], x16.shape[3]))
    x32.copy_(synth.pose_crops(100, B, x16.shape[2], x16.shape[3]))
    trained = trained_like_record(model16, x16, device, steps, world)
    return {"value": round(B * world * steps / el, 2), "unit": "crops/s", "steps": steps, "repeats": rep, "ms_per_step": round(1e3 * el / steps, 4),
            "trained_like_maps": trained,
            "timed_region_s": round(tot, 4), "rel_bound": model16.exact_argmax_rel_bound,
            "no_rerun_path": {"value": round(B * world * steps / el0, 2), "unit": "crops/s", "ms_per_step": round(1e3 * el0 / steps, 4),
                              "repeats": rep0, "timed_region_s": round(tot0, 4)},
            "no_rerun_path_device_decision": {"value": round(B * world * steps / el0d, 2), "unit": "crops/s", "ms_per_step": round(1e3 * el0d / steps, 4),
                                              "repeats": rep0d, "timed_region_s": round(tot0d, 4)},
            "no_rerun_path_in_graph": {"value": round(B * world * steps / el0g, 2), "unit": "crops/s", "ms_per_step": round(1e3 * el0g / steps, 4),
                                       "repeats": rep0g, "timed_region_s": round(tot0g, 4)},
            "rerun_frac": round(rerun["n"] / max(rerun["calls"] * B, 1), 4),
            "argmax_identical_frac": same_b, "argmax_identical_frac_1024_crops": round(same / max(total, 1), 6),
            "rerun_frac_1024_crops": round(flagged / 1024.0, 4),
            "crop_min_margin_quantiles_1024_crops": {q: float(np.quantile(np.concatenate(margins), float(q))) for q in ("0.1", "0.5", "0.9")},
            "crops_below_margin_1024_crops": {str(t): round(float((np.concatenate(margins) < t).mean()), 4) for t in (1e-3, 2e-3, 4e-3, 8e-3)}}


def clip_record(device, n_frames=300):
    """BASELINE configs[4] on one GPU: the 300-frame 512x384 synthetic clip through tools/tracking/demo.run_clip with the real
    nets (fp16, synthetic weights): flow of the 299 pairs + pose of the detector boxes (batched) + the sequential pass
    (propagate, NMS, pose of the propagated boxes, id assignment).  One warm-up run builds every plan / graph, the second is
    timed end to end on the host clock (the pass is host-paced)."""
    from tools.tracking import demo
    targs = types.SimpleNamespace(pose_backbone=50, pose_model="", flow_net="FlowNet2S", flow_model="", fp16=True)
    pose, flow = demo.build_nets(targs, device)
    frames, dets = demo.synthetic_clip(n_frames)
    demo.run_clip(frames, dets, pose, flow, max_boxes="2x")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res, tm = demo.run_clip(frames, dets, pose, flow, max_boxes="2x")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # throughput form: K independent clips on this GPU, their sequential passes interleaved (demo.run_clips); 8 x 150 frames keeps the
    # leg short, the one-after-the-other run of the same clips is the A/B beside it
    K, nf = 8, min(n_frames, 150)
    clips = [demo.synthetic_clip(nf, seed=c) for c in range(K)]
    demo.run_clips(clips, pose, flow, max_boxes="2x")
    multi = {}
    for mode, key in ((False, "one_after_the_other"), (True, "interleaved")):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        _, tmk = demo.run_clips(clips, pose, flow, max_boxes="2x", interleave=mode)
        torch.cuda.synchronize()
        dk = time.perf_counter() - t1
        multi[key] = {"frames_per_s_total": round(K * nf / dk, 1), "wall_s": round(dk, 4)}
    return {"metric": "FlowTrack clip frames/sec (configs[4] on 1 GPU: R50 pose + FlowNet2S fp16, 300 synthetic frames)",
            "clips": {"K": K, "frames_per_clip": nf, **multi},
            "frames": n_frames, "frame_hw": [int(frames.shape[1]), int(frames.shape[2])], "people": 5,
            "frames_per_s": round(n_frames / dt, 1), "wall_s": round(dt, 4), "flow_s": round(tm["flow_s"], 4),
            "pose_s": round(tm["pose_s"], 4), "pass_s": round(tm["track_s"], 4), "pass_frac": round(tm["track_s"] / dt, 3),
            "boxes_per_frame_avg": round(sum(len(f["boxes"]) for f in res) / n_frames, 2)}


def make_pose_runner(args, device, dtype, rank, world, backbone, H, W, B, force_gather=False, rotate=1):
    """(model, x, step): the pose hot path on batches resident in HBM, arg-max inside the plan's graph.  `rotate` > 1: that many
    DIFFERENT batches are resident (one plan replica each: own input + activation buffers, shared packed weights) and the steps
    go round them, so no step re-reads the bytes the previous one left in the 256-MB Infinity Cache.
    N > 1: the [B,17,3] key-point rows of every rank are all-gathered (13 kB per rank) on a communication stream, one step
    behind the compute stream (parallel.RowGatherer): step t's graph replays while step t-1's rows travel."""
    model = build_pose(device, dtype, backbone=backbone)
    model.keypoints_in_plan = True                          # arg-max + 0.25 px nudge run inside the plan's graph
    plans = [model.plan_for(B, H, W, r) for r in range(max(1, rotate))]
    for r, pl in enumerate(plans):                          # zero-copy binding: each batch is resident in HBM at the address
        pl.x_static.copy_(synth.pose_crops(100 + rank + 1000 * r, B, H, W))   # its plan's graph reads, before the timed region
    x = plans[0].x_static
    kp_host = torch.empty((B, 17, 3), dtype=torch.float32).pin_memory()
    # force_gather (single-GPU tests): run the communication-stream path of the N > 1 step on one GPU
    gather = parallel.RowGatherer(B, (17, 3), torch.float32, device, force_stream=force_gather) if (world > 1 or force_gather) else None
    state = {"pending": None, "i": 0}

    def consume(h):
        kp_host.copy_(gather.finish(h)[rank * B:(rank + 1) * B], non_blocking=True)   # (the consumer: the rank's own rows back on the host)

    def step():
        plan = plans[state["i"] % len(plans)]
        state["i"] = (state["i"] + 1) % len(plans)
        rows = model.replay(plan).kp_rows                    # [B,17,3] (x, y, score) rows written by the plan's last launch
        if gather is None:
            kp_host.copy_(rows, non_blocking=True)
            return rows
        h = gather.start(rows)
        if state["pending"] is not None:
            consume(state["pending"])
        state["pending"] = h
        return rows

    def drain():
        if state["pending"] is not None:
            consume(state["pending"])
            state["pending"] = None
    step.drain = drain
    step.gather = gather
    step.plans = plans
    return model, x, step


def make_flow_runner(args, device, dtype, rank, world, name, B, gather_mode="sampled", force_gather=False, rotate=1):
    """FlowNet on B frame pairs resident in HBM.  What leaves the GPU per step is what the consumer of the flow needs:
    the tracking glue reads the field at the previous frame's key points (lib/tracking/flow_utils.py:21-26), so by default
    the step samples the field at 8 x 17 points per pair and (N > 1) all-gathers those rows (1 kB per pair); `full`
    gathers the whole fields (1.57 MB per pair) instead."""
    model = build_flow(device, dtype, name=name)
    plans = [model.plan_for(B, 384, 512, r) for r in range(max(1, rotate))]   # zero-copy binding, `rotate` resident batches (see the pose runner)
    for r, pl in enumerate(plans):
        pl.x_static.copy_(synth.frame_pairs(100 + rank + 1000 * r, B))
    x = plans[0].x_static
    npts = 8 * 17
    pts = torch.from_numpy(synth.uniform01(7, "flow_sample_points", (npts,))).to(device)
    idx = (pts * (384 * 512 - 1)).long()
    full = gather_mode == "full"
    gather = None
    if (world > 1 or force_gather) and gather_mode != "none":
        gather = parallel.RowGatherer(B, (2, 384, 512) if full else (2, npts), torch.float32, device, force_stream=force_gather)
    samples = torch.empty((B, 2, npts), dtype=torch.float32, device=device)
    state = {"pending": None, "i": 0}

    def step():
        plan = plans[state["i"] % len(plans)]
        state["i"] = (state["i"] + 1) % len(plans)
        flow = model.replay(plan).out
        if not full:
            torch.index_select(flow.flatten(2), 2, idx, out=samples)
        if gather is not None:
            h = gather.start(flow if full else samples)
            if state["pending"] is not None:
                gather.finish(state["pending"])
            state["pending"] = h
        return flow

    def drain():
        if state["pending"] is not None:
            gather.finish(state["pending"])
            state["pending"] = None
    step.drain = drain
    step.gather = gather
    step.plans = plans
    return model, x, step


def measure(step, steps, warmup, device, fixed_warmup=False, min_total_s=1.0, max_repeats=400):
    """W untimed warm-up steps, then BLOCKS of exactly `steps` steps, each bracketed by barrier + synchronize on both sides
    (timed_region).  One block is what the contract asks for; with a small `steps` it is a few tens of milliseconds, one
    preempted step away from a 5 % error, so the block is repeated until >= min_total_s has been timed and the MEDIAN block
    is reported.  The repeat count follows from the first block's (max-over-ranks) time, so every rank runs the same number
    of blocks.  Returns (median block seconds, repeats, total timed seconds)."""
    for _ in range(max(warmup, 2)):   # >= 2: first run is eager (+ tile benchmark) + graph capture, second replays the graph
        step()
    if fixed_warmup:                  # profiling runs that count launches: exactly one block
        t = timed_region(step, steps, device)
        return t, 1, t
    spin_up(step, device)
    blocks = [timed_region(step, steps, device)]
    repeats = min(max_repeats, max(1, int(min_total_s / max(blocks[0], 1e-6)) + 1))
    for _ in range(repeats - 1):
        blocks.append(timed_region(step, steps, device))
    blocks.sort()
    return blocks[len(blocks) // 2], len(blocks), sum(blocks)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=800, help="timed steps (default: ~1.1 s of the 64-crop pose step)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", choices=["pose", "flow"], default="pose")
    ap.add_argument("--dtype", choices=["fp16", "fp32"], default="fp16")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default 64 crops / 16 pairs)")
    ap.add_argument("--backbone", choices=["resnet50", "resnet101", "resnet152"], default="resnet50",
                    help="pose trunk; resnet101 + --res 384x288 + --batch 16 is BASELINE.json configs[2]")
    ap.add_argument("--res", default="256x192", help="pose crop HxW (multiples of 32)")
    ap.add_argument("--flow-model", default="FlowNet2S",
                    choices=["FlowNet2S", "FlowNet2C", "FlowNet2CS", "FlowNet2CSS", "FlowNet2SD", "FlowNet2"])
    ap.add_argument("--gather", choices=["sampled", "full", "none"], default="sampled",
                    help="flow, N > 1: all-gather the field sampled at 8 x 17 key points per pair (what the tracker reads), the full fields, or nothing")
    ap.add_argument("--fixed-warmup", action="store_true", help="exactly max(W,2) untimed steps (profiling runs that count launches)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline only: no flow / fp32_parity_mode / parity sub-records")
    ap.add_argument("--layers", action="store_true", help="also print the per-layer table to stderr")
    ap.add_argument("--rotate", type=int, default=4, help="resident batches the timed steps go round (1 = replay one batch)")
    ap.add_argument("--config", choices=["c2", "c3", "c5"], default="c2",
                    help="c2 (default) = BASELINE configs[1] per GPU, weak; c3 = configs[2]: ResNet-101 384x288, 128 crops sharded over the "
                         "N GPUs (128 / N per GPU: STRONG scaling); c5 = configs[4] by clip: every rank tracks its own clips")
    args = ap.parse_args()
    if args.config == "c3":
        args.backbone, args.res = "resnet101", "384x288"

    rank, local_rank, world = parallel.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if os.environ.get("FT_DIST_BACKEND") == "gloo":           # dry run of the N > 1 code on fewer GPUs than ranks (parallel.init_from_env)
        local_rank %= torch.cuda.device_count()
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    dtype = torch.float16 if args.dtype == "fp16" else torch.float32

    if args.config == "c5":
        return main_clips(args, device, rank, world)
    scaling = "weak"
    if args.workload == "pose":
        B = args.batch or 64
        if args.config == "c3" and not args.batch:
            if 128 % world:
                raise SystemExit("--config c3 shards 128 crops: --gpus must divide 128")
            B, scaling = 128 // world, "strong"
        H, W = (int(v) for v in args.res.lower().split("x"))
        depth = args.backbone[len("resnet"):]
        default_cfg = args.backbone == "resnet50" and (H, W) == (256, 192)
        model, x, step = make_pose_runner(args, device, dtype, rank, world, args.backbone, H, W, B, rotate=args.rotate)
        unit, metric = "crops/s", f"pose crops/sec (ResNet-{depth} + 3-deconv head, {H}x{W})"
        cfg = "configs[1]" if default_cfg else ("configs[2]" if (args.backbone, H, W) == ("resnet101", 384, 288) else "variant")
        workload = f"ResNet-{depth} pose head {args.dtype}, batch {B} x {H}x{W} synthetic crops per GPU (BASELINE.json {cfg})"
    else:
        B = args.batch or 16
        default_cfg = args.flow_model == "FlowNet2S"
        model, x, step = make_flow_runner(args, device, dtype, rank, world, args.flow_model, B, args.gather, rotate=args.rotate)
        unit, metric = "pairs/s", f"flow frame-pairs/sec ({args.flow_model}, 512x384)"
        workload = (f"{args.flow_model} {args.dtype}, batch {B} x 512x384 synthetic frame pairs per GPU "
                    f"(BASELINE.json configs[3]{'' if default_cfg else ' shape, other stack'})")

    rccl = parallel.verify_gather(step.gather, rank) if (world > 1 and step.gather is not None) else None
    elapsed, repeats, total_s = measure(step, args.steps, args.warmup, device, args.fixed_warmup)
    value = B * world * args.steps / elapsed
    out = {
        "metric": metric, "value": round(value, 2), "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "repeats": repeats, "timed_region_s": round(total_s, 4),
        "timing": f"median of {repeats} blocks of exactly {args.steps} steps, barrier + synchronize around each, max over ranks",
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "hip_force_dev_kernarg": __import__("flowtrack.pytorch_amd", fromlist=["KERNARG_STATE"]).KERNARG_STATE,
        "dtype": args.dtype, "data": f"synthetic crops / frame pairs and random weights; {max(1, args.rotate)} different HBM-resident "
                                     "batches in rotation",
        "config": {"workload": workload, "per_gpu_batch": B, "resident_batches": max(1, args.rotate),
                   "parallelism": f"dp{world}: batch sharded, one process per GPU" +
                                  (", RCCL all-gather of the output rows on a comm stream, one step behind" if world > 1 else "")},
    }
    if rccl is not None:
        out["rccl"] = rccl
    if args.rotate > 1 and not args.fixed_warmup:
        # the same step replaying ONE resident batch (what rounds 1-4 reported): the A/B for the cache effect of rotation
        keep = list(step.plans)
        step.plans[:] = keep[:1]
        el1, rep1, _ = measure(step, args.steps, 2, device)
        step.plans[:] = keep
        out["same_batch_replay"] = {"value": round(B * world * args.steps / el1, 2), "ms_per_step": round(1e3 * el1 / args.steps, 4), "repeats": rep1}
    if rank == 0:
        plan = next(iter(model._plans.values()))
        out["gflop_per_unit"] = round(plan.prog.flops / B / 1e9, 3)
        out["achieved_tflops_end_to_end"] = round(value / world * plan.prog.flops / B / 1e12, 2)
        if not args.no_roofline:
            roof, per_layer = conv_roofline(plan.prog, args.dtype)
            if args.dtype == "fp16" and not args.batch and default_cfg:   # the committed PMC run is this exact default workload
                roof["traffic"], roof["traffic_source"] = pmc_traffic(args.workload)
                roof["traffic_unit"] = "HBM bytes per conv launch (avg), rocprofv3 PMC, profiles/"
            out["roofline"] = roof
            if args.layers:
                for label, fl, ms in per_layer:
                    print(f"  {label:28s} {fl / 1e9:9.2f} GFLOP {ms * 1e3:9.1f} us {fl / (ms * 1e-3) / 1e12 if ms > 0 else 0:8.1f} TF/s",
                          file=sys.stderr)
        if not args.no_cpu_baseline and world == 1 and default_cfg:
            out["cpu_baseline"] = cpu_baseline_pose() if args.workload == "pose" else cpu_baseline_flow()

    # ---- the rest of BASELINE.json's metric, in the same line (default pose run only) -----------------------------------
    extras = (not args.no_extras and args.workload == "pose" and args.dtype == "fp16" and default_cfg and not args.batch
              and not args.fixed_warmup)
    if extras:
        # (1) FlowNet2S fp16 on configs[3]'s pairs, every rank, same protocol; sized for >= ~1 s like the headline
        fsteps = max(20, min(args.steps, 1200))
        fmodel, fx, fstep = make_flow_runner(args, device, torch.float16, rank, world, "FlowNet2S", 16, args.gather, rotate=args.rotate)
        frccl = parallel.verify_gather(fstep.gather, rank) if (world > 1 and fstep.gather is not None) else None
        fel, frep, ftot = measure(fstep, fsteps, args.warmup, device)
        if rank == 0:
            fplan = next(iter(fmodel._plans.values()))
            fval = 16 * world * fsteps / fel
            rec = {"metric": "flow frame-pairs/sec (FlowNet2S, 512x384)", "value": round(fval, 2), "unit": "pairs/s", "steps": fsteps,
                   "ms_per_step": round(1e3 * fel / fsteps, 4), "repeats": frep, "timed_region_s": round(ftot, 4), "dtype": "fp16",
                   "config": {"workload": "FlowNet2S fp16, 16 x 512x384 pairs per GPU (configs[3])", "per_gpu_batch": 16,
                              "resident_batches": max(1, args.rotate)},
                   "gflop_per_unit": round(fplan.prog.flops / 16 / 1e9, 3)}
            if frccl is not None:
                rec["rccl"] = frccl
            if not args.no_roofline:
                froof, _ = conv_roofline(fplan.prog, "fp16")
                froof["traffic"], froof["traffic_source"] = pmc_traffic("flow")
                froof["traffic_unit"] = "HBM bytes per conv launch (avg), rocprofv3 PMC, profiles/"
                rec["roofline"] = froof
                rec["roofline_ops"] = flow_op_rooflines(device)
            if not args.no_cpu_baseline and world == 1:
                rec["cpu_baseline"] = cpu_baseline_flow(seconds=8.0)
            out["flow"] = rec
        del fmodel, fx, fstep
        # (2) the pose step in fp32 parity mode (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains, peak 157.3 TFLOP/s)
        psteps = max(10, min(args.steps // 10, 80))
        pmodel, px, pstep = make_pose_runner(args, device, torch.float32, rank, world, "resnet50", 256, 192, 64)
        pel, prep, ptot = measure(pstep, psteps, 3, device)
        if rank == 0:
            pplan = next(iter(pmodel._plans.values()))
            pval = 64 * world * psteps / pel
            rec = {"value": round(pval, 2), "unit": "crops/s", "steps": psteps, "ms_per_step": round(1e3 * pel / psteps, 4),
                   "repeats": prep, "timed_region_s": round(ptot, 4), "dtype": "fp32",
                   }
            if not args.no_roofline:
                proof, _ = conv_roofline(pplan.prog, "fp32")
                rec["roofline"] = {k: proof[k] for k in ("bound", "achieved", "peak", "unit", "frac", "launches_per_step", "avg_launch_us",
                                                         "conv_ms_per_step", "graph_replay_ms_per_step")}
            out["fp32_parity_mode"] = rec
            # (3) parity of both modes on the benchmarked batch vs the CPU oracle (checker role only, outside timed regions)
            if world == 1:
                out["parity"] = pose_parity({"fp16": model, "fp32": pmodel}, synth.pose_crops(100 + rank, 64, 256, 192), 1234, 256, 192)
        # (3b) BASELINE configs[2]'s PER-GPU shape: ResNet-101, 16 x 384x288 crops per GPU (128 crops over 8 GPUs) — what each rank
        # of the 8-GPU run executes; same protocol
        csteps = max(10, min(args.steps // 2, 400))
        cmodel, cx, cstep = make_pose_runner(args, device, torch.float16, rank, world, "resnet101", 384, 288, 16, rotate=args.rotate)
        cel, crep, ctot = measure(cstep, csteps, 3, device)
        if rank == 0:
            cplan = next(iter(cmodel._plans.values()))
            rec = {"metric": "pose crops/sec (ResNet-101 + 3-deconv head, 384x288)", "value": round(16 * world * csteps / cel, 2), "unit": "crops/s",
                   "steps": csteps, "repeats": crep, "ms_per_step": round(1e3 * cel / csteps, 4), "timed_region_s": round(ctot, 4), "dtype": "fp16",
                   "config": {"workload": "ResNet-101 fp16, 16 x 384x288 crops per GPU (configs[2]'s per-GPU shape: 128 / 8)", "per_gpu_batch": 16,
                              "resident_batches": max(1, args.rotate)},
                   "gflop_per_unit": round(cplan.prog.flops / 16 / 1e9, 3)}
            if not args.no_roofline:
                croof, _ = conv_roofline(cplan.prog, "fp16")
                rec["roofline"] = {k: croof[k] for k in ("bound", "achieved", "peak", "unit", "frac", "launches_per_step", "avg_launch_us",
                                                         "conv_ms_per_step", "graph_replay_ms_per_step")}
            out["c3_per_gpu"] = rec
        del cmodel, cx, cstep
        # (4) the fast mode with the parity mode's arg-max (every rank steps: the exact path holds no collective)
        esteps = max(10, min(args.steps // 4, 200))
        erec = exact_argmax_record(model, pmodel, x, px, device, esteps, world)
        if rank == 0:
            out["fp16_exact_argmax"] = erec
    if extras and world == 1 and rank == 0:
        out["clip"] = clip_record(device)
    if rank == 0:
        out["summary"] = summary_of(out)                    # LAST key: the sub-records' numbers in a few hundred characters
        print(json.dumps(out, separators=(",", ":")), flush=True)
    parallel.barrier()


def summary_of(out):
    """The numbers of the line a reader wants first, flat and short (the driver keeps only the tail of long lines)."""
    def g(d, *ks):
        for k in ks:
            d = d.get(k) if isinstance(d, dict) else None
        return d
    s = {"pose_crops_s": out.get("value"), "pose_frac": g(out, "roofline", "frac"), "pose_same_batch_crops_s": g(out, "same_batch_replay", "value"),
         "flow_pairs_s": g(out, "flow", "value"), "flow_frac": g(out, "flow", "roofline", "frac"),
         "c3_crops_s": g(out, "c3_per_gpu", "value"), "c3_frac": g(out, "c3_per_gpu", "roofline", "frac"),
         "fp32_crops_s": g(out, "fp32_parity_mode", "value"), "fp32_frac": g(out, "fp32_parity_mode", "roofline", "frac"),
         "fp16_argmax_identical": g(out, "parity", "fp16", "argmax_identical_frac"), "fp16_heatmap_err": g(out, "parity", "fp16", "heatmap_max_abs_err"),
         "fp16_mAP_OKS": g(out, "parity", "fp16", "mAP_at_OKS"), "fp32_argmax_identical": g(out, "parity", "fp32", "argmax_identical_frac"),
         "exact_argmax_crops_s": g(out, "fp16_exact_argmax", "value"), "exact_no_rerun_crops_s": g(out, "fp16_exact_argmax", "no_rerun_path", "value"),
         "exact_no_rerun_device_decision_crops_s": g(out, "fp16_exact_argmax", "no_rerun_path_device_decision", "value"),
         "exact_no_rerun_in_graph_crops_s": g(out, "fp16_exact_argmax", "no_rerun_path_in_graph", "value"),
         "clip_frames_s": g(out, "clip", "frames_per_s"), "clips8_frames_s": g(out, "clip", "clips", "interleaved", "frames_per_s_total"),
         "cpu_crops_s": g(out, "cpu_baseline", "value"), "cpu_pairs_s": g(out, "flow", "cpu_baseline", "value"),
         "rccl_ranks_verified": g(out, "rccl", "ranks_verified")}
    for row in g(out, "flow", "roofline_ops") or []:
        s[row["kernel"].replace("ft_", "").replace("_fwd", "") + "_frac"] = row["frac"]
    return {k: (round(v, 4) if isinstance(v, float) else v) for k, v in s.items() if v is not None}


def main_clips(args, device, rank, world):
    """--config c5 = BASELINE configs[4] scaled out BY CLIP: every rank tracks its own K synthetic clips through
    tools/tracking/demo.run_clips (no exchange between ranks at all: a clip's sequential pass cannot be sharded, clips can);
    value = all ranks' frames / the slowest rank's wall time."""
    from tools.tracking import demo
    targs = types.SimpleNamespace(pose_backbone=50, pose_model="", flow_net="FlowNet2S", flow_model="", fp16=True)
    pose, flow = demo.build_nets(targs, device)
    K, nf = 8, 150
    clips = [demo.synthetic_clip(nf, seed=rank * K + c) for c in range(K)]
    demo.run_clips(clips, pose, flow, max_boxes="2x")       # builds every plan / graph
    steps = max(1, min(args.steps, 3))
    parallel.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        demo.run_clips(clips, pose, flow, max_boxes="2x")
    torch.cuda.synchronize(); parallel.barrier()
    dt = parallel.max_over_ranks(time.perf_counter() - t0, device=device)
    if rank == 0:
        print(json.dumps({"metric": "FlowTrack clip frames/sec (configs[4] by clip: R50 pose + FlowNet2S fp16, 8 clips x 150 frames per GPU)",
                          "value": round(world * K * nf * steps / dt, 1), "unit": "frames/s", "n_gpus": world, "steps": steps, "warmup": 1,
                          "ms_per_step": round(1e3 * dt / steps, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "fp16", "data": "synthetic clips and weights", "config": {"workload": "BASELINE.json configs[4], one process per GPU x 8 clips",
                                                                                               "clips_per_gpu": K, "frames_per_clip": nf}},
                         separators=(",", ":")), flush=True)
    parallel.barrier()


if __name__ == "__main__":
    main()
