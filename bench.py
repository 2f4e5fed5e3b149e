#!/usr/bin/env python
"""Benchmark of the FlowTrack hot path on MI355X (contract: see the task brief / DESIGN.md §Measurement).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the pose hot path over one batch that is already resident in HBM:
NCHW fp32 crops -> NHWC pack -> ResNet-50 + 3x deconv + heatmap conv (HIP graph) -> per-map arg-max /
sub-pixel nudge (ft_heatmap_max_preds) [-> RCCL all-gather of the keypoint rows when N > 1].
Workload = BASELINE.json configs[1]: ResNet-50 pose head, fp16, batch 64 x 256x192 per GPU (weak scaling).
`--workload flow` times FlowNet2S on configs[3]'s 16 x 512x384 frame pairs instead (pairs/s).
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
    path = os.path.join(ROOT, "profiles", f"r01_{workload}_hbm_traffic_pmc.json")
    try:
        with open(path) as f:
            return json.load(f)["conv_kernels"]["hbm_bytes_per_launch_avg"]
    except (OSError, KeyError, ValueError):
        return None


def conv_roofline(prog, dtype_name, iters=5):
    """hipEvent timing of the plan's launches on its own stream.  `achieved` uses events at the boundaries of each run
    of consecutive conv launches (kernels back to back as in the graph); the per-launch pass (an event after every
    launch, ~1 us of overhead each) only feeds the --layers table and the cross-check field."""
    times = prog.time_calls(iters=iters)
    per_launch_conv_ms = sum(ms for name, ms in times if is_conv_call(name))
    conv_ms, other_ms = prog.time_conv_runs(iters=iters)
    total_ms = conv_ms + other_ms
    n_conv = sum(1 for name, _ in times if is_conv_call(name))
    flops = prog.flops
    achieved = flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    peak = MFMA_PEAK_TFLOPS[dtype_name]
    per_layer = []
    for label, call_idx, fl in sorted([r[:3] for r in prog.conv_records] + list(prog.fused_records), key=lambda r: r[1]):
        ms = times[call_idx][1]
        per_layer.append((label, fl, ms))
    return {
        "bound": "mfma", "kernel": "every conv launch of the step (ft_conv2d_fwd / ft_bottleneck_fwd): conv_igemm_dma_kernel + its LDS-patch (halo / stem / pflow), fused-bottleneck, generic and few-output variants", "achieved": round(achieved, 2), "peak": peak,
        "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": None,
        "launches_per_step": n_conv, "flop_per_launch_avg": flops / max(n_conv, 1),
        "avg_launch_us": round(conv_ms * 1e3 / max(n_conv, 1), 2),
        "conv_ms_per_step": round(conv_ms, 4), "all_kernels_ms_per_step_eager_events": round(total_ms, 4),
        "conv_ms_per_step_event_per_launch": round(per_launch_conv_ms, 4),
    }, per_layer


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
    return {"value": round(4 * n / el, 2), "unit": "crops/s", "cores": cores, "kind": "port",
            "sample": f"{n} forwards of batch 4 x 3x256x192 fp32 (BASELINE configs[0]) in {el:.1f} s, torch CPU {torch.__version__}, "
                      f"{cores} threads (fastest of the tried counts; {avail} logical CPUs visible)"}


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
    return {"value": round(n / el, 2), "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"{n} FlowNet2S forwards of 1 x 3x2x384x512 fp32 in {el:.1f} s, torch CPU, {cores} threads "
                      f"(fastest of the tried counts; {avail} logical CPUs visible)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=["pose", "flow"], default="pose")
    ap.add_argument("--dtype", choices=["fp16", "fp32"], default="fp16")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default 64 crops / 16 pairs)")
    ap.add_argument("--backbone", choices=["resnet50", "resnet101", "resnet152"], default="resnet50",
                    help="pose trunk; resnet101 + --res 384x288 + --batch 16 is BASELINE.json configs[2]")
    ap.add_argument("--res", default="256x192", help="pose crop HxW (multiples of 32)")
    ap.add_argument("--flow-model", default="FlowNet2S",
                    choices=["FlowNet2S", "FlowNet2C", "FlowNet2CS", "FlowNet2CSS", "FlowNet2SD", "FlowNet2"])
    ap.add_argument("--fixed-warmup", action="store_true", help="exactly max(W,2) untimed steps (profiling runs that count launches)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--layers", action="store_true", help="also print the per-layer table to stderr")
    args = ap.parse_args()

    rank, local_rank, world = parallel.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    dtype = torch.float16 if args.dtype == "fp16" else torch.float32

    from flowtrack.pytorch_amd.hip_ops import heatmap_max_preds

    if args.workload == "pose":
        B = args.batch or 64
        H, W = (int(v) for v in args.res.lower().split("x"))
        depth = args.backbone[len("resnet"):]
        default_cfg = args.backbone == "resnet50" and (H, W) == (256, 192)
        model = build_pose(device, dtype, backbone=args.backbone)
        model.keypoints_in_plan = True                          # arg-max + 0.25 px nudge run inside the plan's graph
        x = model.static_input(B, H, W)                         # zero-copy binding: the batch is resident in HBM at the
        x.copy_(synth.pose_crops(100 + rank, B, H, W))          # address the plan's graph reads, before the timed region
        unit, metric = "crops/s", f"pose crops/sec (ResNet-{depth} + 3-deconv head, {H}x{W})"
        kp_host = torch.empty((B, 17, 3), dtype=torch.float32).pin_memory()

        def step():
            _, _, score, coords = model.forward_keypoints(x)
            rows = torch.cat((coords, score), dim=2)            # [B,17,3] keypoint rows
            rows = parallel.all_gather_rows(rows, B * world) if world > 1 else rows
            kp_host.copy_(rows[rank * B:(rank + 1) * B] if world > 1 else rows, non_blocking=True)
            return rows
        cfg = "configs[1]" if default_cfg else ("configs[2]" if (args.backbone, H, W) == ("resnet101", 384, 288) else "variant")
        workload = f"ResNet-{depth} pose head {args.dtype}, batch {B} x {H}x{W} synthetic crops per GPU (BASELINE.json {cfg})"
    else:
        B = args.batch or 16
        default_cfg = args.flow_model == "FlowNet2S"
        model = build_flow(device, dtype, name=args.flow_model)
        x = model.static_input(B, 384, 512)                     # zero-copy binding (see the pose branch)
        x.copy_(synth.frame_pairs(100 + rank, B))
        unit, metric = "pairs/s", f"flow frame-pairs/sec ({args.flow_model}, 512x384)"

        def step():
            flow = model(x, copy_output=False)
            return parallel.all_gather_rows(flow, B * world) if world > 1 else flow
        workload = (f"{args.flow_model} {args.dtype}, batch {B} x 512x384 synthetic frame pairs per GPU "
                    f"(BASELINE.json configs[3]{'' if default_cfg else ' shape, other stack'})")

    for _ in range(max(args.warmup, 2)):   # >= 2: first run is eager (+ tile benchmark) + graph capture, second replays the graph
        step()
    # untimed: keep replaying until the GPU has been busy for ~0.5 s, so the timed region does not start on the
    # idle clocks of a freshly woken GPU (observed: an occasional 2x slower 30-step region right after start-up)
    torch.cuda.synchronize()
    t_warm = time.perf_counter()
    while not args.fixed_warmup:
        # every rank must run the same number of steps (step() holds a collective when world > 1): the ranks agree
        # on "keep going" through a MAX all-reduce of their own elapsed-time verdict
        more = 1.0 if time.perf_counter() - t_warm < 0.5 else 0.0
        if parallel.max_over_ranks(more, device=device) < 0.5:
            break
        for _ in range(10):
            step()
        torch.cuda.synchronize()
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    parallel.barrier()
    torch.cuda.synchronize()
    elapsed = parallel.max_over_ranks(time.perf_counter() - t0, device=device)

    value = B * world * args.steps / elapsed
    out = {
        "metric": metric, "value": round(value, 2), "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic (counter-hash crops ~N(0,1) / translated-texture frame pairs, He-scaled random weights)",
        "config": {"workload": workload, "per_gpu_batch": B,
                   "parallelism": f"dp{world}: batch sharded, one process per GPU" + (", RCCL all-gather of outputs" if world > 1 else "")},
    }
    if rank == 0:
        plan = next(iter(model._plans.values()))
        out["gflop_per_unit"] = round(plan.prog.flops / B / 1e9, 3)
        out["achieved_tflops_end_to_end"] = round(value / world * plan.prog.flops / B / 1e12, 2)
        if not args.no_roofline:
            roof, per_layer = conv_roofline(plan.prog, args.dtype)
            if args.dtype == "fp16" and not args.batch and default_cfg:   # the committed PMC run is this exact default workload
                roof["traffic"] = pmc_traffic(args.workload)
                roof["traffic_unit"] = "HBM bytes per conv launch (avg), rocprofv3 PMC, profiles/"
            out["roofline"] = roof
            if args.layers:
                for label, fl, ms in per_layer:
                    print(f"  {label:28s} {fl / 1e9:9.2f} GFLOP {ms * 1e3:9.1f} us {fl / (ms * 1e-3) / 1e12 if ms > 0 else 0:8.1f} TF/s",
                          file=sys.stderr)
        if not args.no_cpu_baseline and world == 1 and default_cfg:
            out["cpu_baseline"] = cpu_baseline_pose() if args.workload == "pose" else cpu_baseline_flow()
        print(json.dumps(out), flush=True)
    parallel.barrier()


if __name__ == "__main__":
    main()
