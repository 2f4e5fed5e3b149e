"""GPU-side view of tools/tracking/demo.run_clips: run under `rocprofv3 --kernel-trace --output-format csv -d <dir> -- python
tools/dev/clips_gpu_busy.py run K T G`, then `python tools/dev/clips_gpu_busy.py report <dir>`: the timed run is the last burst
of kernels (0.5 s of silence before it); prints its span, the union of kernel intervals (GPU busy), the sum of durations
(> union when kernels of different streams overlap) and the top kernels."""
import csv, glob, os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if sys.argv[1] == "run":
    import torch
    from tools.tracking import demo
    K, T, G = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    args = types.SimpleNamespace(pose_backbone=50, pose_model="", flow_net="FlowNet2S", flow_model="", fp16=True)
    dev = torch.device("cuda", 0)
    pose, flow = demo.build_nets(args, dev)
    clips = [demo.synthetic_clip(T, seed=c) for c in range(K)]
    demo.run_clips(clips, pose, flow, max_boxes="2x", groups=G)
    demo.run_clips(clips, pose, flow, max_boxes="2x", groups=G)
    torch.cuda.synchronize(); time.sleep(0.6)
    t0 = time.perf_counter()
    demo.run_clips(clips, pose, flow, max_boxes="2x", groups=G)
    torch.cuda.synchronize()
    print(f"timed run: {K} x {T} frames, groups {G}: {time.perf_counter() - t0:.4f} s wall")
    time.sleep(0.6)
else:
    rows = []
    for path in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # last burst: walk back from the end until a gap of >= 0.4 s
    i = len(rows) - 1
    while i > 0 and rows[i][0] - rows[i - 1][1] < 400_000_000:
        i -= 1
    burst = rows[i:]
    span = burst[-1][1] - burst[0][0]
    busy, cur_s, cur_e = 0, burst[0][0], burst[0][1]
    for s, e, _ in burst[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    total = sum(e - s for s, e, _ in burst)
    print(f"kernels {len(burst)}  span {span / 1e6:.1f} ms  GPU busy (union) {busy / 1e6:.1f} ms = {busy / span:.2f}  sum of durations {total / 1e6:.1f} ms")
    agg = {}
    for s, e, n in burst:
        k = n.split("(")[0][:70]
        a = agg.setdefault(k, [0, 0])
        a[0] += 1; a[1] += e - s
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"  {t / 1e6:8.2f} ms  {c:6d} x {t / c / 1e3:7.1f} us  {k}")
