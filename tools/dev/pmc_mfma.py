"""Aggregate a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass: MFMA-busy fraction per kernel.
usage: pmc_mfma.py <dir> <out.json>
busy fraction = SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs) / (GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 * 1024)."""
import csv, glob, json, os, sys, collections
root, out = sys.argv[1], sys.argv[2]
busy, act, calls = collections.Counter(), collections.Counter(), collections.Counter()
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if row["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
            busy[k] += float(row["Counter_Value"]); calls[k] += 1
        elif row["Counter_Name"] == "GRBM_GUI_ACTIVE":
            act[k] += float(row["Counter_Value"])
res = {}
tb = ta = 0.0
for k in busy:
    if act[k] <= 0: continue
    short = k.replace("(anonymous namespace)::", "").split("(")[0][:110]
    frac = busy[k] / (act[k] / 8.0 * 1024.0)
    res[short] = {"launches": calls[k], "gpu_cycles_total": act[k] / 8.0, "mfma_busy_frac": round(frac, 4)}
    if any(t in k for t in ("conv_", "conv3x3_direct", "conv1x1_stream", "bottleneck_fused", "bottleneck_stream")) and "ft" in k:
        tb += busy[k]; ta += act[k]
top = dict(sorted(res.items(), key=lambda kv: -kv[1]["gpu_cycles_total"])[:12])
summary = {"conv_kernels_mfma_busy_frac": round(tb / (ta / 8.0 * 1024.0), 4) if ta else None, "top_kernels_by_gpu_time": top}
json.dump(summary, open(out, "w"), indent=1)
print(json.dumps({"conv_kernels_mfma_busy_frac": summary["conv_kernels_mfma_busy_frac"]}))
for k, v in list(top.items())[:6]:
    print(f"  {v['mfma_busy_frac']:.3f}  {k[:100]}")
