"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE CSVs into HBM bytes per bench step for the conv kernels.
usage: pmc_traffic.py <dir with fetch/ and write/ subdirs> <steps+warmup+1 eager> <out.json>
gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads ->
read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is in KiB (uncalibrated, taken at face value)."""
import csv, glob, json, os, sys, collections
root, nforward, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
def load(sub, counter):
    tot = collections.Counter(); calls = collections.Counter()
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter: continue
            k = row["Kernel_Name"]
            tot[k] += float(row["Counter_Value"]); calls[k] += 1
    return tot, calls
fetch, fc = load("fetch", "FETCH_SIZE")
write, wc = load("write", "WRITE_SIZE")
res = {"forwards_profiled": nforward, "kernels": {}}
conv_r = conv_w = 0.0; n_conv = 0
for k in sorted(set(fetch) | set(write)):
    r = 2.0 * fetch.get(k, 0.0) * 1024.0 / nforward
    w = write.get(k, 0.0) * 1024.0 / nforward
    short = k.replace("(anonymous namespace)::", "").split("(")[0][:100]
    res["kernels"][short] = {"launches_per_forward": fc.get(k, wc.get(k, 0)) / nforward, "read_MB_per_forward": round(r / 1e6, 2), "write_MB_per_forward": round(w / 1e6, 2)}
    if any(t in k for t in ("conv_igemm", "conv_fewout", "conv_halo", "conv_stem", "conv_pflow", "conv_splitk", "bottleneck_fused", "bottleneck_stream", "conv_direct", "conv3x3_direct", "conv1x1_stream")):
        conv_r += r; conv_w += w
        if "conv_splitk" not in k:      # the reduce launch belongs to the ft_conv2d_fwd call of its split-K conv
            n_conv += fc.get(k, 0) / nforward
res["conv_kernels"] = {"launches_per_forward": n_conv, "hbm_read_MB_per_forward": round(conv_r / 1e6, 1), "hbm_write_MB_per_forward": round(conv_w / 1e6, 1),
                       "hbm_bytes_per_launch_avg": (conv_r + conv_w) / max(n_conv, 1)}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res["conv_kernels"]))
