#!/bin/bash
# usage: sweep_tiles.sh <pose|flow> -- conv_bench under every forced tile / split-K combination (dev: autotune potential)
w=${1:-pose}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/sweep_$w
for cfg in "0 0 0" "256 128 1" "128 128 1" "128 128 2" "128 64 1" "64 128 1" "64 128 2" "64 128 4" "64 64 1" "64 64 2" "64 64 4"; do
  set -- $cfg
  FT_CONV_BP=$1 FT_CONV_BC=$2 FT_CONV_KS=$3 ITERS=10 timeout 120 python $R/tools/dev/conv_bench.py $w fp16 2>/dev/null | grep -v "^/" > $R/gpurun_out/sweep_$w/bp$1_bc$2_ks$3.txt
done
python - <<PY
import glob, os, re, collections
rows = collections.OrderedDict()
cfgs = []
for f in sorted(glob.glob("$R/gpurun_out/sweep_$w/*.txt")):
    c = os.path.basename(f)[:-4]; cfgs.append(c)
    for l in open(f):
        m = re.match(r"(\S+)\s+x(\d+)\s+([\d.]+) us", l)
        if m: rows.setdefault(m.group(1), {"n": int(m.group(2))})[c] = float(m.group(3))
print("%-18s" % "layer", " ".join("%14s" % c for c in cfgs), "  best")
th = tb = 0
for k, r in rows.items():
    best = min((v, c) for c, v in r.items() if c != "n")
    h = r.get("bp0_bc0_ks0", 0)
    th += h * r["n"]; tb += best[0] * r["n"]
    print("%-18s" % k, " ".join("%14.1f" % r.get(c, -1) for c in cfgs), "  %s %.1f" % (best[1], best[0]))
print("heuristic total %.1f us, best-per-layer total %.1f us" % (th, tb))
PY
