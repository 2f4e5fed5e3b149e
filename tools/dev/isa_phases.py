"""Static instruction mix of one kernel of a hipcc -S listing, split at its s_memtime stamps (the kernels' phase markers).
usage: isa_phases.py <file.s> <substring of the kernel's mangled name> [--dump N]   (dev tool; no GPU)"""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i] or lines[i].strip().startswith(".Lfunc_end"))


def klass(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_accvgpr"): return "accvgpr"
    if op.startswith("ds_"): return "ds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_"): return "salu"
    return "other"


seg, segs = collections.Counter(), []
for l in lines[start:end]:
    t = l.split(";")[0].strip()
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        continue
    op = t.split()[0]
    if op == "s_memtime":
        segs.append(seg); seg = collections.Counter()
        continue
    seg[klass(op)] += 1
    seg["op:" + op] += 1
segs.append(seg)
cols = ["mfma", "valu", "accvgpr", "ds", "vmem", "waitcnt", "barrier", "salu", "nop"]
print(f"{'segment':>8s} " + " ".join(f"{c:>8s}" for c in cols) + "   non-mfma/mfma")
tot = collections.Counter()
for i, s in enumerate(segs):
    tot.update(s)
    n = sum(s[c] for c in cols if c != "mfma")
    print(f"{i:8d} " + " ".join(f"{s[c]:8d}" for c in cols) + (f"   {n / s['mfma']:6.2f}" if s["mfma"] else ""))
n = sum(tot[c] for c in cols if c != "mfma")
print(f"{'total':>8s} " + " ".join(f"{tot[c]:8d}" for c in cols) + (f"   {n / tot['mfma']:6.2f}" if tot["mfma"] else ""))
if "--ops" in sys.argv:
    for i, s in enumerate(segs):
        top = sorted(((v, k[3:]) for k, v in s.items() if k.startswith("op:")), reverse=True)[:14]
        print(i, ", ".join(f"{k} {v}" for v, k in top))
