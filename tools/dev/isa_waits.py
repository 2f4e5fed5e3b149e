"""Compiler-inserted s_waitcnt vmcnt(N) in the built ISA of one translation unit (dev tool).

    python tools/dev/isa_waits.py flow_ops.hip [kernel-name substring] [-D...]

hipcc cannot see hand-written waits (inline asm): a register filled by a compiler-visible load before an unrolled loop and first used
behind a branch keeps its "may still be in flight" state at every join, and the compiler then waits again in front of every use —
down to vmcnt(0), which drains look-ahead loads and stores that the hand-counted waits meant to leave in flight (round 6: 23 x sixteen
descending waits in front of the MFMAs of correlation_mfma_rows64_kernel).  Per kernel this prints the MFMA count, the waits the
compiler put directly in front of an MFMA (by count value; many small values = suspicious), its other vmcnt waits, and the
hand-written ones (between ASMSTART / ASMEND) for comparison.  Scratch use (spills) is printed too: scratch loads count in vmcnt."""
import collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "flowtrack", "pytorch_amd", "csrc")


def build_isa(src, flags):
    out = os.path.join(tempfile.gettempdir(), os.path.basename(src).replace(".hip", ".isa_waits.s"))
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include", f"-I{CSRC}", "-S",
           "--cuda-device-only", *flags, src, "-o", out]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return out


def scan(path, want):
    kern, inasm = None, False
    res = collections.OrderedDict()
    lines = open(path).read().split("\n")
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kern = m.group(1)
            res[kern] = {"mfma": 0, "front": collections.Counter(), "other": collections.Counter(), "hand": collections.Counter(), "scratch": 0}
        if "ASMSTART" in l:
            inasm = True
        if "ASMEND" in l:
            inasm = False
        if kern is None:
            continue
        d = res[kern]
        if "v_mfma" in l:
            d["mfma"] += 1
        m = re.search(r"\.private_seg_size, (\d+)", l)
        if m and kern in l:
            d["scratch"] = int(m.group(1))
        m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", l)
        if not m:
            continue
        n = int(m.group(1))
        if inasm:
            d["hand"][n] += 1
            continue
        j = i + 1
        while j < len(lines) and (not lines[j].strip() or lines[j].strip().startswith(";")):
            j += 1
        nxt = lines[j].strip().split()[0] if j < len(lines) and lines[j].strip() else ""
        (d["front"] if nxt.startswith("v_mfma") else d["other"])[n] += 1
    for k, d in res.items():
        if d["mfma"] == 0 or (want and want not in k):
            continue
        demangled = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip() or k
        small = sum(c for n, c in d["front"].items() if n <= 2)
        print(f"{demangled[:150]}\n   MFMAs {d['mfma']}, scratch {d['scratch']} B, compiler waits in front of an MFMA: {sum(d['front'].values())}"
              f" ({small} with vmcnt <= 2) {dict(sorted(d['front'].items()))}\n   other compiler waits {dict(sorted(d['other'].items()))}"
              f"\n   hand-written {dict(sorted(d['hand'].items()))}")


if __name__ == "__main__":
    args = sys.argv[1:]
    if not args:
        sys.exit(__doc__)
    src = args[0] if os.path.isabs(args[0]) else os.path.join(CSRC, args[0])
    flags = [a for a in args[1:] if a.startswith("-")]
    want = next((a for a in args[1:] if not a.startswith("-")), "")
    scan(build_isa(src, flags), want)
