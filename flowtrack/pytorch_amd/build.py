"""Builds libflowtrack_hip.so (gfx950) in-tree with hipcc.

Replaces the reference's nvcc + torch.utils.ffi build (lib/make.sh,
lib/flownet/networks/*/make.sh, build.py): one shared library, no torch headers, C ABI only
(include/flowtrack_hip.h).  hipcc cross-compiles without a GPU, so this runs anywhere the ROCm
toolchain is installed; the resulting .so is git-ignored and travels with the tree.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIB_PATH = os.path.join(HERE, "libflowtrack_hip.so")
STAMP_PATH = os.path.join(HERE, ".libflowtrack_hip.stamp")

SOURCES = ["conv_igemm.hip", "conv_igemm8.hip", "bottleneck.hip", "bottleneck_rstat.hip", "bottleneck_stream.hip", "bottleneck_cluster.hip", "conv_direct.hip", "conv_wstat.hip", "aux_ops.hip", "flow_ops.hip", "crop_ops.hip", "runtime.hip"]
HEADERS = [os.path.join(CSRC, "ft_common.h"), os.path.join(CSRC, "conv_common.h"), os.path.join(CSRC, "conv_wstat.h"), os.path.join(INCLUDE, "flowtrack_hip.h")]
EXPORTS = os.path.join(CSRC, "exports.map")   # version script: only ft_* leaves the library
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _fingerprint() -> str:
    h = hashlib.sha256()
    for path in [os.path.join(CSRC, s) for s in SOURCES] + HEADERS + [EXPORTS]:
        with open(path, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS + [ARCH]).encode())
    return h.hexdigest()


def is_current() -> bool:
    if not (os.path.exists(LIB_PATH) and os.path.exists(STAMP_PATH)):
        return False
    try:
        with open(STAMP_PATH) as f:
            return f.read().strip() == _fingerprint()
    except OSError:
        return False


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile every .hip source for gfx950 and link libflowtrack_hip.so. Returns its path."""
    if not force and is_current():
        return LIB_PATH
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    objs = []
    fingerprint = _fingerprint()          # taken BEFORE compiling: an edit made while hipcc runs must not look built
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        # per-object stamp: one translation unit is recompiled only when it, a header or the flags changed
        h = hashlib.sha256()
        for path in [os.path.join(CSRC, src)] + HEADERS:
            with open(path, "rb") as f:
                h.update(f.read())
        h.update(" ".join(FLAGS + [ARCH]).encode())
        want, stamp = h.hexdigest(), obj + ".stamp"
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read().strip() == want:
            continue
        cmd = [hipcc, f"--offload-arch={ARCH}", *FLAGS, f"-I{INCLUDE}", f"-I{CSRC}", "-c",
               os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd), stamp, want))
    for src, p, stamp, want in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
        with open(stamp, "w") as f:
            f.write(want)
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", f"-Wl,--version-script={EXPORTS}", *objs, "-o", LIB_PATH]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP_PATH, "w") as f:
        f.write(fingerprint)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
