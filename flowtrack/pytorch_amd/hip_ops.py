"""Host-side operator layer over the C ABI (include/flowtrack_hip.h).

PyTorch is used here for device memory, streams and parameter bookkeeping only; every arithmetic
step of the pose / flow hot paths is a launch into libflowtrack_hip.so.  Nothing in this module
falls back to torch ops: a missing library or GPU raises FlowtrackHipError.

Vocabulary: an *activation view* (`ActView`) is an NHWC tensor [N, H, W, cstride] plus a channel
window [coff, coff + C) — the unit the fused conv reads and writes, so that `torch.cat` in the
reference graphs (lib/flownet/networks/FlowNetS.py:73-88) becomes "write into a slice".
"""
from __future__ import annotations

import ctypes
import functools
import json
import os
import sys
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import (FT_ACT_LEAKY, FT_ACT_NONE, FT_ACT_RELU, FT_LAYOUT_NCHW_F32, FT_LAYOUT_NHWC,
                   ConvDesc, ConvGeometry, FlowtrackHipError, check)

ACT_CODES = {None: FT_ACT_NONE, "none": FT_ACT_NONE, "relu": FT_ACT_RELU, "leaky": FT_ACT_LEAKY}


def require_gpu(device: torch.device) -> None:
    if device.type != "cuda" or not torch.cuda.is_available():
        raise FlowtrackHipError(
            "the flowtrack HIP path needs a ROCm GPU (model.cuda()); there is no CPU fallback — "
            "use oracle/ for CPU reference results")


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


@dataclass
class ActView:
    """Channel window [coff, coff+C) of an NHWC buffer t = [N, H, wpitch, cstride].

    Plain buffers have wpitch == W.  *Row-packed* buffers (inputs of the small-Cin 7x7 stems) carry
    `lpad` zero columns left of pixel 0 and zero columns up to `wpitch` on the right, so a whole kernel
    row is one contiguous K-run for the direct-to-LDS conv kernel (include/flowtrack_hip.h, x_wpitch)."""
    t: torch.Tensor
    C: int
    coff: int = 0
    lpad: int = 0
    width: Optional[int] = None   # logical W of a row-packed buffer

    @property
    def N(self): return self.t.shape[0]
    @property
    def H(self): return self.t.shape[1]
    @property
    def W(self): return self.width if self.width is not None else self.t.shape[2]
    @property
    def wpitch(self): return self.t.shape[2]
    @property
    def rowpacked(self): return self.width is not None
    @property
    def cstride(self): return self.t.shape[3]

    def batch_slice(self, lo: int, hi: int) -> "ActView":
        return ActView(self.t[lo:hi], self.C, self.coff, self.lpad, self.width)


def act_stride(C: int) -> int:
    """Channel stride of an NHWC buffer that holds C channels: a multiple of 32 once C >= 32 so the
    direct-to-LDS conv kernel (K-steps of 32 fp16 / 16 fp32 channels) can read it; 8 below that."""
    return round_up(C, 32) if C >= 32 else round_up(C, 8)


def new_rowpacked_act(N: int, H: int, W: int, C: int, pad: int, dtype, device) -> ActView:
    """Zero-filled row-packed input buffer for a stem conv with padding `pad`: channels padded to 4
    (C <= 4) or a multiple of 8, `pad` zero columns on each side (pitch rounded up to even)."""
    cpad = 4 if C <= 4 else round_up(C, 8)
    wpitch = round_up(W + 2 * pad, 2)
    return ActView(torch.zeros((N, H, wpitch, cpad), dtype=dtype, device=device), C, 0, pad, W)


def new_act(N: int, H: int, W: int, C: int, dtype, device, cstride: Optional[int] = None) -> ActView:
    """Zero-filled NHWC buffer; the padding channels [C, cstride) stay zero for the buffer's lifetime
    (the convs read them against zero weights)."""
    cs = act_stride(C) if cstride is None else cstride
    return ActView(torch.zeros((N, H, W, cs), dtype=dtype, device=device), C, 0)


# --------------------------------------------------------------------------------------------
# launch recording: a Program is the static launch sequence of one network at one input shape.
# --------------------------------------------------------------------------------------------
def is_conv_call(name: str) -> bool:
    """Launches whose work is convolution MACs (the roofline's kernels): ft_conv2d_fwd[_ws] and ft_bottleneck_fwd."""
    return name.startswith("ft_conv2d_fwd") or name in ("ft_bottleneck_fwd", "ft_bottleneck_rstat_fwd", "ft_bottleneck_stream_fwd", "ft_bottleneck_cluster_fwd", "ft_conv_direct_fwd")


def _on_plan_device(fn):
    """Run a Program method with the plan's GPU as the current HIP device: the library launches on the stream it is handed,
    but hipFuncSetAttribute / hipGetDevice / event creation inside it act on the CURRENT device, which is not the plan's
    when one process drives several GPUs."""
    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        if self.stream is None:
            return fn(self, *args, **kwargs)
        with torch.cuda.device(self.stream.device):
            return fn(self, *args, **kwargs)
    return wrapper


class Program:
    """Ordered list of C-ABI calls with fully bound arguments, replayable eagerly or as a HIP graph."""

    def __init__(self, stream: torch.cuda.Stream):
        self.lib = _lib.load()
        self.stream = stream
        self.calls: List[Tuple[str, tuple]] = []
        self.lanes: List[int] = []          # 0 = main stream, 1 = side stream (between fork / join)
        self.keepalive: list = []
        self.graph_exec = None
        self.flops = 0.0
        self.conv_records: list = []  # (label, call index, flops, ConvDesc) for per-layer timing / roofline / tuning
        self.fused_records: list = []  # (label, call index, flops) of the multi-conv launches (ft_bottleneck_fwd)
        # one scratch buffer shared by every conv of the plan (split-K partial tiles, ft_conv2d_fwd_ws); the calls hold
        # these two ctypes objects, whose values are filled in by _ensure_workspace() before the first launch
        self._ws_ptr, self._ws_size = ctypes.c_void_p(None), ctypes.c_size_t(0)
        self._ws_need, self._ws_tensor = 0, None
        # parallel branch: calls recorded between fork() and join() with side=True run on a second stream (a parallel
        # branch of the captured graph): two small independent launches share the GPU instead of queueing
        self._side = False
        self._side_stream: Optional[torch.cuda.Stream] = None
        self._choice_flops: list = []     # begin_choice() stack: [flops before the group, flops of its first option]
        self._events: list = []

    @property
    def stream_handle(self) -> ctypes.c_void_p:
        return ctypes.c_void_p(self.stream.cuda_stream)

    def add(self, name: str, *args, keep: Sequence = ()) -> None:
        self.calls.append((name, args))
        self.lanes.append(1 if self._side else 0)
        self.keepalive.extend(keep)

    def fork(self) -> None:
        """Everything recorded until join() is split in two branches that may overlap: `with prog.side():` calls go to
        the second stream, the others stay on the main one.  Both branches start after everything recorded so far."""
        self.calls.append(("__fork__", ()))
        self.lanes.append(0)

    def join(self) -> None:
        self.calls.append(("__join__", ()))
        self.lanes.append(0)

    # -- alternatives -------------------------------------------------------------------------------------------------
    # begin_choice(key) / option() ... / option() ... / end_choice(): the same piece of the network recorded in more than
    # one form (one fused launch or three conv launches; ft_conv_direct_fwd or ft_conv2d_fwd).  Every option writes the
    # same outputs from the same inputs, so until the choice is resolved ALL of them run, in recording order, and the
    # result is valid.  tune_choices() times them in situ and keeps the fastest, resolve_choices() keeps the first one
    # (the recorder's heuristic); both delete the other launches, so a captured plan never holds a choice.  Groups nest.
    def begin_choice(self, key: str) -> None:
        self.calls.append(("__choice__", (key,)))
        self.lanes.append(0)
        self._choice_flops.append([self.flops, None])

    def option(self, name: str = "") -> None:
        """Start the next option of the innermost open choice.  `name` identifies the FORM ("fused" / "convs", "direct" /
        "igemm"): benchmark picks are cached and persisted by name, never by position — the recording order follows
        heuristics (bottleneck_prefers_fused, whether the direct form exists at all) that may change between versions."""
        base = self._choice_flops[-1]
        if base[1] is None and self.calls[-1][0] != "__choice__":
            base[1] = self.flops - base[0]          # flops of the first option: all options are the same arithmetic
        self.flops = base[0]
        self.calls.append(("__option__", (name,)))
        self.lanes.append(0)

    def end_choice(self) -> None:
        base = self._choice_flops.pop()
        if base[1] is not None:
            self.flops = base[0] + base[1]
        self.calls.append(("__endchoice__", ()))
        self.lanes.append(0)

    def _innermost_choices(self) -> list:
        """Unresolved choice groups that contain no other group: dicts {key, marks: [option starts..., end]}."""
        stack, out = [], []
        for i, (name, args) in enumerate(self.calls):
            if name == "__choice__":
                if stack:
                    stack[-1]["nested"] = True
                stack.append({"key": args[0], "start": i, "marks": [], "names": [], "nested": False})
            elif name == "__option__":
                stack[-1]["marks"].append(i)
                stack[-1]["names"].append(args[0] or str(len(stack[-1]["names"])))
            elif name == "__endchoice__":
                g = stack.pop()
                g["marks"].append(i)
                if not g["nested"]:
                    out.append(g)
        if stack:
            raise FlowtrackHipError("begin_choice without end_choice")
        return out

    def _keep_options(self, groups: list, picks: list) -> None:
        """Delete the markers of `groups` and every option but picks[k]; re-index the per-launch records."""
        drop = [False] * len(self.calls)
        for g, pick in zip(groups, picks):
            drop[g["start"]] = True
            marks = g["marks"]
            for j in range(len(marks) - 1):
                for i in range(marks[j], marks[j + 1]):
                    drop[i] = drop[i] or j != pick or i == marks[j]
            drop[marks[-1]] = True
        new_index, n = {}, 0
        for i, d in enumerate(drop):
            if not d:
                new_index[i] = n
                n += 1
        self.calls = [c for c, d in zip(self.calls, drop) if not d]
        self.lanes = [l for l, d in zip(self.lanes, drop) if not d]
        self.conv_records = [(r[0], new_index[r[1]]) + tuple(r[2:]) for r in self.conv_records if r[1] in new_index]
        self.fused_records = [(r[0], new_index[r[1]]) + tuple(r[2:]) for r in self.fused_records if r[1] in new_index]

    def resolve_choices(self, cached_only: bool = False) -> None:
        """Keep the benchmarked pick of every choice where the cache holds one, otherwise its first option
        (cached_only: leave the choices without a cached pick in place, for tune_choices)."""
        if self.graph_exec is not None:
            return
        _load_tile_cache()
        while True:
            groups = self._innermost_choices()
            if cached_only:
                groups = [g for g in groups if _cached_choice(g) is not None]
            if not groups:
                return
            self._keep_options(groups, [_cached_choice(g) or 0 for g in groups])

    @_on_plan_device
    def tune_choices(self, reps: int = 3, verbose: bool = False) -> int:
        """In-situ benchmark of the recorded alternatives, innermost groups first: the whole launch list runs with hipEvents
        at the boundaries of each option; per key (all instances of one layer shape share a pick, as the tile variants
        do) the fastest option wins if it beats the recorder's first choice by 3 %.  Returns #groups that changed."""
        if self.graph_exec is not None:
            raise FlowtrackHipError("tune_choices must run before the plan is captured into a graph")
        self._ensure_workspace()
        _load_tile_cache()
        lib, sh = self.lib, self.stream_handle
        changed = 0
        while True:
            groups = self._innermost_choices()
            if not groups:
                break
            todo = [g for g in groups if _cached_choice(g) is None]
            if todo:
                marks = {}
                for g in todo:
                    g["ev"] = []
                    for i in g["marks"]:
                        e = ctypes.c_void_p()
                        check(lib.ft_event_create(ctypes.byref(e)), "ft_event_create")
                        g["ev"].append(e)
                        marks[i] = e
                    g["ms"] = [float("inf")] * (len(g["marks"]) - 1)
                # One pass per OPTION INDEX (round 5): pass j runs only option j of every group under test (groups with fewer
                # options: their first, untimed), so each option is timed behind the cache state its real predecessors leave.
                # Timing all options of a group back to back in one pass let the later ones find the group's input and the
                # previous option's output warm in L2 / MALL: worth 4-10 us per launch here (net_bench.py, NB_HOT=1), more than
                # the 3 % an option has to win by.  FT_CHOICE_SINGLE_PASS=1 restores the one-pass timing.
                single = os.environ.get("FT_CHOICE_SINGLE_PASS", "0") == "1"
                owner = {}
                for g in todo:
                    m = g["marks"]
                    for j in range(len(m) - 1):
                        for i in range(m[j] + 1, m[j + 1]):
                            owner[i] = (j, len(m) - 1)
                npass = 1 if single else max(len(g["ms"]) for g in todo)
                for _ in range(reps):
                    for jpass in range(npass):
                        with torch.cuda.stream(self.stream):
                            torch.cuda._sleep(4_000_000)
                        for i, (name, args) in enumerate(self.calls):
                            if name.startswith("__"):
                                if i in marks:
                                    check(lib.ft_event_record(marks[i], sh))
                                continue
                            o = owner.get(i)
                            if o is not None and not single and o[0] != (jpass if jpass < o[1] else 0):
                                continue
                            check(getattr(lib, name)(*args, sh), name)
                        check(lib.ft_stream_synchronize(sh), "ft_stream_synchronize")
                        for g in todo:
                            for j in (range(len(g["ms"])) if single else ([jpass] if jpass < len(g["ms"]) else [])):
                                ms = ctypes.c_float()
                                check(lib.ft_event_elapsed_ms(g["ev"][j], g["ev"][j + 1], ctypes.byref(ms)))
                                g["ms"][j] = min(g["ms"][j], ms.value)
                total, names = {}, {}
                for g in todo:
                    names[g["key"]] = g["names"]
                    tot = total.setdefault(g["key"], [0.0] * len(g["ms"]))
                    for j, v in enumerate(g["ms"]):
                        tot[j] += v
                    for e in g["ev"]:
                        lib.ft_event_destroy(e)
                for key, tot in total.items():
                    best = min(range(len(tot)), key=lambda j: tot[j])
                    if tot[best] > 0.97 * tot[0]:
                        best = 0
                    _TILE_CACHE["choice|" + key] = names[key][best]
                    changed += best != 0
                    if verbose:
                        print(f"[choice benchmark] {key:60s} " + "  ".join(f"{n} {v * 1e3:7.1f} us" for n, v in zip(names[key], tot)) +
                              f"  -> {names[key][best]}", file=sys.stderr)
            self._keep_options(groups, [_cached_choice(g) or 0 for g in groups])
        _save_tile_cache()
        return changed

    def side(self):
        prog = self

        class _Side:
            def __enter__(self_inner):
                prog._side = True

            def __exit__(self_inner, *exc):
                prog._side = False
                return False
        return _Side()

    def _event(self, i: int) -> ctypes.c_void_p:
        while len(self._events) <= i:
            e = ctypes.c_void_p()
            check(self.lib.ft_event_create(ctypes.byref(e)), "ft_event_create")
            self._events.append(e)
        return self._events[i]

    def need_workspace(self, nbytes: int) -> None:
        self._ws_need = max(self._ws_need, int(nbytes))

    def _ensure_workspace(self) -> None:
        if self._ws_need > self._ws_size.value:
            if self.graph_exec is not None:
                raise FlowtrackHipError("the plan's workspace cannot grow after graph capture")
            self._ws_tensor = torch.empty(self._ws_need, dtype=torch.uint8, device=self.stream.device)
            self._ws_ptr.value = self._ws_tensor.data_ptr()
            self._ws_size.value = self._ws_need

    #: FT_NO_BRANCHES=1: ignore fork / join (everything in recording order on one stream) — dev A/B switch
    use_branches = os.environ.get("FT_NO_BRANCHES") is None

    @_on_plan_device
    def run_eager(self, branches: bool = True) -> None:
        """Issue the launch list: on the program's stream, with the side-lane calls of each fork/join section on the
        second stream (branches=False: everything in order on the main stream, as the timing passes need it)."""
        self._ensure_workspace()
        sh = self.stream_handle
        lib = self.lib
        branches = branches and self.use_branches
        if branches and any(self.lanes) and self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=self.stream.device)
        side = ctypes.c_void_p(self._side_stream.cuda_stream) if (branches and self._side_stream is not None) else None
        nev = 0
        for (name, args), lane in zip(self.calls, self.lanes):
            if name == "__fork__":
                if side is not None:
                    ev = self._event(nev); nev += 1
                    check(lib.ft_event_record(ev, sh))
                    check(lib.ft_stream_wait_event(side, ev))
            elif name == "__join__":
                if side is not None:
                    ev = self._event(nev); nev += 1
                    check(lib.ft_event_record(ev, side))
                    check(lib.ft_stream_wait_event(sh, ev))
            elif not name.startswith("__"):     # (an unresolved choice runs all of its options: same outputs)
                check(getattr(lib, name)(*args, side if (lane and side is not None) else sh), name)

    @_on_plan_device
    def capture(self) -> None:
        """Record the launch sequence into a HIP graph (hipStreamBeginCapture on our side stream)."""
        if self.graph_exec is not None:
            return
        self.resolve_choices()
        sh = self.stream_handle
        check(self.lib.ft_graph_begin_capture(sh), "ft_graph_begin_capture")
        try:
            self.run_eager()
        finally:
            exec_ = ctypes.c_void_p()
            st = self.lib.ft_graph_end_capture(sh, ctypes.byref(exec_))
        check(st, "ft_graph_end_capture")
        self.graph_exec = exec_

    @_on_plan_device
    def run(self, stream: Optional[torch.cuda.Stream] = None) -> None:
        """Replay the captured graph on `stream` (default: the program's own stream; a graph may be launched on any
        stream, only the capture needed a private one), or run eagerly on the program's stream if not captured."""
        if self.graph_exec is not None:
            sh = self.stream_handle if stream is None else ctypes.c_void_p(stream.cuda_stream)
            check(self.lib.ft_graph_launch(self.graph_exec, sh), "ft_graph_launch")
        else:
            self.run_eager()

    @_on_plan_device
    def time_calls(self, iters: int = 5, repeat_hot: bool = False, median: bool = False):
        """Per-call hipEvent timing on the program's stream (eager). Returns [(name, ms_avg)] (median: the per-call MEDIAN over
        the iterations instead of the mean — one stalled iteration of a 12-us kernel otherwise shows as 80 us).
        repeat_hot (dev): every call is launched TWICE in a row and the second launch is the one timed — its weights and
        inputs are as warm in L2 as they can be; against the plain numbers this shows what a layer pays for arriving cold."""
        self._ensure_workspace()
        lib, sh = self.lib, self.stream_handle
        evs = []
        for _ in range(len(self.calls) + 1):
            e = ctypes.c_void_p()
            check(lib.ft_event_create(ctypes.byref(e)), "ft_event_create")
            evs.append(e)
        acc = [0.0] * len(self.calls)
        samples = [[] for _ in self.calls]
        for _ in range(iters):
            with torch.cuda.stream(self.stream):
                torch.cuda._sleep(4_000_000)   # device-side head start: the intervals below hold no host launch latency
            check(lib.ft_event_record(evs[0], sh))
            if repeat_hot:
                hot = []
                for i, (name, args) in enumerate(self.calls):
                    if name.startswith("__"):
                        continue
                    check(getattr(lib, name)(*args, sh), name)
                    check(lib.ft_event_record(evs[i], sh))
                    check(getattr(lib, name)(*args, sh), name)
                    e2 = ctypes.c_void_p()
                    check(lib.ft_event_create(ctypes.byref(e2)), "ft_event_create")
                    check(lib.ft_event_record(e2, sh))
                    hot.append((i, e2))
                check(lib.ft_event_synchronize(hot[-1][1]))
                for i, e2 in hot:
                    ms = ctypes.c_float()
                    check(lib.ft_event_elapsed_ms(evs[i], e2, ctypes.byref(ms)))
                    acc[i] += ms.value
                    lib.ft_event_destroy(e2)
                continue
            for i, (name, args) in enumerate(self.calls):
                if not name.startswith("__"):          # fork / join markers: the timing passes run everything in order
                    check(getattr(lib, name)(*args, sh), name)
                check(lib.ft_event_record(evs[i + 1], sh))
            check(lib.ft_event_synchronize(evs[-1]))
            for i in range(len(self.calls)):
                ms = ctypes.c_float()
                check(lib.ft_event_elapsed_ms(evs[i], evs[i + 1], ctypes.byref(ms)))
                acc[i] += ms.value
                samples[i].append(ms.value)
        for e in evs:
            lib.ft_event_destroy(e)
        if median and not repeat_hot:
            return [(self.calls[i][0], sorted(samples[i])[len(samples[i]) // 2]) for i in range(len(self.calls))]
        return [(self.calls[i][0], acc[i] / iters) for i in range(len(self.calls))]

    @_on_plan_device
    def time_conv_runs(self, iters: int = 5):
        """GPU time of the conv launches of one step, with hipEvents only at the boundaries of each maximal run of
        consecutive `ft_conv2d_fwd` calls (so the kernels run back to back exactly as in the graph and the intervals
        carry no per-launch event overhead) and a device-side head start (no host launch latency).
        Returns (conv_ms, other_ms) averaged over `iters` passes."""
        self._ensure_workspace()
        lib, sh = self.lib, self.stream_handle
        is_conv = [is_conv_call(name) for name, _ in self.calls]
        bounds = [0] + [i for i in range(1, len(is_conv)) if is_conv[i] != is_conv[i - 1]] + [len(is_conv)]
        evs = []
        for _ in bounds:
            e = ctypes.c_void_p()
            check(lib.ft_event_create(ctypes.byref(e)), "ft_event_create")
            evs.append(e)
        conv_ms = other_ms = 0.0
        for _ in range(iters):
            with torch.cuda.stream(self.stream):
                torch.cuda._sleep(4_000_000)
            for b in range(len(bounds) - 1):
                check(lib.ft_event_record(evs[b], sh))
                for name, args in self.calls[bounds[b]:bounds[b + 1]]:
                    if not name.startswith("__"):
                        check(getattr(lib, name)(*args, sh), name)
            check(lib.ft_event_record(evs[-1], sh))
            check(lib.ft_event_synchronize(evs[-1]))
            for b in range(len(bounds) - 1):
                ms = ctypes.c_float()
                check(lib.ft_event_elapsed_ms(evs[b], evs[b + 1], ctypes.byref(ms)))
                if is_conv[bounds[b]]:
                    conv_ms += ms.value
                else:
                    other_ms += ms.value
        for e in evs:
            lib.ft_event_destroy(e)
        return conv_ms / iters, other_ms / iters

    @_on_plan_device
    def tune_tiles(self, reps: int = 3, verbose: bool = False) -> int:
        """In-situ benchmark of the conv tile variants (the reference's `cudnn.benchmark = True`): the whole launch
        list runs eagerly `reps` times per candidate round with hipEvents around every conv, so each variant is timed
        with the cache state its real predecessors leave behind (isolated back-to-back timings of ONE layer flatter the
        big tiles and did not carry over to the network).  A device-side spin ahead of each pass lets the host enqueue
        everything before the first kernel starts, so the intervals hold no launch latency.  Returns #layers changed."""
        if self.graph_exec is not None:
            raise FlowtrackHipError("tune_tiles must run before the plan is captured into a graph")
        self._ensure_workspace()
        lib, sh = self.lib, self.stream_handle
        convs = [(i, rec[3]) for rec in self.conv_records for i in (rec[1],)]
        if not convs:
            return 0
        _load_tile_cache()
        keys = [("direct|" if self.calls[i][0] == "ft_conv_direct_fwd" else "") + _desc_key(d) for i, d in convs]
        if all(k in _TILE_CACHE for k in keys):          # every layer already benchmarked (this process or the file)
            for (_, d), k in zip(convs, keys):
                d.tile_hint = _TILE_CACHE[k]
            return sum(1 for k in keys if _TILE_CACHE[k])
        cands = []
        hints = (ctypes.c_int * 64)()   # ft_conv_tile_candidates enumerates at most ~45 forms (32 truncated the halo / split-K ones)
        for (_, d), key in zip(convs, keys):     # (_ = index of the launch in self.calls)
            if key in _TILE_CACHE:        # picks are sticky within a process: two plans of one model (other batch-
                cands.append([_TILE_CACHE[key]])   # independent keys aside) must run the same variants, bit for bit
                continue
            if self.calls[_][0] == "ft_conv_direct_fwd":   # one kernel, nothing to pick
                cands.append([0])
                continue
            n = lib.ft_conv_tile_candidates(ctypes.byref(d), hints, 64)
            if n < 0:
                check(-n, "ft_conv_tile_candidates")
            cands.append([0] + [int(h) for h in hints[:n]] if n > 1 else [0])
        rounds = max(len(c) for c in cands)
        if rounds == 1:
            for k, (_, d) in enumerate(convs):
                d.tile_hint = cands[k][0]
                _TILE_CACHE.setdefault(keys[k], cands[k][0])
            return 0
        conv_pos = {i: k for k, (i, _) in enumerate(convs)}
        ev = []
        for _ in range(2 * len(convs)):
            e = ctypes.c_void_p()
            check(lib.ft_event_create(ctypes.byref(e)), "ft_event_create")
            ev.append(e)
        times = [[float("inf")] * len(c) for c in cands]
        for r in range(rounds):
            for k, (_, d) in enumerate(convs):
                d.tile_hint = cands[k][r] if r < len(cands[k]) else 0
            for _ in range(reps):
                with torch.cuda.stream(self.stream):
                    torch.cuda._sleep(4_000_000)          # ~2 ms head start for the host
                for i, (name, args) in enumerate(self.calls):
                    if name.startswith("__"):
                        continue
                    k = conv_pos.get(i)
                    if k is not None:
                        check(lib.ft_event_record(ev[2 * k], sh))
                    check(getattr(lib, name)(*args, sh), name)
                    if k is not None:
                        check(lib.ft_event_record(ev[2 * k + 1], sh))
                check(lib.ft_stream_synchronize(sh), "ft_stream_synchronize")
                for k in range(len(convs)):
                    if r < len(cands[k]):
                        ms = ctypes.c_float()
                        check(lib.ft_event_elapsed_ms(ev[2 * k], ev[2 * k + 1], ctypes.byref(ms)))
                        times[k][r] = min(times[k][r], ms.value)
        for e in ev:
            lib.ft_event_destroy(e)
        # one pick per distinct layer description (all instances of e.g. layer3.[1-5].conv1 share it, judged on their
        # summed time): the variants differ in summation order, so a per-instance pick would make a freshly benchmarked
        # plan and a later plan served from the cache disagree in the last bits
        group_time = {}
        for k, key in enumerate(keys):
            tot = group_time.setdefault(key, [0.0] * len(cands[k]))
            for r in range(len(cands[k])):
                tot[r] += times[k][r]
        changed = 0
        for k, (_, d) in enumerate(convs):
            tot = group_time[keys[k]]
            best = min(range(len(cands[k])), key=lambda r: tot[r])
            if tot[best] > 0.97 * tot[0]:
                best = 0
            d.tile_hint = cands[k][best]
            _TILE_CACHE[keys[k]] = cands[k][best]
            changed += best != 0
            if verbose:
                h = cands[k][best]
                print(f"[tile benchmark] {self.conv_records[k][0]:28s} heuristic {times[k][0] * 1e3:7.1f} us  best {times[k][best] * 1e3:7.1f} us"
                      f"  -> bp {h & 0xfff} bc {(h >> 12) & 0x1ff} ks {(h >> 24) & 0xf} wide {(h >> 28) & 3}{' halo' if (h >> 30) & 1 else ''}"
                      f"{' splitK x%d' % ((1, 2, 4, 8, 3, 5, 6, 7)[(h >> 21) & 7]) if (h >> 21) & 7 else ''}", file=sys.stderr)
        _save_tile_cache()
        return changed

    def __del__(self):
        try:
            if self.graph_exec is not None:
                self.lib.ft_graph_destroy(self.graph_exec)
        except Exception:
            pass


# --------------------------------------------------------------------------------------------
# fused conv / transposed conv
# --------------------------------------------------------------------------------------------
def conv_geometry(d: ConvDesc) -> ConvGeometry:
    g = ConvGeometry()
    check(_lib.load().ft_conv_pack_geometry(ctypes.byref(d), ctypes.byref(g)), "ft_conv_pack_geometry")
    return g


def pack_conv_weights(weight: torch.Tensor, d: ConvDesc, *, dtype: torch.dtype,
                      device: torch.device) -> Tuple[torch.Tensor, int]:
    """Re-lay reference weights for the implicit-GEMM kernels, for the layer described by `d`.

    weight: Conv2d [Cout, Cin, kh, kw] or ConvTranspose2d [Cin, Cout, 4, 4] (reference layouts,
    SURVEY Appendix B).  Returns ([nphases, Cout_pad, Kpad] tensor of `dtype` on `device`,
    Cout_pad); k = tap * cin_pad + sub * run_cpad + ci, zeros in all padding.  Both the geometry (which
    depends on the kernel the library will choose for `d`) and the (tap, sub) -> (ky, kx) map come from
    the library (ft_conv_pack_geometry / ft_conv_tap_source), so packer and kernels cannot disagree."""
    lib = _lib.load()
    w = weight.detach().to(torch.float32).cpu()
    transposed = bool(d.transposed)
    cin, cout = (w.shape[0], w.shape[1]) if transposed else (w.shape[1], w.shape[0])
    g = conv_geometry(d)
    packed = torch.zeros((g.nphases, g.cout_pad, g.kpad), dtype=torch.float32)
    ky, kx = ctypes.c_int(), ctypes.c_int()
    for ph in range(g.nphases):
        for t in range(g.ntaps):
            for sub in range(g.run_taps):
                check(lib.ft_conv_tap_source(ctypes.byref(d), ph, t, sub, ctypes.byref(ky), ctypes.byref(kx)),
                      "ft_conv_tap_source")
                tap_w = w[:, :, ky.value, kx.value]
                if transposed:
                    tap_w = tap_w.t()                     # [Cout, Cin]
                k0 = t * g.cin_pad + sub * g.run_cpad
                packed[ph, :cout, k0:k0 + cin] = tap_w
    return packed.to(device=device, dtype=dtype).contiguous(), g.cout_pad


def fold_scale_shift(cout: int, cout_pad: int, bias: Optional[torch.Tensor], bn: Optional[dict],
                     device: torch.device) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
    """Eval-mode BatchNorm2d (running stats) and/or conv bias folded into fp32 scale/shift:
    y = conv * scale + shift, scale = gamma / sqrt(var + eps), shift = beta - mean * scale (+ bias * scale)."""
    scale = shift = None
    if bn is not None:
        g = bn["weight"].detach().double().cpu()
        b = bn["bias"].detach().double().cpu()
        m = bn["running_mean"].detach().double().cpu()
        v = bn["running_var"].detach().double().cpu()
        s = g / torch.sqrt(v + bn.get("eps", 1e-5))
        sh = b - m * s
        if bias is not None:
            sh = sh + bias.detach().double().cpu() * s
        scale = torch.ones(cout_pad, dtype=torch.float32)
        shift = torch.zeros(cout_pad, dtype=torch.float32)
        scale[:cout] = s.float()
        shift[:cout] = sh.float()
    elif bias is not None:
        shift = torch.zeros(cout_pad, dtype=torch.float32)
        shift[:cout] = bias.detach().float().cpu()
    if scale is not None:
        scale = scale.to(device)
    if shift is not None:
        shift = shift.to(device)
    return scale, shift


#: first-run benchmark of the conv tile variants (Program.tune_tiles; the reference's `cudnn.benchmark = True`,
#: tools/pose/main.py:59).  FT_CONV_BENCHMARK=0 turns it off (the library's heuristic picks every tile).
benchmark = os.environ.get("FT_CONV_BENCHMARK", "1") != "0"
#: optional JSON file that persists the benchmark's picks across processes (descriptor -> tile_hint), the role
#: MIOpen's perf-db plays for the reference's cudnn/MIOpen path.  FT_TILE_CACHE=<path>; unset = in-process only.
tile_cache_path = os.environ.get("FT_TILE_CACHE") or None
_TILE_CACHE: dict = {}
_TILE_CACHE_LOADED = False


def _cached_choice(group: dict):
    """Index of the cached pick of a choice group (picks are stored by option NAME), or None when the cache has none or names
    a form this recording does not offer (a stale file: the group is benchmarked again)."""
    name = _TILE_CACHE.get("choice|" + group["key"])
    if not isinstance(name, str) or name not in group["names"]:
        return None
    return group["names"].index(name)


def _desc_key(d: ConvDesc) -> str:
    return ",".join(str(getattr(d, f)) for f, _ in ConvDesc._fields_ if f != "tile_hint")


def _load_tile_cache() -> None:
    global _TILE_CACHE_LOADED
    if _TILE_CACHE_LOADED:
        return
    _TILE_CACHE_LOADED = True
    if tile_cache_path and os.path.isfile(tile_cache_path):
        try:
            with open(tile_cache_path) as f:
                # tile picks are integers (tile_hint values); choice picks are option names (a pre-name file's bare indices
                # are dropped: the order they referred to is not recoverable)
                _TILE_CACHE.update({k: (str(v) if k.startswith("choice|") else int(v)) for k, v in json.load(f).items()
                                    if not (k.startswith("choice|") and not isinstance(v, str))})
        except (OSError, ValueError):
            pass


def _save_tile_cache() -> None:
    if not tile_cache_path:
        return
    tmp = f"{tile_cache_path}.{os.getpid()}.tmp"
    try:
        with open(tmp, "w") as f:
            json.dump(_TILE_CACHE, f, indent=0, sort_keys=True)
        os.replace(tmp, tile_cache_path)
    except OSError:
        pass


class FusedConv:
    """One Conv2d / ConvTranspose2d(4,2,1) with its folded BN / bias and activation, packed for HIP."""

    def __init__(self, weight: torch.Tensor, *, dtype: torch.dtype, device: torch.device, stride: int = 1,
                 pad: int = 0, transposed: bool = False, bias: Optional[torch.Tensor] = None,
                 bn: Optional[dict] = None, act: Optional[str] = None, slope: float = 0.0, label: str = "",
                 tail_weight: Optional[torch.Tensor] = None, tail_bias: Optional[torch.Tensor] = None):
        """tail_weight [n <= 32, Cout, 1, 1] (+ tail_bias [n]): a 1x1 conv fused behind this layer's activation
        (ft_conv_desc.tail_cout); record() then writes the TAIL's output and this layer's own output never exists."""
        require_gpu(device)
        self.lib = _lib.load()
        self.dtype, self.device = dtype, device
        self.code = _lib.dtype_code(dtype)
        self.transposed = transposed
        if transposed:
            self.cin, self.cout, self.k, _ = weight.shape
        else:
            self.cout, self.cin, self.k, _ = weight.shape
        self.stride, self.pad = stride, pad
        self.act, self.slope, self.label = ACT_CODES[act], float(slope), label
        if weight.shape[2] != weight.shape[3]:
            raise FlowtrackHipError("only square kernels are used by the FlowTrack hot paths")
        self._weight = weight.detach().to(torch.float32).cpu()
        self._bias = None if bias is None else bias.detach().to(torch.float32).cpu()
        self._bn = None if bn is None else {k: (v.detach().to(torch.float32).cpu() if torch.is_tensor(v) else v)
                                            for k, v in bn.items()}
        self._packed = {}  # (cin_pad-dependent) geometry key -> (w, cout_pad, scale, shift)
        self.tail_cout, self._tail = 0, None
        if tail_weight is not None:
            n = tail_weight.shape[0]
            if dtype != torch.float16 or tuple(tail_weight.shape[1:]) != (self.cout, 1, 1) or n > 32 or self.cout not in (64, 128, 256):
                raise FlowtrackHipError(f"{label}: the fused tail needs fp16, a 1x1 conv to <= 32 channels on 64/128/256 inputs")
            # fp32-grade tail weights as an fp16 hi / lo pair (w = hi + lo up to ~2^-22 relative): the heatmap conv decides
            # the arg-max, its weights are not rounded to fp16 (the activations of the last deconv still are)
            w32 = tail_weight.detach().float().cpu()[:, :, 0, 0]
            w16 = torch.zeros((32, self.cout), dtype=torch.float16)
            w16[:n] = w32.half()
            wlo = torch.zeros((32, self.cout), dtype=torch.float16)
            wlo[:n] = (w32 - w16[:n].float()).half()
            b32 = torch.zeros(32, dtype=torch.float32)
            if tail_bias is not None:
                b32[:n] = tail_bias.detach().float().cpu()
            parts = [w16.view(torch.uint8).flatten(), b32.view(torch.uint8).flatten(), wlo.view(torch.uint8).flatten()]
            if self.cout == 256:
                # 4th section (include/flowtrack_hip.h, tail pack): the same hi / lo weights in the 8-phase tile's operand order,
                # fp16 [channel half wc][step st][hi, lo][lane = lhi * 32 + tail output][8]: channels wc * 128 + st * 16 + 4 lhi +
                # {0..3} and + 8 + {0..3} — the K order in which that kernel's accumulator registers hold a pixel's channels
                wc, st, lhi, e = torch.meshgrid(torch.arange(2), torch.arange(8), torch.arange(2), torch.arange(8), indexing="ij")
                ch = wc * 128 + st * 16 + 4 * lhi + (e % 4) + 8 * (e // 4)                  # [2, 8, 2, 8]
                perm = torch.stack([t[:, ch] for t in (w16, wlo)], dim=0)                    # [hl, 32 rows, wc, st, lhi, e]
                perm = perm.permute(2, 3, 0, 4, 1, 5).contiguous()                           # [wc, st, hl, lhi, row, e]
                parts.append(perm.view(torch.uint8).flatten())
            self._tail = torch.cat(parts).to(device)
            self.tail_cout = n

    def _packed_for(self, d: ConvDesc):
        """Packed weights + folded scale/shift for the kernel the library picks for `d` (cached per layout)."""
        key = conv_geometry(d).key()
        hit = self._packed.get(key)
        if hit is None:
            w, cout_pad = pack_conv_weights(self._weight, d, dtype=self.dtype, device=self.device)
            scale, shift = fold_scale_shift(self.cout, cout_pad, self._bias, self._bn, self.device)
            hit = self._packed[key] = (w, cout_pad, scale, shift)
        return hit

    def out_hw(self, H: int, W: int) -> Tuple[int, int]:
        if self.transposed:
            return 2 * H, 2 * W
        return (H + 2 * self.pad - self.k) // self.stride + 1, (W + 2 * self.pad - self.k) // self.stride + 1

    def record(self, prog: Program, x: ActView, y, residual: Optional[ActView] = None, pool: bool = False,
               shift_n=None, x_nchw: Optional[torch.Tensor] = None) -> None:
        """Append this layer to `prog`. y is an ActView (NHWC) or a contiguous NCHW fp32 tensor.
        pool=True (the ResNet stem): the 3x3/s2/p1 max-pool runs inside the launch, y is the pooled map.
        shift_n: callable(scale, shift) -> fp32 [N, Cout] tensor of PER-SAMPLE shifts (ft_conv_desc.shift_nstride); it is
        called with the layer's folded tables right before the conv launch is appended, so it can record the launch that
        fills the tensor (FlowNet2S's rgb mean folded into conv1: ft_flow_mean_fold).
        x_nchw (with pool=True): the network's NCHW fp32 input [N, Cin, H, W] itself (ft_conv_desc.x_nchw_f32) — `x` then is
        only the GEOMETRY of the row-packed view the kernel builds in LDS (its tensor may live on the `meta` device) and no
        pack launch is needed."""
        if x.C != self.cin:
            raise FlowtrackHipError(f"{self.label}: input has {x.C} channels, layer expects {self.cin}")
        if x.t.dtype != self.dtype or not x.t.is_contiguous():
            raise FlowtrackHipError(f"{self.label}: input buffer must be contiguous {self.dtype}")
        if x_nchw is not None:
            if not pool or not x.rowpacked or tuple(x_nchw.shape) != (x.N, self.cin, x.H, x.W) or x_nchw.dtype != torch.float32 \
                    or not x_nchw.is_contiguous():
                raise FlowtrackHipError(f"{self.label}: x_nchw must be the contiguous fp32 [N, Cin, H, W] input of a pooled stem")
        elif x.t.device.type == "meta":
            raise FlowtrackHipError(f"{self.label}: a geometry-only input view needs x_nchw")
        Ho, Wo = self.out_hw(x.H, x.W)
        d = ConvDesc()
        d.dtype = self.code
        d.N, d.Hi, d.Wi = x.N, x.H, x.W
        d.Cin, d.x_cstride, d.x_coff = self.cin, x.cstride, x.coff
        if x.rowpacked:
            d.x_lpad, d.x_wpitch = x.lpad, x.wpitch
        d.Cout, d.kh, d.kw = self.cout, self.k, self.k
        d.stride, d.pad, d.transposed = self.stride, self.pad, int(self.transposed)
        d.Ho, d.Wo = Ho, Wo
        out_c = self.tail_cout or self.cout
        if pool:
            if Ho % 2 or Wo % 2 or not isinstance(y, ActView) or residual is not None or self.tail_cout:
                raise FlowtrackHipError(f"{self.label}: the fused max-pool needs an even conv output and a plain NHWC result")
            d.pool = 1
        yH, yW = (Ho // 2, Wo // 2) if pool else (Ho, Wo)
        if isinstance(y, ActView):
            if (y.N, y.H, y.W) != (x.N, yH, yW) or y.C != out_c or y.t.dtype != self.dtype:
                raise FlowtrackHipError(f"{self.label}: output view mismatch {tuple(y.t.shape)} C={y.C}")
            d.y_cstride, d.y_coff, d.out_layout = y.cstride, y.coff, FT_LAYOUT_NHWC
            yt = y.t
        else:
            if tuple(y.shape) != (x.N, out_c, Ho, Wo) or y.dtype != torch.float32 or not y.is_contiguous():
                raise FlowtrackHipError(f"{self.label}: NCHW output must be contiguous fp32 {(x.N, out_c, Ho, Wo)}")
            d.y_cstride, d.y_coff, d.out_layout = 0, 0, FT_LAYOUT_NCHW_F32
            yt = y
        res_ptr = None
        if residual is not None:
            if (residual.N, residual.H, residual.W) != (x.N, Ho, Wo) or residual.C != self.cout:
                raise FlowtrackHipError(f"{self.label}: residual view mismatch")
            d.has_residual, d.res_cstride, d.res_coff = 1, residual.cstride, residual.coff
            res_ptr = residual.t.data_ptr()
        d.act, d.slope = self.act, self.slope
        if self.tail_cout:
            if residual is not None:
                raise FlowtrackHipError(f"{self.label}: a fused tail excludes a residual input")
            d.tail_cout = self.tail_cout
            res_ptr = self._tail.data_ptr()
        x_ptr, x_keep = x.t.data_ptr() if x_nchw is None else x_nchw.data_ptr(), x.t if x_nchw is None else x_nchw
        if x_nchw is not None:
            d.x_nchw_f32 = 1
        w, _, scale, shift = self._packed_for(d)
        if shift_n is not None:
            d.shift_nstride = self.cout
            if self.lib.ft_conv_shift_nstride_supported(ctypes.byref(d)) != 0:
                raise FlowtrackHipError(f"{self.label}: per-sample shift is not supported for this layer / shape")
            shift = shift_n(scale, shift)
            if tuple(shift.shape) != (x.N, self.cout) or shift.dtype != torch.float32 or not shift.is_contiguous():
                raise FlowtrackHipError(f"{self.label}: per-sample shift must be contiguous fp32 {(x.N, self.cout)}")
        flops = float(self.lib.ft_conv_flops(ctypes.byref(d)))
        ws = _direct_stream(self, d, w, x.t.device) if (self.k in (1, 3, 5) and isinstance(y, ActView) and not self.tail_cout and not pool) else None
        if ws is not None:
            # two forms of the same launch; the in-situ benchmark (Program.tune_choices) keeps the faster one
            dd = ConvDesc.from_buffer_copy(d)
            prog.begin_choice("conv|" + _desc_key(d))
            prog.option("direct")
            prog.flops += flops
            prog.conv_records.append((self.label, len(prog.calls), flops, dd))
            prog.add("ft_conv_direct_fwd", ctypes.byref(dd), x.t.data_ptr(), ws.data_ptr(),
                     scale.data_ptr() if scale is not None else None, shift.data_ptr() if shift is not None else None, res_ptr,
                     yt.data_ptr(), keep=(dd, x.t, yt, ws, scale, shift, residual.t if residual is not None else None))
            prog.option("igemm")
        self._record_igemm(prog, d, x, w, scale, shift, res_ptr, yt, residual, flops, x_ptr, x_keep)
        if ws is not None:
            prog.end_choice()

    def _record_igemm(self, prog: Program, d: ConvDesc, x: ActView, w, scale, shift, res_ptr, yt, residual, flops: float,
                      x_ptr=None, x_keep=None) -> None:
        x_ptr = x.t.data_ptr() if x_ptr is None else x_ptr
        x_keep = x.t if x_keep is None else x_keep
        prog.flops += flops
        prog.conv_records.append((self.label, len(prog.calls), flops, d))
        if prog._side:     # side-branch launches may overlap main-branch ones: they must not share the plan's workspace
            prog.add("ft_conv2d_fwd", ctypes.byref(d), x_ptr, w.data_ptr(),
                     scale.data_ptr() if scale is not None else None,
                     shift.data_ptr() if shift is not None else None, res_ptr, yt.data_ptr(),
                     keep=(d, x_keep, yt, w, scale, shift, residual.t if residual is not None else None, self._tail))
            return
        prog.need_workspace(self.lib.ft_conv_workspace_bytes(ctypes.byref(d)))
        prog.add("ft_conv2d_fwd_ws", ctypes.byref(d), x_ptr, w.data_ptr(),
                 scale.data_ptr() if scale is not None else None,
                 shift.data_ptr() if shift is not None else None, res_ptr, yt.data_ptr(), prog._ws_ptr, prog._ws_size,
                 keep=(d, x_keep, yt, w, scale, shift, residual.t if residual is not None else None, self._tail))


#: 1x1 layers with few pixels and a long K run on ft_conv_direct_fwd (weights straight to registers); FT_CONV_DIRECT=0 keeps
#: them on ft_conv2d_fwd, FT_CONV_DIRECT_MAX_PIXELS moves the pixel bound (default 65536: ResNet layer2.0's block exit, layer3
#: and layer4 at batch 64; measured same-box on R50: layer3.0.conv1 34 -> 26 us, layer2.0.conv3+downsample 44 -> 36 us, but
#: layer2.0.conv1 at 196608 pixels 50 -> 95 us)
CONV_DIRECT = os.environ.get("FT_CONV_DIRECT", "1") != "0"
CONV_DIRECT_MAX_PIXELS = int(os.environ.get("FT_CONV_DIRECT_MAX_PIXELS", "65536"))


def _direct_stream(owner, d: ConvDesc, w: torch.Tensor, device) -> Optional[torch.Tensor]:
    """The ft_conv_direct weight stream of `d` (built once per packed weight set, cached on the layer), or None when the
    layer does not qualify."""
    lib = _lib.load()
    # (5x5 / stride 2 on 64 channels — FlowNet's conv2 — has a form of its own: weights stationary in registers, any pixel count)
    wstat = d.kh == 5 and d.Cin == 64 and d.stride == 2 and not d.x2_cin
    if not CONV_DIRECT or d.dtype != _lib.FT_F16 or (d.Cin + d.x2_cin < 256 and not wstat) or (d.kh == 5 and not wstat):
        return None
    if not wstat and d.N * d.Ho * d.Wo > CONV_DIRECT_MAX_PIXELS and not (CONV_DIRECT_MAX_PIXELS >= 65536 and d.kh == 1 and d.Cin in (256, 512)
                                                          and not d.x2_cin and not d.has_residual):
        return None     # (beyond the bound only the weight-stationary short-K forms exist: ResNet layer2.0 / layer3.0 conv1)
    if lib.ft_conv_direct_supported(ctypes.byref(d)) != 0:
        return None
    # one stream per LAYOUT: the kernel form (K split 1 / 4, weight-stationary, whole-map 3x3) follows the pixel count, every
    # form orders the fragments differently and all of them have the same byte count — ft_conv_direct_stream_id tells them
    # apart, so two plans of one model at batch sizes on either side of a form boundary each get their own stream
    sid = int(lib.ft_conv_direct_stream_id(ctypes.byref(d)))
    if sid < 0:
        return None
    key = ("direct", sid, w.data_ptr())
    hit = owner._packed.get(key)
    if hit is None:
        ws = torch.empty(int(lib.ft_conv_direct_weight_bytes(ctypes.byref(d))), dtype=torch.uint8, device=device)
        check(lib.ft_conv_direct_pack(ctypes.byref(d), w.data_ptr(), int(w.shape[-1]), int(w.shape[-2]), ws.data_ptr(),
                                      current_stream_handle(device)), "ft_conv_direct_pack")
        torch.cuda.current_stream(device).synchronize()
        hit = owner._packed[key] = ws
    return hit


class FusedShortcutConv:
    """conv3 + bn3 of a bottleneck block and its projection shortcut (`downsample` = 1x1 stride-s conv + bn,
    lib/pose/models/blocks.py:104-119) as ONE launch:  out = relu(bn3(conv3(t2)) + bn_d(conv_d(x))).
    Both BatchNorms are folded into the weights (W' = diag(scale) W), the two 1x1 convs become one GEMM over the
    concatenated K = [channels of t2 | channels of x] (`ft_conv_desc.x2_*`), the shifts add up.  The shortcut tensor
    is never written to / re-read from HBM and its launch disappears."""

    def __init__(self, w3: torch.Tensor, bn3: dict, wd: torch.Tensor, bnd: dict, stride_d: int, *, dtype: torch.dtype,
                 device: torch.device, act: Optional[str] = "relu", label: str = ""):
        require_gpu(device)
        self.lib = _lib.load()
        self.dtype, self.device, self.code = dtype, device, _lib.dtype_code(dtype)
        self.cout, self.cin = w3.shape[0], w3.shape[1]
        self.cin2, self.stride_d = wd.shape[1], stride_d
        if wd.shape[0] != self.cout or w3.shape[2:] != (1, 1) or wd.shape[2:] != (1, 1):
            raise FlowtrackHipError(f"{label}: conv3 / downsample must be 1x1 convs with the same output channels")
        self.act, self.label = ACT_CODES[act], label

        def folded(w, bn):
            s, sh = fold_scale_shift(self.cout, self.cout, None, bn, torch.device("cpu"))
            return w.detach().to(torch.float32).cpu()[:, :, 0, 0] * s[:, None], sh
        self._w3, sh3 = folded(w3, bn3)
        self._wd, shd = folded(wd, bnd)
        self._shift = sh3 + shd
        self._packed = {}

    def _packed_for(self, d: ConvDesc):
        g = conv_geometry(d)
        key = g.key()
        hit = self._packed.get(key)
        if hit is None:
            w = torch.zeros((1, g.cout_pad, g.kpad), dtype=torch.float32)
            w[0, :self.cout, :self.cin] = self._w3
            w[0, :self.cout, g.cin_pad:g.cin_pad + self.cin2] = self._wd
            shift = torch.zeros(g.cout_pad, dtype=torch.float32)
            shift[:self.cout] = self._shift
            hit = self._packed[key] = (w.to(device=self.device, dtype=self.dtype).contiguous(), shift.to(self.device))
        return hit

    def record(self, prog: "Program", t2: ActView, x: ActView, y: ActView) -> None:
        if t2.C != self.cin or x.C != self.cin2 or y.C != self.cout:
            raise FlowtrackHipError(f"{self.label}: channel mismatch")
        if (x.N, (x.H - 1) // self.stride_d + 1, (x.W - 1) // self.stride_d + 1) != (t2.N, t2.H, t2.W) or (y.N, y.H, y.W) != (t2.N, t2.H, t2.W):
            raise FlowtrackHipError(f"{self.label}: shortcut / main / output sizes do not line up")
        d = ConvDesc()
        d.dtype = self.code
        d.N, d.Hi, d.Wi = t2.N, t2.H, t2.W
        d.Cin, d.x_cstride, d.x_coff = self.cin, t2.cstride, t2.coff
        d.Cout, d.kh, d.kw, d.stride, d.pad, d.transposed = self.cout, 1, 1, 1, 0, 0
        d.Ho, d.Wo = t2.H, t2.W
        d.y_cstride, d.y_coff, d.out_layout = y.cstride, y.coff, FT_LAYOUT_NHWC
        d.act, d.slope = self.act, 0.0
        d.x2_cin, d.x2_hi, d.x2_wi, d.x2_cstride, d.x2_coff, d.x2_stride = self.cin2, x.H, x.W, x.cstride, x.coff, self.stride_d
        w, shift = self._packed_for(d)
        flops = float(self.lib.ft_conv_flops(ctypes.byref(d)))
        ws = _direct_stream(self, d, w, t2.t.device)
        if ws is not None:
            dd = ConvDesc.from_buffer_copy(d)
            prog.begin_choice("conv|" + _desc_key(d))
            prog.option("direct")
            prog.flops += flops
            prog.conv_records.append((self.label, len(prog.calls), flops, dd))
            prog.add("ft_conv_direct_fwd", ctypes.byref(dd), t2.t.data_ptr(), ws.data_ptr(), None, shift.data_ptr(), x.t.data_ptr(),
                     y.t.data_ptr(), keep=(dd, t2.t, x.t, y.t, ws, shift))
            prog.option("igemm")
        prog.flops += flops
        prog.conv_records.append((self.label, len(prog.calls), flops, d))
        prog.add("ft_conv2d_fwd", ctypes.byref(d), t2.t.data_ptr(), w.data_ptr(), None, shift.data_ptr(), x.t.data_ptr(),
                 y.t.data_ptr(), keep=(d, t2.t, x.t, y.t, w, shift))
        if ws is not None:
            prog.end_choice()


def bottleneck_fusable(c1: "FusedConv", c2: "FusedConv", c3: "FusedConv", x: ActView, y: ActView) -> bool:
    """True when ft_bottleneck_fwd covers this identity-shortcut block (fp16, 256 -> 64 -> 64 -> 256, stride 1)."""
    if x.t.dtype != torch.float16 or x.rowpacked or (x.N, x.H, x.W, x.C) != (y.N, y.H, y.W, y.C):
        return False
    if (c1.k, c2.k, c3.k) != (1, 3, 1) or (c2.stride, c2.pad) != (1, 1) or c1.stride != 1 or c3.stride != 1:
        return False
    if (c1.cin, c1.cout, c2.cin, c2.cout, c3.cin, c3.cout) != (x.C, c2.cin, c2.cin, c2.cin, c2.cin, x.C):
        return False
    if any(c.transposed or c.tail_cout or c.act != ACT_CODES["relu"] or c._bn is None for c in (c1, c2, c3)):
        return False
    d = _bottleneck_desc(x, y, c2.cin)
    lib = _lib.load()
    return lib.ft_bottleneck_supported(ctypes.byref(d)) == 0 or (
        FUSE_BOTTLENECK_STREAM and lib.ft_bottleneck_stream_supported(ctypes.byref(d)) == 0)


#: the streamed-weights fused bottleneck (ft_bottleneck_stream_fwd) for the 128- / 256-plane stages; FT_FUSE_BOTTLENECK_STREAM=0
#: keeps their three conv launches
FUSE_BOTTLENECK_STREAM = os.environ.get("FT_FUSE_BOTTLENECK_STREAM", "1") != "0"


def bottleneck_prefers_fused(x: ActView, planes: int) -> bool:
    """The recorder's first choice for a fusable identity block (what runs when the first-call benchmark is off): the
    64-plane kernel always; the streamed-weights kernels only with enough pixels to spread their per-workgroup weight
    stream over (measured: 256 planes at 3072 px 50 us fused vs 27 us as three launches, at 12288 px 53 vs 82 us)."""
    px = x.N * x.H * x.W
    return planes == 64 or (planes == 128 and px >= 16384) or (planes == 256 and px >= 8192)


def _bottleneck_desc(x: ActView, y: ActView, planes: int, head_only: bool = False) -> _lib.BottleneckDesc:
    d = _lib.BottleneckDesc()
    d.dtype = _lib.dtype_code(x.t.dtype)
    d.N, d.H, d.W, d.C, d.P = x.N, x.H, x.W, x.C, planes
    d.x_cstride, d.x_coff, d.y_cstride, d.y_coff = x.cstride, x.coff, y.cstride, y.coff
    d.head_only = int(head_only)
    return d


def bottleneck_head_fusable(c1: "FusedConv", c2: "FusedConv", x: ActView, t2: ActView) -> bool:
    """True when ft_bottleneck_fwd(head_only) covers conv1 + conv2 of a stage's entry block (fp16, 64 -> 64 -> 64, stride 1)."""
    if x.t.dtype != torch.float16 or x.rowpacked or (x.N, x.H, x.W) != (t2.N, t2.H, t2.W):
        return False
    if (c1.k, c2.k) != (1, 3) or (c1.stride, c2.stride, c2.pad) != (1, 1, 1):
        return False
    if (c1.cin, c1.cout, c2.cin, c2.cout, t2.C) != (x.C, c2.cin, c2.cin, c2.cin, c2.cin):
        return False
    if any(c.transposed or c.tail_cout or c.act != ACT_CODES["relu"] or c._bn is None for c in (c1, c2)):
        return False
    d = _bottleneck_desc(x, t2, c2.cin, head_only=True)
    return _lib.load().ft_bottleneck_supported(ctypes.byref(d)) == 0


def _bottleneck_packed(conv: "FusedConv", x: ActView, want, label: str):
    d = ConvDesc()
    d.dtype = conv.code
    d.N, d.Hi, d.Wi, d.Ho, d.Wo = x.N, x.H, x.W, x.H, x.W
    d.Cin, d.x_cstride, d.x_coff = conv.cin, act_stride(conv.cin), 0
    d.Cout, d.kh, d.kw, d.stride, d.pad = conv.cout, conv.k, conv.k, 1, conv.pad
    d.y_cstride, d.y_coff, d.out_layout = act_stride(conv.cout), 0, FT_LAYOUT_NHWC
    d.act = conv.act
    g = conv_geometry(d)
    if (g.cout_pad, g.kpad, g.nphases) != want:
        raise FlowtrackHipError(f"{label}: unexpected packed layout {(g.cout_pad, g.kpad)} for the fused bottleneck")
    w, _, scale, shift = conv._packed_for(d)
    return w, scale, shift


def record_bottleneck_head(prog: Program, c1: "FusedConv", c2: "FusedConv", x: ActView, t2: ActView, label: str) -> None:
    """conv1 + bn1 + relu -> conv2 + bn2 + relu of a stage's entry block as one launch (ft_bottleneck_fwd, head_only)."""
    lib = _lib.load()
    planes = c2.cin
    w1, s1, b1 = _bottleneck_packed(c1, x, (planes, x.C, 1), label)
    w2, s2, b2 = _bottleneck_packed(c2, x, (planes, 9 * planes, 1), label)
    d = _bottleneck_desc(x, t2, planes, head_only=True)
    check(lib.ft_bottleneck_supported(ctypes.byref(d)), "ft_bottleneck_supported")
    flops = float(lib.ft_bottleneck_flops(ctypes.byref(d)))
    prog.flops += flops
    prog.fused_records.append((label, len(prog.calls), flops))
    table = torch.cat([t.flatten()[:planes] for t in (s1, b1, s2, b2)]).contiguous()
    prog.add("ft_bottleneck_fwd", ctypes.byref(d), x.t.data_ptr(), w1.data_ptr(), w2.data_ptr(), None, table.data_ptr(),
             t2.t.data_ptr(), keep=(d, x.t, t2.t, w1, w2, table))


def bottleneck_head_stream_fusable(c1: "FusedConv", c2: "FusedConv", x: ActView, t2: ActView) -> bool:
    """True when ft_bottleneck_stream_fwd(head_only, stride 2) covers conv1 + conv2 of a stage's entry block (fp16, 512 -> 256
    -> 256 with the stride on conv2: layer3.0, resnet.py:44-49)."""
    if not FUSE_BOTTLENECK_STREAM or x.t.dtype != torch.float16 or x.rowpacked or (x.N, x.H // 2, x.W // 2) != (t2.N, t2.H, t2.W):
        return False
    if (c1.k, c2.k) != (1, 3) or (c1.stride, c2.stride, c2.pad) != (1, 2, 1):
        return False
    if (c1.cin, c1.cout, c2.cin, c2.cout, t2.C) != (x.C, c2.cin, c2.cin, c2.cin, c2.cin):
        return False
    if any(c.transposed or c.tail_cout or c.act != ACT_CODES["relu"] or c._bn is None for c in (c1, c2)):
        return False
    d = _bottleneck_desc(x, t2, c2.cin, head_only=True)
    d.stride = 2
    return _lib.load().ft_bottleneck_stream_supported(ctypes.byref(d)) == 0


def record_bottleneck_head_stream(prog: Program, c1: "FusedConv", c2: "FusedConv", x: ActView, t2: ActView, label: str) -> None:
    """conv1 + bn1 + relu -> conv2 (stride 2) + bn2 + relu of a 256-plane entry block as one launch: t1 stays in LDS at the
    input resolution, the weights stream from L2 (ft_bottleneck_stream_fwd, head_only + stride 2)."""
    lib = _lib.load()
    planes = c2.cin
    w1, s1, b1 = _bottleneck_packed(c1, x, (planes, x.C, 1), label)
    w2, s2, b2 = _bottleneck_packed(c2, x, (planes, 9 * planes, 1), label)
    d = _bottleneck_desc(x, t2, planes, head_only=True)
    d.stride = 2
    check(lib.ft_bottleneck_stream_supported(ctypes.byref(d)), "ft_bottleneck_stream_supported")
    key = ("bns_head_stream", x.N, x.H, x.W)
    cached = c1._packed.get(key) if hasattr(c1, "_packed") else None
    if cached is None:
        nbytes = int(lib.ft_bottleneck_stream_weight_bytes(ctypes.byref(d)))
        wstream = torch.empty(nbytes, dtype=torch.uint8, device=x.t.device)
        check(lib.ft_bottleneck_stream_pack(ctypes.byref(d), w1.data_ptr(), w2.data_ptr(), None, wstream.data_ptr(),
                                            current_stream_handle(x.t.device)), "ft_bottleneck_stream_pack")
        torch.cuda.current_stream(x.t.device).synchronize()
        P = planes
        tables = torch.zeros(6 * 2 * P, dtype=torch.float32, device=x.t.device)          # the kernel's descriptor spans six tables
        tables[:4 * P] = torch.cat([s1.flatten()[:P], b1.flatten()[:P], s2.flatten()[:P], b2.flatten()[:P]]).float()
        cached = (wstream, tables)
        if hasattr(c1, "_packed"):
            c1._packed[key] = cached
    wstream, tables = cached
    flops = float(lib.ft_bottleneck_flops(ctypes.byref(d)))
    prog.flops += flops
    prog.fused_records.append((label, len(prog.calls), flops))
    prog.add("ft_bottleneck_stream_fwd", ctypes.byref(d), x.t.data_ptr(), wstream.data_ptr(), tables.data_ptr(), t2.t.data_ptr(),
             keep=(d, x.t, t2.t, wstream, tables))


def bottleneck_entry_fusable(c1: "FusedConv", c2: "FusedConv", sc: "FusedShortcutConv", x: ActView, y: ActView) -> bool:
    """True when ft_bottleneck_fwd(projection) covers a stage's WHOLE entry block (fp16, 64 -> 64 -> 64 -> 256, stride 1,
    projection shortcut K-concatenated with conv3)."""
    if x.t.dtype != torch.float16 or x.rowpacked or (x.N, x.H, x.W) != (y.N, y.H, y.W):
        return False
    if (c1.k, c2.k) != (1, 3) or (c1.stride, c2.stride, c2.pad, sc.stride_d) != (1, 1, 1, 1):
        return False
    if (c1.cin, c1.cout, c2.cin, c2.cout, sc.cin, sc.cin2, sc.cout, y.C) != (x.C, c2.cin, c2.cin, c2.cin, c2.cin, x.C, 4 * c2.cin, 4 * c2.cin):
        return False
    if any(c.transposed or c.tail_cout or c.act != ACT_CODES["relu"] or c._bn is None for c in (c1, c2)) or sc.act != ACT_CODES["relu"]:
        return False
    d = _bottleneck_desc(x, y, c2.cin)
    d.projection = 1
    return _lib.load().ft_bottleneck_supported(ctypes.byref(d)) == 0


def record_bottleneck_entry(prog: Program, c1: "FusedConv", c2: "FusedConv", sc: "FusedShortcutConv", x: ActView, y: ActView,
                            label: str) -> None:
    """A stage's entry block as ONE launch (ft_bottleneck_fwd, projection = 1): conv1 + bn1 + relu -> conv2 + bn2 + relu ->
    relu(bn3(conv3) + bn_d(conv_d(x))); t1 / t2 stay in LDS, the last step is FusedShortcutConv's K-concatenated GEMM with
    the very weights that class packs."""
    lib = _lib.load()
    planes = c2.cin
    w1, s1, b1 = _bottleneck_packed(c1, x, (planes, x.C, 1), label)
    w2, s2, b2 = _bottleneck_packed(c2, x, (planes, 9 * planes, 1), label)
    dc = ConvDesc()
    dc.dtype = sc.code
    dc.N, dc.Hi, dc.Wi, dc.Ho, dc.Wo = x.N, x.H, x.W, x.H, x.W
    dc.Cin, dc.x_cstride, dc.x_coff = sc.cin, act_stride(sc.cin), 0
    dc.Cout, dc.kh, dc.kw, dc.stride, dc.pad, dc.transposed = sc.cout, 1, 1, 1, 0, 0
    dc.y_cstride, dc.y_coff, dc.out_layout = y.cstride, y.coff, FT_LAYOUT_NHWC
    dc.act = sc.act
    dc.x2_cin, dc.x2_hi, dc.x2_wi, dc.x2_cstride, dc.x2_coff, dc.x2_stride = sc.cin2, x.H, x.W, x.cstride, x.coff, 1
    g = conv_geometry(dc)
    if (g.cout_pad, g.kpad, g.cin_pad, g.nphases) != (4 * planes, planes + x.C, planes, 1):
        raise FlowtrackHipError(f"{label}: unexpected K-concatenated layout {(g.cout_pad, g.kpad, g.cin_pad)} for the fused entry block")
    w3, shift3 = sc._packed_for(dc)
    d = _bottleneck_desc(x, y, planes)
    d.projection = 1
    check(lib.ft_bottleneck_supported(ctypes.byref(d)), "ft_bottleneck_supported")
    flops = float(lib.ft_bottleneck_flops(ctypes.byref(d)))
    prog.flops += flops
    prog.fused_records.append((label, len(prog.calls), flops))
    table = torch.cat([t.flatten()[:planes].float() for t in (s1, b1, s2, b2)] +
                      [torch.ones(4 * planes, dtype=torch.float32, device=x.t.device), shift3.flatten()[:4 * planes].float()]).contiguous()
    prog.add("ft_bottleneck_fwd", ctypes.byref(d), x.t.data_ptr(), w1.data_ptr(), w2.data_ptr(), w3.data_ptr(), table.data_ptr(),
             y.t.data_ptr(), keep=(d, x.t, y.t, w1, w2, w3, table))


def bottleneck_cluster_supported(x: ActView, y: ActView, planes: int) -> bool:
    """True when ft_bottleneck_cluster_fwd covers this identity block (fp16, 256 planes, a whole map of <= 192 pixels per
    cluster of four workgroups: layer3 of the ResNets at 256 x 192)."""
    if x.t.dtype != torch.float16 or x.rowpacked or not FUSE_BOTTLENECK_STREAM:
        return False
    d = _bottleneck_desc(x, y, planes)
    return _lib.load().ft_bottleneck_cluster_supported(ctypes.byref(d)) == 0


#: folded operands for the streamed-weights bottleneck kernels (round 6; FT_BNS_FOLD=0 = the scale / shift tables of round 2)
FOLD_BOTTLENECK_STREAM = os.environ.get("FT_BNS_FOLD", "1") != "0"


def _folded_kmajor(conv: "FusedConv"):
    """fp16 [cout][(ky * kw + kx) * cin + ci] weights with the folded BatchNorm SCALE multiplied in (one rounding from the fp32 weights,
    like the plain fp16 weights), and the fp32 shift: the operands of the kernels that add the shift by MFMA."""
    scale, shift = fold_scale_shift(conv.cout, conv.cout, conv._bias, conv._bn, torch.device("cpu"))
    w = conv._weight.detach().float().cpu().permute(0, 2, 3, 1).reshape(conv.cout, -1)
    if scale is not None:
        w = w * scale[:, None]
    return w.half().contiguous(), (shift if shift is not None else torch.zeros(conv.cout))


def _fold_representable(convs) -> bool:
    """The folded form keeps scale * weight and the shift's high part in fp16: a BatchNorm whose folded scale pushes a weight, or
    whose shift lies, beyond the fp16 range (|v| > 65504: never seen in a pose / flow checkpoint, but legal) must take the table form,
    which applies both in fp32."""
    for c in convs:
        w, sh = _folded_kmajor(c)
        if not bool(torch.isfinite(w.float()).all()) or not bool(torch.isfinite(sh.float().half().float()).all()):
            return False
    return True


def shift_pairs(shift: torch.Tensor) -> torch.Tensor:
    """fp32 shifts as (hi, lo) fp16 pairs in one int32 each: hi = fp16(shift) in bits 0-15, lo = fp16(shift - hi) in bits 16-31
    (ft_bottleneck_desc.folded): the two halves are two k-slots of the MFMA slice that adds the shift, so it arrives with ~22 bits."""
    hi = shift.float().half()
    lo = (shift.float() - hi.float()).half()
    return torch.stack([hi, lo], dim=-1).contiguous().view(torch.int32).reshape(-1)


def _bottleneck_stream_operands(c1: "FusedConv", d, x: ActView, planes: int, p1, p2, p3, convs=None):
    """(weight stream, tables) of ft_bottleneck_stream_fwd / ft_bottleneck_cluster_fwd, built once per weight set.  `convs` = the three
    FusedConv layers: the FOLDED operands (d.folded = 1: scales in the weights, tables = shift pairs); None = the table form."""
    lib = _lib.load()
    (w1, s1, b1), (w2, s2, b2), (w3, s3, b3) = p1, p2, p3
    folded = convs is not None
    key = ("bns_stream_folded" if folded else "bns_stream", x.N, x.H, x.W)
    cached = c1._packed.get(key) if hasattr(c1, "_packed") else None
    if cached is None:
        P = planes
        if folded:
            dev = x.t.device
            (w1, sh1), (w2, sh2), (w3, sh3) = (_folded_kmajor(c) for c in convs)
            if (tuple(w1.shape), tuple(w2.shape), tuple(w3.shape)) != ((P, 4 * P), (P, 9 * P), (4 * P, P)):
                raise FlowtrackHipError(f"folded bottleneck weights: unexpected shapes {tuple(w1.shape)} {tuple(w2.shape)} {tuple(w3.shape)}")
            w1, w2, w3 = w1.to(dev), w2.to(dev), w3.to(dev)
            tables = torch.cat([shift_pairs(sh1), shift_pairs(sh2), shift_pairs(sh3)]).contiguous().to(dev)
        else:
            tables = torch.cat([s1.flatten()[:P], b1.flatten()[:P], s2.flatten()[:P], b2.flatten()[:P]] +
                               [t.flatten()[q * P:(q + 1) * P] for q in range(4) for t in (s3, b3)]).float().contiguous()
        nbytes = int(lib.ft_bottleneck_stream_weight_bytes(ctypes.byref(d)))
        wstream = torch.empty(nbytes, dtype=torch.uint8, device=x.t.device)
        check(lib.ft_bottleneck_stream_pack(ctypes.byref(d), w1.data_ptr(), w2.data_ptr(), w3.data_ptr(), wstream.data_ptr(),
                                            current_stream_handle(x.t.device)), "ft_bottleneck_stream_pack")
        torch.cuda.current_stream(x.t.device).synchronize()     # plan-build time: the plan may replay on another stream
        cached = (wstream, tables)
        if hasattr(c1, "_packed"):
            c1._packed[key] = cached
    return cached


def check_cluster_status(prog: Program) -> None:
    """Raise if a cluster-form launch recorded in `prog` ever reported a timed-out hand-off (ft_bottleneck_cluster_fwd sets only the
    status word of its workspace).  Synchronises the program's stream; no-op for plans without a cluster launch."""
    words = prog.__dict__.get("_cluster_status")
    if not words:
        return
    prog.stream.synchronize()
    for shape, (ws, soff) in words.items():
        code = int(ws[soff:soff + 4].view(torch.int32).item())
        if code:
            raise FlowtrackHipError(f"ft_bottleneck_cluster_fwd: a hand-off between the workgroups of a cluster timed out (status {code}, map "
                                    f"{shape}): the block's output is wrong; run without FT_CLUSTER_KERNELS=1 (the strip form)")


def _bottleneck_rstat_weights(c1: "FusedConv", c2: "FusedConv", c3: "FusedConv", device) -> torch.Tensor:
    """Weight buffer of ft_bottleneck_rstat_fwd (layout: include/flowtrack_hip.h), built once per weight set from the fp32 weights:
    each conv's BatchNorm scale folded into its fp16 weights (ONE rounding, like the plain fp16 weights), the shift as a
    (hi, lo) fp16 pair in 16 extra K columns."""
    cached = c1._packed.get(("bnr_weights",))
    if cached is not None:
        return cached
    parts = []
    for conv in (c1, c2, c3):
        scale, shift = fold_scale_shift(conv.cout, conv.cout, conv._bias, conv._bn, torch.device("cpu"))
        w = conv._weight.permute(0, 2, 3, 1).reshape(conv.cout, -1).float()          # [co][(ky * 3 + kx) * cin + ci]
        if scale is not None:
            w = w * scale[:, None]
        sh = shift if shift is not None else torch.zeros(conv.cout)
        extra = torch.zeros((conv.cout, 16), dtype=torch.float16)
        hi = sh.half()
        extra[:, 0] = hi
        extra[:, 1] = (sh - hi.float()).half()
        parts.append(torch.cat([w.half(), extra], dim=1).reshape(-1))
    buf = torch.cat(parts).contiguous().to(device)
    if buf.numel() * 2 != int(_lib.load().ft_bottleneck_rstat_weight_bytes()):
        raise FlowtrackHipError(f"ft_bottleneck_rstat weight buffer: {buf.numel() * 2} bytes built, library expects "
                                f"{int(_lib.load().ft_bottleneck_rstat_weight_bytes())}")
    c1._packed[("bnr_weights",)] = buf
    return buf


def bottleneck_strips_supported(x: ActView, y: ActView, planes: int) -> bool:
    """True when ft_bottleneck_rstat_fwd (register-stationary strips, csrc/bottleneck_rstat.hip) covers this identity block."""
    if x.t.dtype != torch.float16 or x.rowpacked:
        return False
    d = _bottleneck_desc(x, y, planes)
    return _lib.load().ft_bottleneck_rstat_supported(ctypes.byref(d)) == 0


def record_bottleneck(prog: Program, c1: "FusedConv", c2: "FusedConv", c3: "FusedConv", x: ActView, y: ActView,
                      label: str, cluster: bool = False, form: str = "auto", fold: Optional[bool] = None) -> None:
    """conv1 + bn1 + relu -> conv2 + bn2 + relu -> conv3 + bn3 + residual(x) + relu as ONE launch (ft_bottleneck_fwd);
    the packed weights / folded BN are those the three FusedConv layers would use on channel-aligned views.
    `form` (64-plane blocks): "patch" = ft_bottleneck_fwd, "strips" = ft_bottleneck_rstat_fwd, "auto" = strips where the library's
    cost rule takes them.  `fold` (128- / 256-plane blocks): the folded operands of ft_bottleneck_desc.folded (None = FOLD_BOTTLENECK_STREAM,
    i.e. on unless FT_BNS_FOLD=0; the cluster form only has the table form)."""
    lib = _lib.load()
    if y.t.data_ptr() == x.t.data_ptr():
        raise FlowtrackHipError(f"{label}: the fused bottleneck cannot run in place")
    planes = c2.cin

    w1, s1, b1 = _bottleneck_packed(c1, x, (planes, x.C, 1), label)
    w2, s2, b2 = _bottleneck_packed(c2, x, (planes, 9 * planes, 1), label)
    w3, s3, b3 = _bottleneck_packed(c3, x, (x.C, planes, 1), label)
    d = _bottleneck_desc(x, y, planes)
    if lib.ft_bottleneck_supported(ctypes.byref(d)) != 0:
        # 128 / 256 planes: the streamed-weights kernel; its weight stream is built once per weight set by the library
        check(lib.ft_bottleneck_stream_supported(ctypes.byref(d)), "ft_bottleneck_stream_supported")
        flops = float(lib.ft_bottleneck_flops(ctypes.byref(d)))
        fold = (FOLD_BOTTLENECK_STREAM if fold is None else fold) and not cluster and lib.ft_bottleneck_stream_folds(ctypes.byref(d)) == 1
        if fold:
            ok = c1._packed.get(("bns_fold_ok",))
            if ok is None:
                ok = c1._packed[("bns_fold_ok",)] = _fold_representable((c1, c2, c3))
            fold = ok
        d.folded = 1 if fold else 0
        wstream, tables = _bottleneck_stream_operands(c1, d, x, planes, (w1, s1, b1), (w2, s2, b2), (w3, s3, b3),
                                                      convs=(c1, c2, c3) if fold else None)
        prog.flops += flops
        prog.fused_records.append((label, len(prog.calls), flops))
        if cluster:
            # the CLUSTER form (csrc/bottleneck_cluster.hip): four workgroups per image exchange t1 / t2 inside the launch;
            # same weight stream and tables.  Its workspace (exchange buffers + the clusters' monotonic arrival counters, zeroed
            # ONCE) is shared by every cluster block of the plan: the launches of a plan are serialised on its stream.
            check(lib.ft_bottleneck_cluster_supported(ctypes.byref(d)), "ft_bottleneck_cluster_supported")
            nbytes = int(lib.ft_bottleneck_cluster_workspace_bytes(ctypes.byref(d)))
            pool = prog.__dict__.setdefault("_cluster_ws", {})
            ws = pool.get((x.N, x.H, x.W))
            if ws is None or ws.numel() < nbytes:
                ws = pool[(x.N, x.H, x.W)] = torch.zeros(nbytes, dtype=torch.uint8, device=x.t.device)
            prog.add("ft_bottleneck_cluster_fwd", ctypes.byref(d), x.t.data_ptr(), wstream.data_ptr(), tables.data_ptr(), y.t.data_ptr(),
                     ws.data_ptr(), keep=(d, x.t, y.t, wstream, tables, ws))
            # the 32-bit status word of that workspace: non-zero once a hand-off of the cluster form timed out (its output is then
            # wrong); HipModule._run_plan reads the recorded words after the first run / the choice benchmark (check_cluster_status)
            prog.__dict__.setdefault("_cluster_status", {})[(x.N, x.H, x.W)] = (ws, int(lib.ft_bottleneck_cluster_status_offset(ctypes.byref(d))))
            return
        prog.add("ft_bottleneck_stream_fwd", ctypes.byref(d), x.t.data_ptr(), wstream.data_ptr(), tables.data_ptr(), y.t.data_ptr(),
                 keep=(d, x.t, y.t, wstream, tables))
        return
    flops = float(lib.ft_bottleneck_flops(ctypes.byref(d)))
    prog.flops += flops
    prog.fused_records.append((label, len(prog.calls), flops))
    if form == "strips":
        check(lib.ft_bottleneck_rstat_supported(ctypes.byref(d)), "ft_bottleneck_rstat_supported")
    if form != "patch" and lib.ft_bottleneck_rstat_supported(ctypes.byref(d)) == 0:
        # maps up to 62 wide with enough pixels per CU: the register-stationary strip form (csrc/bottleneck_rstat.hip)
        wpack = _bottleneck_rstat_weights(c1, c2, c3, x.t.device)
        prog.add("ft_bottleneck_rstat_fwd", ctypes.byref(d), x.t.data_ptr(), wpack.data_ptr(), y.t.data_ptr(), keep=(d, x.t, y.t, wpack))
        return
    table = torch.cat([t.flatten()[:n] for t, n in ((s1, planes), (b1, planes), (s2, planes), (b2, planes), (s3, x.C), (b3, x.C))]).contiguous()
    prog.add("ft_bottleneck_fwd", ctypes.byref(d), x.t.data_ptr(), w1.data_ptr(), w2.data_ptr(), w3.data_ptr(), table.data_ptr(),
             y.t.data_ptr(), keep=(d, x.t, y.t, w1, w2, w3, table))


# --------------------------------------------------------------------------------------------
# thin wrappers for the remaining entry points (recorded into a Program)
# --------------------------------------------------------------------------------------------
def record_pack_input(prog: Program, x_nchw: torch.Tensor, y: ActView) -> None:
    N, C, H, W = x_nchw.shape
    prog.add("ft_pack_nchw_to_nhwc", x_nchw.data_ptr(), y.t.data_ptr(), N, C, H, W, y.cstride, y.lpad, y.wpitch,
             _lib.dtype_code(y.t.dtype), keep=(x_nchw, y.t))


def record_maxpool(prog: Program, x: ActView, y: ActView) -> None:
    if x.coff or y.coff or x.cstride != x.C or y.cstride != y.C:
        raise FlowtrackHipError("maxpool works on dense NHWC buffers")
    prog.add("ft_maxpool3x3s2_fwd", x.t.data_ptr(), y.t.data_ptr(), x.N, x.H, x.W, x.C, _lib.dtype_code(x.t.dtype),
             keep=(x.t, y.t))


def record_upsample4x(prog: Program, x: torch.Tensor, y: torch.Tensor, mul: float) -> None:
    N, C, h, w = x.shape
    prog.add("ft_upsample_bilinear4x", x.data_ptr(), y.data_ptr(), N, C, h, w, ctypes.c_float(mul), keep=(x, y))


def current_stream_handle(device=None) -> ctypes.c_void_p:
    """Handle of torch's current stream on `device` (default: the current device)."""
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def heatmap_max_preds(heatmaps: torch.Tensor, adjust_coords: bool):
    """Device part of max_preds/final_preds (lib/pose/utils/evaluation.py:11-33).
    Returns (idx int32 [N,K], scores fp32 [N,K,1], coords fp32 [N,K,2]) on the device."""
    require_gpu(heatmaps.device)
    lib = _lib.load()
    hm = heatmaps.contiguous().float()
    N, K, H, W = hm.shape
    idx = torch.empty((N, K), dtype=torch.int32, device=hm.device)
    score = torch.empty((N, K, 1), dtype=torch.float32, device=hm.device)
    coords = torch.empty((N, K, 2), dtype=torch.float32, device=hm.device)
    check(lib.ft_heatmap_max_preds(hm.data_ptr(), N, K, H, W, int(bool(adjust_coords)), idx.data_ptr(),
                                   score.data_ptr(), coords.data_ptr(), current_stream_handle()),
          "ft_heatmap_max_preds")
    return idx, score, coords


def bn_batch_stats(x: ActView):
    """Per-channel (mean, biased var) over N*H*W of an NHWC view — nn.BatchNorm2d's training-mode reduction."""
    require_gpu(x.t.device)
    if x.coff or x.rowpacked:
        raise FlowtrackHipError("bn_batch_stats works on plain NHWC buffers (coff == 0)")
    lib = _lib.load()
    C = x.C
    ws = torch.empty(2 * C, dtype=torch.float32, device=x.t.device)
    mean = torch.empty(C, dtype=torch.float32, device=x.t.device)
    var = torch.empty(C, dtype=torch.float32, device=x.t.device)
    check(lib.ft_bn_batch_stats(x.t.data_ptr(), x.N, x.H, x.W, C, x.cstride, _lib.dtype_code(x.t.dtype), ws.data_ptr(),
                                mean.data_ptr(), var.data_ptr(), current_stream_handle()), "ft_bn_batch_stats")
    return mean, var
