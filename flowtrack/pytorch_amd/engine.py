"""Base class of the HIP-backed models: plan cache, packed-weight invalidation, stream hand-off.

A *plan* is the static launch sequence (hip_ops.Program) of one network for one input shape, with
all activation buffers pre-allocated in HBM, optionally captured into a HIP graph.  The module's
parameters stay in the reference's layout; plans hold the packed copies.
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import _lib, hip_ops
from .hip_ops import FlowtrackHipError, Program, require_gpu


class HipModule(nn.Module):
    #: set to torch.float16 / torch.float32 to override the dtype implied by the parameters
    compute_dtype: Optional[torch.dtype] = None
    #: capture each plan into a HIP graph after its first (eager, validating) run
    use_graph: bool = True

    def __init__(self):
        super().__init__()
        self._plans: Dict[tuple, object] = {}
        self._layers: Dict[tuple, object] = {}   # (label, dtype, device) -> FusedConv: packed weights shared by all plans
        self._stream: Optional[torch.cuda.Stream] = None
        self._fingerprint: Optional[int] = None     # sum of the version counters of every parameter / buffer the plans were built from
        self._fp_tensors: Optional[list] = None     # those tensors (cached: walking the module tree costs 0.3 ms)

    # -- parameter changes invalidate packed weights ------------------------------------------
    def _invalidate(self):
        self._plans = {}
        self._layers = {}
        self._fp_tensors = None

    def refresh(self) -> None:
        """Drop every packed weight set and captured plan; the next forward rebuilds them from the current parameters.
        Needed only after edits the version check of _check_fingerprint() cannot see (writes through `p.data`)."""
        self._invalidate()

    def fused(self, label: str, weight, *, dtype, device, **kw):
        """FusedConv for `label`, built once per (dtype, device) and shared by every plan (input shape) of this
        module — weight packing / BN folding is per layer, not per plan."""
        from .hip_ops import FusedConv
        key = (label, dtype, str(device))
        layer = self._layers.get(key)
        if layer is None:
            layer = self._layers[key] = FusedConv(weight, dtype=dtype, device=device, label=label, **kw)
        return layer

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._invalidate()
        return out

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self._invalidate()
        return out

    # -- helpers ------------------------------------------------------------------------------
    def _param_device_dtype(self):
        p = next(self.parameters())
        return p.device, p.dtype

    def _check_fingerprint(self) -> None:
        """Packed weights / captured graphs are copies of the parameters: rebuild them when any parameter or buffer of
        this module tree changed since they were made — a load_state_dict through a child or a wrapper, in-place edits
        under no_grad such as nn.init (all bump the tensor's version counter).  Writes through `p.data` bump no counter
        and a re-assigned Parameter object is not in the cached list: call refresh() after those.  The check is a sum
        over a cached tensor list (~20 us of host time for a ResNet-101; walking named_parameters() every call cost 0.3 ms,
        more than a small-batch step)."""
        tensors = self._fp_tensors
        if tensors is None:
            tensors = self._fp_tensors = list(self.parameters()) + list(self.buffers())
        fp = 0
        for t in tensors:
            fp += t._version
        if fp != self._fingerprint:
            if self._fingerprint is not None and (self._plans or self._layers):
                self._invalidate()
            self._fingerprint = fp

    def _resolve(self):
        self._check_fingerprint()
        device, pdtype = self._param_device_dtype()
        require_gpu(device)
        dtype = self.compute_dtype or pdtype
        if dtype not in (torch.float16, torch.float32):
            raise FlowtrackHipError(f"compute dtype {dtype} unsupported (fp16 or fp32)")
        _lib.load()
        return device, dtype

    def _side_stream(self, device) -> torch.cuda.Stream:
        if self._stream is None or self._stream.device != device:
            self._stream = torch.cuda.Stream(device=device)
        return self._stream

    def _check_eval(self):
        if self.training:
            raise FlowtrackHipError(
                f"{type(self).__name__}: only the inference path is implemented on HIP — call .eval() "
                "(training-mode BatchNorm / multi-scale outputs are out of scope, SURVEY §8)")

    def _run_plan(self, prog: Program, first: bool) -> None:
        """First call: eager run (+ tile benchmark + graph capture) on the plan's private stream, ordered after / before
        the caller's current stream.  Later calls: the captured graph is launched directly on the caller's current
        stream — no cross-stream event waits (each cost 10-20 us of idle GPU per step in the kernel trace)."""
        cur = torch.cuda.current_stream(prog.stream.device)
        if first or prog.graph_exec is None:
            prog.stream.wait_stream(cur)
            if first:
                if hip_ops.benchmark:
                    prog.resolve_choices(cached_only=True)   # picks of an earlier plan / the FT_TILE_CACHE file: nothing to time
                prog.run_eager()          # surfaces argument errors before any capture
                if hip_ops.benchmark:     # cudnn.benchmark counterpart: pick each conv's tile variant in situ (every
                    prog.stream.synchronize()   # pass is a complete, valid forward: the outputs stay those of this input)
                    verbose = bool(os.environ.get("FT_CONV_BENCHMARK_VERBOSE"))
                    prog.tune_tiles(verbose=verbose)
                    prog.tune_choices(verbose=verbose)   # fused / unfused blocks, direct / implicit-GEMM 1x1s: keep the faster
                    prog.run_eager()      # outputs of this call come from the chosen variants (split-K variants sum in
                                          # another order: the first call must be bit-identical to the replays)
                else:
                    prog.resolve_choices()    # no benchmark: the recorder's first option of every alternative
                hip_ops.check_cluster_status(prog)    # (only plans recorded with FT_CLUSTER_KERNELS=1 hold such launches)
                if self.use_graph:
                    prog.stream.synchronize()
                    prog.capture()
            else:
                prog.run()
            cur.wait_stream(prog.stream)
        else:
            prog.run(cur)
