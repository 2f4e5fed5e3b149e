"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's algorithms for the FlowTrack pose / flow hot paths.  Only
tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this package, and only
as the checker — the product path (flowtrack/pytorch_amd) never routes through it.

Contents
  ops_ref.py        ctypes binding of flow_ops_ref.c (plain C, follows the reference .cu kernels) and
                    independent numpy formulations of the same operators
  pose_ref.py       functional torch-CPU fp32 graph of DeconvResnet (the reference's own arithmetic
                    for the stock layers is torch CPU: SURVEY §8(c) "third-party arithmetic")
  flow_ref.py       functional torch-CPU fp32 graphs of FlowNet2S / FlowNet2C / FlowNet2CS
  keypoints_ref.py  numpy restatement of max_preds / final_preds / OKS with torch-0.4 semantics

Pinning: pose_ref / flow_ref(FlowNet2S) / keypoints_ref are checked against the imported reference
classes by tests/golden/make_golden.py (run in the build container only; the vectors it wrote are
committed under tests/golden/).  The three CUDA-only ops have no runnable reference => parity
UNPINNED by the reference for those (see flow_ops_ref.c header).
"""
