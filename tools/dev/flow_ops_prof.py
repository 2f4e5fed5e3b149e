"""The custom operators of the FLOW path at BASELINE configs[3] shapes, alone: correlation [16,256,48,64] x2 -> 441 channels,
warp + diff + norm + concat, Resample2d, ChannelNorm at [16,*,384,512] (bench.flow_op_rooflines: one eager pass + 20 timed).
Run under rocprofv3 (tools/dev/prof_round.sh) for the kernel trace and the FETCH / WRITE passes; --json prints the rooflines."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
rows = bench.flow_op_rooflines(torch.device("cuda:0"))
if "--json" in sys.argv:
    print(json.dumps(rows, indent=1))
else:
    for r in rows:
        print(r)
