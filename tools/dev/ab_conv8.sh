#!/bin/bash
# Dev: ablation of conv_igemm8_kernel through FT_CONV_DBG (4 = no epilogue, 64 = no pixel loads, 256 = no weight loads)
cd "$(dirname "$0")/../.."
for dbg in 0 4 68 260 324; do
  echo "== FT_CONV_DBG=$dbg"
  FT_CONV_DBG=$dbg timeout 300 python tools/dev/conv8_bench.py all "deconv6_256,deconv3_256,deconv0_2048,f.conv3_5x5,f.conv3_1" 2>&1 | grep -v amdgpu.ids | sed 's/heuristic.*| 8-phase/| 8-phase/'
done
