#!/bin/bash
# round-2 GPU session 13: 64-channel layers at 256^2: 16-row 8-wave tile (one workgroup per CU) vs 8-row 4-wave tile (two per CU)
set -o pipefail
mkdir -p gpurun_out/r02
L=mu-diff_amd/mudiff_hip/libmudiff_hip.so
for i in 1 2; do
AB_SHAPES=10,11,14 python scripts/ab_conv.py 16 5 $L 2>/dev/null | sed 's/^/16x1 : /'
AB_SHAPES=10,11,14 MUD_CONV_NO16=1 python scripts/ab_conv.py 16 5 $L 2>/dev/null | sed 's/^/MT2  : /'
done
AB_SHAPES=10,11,14 python scripts/ab_conv.py 4 5 $L 2>/dev/null | sed 's/^/B4 16x1 : /'
AB_SHAPES=10,11,14 MUD_CONV_NO16=1 python scripts/ab_conv.py 4 5 $L 2>/dev/null | sed 's/^/B4 MT2  : /'
