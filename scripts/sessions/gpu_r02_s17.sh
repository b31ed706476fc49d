#!/bin/bash
# round-2 GPU session 17: 64-output-channel layers: 8 waves x 1 row (two workgroups per CU, four waves per SIMD) vs the shipped tiles
set -o pipefail
L=mu-diff_amd/mudiff_hip/libmudiff_hip.so
for i in 1 2; do
AB_SHAPES=3,10,11,14,9 python scripts/ab_conv.py 16 5 $L 2>/dev/null | sed 's/^/shipped : /'
AB_SHAPES=3,10,11,14,9 MUD_CONV_8X1R=1 python scripts/ab_conv.py 16 5 $L 2>/dev/null | sed 's/^/8x1row  : /'
done
