#!/bin/bash
# round-2 GPU session 19: 128 / 256-channel layers on a 4-row x 128-channel 8-wave tile (81.8 KB LDS, <= 128 VGPRs: two workgroups per CU)
set -o pipefail
L=mu-diff_amd/mudiff_hip/libmudiff_hip.so
for i in 1 2; do
AB_SHAPES=2,4,13,5,6,8,12 python scripts/ab_conv.py 16 5 $L 2>/dev/null | sed 's/^/shipped 8x2 (8 rows)  : /'
AB_SHAPES=2,4,13,5,6,8,12 MUD_CONV_MT=42 python scripts/ab_conv.py 16 5 $L 2>/dev/null | sed 's/^/4 rows, two per CU    : /'
done
