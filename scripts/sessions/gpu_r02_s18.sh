#!/bin/bash
# round-2 GPU session 18: does two-workgroups-per-CU pay for the 128-channel layers?  proxy: the 4-wave 8-row x 64-channel tile
# (79 KiB LDS: two per CU, but every 64-channel column re-stages the input) against the shipped 8-wave 8-row x 128-channel tile
set -o pipefail
L=mu-diff_amd/mudiff_hip/libmudiff_hip.so
for i in 1 2; do
AB_SHAPES=2,4,13,5,6 python scripts/ab_conv.py 16 5 $L 2>/dev/null | sed 's/^/shipped 8x2    : /'
AB_SHAPES=2,4,13,5,6 MUD_CONV_MT=2 python scripts/ab_conv.py 16 5 $L 2>/dev/null | sed 's/^/4-wave 2 per CU: /'
done
