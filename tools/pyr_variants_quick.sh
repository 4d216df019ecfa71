#!/bin/bash
# (round 5) the A/B switches below exist in the experiments build of the library only
export ODHIP_LIB=${ODHIP_LIB:-$(cd "$(dirname "$0")/.." && pwd)/daala_amd/lib/libdaalahip_exp.so}
# Development: the luma pyramid's kernel shapes side by side on one box (tools/pyr_stalls.py child: all
# levels stored to separate torch allocations / nothing stored / all levels into one allocation).
for v in 2 3 6 0; do echo "ODHIP_PYR_VARIANT=$v"; ODHIP_PYR_VARIANT=$v ODHIP_PYR_LDS_PAD=0 python tools/pyr_stalls.py child 2>/dev/null | grep -E "all five|no level|one allocation"; done
echo "ODHIP_PYRAMID_X1=1 (one superblock per 256-thread workgroup)"; ODHIP_PYRAMID_X1=1 ODHIP_PYR_LDS_PAD=0 python tools/pyr_stalls.py child 2>/dev/null | grep -E "all five|no level|one allocation"
