#!/bin/bash
# A/B of two library builds on one box: tools/ab.sh <script> [args]  (ODHIP_LIB=daala_amd/lib_ab/libdaalahip.so is B)
for rep in 1 2; do
  echo "A (default build)"; python "$@" 2>/dev/null | head -${AB_LINES:-3}
  echo "B (lib_ab)"; ODHIP_LIB=$GRAFT_REPO_ROOT/daala_amd/lib_ab/libdaalahip.so python "$@" 2>/dev/null | head -${AB_LINES:-3}
done
