for d in 0 1 2 4 7; do
  export ODHIP_PREP_DEBUG=$d
  echo "== debug $d"
  bash tools/prof_quick.sh 2>&1 | grep -E "k_prep_lane|k_prep_wide"
done
