for l in 0 5000 10000 20000 40000; do
  export ODHIP_LDS4=$l ODHIP_LDS8=$l
  echo "== dummy LDS $l"
  bash tools/prof_quick.sh 2>&1 | grep -E "k_prep_corner"
done
