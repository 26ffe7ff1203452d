for xcd in 0 1; do for sym in 0 1; do
  echo "xcd $xcd sym $sym"
  for args in "32 512 64 volume" "8 512 64 volume" "1 512 64 volume"; do
    NDMPS_TRD_XCD=$xcd NDMPS_TRD_SYM=$sym timeout -k 10 120 python tools/band_probe.py $args 2>&1 | grep "^band" || exit 1
  done
done; done
