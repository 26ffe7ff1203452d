for xcd in 1 2; do
  echo "xcd mode $xcd"
  for args in "1 512 64 volume" "2 512 64 volume" "4 512 64 volume" "8 512 64 volume" "12 512 64 volume" "20 512 64 volume" "32 320 32 graded" "3 200 32 graded"; do
    NDMPS_TRD_XCD=$xcd timeout -k 10 120 python tools/band_probe.py $args 2>&1 | grep "^band" || exit 1
  done
done
