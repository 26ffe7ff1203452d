for args in "32 512 64 volume" "32 500 64 random" "32 320 32 graded" "40 480 64 volume"; do
  NDMPS_TRD_SYM=1 timeout -k 10 120 python tools/band_probe.py $args 2>&1 | grep "^band" || exit 1
done
timeout -k 10 120 python tools/band_probe.py 32 512 64 volume 2>&1 | grep "^band"
