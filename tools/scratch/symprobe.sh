for per in 2 3 4; do
  echo "narrow workgroups per CU $per"
  for args in "10 512 64 volume" "12 512 64 volume" "16 512 64 volume"; do
    NDMPS_TRD_NARROW_PER_CU=$per timeout -k 10 120 python tools/band_probe.py $args 2>&1 | grep "^band" || exit 1
  done
done
