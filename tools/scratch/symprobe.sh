L=img-compression-mps_amd
cp $L/libndmps_hip.so /tmp/lib158.so
for v in 158 128; do
  if [ $v = 128 ]; then cp $L/libndmps_hip_128.so $L/libndmps_hip.so; fi
  echo "tail with $v registers"
  for args in "1 512 64 volume" "32 512 64 volume" "3 100 16 random"; do
    timeout -k 10 120 python tools/band_probe.py $args 2>&1 | grep "^band" || exit 1
  done
  rm -f gpurun_out/bs_$v.txt
  bash tools/batch_sweep.sh gpurun_out/bs_$v.txt "64 2" "64 2" "8 1" "1 1" > /dev/null; cat gpurun_out/bs_$v.txt
done
echo "tail in LDS"
cp /tmp/lib158.so $L/libndmps_hip.so
rm -f gpurun_out/bs_lds.txt
NDMPS_TRD_TAIL_LDS=1 bash tools/batch_sweep.sh gpurun_out/bs_lds.txt "64 2" "64 2" "8 1" "1 1" > /dev/null; cat gpurun_out/bs_lds.txt
