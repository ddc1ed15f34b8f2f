for D in "" "0,0" "0,0,0,0"; do
  export TRXHIP_DEVICES=$D; [ -z "$D" ] && unset TRXHIP_DEVICES
  echo "TRXHIP_DEVICES=$D"
  python tools/bench_gather.py 16 4096 200 1 256 4 4096
  python tools/bench_gather.py 16 4096 200 1 256 8 4096
  python tools/bench_gather.py 16 1024 200 1 256 8 4096
done > gpurun_out/r05_gather0.log 2>&1
nproc >> gpurun_out/r05_gather0.log
cat gpurun_out/r05_gather0.log
