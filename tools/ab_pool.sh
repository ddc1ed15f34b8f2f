mkdir -p gpurun_out/r03
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4
for i in 1 2 3; do
  for L in prev pool; do
    v=$(TRXHIP_LIB=$PWD/osmo_trx_amd/lib/libtrxhip_$L.so timeout 300 python3 bench.py --main-only --steps 40 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])")
    echo "$L $v"
  done
  v=$(TRXHIP_NO_POOL=1 TRXHIP_LIB=$PWD/osmo_trx_amd/lib/libtrxhip_pool.so timeout 300 python3 bench.py --main-only --steps 40 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])")
  echo "pool_lib_static $v"
done
