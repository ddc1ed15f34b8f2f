L=osmo_trx_amd/lib
for r in 1 2; do for l in felds fesgpr; do echo "== $l"; TRXHIP_LIB=$PWD/$L/libtrxhip_$l.so python tools/bench_frontend.py 2>&1 | grep -i "fused\|front"; done; done > gpurun_out/r05_fe1.log 2>&1
python -m pytest tests/test_gpu_aux_kernels.py -q -m gpu -x 2>&1 | tail -3 >> gpurun_out/r05_fe1.log
cat gpurun_out/r05_fe1.log
