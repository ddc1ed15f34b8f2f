L=osmo_trx_amd/lib
python -m pytest tests/test_gpu_parity.py tests/test_gpu_trxd_hostpipe.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/r05_t12.log
for i in 1 2; do python tools/bench_exact.py 2>&1 | tail -2; python tools/bench_1sps.py 2>&1 | tail -1; done >> gpurun_out/r05_t12.log 2>&1
cat gpurun_out/r05_t12.log
