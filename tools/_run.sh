python -m pytest tests/test_gpu_trxd_hostpipe.py tests/test_gpu_host_shim.py -q -m gpu -x -k "use_va or batch_va or adapter" 2>&1 | tail -25 > gpurun_out/r05_t9.log
cat gpurun_out/r05_t9.log
