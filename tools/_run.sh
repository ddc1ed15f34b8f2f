python -m pytest tests/test_gpu_sharded.py -q -m gpu -x --durations=6 2>&1 | tail -25 > gpurun_out/r05_t8.log
cat gpurun_out/r05_t8.log
