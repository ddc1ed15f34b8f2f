python -m pytest tests -q -m gpu -x 2>&1 | tail -25 > gpurun_out/r05_t5.log
cat gpurun_out/r05_t5.log
