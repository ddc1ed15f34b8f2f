L=osmo_trx_amd/lib
bash tools/ab_legs.sh 2 $L/libtrxhip_g0.so $L/libtrxhip_g1.so > gpurun_out/r05_ab15.log 2>&1
python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -2 >> gpurun_out/r05_ab15.log
cat gpurun_out/r05_ab15.log
