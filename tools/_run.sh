L=osmo_trx_amd/lib
bash tools/pmc_insts.sh $L/libtrxhip_late.so $L/libtrxhip_early.so > gpurun_out/r05_pmc9.log 2>&1
bash tools/ab_multi.sh 3 30 $L/libtrxhip_late.so $L/libtrxhip_early.so $L/libtrxhip_d6.so $L/libtrxhip_d2.so > gpurun_out/r05_ab9.log 2>&1
cat gpurun_out/r05_pmc9.log gpurun_out/r05_ab9.log
