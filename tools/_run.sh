L=osmo_trx_amd/lib
bash tools/pmc_insts.sh $L/libtrxhip_ge.so $L/libtrxhip_gl.so $L/libtrxhip_nosel.so > gpurun_out/r05_pmc8.log 2>&1
bash tools/ab_multi.sh 3 30 $L/libtrxhip_ge.so $L/libtrxhip_gl.so $L/libtrxhip_nosel.so > gpurun_out/r05_ab8.log 2>&1
cat gpurun_out/r05_pmc8.log gpurun_out/r05_ab8.log
