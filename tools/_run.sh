L=osmo_trx_amd/lib
bash tools/ab_multi.sh 2 30 $L/libtrxhip_full.so $L/libtrxhip_nodemod.so $L/libtrxhip_nopeak.so $L/libtrxhip_nodet.so $L/libtrxhip_nofir.so $L/libtrxhip_noci.so $L/libtrxhip_notail.so > gpurun_out/r05_abl.log 2>&1
cat gpurun_out/r05_abl.log
