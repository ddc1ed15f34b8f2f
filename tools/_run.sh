set -x
L=osmo_trx_amd/lib
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u | tr '\n' ' ' > gpurun_out/r05_counters.txt
bash tools/pmc_insts.sh $L/libtrxhip_base.so $L/libtrxhip_f0.so $L/libtrxhip_f1.so $L/libtrxhip_nofw.so > gpurun_out/r05_pmc5.log 2>&1
bash tools/ab_multi.sh 3 30 $L/libtrxhip_base.so $L/libtrxhip_f0.so $L/libtrxhip_f1.so $L/libtrxhip_nofw.so > gpurun_out/r05_ab5.log 2>&1
cat gpurun_out/r05_counters.txt gpurun_out/r05_pmc5.log gpurun_out/r05_ab5.log
