set -x
L=osmo_trx_amd/lib
for l in t24 w6n24 w6n26 w5n26 w4n28; do TRXHIP_LIB=$PWD/$L/libtrxhip_$l.so python tools/fused_taps_report.py; done > gpurun_out/r05_taps2.log 2>&1
bash tools/ab_multi.sh 3 30 $L/libtrxhip_t24.so $L/libtrxhip_w6n24.so $L/libtrxhip_w6n26.so $L/libtrxhip_w5n26.so $L/libtrxhip_w4n28.so > gpurun_out/r05_ab7.log 2>&1
grep -v "^+\|amdgpu.ids" gpurun_out/r05_taps2.log; cat gpurun_out/r05_ab7.log
