L=osmo_trx_amd/lib
for i in 1 2 3; do for l in nt0 nt1; do TRXHIP_LIB=$PWD/$L/libtrxhip_$l.so python3 bench.py --main-only --steps 30 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['config']['sustained']; print('$l', d['value'], d['roofline']['kernel_ms'], s['mbursts_per_s_all_gpus'], s['sclk_mhz_under_load'])"; done; done > gpurun_out/r05_ab16.log 2>&1
cat gpurun_out/r05_ab16.log
