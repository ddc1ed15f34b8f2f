cd /tmp && export TMPDIR=/tmp
export NOHINT=1 TRXHIP_NO_BACKOFF=1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_mix -o mix -- python3 $GRAFT_REPO_ROOT/tools/trace_mixed.py 30 > /dev/null 2>&1
head -8 $GRAFT_REPO_ROOT/gpurun_out/prof_mix/mix_kernel_stats.csv | cut -c1-200
find $GRAFT_REPO_ROOT/gpurun_out/prof_mix -name "*kernel_trace.csv" -delete
