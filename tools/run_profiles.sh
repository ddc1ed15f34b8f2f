#!/bin/bash
# Profile recipe of profiles/<tag>_*: run on the GPU box from the repo root:  bash tools/run_profiles.sh r01d
# (kernel-trace/stats and each --pmc set are separate rocprofv3 runs; counters are restricted to the hot kernel)
TAG=${1:-r01x}
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o $TAG -- python3 $R/bench.py --steps 10 --warmup 3 --main-only > $O/${TAG}_bench.log 2>&1
grep '^{"metric"' $O/${TAG}_bench.log | tail -1 > $O/${TAG}_bench.json
for C in FETCH_SIZE WRITE_SIZE; do
	d=$O/prof_$(echo $C | tr A-Z a-z | sed 's/_size//')
	timeout 400 rocprofv3 --output-format csv --kernel-include-regex pull4 --pmc $C -d $d -o $TAG -- python3 $R/bench.py --steps 3 --warmup 1 --main-only > $O/${TAG}_$C.log 2>&1
done
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH GRBM_GUI_ACTIVE" \
	   "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
	   "SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU"; do
	i=$((i + 1))
	timeout 400 rocprofv3 --output-format csv --kernel-include-regex pull4 --pmc $SET -d $O/prof_sq$i -o $TAG -- python3 $R/bench.py --steps 2 --warmup 1 --main-only > $O/${TAG}_sq$i.log 2>&1
done
# keep the merge-back small: drop traces, keep stats and the hot kernel's counter rows
find $O -name "*kernel_trace.csv" -delete
find $O -name "*agent_info.csv" -delete
for f in $(find $O -name "*counter_collection.csv"); do
	(head -1 $f; grep pull4 $f) > $f.tmp && mv $f.tmp $f
done
du -sh $O
