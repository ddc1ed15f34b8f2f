#!/bin/bash
# rocprofv3 evidence for every kernel other than the timed one: tools/bench_aux.py under --kernel-trace --stats and
# under separate --pmc FETCH_SIZE / WRITE_SIZE passes.   bash tools/run_profiles_aux.sh r02   (GPU box, repo root)
# -> gpurun_out/<tag>_aux.jsonl (the program's own HIP-event timings), prof_aux_stats/, prof_aux_fetch/, prof_aux_write/
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export REPS=${REPS:-5}
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_aux_stats -o ${TAG}_aux -- python3 $R/tools/bench_aux.py > $O/${TAG}_aux.jsonl 2> $O/${TAG}_aux.err
export REPS=1
for C in FETCH_SIZE WRITE_SIZE; do
	d=$O/prof_aux_$(echo $C | tr A-Z a-z | sed 's/_size//')
	timeout 900 rocprofv3 --output-format csv --pmc $C -d $d -o ${TAG}_aux -- python3 $R/tools/bench_aux.py > $O/${TAG}_aux_$C.log 2>&1
done
find $O/prof_aux_stats $O/prof_aux_fetch $O/prof_aux_write -name "*kernel_trace.csv" -delete
find $O/prof_aux_stats $O/prof_aux_fetch $O/prof_aux_write -name "*agent_info.csv" -delete
# counter files: keep the library's kernels only (drop torch's fill / copy / rng kernels)
for f in $(find $O/prof_aux_fetch $O/prof_aux_write -name "*counter_collection.csv"); do
	(head -1 $f; grep -E "burst_pull|pack_trxd|va_demod|channelize|resample|frontend_fused|convolve_kernel|convolve_lds_kernel|convert_short|delay_vector|energy_detect|vector_slicer|sch_detect|save_" $f) > $f.tmp && mv $f.tmp $f
done
du -sh $O/prof_aux_*
cat $O/${TAG}_aux.jsonl
