#!/bin/bash
# rocprofv3 --pmc passes over the front-end kernels (tools/bench_frontend.py): instruction mix, LDS and wait cycles.
#   bash tools/pmc_frontend.sh <tag>   -> gpurun_out/<tag>_pmcfe.txt (per-launch averages)
set -eu
TAG=${1:-fe}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
	   "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT" \
	   "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
	i=$((i + 1))
	timeout 300 rocprofv3 --output-format csv --kernel-include-regex "channelize_kernel|resample_kernel|frontend_fused_kernel|frontend_wave_kernel" --pmc $SET -d $O/pmcfe$i -o $TAG -- python3 $R/tools/bench_frontend.py > $O/${TAG}_pmcfe$i.log 2>&1 || true
done
python3 - "$O" "$TAG" > $O/${TAG}_pmcfe.txt <<'PY'
import csv, glob, sys, collections
O, tag = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(f"{O}/pmcfe*/**/{tag}_counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"].split("(")[0][:24], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(f"{k[0]:26s} {k[1]:30s} {sum(v) / len(v):16.0f}  (n={len(v)})")
PY
rm -rf $O/pmcfe1 $O/pmcfe2 $O/pmcfe3
cat $O/${TAG}_pmcfe.txt
