#!/bin/bash
# One rocprofv3 --pmc pass over the timed leg of bench.py: instruction mix and wave-cycle split of the hot kernel.
#   bash tools/pmc_quick.sh <tag>   -> gpurun_out/<tag>_pmcq.txt (per-burst values)
set -eu
TAG=${1:-q}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
	   "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT"; do
	i=$((i + 1))
	timeout 300 rocprofv3 --output-format csv --kernel-include-regex burst_pull4 --pmc $SET -d $O/pmcq$i -o $TAG -- python3 $R/bench.py --steps 2 --warmup 1 --main-only > $O/${TAG}_pmcq$i.log 2>&1
done
python3 - "$O" "$TAG" > $O/${TAG}_pmcq.txt <<'PY'
import csv, glob, sys, collections
O, tag = sys.argv[1], sys.argv[2]
n = 1 << 20
for f in sorted(glob.glob(f"{O}/pmcq*/**/{tag}_counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "burst_pull4_kernel<false, false, true>" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:26s} {sum(v) / len(v) / n:10.2f}")
PY
rm -rf $O/pmcq1 $O/pmcq2
cat $O/${TAG}_pmcq.txt
