#!/bin/bash
# Two rocprofv3 --pmc passes over the timed leg of bench.py: instruction mix and wave-cycle split of the hot kernels
# (normal-burst kernel + the general kernel behind it), per burst of the batch.
#   bash tools/pmc_nb.sh <tag>   -> gpurun_out/<tag>_pmcnb.txt
set -eu
TAG=${1:-q}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" \
	   "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
	   "SQ_INSTS_VALU SQ_IFETCH SQ_INST_LEVEL_LDS SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_MISC"; do
	i=$((i + 1))
	timeout 300 rocprofv3 --output-format csv --kernel-include-regex "pull4" --pmc $SET -d $O/pmcnb$i -o $TAG -- python3 $R/bench.py --steps 2 --warmup 1 --main-only > $O/${TAG}_pmcnb$i.log 2>&1 || true
done
python3 - "$O" "$TAG" > $O/${TAG}_pmcnb.txt <<'PY'
import csv, glob, sys, collections
O, tag = sys.argv[1], sys.argv[2]
n = 1 << 20
for f in sorted(glob.glob(f"{O}/pmcnb*/**/{tag}_counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = "nb" if "nb_pull4" in k else ("list" if "true, true>" in k else "general")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kn, d in acc.items():
        for k, v in d.items():
            print(f"{kn:8s} {k:26s} {sum(v) / len(v) / n:10.2f}")
PY
rm -rf $O/pmcnb1 $O/pmcnb2 $O/pmcnb3
cat $O/${TAG}_pmcnb.txt
