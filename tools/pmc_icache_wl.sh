#!/bin/bash
# Instruction-cache counters of the hot kernel per workload:  bash tools/pmc_icache_wl.sh  -> per-burst values for normal / rach / mixed
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for W in normal rach mixed; do
	rm -rf $O/icw
	WORKLOAD=$W timeout 300 rocprofv3 --output-format csv --kernel-include-regex burst_pull4 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY -d $O/icw -o ic -- python3 $R/tools/pmc_mixed.py > $O/icw.log 2>&1
	python3 - "$O" "$W" <<'PY'
import csv, glob, sys, collections
O, W = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(f"{O}/icw/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(W, " ".join(f"{k}={sum(v)/len(v)/(1<<20):.2f}" for k, v in sorted(acc.items())))
PY
done
rm -rf $O/icw
