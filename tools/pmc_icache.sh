#!/bin/bash
# Instruction-cache behaviour of the timed kernel (the COMMON instantiation is ~65 KB of code; the cache is 64 KB per CU pair).
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --output-format csv --kernel-include-regex burst_pull4 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_INSTS_VALU -d $O/icache -o ic -- python3 $R/bench.py --steps 2 --warmup 1 --main-only > $O/icache.log 2>&1
python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(f"{O}/icache/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "burst_pull4_kernel<false, false, true>" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"{k:30s} {sum(v)/len(v)/(1<<20):12.3f} per burst")
PY
rm -rf $O/icache
