#!/bin/bash
# Instruction counts per burst of the hot kernel (main workload of bench.py) for several builds of the library, one rocprofv3
# --pmc pass each:   bash tools/pmc_insts.sh lib1.so lib2.so ...   -> one line per library: VALU SALU LDS per burst, kernel ms
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
	export TRXHIP_LIB=$R/$L
	rm -rf $O/pmci
	timeout 300 rocprofv3 --output-format csv --kernel-include-regex burst_pull4 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/pmci -o x -- python3 $R/bench.py --steps 2 --warmup 1 --main-only > $O/pmci.log 2>&1
	python3 - "$O" "$(basename $L)" <<'PY'
import csv, glob, sys, collections
O, name = sys.argv[1], sys.argv[2]
n = 1 << 20
acc = collections.defaultdict(list)
for f in glob.glob(f"{O}/pmci/**/x_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "burst_pull4_kernel<false, false, true>" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(name, " ".join(f"{k[3:]}={sum(v) / len(v) / n:.1f}" for k, v in sorted(acc.items())))
PY
done
rm -rf $O/pmci
