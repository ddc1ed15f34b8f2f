#!/bin/bash
# SQ counters of the 4-SPS kernel on one workload (tools/pmc_mixed.py): per-burst instruction counts, wave / wait / LDS cycles.
#   WORKLOAD=normal|rach|ext|mixed [EXACT=1] bash tools/pmc_workload.sh <tag>   -> gpurun_out/<tag>_pmcw.txt
set -eu
TAG=${1:-w}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
	   "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT"; do
	i=$((i + 1))
	timeout 300 rocprofv3 --output-format csv --kernel-include-regex burst_pull --pmc $SET -d $O/pmcw$i -o $TAG -- python3 $R/tools/pmc_mixed.py > $O/${TAG}_pmcw$i.log 2>&1
done
python3 - "$O" "$TAG" > $O/${TAG}_pmcw.txt <<'PY'
import csv, glob, sys, collections
O, tag = sys.argv[1], sys.argv[2]
n = 1 << 20
acc = collections.defaultdict(list)
for f in sorted(glob.glob(f"{O}/pmcw*/**/{tag}_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
per = {k: sum(v) / len(v) / n for k, v in acc.items()}
for k in sorted(per):
    print(f"{k[0]:44s} {k[1]:24s} {per[k]:10.2f}")
for kern in sorted(set(k[0] for k in per)):
    g = lambda c: per.get((kern, c), 0.0)
    if g("GRBM_GUI_ACTIVE"):
        kcyc = g("GRBM_GUI_ACTIVE") * n / 8
        print(f"{kern}: valu_busy {4 * g('SQ_ACTIVE_INST_VALU') * n / (kcyc * 1024):.3f} lds_busy {g('SQ_LDS_IDX_ACTIVE') * n / (kcyc * 256):.3f} "
              f"waves/SIMD {4 * g('SQ_WAVE_CYCLES') * n / (kcyc * 1024):.2f} wait_share {g('SQ_WAIT_ANY') / max(g('SQ_WAVE_CYCLES'), 1e-9):.3f}")
PY
rm -rf $O/pmcw1 $O/pmcw2
cat $O/${TAG}_pmcw.txt
