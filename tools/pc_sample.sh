#!/bin/bash
# PC-sampling profile of the timed kernel (rocprofv3 beta feature): where the waves' time goes, per instruction.
#   bash tools/pc_sample.sh <tag> [stochastic|host_trap]  -> gpurun_out/<tag>_pcs_*.txt (header, sample rows aggregated per PC)
TAG=${1:-pcs}; M=${2:-stochastic}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ "$M" = stochastic ]; then U=cycles; I=${3:-65536}; else U=time; I=${3:-1}; fi
timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $M --pc-sampling-unit $U --pc-sampling-interval $I \
	--kernel-trace --output-format csv -d $O/pcs_$TAG -o $TAG -- python3 $R/bench.py --main-only --steps 20 --warmup 2 > $O/${TAG}_pcs.log 2>&1
echo "rc=$?" >> $O/${TAG}_pcs.log
ls -la $O/pcs_$TAG/* >> $O/${TAG}_pcs.log 2>&1
python3 - "$O" "$TAG" <<'PY'
import csv, glob, sys, collections, os
O, tag = sys.argv[1], sys.argv[2]
for f in glob.glob(f"{O}/pcs_{tag}/**/*.csv", recursive=True):
    if "pc_sampling" not in f:
        continue
    out = open(f"{O}/{tag}_pcs_{os.path.basename(f)}.txt", "w")
    rd = csv.reader(open(f))
    hdr = next(rd)
    out.write("HEADER " + ",".join(hdr) + "\n")
    rows = 0
    agg = collections.Counter()
    keep = []
    for r in rd:
        rows += 1
        if rows <= 10:
            keep.append(",".join(r))
        d = dict(zip(hdr, r))
        key = tuple(d.get(k, "") for k in hdr if any(t in k.lower() for t in ("offset", "instruction", "stall", "reason", "type", "code_object", "arb", "issued")) and "time" not in k.lower() and "id" != k.lower())
        agg[key] += 1
    out.write(f"ROWS {rows}\n" + "\n".join(keep) + "\n")
    for k, n in agg.most_common(6000):
        out.write(f"{n}\t" + "\t".join(k) + "\n")
    out.close()
PY
find $O/pcs_$TAG -name "*.csv" -size +1M -delete
du -sh $O
