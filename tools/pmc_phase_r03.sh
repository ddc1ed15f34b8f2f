#!/bin/bash
# Per-phase instruction counts of the final hot kernel (burst_pull4_kernel<false,false,true>): one --pmc pass per
# compile-time ablation variant (tools/build_variants.py abl_*), differences = phases.  Run on the GPU box from the repo root:
#   python tools/build_variants.py abl_full:-DTRX_ABL_MASK=0 abl_nodemod:-DTRX_ABL_MASK=0x1 abl_nodetect:-DTRX_ABL_MASK=0x8 \
#          abl_argmax:-DTRX_ABL_MASK=0x201 abl_nopeak:-DTRX_ABL_MASK=0x3 abl_noci:-DTRX_ABL_MASK=0x81 abl_nofir:-DTRX_ABL_MASK=0x20
#   bash tools/pmc_phase_r03.sh        -> gpurun_out/r03/phase_counters.json
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for V in full nodemod nodetect argmax nopeak noci nofir; do
	TRXHIP_LIB=$R/osmo_trx_amd/lib/libtrxhip_abl_$V.so timeout 300 rocprofv3 --output-format csv --kernel-include-regex burst_pull4 \
		--pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE -d $O/ph_$V -o ph \
		-- python3 $R/bench.py --steps 2 --warmup 1 --main-only > $O/ph_$V.log 2>&1
done
python3 - "$O" > $O/phase_counters.json <<'PY'
import csv, glob, sys, json, collections
O = sys.argv[1]
n = 1 << 20
out = {}
for v in "full nodemod nodetect argmax nopeak noci nofir".split():
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{O}/ph_{v}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "burst_pull4_kernel<false, false, true>" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out[v] = {k: round(sum(x) / len(x) / n, 2) for k, x in acc.items()}
print(json.dumps(out, indent=1))
PY
rm -rf $O/ph_full $O/ph_nodemod $O/ph_nodetect $O/ph_argmax $O/ph_nopeak $O/ph_noci $O/ph_nofir
cat $O/phase_counters.json
