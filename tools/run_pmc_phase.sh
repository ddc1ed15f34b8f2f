set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
export TRXHIP_LIB=$R/osmo_trx_amd/lib/libtrxhip_diag.so
export DIAG_MASKS=0,0x1,0x2,0x8,0x200,0x80,0x40,0x20,0x10,0x4
timeout 420 rocprofv3 --output-format csv --kernel-include-regex burst_pull4 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $R/gpurun_out/pmc_phase -o ph -- python3 $R/tools/pmc_phase.py > $R/gpurun_out/pmc_phase.log 2>&1
cd $R/gpurun_out/pmc_phase && find . -name "*counter_collection.csv" | head -3
python3 - > $R/gpurun_out/pmc_phase_summary.txt <<'PY'
import csv, glob, collections
f = glob.glob("**/*counter_collection.csv", recursive=True)[0]
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "burst_pull4" not in r["Kernel_Name"]: continue
    rows.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
masks = "0,0x1,0x2,0x8,0x200,0x80,0x40,0x20,0x10,0x4".split(",")
n = 1 << 17
for (d, c), m in zip(rows.items(), masks):
    print(m, {k: round(v / n, 1) for k, v in c.items()})
PY
rm -rf "$R/gpurun_out/pmc_phase"
