#!/bin/bash
# Same-box A/B of two builds of the library (boxes differ by a few percent between gpurun calls):
#   bash tools/ab.sh osmo_trx_amd/lib/libtrxhip_A.so osmo_trx_amd/lib/libtrxhip.so [rounds]
A=$1; B=$2; N=${3:-3}
for i in $(seq $N); do
	for L in $A $B; do
		v=$(TRXHIP_LIB=$PWD/$L python3 bench.py --main-only --steps 40 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
		echo "$(basename $L) $v"
	done
done
