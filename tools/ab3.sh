#!/bin/bash
# Same-box comparison: bash tools/ab3.sh <rounds> <steps> libA.so libB.so ...  ("old" = the default library with TRXHIP_NO_NB_KERNEL=1)
N=$1; K=$2; shift 2
for i in $(seq $N); do
	for L in "$@"; do
		if [ "$L" = "old" ]; then
			v=$(TRXHIP_NO_NB_KERNEL=1 python3 bench.py --main-only --steps $K 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])")
		else
			v=$(TRXHIP_LIB=$PWD/$L python3 bench.py --main-only --steps $K 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])")
		fi
		echo "$(basename $L) $v"
	done
done
