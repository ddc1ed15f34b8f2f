#!/bin/bash
# Same-box comparison of several builds of the library (the boxes of the pool differ by a few percent between gpurun calls):
#   bash tools/ab_multi.sh <rounds> <steps> lib1.so lib2.so ...   -> one line per (round, library): value of bench.py --main-only
N=$1; K=$2; shift 2
for i in $(seq $N); do
	for L in "$@"; do
		v=$(TRXHIP_LIB=$PWD/$L python3 bench.py --main-only --steps $K 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])")
		echo "$(basename $L) $v"
	done
done
