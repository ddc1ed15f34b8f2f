#!/bin/bash
# Same-box comparison of libraries on every slot type: bash tools/ab_types.sh <rounds> <steps> libA.so libB.so ...
N=$1; K=$2; shift 2
for i in $(seq $N); do
	for L in "$@"; do
		echo "== $(basename $L)"
		TRXHIP_LIB=$PWD/$L python3 tools/bench_types.py $K 2>/dev/null | grep -v amdgpu.ids
		TRXHIP_LIB=$PWD/$L python3 tools/bench_exact.py 2>/dev/null | tail -1
	done
done
