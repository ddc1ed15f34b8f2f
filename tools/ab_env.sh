#!/bin/bash
# Same-box comparison of environment switches: bash tools/ab_env.sh <rounds> <steps> "" "VAR=1" ...
N=$1; K=$2; shift 2
for i in $(seq $N); do
	for E in "$@"; do
		v=$(env $E python3 bench.py --main-only --steps $K 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])")
		echo "[$E] $v"
	done
done
