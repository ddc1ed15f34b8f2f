#!/bin/bash
# Same-box comparison of several builds of the library on the main workload AND the access-burst legs of bench.py:
#   bash tools/ab_legs.sh <rounds> lib1.so lib2.so ...   -> per (round, library): main value, RACH, EXT_RACH Mbursts/s
N=$1; shift 1
for i in $(seq $N); do
	for L in "$@"; do
		v=$(TRXHIP_LIB=$PWD/$L python3 bench.py --legs c2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); oc=d['config']['other_configs']
print(d['value'], oc['configs[2]']['mbursts_per_s_all_gpus'], oc['configs[2]_ext_rach']['mbursts_per_s_all_gpus'])")
		echo "$(basename $L) $v"
	done
done
