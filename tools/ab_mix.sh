#!/bin/bash
# Same-box, settled-clock comparison of several builds on normal / access / 7:1 mixed batches (tools/mix_check.py):
#   bash tools/ab_mix.sh lib1.so lib2.so ...
for L in "$@"; do
	echo "$(basename $L): $(TRXHIP_LIB=$PWD/$L python3 tools/mix_check.py 2>/dev/null | tail -1)"
done
