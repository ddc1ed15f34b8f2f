#!/bin/bash
# Samples the shader clock and socket power while the timed leg of bench.py runs (diagnostic: is the kernel clock- or
# power-limited?).   bash tools/clock_watch.sh   -> gpurun_out/clock_watch.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
python3 $R/bench.py --main-only --steps 16000 --warmup 3 > $O/clock_watch_bench.json 2>/dev/null &
BP=$!
while kill -0 $BP 2>/dev/null; do
	rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|Package Power|GPU use" | sed 's/.*: //' | tr '\n' ' '
	echo
	sleep 2
done > $O/clock_watch.txt
wait $BP
cat $O/clock_watch.txt
cut -c1-200 $O/clock_watch_bench.json
