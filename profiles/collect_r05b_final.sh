#!/bin/bash
# the last commit's subset of collect_r05b.sh: the default bench line, the driver's command, the full GPU suite
cd "$(dirname "$0")/.."
O=gpurun_out/r05y; mkdir -p $O
( time python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; tail -3 $O/bench.time
python bench.py --steps 20 --warmup 5 --no-full-step --no-cpu-baseline --no-stress > $O/bench_driver_cmd.json 2> /dev/null
( time python -m pytest tests -m gpu -q 2>&1 | tail -4 ) > $O/gputests_head.txt 2>&1; cat $O/gputests_head.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
