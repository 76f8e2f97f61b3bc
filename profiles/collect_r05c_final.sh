#!/bin/bash
# the last commit's subset of collect_r05c.sh: the driver's command, the default bench line, the full GPU suite, smoke
cd "$(dirname "$0")/.."
O=gpurun_out/r05zz; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd_full.json 2> /dev/null
( time python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; tail -3 $O/bench.time
( time python -m pytest tests -m gpu -q 2>&1 | tail -4 ) > $O/gputests_head.txt 2>&1; cat $O/gputests_head.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
