#!/bin/bash
# the driver's command (--steps 20 --warmup 5) under different numbers of untimed pre-warm steps: does the 3-ms region still see a GPU that is ramping up?
cd "$(dirname "$0")/.."
O=gpurun_out/r05c; mkdir -p $O
F="--no-full-step --no-cpu-baseline --no-stress"
pick='import json,sys; d=json.loads(sys.stdin.readline()); print(sys.argv[1], d["value"], d["value_blocks"]["views_per_s"], d["roofline"]["avg_launch_us"])'
for rep in 1 2 3; do
for PW in 300 1000 3000 10000; do
  python bench.py --steps 20 --warmup 5 --prewarm $PW $F 2> /dev/null | python -c "$pick" "K20 prewarm=$PW rep=$rep" | tee -a $O/prewarm_sweep.txt
done; done
