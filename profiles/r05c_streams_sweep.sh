#!/bin/bash
# Same-box sweep of bench.py's --streams (views in flight) under the driver's command (--steps 20 --warmup 5: a 3-ms region, so the
# fill / drain of the stream pipeline and 20 mod streams weigh) and under the default 2000-step region.  One gpurun call.
cd "$(dirname "$0")/.."
O=gpurun_out/r05c; mkdir -p $O
F="--no-full-step --no-cpu-baseline --no-stress"
pick='import json,sys; d=json.loads(sys.stdin.readline()); print(sys.argv[1], d["value"], d.get("value_blocks",{}).get("views_per_s"), d.get("roofline",{}).get("frac"))'
for rep in 1 2; do
for S in 3 4 2 5 6; do
  python bench.py --steps 20 --warmup 5 --streams $S $F 2> /dev/null | tee $O/drv_s${S}_r${rep}.json | python -c "$pick" "K20 streams=$S rep=$rep" >> $O/streams_sweep.txt
done; done
for S in 3 4 5; do
  python bench.py --streams $S $F 2> /dev/null | tee $O/long_s${S}.json | python -c "$pick" "K2000 streams=$S" >> $O/streams_sweep.txt
done
cat $O/streams_sweep.txt
