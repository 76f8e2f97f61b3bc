#!/bin/bash
# Same-box A/B of libag_hip.so builds on the raster bench (per-kernel HIP-event breakdown): profiles/raster_ab.sh <tag> <lib> [...]
cd "$(dirname "$0")/.."
while [ $# -ge 2 ]; do
  tag=$1; lib=$2; shift 2
  AG_LIB_PATH=$lib python bench.py --steps 600 --warmup 150 --breakdown --no-full-step --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_us']
print('$tag', 'views/s', d['value'], 'seq', d['sequential']['views_per_s'], ' '.join(f'{n.replace(\"_kernel\",\"\")}={v}' for n,v in k.items() if v))"
done
