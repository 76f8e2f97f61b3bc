#!/bin/bash
# absmax grid sized to the largest job (AG_AMAX_SIZED_GRID): parity tests with the switch on (default), then the same-box A/B by device kernel time
cd "$(dirname "$0")/.."
O=gpurun_out/r05c; mkdir -p $O
python -m pytest tests/test_conv_gpu.py tests/test_grouped_gpu.py tests/test_styleunet_net.py tests/test_zz_guards_gpu.py -q -x 2>&1 | tail -2
for rep in 1 2; do
for kv in AG_AMAX_SIZED_GRID=0 AG_AMAX_SIZED_GRID=1; do
env $kv python profiles/per_view_breakdown.py 1 3 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once" > $O/pvb_${kv}_$rep.txt; echo "$kv: $(head -2 $O/pvb_${kv}_$rep.txt | tr '\n' ' ')"
AG_DUMP=1 env $kv python profiles/per_view_breakdown.py 1 3 2>&1 | grep "absmax" | head -3
done; done
