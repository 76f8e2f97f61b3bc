#!/bin/bash
# parallel slice reduction of the weight gradient (AG_WGRAD_REDUCE_PAR): parity tests with the switch on (default), then the same-box A/B by device kernel time
cd "$(dirname "$0")/.."
O=gpurun_out/r05c; mkdir -p $O
python -m pytest tests/test_conv_gpu.py tests/test_grouped_gpu.py -q -x 2>&1 | tail -3
AG_WGRAD_SPLITS=64 python -m pytest tests/test_conv_gpu.py -q -x -k "forward_backward or deterministic" 2>&1 | tail -2
for rep in 1 2; do
for kv in AG_WGRAD_REDUCE_PAR=0 AG_WGRAD_REDUCE_PAR=1; do
env $kv python profiles/per_view_breakdown.py 1 3 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once" > $O/pvb_${kv}_$rep.txt; echo "$kv: $(head -2 $O/pvb_${kv}_$rep.txt | tr '\n' ' ') | wgrad_reduce $(grep wgrad_reduce $O/pvb_${kv}_$rep.txt | awk '{s+=$1} END {print s}') us/view"
done; done
