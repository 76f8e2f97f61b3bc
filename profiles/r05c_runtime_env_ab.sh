#!/bin/bash
# same-box A/B of HIP runtime settings on the 1-view training step (814 dependent launches per step, ~2 us of idle time between them)
cd "$(dirname "$0")/.."
O=gpurun_out/r05c; mkdir -p $O
for rep in 1 2; do
for kv in X=0 HIP_FORCE_DEV_KERNARG=1 HIP_FORCE_DEV_KERNARG=0 GPU_MAX_HW_QUEUES=2 HSA_ENABLE_INTERRUPT=0; do
echo "$kv: $(env $kv python profiles/views_scaling.py 1 1 1 2>/dev/null | tr '\n' ' ')" | tee -a $O/runtime_env_ab.txt
done; done
