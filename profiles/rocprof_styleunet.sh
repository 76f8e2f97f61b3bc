#!/bin/bash
# rocprofv3 kernel summary of DualStyleUNet forward + backward:  profiles/rocprof_styleunet.sh <tag>   -> gpurun_out/<tag>_styleunet_kernel_stats.csv
cd "$(dirname "$0")/.."
tag=${1:-run}
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
rm -rf "$out"
rocprofv3 --kernel-trace --stats -d "$out" -o su -- python profiles/styleunet_bench.py 4 > gpurun_out/${tag}_styleunet_bench.log 2>&1
f=$(find "$out" -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/${tag}_styleunet_kernel_stats.csv
head -30 gpurun_out/${tag}_styleunet_kernel_stats.csv | cut -c1-160
tail -3 gpurun_out/${tag}_styleunet_bench.log
