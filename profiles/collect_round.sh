#!/bin/bash
# Round artefacts for profiles/ from ONE box:  profiles/collect_round.sh <tag>    (run through gpurun; outputs under gpurun_out/)
#   <tag>_bench.json                 python bench.py (the driver's default command)
#   <tag>_bench_kernel_stats.csv     rocprofv3 --kernel-trace --stats of the same command, summarised per kernel
#   <tag>_conv_layers.csv            per-layer forward / dgrad / wgrad table of the DualStyleUNet convolutions
#   <tag>_styleunet_kernel_stats.csv rocprofv3 summary of DualStyleUNet forward and forward + backward passes
#   <tag>_conv_math_ab.txt           interleaved A/B of the three convolution arithmetic modes (network pass, training step)
#   <tag>_host_vs_gpu.txt            host issue time against GPU time of the network pass and the training step
cd "$(dirname "$0")/.."
tag=${1:-rXX}
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -c 600 gpurun_out/${tag}_bench.json; echo
rm -rf /tmp/prof_bench
rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o b -- python bench.py > gpurun_out/${tag}_bench_profiled.json 2> /dev/null
db=$(find /tmp/prof_bench -name "*.db" | head -1)
python profiles/summarize_rocprof.py "$db" gpurun_out/${tag}_bench_kernel_stats.csv | head -14 | cut -c1-140
python profiles/conv_layers.py gpurun_out/${tag}_conv_layers.csv | tail -1
rm -rf /tmp/prof_su
rocprofv3 --kernel-trace --stats -d /tmp/prof_su -o su -- python profiles/styleunet_bench.py 4 > gpurun_out/${tag}_styleunet_bench.log 2>&1
db=$(find /tmp/prof_su -name "*.db" | head -1)
python profiles/summarize_rocprof.py "$db" gpurun_out/${tag}_styleunet_kernel_stats.csv | head -8 | cut -c1-140
grep -E "^fwd" gpurun_out/${tag}_styleunet_bench.log
python profiles/conv_math_ab.py 3 > gpurun_out/${tag}_conv_math_ab.txt 2>&1; tail -3 gpurun_out/${tag}_conv_math_ab.txt
python profiles/host_vs_gpu.py > gpurun_out/${tag}_host_vs_gpu.txt 2>&1; tail -3 gpurun_out/${tag}_host_vs_gpu.txt
