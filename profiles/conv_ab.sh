#!/bin/bash
# Same-box A/B of libag_hip.so builds on the per-layer convolution table: profiles/conv_ab.sh <tag> <lib> [<tag> <lib> ...]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
while [ $# -ge 2 ]; do
  tag=$1; lib=$2; shift 2
  AG_LIB_PATH=$lib python profiles/conv_layers.py gpurun_out/conv_layers_$tag.csv | tail -1
done
