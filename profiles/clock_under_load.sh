#!/bin/bash
# Shader clock and socket power while a workload runs (rocm-smi sampled beside it): is the matrix pipe's 20.7 ns on real operands a clock drop?
#   bash profiles/clock_under_load.sh
R=${GRAFT_REPO_ROOT:-$PWD}
sample() { # label
  for i in 1 2 3 4; do
    sleep 0.6
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Graphics Package Power" | sed "s/^/$1  /"
  done
}
echo "== idle"; sample idle
echo "== convolution 256->256 @256^2, split_f16 (three products), back to back"
( AG_CONV_MATH=split_f16 python $R/profiles/conv_one.py 256 256 256 256 3 1 1 40000 fwd > /dev/null 2>&1 & echo $! > /tmp/load.pid ); sleep 4; sample split_f16; kill $(cat /tmp/load.pid) 2>/dev/null; sleep 1
echo "== the same, f16 (one product)"
( AG_CONV_MATH=f16 python $R/profiles/conv_one.py 256 256 256 256 3 1 1 60000 fwd > /dev/null 2>&1 & echo $! > /tmp/load.pid ); sleep 4; sample f16; kill $(cat /tmp/load.pid) 2>/dev/null; sleep 1
echo "== pure fp16 MFMA stream (mfma_floor_f16: constants, then random high parts, then h / l mix; ~0.5 s each at 1 / 2 / 4 waves per SIMD)"
( for i in 1 2 3 4 5 6; do $R/profiles/ub/mfma_floor_f16 > /dev/null 2>&1; done & echo $! > /tmp/load.pid ); sleep 0.3
for i in $(seq 1 14); do sleep 0.35; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Graphics Package Power" | sed "s/^/mfma  /"; done
kill $(cat /tmp/load.pid) 2>/dev/null
