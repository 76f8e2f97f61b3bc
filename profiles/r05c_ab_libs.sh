#!/bin/bash
# same-box A/B of builds of the native library on the raster headline: bash profiles/r05c_ab_libs.sh <variant> [<variant> ...]   (profiles/ub/ko/libag_<variant>.so; "head" = the product library)
cd "$(dirname "$0")/.."
O=gpurun_out/r05c; mkdir -p $O
pick='import json,sys; d=json.loads(sys.stdin.readline()); r=d["roofline_raster_kernels"]; print(sys.argv[1], "value", d["value"], "blocks", d["value_blocks"]["views_per_s"][:3], "seq", d["sequential"]["views_per_s"], "one-stream us:", {k[:14]: r[k]["avg_launch_us"] for k in r if isinstance(r[k], dict)}, "sum", r["one_stream_sum_us"], "bwd overlapped us", d["roofline"]["avg_launch_us"])'
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = head ]; then unset AG_LIB_PATH; else export AG_LIB_PATH=$PWD/profiles/ub/ko/libag_$v.so; fi
  python bench.py --no-full-step --no-cpu-baseline --no-stress 2> /dev/null | python -c "$pick" "$v rep$rep" | tee -a $O/ab_libs.txt
done; done
