# same-box A/B of two builds of the native library by device kernel time per training step: bash profiles/r05b_ab_lib.sh <variant name under profiles/ub/ko>
mkdir -p gpurun_out/r05b
for rep in 1 2; do
AG_LIB_PATH=$PWD/profiles/ub/ko/libag_$1.so python profiles/per_view_breakdown.py 1 3 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once" > gpurun_out/r05b/pvb_lib_$1_$rep.txt; echo "$1: $(head -2 gpurun_out/r05b/pvb_lib_$1_$rep.txt | tr '\n' ' ')"
python profiles/per_view_breakdown.py 1 3 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once" > gpurun_out/r05b/pvb_lib_head_$rep.txt; echo "head: $(head -2 gpurun_out/r05b/pvb_lib_head_$rep.txt | tr '\n' ' ')"
done
