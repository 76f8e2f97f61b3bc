mkdir -p gpurun_out/r04n; O=gpurun_out/r04n
for m in split_bf16x3 default split_bf16x3 default; do echo "--- AG_CONV_MATH=$m"; if [ $m = default ]; then python profiles/host_vs_gpu.py; else AG_CONV_MATH=$m python profiles/host_vs_gpu.py; fi 2>&1 | grep -v amdgpu.ids | tee -a $O/host_vs_gpu_$m.txt; done
