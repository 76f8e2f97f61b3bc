mkdir -p gpurun_out/r04o; O=gpurun_out/r04o
python -m pytest tests/test_conv_gpu.py -m gpu -q -p no:cacheprovider -x -s -k "fp32_grade or split_f16" 2>&1 | grep -v "^$" | tail -60 | tee $O/conv_tests.txt
python -m pytest tests/test_styleunet_net.py -m gpu -q -p no:cacheprovider -x -s -k "golden" 2>&1 | tail -30 | tee $O/net_tests.txt
for m in split_f16 split_bf16 split_f16 split_bf16; do echo "--- AG_CONV_MATH=$m"; AG_CONV_MATH=$m python profiles/host_vs_gpu.py 2>&1 | grep -v amdgpu.ids | tee -a $O/host_vs_gpu_$m.txt; done
