mkdir -p gpurun_out/r04s; O=gpurun_out/r04s; export TMPDIR=/tmp
R=$PWD
python -m pytest tests/test_conv_gpu.py -m gpu -q -p no:cacheprovider -x -s -k "fp32_grade or split_f16" 2>&1 | grep -v "^$" | grep "split_f16\|passed\|failed\|Error\|assert" | tee $O/conv_tests.txt
python -m pytest tests/test_styleunet_net.py -m gpu -q -p no:cacheprovider -x -k "golden" 2>&1 | tail -3 | tee $O/net_tests.txt
for m in split_f16 split_bf16 split_f16 split_bf16; do echo "--- AG_CONV_MATH=$m"; AG_CONV_MATH=$m python profiles/host_vs_gpu.py 2>&1 | grep -v amdgpu.ids | tee -a $O/host_vs_gpu_$m.txt; done
rm -rf /tmp/prof_fs; ( cd /tmp && AG_CONV_MATH=split_f16 rocprofv3 --kernel-trace --stats -d /tmp/prof_fs -o fs -- python $R/profiles/fullstep_prof.py 8 > /dev/null 2>&1 )
db=$(find /tmp/prof_fs -name "*.db" | head -1); python profiles/summarize_rocprof.py "$db" $O/fullstep_kernel_stats_split_f16.csv | head -8 | cut -c1-130
