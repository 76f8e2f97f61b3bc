mkdir -p gpurun_out/r04m; O=gpurun_out/r04m; export TMPDIR=/tmp
python -m pytest tests/test_styleunet_ops.py tests/test_grouped_gpu.py tests/test_styleunet_net.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -5
for m in 1 0 1 0; do echo "--- AG_FUSED_ACT=$m"; AG_FUSED_ACT=$m python profiles/host_vs_gpu.py 2>&1 | grep -v amdgpu.ids | tee -a $O/host_vs_gpu_fused$m.txt; done
R=$PWD; rm -rf /tmp/prof_fs; ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_fs -o fs -- python $R/profiles/fullstep_prof.py 8 > /dev/null 2>&1 )
db=$(find /tmp/prof_fs -name "*.db" | head -1); python profiles/summarize_rocprof.py "$db" $O/fullstep_kernel_stats.csv | head -3 | cut -c1-100
grep -i "fir4x4\|noise_bias" $O/fullstep_kernel_stats.csv | cut -c1-160
