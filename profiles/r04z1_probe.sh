mkdir -p gpurun_out/r04z1; O=gpurun_out/r04z1; export TMPDIR=/tmp
R=$PWD
python -m pytest tests/test_conv_gpu.py tests/test_grouped_gpu.py tests/test_styleunet_net.py tests/test_styleunet_ops.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -3 | tee $O/tests.txt
for m in 2 1 2 1; do echo "--- AG_WGRAD_BVEC=$m"; AG_WGRAD_BVEC=$m python profiles/host_vs_gpu.py 2>&1 | grep -v amdgpu.ids | tee -a $O/host_vs_gpu_bvec$m.txt; done
rm -rf /tmp/prof_fs; ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_fs -o fs -- python $R/profiles/fullstep_prof.py 8 > /dev/null 2>&1 )
db=$(find /tmp/prof_fs -name "*.db" | head -1); python profiles/summarize_rocprof.py "$db" $O/fullstep_kernel_stats.csv | grep "wgrad\|modulate_finish" | cut -c1-130
