mkdir -p gpurun_out/r05b
timeout 900 python -m pytest tests/test_linear_gpu.py tests/test_styleunet_net.py tests/test_grouped_gpu.py tests/test_avatar_net_gpu.py tests/test_trainer_surface.py -x -q -m gpu > gpurun_out/r05b/linear_tests.txt 2>&1; tail -5 gpurun_out/r05b/linear_tests.txt
for rep in 1 2; do
AG_PKG_ROOT=profiles/ub/ko/pkg_old python profiles/per_view_breakdown.py 1 3 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once" > gpurun_out/r05b/pvb_old_$rep.txt; echo "old: $(head -2 gpurun_out/r05b/pvb_old_$rep.txt | tr '\n' ' ')"
python profiles/per_view_breakdown.py 1 3 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once" > gpurun_out/r05b/pvb_new_$rep.txt; echo "new: $(head -2 gpurun_out/r05b/pvb_new_$rep.txt | tr '\n' ' ')"
done
