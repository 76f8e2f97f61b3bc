mkdir -p gpurun_out/r05b
python -m pytest tests/test_avatar_net_gpu.py tests/test_trainer_surface.py tests/test_avatar_gpu.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2; do
AG_PKG_ROOT=profiles/ub/ko/pkg_prev python profiles/per_view_breakdown.py 1 4 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once" > gpurun_out/r05b/pvb_prev_$rep.txt; echo "prev: $(head -2 gpurun_out/r05b/pvb_prev_$rep.txt | tr '\n' ' ')"
python profiles/per_view_breakdown.py 1 4 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once" > gpurun_out/r05b/pvb_new_$rep.txt; echo "new: $(head -2 gpurun_out/r05b/pvb_new_$rep.txt | tr '\n' ' ')"
done
