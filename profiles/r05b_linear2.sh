mkdir -p gpurun_out/r05b
timeout 900 python -m pytest tests/test_linear_gpu.py -x -q -m gpu 2>&1 | tail -2
python profiles/per_view_breakdown.py 1 3 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once" > gpurun_out/r05b/per_view_breakdown_linear.txt; head -2 gpurun_out/r05b/per_view_breakdown_linear.txt; grep bilinear gpurun_out/r05b/per_view_breakdown_linear.txt
