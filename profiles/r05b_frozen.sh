mkdir -p gpurun_out/r05b
timeout 900 python -m pytest tests/test_frozen_weights_gpu.py tests/test_avatar_net_gpu.py tests/test_grouped_gpu.py tests/test_optim_gpu.py -x -q -m gpu 2>&1 | tail -12
for rep in 1 2; do
AG_FROZEN_WEIGHTS=0 python bench_avatar.py --infer --steps 40 --warmup 10 2>&1 | grep -o '"value": [0-9.]*, "unit": "views/s"' | sed 's/^/off: /'
python bench_avatar.py --infer --steps 40 --warmup 10 2>&1 | grep -o '"value": [0-9.]*, "unit": "views/s"' | sed 's/^/on:  /'
done
python profiles/infer_kernel_table.py 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once" | head -14
