mkdir -p gpurun_out/r04e; O=gpurun_out/r04e
python -m pytest tests/test_grouped_gpu.py -m gpu -q -p no:cacheprovider -x -s 2>&1 | tail -5
AG_GROUPED_STREAMS=2 python -m pytest tests/test_grouped_gpu.py -m gpu -q -p no:cacheprovider -x -k "three_networks or multi_view" 2>&1 | tail -3
for m in 1 2 1 2; do echo "--- AG_GROUPED_STREAMS=$m"; AG_GROUPED_STREAMS=$m python profiles/host_vs_gpu.py 2>&1 | grep -v amdgpu.ids | tee -a $O/host_vs_gpu_streams$m.txt; done
