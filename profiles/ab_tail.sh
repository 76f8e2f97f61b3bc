# same-box A/B: Blur + noise/bias/activation of the up-sampling StyledConvs as one pass (AG_FUSED_TAIL bit 0 forward, bit 1 backward)
for v in 0 1 3 0 1 3; do echo "AG_FUSED_TAIL=$v"; AG_FUSED_TAIL=$v python profiles/host_vs_gpu.py 2>&1 | grep -v amdgpu.ids | head -2; done
