# same-box A/B of the Python package (profiles/ub/ko/pkg_old = the package at the start of this session) by device kernel time per training step
mkdir -p gpurun_out/r05b
for rep in 1 2; do
AG_PKG_ROOT=profiles/ub/ko/pkg_old python profiles/per_view_breakdown.py 1 3 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once" > gpurun_out/r05b/pvb_old_$rep.txt; echo "old: $(head -2 gpurun_out/r05b/pvb_old_$rep.txt | tr '\n' ' ')"
python profiles/per_view_breakdown.py 1 3 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once" > gpurun_out/r05b/pvb_new_$rep.txt; echo "new: $(head -2 gpurun_out/r05b/pvb_new_$rep.txt | tr '\n' ' ')"
done
