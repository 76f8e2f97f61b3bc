# same-box A/B of an environment switch by device kernel time per training step:  bash profiles/r05b_ab_env.sh VAR=off VAR=on
mkdir -p gpurun_out/r05b
for rep in 1 2; do
for kv in "$@"; do
env $kv python profiles/per_view_breakdown.py 1 3 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once" > gpurun_out/r05b/pvb_${kv}_$rep.txt; echo "$kv: $(head -2 gpurun_out/r05b/pvb_${kv}_$rep.txt | tr '\n' ' ') | wgrad $(grep wgrad gpurun_out/r05b/pvb_${kv}_$rep.txt | awk '{s+=$1} END {print s}') us/view"
done; done
