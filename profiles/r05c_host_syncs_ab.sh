#!/bin/bash
# the camera tensors' identity cache (no device->host copy per render call) + the background colour uploaded once: same-box A/B on the 1-view step, the
# idle-time profile, then the full GPU suite and smoke at this state
cd "$(dirname "$0")/.."
O=gpurun_out/r05c; mkdir -p $O
for kv in AG_CAMERA_IDENT_CACHE=0 AG_CAMERA_IDENT_CACHE=1 AG_CAMERA_IDENT_CACHE=0 AG_CAMERA_IDENT_CACHE=1; do
echo "$kv: $(env $kv python profiles/views_scaling.py 1 1 4 2>/dev/null | tr '\n' ' ')" | tee -a $O/host_syncs_ab.txt
done
python profiles/step_gaps.py 1 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once" > $O/step_gaps_after.txt; head -3 $O/step_gaps_after.txt
( time python -m pytest tests -m gpu -q -x 2>&1 | tail -3 ) > $O/gputests_after_host_syncs.txt 2>&1; cat $O/gputests_after_host_syncs.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
