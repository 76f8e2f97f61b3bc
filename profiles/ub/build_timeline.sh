#!/bin/bash
# Builds profiles/ub/ko/libag_timeline.so = libag_hip.so with the blend backward compiled under -DAG_BWD_TIMELINE (per-item time stamps,
# read by profiles/bwd_wg_times.py through ag_debug_bwd_timeline).  Diagnostic only.
set -eo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$HERE/../.."
CSRC="$ROOT/animatablegaussians_amd/csrc"
OBJ="$ROOT/animatablegaussians_amd/lib/obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
mkdir -p "$HERE/ko"
[ -f "$OBJ/ag_abi.o" ] || bash "$CSRC/build.sh"
NOPK="-Xclang -target-feature -Xclang -packed-fp32-ops"
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -ffp-contract=fast $NOPK -DAG_BWD_TIMELINE \
    -c "$CSRC/ag_blend_backward.hip" -o "$HERE/ko/ag_blend_backward_tl.o" 2> >(grep -v "packed-fp32-ops' is not a recognized feature" >&2)
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$HERE/ko/libag_timeline.so" $(ls "$OBJ"/*.o | grep -v '/ag_blend_backward.o$') "$HERE/ko/ag_blend_backward_tl.o"
echo "built $HERE/ko/libag_timeline.so"
