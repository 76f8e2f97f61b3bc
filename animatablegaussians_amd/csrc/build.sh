#!/bin/bash
# Builds libag_hip.so for gfx950 (cross-compiles without a GPU).  Output: animatablegaussians_amd/lib/libag_hip.so
set -eo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../lib"
OBJ="$HERE/../lib/obj"
mkdir -p "$OUT" "$OBJ"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
# fp32 op order is a parity contract in the preprocess kernels: no FMA contraction there.
EXACT="-ffp-contract=off"
FAST="-ffp-contract=fast"
compile() { # src flags
  local src="$1"; shift
  local obj="$OBJ/$(basename "${src%.hip}").o"
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ "$HERE/ag_common.h" -nt "$obj" ] || [ "$HERE/../../include/ag_raster.h" -nt "$obj" ] || [ "$HERE/../../include/ag_avatar.h" -nt "$obj" ] || [ "$HERE/../../include/ag_styleunet.h" -nt "$obj" ] || [ "$HERE/../../include/ag_conv.h" -nt "$obj" ] || [ "$HERE/../../include/ag_lpips.h" -nt "$obj" ] || [ "$HERE/../../include/ag_smplx.h" -nt "$obj" ] || [ "$HERE/ag_sh.h" -nt "$obj" ]; then
    echo "hipcc $(basename "$src") $*"
    rm -f "$obj"
    "$HIPCC" $COMMON "$@" -c "$src" -o "$obj"
  fi
}
compile "$HERE/ag_abi.hip" $FAST &
compile "$HERE/ag_preprocess.hip" $EXACT &
compile "$HERE/ag_binning.hip" $EXACT &
compile "$HERE/ag_blend_forward.hip" $FAST &
compile "$HERE/ag_blend_backward.hip" $FAST &
compile "$HERE/ag_preprocess_backward.hip" $FAST &
compile "$HERE/ag_avatar.hip" $FAST &
compile "$HERE/ag_styleunet_ops.hip" $FAST &
compile "$HERE/ag_conv.hip" $FAST &
compile "$HERE/ag_conv_pointwise.hip" $FAST &
compile "$HERE/ag_lpips.hip" $FAST &
compile "$HERE/ag_smplx.hip" $FAST &
fail=0
for job in $(jobs -p); do wait "$job" || fail=1; done
if [ "$fail" -ne 0 ]; then echo "build.sh: compilation failed" >&2; exit 1; fi
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT/libag_hip.so" "$OBJ"/*.o
echo "built $OUT/libag_hip.so"
