#!/bin/bash
# profiles/ub/build_variant.sh <name> <unit> [extra compiler flags...]
# Builds profiles/ub/ko/libag_<name>.so = the product library with ONE translation unit (e.g. ag_blend_backward) recompiled under extra
# flags (-D switches of diagnostic / A-B variants).  Select it with AG_LIB_PATH for a same-box A/B.
set -eo pipefail
name="$1"; unit="$2"; shift 2
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$HERE/../.."
CSRC="$ROOT/animatablegaussians_amd/csrc"
OBJ="$ROOT/animatablegaussians_amd/lib/obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
mkdir -p "$HERE/ko"
[ -f "$OBJ/ag_abi.o" ] || bash "$CSRC/build.sh"
NOPK="-Xclang -target-feature -Xclang -packed-fp32-ops"
CONTRACT="-ffp-contract=fast"
case "$unit" in ag_preprocess|ag_binning) CONTRACT="-ffp-contract=off";; esac
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $CONTRACT $NOPK "$@" \
    -c "$CSRC/$unit.hip" -o "$HERE/ko/${unit}_$name.o" 2> >(grep -v "packed-fp32-ops' is not a recognized feature" >&2)
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$HERE/ko/libag_$name.so" $(ls "$OBJ"/*.o | grep -v "/$unit.o$") "$HERE/ko/${unit}_$name.o"
echo "built $HERE/ko/libag_$name.so"
