#!/bin/bash
# Builds the stand-alone packed-fp32 hazard reproducer (profiles/ub/pk_hazard) and the knock-out builds of the convolution library it
# dlopens (profiles/ub/ko/libag_ko<mask>.so = libag_hip.so with ag_conv.hip compiled under -DAG_CONV_KNOCKOUT=<mask>).  Needs the
# product objects (animatablegaussians_amd/csrc/build.sh) to exist.  Cross-compiles without a GPU.
# NOTE (round 4): the knock-out switches (-DAG_CONV_KNOCKOUT) were removed from ag_conv.hip once their results were recorded
# (profiles/r03_packed_fp32_hazard.md); the knock-out library variants build from the round-3 tree: `git checkout fa8b937 -- animatablegaussians_amd/csrc/ag_conv.hip`
# in a scratch worktree.  The stand-alone reproducer (pk_hazard.hip, its own synthetic aggressor kernel) builds from this tree as before.
set -eo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$HERE/../.."
CSRC="$ROOT/animatablegaussians_amd/csrc"
OBJ="$ROOT/animatablegaussians_amd/lib/obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
mkdir -p "$HERE/ko"
[ -f "$OBJ/ag_abi.o" ] || bash "$CSRC/build.sh"
NOPK="-Xclang -target-feature -Xclang -packed-fp32-ops"
others=$(ls "$OBJ"/*.o | grep -v '/ag_conv.o$')
pids=()
for mask in ${KO_MASKS:-1 2 4 8 16 24 7 15 22}; do
  (
    out="$HERE/ko/ag_conv_ko$mask.o"
    if [ ! -f "$out" ] || [ "$CSRC/ag_conv.hip" -nt "$out" ]; then
      "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -ffp-contract=fast $NOPK -DAG_CONV_KNOCKOUT=$mask \
          -c "$CSRC/ag_conv.hip" -o "$out" 2> >(grep -v "packed-fp32-ops' is not a recognized feature" >&2)
    fi
    "$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$HERE/ko/libag_ko$mask.so" $others "$out"
  ) &
  pids+=($!)
done
# the reproducer itself: packed fp32 ON (default target features), so the included pointwise kernels are the disturbed form
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -Wno-unused-function -o "$HERE/pk_hazard" "$HERE/pk_hazard.hip" -ldl
fail=0
for p in "${pids[@]}"; do wait "$p" || fail=1; done
[ "$fail" -eq 0 ] || { echo "build_pk_hazard.sh: a knock-out build failed" >&2; exit 1; }
echo "built $HERE/pk_hazard and $(ls "$HERE"/ko/*.so | wc -l) knock-out libraries"
