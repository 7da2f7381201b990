#!/bin/bash
# Build the C-ABI library of ANOTHER revision of goliath_amd/csrc next to the current one, for same-box A/B runs:
#   bash tools/build_ab_lib.sh <git-rev> <tag>      ->  goliath_amd/lib/libgoliath_hip_<tag>.so
# Run HERE (needs .git); the .so is git-ignored but travels to the GPU box, where
#   GOLIATH_HIP_LIB=goliath_amd/lib/libgoliath_hip_<tag>.so python bench.py ...
# times the old kernels under the current host code (the ABI must not have changed between the two revisions).
set -e
REV=${1:?git revision}; TAG=${2:?tag}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
git -C "$ROOT" archive "$REV" goliath_amd/csrc include | tar -x -C "$TMP"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=fast -Wno-unused-function"
cd "$TMP/goliath_amd/csrc"
ls *.hip | xargs -P 8 -I{} sh -c "hipcc $FLAGS -c {} -o {}.o"
hipcc --offload-arch=gfx950 -shared -fPIC *.o -o "$ROOT/goliath_amd/lib/libgoliath_hip_$TAG.so"
echo "$ROOT/goliath_amd/lib/libgoliath_hip_$TAG.so  (csrc of $(git -C "$ROOT" rev-parse --short "$REV"))"
