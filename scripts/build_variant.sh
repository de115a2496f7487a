#!/bin/bash
# Build an experiment variant of libmnrf_hip.so with extra -D flags into exp_libs/<name>.so WITHOUT touching the in-tree
# objects: the sources are copied to exp_libs/_build/<name>/ and built there (a variant once left an object compiled with
# its -D flag behind in csrc/, which the next plain `make` happily linked into the default library).
# Usage: scripts/build_variant.sh <name> "-DMNRF_EXP_..."      (MNRF_LIB=exp_libs/<name>.so selects it)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME=$1; FLAGS=$2
B="$ROOT/exp_libs/_build/$NAME"
rm -rf "$B"; mkdir -p "$B/mirror_nerf_amd/csrc" "$B/include"
cp "$ROOT"/mirror_nerf_amd/csrc/*.hip "$ROOT"/mirror_nerf_amd/csrc/*.inc "$ROOT"/mirror_nerf_amd/csrc/*.h "$ROOT"/mirror_nerf_amd/csrc/Makefile "$B/mirror_nerf_amd/csrc/"
cp "$ROOT"/include/*.h "$B/include/"
make -C "$B/mirror_nerf_amd/csrc" -j8 CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function -mllvm -pragma-unroll-threshold=1000000 $FLAGS" > /dev/null
cp "$B/mirror_nerf_amd/libmnrf_hip.so" "$ROOT/exp_libs/$NAME.so"
rm -rf "$B"
echo "built exp_libs/$NAME.so with $FLAGS"
