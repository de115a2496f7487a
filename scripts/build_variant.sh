#!/bin/bash
# Build an experiment variant of libmnrf_hip.so with extra -D flags into exp_libs/<name>.so (the default library is rebuilt
# afterwards by a plain `make`).  Usage: scripts/build_variant.sh <name> "-DMNRF_EXP_..." [object ...]   (MNRF_LIB selects it)
set -e
cd "$(dirname "$0")/../mirror_nerf_amd/csrc"
NAME=$1; FLAGS=$2; shift 2
OBJS=${@:-mnrf_field_split.o}
mkdir -p ../../exp_libs
for o in $OBJS; do rm -f $o; done
make -j8 CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function -mllvm -pragma-unroll-threshold=1000000 $FLAGS" OUT=../../exp_libs/$NAME.so > /dev/null
for o in $OBJS; do rm -f $o; done
echo "built exp_libs/$NAME.so with $FLAGS"
