#!/bin/bash
# build a tuning variant of libcstone_hip.so: tools/build_variant.sh NAME "-DFLAG ..."  -> cornerstone-octree_amd/lib/variants/NAME.so
# (only sort.hip is recompiled with the extra flags; the other objects come from the regular build)
set -e
cd "$(dirname "$0")/../cornerstone-octree_amd"
make -s
mkdir -p lib/variants build/variants
hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=off -Wno-unused-function -I../include -Icsrc $2 \
      -c csrc/sort.hip -o build/variants/sort_$1.o
objs="build/ctx.o build/sfc.o build/scan.o build/resort.o build/primitives.o build/tree.o build/halos.o build/neighbors.o build/groups.o build/focus.o build/extras.o build/btree.o build/domain.o build/domain_mr.o build/comm_rccl.o"
hipcc --offload-arch=gfx950 -shared -fPIC -o lib/variants/$1.so $objs build/variants/sort_$1.o -ldl
echo "built lib/variants/$1.so"
