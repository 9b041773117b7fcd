#!/bin/bash
# build a tuning variant of libcstone_hip.so: tools/build_variant.sh NAME "-DFLAG ..." [FILE]
#   -> cornerstone-octree_amd/lib/variants/NAME.so
# (only csrc/FILE.hip, default sort, is recompiled with the extra flags; the other objects come from the regular build)
set -e
cd "$(dirname "$0")/../cornerstone-octree_amd"
make -s
f=${3:-sort}
mkdir -p lib/variants build/variants
hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=off -Wno-unused-function -I../include -Icsrc $2 \
      -c csrc/$f.hip -o build/variants/${f}_$1.o
objs=""
for o in ctx let_ops sfc sort scan resort primitives tree halos neighbors groups focus extras btree domain domain_mr comm_rccl; do
  if [ $o = $f ]; then objs="$objs build/variants/${f}_$1.o"; else objs="$objs build/$o.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o lib/variants/$1.so $objs -ldl
echo "built lib/variants/$1.so"
