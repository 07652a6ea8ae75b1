#!/bin/bash
# experiment builds of libmarius_hip.so with -DGB6_ABLATE=<mask> (lp_split_grad.hip) into build_abl/
set -e
cd "$(dirname "$0")/.."
OBJS=$(for f in marius_amd/csrc/kernels/*.hip; do b=$(basename $f .hip); [ "$b" != lp_split_grad ] && echo marius_amd/lib/obj/$b.o; done)
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Imarius_amd/csrc/kernels -DGB6_ABLATE=$m -c marius_amd/csrc/kernels/lp_split_grad.hip -o build_abl/lp_sg_$m.o &
done
wait
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_abl/libgb6_$m.so $OBJS build_abl/lp_sg_$m.o
done
