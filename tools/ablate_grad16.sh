#!/bin/bash
# Build experiment variants of libmarius_hip.so with -DGRAD_ABLATE=<mask> (see lp_res.hip) into build_abl/ (run in the dev container),
# then on the GPU:  for m in 0 2 4 8 12 14; do MARIUS_HIP_LIB=build_abl/libabl_$m.so python bench.py --driver py ...; done
set -e
cd "$(dirname "$0")/.."
OBJS=$(for f in marius_amd/csrc/kernels/*.hip; do b=$(basename $f .hip); [ "$b" != lp_res ] && echo marius_amd/lib/obj/$b.o; done)
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Imarius_amd/csrc/kernels -DGRAD_ABLATE=$m -c marius_amd/csrc/kernels/lp_res.hip -o build_abl/lp_res_$m.o &
done
wait
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_abl/libabl_$m.so $OBJS build_abl/lp_res_$m.o
done
ls -la build_abl/*.so
