#!/bin/bash
# Builds benchmarks/mlp3_stamps/libfv3hip_stamps.so: libfv3hip.so with the split-bf16 kernel's per-phase cycle stamps compiled in
# (-DMLP3_STAMPS).  Needs the objects of a normal build (make -C fv3net_amd/csrc).  Extra flags are passed to hipcc.
set -e
here=$(cd "$(dirname "$0")" && pwd)
cd "$here/../../fv3net_amd/csrc"
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-value -Wno-invalid-offsetof -Wno-inline-asm \
      -mllvm -pragma-unroll-threshold=262144 -DMLP3_STAMPS "$@" -c mlp_bf16x3.hip -o /tmp/mlp3_stamps.o
hipcc --offload-arch=gfx950 -shared -fPIC -o "$here/libfv3hip_stamps.so" capi.o coarsen.o vertical.o remap.o mlp.o /tmp/mlp3_stamps.o emulation.o local.o fit.o
