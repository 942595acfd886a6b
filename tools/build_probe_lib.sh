#!/bin/bash
# Tuning build of the library with the fused-MLP cycle stamps (-DSM_PROBE) -> tools/libpvn3d_probe.so
# (used by tools/mlp_probe.py through PVN3D_HIP_LIB; never the shipped library).
set -e
cd "$(dirname "$0")/../pvn3d_amd/csrc"
make -s -j8
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DSM_PROBE -c sa_mlp.hip -o /tmp/sa_mlp_probe.o
OBJS=$(ls *.o | grep -v '^sa_mlp.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libpvn3d_probe.so $OBJS /tmp/sa_mlp_probe.o
echo built tools/libpvn3d_probe.so
