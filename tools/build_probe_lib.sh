#!/bin/bash
# Tuning build of the library with the fused-MLP cycle stamps (-DSM_PROBE) -> tools/libpvn3d_probe.so
# (used by tools/mlp_probe.py through PVN3D_HIP_LIB; never the shipped library).
set -e
cd "$(dirname "$0")/../pvn3d_amd/csrc"
make -s -j8
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DSM_PROBE -c sa_mlp.hip -o /tmp/sa_mlp_probe.o
# split-bf16 chains with the PVN3D_S3_DBG switches and cycle stamps (tools/s3_prof.py, tools/s3_time.py)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DPVN3D_S3_TUNING -c sa_mlp_split.hip -o /tmp/sa_mlp_split_probe.o
# multi-wave FPS with cycle counters per phase of a round (tools/fps_prof.py)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-inline-asm -DPVN3D_FC_PROF -c fps_cells.hip -o /tmp/fps_cells_probe.o
OBJS=$(ls *.o | grep -v '^sa_mlp.o$' | grep -v '^sa_mlp_split.o$' | grep -v '^fps_cells.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libpvn3d_probe.so $OBJS /tmp/sa_mlp_probe.o /tmp/sa_mlp_split_probe.o /tmp/fps_cells_probe.o
echo built tools/libpvn3d_probe.so
