#!/bin/bash
# Tuning builds of the library with pieces of the LDS-DMA split GEMM switched off (-DPVN3D_SG_DBG=n: 1 no MFMAs, 2 no
# stores, 4 no operand loads) -> tools/sgv/libsg_<n>.so, for `PVN3D_HIP_LIB=tools/sgv/libsg_<n>.so python tools/sg_time.py`.
set -e
cd "$(dirname "$0")/../pvn3d_amd/csrc"
make -s -j8
mkdir -p ../../tools/sgv
OBJS=$(ls *.o | grep -v '^split_gemm.o$')
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DPVN3D_SG_DBG=$n -c split_gemm.hip -o /tmp/split_gemm_dbg$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/sgv/libsg_$n.so $OBJS /tmp/split_gemm_dbg$n.o
done
echo built "$@"
