#!/bin/bash
# A/B build of the library with the straightforward fp16 x 2 split (-DPVN3D_SPLIT_PLAIN: v_cvt_f32_f16 + v_pk_add_f32 +
# v_cvt_pk_f16_f32 instead of v_fma_mixlo/hi_f16, csrc/common.h) -> tools/ab/libplain.so, for tools/split_ab.py.
set -e
cd "$(dirname "$0")/../pvn3d_amd/csrc"
make -s -j8
mkdir -p ../../tools/ab
OBJS=$(ls *.o | grep -v '^split_gemm.o$' | grep -v '^sa_mlp_split.o$')
for f in split_gemm sa_mlp_split; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DPVN3D_SPLIT_PLAIN -c $f.hip -o /tmp/${f}_plain.o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/libplain.so $OBJS /tmp/split_gemm_plain.o /tmp/sa_mlp_split_plain.o
echo built tools/ab/libplain.so
