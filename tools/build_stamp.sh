#!/bin/bash
# Diagnostic build of libp2i_hip with in-kernel s_memtime stamps in the fwd/dgrad chunk loop (-DP2I_STAMP): build/ab/libp2i_hip_stamp.so
# Used with P2I_HIP_LIB=... by tools/stamp_conv.py.  Never shipped: its fences forbid overlaps the real kernel has (read shares, not times).
set -e
cd "$(dirname "$0")/../p2i-gan-benchmark_amd/csrc"
B=../../build/stamp; mkdir -p $B ../../build/ab
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -I../../include -I. -DP2I_STAMP -fgpu-rdc"
for f in conv conv_dma_g0 conv_dma_g1 conv_dma_g2 conv_dma_g3 conv_fused conv_x6c wgrad wgrad_x6 conv_c1 weights glue idw loss metrics tape; do
  EX=""; [ $f = idw ] && EX="-ffp-contract=off"
  /opt/rocm/bin/hipcc $FLAGS $EX -c $f.hip -o $B/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fgpu-rdc --hip-link -shared -fPIC $B/*.o -o ../../build/ab/libp2i_hip_stamp.so
echo built build/ab/libp2i_hip_stamp.so
