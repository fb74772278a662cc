// Calibration: what the chip sustains on the INNER LOOP SHAPE of a bf16-split ("x6") convolution: per 16-deep k-step of a
// 64x64 wave tile, 3+3 operand planes x 2 tiles = 12 ds_read_b128 and 6 products x 4 tiles = 24 v_mfma_f32_32x32x16_bf16,
// operands random bf16 in LDS, reads of step s+1 interleaved with the MFMAs of step s.  Prints fp32-equivalent TFLOP/s
// (= bf16 flops / 6).   hipcc --offload-arch=gfx950 -O3 tools/mfma_x6_lds.hip -o tools/bin/mfma_x6_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int WAVES, bool ILV>
__global__ __launch_bounds__(WAVES * 64) void k(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned lds[];
  unsigned seed = threadIdx.x * 2654435761u + 12345u;
  for (int i = threadIdx.x; i < 24576; i += blockDim.x) {          // 96 KB of random bf16 pairs in [-1,1)
    seed = seed * 1664525u + 1013904223u;
    const unsigned a = 0x3F000000u | ((seed >> 9) & 0x007F0000u) | (seed & 0x80000000u);
    seed = seed * 1664525u + 1013904223u;
    const unsigned b = 0x3F000000u | ((seed >> 9) & 0x007F0000u) | (seed & 0x80000000u);
    lds[i] = (a >> 16) | (b & 0xFFFF0000u);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // A image: [plane 3][tile 2][lane 64] x 16 B, B image likewise, shifted per step (8 steps ring)
  const bf16x8* base = reinterpret_cast<const bf16x8*>(lds) + lane + (wave & 3) * 64;
  bf16x8 a[3][2], b[3][2], an[3][2], bn[3][2];
  auto load = [&](int s, bf16x8 (&A)[3][2], bf16x8 (&B)[3][2]) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        A[p][t] = base[((s & 7) * 12 + p * 2 + t) * 64 * 4 / 4];
        B[p][t] = base[((s & 7) * 12 + 6 + p * 2 + t) * 64 * 4 / 4];
      }
  };
  auto mm = [&](bf16x8 (&A)[3][2], bf16x8 (&B)[3][2]) {
    // hh, hm, mh, hl, lh, mm
    const int pa[6] = {0, 0, 1, 0, 2, 1}, pb[6] = {0, 1, 0, 2, 0, 1};
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[pa[q]][i], B[pb[q]][j], acc[i][j], 0, 0, 0);
  };
  load(0, a, b);
  for (int s = 0; s < iters; s += 2) {
    load(s + 1, an, bn);
    mm(a, b);
    if (ILV) {
#pragma unroll
      for (int i = 0; i < 12; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
    }
    __builtin_amdgcn_sched_barrier(0);
    load(s + 2, a, b);
    mm(an, bn);
    if (ILV) {
#pragma unroll
      for (int i = 0; i < 12; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float sacc = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sacc;
}
template <int WAVES, bool ILV> void run(const char* name) {
  float* out; (void)hipMalloc(&out, 4 * 512 * 256);
  const int iters = 4000, blocks = 256;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipFuncSetAttribute((const void*)k<WAVES, ILV>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  k<WAVES, ILV><<<blocks, WAVES * 64, 100 * 1024>>>(out, 16);
  (void)hipDeviceSynchronize();
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0); k<WAVES, ILV><<<blocks, WAVES * 64, 100 * 1024>>>(out, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)blocks * WAVES * iters * 24 * 32768.0;
    printf("%s: %.2f ms  %.0f bf16 TFLOP/s = %.0f fp32-equivalent TFLOP/s\n", name, ms, fl / ms / 1e9, fl / ms / 1e9 / 6);
  }
}
int main() {
  run<4, true>("4 waves/CU, reads interleaved");
  run<8, true>("8 waves/CU, reads interleaved");
  run<8, false>("8 waves/CU, reads clustered");
  return 0;
}
