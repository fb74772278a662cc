#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC, int WPS>
__global__ __launch_bounds__(256 * WPS) void k(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
  }
  float s = 0.f;
  for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, int WPS> void run(const char* name) {
  float* out; hipMalloc(&out, 4 * 256 * WPS * 1024);
  const int iters = 4000, blocks = 256 * (WPS == 1 ? 2 : 1);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC, WPS><<<blocks, 256 * WPS>>>(out, 10, 1.f, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0); k<NACC, WPS><<<blocks, 256 * WPS>>>(out, iters, 0.5f, 0.25f); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double fl = (double)blocks * (256 * WPS / 64) * iters * NACC * 4096.0;
  printf("%s: %.2f ms  %.1f TFLOP/s\n", name, ms, fl / ms / 1e9);
}
int main() {
  run<4, 1>("4 acc, 4 waves/block x2 blocks/CU (2 waves/SIMD)");
  run<9, 1>("9 acc, 2 waves/SIMD");
  run<9, 2>("9 acc, 8-wave blocks (2 waves/SIMD)");
  run<1, 1>("1 acc (dependent chain), 2 waves/SIMD");
  return 0;
}
