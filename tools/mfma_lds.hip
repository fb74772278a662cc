#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
// mode 0: pipelined like wgrad (loads for next half before MFMAs, sched_barrier), mode 1: no sched barriers (compiler free)
template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, int stride) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 1e-3f * i;
  __syncthreads();
  f32x16 acc[9];
  for (int t = 0; t < 9; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int lane = threadIdx.x & 63;
  const float* xa = lds + (lane & 31) * 35 + (lane >> 5);
  const float* yb = lds + 8192 + (lane & 31) * 68;
  int toff[9];
  for (int t = 0; t < 9; ++t) toff[t] = (t / 3) * stride + (t % 3);
  float a0[9], a1[9];
  float4 v = *reinterpret_cast<const float4*>(yb), vn = v;
  for (int t = 0; t < 9; ++t) a0[t] = xa[toff[t]];
  for (int i = 0; i < iters; ++i) {
    const float* xp = xa + (i & 7) * 4;
#pragma unroll
    for (int t = 0; t < 9; ++t) a1[t] = xp[toff[t] + 2];
    if (MODE == 0) __builtin_amdgcn_sched_barrier(0);
    const float b0 = (lane >> 5) ? v.y : v.x, b1 = (lane >> 5) ? v.w : v.z;
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b0, acc[t], 0, 0, 0);
    if (MODE == 0) __builtin_amdgcn_sched_barrier(0);
    const float* xn = xa + ((i + 1) & 7) * 4;
#pragma unroll
    for (int t = 0; t < 9; ++t) a0[t] = xn[toff[t]];
    vn = *reinterpret_cast<const float4*>(yb + ((i + 1) & 7) * 4);
    if (MODE == 0) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b1, acc[t], 0, 0, 0);
    if (MODE == 0) __builtin_amdgcn_sched_barrier(0);
    v = vn;
  }
  float s = 0.f;
  for (int t = 0; t < 9; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name) {
  float* out; (void)hipMalloc(&out, 4 * 512 * 256);
  const int iters = 2000, blocks = 256;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  k<MODE><<<blocks, 512, 100 * 1024>>>(out, 10, 2240);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); k<MODE><<<blocks, 512, 100 * 1024>>>(out, iters, 2240); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double fl = (double)blocks * 8 * iters * 18 * 4096.0;
  printf("%s: %.2f ms  %.1f TFLOP/s\n", name, ms, fl / ms / 1e9);
}
int main() { run<0>("wgrad-like loop, sched_barrier pipelined"); run<1>("wgrad-like loop, compiler scheduled"); return 0; }
