// Shared helpers for libp2i_hip (gfx950 only: wave64, MFMA f32 32x32x2).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "p2i_hip.h"
#include "tape.h"

namespace p2i {

void set_error(const char* fmt, ...);

#define P2I_REQUIRE(cond, ...)                  \
  do {                                          \
    if (!(cond)) {                              \
      p2i::set_error(__VA_ARGS__);              \
      return P2I_EINVAL;                        \
    }                                           \
  } while (0)

static inline int launch_status() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("launch failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return P2I_OK;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int pow2_ceil(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}
static inline int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}
// exact floor(n/d) for 0 <= n < 65536, 1 <= d < 65536 via one mul_hi
static inline unsigned magic_u16(int d) { return (unsigned)(0x100000000ull / (unsigned)d) + 1u; }
__device__ __forceinline__ int fast_div(int n, unsigned magic) { return (int)__umulhi((unsigned)n, magic); }

__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    case P2I_ACT_RELU: return v > 0.f ? v : 0.f;
    case P2I_ACT_LEAKY: return v > 0.f ? v : 0.2f * v;
    case P2I_ACT_TANH: return tanhf(v);
    default: return v;
  }
}
// d act / d pre-activation expressed through the saved post-activation y
__device__ __forceinline__ float act_grad(float g, float y, int act) {
  switch (act) {
    case P2I_ACT_RELU: return y > 0.f ? g : 0.f;
    case P2I_ACT_LEAKY: return y > 0.f ? g : 0.2f * g;
    case P2I_ACT_TANH: return g * (1.f - y * y);
    default: return g;
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// block-wide sum; `red` is >= 16 floats of LDS; result valid in every thread
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float s = 0.f;
  for (int i = 0; i < nw; ++i) s += red[i];
  return s;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float s = red[0];
  for (int i = 1; i < nw; ++i) s = fmaxf(s, red[i]);
  return s;
}

// ---- deterministic cross-workgroup sums.  The small reductions of the step (bias / position / attention-weight / alpha gradients, a
// few dot products) used to end in float atomics, whose order -- hence whose rounding -- changes from run to run.  With a caller-owned
// scratch (p2i_det_workspace) they are summed in a FIXED order instead, in two stages: the kernel's workgroups STORE their partials,
// det_reduce (one more small launch) adds them in workgroup order.  (First version, measured and dropped: the last workgroup to
// finish added the partials inside the same launch.  Its __threadfence() is an L2 write-back on this multi-die part -- every workgroup
// paid for flushing what the kernel had just written: act_bwd_bias 12 -> 126 us, the step +1.9 ms.)
struct DetWs {
  float* part;          // partial sums (layout: per kernel); nullptr: no scratch registered -> float atomics
  unsigned* counter;    // (reserved)
};
// host: this call's share of the scratch registered with p2i_det_workspace ({nullptr, nullptr}: none registered, or too small --
// the kernels then keep their float atomics)
DetWs det_take(size_t floats, int counters);
// out[g] += sum_{k < n} part[g * gs + k * ks] for g < groups, k ascending.  Up to four output segments: group g of segment i
// (lengths seg_len, consecutive g) goes to seg_out[i][g - start_i].
struct DetSegs { float* out[4]; int len[4]; };
int det_reduce(const float* part, int groups, int n, long long gs, long long ks, const DetSegs& segs, hipStream_t s);

}  // namespace p2i
