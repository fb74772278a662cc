// Weight preparation for the conv engine: DO-Conv fold (deconv_pytorch.py:111-127) and its
// backward, plain <-> packed weight layouts, spectral-norm power iteration
// (torch.nn.utils.spectral_norm; call sites layer.py:402-407, p2igan.py:141).
#include <stdarg.h>
#include "common.h"

namespace p2i {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

__device__ __forceinline__ int pad32(int v) { return (v + 31) & ~31; }

// ---- DO-Conv fold.  One thread per flat pair q = o'*I + i of the (O/g, I, 9) view; a block owns a
// 16 (o') x 16 (i) tile so that BOTH packed layouts are written in 64-B runs (wp_f: o contiguous,
// wp_d: cin contiguous) via an LDS transpose.  groups == 1 only; grouped / 1x1 layers use the simple kernel.
constexpr int FOLD_MAX_LAYERS = 16;
// same-shape layers of one launch (blockIdx.z = layer): the fold kernels are launch-latency bound (8-12 us each)
struct FoldBatch {
  const float* W[FOLD_MAX_LAYERS];
  const float* D[FOLD_MAX_LAYERS];
  const float* Dd[FOLD_MAX_LAYERS];
  float* a[FOLD_MAX_LAYERS];          // fwd: wp_f   bwd: dW
  float* b[FOLD_MAX_LAYERS];          // fwd: wp_d   bwd: dD
  const float* g[FOLD_MAX_LAYERS];    // bwd: packed weight gradient
};

__device__ __forceinline__ void fold_fwd_tile_body(const float* __restrict__ W, const float* __restrict__ D,
                                                   const float* __restrict__ Dd, int O, int I, float* wp_f, float* wp_d) {
  __shared__ float tile[9][16][17];
  const int ti = threadIdx.x & 15, to = threadIdx.x >> 4;     // thread computes (o' = o0+to, i = i0+ti): W reads 36-B runs
  const int o0 = blockIdx.y * 16, i0 = blockIdx.x * 16;
  const int o = o0 + to, i = i0 + ti;
  if (o < O && i < I) {
    float w[9];
    const float* wq = W + ((size_t)o * I + i) * 9;
#pragma unroll
    for (int s2 = 0; s2 < 9; ++s2) w[s2] = wq[s2];
    const float* d = D + (size_t)i * 81;
    const float* dd = Dd + (size_t)i * 81;
#pragma unroll
    for (int m = 0; m < 9; ++m) {
      float acc = 0.f;
#pragma unroll
      for (int s2 = 0; s2 < 9; ++s2) acc += (d[m * 9 + s2] + dd[m * 9 + s2]) * w[s2];
      tile[m][to][ti] = acc;
    }
  }
  __syncthreads();
  const int a = threadIdx.x & 15, b = threadIdx.x >> 4;
#pragma unroll
  for (int m = 0; m < 9; ++m) {
    // wp_f[m][i][o]: lanes along o
    if (i0 + b < I && o0 + a < O) wp_f[((size_t)m * I + i0 + b) * O + o0 + a] = tile[m][a][b];
    // wp_d[m][o][i]: lanes along i
    if (wp_d && o0 + b < O && i0 + a < I) wp_d[((size_t)m * O + o0 + b) * I + i0 + a] = tile[m][b][a];
  }
}
__global__ __launch_bounds__(256) void fold_fwd_tile_kernel(const float* __restrict__ W, const float* __restrict__ D,
                                                           const float* __restrict__ Dd, int O, int I, float* wp_f, float* wp_d) {
  fold_fwd_tile_body(W, D, Dd, O, I, wp_f, wp_d);
}
// 32 x 32 (o, i) tile for O, I multiples of 32 (the batched path).  The 16 x 16 body above reads D + D_diag (2 x 81 floats) from
// global memory in EVERY (o, i) thread -- 162 of its 171 loads -- and stores 64-byte runs: 130 us for the eight 512-channel layers,
// 1.7 TB/s.  Here D + D_diag of the tile's 32 input channels is summed once into LDS (conflict-free: 81 floats per channel, odd
// pitch), a thread folds FOUR output channels against one read of it, and both packed layouts are written in full 128-byte lines.
__global__ __launch_bounds__(256) void fold_fwd_tile_batched_kernel(const FoldBatch fb, int O, int I) {
  __shared__ float dsum[32][81];
  __shared__ float tile[9][32][33];
  const int L = blockIdx.z;
  const float* __restrict__ W = fb.W[L];
  const float* __restrict__ D = fb.D[L];
  const float* __restrict__ Dd = fb.Dd[L];
  float* wp_f = fb.a[L];
  float* wp_d = fb.b[L];
  const int o0 = blockIdx.y * 32, i0 = blockIdx.x * 32;
  for (int e = threadIdx.x; e < 32 * 81; e += 256) (&dsum[0][0])[e] = D[(size_t)i0 * 81 + e] + Dd[(size_t)i0 * 81 + e];
  const int ti = threadIdx.x & 31, to = threadIdx.x >> 5;     // i = i0 + ti; o = o0 + to + 8 k, k = 0..3
  float w[4][9];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float* wq = W + ((size_t)(o0 + to + 8 * k) * I + i0 + ti) * 9;
#pragma unroll
    for (int s2 = 0; s2 < 9; ++s2) w[k][s2] = wq[s2];
  }
  __syncthreads();
#pragma unroll
  for (int m = 0; m < 9; ++m) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s2 = 0; s2 < 9; ++s2) {
      const float d = dsum[ti][m * 9 + s2];
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] += d * w[k][s2];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) tile[m][to + 8 * k][ti] = acc[k];
  }
  __syncthreads();
  const int a = threadIdx.x & 31, b = threadIdx.x >> 5;
#pragma unroll
  for (int m = 0; m < 9; ++m)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      wp_f[((size_t)m * I + i0 + b + 8 * k) * O + o0 + a] = tile[m][a][b + 8 * k];           // wp_f[m][i][o]: lanes along o
      if (wp_d) wp_d[((size_t)m * O + o0 + b + 8 * k) * I + i0 + a] = tile[m][b + 8 * k][a];  // wp_d[m][o][i]: lanes along i
    }
}

__global__ void fold_fwd_kernel(const float* __restrict__ W, const float* __restrict__ D, const float* __restrict__ Dd,
                                int O, int I, int groups, int ksz, int identity_rep, float* wp_f, float* wp_d) {
  const int Ig = I / groups, Og = O / groups;
  const int Opad = pad32(O), Ipad = pad32(I);
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (ksz == 1) {                       // DoW = W.reshape(O, I/g, 1, 1)
    if (q >= O * Ig) return;
    const int o = q / Ig, ci = q - o * Ig;
    const int cin = (o / Og) * Ig + ci;
    const float v = W[q];
    wp_f[(size_t)cin * Opad + o] = v;
    if (wp_d) wp_d[(size_t)o * Ipad + cin] = v;
    return;
  }
  if (q >= Og * I) return;
  const int i = q % I;                  // index into D (I, 9, 9)
  const int o = q / Ig, ci = q - o * Ig;   // memory reinterpretation (O/g, I, 9) -> (O, I/g, 3, 3)
  const int cin = (o / Og) * Ig + ci;
  float w[9];
#pragma unroll
  for (int s = 0; s < 9; ++s) w[s] = W[(size_t)q * 9 + s];
#pragma unroll
  for (int m = 0; m < 9; ++m) {
    float acc = 0.f;
#pragma unroll
    for (int s = 0; s < 9; ++s) acc += (D[(i * 9 + m) * 9 + s] + Dd[(i * 9 + m) * 9 + s]) * w[s];
    if (identity_rep > 0 && m == 4 && cin == o / identity_rep) acc += 1.f;   // + x.repeat_interleave(rep,1), p2igan.py:79
    wp_f[((size_t)m * I + cin) * Opad + o] = acc;
    if (wp_d) wp_d[((size_t)m * O + o) * Ipad + cin] = acc;
  }
}

// dW[o'][i][s] = sum_m dDoW[o',i,m] (D+Dd)[i][m][s] for groups == 1: dDoW tile gathered with o-contiguous reads
__device__ __forceinline__ void fold_bwd_w_tile_body(const float* __restrict__ dwp, const float* __restrict__ D,
                                                     const float* __restrict__ Dd, int O, int I, float* dW) {
  __shared__ float tile[9][16][17];
  const int o0 = blockIdx.y * 16, i0 = blockIdx.x * 16;
  const int a = threadIdx.x & 15, b = threadIdx.x >> 4;
#pragma unroll
  for (int m = 0; m < 9; ++m)
    tile[m][a][b] = (i0 + b < I && o0 + a < O) ? dwp[((size_t)m * I + i0 + b) * O + o0 + a] : 0.f;    // [m][o][i]
  __syncthreads();
  const int ti = threadIdx.x & 15, to = threadIdx.x >> 4;
  const int o = o0 + to, i = i0 + ti;
  if (o >= O || i >= I) return;
  float g[9];
#pragma unroll
  for (int m = 0; m < 9; ++m) g[m] = tile[m][to][ti];
  const float* d = D + (size_t)i * 81;
  const float* dd = Dd + (size_t)i * 81;
  float* out = dW + ((size_t)o * I + i) * 9;
#pragma unroll
  for (int s2 = 0; s2 < 9; ++s2) {
    float acc = 0.f;
#pragma unroll
    for (int m = 0; m < 9; ++m) acc += g[m] * (d[m * 9 + s2] + dd[m * 9 + s2]);
    out[s2] = acc;
  }
}
__global__ __launch_bounds__(256) void fold_bwd_w_tile_kernel(const float* __restrict__ dwp, const float* __restrict__ D,
                                                             const float* __restrict__ Dd, int O, int I, float* dW) {
  fold_bwd_w_tile_body(dwp, D, Dd, O, I, dW);
}
// 32 x 32 (o, i) tile, D + D_diag of the tile's input channels summed once into LDS, four output channels per thread (same
// restructuring as fold_fwd_tile_batched_kernel; O, I multiples of 32)
__global__ __launch_bounds__(256) void fold_bwd_w_tile_batched_kernel(const FoldBatch fb, int O, int I) {
  __shared__ float dsum[32][81];
  __shared__ float tile[9][32][33];
  const int L = blockIdx.z;
  const float* __restrict__ dwp = fb.g[L];
  const float* __restrict__ D = fb.D[L];
  const float* __restrict__ Dd = fb.Dd[L];
  float* dW = fb.a[L];
  const int o0 = blockIdx.y * 32, i0 = blockIdx.x * 32;
  for (int e = threadIdx.x; e < 32 * 81; e += 256) (&dsum[0][0])[e] = D[(size_t)i0 * 81 + e] + Dd[(size_t)i0 * 81 + e];
  const int a = threadIdx.x & 31, b = threadIdx.x >> 5;
#pragma unroll
  for (int m = 0; m < 9; ++m)
#pragma unroll
    for (int k = 0; k < 4; ++k) tile[m][a][b + 8 * k] = dwp[((size_t)m * I + i0 + b + 8 * k) * O + o0 + a];    // [m][o][i], lanes along o
  __syncthreads();
  const int ti = threadIdx.x & 31, to = threadIdx.x >> 5;     // i = i0 + ti; o = o0 + to + 8 k
  float g[4][9], out[4][9];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int m = 0; m < 9; ++m) { g[k][m] = tile[m][to + 8 * k][ti]; out[k][m] = 0.f; }
#pragma unroll
  for (int m = 0; m < 9; ++m)
#pragma unroll
    for (int s2 = 0; s2 < 9; ++s2) {
      const float d = dsum[ti][m * 9 + s2];
#pragma unroll
      for (int k = 0; k < 4; ++k) out[k][s2] += g[k][m] * d;
    }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float* o = dW + ((size_t)(o0 + to + 8 * k) * I + i0 + ti) * 9;
#pragma unroll
    for (int s2 = 0; s2 < 9; ++s2) o[s2] = out[k][s2];
  }
}

// dW (O/g, I, 9) from packed dDoW: dW[q][s] = sum_m dDoW[q][m] * (D+Dd)[i][m][s]
__global__ void fold_bwd_w_kernel(const float* __restrict__ dwp, const float* __restrict__ D, const float* __restrict__ Dd,
                                  int O, int I, int groups, int ksz, float* dW) {
  const int Ig = I / groups, Og = O / groups, Opad = pad32(O);
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (ksz == 1) {
    if (q >= O * Ig) return;
    const int o = q / Ig, ci = q - o * Ig;
    dW[q] = dwp[(size_t)((o / Og) * Ig + ci) * Opad + o];
    return;
  }
  if (q >= Og * I) return;
  const int i = q % I, o = q / Ig, ci = q - o * Ig, cin = (o / Og) * Ig + ci;
  float g[9];
#pragma unroll
  for (int m = 0; m < 9; ++m) g[m] = dwp[((size_t)m * I + cin) * Opad + o];
#pragma unroll
  for (int s = 0; s < 9; ++s) {
    float acc = 0.f;
#pragma unroll
    for (int m = 0; m < 9; ++m) acc += g[m] * (D[(i * 9 + m) * 9 + s] + Dd[(i * 9 + m) * 9 + s]);
    dW[(size_t)q * 9 + s] = acc;
  }
}

// dD[i][m][s] = sum_{o'} dDoW[o',i,m] * W[o',i,s]; block per i, threads stride over o' (coalesced dDoW reads),
// 81 register accumulators per thread, wave shuffle + LDS combine
__device__ __forceinline__ void fold_bwd_d_body(const float* __restrict__ dwp, const float* __restrict__ W, int O, int I,
                                                int groups, float* dD) {
  __shared__ float part[4][81];
  const int i = blockIdx.x;
  const int Ig = I / groups, Og = O / groups, Opad = pad32(O);
  float acc[81];
#pragma unroll
  for (int k = 0; k < 81; ++k) acc[k] = 0.f;
  for (int op = threadIdx.x; op < Og; op += blockDim.x) {
    const int q = op * I + i, o = q / Ig, ci = q - o * Ig, cin = (o / Og) * Ig + ci;
    float gm[9], ws[9];
#pragma unroll
    for (int m = 0; m < 9; ++m) gm[m] = dwp[((size_t)m * I + cin) * Opad + o];
#pragma unroll
    for (int s2 = 0; s2 < 9; ++s2) ws[s2] = W[(size_t)q * 9 + s2];
#pragma unroll
    for (int m = 0; m < 9; ++m)
#pragma unroll
      for (int s2 = 0; s2 < 9; ++s2) acc[m * 9 + s2] += gm[m] * ws[s2];
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // butterfly over the wave, one exchange STAGE for all 81 sums at a time: 81 independent ds_bpermute in flight per
  // wait instead of 486 serialised exchange->wait->add steps (that chain was 20 of this kernel's 21 us)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    float t[81];
#pragma unroll
    for (int k = 0; k < 81; ++k) t[k] = __shfl_xor(acc[k], off, 64);
#pragma unroll
    for (int k = 0; k < 81; ++k) acc[k] += t[k];
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 81; ++k) part[wave][k] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x < 81) dD[i * 81 + threadIdx.x] = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
}
__global__ __launch_bounds__(256) void fold_bwd_d_kernel(const float* __restrict__ dwp, const float* __restrict__ W, int O, int I,
                                                        int groups, float* dD) {
  fold_bwd_d_body(dwp, W, O, I, groups, dD);
}
// dD for 32 input channels per block (O, I multiples of 32, groups 1).  The per-channel body above reads W in 36-byte pieces one
// cache line apart (104 us for the eight 512-channel layers).  Here dDoW tiles [9][32 i][32 o] are staged through LDS with lanes
// along o, W is read with lanes along i (both coalesced), thread = (input channel, one of 8 output-channel slices) keeps the 81
// sums in registers over all of O, and the 8 slices are combined through LDS at the end.
__global__ __launch_bounds__(256) void fold_bwd_d_batched_kernel(const FoldBatch fb, int O, int I) {
  __shared__ float tile[9][32][33];                    // [m][i][o]
  const int L = blockIdx.z;
  const float* __restrict__ dwp = fb.g[L];
  const float* __restrict__ W = fb.W[L];
  float* dD = fb.b[L];
  const int i0 = blockIdx.x * 32;
  const int ti = threadIdx.x & 31, sl = threadIdx.x >> 5;       // channel i0 + ti, output slice sl: o = o0 + sl + 8 k
  float acc[81];
#pragma unroll
  for (int k = 0; k < 81; ++k) acc[k] = 0.f;
  for (int o0 = 0; o0 < O; o0 += 32) {
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 9; ++m)
#pragma unroll
      for (int k = 0; k < 4; ++k) tile[m][sl + 8 * k][ti] = dwp[((size_t)m * I + i0 + sl + 8 * k) * O + o0 + ti];   // lanes along o
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int ol = sl + 8 * k;
      float gm[9], ws[9];
#pragma unroll
      for (int m = 0; m < 9; ++m) gm[m] = tile[m][ti][ol];
      const float* wq = W + ((size_t)(o0 + ol) * I + i0 + ti) * 9;
#pragma unroll
      for (int s2 = 0; s2 < 9; ++s2) ws[s2] = wq[s2];
#pragma unroll
      for (int m = 0; m < 9; ++m)
#pragma unroll
        for (int s2 = 0; s2 < 9; ++s2) acc[m * 9 + s2] += gm[m] * ws[s2];
    }
  }
  // combine the 8 slices: [slice][i][81] floats = 83 KB would not fit beside the tile; do it in 3 passes of 27 sums through the tile
  float* red = &tile[0][0][0];                         // 9 * 32 * 33 = 9504 floats >= 8 * 32 * 27 = 6912
#pragma unroll
  for (int pss = 0; pss < 3; ++pss) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 27; ++k) red[(sl * 32 + ti) * 27 + k] = acc[pss * 27 + k];
    __syncthreads();
    for (int e = threadIdx.x; e < 32 * 27; e += 256) {
      const int il = e / 27, k = e - il * 27;
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) v += red[(q * 32 + il) * 27 + k];
      dD[(size_t)(i0 + il) * 81 + pss * 27 + k] = v;
    }
  }
}

// ---- plain (O, I, NT) <-> packed
__global__ void pack_kernel(const float* __restrict__ w, int O, int I, int NT, const float* div_ptr, float* wp_f, float* wp_d) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= O * I * NT) return;
  const int tap = idx % NT, i = (idx / NT) % I, o = idx / (NT * I);
  float v = w[idx];
  if (div_ptr) v = v / *div_ptr;
  if (wp_f) wp_f[((size_t)tap * I + i) * pad32(O) + o] = v;
  if (wp_d) wp_d[((size_t)tap * O + o) * pad32(I) + i] = v;
}

__global__ void unpack_dot_kernel(const float* __restrict__ dwp, const float* __restrict__ w, int O, int I, int NT, float* dot, DetWs ws) {
  __shared__ float red[16];
  float acc = 0.f;
  const int n = O * I * NT;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
    const int tap = idx % NT, i = (idx / NT) % I, o = idx / (NT * I);
    acc += dwp[((size_t)tap * I + i) * pad32(O) + o] * w[idx];
  }
  acc = block_sum(acc, red);
  if (ws.part) {
    if (threadIdx.x == 0) ws.part[blockIdx.x] = acc;
    return;
  }
  if (threadIdx.x == 0) atomicAdd(dot, acc);
}

__global__ void unpack_kernel(const float* __restrict__ dwp, int O, int I, int NT, const float* sigma_ptr, const float* dot,
                              const float* __restrict__ u, const float* __restrict__ v, float* dw) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= O * I * NT) return;
  const int tap = idx % NT, i = (idx / NT) % I, o = idx / (NT * I);
  float g = dwp[((size_t)tap * I + i) * pad32(O) + o];
  if (sigma_ptr) {
    const float sg = *sigma_ptr;
    g = g / sg - (*dot / (sg * sg)) * u[o] * v[i * NT + tap];
  }
  dw[idx] = g;
}

// ---- spectral norm
// t[k] += sum_{o in slab} W[o][k] u[o]; grid (K/64, O/32); block 256 = 64 columns x 4 row groups (t zeroed by caller)
__global__ __launch_bounds__(256) void sn_wtu_kernel(const float* __restrict__ w, const float* __restrict__ u, int O, int K, float* t) {
  __shared__ float part[4][64];
  const int col = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + col, o0 = blockIdx.y * 32 + rg * 8;
  float acc = 0.f;
  if (k < K) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int o = o0 + r;
      if (o < O) acc += w[(size_t)o * K + k] * u[o];
    }
  }
  part[rg][col] = acc;
  __syncthreads();
  if (rg == 0 && k < K) atomicAdd(t + k, part[0][col] + part[1][col] + part[2][col] + part[3][col]);
}
__global__ void sn_wv_kernel(const float* __restrict__ w, const float* __restrict__ v, int O, int K, float* s) {
  const int o = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (o >= O) return;
  float acc = 0.f;
  for (int k = lane; k < K; k += 64) acc += w[(size_t)o * K + k] * v[k];
  acc = wave_sum(acc);
  if (lane == 0) s[o] = acc;
}
// out = in / max(||in||, eps); optionally sigma = sum(out * in) (= u^T W v with the NEW u)
__global__ void sn_normalize_kernel(const float* __restrict__ in, int n, float* out, float* sigma) {
  __shared__ float red[16];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) acc += in[i] * in[i];
  const float nrm = fmaxf(sqrtf(block_sum(acc, red)), 1e-12f);
  float dot = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float o = in[i] / nrm;
    out[i] = o;
    dot += o * in[i];
  }
  if (sigma) {
    dot = block_sum(dot, red);
    if (threadIdx.x == 0) *sigma = dot;
  }
}
// ---- the same power iteration for ALL spectral-norm layers of a network in 4 launches (blockIdx.z = layer) instead of
//      4-5 launches per layer: these kernels are launch-latency bound (4-8 us each, 120 of them per training step)
constexpr int SN_MAX_LAYERS = 16;
struct SnBatch {
  const float* w[SN_MAX_LAYERS];
  float* u[SN_MAX_LAYERS];
  float* v[SN_MAX_LAYERS];
  float* sigma[SN_MAX_LAYERS];
  float* scratch[SN_MAX_LAYERS];     // >= O + K + 4 floats each: t[K] then sv[O]
  float* usnap[SN_MAX_LAYERS];       // optional copies of the UPDATED u / v (what backward needs; the next forward
  float* vsnap[SN_MAX_LAYERS];       // call overwrites u, v in place)
  int O[SN_MAX_LAYERS], K[SN_MAX_LAYERS];
  int n;
};
// t[k] = sum_o W[o][k] u[o]: block = 64 columns x 4 row groups, each group strides over all rows (no atomics, no memset)
__global__ __launch_bounds__(256) void sn_wtu_batched_kernel(const SnBatch b) {
  __shared__ float part[4][64];
  const int L = blockIdx.z, O = b.O[L], K = b.K[L];
  if ((int)blockIdx.x * 64 >= K) return;
  const float* w = b.w[L];
  const float* u = b.u[L];
  const int col = threadIdx.x & 63, rg = threadIdx.x >> 6, k = blockIdx.x * 64 + col;
  float acc = 0.f;
  if (k < K)
    for (int o = rg; o < O; o += 4) acc += w[(size_t)o * K + k] * u[o];
  part[rg][col] = acc;
  __syncthreads();
  if (rg == 0 && k < K) b.scratch[L][k] = part[0][col] + part[1][col] + part[2][col] + part[3][col];
}
__global__ void sn_wv_batched_kernel(const SnBatch b) {
  const int L = blockIdx.z, O = b.O[L], K = b.K[L];
  const int o = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (o >= O) return;
  const float* w = b.w[L];
  const float* v = b.v[L];
  float acc = 0.f;
  for (int k = lane; k < K; k += 64) acc += w[(size_t)o * K + k] * v[k];
  acc = wave_sum(acc);
  if (lane == 0) b.scratch[L][K + o] = acc;
}
// phase 0: v = normalize(t);  phase 1: u = normalize(sv), sigma = u . sv;  phase 2 (eval): sigma = u . sv
__global__ void sn_finish_batched_kernel(const SnBatch b, int phase) {
  __shared__ float red[16];
  const int L = blockIdx.z, O = b.O[L], K = b.K[L];
  const float* in = phase == 0 ? b.scratch[L] : b.scratch[L] + K;
  const int n = phase == 0 ? K : O;
  if (phase == 2) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc += b.u[L][i] * in[i];
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) *b.sigma[L] = acc;
    return;
  }
  float* out = phase == 0 ? b.v[L] : b.u[L];
  float* snap = phase == 0 ? b.vsnap[L] : b.usnap[L];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) acc += in[i] * in[i];
  const float nrm = fmaxf(sqrtf(block_sum(acc, red)), 1e-12f);
  float dot = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float o = in[i] / nrm;
    out[i] = o;
    if (snap) snap[i] = o;
    dot += o * in[i];
  }
  if (phase == 1) {
    dot = block_sum(dot, red);
    if (threadIdx.x == 0) *b.sigma[L] = dot;
  }
}
__global__ void sn_dot_kernel(const float* __restrict__ a, const float* __restrict__ b, int n, float* out) {
  __shared__ float red[16];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) acc += a[i] * b[i];
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) *out = acc;
}

}  // namespace p2i
using namespace p2i;

extern "C" int p2i_abi_version(void) { return 1; }
extern "C" const char* p2i_last_error(void) { return p2i::g_err; }

extern "C" int p2i_doconv_fold_fwd(const float* W, const float* D, const float* D_diag, int O, int I, int groups, int ksz,
                                   int identity_rep, float* wp_f, float* wp_d, void* stream) {
  P2I_REQUIRE(W && wp_f, "null pointer");
  P2I_REQUIRE(ksz == 1 || ksz == 3, "DO-Conv kernel size must be 1 or 3");
  P2I_REQUIRE(groups >= 1 && O % groups == 0 && I % groups == 0, "channels not divisible by groups");
  P2I_REQUIRE(ksz == 1 || (D && D_diag), "3x3 DO-Conv needs D and D_diag");
  hipStream_t s = (hipStream_t)stream;
  const int nt = ksz * ksz, Opad = (O + 31) / 32 * 32, Ipad = (I + 31) / 32 * 32;
  if (groups > 1 || Opad != O) (void)p2i::memset_async(wp_f, 0, sizeof(float) * (size_t)nt * I * Opad, s);
  if (wp_d && (groups > 1 || Ipad != I)) (void)p2i::memset_async(wp_d, 0, sizeof(float) * (size_t)nt * O * Ipad, s);
  if (ksz == 3 && groups == 1 && identity_rep == 0 && Opad == O && Ipad == I) {
    P2I_LAUNCH(fold_fwd_tile_kernel, dim3(ceil_div(I, 16), ceil_div(O, 16)), dim3(256), 0, s, W, D, D_diag, O, I, wp_f, wp_d);
    return launch_status();
  }
  const int n = (ksz == 1) ? O * (I / groups) : (O / groups) * I;
  P2I_LAUNCH(fold_fwd_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, W, D, D_diag, O, I, groups, ksz, identity_rep, wp_f, wp_d);
  return launch_status();
}

extern "C" int p2i_doconv_fold_bwd(const float* dwp_f, const float* W, const float* D, const float* D_diag, int O, int I,
                                   int groups, int ksz, float* dW, float* dD, void* stream) {
  P2I_REQUIRE(dwp_f && W && dW, "null pointer");
  P2I_REQUIRE(ksz == 1 || (D && D_diag && dD), "3x3 DO-Conv needs D, D_diag, dD");
  hipStream_t s = (hipStream_t)stream;
  const int n = (ksz == 1) ? O * (I / groups) : (O / groups) * I;
  if (ksz == 3 && groups == 1 && O % 32 == 0)
    P2I_LAUNCH(fold_bwd_w_tile_kernel, dim3(ceil_div(I, 16), ceil_div(O, 16)), dim3(256), 0, s, dwp_f, D, D_diag, O, I, dW);
  else
    P2I_LAUNCH(fold_bwd_w_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, dwp_f, D, D_diag, O, I, groups, ksz, dW);
  if (ksz == 3) P2I_LAUNCH(fold_bwd_d_kernel, dim3(I), dim3(256), 0, s, dwp_f, W, O, I, groups, dD);
  return launch_status();
}

extern "C" int p2i_weight_pack(const float* w, int O, int I, int ntaps, const float* inv_div_ptr, float* wp_f, float* wp_d,
                               void* stream) {
  P2I_REQUIRE(w && (wp_f || wp_d), "null pointer");
  hipStream_t s = (hipStream_t)stream;
  const int Opad = (O + 31) / 32 * 32, Ipad = (I + 31) / 32 * 32;
  if (wp_f && Opad != O) (void)p2i::memset_async(wp_f, 0, sizeof(float) * (size_t)ntaps * I * Opad, s);
  if (wp_d && Ipad != I) (void)p2i::memset_async(wp_d, 0, sizeof(float) * (size_t)ntaps * O * Ipad, s);
  P2I_LAUNCH(pack_kernel, dim3(ceil_div(O * I * ntaps, 256)), dim3(256), 0, s, w, O, I, ntaps, inv_div_ptr, wp_f, wp_d);
  return launch_status();
}

extern "C" int p2i_weight_unpack_grad(const float* dwp_f, int O, int I, int ntaps, const float* w_orig, const float* sigma_ptr,
                                      const float* u, const float* v, float* scratch, float* dw, void* stream) {
  P2I_REQUIRE(dwp_f && dw, "null pointer");
  P2I_REQUIRE(!sigma_ptr || (w_orig && u && v && scratch), "spectral-norm unpack needs w_orig, u, v, scratch");
  hipStream_t s = (hipStream_t)stream;
  const int n = O * I * ntaps;
  if (sigma_ptr) {
    (void)p2i::memset_async(scratch, 0, sizeof(float), s);
    const int nbd = min(ceil_div(n, 256), 256);
    const DetWs ws = det_take((size_t)nbd, 0);
    P2I_LAUNCH(unpack_dot_kernel, dim3(nbd), dim3(256), 0, s, dwp_f, w_orig, O, I, ntaps, scratch, ws);
    if (ws.part)
      if (int e = det_reduce(ws.part, 1, nbd, 0, 1, DetSegs{{scratch, nullptr, nullptr, nullptr}, {1, 0, 0, 0}}, s)) return e;
  }
  P2I_LAUNCH(unpack_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, dwp_f, O, I, ntaps, sigma_ptr, scratch, u, v, dw);
  return launch_status();
}

extern "C" int p2i_spectral_norm(const float* w, int O, int K, float* u, float* v, int training, float* sigma, float* scratch,
                                 void* stream) {
  P2I_REQUIRE(w && u && v && sigma && scratch, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  float* t = scratch;        // K
  float* sv = scratch + K;   // O
  if (training) {
    (void)p2i::memset_async(t, 0, sizeof(float) * K, s);
    P2I_LAUNCH(sn_wtu_kernel, dim3(ceil_div(K, 64), ceil_div(O, 32)), dim3(256), 0, s, w, u, O, K, t);
    P2I_LAUNCH(sn_normalize_kernel, dim3(1), dim3(1024), 0, s, t, K, v, (float*)nullptr);
    P2I_LAUNCH(sn_wv_kernel, dim3(ceil_div(O, 4)), dim3(256), 0, s, w, v, O, K, sv);
    P2I_LAUNCH(sn_normalize_kernel, dim3(1), dim3(1024), 0, s, sv, O, u, sigma);
  } else {
    P2I_LAUNCH(sn_wv_kernel, dim3(ceil_div(O, 4)), dim3(256), 0, s, w, v, O, K, sv);
    P2I_LAUNCH(sn_dot_kernel, dim3(1), dim3(1024), 0, s, u, sv, O, sigma);
  }
  return launch_status();
}

extern "C" int p2i_spectral_norm_batched(const float* const* w, const int* O, const int* K, float* const* u, float* const* v,
                                         int training, float* const* sigma, float* const* scratch, float* const* u_snap,
                                         float* const* v_snap, int n, void* stream) {
  P2I_REQUIRE(w && O && K && u && v && sigma && scratch && n >= 1 && n <= SN_MAX_LAYERS, "1..%d layers", SN_MAX_LAYERS);
  SnBatch b;
  int maxO = 0, maxK = 0;
  for (int i = 0; i < SN_MAX_LAYERS; ++i) {
    const int j = i < n ? i : 0;
    P2I_REQUIRE(w[j] && u[j] && v[j] && sigma[j] && scratch[j] && O[j] > 0 && K[j] > 0, "bad spectral-norm layer %d", j);
    b.w[i] = w[j]; b.u[i] = u[j]; b.v[i] = v[j]; b.sigma[i] = sigma[j]; b.scratch[i] = scratch[j]; b.O[i] = O[j]; b.K[i] = K[j];
    b.usnap[i] = (training && u_snap) ? u_snap[j] : nullptr; b.vsnap[i] = (training && v_snap) ? v_snap[j] : nullptr;
    if (O[j] > maxO) maxO = O[j];
    if (K[j] > maxK) maxK = K[j];
  }
  b.n = n;
  hipStream_t s = (hipStream_t)stream;
  if (training) {
    P2I_LAUNCH(sn_wtu_batched_kernel, dim3(ceil_div(maxK, 64), 1, n), dim3(256), 0, s, b);
    P2I_LAUNCH(sn_finish_batched_kernel, dim3(1, 1, n), dim3(1024), 0, s, b, 0);
    P2I_LAUNCH(sn_wv_batched_kernel, dim3(ceil_div(maxO, 4), 1, n), dim3(256), 0, s, b);
    P2I_LAUNCH(sn_finish_batched_kernel, dim3(1, 1, n), dim3(1024), 0, s, b, 1);
  } else {
    P2I_LAUNCH(sn_wv_batched_kernel, dim3(ceil_div(maxO, 4), 1, n), dim3(256), 0, s, b);
    P2I_LAUNCH(sn_finish_batched_kernel, dim3(1, 1, n), dim3(1024), 0, s, b, 2);
  }
  return launch_status();
}

// n <= 16 same-shape 3x3 DO-Conv layers (groups 1, O and I multiples of 32: the generator's residual stack) in ONE launch
extern "C" int p2i_doconv_fold_fwd_batched(const float* const* W, const float* const* D, const float* const* D_diag, int n, int O, int I,
                                           float* const* wp_f, float* const* wp_d, void* stream) {
  P2I_REQUIRE(W && D && D_diag && wp_f && n >= 1 && n <= FOLD_MAX_LAYERS, "1..%d layers", FOLD_MAX_LAYERS);
  P2I_REQUIRE(O > 0 && I > 0 && O % 32 == 0 && I % 32 == 0, "batched fold needs O, I multiples of 32");
  FoldBatch fb{};
  for (int i = 0; i < n; ++i) {
    P2I_REQUIRE(W[i] && D[i] && D_diag[i] && wp_f[i], "null pointer (layer %d)", i);
    fb.W[i] = W[i]; fb.D[i] = D[i]; fb.Dd[i] = D_diag[i]; fb.a[i] = wp_f[i]; fb.b[i] = wp_d ? wp_d[i] : nullptr;
  }
  P2I_LAUNCH(fold_fwd_tile_batched_kernel, dim3(I / 32, O / 32, n), dim3(256), 0, (hipStream_t)stream, fb, O, I);
  return launch_status();
}
extern "C" int p2i_doconv_fold_bwd_batched(const float* const* dwp_f, const float* const* W, const float* const* D,
                                           const float* const* D_diag, int n, int O, int I, float* const* dW, float* const* dD,
                                           void* stream) {
  P2I_REQUIRE(dwp_f && W && D && D_diag && dW && dD && n >= 1 && n <= FOLD_MAX_LAYERS, "1..%d layers", FOLD_MAX_LAYERS);
  P2I_REQUIRE(O > 0 && I > 0 && O % 32 == 0 && I % 32 == 0, "batched fold needs O, I multiples of 32");
  FoldBatch fb{};
  for (int i = 0; i < n; ++i) {
    P2I_REQUIRE(dwp_f[i] && W[i] && D[i] && D_diag[i] && dW[i] && dD[i], "null pointer (layer %d)", i);
    fb.g[i] = dwp_f[i]; fb.W[i] = W[i]; fb.D[i] = D[i]; fb.Dd[i] = D_diag[i]; fb.a[i] = dW[i]; fb.b[i] = dD[i];
  }
  hipStream_t s = (hipStream_t)stream;
  P2I_LAUNCH(fold_bwd_w_tile_batched_kernel, dim3(I / 32, O / 32, n), dim3(256), 0, s, fb, O, I);
  P2I_LAUNCH(fold_bwd_d_batched_kernel, dim3(I / 32, 1, n), dim3(256), 0, s, fb, O, I);
  return launch_status();
}

// ---- pack / unpack of up to 16 layers of DIFFERENT shapes in one launch each (blockIdx.z = layer): the discriminator's ten
//      spectral-norm layers are packed once per forward and unpacked once per backward
namespace p2i {
constexpr int PACK_MAX_LAYERS = 16;
struct PackBatch {
  const float* w[PACK_MAX_LAYERS];
  const float* div[PACK_MAX_LAYERS];
  float* f[PACK_MAX_LAYERS];
  float* d[PACK_MAX_LAYERS];
  const float* g[PACK_MAX_LAYERS];     // unpack: packed gradient
  const float* u[PACK_MAX_LAYERS];
  const float* v[PACK_MAX_LAYERS];
  float* dot[PACK_MAX_LAYERS];
  int O[PACK_MAX_LAYERS], I[PACK_MAX_LAYERS], NT[PACK_MAX_LAYERS];
  int accumulate;                      // unpack: dw += instead of dw =
};
__global__ void pack_batched_kernel(const PackBatch b) {
  const int L = blockIdx.z, O = b.O[L], I = b.I[L], NT = b.NT[L];
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < O * I * NT; idx += gridDim.x * blockDim.x) {
    const int tap = idx % NT, i = (idx / NT) % I, o = idx / (NT * I);
    float v = b.w[L][idx];
    if (b.div[L]) v = v / *b.div[L];
    if (b.f[L]) b.f[L][((size_t)tap * I + i) * pad32(O) + o] = v;
    if (b.d[L]) b.d[L][((size_t)tap * O + o) * pad32(I) + i] = v;
  }
}
__global__ void unpack_dot_batched_kernel(const PackBatch b, DetWs ws) {
  __shared__ float red[16];
  const int L = blockIdx.z, O = b.O[L], I = b.I[L], NT = b.NT[L];
  if (!b.div[L]) return;                              // plain unpack: no sigma term
  float acc = 0.f;
  const int n = O * I * NT;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
    const int tap = idx % NT, i = (idx / NT) % I, o = idx / (NT * I);
    acc += b.g[L][((size_t)tap * I + i) * pad32(O) + o] * b.w[L][idx];
  }
  acc = block_sum(acc, red);
  if (ws.part) {                                       // ws.part = [layers][gridDim.x]: added in workgroup order by det_reduce
    if (threadIdx.x == 0) ws.part[(size_t)L * gridDim.x + blockIdx.x] = acc;
    return;
  }
  if (threadIdx.x == 0 && acc != 0.f) atomicAdd(b.dot[L], acc);
}
__global__ void unpack_batched_kernel(const PackBatch b) {
  const int L = blockIdx.z, O = b.O[L], I = b.I[L], NT = b.NT[L];
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < O * I * NT; idx += gridDim.x * blockDim.x) {
    const int tap = idx % NT, i = (idx / NT) % I, o = idx / (NT * I);
    float g = b.g[L][((size_t)tap * I + i) * pad32(O) + o];
    if (b.div[L]) {
      const float sg = *b.div[L];
      g = g / sg - (*b.dot[L] / (sg * sg)) * b.u[L][o] * b.v[L][i * NT + tap];
    }
    if (b.accumulate) b.f[L][idx] += g; else b.f[L][idx] = g;
  }
}
}  // namespace p2i

extern "C" int p2i_weight_pack_batched(const float* const* w, const int* O, const int* I, const int* ntaps, const float* const* inv_div,
                                       float* const* wp_f, float* const* wp_d, int n, void* stream) {
  P2I_REQUIRE(w && O && I && ntaps && (wp_f || wp_d) && n >= 1 && n <= PACK_MAX_LAYERS, "1..%d layers", PACK_MAX_LAYERS);
  PackBatch b{};
  int maxn = 0;
  for (int i = 0; i < n; ++i) {
    P2I_REQUIRE(w[i] && O[i] > 0 && I[i] > 0 && ntaps[i] > 0, "bad layer %d", i);
    b.w[i] = w[i]; b.div[i] = inv_div ? inv_div[i] : nullptr; b.f[i] = wp_f ? wp_f[i] : nullptr; b.d[i] = wp_d ? wp_d[i] : nullptr;
    b.O[i] = O[i]; b.I[i] = I[i]; b.NT[i] = ntaps[i];
    if (O[i] * I[i] * ntaps[i] > maxn) maxn = O[i] * I[i] * ntaps[i];
  }
  const int blocks = ceil_div(maxn, 256);
  P2I_LAUNCH(pack_batched_kernel, dim3(blocks > 512 ? 512 : blocks, 1, n), dim3(256), 0, (hipStream_t)stream, b);
  return launch_status();
}

extern "C" int p2i_weight_unpack_grad_batched_acc(const float* const* dwp_f, const int* O, const int* I, const int* ntaps,
                                                  const float* const* w_orig, const float* const* sigma, const float* const* u,
                                                  const float* const* v, float* dots, float* const* dw, int n, int accumulate,
                                                  void* stream) {
  P2I_REQUIRE(dwp_f && O && I && ntaps && dw && dots && n >= 1 && n <= PACK_MAX_LAYERS, "1..%d layers", PACK_MAX_LAYERS);
  PackBatch b{};
  b.accumulate = accumulate;
  int maxn = 0;
  bool any_sigma = false;
  for (int i = 0; i < n; ++i) {
    P2I_REQUIRE(dwp_f[i] && dw[i] && O[i] > 0 && I[i] > 0 && ntaps[i] > 0, "bad layer %d", i);
    const bool sg = sigma && sigma[i];
    P2I_REQUIRE(!sg || (w_orig && w_orig[i] && u && u[i] && v && v[i]), "spectral-norm unpack needs w_orig, u, v (layer %d)", i);
    b.g[i] = dwp_f[i]; b.f[i] = dw[i]; b.div[i] = sg ? sigma[i] : nullptr; b.w[i] = sg ? w_orig[i] : nullptr;
    b.u[i] = sg ? u[i] : nullptr; b.v[i] = sg ? v[i] : nullptr; b.dot[i] = dots + i;
    b.O[i] = O[i]; b.I[i] = I[i]; b.NT[i] = ntaps[i];
    any_sigma = any_sigma || sg;
    if (O[i] * I[i] * ntaps[i] > maxn) maxn = O[i] * I[i] * ntaps[i];
  }
  hipStream_t s = (hipStream_t)stream;
  const int blocks = ceil_div(maxn, 256);
  if (any_sigma) {
    (void)p2i::memset_async(dots, 0, sizeof(float) * n, s);
    const int nbd = blocks > 128 ? 128 : blocks;
    const DetWs ws = det_take((size_t)nbd * n, 0);
    if (ws.part) (void)p2i::memset_async(ws.part, 0, sizeof(float) * (size_t)nbd * n, s);     // (layers without sigma store nothing)
    P2I_LAUNCH(unpack_dot_batched_kernel, dim3(nbd, 1, n), dim3(256), 0, s, b, ws);
    if (ws.part)
      if (int e = det_reduce(ws.part, n, nbd, nbd, 1, DetSegs{{dots, nullptr, nullptr, nullptr}, {n, 0, 0, 0}}, s)) return e;
  }
  P2I_LAUNCH(unpack_batched_kernel, dim3(blocks > 512 ? 512 : blocks, 1, n), dim3(256), 0, s, b);
  return launch_status();
}

extern "C" int p2i_weight_unpack_grad_batched(const float* const* dwp_f, const int* O, const int* I, const int* ntaps,
                                              const float* const* w_orig, const float* const* sigma, const float* const* u,
                                              const float* const* v, float* dots, float* const* dw, int n, void* stream) {
  return p2i_weight_unpack_grad_batched_acc(dwp_f, O, I, ntaps, w_orig, sigma, u, v, dots, dw, n, 0, stream);
}
