// Single-input-channel convolutions (the discriminator's first 3-D layer, Conv3d(1, 32, 3, stride (1,2,2)),
// p2igan.py:133): their data gradient has ONE output channel and their weight gradient one x channel, so
// the MFMA tiles of the general engine would be 1/32 full.  These are K = 27, HBM/LDS-bound problems:
// plain VALU kernels.
#include "conv_common.h"

namespace p2i {

// dx[b,0,ti,hi,wi] = (sum_{o,tap valid} Wd[tap][o][0] * dy[b,o,to,ho,wo] + add) * act'(mask)
__global__ __launch_bounds__(256) void c1_dgrad_kernel(const p2i_conv_desc d, const float* __restrict__ dy, const float* __restrict__ wp_d,
                                                      const float* __restrict__ add, const float* __restrict__ mask_y, int mask_act,
                                                      float* __restrict__ dx) {
  extern __shared__ float sw[];                      // [ntaps][Cout]
  const int ntaps = d.kt * d.kh * d.kw;
  for (int i = threadIdx.x; i < ntaps * d.Cout; i += blockDim.x) sw[i] = wp_d[(size_t)i * 32];      // Ipad = 32, channel 0
  __syncthreads();
  const int HWi = d.Hi * d.Wi, HWo = d.Ho * d.Wo;
  const size_t total = (size_t)d.B * d.Ti * HWi;
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int wi = idx % d.Wi, hi = (idx / d.Wi) % d.Hi, ti = (idx / HWi) % d.Ti;
    const int b = idx / ((size_t)HWi * d.Ti);
    float acc = 0.f;
    // only the taps congruent to (i + pad) modulo the stride reach an output position: step through those
    for (int a = (ti + d.pt) % d.st; a < d.kt; a += d.st) {
      const int to = (ti + d.pt - a) / d.st;
      if (ti + d.pt - a < 0 || to >= d.To) continue;
      for (int bb = (hi + d.ph) % d.sh; bb < d.kh; bb += d.sh) {
        const int ho = (hi + d.ph - bb) / d.sh;
        if (hi + d.ph - bb < 0 || ho >= d.Ho) continue;
        for (int c = (wi + d.pw) % d.sw; c < d.kw; c += d.sw) {
          const int wo = (wi + d.pw - c) / d.sw;
          if (wi + d.pw - c < 0 || wo >= d.Wo) continue;
          const float* w = sw + ((a * d.kh + bb) * d.kw + c) * d.Cout;
          const float* g = dy + (((size_t)b * d.Cout) * d.To + to) * HWo + ho * d.Wo + wo;
          const size_t cs = (size_t)d.To * HWo;
#pragma unroll 8
          for (int o = 0; o < d.Cout; ++o) acc += w[o] * g[o * cs];
        }
      }
    }
    if (add) acc += add[idx];
    if (mask_y) acc = act_grad(acc, mask_y[idx], mask_act);
    dx[idx] = acc;
  }
}

// dWp[tap][0][o] += sum_pix x[pix*s + tap - pad] * dy[o][pix];  block = persistent over 64-pixel tiles (one output
// row segment pair), thread = (o = tid & 31, tap slot = tid >> 5), register accumulators, one atomic per (tap,o)
template <int NPW>      // output pixels per tile row (32)
__global__ __launch_bounds__(256) void c1_wgrad_kernel(const p2i_conv_desc d, const float* __restrict__ x, const float* __restrict__ dy,
                                                      float* dwp, float* dbias, int ntiles_w, int ntiles) {
  extern __shared__ float sm[];
  const int eW = (NPW - 1) * d.sw + d.kw, eH = d.sh + d.kh;          // 2 output rows per tile
  float* sx = sm;                                    // [kt][eH][eW]
  float* sy = sm + d.kt * eH * eW;                   // [Cout][2*NPW + 1]
  const int PP = 2 * NPW + 1;
  const int o = threadIdx.x & 31, slot = threadIdx.x >> 5;
  const int ntaps = d.kt * d.kh * d.kw;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  int toff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int tap = slot + 8 * j;
    const int a = tap / (d.kh * d.kw), r = tap % (d.kh * d.kw);
    toff[j] = tap < ntaps ? (a * eH + r / d.kw) * eW + r % d.kw : -1;
  }
  const int HWi = d.Hi * d.Wi, HWo = d.Ho * d.Wo;
  const int nth = (d.Ho + 1) / 2;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int tl = tile;
    const int tw = tl % ntiles_w; tl /= ntiles_w;
    const int th = tl % nth; tl /= nth;
    const int to = tl % d.To;
    const int b = tl / d.To;
    const int h0 = th * 2, w0 = tw * NPW;
    __syncthreads();
    for (int e = threadIdx.x; e < d.kt * eH * eW; e += 256) {
      const int xx = e % eW, yy = (e / eW) % eH, a = e / (eW * eH);
      const int t = to * d.st + a - d.pt, h = h0 * d.sh + yy - d.ph, w = w0 * d.sw + xx - d.pw;
      sx[e] = ((unsigned)t < (unsigned)d.Ti && (unsigned)h < (unsigned)d.Hi && (unsigned)w < (unsigned)d.Wi)
                  ? x[((size_t)b * d.Ti + t) * HWi + h * d.Wi + w] : 0.f;
    }
    for (int e = threadIdx.x; e < d.Cout * 2 * NPW; e += 256) {
      const int p = e % (2 * NPW), oc = e / (2 * NPW);
      const int h = h0 + p / NPW, w = w0 + p % NPW;
      sy[oc * PP + p] = (h < d.Ho && w < d.Wo) ? dy[(((size_t)b * d.Cout + oc) * d.To + to) * HWo + h * d.Wo + w] : 0.f;
    }
    __syncthreads();
    if (o < d.Cout) {
      const float* yo = sy + o * PP;
#pragma unroll 4
      for (int p = 0; p < 2 * NPW; ++p) {
        const float gy = yo[p];
        const int base = (p / NPW) * d.sh * eW + (p % NPW) * d.sw;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (toff[j] >= 0) acc[j] += gy * sx[base + toff[j]];
        if (slot == 0) bsum += gy;
      }
    }
  }
  if (o < d.Cout) {
    const int CoPad = (d.Cout + 31) / 32 * 32;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (toff[j] >= 0) atomicAdd(dwp + (size_t)(slot + 8 * j) * CoPad + o, acc[j]);       // Cin == 1: [tap][0][o]
    if (dbias && slot == 0) atomicAdd(dbias + o, bsum);
  }
}

int c1_dgrad(const p2i_conv_desc* d, const float* dy, const float* wp_d, const float* add, const float* mask_y, int mask_act,
             float* dx, hipStream_t s) {
  const int ntaps = d->kt * d->kh * d->kw;
  const size_t total = (size_t)d->B * d->Ti * d->Hi * d->Wi;
  const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  hipLaunchKernelGGL(c1_dgrad_kernel, dim3(grid), dim3(256), sizeof(float) * ntaps * d->Cout, s, *d, dy, wp_d, add, mask_y, mask_act, dx);
  return launch_status();
}

int c1_wgrad(const p2i_conv_desc* d, const float* x, const float* dy, float* dwp, float* dbias, hipStream_t s) {
  constexpr int NPW = 32;
  const int ntw = ceil_div(d->Wo, NPW), nth = (d->Ho + 1) / 2;
  const int ntiles = d->B * d->To * nth * ntw;
  const int eW = (NPW - 1) * d->sw + d->kw, eH = d->sh + d->kh;
  const size_t lds = sizeof(float) * ((size_t)d->kt * eH * eW + (size_t)d->Cout * (2 * NPW + 1));
  const int grid = ntiles < 1024 ? ntiles : 1024;
  hipLaunchKernelGGL(c1_wgrad_kernel<NPW>, dim3(grid), dim3(256), lds, s, *d, x, dy, dwp, dbias, ntw, ntiles);
  return launch_status();
}

}  // namespace p2i
