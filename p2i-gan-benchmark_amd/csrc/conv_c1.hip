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
                                                      float* dwp, float* dbias, int ntiles_w, int ntiles, DetWs ws) {
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
  if (ws.part) {                                       // ws.part = [workgroups][33 rows x 32]: rows = taps (<= 32), then the bias row
    float* pp = ws.part + (size_t)blockIdx.x * (33 * 32);
#pragma unroll
    for (int j = 0; j < 4; ++j) pp[(slot + 8 * j) * 32 + o] = (toff[j] >= 0 && o < d.Cout) ? acc[j] : 0.f;
    if (slot == 0) pp[32 * 32 + o] = o < d.Cout ? bsum : 0.f;
    return;
  }
  if (o < d.Cout) {
    const int CoPad = (d.Cout + 31) / 32 * 32;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (toff[j] >= 0) atomicAdd(dwp + (size_t)(slot + 8 * j) * CoPad + o, acc[j]);       // Cin == 1: [tap][0][o]
    if (dbias && slot == 0) atomicAdd(dbias + o, bsum);
  }
}

static bool c1_shape_ok(const p2i_conv_desc* d);
static int c1_dgrad_fast(const p2i_conv_desc* d, const float* dy, const float* wp_d, const float* add, const float* mask_y, int mask_act,
                         float* dx, hipStream_t s);
static int c1_wgrad_mfma(const p2i_conv_desc* d, const float* x, const float* dy, float* dwp, float* dbias, hipStream_t s);
static const int c1_fast_on = getenv("P2I_C1_FAST") ? atoi(getenv("P2I_C1_FAST")) : 1;

int c1_dgrad(const p2i_conv_desc* d, const float* dy, const float* wp_d, const float* add, const float* mask_y, int mask_act,
             float* dx, hipStream_t s) {
  if (c1_fast_on && c1_shape_ok(d)) {
    const int rc = c1_dgrad_fast(d, dy, wp_d, add, mask_y, mask_act, dx, s);
    if (rc != 1) return rc;
  }
  const int ntaps = d->kt * d->kh * d->kw;
  const size_t total = (size_t)d->B * d->Ti * d->Hi * d->Wi;
  const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  P2I_LAUNCH(c1_dgrad_kernel, dim3(grid), dim3(256), sizeof(float) * ntaps * d->Cout, s, *d, dy, wp_d, add, mask_y, mask_act, dx);
  return launch_status();
}

int c1_wgrad(const p2i_conv_desc* d, const float* x, const float* dy, float* dwp, float* dbias, hipStream_t s) {
  if (c1_fast_on && c1_shape_ok(d)) {
    const int rc = c1_wgrad_mfma(d, x, dy, dwp, dbias, s);
    if (rc != 1) return rc;
  }
  constexpr int NPW = 32;
  const int ntw = ceil_div(d->Wo, NPW), nth = (d->Ho + 1) / 2;
  const int ntiles = d->B * d->To * nth * ntw;
  const int eW = (NPW - 1) * d->sw + d->kw, eH = d->sh + d->kh;
  const size_t lds = sizeof(float) * ((size_t)d->kt * eH * eW + (size_t)d->Cout * (2 * NPW + 1));
  const int grid = ntiles < 1024 ? ntiles : 1024;
  const int ntaps = d->kt * d->kh * d->kw;
  // (Cout <= 32: CoPad == 32, the packed gradient is [tap][32]; deterministic mode: partials per workgroup, summed by det_reduce)
  const DetWs ws = (ntaps <= 32 && d->Cout <= 32) ? det_take((size_t)grid * 33 * 32, 0) : DetWs{nullptr, nullptr};
  P2I_LAUNCH(c1_wgrad_kernel<NPW>, dim3(grid), dim3(256), lds, s, *d, x, dy, dwp, dbias, ntw, ntiles, ws);
  if (ws.part) {
    if (int e = det_reduce(ws.part, ntaps * 32, grid, 1, 33 * 32, DetSegs{{dwp, nullptr, nullptr, nullptr}, {ntaps * 32, 0, 0, 0}}, s)) return e;
    if (dbias) return det_reduce(ws.part + 32 * 32, d->Cout, grid, 1, 33 * 32, DetSegs{{dbias, nullptr, nullptr, nullptr}, {d->Cout, 0, 0, 0}}, s);
    return P2I_OK;
  }
  return launch_status();
}

}  // namespace p2i

// =====================================================================================================================
// Fast paths for THE layer these cases come from: Conv3d(1, Cout <= 32, k 3x3x3, stride (1,2,2), pad 1) (p2igan.py:133).
// All three directions are HBM-bound (x is 1/8 of y): forward and weight gradient put the 27 taps on the MFMA's K / M side
// (one 32x32 tile: taps x channels), the data gradient is a VALU stencil over 2x2x4 output blocks.
// =====================================================================================================================
namespace p2i {

static bool c1_shape_ok(const p2i_conv_desc* d) {
  return d->Cin == 1 && d->Cout <= 32 && d->kt == 3 && d->kh == 3 && d->kw == 3 && d->st == 1 && d->sh == 2 && d->sw == 2 &&
         d->pt == 1 && d->ph == 1 && d->pw == 1;
}

constexpr int C1_ROWS = 4, C1_TW = 64, C1_EH = 2 * C1_ROWS + 1, C1_EW = 2 * C1_TW + 1;      // output tile and its x patch (odd pitch)

// x patch [3][EH][EW] of output tile (b, to, h0.., w0..): x[b, to-1+a, 2*h0-1+yy, 2*w0-1+xx], zero outside.  Split into
// "all loads to registers" and "all LDS stores" (fixed trip count) so that the 14 loads of a thread are in flight together and
// can be issued a tile ahead; the load is unconditional (clamped address) and the select follows it: a load under a per-element
// condition makes hipcc branch around it and wait for each one separately.
// Thread (g = tid >> 7, col = tid & 127) loads column `col` of the patch rows g*14 .. g*14+13 (27 rows = 3 frames x 9 rows; row
// validity is wave-uniform, the column test a per-thread constant: no div/mod in the loop); column 128 of row r is loaded by
// thread r.  (Index arithmetic with e % 129, e / 129 % 9 ... per element cost more VALU time than the whole tile's MFMAs.)
constexpr int C1_NXE = 15;
__device__ __forceinline__ void c1_load_x(const p2i_conv_desc& d, const float* __restrict__ x, float (&xr)[C1_NXE], int b, int to, int h0, int w0) {
  const int HWi = d.Hi * d.Wi;
  const int g = threadIdx.x >> 7, col = threadIdx.x & 127;
  const int w = 2 * w0 - 1 + col;
  const bool wok = (unsigned)w < (unsigned)d.Wi;
  const float* xb = x + (size_t)b * d.Ti * HWi;
#pragma unroll
  for (int i = 0; i < 14; ++i) {
    const int r = g * 14 + i;                          // patch row (a, yy); r = 27 does not exist
    const int a = r / C1_EH, yy = r - a * C1_EH;
    const int t = to - 1 + a, h = 2 * h0 - 1 + yy;
    const bool ok = r < 27 && wok && (unsigned)t < (unsigned)d.Ti && (unsigned)h < (unsigned)d.Hi;
    const float v = xb[ok ? t * HWi + h * d.Wi + w : 0];
    xr[i] = ok ? v : 0.f;
  }
  {                                                    // last column (xx = 128) of row tid
    const int r = threadIdx.x, a = r / C1_EH, yy = r - a * C1_EH;
    const int t = to - 1 + a, h = 2 * h0 - 1 + yy, wl = 2 * w0 - 1 + 128;
    const bool ok = r < 27 && (unsigned)wl < (unsigned)d.Wi && (unsigned)t < (unsigned)d.Ti && (unsigned)h < (unsigned)d.Hi;
    const float v = xb[ok ? t * HWi + h * d.Wi + wl : 0];
    xr[14] = ok ? v : 0.f;
  }
}
__device__ __forceinline__ void c1_store_x(float* sx, const float (&xr)[C1_NXE]) {
  const int g = threadIdx.x >> 7, col = threadIdx.x & 127;
#pragma unroll
  for (int i = 0; i < 14; ++i) {
    const int r = g * 14 + i;
    if (r < 27) sx[r * C1_EW + col] = xr[i];
  }
  if (threadIdx.x < 27) sx[threadIdx.x * C1_EW + 128] = xr[14];
}

// forward: y[b,o,to,ho,wo] = act(bias[o] + sum_tap w[o][tap] * patch).  MFMA: A = weights (M = o, K = tap: 14 steps of 2),
// B = patch gather (N = 32 output columns).  Wave = one output row of the tile (2 x 32 columns).
__global__ __launch_bounds__(256) void c1_fwd_kernel(const p2i_conv_desc d, const float* __restrict__ x, const float* __restrict__ wp,
                                                    const float* __restrict__ bias, float* __restrict__ y, int act, int nth, int ntw, int ntiles) {
  __shared__ float sx[3 * C1_EH * C1_EW + 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lhi = lane >> 5;
  float wa[14];
  int toff[14];
#pragma unroll
  for (int s = 0; s < 14; ++s) {
    const int tap = 2 * s + lhi;
    wa[s] = (tap < 27 && l31 < d.Cout) ? wp[tap * 32 + l31] : 0.f;          // packed [tap][Cin = 1][32]
    const int tp = tap < 27 ? tap : 26;
    toff[s] = ((tp / 9) * C1_EH + (tp / 3) % 3) * C1_EW + tp % 3;
  }
  float bo[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int o = (r & 3) + 8 * (r >> 2) + 4 * lhi;
    bo[r] = (bias && o < d.Cout) ? bias[o] : 0.f;
  }
  const size_t HWo = (size_t)d.Ho * d.Wo;
  auto decode = [&](int tile, int& b, int& to, int& h0, int& w0) {
    const int tw = tile % ntw; tile /= ntw;
    const int th = tile % nth; tile /= nth;
    to = tile % d.To; b = tile / d.To;
    h0 = th * C1_ROWS; w0 = tw * C1_TW;
  };
  float xr[C1_NXE];
  int b, to, h0, w0;
  if ((int)blockIdx.x < ntiles) { decode(blockIdx.x, b, to, h0, w0); c1_load_x(d, x, xr, b, to, h0, w0); }
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    decode(tile, b, to, h0, w0);
    __syncthreads();                                   // everybody is done reading the previous patch
    c1_store_x(sx, xr);
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) {              // next tile's patch travels while this one is multiplied and stored
      int nb_, nto, nh0, nw0;
      decode(tile + gridDim.x, nb_, nto, nh0, nw0);
      c1_load_x(d, x, xr, nb_, nto, nh0, nw0);
    }
    const int ho = h0 + wave;
    if (ho >= d.Ho) continue;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int q = nb * 32 + l31;
      const float* xp = sx + (2 * wave) * C1_EW + 2 * q;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = bo[r];
#pragma unroll
      for (int s = 0; s < 14; ++s) {
        float bv = xp[toff[s]];
        if (s == 13) bv = lhi ? 0.f : bv;                                    // tap 27 does not exist (0 * inf guard)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[s], bv, acc, 0, 0, 0);
      }
      const int wo = w0 + q;
      if (wo < d.Wo) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = (r & 3) + 8 * (r >> 2) + 4 * lhi;
          if (o < d.Cout) y[(((size_t)b * d.Cout + o) * d.To + to) * HWo + (size_t)ho * d.Wo + wo] = act_apply(acc[r], act);
        }
      }
    }
  }
}

// data gradient: thread = 4 frames x 2 x 2 pixels of dx (one output-parity quad per frame).  Per dy channel: 6 frames x 2 x 2 dy
// values feed 108 FMAs with the channel's 27 weights (LDS broadcast).  dx = (sum + add) * act'(mask).
__global__ __launch_bounds__(256) void c1_dgrad_fast_kernel(const p2i_conv_desc d, const float* __restrict__ dy, const float* __restrict__ wp_d,
                                                           const float* __restrict__ add, const float* __restrict__ mask_y, int mask_act,
                                                           float* __restrict__ dx, int nJ, int nI, int nTB) {
  __shared__ float sw[32 * 28];
  for (int i = threadIdx.x; i < 27 * d.Cout; i += 256) {
    const int tap = i / d.Cout, o = i % d.Cout;
    sw[o * 28 + tap] = wp_d[(size_t)i * 32];                                  // packed [tap][Cout][32], input channel 0
  }
  __syncthreads();
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const int j = (int)(gid % nJ);
  long long rest = gid / nJ;
  const int i = (int)(rest % nI); rest /= nI;
  const int tb = (int)(rest % nTB);
  const int b = (int)(rest / nTB);
  if (b >= d.B) return;
  const int t0 = 4 * tb;
  const size_t HWo = (size_t)d.Ho * d.Wo, cs = (size_t)d.To * HWo;
  // dy addresses of the 6 x 2 x 2 block (clamped) and their validity
  int goff[6][2][2];
  bool gok[6][2][2];
#pragma unroll
  for (int f = 0; f < 6; ++f)
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int to = t0 - 1 + f, ho = i + r, wo = j + q;
        gok[f][r][q] = (unsigned)to < (unsigned)d.To && ho < d.Ho && wo < d.Wo;
        goff[f][r][q] = gok[f][r][q] ? (int)((size_t)to * HWo + (size_t)ho * d.Wo + wo) : 0;
      }
  float acc[4][2][2];
#pragma unroll
  for (int tt = 0; tt < 4; ++tt)
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[tt][p >> 1][p & 1] = 0.f;
  const float* gb = dy + (size_t)b * d.Cout * cs;
  // channel o + 1's 24 loads are issued before channel o's 108 FMAs (unconditional loads at clamped offsets, masked after)
  float G[6][2][2], Gn[6][2][2];
#pragma unroll
  for (int f = 0; f < 6; ++f)
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int q = 0; q < 2; ++q) G[f][r][q] = gb[goff[f][r][q]];
  for (int o = 0; o < d.Cout; ++o) {
    const float* gn = gb + (size_t)(o + 1 < d.Cout ? o + 1 : o) * cs;
#pragma unroll
    for (int f = 0; f < 6; ++f)
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 2; ++q) Gn[f][r][q] = gn[goff[f][r][q]];
#pragma unroll
    for (int f = 0; f < 6; ++f)
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 2; ++q) G[f][r][q] = gok[f][r][q] ? G[f][r][q] : 0.f;
    const float* w = sw + o * 28;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      // kernel rows / columns reaching an even (ph = 0) pixel: bb = 1 (ho = i); an odd one: bb = 0 (ho = i + 1), bb = 2 (ho = i)
      const float w00 = w[a * 9 + 0], w01 = w[a * 9 + 1], w02 = w[a * 9 + 2];
      const float w10 = w[a * 9 + 3], w11 = w[a * 9 + 4], w12 = w[a * 9 + 5];
      const float w20 = w[a * 9 + 6], w21 = w[a * 9 + 7], w22 = w[a * 9 + 8];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const int f = tt + 2 - a;
        acc[tt][0][0] += w11 * G[f][0][0];
        acc[tt][0][1] += w10 * G[f][0][1] + w12 * G[f][0][0];
        acc[tt][1][0] += w01 * G[f][1][0] + w21 * G[f][0][0];
        acc[tt][1][1] += w00 * G[f][1][1] + w02 * G[f][1][0] + w20 * G[f][0][1] + w22 * G[f][0][0];
      }
    }
#pragma unroll
    for (int f = 0; f < 6; ++f)
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 2; ++q) G[f][r][q] = Gn[f][r][q];
  }
  const size_t HWi = (size_t)d.Hi * d.Wi;
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) {
    const int t = t0 + tt;
    if (t >= d.Ti) break;
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      const int h = 2 * i + ph;
      if (h >= d.Hi) continue;
#pragma unroll
      for (int pw = 0; pw < 2; ++pw) {
        const int w = 2 * j + pw;
        if (w >= d.Wi) continue;
        const size_t idx = ((size_t)b * d.Ti + t) * HWi + (size_t)h * d.Wi + w;
        float v = acc[tt][ph][pw];
        if (add) v += add[idx];
        if (mask_y) v = act_grad(v, mask_y[idx], mask_act);
        dx[idx] = v;
      }
    }
  }
}

// weight (+ bias) gradient: one 32x32 MFMA tile, rows = taps (row 27 = bias: A operand 1.0), columns = dy channels, K = output
// positions.  Persistent workgroups over (b, to, 4 rows x 64 columns) tiles; wave = one tile row; per-wave partial tiles are
// summed through LDS and added to dwp / dbias with one float atomic per element per workgroup (256-B segments: full rate).
__global__ __launch_bounds__(256) void c1_wgrad_mfma_kernel(const p2i_conv_desc d, const float* __restrict__ x, const float* __restrict__ dy,
                                                           float* dwp, float* dbias, int nth, int ntw, int ntiles, DetWs ws) {
  constexpr int PP = C1_ROWS * C1_TW + 4;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sx = sm;                                        // [3][EH][EW] + {1, 0}
  float* sy = sm + ((3 * C1_EH * C1_EW + 2 + 3) & ~3);   // [32][PP]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lhi = lane >> 5;
  if (threadIdx.x == 0) { sx[3 * C1_EH * C1_EW] = 1.f; sx[3 * C1_EH * C1_EW + 1] = 0.f; }
  // A operand address of this lane: tap row l31 (27: the constant 1, 28..31: the constant 0)
  const int tap = l31;
  const int aoff = tap < 27 ? ((tap / 9) * C1_EH + (tap / 3) % 3 + 2 * wave) * C1_EW + tap % 3 + 2 * lhi : 3 * C1_EH * C1_EW + (tap == 27 ? 0 : 1);
  const int astep = tap < 27 ? 4 : 0;
  const int boff = l31 * PP + wave * C1_TW + lhi;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const size_t HWo = (size_t)d.Ho * d.Wo;
  const bool v4 = (d.Wo & 3) == 0;
  constexpr int NYE = 32 * C1_ROWS * (C1_TW / 4) / 256;                      // float4 of the dy tile per thread
  float xr[C1_NXE];
  float4 yr[NYE];
  auto load_tile = [&](int tile) {
    const int tw = tile % ntw; tile /= ntw;
    const int th = tile % nth; tile /= nth;
    const int to = tile % d.To, b = tile / d.To;
    const int h0 = th * C1_ROWS, w0 = tw * C1_TW;
    c1_load_x(d, x, xr, b, to, h0, w0);
#pragma unroll
    for (int i = 0; i < NYE; ++i) {                                          // dy tile [o][row][64], zero outside / for o >= Cout
      const int e = i * 256 + threadIdx.x;
      const int q4 = e % (C1_TW / 4), r = (e / (C1_TW / 4)) % C1_ROWS, o = e / (C1_TW / 4 * C1_ROWS);
      const int ho = h0 + r, wo = w0 + 4 * q4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      const bool ok = o < d.Cout && ho < d.Ho && wo < d.Wo;
      const float* src = dy + (ok ? (((size_t)b * d.Cout + o) * d.To + to) * HWo + (size_t)ho * d.Wo + wo : 0);
      if (v4) {
        const float4 t4 = *reinterpret_cast<const float4*>(src);             // unconditional, then masked (see c1_load_x)
        if (ok) v = t4;
      } else if (ok) {
        v.x = src[0]; if (wo + 1 < d.Wo) v.y = src[1]; if (wo + 2 < d.Wo) v.z = src[2]; if (wo + 3 < d.Wo) v.w = src[3];
      }
      yr[i] = v;
    }
  };
  if ((int)blockIdx.x < ntiles) load_tile(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    __syncthreads();
    c1_store_x(sx, xr);
#pragma unroll
    for (int i = 0; i < NYE; ++i) {
      const int e = i * 256 + threadIdx.x;
      const int q4 = e % (C1_TW / 4), r = (e / (C1_TW / 4)) % C1_ROWS, o = e / (C1_TW / 4 * C1_ROWS);
      *reinterpret_cast<float4*>(sy + o * PP + r * C1_TW + 4 * q4) = yr[i];
    }
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);         // in flight during the MFMAs below
    const float* ap = sx + aoff;
    const float* bp = sy + boff;
#pragma unroll 8
    for (int q = 0; q < C1_TW; q += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[(q >> 1) * astep], bp[q], acc, 0, 0, 0);
  }
  // sum the four waves' tiles, then one atomic per element
  __syncthreads();
  float* red = sm;                                       // 4 x 16 x 64 floats
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
  __syncthreads();
  if (ws.part) {                                       // ws.part = [workgroups][28 x 32]: added in workgroup order by det_reduce
    if (wave == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = (red[r * 64 + lane] + red[(16 + r) * 64 + lane]) + (red[(32 + r) * 64 + lane] + red[(48 + r) * 64 + lane]);
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (row < 28) ws.part[(size_t)blockIdx.x * (28 * 32) + row * 32 + l31] = v;
      }
    }
    return;
  }
  if (wave == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = (red[r * 64 + lane] + red[(16 + r) * 64 + lane]) + (red[(32 + r) * 64 + lane] + red[(48 + r) * 64 + lane]);
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;     // tap
      if (l31 < d.Cout) {
        if (row < 27) atomicAdd(dwp + row * 32 + l31, v);   // packed [tap][Cin = 1][32]
        else if (row == 27 && dbias) atomicAdd(dbias + l31, v);
      }
    }
  }
}

int c1_fwd(const p2i_conv_desc* d, const float* x, const float* wp, const float* bias, float* y, int act, hipStream_t s) {
  if (!c1_fast_on || !c1_shape_ok(d)) return 1;
  const int nth = ceil_div(d->Ho, C1_ROWS), ntw = ceil_div(d->Wo, C1_TW);
  const long long ntiles = (long long)d->B * d->To * nth * ntw;
  if (ntiles > 0x7fffffff) return 1;
  const int grid = (int)ntiles;                               // (the kernel loops; > 1 tile per workgroup measured slower)
  P2I_LAUNCH(c1_fwd_kernel, dim3(grid), dim3(256), 0, s, *d, x, wp, bias, y, act, nth, ntw, (int)ntiles);
  return launch_status();
}

static int c1_dgrad_fast(const p2i_conv_desc* d, const float* dy, const float* wp_d, const float* add, const float* mask_y, int mask_act,
                         float* dx, hipStream_t s) {
  const int nJ = (d->Wi + 1) / 2, nI = (d->Hi + 1) / 2, nTB = (d->Ti + 3) / 4;
  const long long nthr = (long long)d->B * nTB * nI * nJ;
  if ((long long)d->To * d->Ho * d->Wo >= (1ll << 31)) return 1;
  P2I_LAUNCH(c1_dgrad_fast_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, s, *d, dy, wp_d, add, mask_y, mask_act, dx,
                     nJ, nI, nTB);
  return launch_status();
}

static int c1_wgrad_mfma(const p2i_conv_desc* d, const float* x, const float* dy, float* dwp, float* dbias, hipStream_t s) {
  const int nth = ceil_div(d->Ho, C1_ROWS), ntw = ceil_div(d->Wo, C1_TW);
  const long long ntiles = (long long)d->B * d->To * nth * ntw;
  if (ntiles > 0x7fffffff) return 1;
  constexpr int PP = C1_ROWS * C1_TW + 4;
  const size_t lds = sizeof(float) * (((3 * C1_EH * C1_EW + 2 + 3) & ~3) + 32 * PP);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)c1_wgrad_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    attr_set = true;
  }
  const int grid = ntiles < 512 ? (int)ntiles : 512;
  const DetWs ws = det_take((size_t)grid * 28 * 32, 0);
  P2I_LAUNCH(c1_wgrad_mfma_kernel, dim3(grid), dim3(256), lds, s, *d, x, dy, dwp, dbias, nth, ntw, (int)ntiles, ws);
  // taps x 32 columns (the columns >= Cout of the partials are zero: nothing of dy there), then the bias row
  if (ws.part) return det_reduce(ws.part, 27 * 32 + (dbias ? d->Cout : 0), grid, 1, 28 * 32, DetSegs{{dwp, dbias, nullptr, nullptr}, {27 * 32, dbias ? d->Cout : 0, 0, 0}}, s);
  return launch_status();
}

// ---- single-OUTPUT-channel forward (the 2-D discriminator's last layer, Conv2d(256, 1, 3, padding 1): p2igan.py:129).  One output
// channel leaves 31 of 32 MFMA rows empty in the patch-GEMM engine (50 us for 8.4 MB of input at B = 8); this is a bandwidth
// problem: x is read ONCE.  Workgroup = 2 output rows x 32 columns of one image, wave w = channels w, w+8, ...; a lane holds pixel
// (row lane >> 5, column lane & 31), loads the three vertically adjacent values of its column per channel (full 128-B rows) and gets
// the horizontal neighbours from its lane neighbours; the eight waves' partial sums meet in LDS.  Deterministic (no atomics).
__global__ __launch_bounds__(512) void o1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias,
                                                    float* __restrict__ y, int Cin, int H, int W, int nth, int ntw, int act) {
  extern __shared__ float osm[];                      // [Cin][9] weights, then [8][64] partial sums
  float* sw = osm;
  float* red = osm + ((Cin * 9 + 63) & ~63);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < Cin * 9; i += 512) {
    const int c = i / 9, tap = i - c * 9;
    sw[i] = wp[((size_t)tap * Cin + c) * 32];         // packed [tap][c][pad32(1)], output channel 0
  }
  int t = blockIdx.x;
  const int tw = t % ntw; t /= ntw;
  const int th = t % nth;
  const int b = t / nth;
  const int r = lane >> 5, col = lane & 31;
  const int h = th * 2 + r, w = tw * 32 + col;
  const bool wide = ntw > 1;                          // wave-uniform: tile edges have neighbours in the next tile
  const bool in_w = w < W;
  const bool up = h - 1 >= 0 && h - 1 < H && in_w, mid = h < H && in_w, dn = h + 1 < H && in_w;
  const int HW = H * W;
  const float* xb = x + (size_t)b * Cin * HW + (size_t)h * W + w;
  __syncthreads();
  float acc = 0.f;
  // loads of a channel are unconditional (clamped offsets, values masked afterwards) so that the loop unrolls and the loads of
  // several channels are in flight together: a conditional load makes hipcc branch around it and wait before the next one
  // (a lane below the image's last row -- odd H -- is `up` but not `mid`: it reads from the tensor base x, where a row offset of -W
  // would leave the allocation; its value only feeds pixels that are never stored)
  const int o_up = (up && mid) ? -W : 0, o_dn = (dn && mid) ? W : 0;
  const int nc = (Cin - wave + 7) >> 3;              // channels of this wave: wave, wave + 8, ...
  if (!wide) {
    const bool lz = col == 0, rz = col == 31;
    const float* xm = mid ? xb : x;                  // (a pixel of the tile outside the image reads element 0 and is masked)
    auto tap9 = [&](int c, float v0, float v1, float v2) {
      v0 = up ? v0 : 0.f; v1 = mid ? v1 : 0.f; v2 = dn ? v2 : 0.f;
      float l0 = __shfl_up(v0, 1, 32), l1 = __shfl_up(v1, 1, 32), l2 = __shfl_up(v2, 1, 32);
      float r0 = __shfl_down(v0, 1, 32), r1 = __shfl_down(v1, 1, 32), r2 = __shfl_down(v2, 1, 32);
      l0 = lz ? 0.f : l0; l1 = lz ? 0.f : l1; l2 = lz ? 0.f : l2;
      r0 = rz ? 0.f : r0; r1 = rz ? 0.f : r1; r2 = rz ? 0.f : r2;
      const float* k = sw + c * 9;
      acc += k[0] * l0 + k[1] * v0 + k[2] * r0 + k[3] * l1 + k[4] * v1 + k[5] * r1 + k[6] * l2 + k[7] * v2 + k[8] * r2;
    };
    // eight channels per trip, their 24 loads issued together (hipcc does not unroll this loop by itself: one channel per
    // trip means one exposed memory latency per channel)
    int ci = 0;
    for (; ci + 8 <= nc; ci += 8) {
      float v[8][3];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float* xc = xm + (size_t)(wave + 8 * (ci + u)) * HW;
        v[u][0] = xc[o_up]; v[u][1] = xc[0]; v[u][2] = xc[o_dn];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) tap9(wave + 8 * (ci + u), v[u][0], v[u][1], v[u][2]);
    }
    for (; ci < nc; ++ci) {
      const float* xc = xm + (size_t)(wave + 8 * ci) * HW;
      tap9(wave + 8 * ci, xc[o_up], xc[0], xc[o_dn]);
    }
  } else {
    // several column tiles: every lane reads its own left and right neighbours (columns w - 1, w + 1 of the image)
    const bool lok = w > 0 && w - 1 < W, rok = w + 1 < W;
    const int h_c = h < H ? h : H - 1;
    const float* xr = x + (size_t)b * Cin * HW + (size_t)h_c * W;      // row start (clamped row)
    const int wl = lok ? w - 1 : 0, wm = in_w ? w : 0, wr = rok ? w + 1 : 0;
    const bool hu = h - 1 >= 0 && h - 1 < H, hm = h < H, hd = h + 1 < H;
    const int ou = (hu && hm) ? -W : 0, od = hd ? W : 0;      // (row h == H is clamped to H - 1: no row above it when H == 1)
#pragma unroll 2
    for (int ci = 0; ci < nc; ++ci) {
      const int c = wave + 8 * ci;
      const float* xc = xr + (size_t)c * HW;
      const float* k = sw + c * 9;
      const float a0 = xc[ou + wl], a1 = xc[ou + wm], a2 = xc[ou + wr], b0 = xc[wl], b1 = xc[wm], b2 = xc[wr];
      const float c0 = xc[od + wl], c1 = xc[od + wm], c2 = xc[od + wr];
      acc += k[0] * (hu && lok ? a0 : 0.f) + k[1] * (hu && in_w ? a1 : 0.f) + k[2] * (hu && rok ? a2 : 0.f) +
             k[3] * (hm && lok ? b0 : 0.f) + k[4] * (hm && in_w ? b1 : 0.f) + k[5] * (hm && rok ? b2 : 0.f) +
             k[6] * (hd && lok ? c0 : 0.f) + k[7] * (hd && in_w ? c1 : 0.f) + k[8] * (hd && rok ? c2 : 0.f);
    }
  }
  red[wave * 64 + lane] = acc;
  __syncthreads();
  if (wave == 0) {
    float v = bias ? bias[0] : 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) v += red[q * 64 + lane];
    if (mid) y[(size_t)b * HW + (size_t)h * W + w] = act_apply(v, act);
  }
}

// 1 = not a case of this kernel
int o1_fwd(const p2i_conv_desc* d, const float* x, const float* wp, const float* bias, float* y, int act, hipStream_t s) {
  static const int on = getenv("P2I_O1_FWD") ? atoi(getenv("P2I_O1_FWD")) : 1;
  if (!on || d->Cout != 1 || d->kt != 1 || d->kh != 3 || d->kw != 3 || d->st != 1 || d->sh != 1 || d->sw != 1 || d->pt != 0 || d->ph != 1 ||
      d->pw != 1 || d->Ti != 1 || d->Cin < 8 || d->Cin > 2048)
    return 1;
  const int nth = ceil_div(d->Ho, 2), ntw = ceil_div(d->Wo, 32);
  const long long nt = (long long)d->B * nth * ntw;
  if (nt > 0x7fffffff) return 1;
  const size_t lds = sizeof(float) * (size_t)(((d->Cin * 9 + 63) & ~63) + 8 * 64);
  P2I_LAUNCH(o1_fwd_kernel, dim3((unsigned)nt), dim3(512), lds, s, x, wp, bias, y, d->Cin, d->Hi, d->Wi, nth, ntw, act);
  return launch_status();
}

}  // namespace p2i
