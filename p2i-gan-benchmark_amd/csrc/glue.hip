// HBM-bound glue kernels of the P2I-GAN hot path: AttentionBlock (layer.py:296-304),
// DownsampleDuplicateChannels (layer.py:205-214), UPPos upsample+modulation (layer.py:392-396),
// discriminator tail (p2igan.py:165-173), bias gradient, axpy, Adam (train.py:125-136).
#include "common.h"

namespace p2i {

// ------------------------------------------------------------------ AttentionBlock x2
// x (B,T,HW): thread = one pixel, the T=16 vector lives in registers.
template <int T>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w0,
                                                       const float* __restrict__ b0, const float* __restrict__ w1,
                                                       const float* __restrict__ b1, float* out, int B, int HW) {
  __shared__ float sw[2][T * T + T];
  for (int i = threadIdx.x; i < T * T; i += blockDim.x) { sw[0][i] = w0[i]; sw[1][i] = w1[i]; }
  for (int i = threadIdx.x; i < T; i += blockDim.x) { sw[0][T * T + i] = b0[i]; sw[1][T * T + i] = b1[i]; }
  __syncthreads();
  const int p = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (p >= HW) return;
  float h[T];
#pragma unroll
  for (int t = 0; t < T; ++t) h[t] = x[((size_t)b * T + t) * HW + p];
#pragma unroll
  for (int l = 0; l < 2; ++l) {
    float n[T];
#pragma unroll
    for (int i = 0; i < T; ++i) {
      float g = sw[l][T * T + i];
#pragma unroll
      for (int j = 0; j < T; ++j) g += sw[l][i * T + j] * h[j];
      const float a = h[i] + h[i] * g;
      n[i] = a > 0.f ? a : 0.f;
    }
#pragma unroll
    for (int i = 0; i < T; ++i) h[i] = n[i];
  }
#pragma unroll
  for (int t = 0; t < T; ++t) out[((size_t)b * T + t) * HW + p] = h[t];
}

// parameter gradients only (the block's input is data).  Pixels whose dout is all zero (every non-gauge pixel: the IDW scatter
// touches gauge voxels only) are skipped.  A pixel's contribution is two rank-1 updates (dg (x) h, T x T each) + two bias rows.
// Round 1 let each active LANE add its 2 (T*T + T) products into LDS with atomics (~550 dependent LDS atomics per lane, 122 us per
// launch with ~1 active pixel per block); rounds 2-3 listed the active pixels' four T-vectors in LDS and let every thread own
// parameter elements and sum over the list -- but each active lane still ran the whole per-pixel chain (two T x T mat-vecs forward,
// one backward) on nine T-vectors in registers: at T = 32 that spilled 760 VGPRs to 3 KB of scratch per lane.
// Now the per-pixel chain is COOPERATIVE too: the block's active pixels are compacted (ballot ranks), and in rounds of ATTN_LIST
// pixels thread (m, i) computes element i of pixel m's vectors -- row i of each mat-vec from LDS, the T-vectors exchanged through the
// list -- so a thread holds a handful of scalars whatever T is.  Same summation order per element as before (bias first, then j = 0 ..
// T-1), a fixed order inside the block, one global atomic per parameter and block at the end (blocks without an active pixel leave).
constexpr int ATTN_LIST = 32;

// Round 4: a workgroup walks SEVERAL 256-pixel chunks of its sample (grid.x chunks-strided) and keeps summing into the same
// registers, so a launch ends with grid.x * B partial vectors instead of one per chunk (512 at B = 8): with a scratch (ws.part)
// each workgroup stores its partial and det_reduce adds them in workgroup order -- deterministic, and a sum over a few dozen
// vectors; without one the partials are added with float atomics as before.
template <int T>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w0,
                                                       const float* __restrict__ b0, const float* __restrict__ w1,
                                                       const float* __restrict__ b1, const float* __restrict__ dout,
                                                       float* dw0, float* db0, float* dw1, float* db1, int B, int HW, DetWs ws) {
  constexpr int NP = T * T + T;                        // parameters per layer: weight rows then bias
  constexpr int NE = (2 * NP + 255) / 256;             // parameter elements per thread
  constexpr int PPT = (ATTN_LIST * T + 255) / 256;     // (listed pixel, element) pairs per thread: 4 / 2 / 1 at T = 32 / 16 / 8
  constexpr int H0 = 0, H1 = 1, DG2 = 2, DG1 = 3;      // slots of a listed pixel
  __shared__ float sw[2][NP];                          // [layer][i * T + j], then the bias
  __shared__ float swT[2][T * T];                      // [layer][j * T + i]: row i of a mat-vec read by consecutive threads i
  __shared__ float lst[ATTN_LIST][4][T];
  __shared__ int act[256];                             // pixel of the chunk's r-th active thread
  __shared__ int wave_cnt[4];
  for (int i = threadIdx.x; i < T * T; i += blockDim.x) {
    const float a = w0[i], c = w1[i];
    sw[0][i] = a; sw[1][i] = c;
    const int r = i / T, q = i - r * T;
    swT[0][q * T + r] = a; swT[1][q * T + r] = c;
  }
  for (int i = threadIdx.x; i < T; i += blockDim.x) { sw[0][T * T + i] = b0[i]; sw[1][T * T + i] = b1[i]; }
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // element e of this thread: layer l = e / NP, index q = e % NP; q < T*T: weight (i, j) = (q / T, q % T) <- dg[i] * h[j]; else bias
  float acc[NE];
#pragma unroll
  for (int k = 0; k < NE; ++k) acc[k] = 0.f;
  const int nchunk = (HW + 255) / 256;
  for (int chunk = blockIdx.x; chunk < nchunk; chunk += gridDim.x) {
    const int p = chunk * 256 + threadIdx.x;
    bool any = false;
    if (p < HW) {
#pragma unroll
      for (int t = 0; t < T; ++t) any |= (dout[((size_t)b * T + t) * HW + p] != 0.f);
    }
    // rank of this thread among the chunk's active pixels (wave ballots + wave offsets)
    const unsigned long long bal = __ballot(any);
    __syncthreads();                                   // sw / swT visible (first trip); act / wave_cnt of the previous chunk consumed
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    int rank = __popcll(bal & ((1ull << lane) - 1ull)), nact = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { if (w < wave) rank += wave_cnt[w]; nact += wave_cnt[w]; }
    if (nact == 0) continue;                           // block-uniform
    if (any) act[rank] = p;
    for (int r0 = 0; r0 < nact; r0 += ATTN_LIST) {
      const int n = min(ATTN_LIST, nact - r0);
      __syncthreads();                                 // act visible / previous round consumed
      int pm[PPT], pi[PPT];
      bool ok[PPT];
      float go[PPT], h0v[PPT], a1[PPT], dh1[PPT];
#pragma unroll
      for (int k = 0; k < PPT; ++k) {
        const int pair = threadIdx.x + 256 * k;
        pm[k] = pair / T; pi[k] = pair - pm[k] * T;
        ok[k] = pm[k] < n;                             // (pair < ATTN_LIST * T follows: n <= ATTN_LIST)
        if (ok[k]) {
          const size_t o = ((size_t)b * T + pi[k]) * HW + act[r0 + pm[k]];
          go[k] = dout[o];
          h0v[k] = x[o];
          lst[pm[k]][H0][pi[k]] = h0v[k];
        }
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < PPT; ++k) {
        if (!ok[k]) continue;
        float g = sw[0][T * T + pi[k]];
#pragma unroll 8
        for (int j = 0; j < T; ++j) g += swT[0][j * T + pi[k]] * lst[pm[k]][H0][j];
        a1[k] = h0v[k] + h0v[k] * g;
        lst[pm[k]][H1][pi[k]] = a1[k] > 0.f ? a1[k] : 0.f;
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < PPT; ++k) {
        if (!ok[k]) continue;
        float g = sw[1][T * T + pi[k]];
#pragma unroll 8
        for (int j = 0; j < T; ++j) g += swT[1][j * T + pi[k]] * lst[pm[k]][H1][j];
        const float h1i = lst[pm[k]][H1][pi[k]];
        const float a2 = h1i + h1i * g;
        const float da2 = a2 > 0.f ? go[k] : 0.f;
        lst[pm[k]][DG2][pi[k]] = da2 * h1i;
        dh1[k] = da2 * (1.f + g);
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < PPT; ++k) {
        if (!ok[k]) continue;
        float sm = 0.f;
#pragma unroll 8
        for (int i = 0; i < T; ++i) sm += sw[1][i * T + pi[k]] * lst[pm[k]][DG2][i];
        const float d = dh1[k] + sm;
        const float da1 = a1[k] > 0.f ? d : 0.f;
        lst[pm[k]][DG1][pi[k]] = da1 * h0v[k];
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < NE; ++k) {
        const int e = threadIdx.x + 256 * k;
        if (e >= 2 * NP) continue;
        const int l = e >= NP ? 1 : 0, q = e - l * NP;
        const int sd = l ? DG2 : DG1, sh = l ? H1 : H0;
        float a = acc[k];
        if (q < T * T) {
          const int i = q / T, j = q % T;
          for (int m = 0; m < n; ++m) a += lst[m][sd][i] * lst[m][sh][j];
        } else {
          for (int m = 0; m < n; ++m) a += lst[m][sd][q - T * T];
        }
        acc[k] = a;
      }
    }
  }
  if (ws.part) {                                       // ws.part = [workgroups][2 NP]: added in workgroup order by det_reduce
    const unsigned blk = blockIdx.y * gridDim.x + blockIdx.x;
#pragma unroll
    for (int k = 0; k < NE; ++k) {
      const int e = threadIdx.x + 256 * k;
      if (e < 2 * NP) ws.part[(size_t)blk * 2 * NP + e] = acc[k];
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < NE; ++k) {
    const int e = threadIdx.x + 256 * k;
    if (e >= 2 * NP || acc[k] == 0.f) continue;
    const int l = e >= NP ? 1 : 0, q = e - l * NP;
    float* dw = l ? dw1 : dw0;
    float* db = l ? db1 : db0;
    atomicAdd(q < T * T ? dw + q : db + (q - T * T), acc[k]);
  }
}

// ------------------------------------------------------------------ pool + duplicate
// y[b, co] = maxpool2x2(x[b, co/2]) : view(b*t, c/t).repeat_interleave(2, dim=1) == channel co <- co/2
__global__ void pooldup_fwd_kernel(const float* __restrict__ x, float* y, int BC, int H, int W) {
  const int H2 = H >> 1, W2 = W >> 1;
  const size_t n = (size_t)BC * H2 * W2;
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
    const int w2 = idx % W2, h2 = (idx / W2) % H2;
    const size_t bc = idx / ((size_t)W2 * H2);
    const float* px = x + (bc * H + 2 * h2) * W + 2 * w2;
    const float2 r0 = *reinterpret_cast<const float2*>(px);
    const float2 r1 = *reinterpret_cast<const float2*>(px + W);
    const float m = fmaxf(fmaxf(r0.x, r0.y), fmaxf(r1.x, r1.y));
    const size_t yo = ((bc * 2) * H2 + h2) * W2 + w2;
    y[yo] = m;
    y[yo + (size_t)H2 * W2] = m;
  }
}
// gradient goes to the FIRST maximum in window scan order (max_pool2d backward)
__global__ void pooldup_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* dx, int BC, int H, int W) {
  const int H2 = H >> 1, W2 = W >> 1;
  const size_t n = (size_t)BC * H2 * W2;
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
    const int w2 = idx % W2, h2 = (idx / W2) % H2;
    const size_t bc = idx / ((size_t)W2 * H2);
    const size_t xo = (bc * H + 2 * h2) * W + 2 * w2;
    const float2 r0 = *reinterpret_cast<const float2*>(x + xo);
    const float2 r1 = *reinterpret_cast<const float2*>(x + xo + W);
    int am = 0; float mv = r0.x;
    if (r0.y > mv) { mv = r0.y; am = 1; }
    if (r1.x > mv) { mv = r1.x; am = 2; }
    if (r1.y > mv) { mv = r1.y; am = 3; }
    const size_t yo = ((bc * 2) * H2 + h2) * W2 + w2;
    const float g = dy[yo] + dy[yo + (size_t)H2 * W2];
    *reinterpret_cast<float2*>(dx + xo) = make_float2(am == 0 ? g : 0.f, am == 1 ? g : 0.f);
    *reinterpret_cast<float2*>(dx + xo + W) = make_float2(am == 2 ? g : 0.f, am == 3 ? g : 0.f);
  }
}

// ------------------------------------------------------------------ UPPos front half
__device__ __forceinline__ void ac_src(int o, float scale, int S, int& i0, int& i1, float& l1) {
  const float src = scale * (float)o;           // align_corners=True: src = dst * (in-1)/(out-1)
  i0 = (int)src;
  i1 = i0 + (i0 < S - 1 ? 1 : 0);
  l1 = src - (float)i0;
}
// x (BC, Sh, Sw) -> u (BC, 2Sh, 2Sw); pos (2Sh, 2Sw).  Thread = one output pixel for a chunk of channels (blockIdx.y): the
// bilinear taps and the modulation 1 + (2 sigmoid(pos) - 1) depend on the pixel only and are computed once, not per channel.
// bias != null (round 3, UPPos with the 1x1 projection moved IN FRONT of the upsampling -- models/p2igan.py uppos): u = act(v (1 + pm)
// + bias[channel]), C = channels per sample.
__global__ __launch_bounds__(256) void upmod_fwd_kernel(const float* __restrict__ x, const float* __restrict__ pos, float* u, int BC, int Sh, int Sw,
                                                       int chunk, const float* __restrict__ bias, int act, int C) {
  const int Oh = 2 * Sh, Ow = 2 * Sw;
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= Oh * Ow) return;
  const float sch = (float)(Sh - 1) / (float)(Oh - 1), scw = (float)(Sw - 1) / (float)(Ow - 1);
  const int ox = pix % Ow, oy = pix / Ow;
  int y0, y1, x0, x1; float ly, lx;
  ac_src(oy, sch, Sh, y0, y1, ly);
  ac_src(ox, scw, Sw, x0, x1, lx);
  const float pm = 2.f / (1.f + expf(-pos[pix])) - 1.f;
  const int i00 = y0 * Sw + x0, i01 = y0 * Sw + x1, i10 = y1 * Sw + x0, i11 = y1 * Sw + x1;
  const int bc0 = blockIdx.y * chunk, bc1 = min(BC, bc0 + chunk);
  const size_t SS = (size_t)Sh * Sw, OO = (size_t)Oh * Ow;
#pragma unroll 4
  for (int bc = bc0; bc < bc1; ++bc) {
    const float* px = x + bc * SS;
    const float v = (1.f - ly) * ((1.f - lx) * px[i00] + lx * px[i01]) + ly * ((1.f - lx) * px[i10] + lx * px[i11]);
    float r = v + v * pm;
    if (bias) r = act_apply(r + bias[bc % C], act);
    u[bc * OO + pix] = r;
  }
}
// dpos[y,x] += sum_{bc in chunk} du * v * 2 s (1-s);   grid.y = bc chunks
__global__ void upmod_bwd_pos_kernel(const float* __restrict__ x, const float* __restrict__ pos, const float* __restrict__ du,
                                     float* dpos, int BC, int Sh, int Sw, int chunk, DetWs ws) {
  const int Oh = 2 * Sh, Ow = 2 * Sw;
  const float sch = (float)(Sh - 1) / (float)(Oh - 1), scw = (float)(Sw - 1) / (float)(Ow - 1);
  const int pix0 = blockIdx.x * blockDim.x + threadIdx.x;
  const bool pvalid = pix0 < Oh * Ow;
  const int pix = pix0;
  if (!pvalid) return;
  const int ox = pix % Ow, oy = pix / Ow;
  int y0, y1, x0, x1; float ly, lx;
  ac_src(oy, sch, Sh, y0, y1, ly);
  ac_src(ox, scw, Sw, x0, x1, lx);
  const int bc0 = blockIdx.y * chunk, bc1 = min(BC, bc0 + chunk);
  float acc = 0.f;
  for (int bc = bc0; bc < bc1; ++bc) {
    const float* px = x + (size_t)bc * Sh * Sw;
    const float v = (1.f - ly) * ((1.f - lx) * px[y0 * Sw + x0] + lx * px[y0 * Sw + x1]) +
                    ly * ((1.f - lx) * px[y1 * Sw + x0] + lx * px[y1 * Sw + x1]);
    acc += du[(size_t)bc * Oh * Ow + pix] * v;
  }
  const float sg = 1.f / (1.f + expf(-pos[pix]));
  const float contrib = acc * 2.f * sg * (1.f - sg);
  if (ws.part) {                                       // ws.part = [gridDim.y chunks][pixels]: added in chunk order by det_reduce
    if (pvalid) ws.part[(size_t)blockIdx.y * (Oh * Ow) + pix] = contrib;
    return;
  }
  atomicAdd(dpos + pix, contrib);
}
// dx[bc, yi, xi] = sum over the output pixels that sample (yi, xi) of weight * du * (1 + posm).  Thread = one INPUT pixel for a
// chunk of channels: the <= 6 x 6 (weight * modulation) factors are computed once per thread (they held an expf and two
// source-index computations per output sample and channel before), the channel loop is loads and FMAs only.
__global__ __launch_bounds__(256) void upmod_bwd_x_kernel(const float* __restrict__ pos, const float* __restrict__ du, float* dx, int BC, int Sh, int Sw,
                                                         int chunk) {
  const int Oh = 2 * Sh, Ow = 2 * Sw;
  const int pin = blockIdx.x * blockDim.x + threadIdx.x;
  if (pin >= Sh * Sw) return;
  const float sch = (float)(Sh - 1) / (float)(Oh - 1), scw = (float)(Sw - 1) / (float)(Ow - 1);
  const int xi = pin % Sw, yi = pin / Sw;
  float wy[6], wx[6];
  int oyv[6], oxv[6];
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    const int oy = 2 * yi - 2 + a, ox = 2 * xi - 2 + a;
    int i0, i1; float l;
    wy[a] = 0.f; wx[a] = 0.f;
    oyv[a] = min(max(oy, 0), Oh - 1); oxv[a] = min(max(ox, 0), Ow - 1);
    if (oy >= 0 && oy < Oh) { ac_src(oy, sch, Sh, i0, i1, l); wy[a] = (i0 == yi ? 1.f - l : 0.f) + (i1 == yi ? l : 0.f); }
    if (ox >= 0 && ox < Ow) { ac_src(ox, scw, Sw, i0, i1, l); wx[a] = (i0 == xi ? 1.f - l : 0.f) + (i1 == xi ? l : 0.f); }
  }
  float wt[6][6];
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      const float w = wy[a] * wx[b];
      wt[a][b] = w != 0.f ? w * (2.f / (1.f + expf(-pos[oyv[a] * Ow + oxv[b]]))) : 0.f;     // 1 + (2 sigmoid - 1)
    }
  const int bc0 = blockIdx.y * chunk, bc1 = min(BC, bc0 + chunk);
  const size_t SS = (size_t)Sh * Sw, OO = (size_t)Oh * Ow;
  for (int bc = bc0; bc < bc1; ++bc) {
    const float* g = du + bc * OO;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      if (wy[a] == 0.f) continue;
#pragma unroll
      for (int b = 0; b < 6; ++b) acc += wt[a][b] * g[oyv[a] * Ow + oxv[b]];
    }
    dx[bc * SS + pin] = acc;
  }
}

// ------------------------------------------------------------------ discriminator tail
__device__ __forceinline__ void hp_src(int o, float scale, int S, int& i0, int& i1, float& l1) {
  float src = ((float)o + 0.5f) * scale - 0.5f;     // align_corners=False
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  i1 = i0 + (i0 < S - 1 ? 1 : 0);
  l1 = src - (float)i0;
}
__global__ void dtail_fwd_kernel(const float* __restrict__ o2, const float* __restrict__ o3, const float* __restrict__ alpha,
                                 float* fused, int B, int H2, int W2, int T3, int H3, int W3) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * H2 * W2) return;
  const int x = idx % W2, y = (idx / W2) % H2, b = idx / (W2 * H2);
  const float sh = (float)H3 / (float)H2, sw = (float)W3 / (float)W2;
  int y0, y1, x0, x1; float ly, lx;
  hp_src(y, sh, H3, y0, y1, ly);
  hp_src(x, sw, W3, x0, x1, lx);
  float m00 = 0.f, m01 = 0.f, m10 = 0.f, m11 = 0.f;
  for (int t = 0; t < T3; ++t) {
    const float* p = o3 + ((size_t)b * T3 + t) * H3 * W3;
    m00 += p[y0 * W3 + x0]; m01 += p[y0 * W3 + x1]; m10 += p[y1 * W3 + x0]; m11 += p[y1 * W3 + x1];
  }
  const float it = 1.f / (float)T3;
  m00 *= it; m01 *= it; m10 *= it; m11 *= it;
  float up;
  if (H3 == H2 && W3 == W2) up = m00;
  else up = (1.f - ly) * ((1.f - lx) * m00 + lx * m01) + ly * ((1.f - lx) * m10 + lx * m11);
  const float sg = 1.f / (1.f + expf(-*alpha));
  fused[idx] = sg * o2[idx] + up;
}
__global__ void dtail_bwd2_kernel(const float* __restrict__ o2, const float* __restrict__ alpha, const float* __restrict__ df,
                                  float* do2, float* dalpha, int n, DetWs ws) {
  __shared__ float red[16];
  const float sg = 1.f / (1.f + expf(-*alpha));
  float acc = 0.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float g = df[i];
    if (do2) do2[i] = sg * g;
    acc += g * o2[i];
  }
  acc = block_sum(acc, red);
  if (dalpha && ws.part) {                             // ws.part = [workgroups]: added in workgroup order by det_reduce
    if (threadIdx.x == 0) ws.part[blockIdx.x] = acc * sg * (1.f - sg);
    return;
  }
  if (dalpha && threadIdx.x == 0) atomicAdd(dalpha, acc * sg * (1.f - sg));
}
__global__ void dtail_bwd3_kernel(const float* __restrict__ df, float* do3, int B, int H2, int W2, int T3, int H3, int W3) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * H3 * W3) return;
  const int xi = idx % W3, yi = (idx / W3) % H3, b = idx / (W3 * H3);
  const float sh = (float)H3 / (float)H2, sw = (float)W3 / (float)W2;
  const int ry = H2 / H3 > 0 ? H2 / H3 : 1, rx = W2 / W3 > 0 ? W2 / W3 : 1;
  const int ylo = max(0, (yi - 1) * ry - 1), yhi = min(H2 - 1, (yi + 2) * ry);
  const int xlo = max(0, (xi - 1) * rx - 1), xhi = min(W2 - 1, (xi + 2) * rx);
  float acc = 0.f;
  const bool same = (H3 == H2 && W3 == W2);
  for (int y = ylo; y <= yhi; ++y) {
    int y0, y1; float ly;
    hp_src(y, sh, H3, y0, y1, ly);
    const float wy = same ? (y == yi ? 1.f : 0.f) : ((y0 == yi ? 1.f - ly : 0.f) + (y1 == yi ? ly : 0.f));
    if (wy == 0.f) continue;
    for (int x = xlo; x <= xhi; ++x) {
      int x0, x1; float lx;
      hp_src(x, sw, W3, x0, x1, lx);
      const float wx = same ? (x == xi ? 1.f : 0.f) : ((x0 == xi ? 1.f - lx : 0.f) + (x1 == xi ? lx : 0.f));
      if (wx != 0.f) acc += wy * wx * df[((size_t)b * H2 + y) * W2 + x];
    }
  }
  acc /= (float)T3;
  for (int t = 0; t < T3; ++t) do3[(((size_t)b * T3 + t) * H3 + yi) * W3 + xi] = acc;
}

// ------------------------------------------------------------------ misc
// db[c] += sum_{b, inner} dy * act'(y); grid (C, chunks): block partial sums combined by one atomic each
__global__ void bias_grad_kernel(const float* __restrict__ dy, const float* __restrict__ y, int act, float* db, int B, int C,
                                 int64_t inner, DetWs ws) {
  __shared__ float red[16];
  const int c = blockIdx.x;
  const int64_t total = (int64_t)B * inner;
  const int64_t per = (total + gridDim.y - 1) / gridDim.y;
  const int64_t lo = per * blockIdx.y, hi = lo + per < total ? lo + per : total;
  float acc = 0.f;
  for (int64_t j = lo + threadIdx.x; j < hi; j += blockDim.x) {
    const int64_t b = j / inner, i = j - b * inner;
    const size_t idx = ((size_t)b * C + c) * inner + i;
    float g = dy[idx];
    if (y) g = act_grad(g, y[idx], act);
    acc += g;
  }
  acc = block_sum(acc, red);
  if (ws.part) {                                       // ws.part = [C][gridDim.y chunks]: added in chunk order by det_reduce
    if (threadIdx.x == 0) ws.part[(size_t)c * gridDim.y + blockIdx.y] = acc;
    return;
  }
  if (threadIdx.x == 0) atomicAdd(db + c, acc);
}
// out = dy * act'(y) AND db[c] += sum of it, one pass (UPPos backward: the masked gradient feeds the upsampling's adjoint, its channel
// sums are the bias gradient).  Block = (channel c, sample b, chunk of the plane); float4 streams, inner % 4 == 0.
__global__ __launch_bounds__(256) void act_bwd_bias_kernel(const float* __restrict__ dy, const float* __restrict__ y, int act, float* __restrict__ out,
                                                          float* db, int C, int64_t inner4, DetWs ws) {
  __shared__ float red[16];
  const int c = blockIdx.x, b = blockIdx.y;
  const int64_t base = ((int64_t)b * C + c) * inner4;
  const int64_t per = (inner4 + gridDim.z - 1) / gridDim.z;
  const int64_t lo = per * blockIdx.z, hi = lo + per < inner4 ? lo + per : inner4;
  float acc = 0.f;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
    const float4 g = reinterpret_cast<const float4*>(dy)[base + i], v = reinterpret_cast<const float4*>(y)[base + i];
    const float4 r = make_float4(act_grad(g.x, v.x, act), act_grad(g.y, v.y, act), act_grad(g.z, v.z, act), act_grad(g.w, v.w, act));
    reinterpret_cast<float4*>(out)[base + i] = r;
    acc += (r.x + r.y) + (r.z + r.w);
  }
  acc = block_sum(acc, red);
  if (ws.part) {                                       // ws.part = [C][B][chunks]: added in (sample, chunk) order by det_reduce
    if (threadIdx.x == 0) ws.part[((size_t)c * gridDim.y + blockIdx.y) * gridDim.z + blockIdx.z] = acc;
    return;
  }
  if (threadIdx.x == 0) atomicAdd(db + c, acc);
}
// out = dy * act'(y) (y = saved post-activation tensor); float4 streams
__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, int act, float* out, int64_t n4, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 g = reinterpret_cast<const float4*>(dy)[i], v = reinterpret_cast<const float4*>(y)[i];
    reinterpret_cast<float4*>(out)[i] = make_float4(act_grad(g.x, v.x, act), act_grad(g.y, v.y, act), act_grad(g.z, v.z, act), act_grad(g.w, v.w, act));
  }
  if (blockIdx.x == 0 && threadIdx.x < n - n4 * 4) {
    const int64_t i = n4 * 4 + threadIdx.x;
    out[i] = act_grad(dy[i], y[i], act);
  }
}
__global__ void add2_kernel(float* out, const float* __restrict__ a, const float* __restrict__ b, int64_t n4, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
    reinterpret_cast<float4*>(out)[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
  }
  if (blockIdx.x == 0) for (int64_t i = 4 * n4 + threadIdx.x; i < n; i += blockDim.x) out[i] = a[i] + b[i];
}
__global__ void axpy_kernel(float* y, const float* __restrict__ x, float a, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] += a * x[i];
}
// torch.optim.Adam single-tensor math: m.lerp_(g, 1-b1); v = b2 v + (1-b2) g g;
// p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ void adam_kernel(float* p, const float* __restrict__ g, float* m, float* v, int64_t n, float step_size, float beta1,
                            float beta2, float eps, float bc2_sqrt) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i];
    const float mi = m[i] + (gi - m[i]) * (1.f - beta1);
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
  }
}

// Adam with the step counter on the device (hipGraph replay: a captured launch cannot take a new host scalar per step)
__global__ void adam_coef_kernel(int32_t* step_dev, float* coef, float lr, float beta1, float beta2) {
  const int step = step_dev[0] + 1;
  step_dev[0] = step;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  coef[0] = (float)((double)lr / bc1);
  coef[1] = (float)sqrt(bc2);
}
__global__ void adam_dev_kernel(float* p, const float* __restrict__ g, float* m, float* v, int64_t n, const float* __restrict__ coef,
                                float beta1, float beta2, float eps) {
  const float step_size = coef[0], bc2_sqrt = coef[1];
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i];
    const float mi = m[i] + (gi - m[i]) * (1.f - beta1);
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
  }
}

static inline int grid_for(int64_t n, int block = 256, int cap = 8192) {
  int64_t g = (n + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// second stage of the deterministic sums (common.h): thread = one output element, its partials added in ascending order
__global__ __launch_bounds__(256) void det_reduce_kernel(const float* __restrict__ part, int groups, int n, long long gs, long long ks, const DetSegs segs) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= groups) return;
  const float* p = part + (long long)g * gs;
  float sacc = 0.f;
  int k = 0;
  for (; k + 8 <= n; k += 8) {                         // eight loads in flight, added in order
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(long long)(k + u) * ks];
#pragma unroll
    for (int u = 0; u < 8; ++u) sacc += v[u];
  }
  for (; k < n; ++k) sacc += p[(long long)k * ks];
  int r = g;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (r < segs.len[i]) { segs.out[i][r] += sacc; return; }
    r -= segs.len[i];
  }
}
int det_reduce(const float* part, int groups, int n, long long gs, long long ks, const DetSegs& segs, hipStream_t s) {
  P2I_LAUNCH(det_reduce_kernel, dim3(ceil_div(groups, 256)), dim3(256), 0, s, part, groups, n, gs, ks, segs);
  return launch_status();
}
}  // namespace p2i
using namespace p2i;

extern "C" int p2i_attn_fwd(const float* x, const float* w0, const float* b0, const float* w1, const float* b1, float* out,
                            int B, int T, int HW, void* stream) {
  P2I_REQUIRE(x && w0 && b0 && w1 && b1 && out, "null pointer");
  // T = 16 is the reference (layer.py:310); 8 and 32 are this build's generalisation (parity unpinned)
  P2I_REQUIRE(T == 8 || T == 16 || T == 32, "AttentionBlock kernels exist for T in {8, 16, 32}");
  const dim3 grid(ceil_div(HW, 256), B);
  hipStream_t s = (hipStream_t)stream;
  if (T == 16) P2I_LAUNCH(attn_fwd_kernel<16>, grid, dim3(256), 0, s, x, w0, b0, w1, b1, out, B, HW);
  else if (T == 32) P2I_LAUNCH(attn_fwd_kernel<32>, grid, dim3(256), 0, s, x, w0, b0, w1, b1, out, B, HW);
  else P2I_LAUNCH(attn_fwd_kernel<8>, grid, dim3(256), 0, s, x, w0, b0, w1, b1, out, B, HW);
  return launch_status();
}
extern "C" int p2i_attn_bwd(const float* x, const float* w0, const float* b0, const float* w1, const float* b1,
                            const float* dout, float* dw0, float* db0, float* dw1, float* db1, int B, int T, int HW, void* stream) {
  P2I_REQUIRE(x && w0 && b0 && w1 && b1 && dout && dw0 && db0 && dw1 && db1, "null pointer");
  P2I_REQUIRE(T == 8 || T == 16 || T == 32, "AttentionBlock kernels exist for T in {8, 16, 32}");
  // a workgroup walks several 256-pixel chunks of its sample: 64 workgroups in all (4 per CU quarter is plenty: the kernel reads
  // dout once -- 8 MB at B = 8 -- and does real work for the few pixels with a gradient)
  int gx = ceil_div(HW, 256);
  const int want = 64 / (B < 64 ? B : 64) > 0 ? 64 / (B < 64 ? B : 64) : 1;
  if (gx > want) gx = want;
  const dim3 grid(gx, B);
  hipStream_t s = (hipStream_t)stream;
  const size_t nblk = (size_t)grid.x * grid.y;
  const DetWs ws = det_take(nblk * 2 * (size_t)(T * T + T), 1);
  if (T == 16) P2I_LAUNCH(attn_bwd_kernel<16>, grid, dim3(256), 0, s, x, w0, b0, w1, b1, dout, dw0, db0, dw1, db1, B, HW, ws);
  else if (T == 32) P2I_LAUNCH(attn_bwd_kernel<32>, grid, dim3(256), 0, s, x, w0, b0, w1, b1, dout, dw0, db0, dw1, db1, B, HW, ws);
  else P2I_LAUNCH(attn_bwd_kernel<8>, grid, dim3(256), 0, s, x, w0, b0, w1, b1, dout, dw0, db0, dw1, db1, B, HW, ws);
  if (ws.part) {
    const int NP = T * T + T;
    return det_reduce(ws.part, 2 * NP, (int)nblk, 1, 2 * NP, DetSegs{{dw0, db0, dw1, db1}, {T * T, T, T * T, T}}, s);
  }
  return launch_status();
}
extern "C" int p2i_pooldup_fwd(const float* x, float* y, int B, int C, int H, int W, void* stream) {
  P2I_REQUIRE(x && y, "null pointer");
  P2I_REQUIRE(H % 2 == 0 && W % 2 == 0, "pooldup needs even H, W");
  const int64_t n = (int64_t)B * C * (H / 2) * (W / 2);
  P2I_LAUNCH(pooldup_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, B * C, H, W);
  return launch_status();
}
extern "C" int p2i_pooldup_bwd(const float* x, const float* dy, float* dx, int B, int C, int H, int W, void* stream) {
  P2I_REQUIRE(x && dy && dx, "null pointer");
  P2I_REQUIRE(H % 2 == 0 && W % 2 == 0, "pooldup needs even H, W");
  const int64_t n = (int64_t)B * C * (H / 2) * (W / 2);
  P2I_LAUNCH(pooldup_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, B * C, H, W);
  return launch_status();
}
extern "C" int p2i_upmod_fwd(const float* x, const float* pos, float* u, int B, int C, int S, int S2w, void* stream) {
  P2I_REQUIRE(x && pos && u, "null pointer");
  {
    const int BC = B * C, npix = 4 * S * S2w;
    int chunk = BC;                                  // enough (pixel tile, channel chunk) blocks to fill the chip ~8 times
    while (chunk > 8 && (long long)ceil_div(npix, 256) * ceil_div(BC, chunk) < 2048) chunk = (chunk + 1) / 2;
    P2I_LAUNCH(upmod_fwd_kernel, dim3(ceil_div(npix, 256), ceil_div(BC, chunk)), dim3(256), 0, (hipStream_t)stream, x, pos, u, BC, S, S2w, chunk,
                       (const float*)nullptr, P2I_ACT_NONE, C);
  }
  return launch_status();
}
extern "C" int p2i_upmod_fwd_ba(const float* x, const float* pos, const float* bias, int act, float* u, int B, int C, int S, int S2w, void* stream) {
  P2I_REQUIRE(x && pos && bias && u, "null pointer");
  const int BC = B * C, npix = 4 * S * S2w;
  int chunk = BC;
  while (chunk > 8 && (long long)ceil_div(npix, 256) * ceil_div(BC, chunk) < 2048) chunk = (chunk + 1) / 2;
  P2I_LAUNCH(upmod_fwd_kernel, dim3(ceil_div(npix, 256), ceil_div(BC, chunk)), dim3(256), 0, (hipStream_t)stream, x, pos, u, BC, S, S2w, chunk,
                     bias, act, C);
  return launch_status();
}
extern "C" int p2i_upmod_bwd(const float* x, const float* pos, const float* du, float* dx, float* dpos, int B, int C, int S,
                             int S2w, void* stream) {
  P2I_REQUIRE(x && pos && du, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  const int BC = B * C;
  if (dpos) {
    const int chunk = 16;
    const dim3 gp(ceil_div(4 * S * S2w, 256), ceil_div(BC, chunk));
    const int npix = 4 * S * S2w;
    const DetWs ws = det_take((size_t)npix * gp.y, 0);
    P2I_LAUNCH(upmod_bwd_pos_kernel, gp, dim3(256), 0, s, x, pos, du, dpos, BC, S, S2w, chunk, ws);
    if (ws.part)
      if (int e = det_reduce(ws.part, npix, (int)gp.y, 1, npix, DetSegs{{dpos, nullptr, nullptr, nullptr}, {npix, 0, 0, 0}}, s)) return e;
  }
  if (dx) {
    {
      const int npin = S * S2w;
      int chunk2 = BC;
      while (chunk2 > 8 && (long long)ceil_div(npin, 256) * ceil_div(BC, chunk2) < 2048) chunk2 = (chunk2 + 1) / 2;
      P2I_LAUNCH(upmod_bwd_x_kernel, dim3(ceil_div(npin, 256), ceil_div(BC, chunk2)), dim3(256), 0, s, pos, du, dx, BC, S, S2w, chunk2);
    }
  }
  return launch_status();
}
extern "C" int p2i_dtail_fwd(const float* out2d, const float* out3d, const float* alpha2d, float* fused, int B, int H2, int W2,
                             int T3, int H3, int W3, void* stream) {
  P2I_REQUIRE(out2d && out3d && alpha2d && fused, "null pointer");
  P2I_LAUNCH(dtail_fwd_kernel, dim3(ceil_div(B * H2 * W2, 256)), dim3(256), 0, (hipStream_t)stream, out2d, out3d, alpha2d,
                     fused, B, H2, W2, T3, H3, W3);
  return launch_status();
}
extern "C" int p2i_dtail_bwd(const float* out2d, const float* alpha2d, const float* dfused, float* dout2d, float* dout3d,
                             float* dalpha2d, int B, int H2, int W2, int T3, int H3, int W3, void* stream) {
  P2I_REQUIRE(out2d && alpha2d && dfused, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  const int n = B * H2 * W2;
  const int nb2 = min(ceil_div(n, 256), 64);
  const DetWs ws = dalpha2d ? det_take((size_t)nb2, 1) : DetWs{nullptr, nullptr};
  P2I_LAUNCH(dtail_bwd2_kernel, dim3(nb2), dim3(256), 0, s, out2d, alpha2d, dfused, dout2d, dalpha2d, n, ws);
  if (ws.part)
    if (int e = det_reduce(ws.part, 1, nb2, 0, 1, DetSegs{{dalpha2d, nullptr, nullptr, nullptr}, {1, 0, 0, 0}}, s)) return e;
  if (dout3d)
    P2I_LAUNCH(dtail_bwd3_kernel, dim3(ceil_div(B * H3 * W3, 256)), dim3(256), 0, s, dfused, dout3d, B, H2, W2, T3, H3, W3);
  return launch_status();
}
extern "C" int p2i_bias_grad(const float* dy, const float* y_act, int act, float* db, int B, int C, int64_t inner, void* stream) {
  P2I_REQUIRE(dy && db, "null pointer");
  const int64_t total = (int64_t)B * inner;
  int chunks = (int)((total + 16383) / 16384);
  if (chunks > 64) chunks = 64;
  if (chunks < 1) chunks = 1;
  const DetWs ws = det_take((size_t)C * chunks, 0);
  P2I_LAUNCH(bias_grad_kernel, dim3(C, chunks), dim3(256), 0, (hipStream_t)stream, dy, y_act, act, db, B, C, inner, ws);
  if (ws.part) return det_reduce(ws.part, C, chunks, chunks, 1, DetSegs{{db, nullptr, nullptr, nullptr}, {C, 0, 0, 0}}, (hipStream_t)stream);
  return launch_status();
}
extern "C" int p2i_act_bwd(const float* dy, const float* y, int act, float* out, int64_t n, void* stream) {
  P2I_REQUIRE(dy && y && out && n > 0, "null pointer");
  P2I_LAUNCH(act_bwd_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, dy, y, act, out, n / 4, n);
  return launch_status();
}
extern "C" int p2i_act_bwd_bias(const float* dy, const float* y, int act, float* out, float* db, int B, int C, int64_t inner, void* stream) {
  P2I_REQUIRE(dy && y && out && db && B > 0 && C > 0 && inner > 0, "null pointer");
  P2I_REQUIRE((inner & 3) == 0 && (((uintptr_t)dy | (uintptr_t)y | (uintptr_t)out) & 15) == 0, "inner % 4 and 16-byte alignment");
  int chunks = (int)((inner / 4 + 2047) / 2048);
  if (chunks > 16) chunks = 16;
  const DetWs ws = det_take((size_t)C * B * chunks, 0);
  P2I_LAUNCH(act_bwd_bias_kernel, dim3(C, B, chunks), dim3(256), 0, (hipStream_t)stream, dy, y, act, out, db, C, inner / 4, ws);
  if (ws.part) return det_reduce(ws.part, C, B * chunks, B * chunks, 1, DetSegs{{db, nullptr, nullptr, nullptr}, {C, 0, 0, 0}}, (hipStream_t)stream);
  return launch_status();
}
extern "C" int p2i_axpy(float* y, const float* x, float a, int64_t n, void* stream) {
  P2I_REQUIRE(y && x, "null pointer");
  P2I_LAUNCH(axpy_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, y, x, a, n);
  return launch_status();
}
extern "C" int p2i_add2(float* out, const float* a, const float* b, int64_t n, void* stream) {
  P2I_REQUIRE(out && a && b && n >= 0, "null pointer");
  const bool al = (((uintptr_t)out | (uintptr_t)a | (uintptr_t)b) & 15) == 0;
  const int64_t n4 = al ? n / 4 : 0;
  P2I_LAUNCH(add2_kernel, dim3(grid_for(n4 + 1)), dim3(256), 0, (hipStream_t)stream, out, a, b, n4, n);
  return launch_status();
}
extern "C" int p2i_zero(float* p, int64_t n, void* stream) {
  P2I_REQUIRE(p && n >= 0, "null pointer");
  if (n > 0 && p2i::memset_async(p, 0, sizeof(float) * (size_t)n, (hipStream_t)stream) != hipSuccess) { set_error("memset failed"); return P2I_EINVAL; }
  return P2I_OK;
}
extern "C" int p2i_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                        int step, void* stream) {
  P2I_REQUIRE(p && g && m && v && step >= 1, "bad adam arguments");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  P2I_LAUNCH(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, (float)(lr / bc1), beta1,
                     beta2, eps, (float)sqrt(bc2));
  return launch_status();
}

extern "C" int p2i_adam_dev(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                            int32_t* step_dev, float* coef2, void* stream) {
  P2I_REQUIRE(p && g && m && v && step_dev && coef2, "bad adam arguments");
  P2I_LAUNCH(adam_coef_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev, coef2, lr, beta1, beta2);
  P2I_LAUNCH(adam_dev_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, coef2, beta1, beta2, eps);
  return launch_status();
}


// ------------------------------------------------------------------------------------ batch assembly
// post_process of the reference loaders (sti_dataset.py:209,223-224: video/255, masked = video*mask) and the
// (B,T,H,W,C=1)->(B,T,C,H,W) permute of Trainer._prepare_batch (train.py:468-473), on the device: the loader ships
// uint8 frames and a uint8 mask (12x fewer PCIe bytes than three fp32 tensors).  One pass, 16 pixels per thread.
namespace p2i {
__global__ __launch_bounds__(256) void assemble_batch_kernel(const uint8_t* __restrict__ fr, const uint8_t* __restrict__ mk,
                                                             float* __restrict__ frames, float* __restrict__ masked,
                                                             float* __restrict__ masks, long long n, long long mask_period) {
  // mask index = i % mask_period  (mask_period = H*W: one (H,W) mask for every frame of every sample;
  //                                T*H*W: per-frame masks shared by the samples; B*T*H*W: a mask per voxel)
  const long long stride = (long long)gridDim.x * 256 * 4;
  for (long long i0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i0 < n; i0 += stride) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const long long i = i0 + q;
      if (i < n) {
        const float v = __fdiv_rn((float)fr[i], 255.0f);            // numpy: uint8.astype(float32) / 255.0
        const float m = mk[i % mask_period] ? 1.0f : 0.0f;
        frames[i] = v;
        masks[i] = m;
        masked[i] = v * m;
      }
    }
  }
}
}  // namespace p2i

// ---- sliding-window inference (infer.py:188-262): window w = frames w*step .. w*step + win - 1 of an (L, HW) event, frames past the
// end repeat the last one; the prediction of frame l is the mean over the windows that hold it (not counting the repeated copies),
// scaled and clipped at 0.
__global__ __launch_bounds__(256) void window_gather_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ wa,
                                                           float* __restrict__ wb, int L, int64_t HW4, int w0, int nw, int win, int step) {
  const int64_t per = (int64_t)win * HW4;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < (int64_t)nw * per; i += (int64_t)gridDim.x * 256) {
    const int w = (int)(i / per);
    const int64_t r = i - (int64_t)w * per;
    const int k = (int)(r / HW4);
    const int64_t p = r - (int64_t)k * HW4;
    const int l = min((w0 + w) * step + k, L - 1);
    reinterpret_cast<float4*>(wa)[i] = reinterpret_cast<const float4*>(a)[(int64_t)l * HW4 + p];
    if (b) reinterpret_cast<float4*>(wb)[i] = reinterpret_cast<const float4*>(b)[(int64_t)l * HW4 + p];
  }
}
__global__ __launch_bounds__(256) void window_mean_kernel(const float* __restrict__ pw, float* __restrict__ out, int L, int64_t HW4, int nwin, int win,
                                                         int step, float scale) {
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < (int64_t)L * HW4; i += (int64_t)gridDim.x * 256) {
    const int l = (int)(i / HW4);
    const int64_t p = i - (int64_t)l * HW4;
    // windows w with 0 <= l - w*step < win, in window order (the reference accumulates them in that order)
    const int wlo = l - win + 1 > 0 ? (l - win + 1 + step - 1) / step : 0, whi = min(l / step, nwin - 1);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int w = wlo; w <= whi; ++w) {
      const float4 v = reinterpret_cast<const float4*>(pw)[((int64_t)w * win + (l - w * step)) * HW4 + p];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const float c = fmaxf((float)(whi - wlo + 1), 1e-5f);
    acc.x = fmaxf(acc.x / c * scale, 0.f); acc.y = fmaxf(acc.y / c * scale, 0.f);
    acc.z = fmaxf(acc.z / c * scale, 0.f); acc.w = fmaxf(acc.w / c * scale, 0.f);
    reinterpret_cast<float4*>(out)[i] = acc;
  }
}
extern "C" int p2i_window_gather(const float* a, const float* b, float* wa, float* wb, int L, int64_t HW, int w0, int nw, int win, int step,
                                 void* stream) {
  P2I_REQUIRE(a && wa && (b == nullptr) == (wb == nullptr), "null pointer");
  P2I_REQUIRE(L > 0 && nw > 0 && win > 0 && step > 0 && HW > 0 && (HW & 3) == 0, "bad window geometry (H*W must be a multiple of 4)");
  P2I_REQUIRE((((uintptr_t)a | (uintptr_t)b | (uintptr_t)wa | (uintptr_t)wb) & 15) == 0, "16-byte alignment");
  P2I_LAUNCH(window_gather_kernel, dim3(grid_for((int64_t)nw * win * (HW / 4))), dim3(256), 0, (hipStream_t)stream, a, b, wa, wb, L, HW / 4,
                     w0, nw, win, step);
  return launch_status();
}
extern "C" int p2i_window_mean(const float* pred_windows, float* out, int L, int64_t HW, int nwin, int win, int step, float scale, void* stream) {
  P2I_REQUIRE(pred_windows && out, "null pointer");
  P2I_REQUIRE(L > 0 && nwin > 0 && win > 0 && step > 0 && HW > 0 && (HW & 3) == 0 && (nwin - 1) * step < L, "bad window geometry");
  P2I_REQUIRE((((uintptr_t)pred_windows | (uintptr_t)out) & 15) == 0, "16-byte alignment");
  P2I_LAUNCH(window_mean_kernel, dim3(grid_for((int64_t)L * (HW / 4))), dim3(256), 0, (hipStream_t)stream, pred_windows, out, L, HW / 4, nwin,
                     win, step, scale);
  return launch_status();
}
extern "C" int p2i_assemble_batch(const uint8_t* frames_u8, const uint8_t* mask_u8, int64_t mask_numel, float* frames,
                                  float* masked, float* masks, int B, int T, int H, int W, void* stream) {
  P2I_REQUIRE(frames_u8 && mask_u8 && frames && masked && masks, "null pointer");
  P2I_REQUIRE(B > 0 && T > 0 && H > 0 && W > 0, "bad dims");
  const long long hw = (long long)H * W, n = (long long)B * T * hw;
  P2I_REQUIRE(mask_numel == hw || mask_numel == (long long)T * hw || mask_numel == n, "mask must be (H,W), (T,H,W) or (B,T,H,W)");
  const long long blocks = (n + 1023) / 1024;
  P2I_LAUNCH(p2i::assemble_batch_kernel, dim3((unsigned)(blocks > 65535 ? 65535 : blocks)), dim3(256), 0, (hipStream_t)stream,
                     frames_u8, mask_u8, frames, masked, masks, n, (long long)mask_numel);
  return p2i::launch_status();
}
