// Exact-fp32 convolution on the bf16 matrix pipe of gfx950 ("x6" patch GEMM).
//
// CDNA4's f32-input MFMA runs at 1/16 of the bf16 rate (157 TF vs 2.5 PF) and has no xf32/tf32 form, while the
// P2I-GAN path needs fp32 results (1e-4 parity through 32 stacked convolutions, north_star).  This kernel keeps
// fp32 accuracy and moves the contraction to v_mfma_f32_32x32x16_bf16 by splitting every fp32 operand EXACTLY
// into three bf16 terms,  x = hi + mid + lo  (truncation split: each term takes the next 8 significant bits, so
// the three terms carry all 24 bits and the subtractions that produce the remainders are exact), and summing
// the six largest cross products in the fp32 accumulator:
//     a*b ~= hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid        (dropped: mid*lo, lo*mid, lo*lo <= 2^-23 |a*b|)
// Every bf16 x bf16 product is exact in fp32, so the result differs from an fp32 fma chain only by the dropped
// terms (~2 ulp of ONE product) and by the accumulation order.  6 bf16 MFMAs of K=16 replace 8 f32 MFMAs of K=2
// at 1/2 the cycles each: 2.67x the f32-MFMA roofline (2.5 PF / 6 = 417 TF fp32-equivalent).
//
// Same formulation and tiling vocabulary as conv.hip (dest[b,m,j] = epi(sum_{tap,k} W[tap][k][m] * src[b,k,j*S+d_tap])):
//   * weights arrive pre-split (wsplit_kernel) as Wb[plane][tap][k/8][m][8] bf16 -> one 16-B LDS-DMA element per
//     (m, 8 k's); a pipeline STAGE is TG taps of one 16-channel chunk, its weights are DMA'd R-1 stages ahead into
//     an R-slot LDS ring;
//   * the fp32 source patch (box*S + halo) of the next chunk is DMA'd into an LDS staging area while the current
//     chunk is multiplied, then split once per element into [plane][k/8][patch pixel][8 k's]: every MFMA operand
//     (A and B) is ONE ds_read_b128 and the split is paid once per staged element, not once per tap;
//   * one s_barrier per stage behind a counted s_waitcnt; fp32 accumulators; the epilogue (bias, activation,
//     residual, act'(mask)) is the one of patch_gemm_dma_kernel.
//
// STATUS (round 1): bit-for-bit within fp32 rounding of the f32-MFMA kernels on every covered layer (tests), but not
// yet faster at the bench batch size (profiles/README.md, "x6 engine"): the MFMA phase is issue-bound by per-stage
// LDS latency and uniform-branch overhead.  It is therefore opt-in (P2I_CONV_ENGINE=x6); the default engine is f32.
#include "conv_common.h"

namespace p2i {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack_hi16(float lo_elem, float hi_elem) {
  return (__float_as_uint(lo_elem) >> 16) | (__float_as_uint(hi_elem) & 0xFFFF0000u);
}
__device__ __forceinline__ float trunc_bf16(float v) { return __uint_as_float(__float_as_uint(v) & 0xFFFF0000u); }

// exact 3-way truncation split of 8 consecutive-k values into three packed bf16x8 operands
__device__ __forceinline__ void split8(const float (&v)[8], u32x4& hi, u32x4& mid, u32x4& lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = v[2 * j], b = v[2 * j + 1];
    hi[j] = pack_hi16(a, b);
    const float ra = a - trunc_bf16(a), rb = b - trunc_bf16(b);          // exact
    mid[j] = pack_hi16(ra, rb);
    const float sa = ra - trunc_bf16(ra), sb = rb - trunc_bf16(rb);      // exact, <= 8 significant bits left
    lo[j] = pack_hi16(sa, sb);
  }
}

// Wp fp32 [tap][Ck][CmPad]  ->  Wb bf16 [plane][tap][Ck/8][CmPad][8]
__global__ __launch_bounds__(256) void wsplit_kernel(const float* __restrict__ wp, u32x4* __restrict__ wb, int ntaps, int Ck, int CmPad) {
  const int KCt = Ck >> 3;
  const int total = ntaps * KCt * CmPad;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int m = i % CmPad;
    const int r = i / CmPad;
    const int kc = r % KCt, tap = r / KCt;
    const float* p = wp + ((size_t)tap * Ck + kc * 8) * CmPad + m;
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = p[(size_t)q * CmPad];
    u32x4 h, mi, lo;
    split8(v, h, mi, lo);
    wb[i] = h;
    wb[total + i] = mi;
    wb[2 * total + i] = lo;
  }
}

constexpr int X6_MAXE = 3;        // patch pixels per thread and 8-channel group (CSl <= 768)
constexpr int X6_MAXTG = 3;
constexpr int X6_MAXR = 12;

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the immediate must be a constant); n > 63 waits for <= 63
__device__ __forceinline__ void wait_vm(int n) {
#define P2I_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n < 0 ? 0 : n) {
    P2I_W(0) P2I_W(1) P2I_W(2) P2I_W(3) P2I_W(4) P2I_W(5) P2I_W(6) P2I_W(7) P2I_W(8) P2I_W(9)
    P2I_W(10) P2I_W(11) P2I_W(12) P2I_W(13) P2I_W(14) P2I_W(15) P2I_W(16) P2I_W(17) P2I_W(18) P2I_W(19)
    P2I_W(20) P2I_W(21) P2I_W(22) P2I_W(23) P2I_W(24) P2I_W(25) P2I_W(26) P2I_W(27) P2I_W(28) P2I_W(29)
    P2I_W(30) P2I_W(31) P2I_W(32) P2I_W(33) P2I_W(34) P2I_W(35) P2I_W(36) P2I_W(37) P2I_W(38) P2I_W(39)
    P2I_W(40) P2I_W(41) P2I_W(42) P2I_W(43) P2I_W(44) P2I_W(45) P2I_W(46) P2I_W(47) P2I_W(48) P2I_W(49)
    P2I_W(50) P2I_W(51) P2I_W(52) P2I_W(53) P2I_W(54) P2I_W(55) P2I_W(56) P2I_W(57) P2I_W(58) P2I_W(59)
    P2I_W(60) P2I_W(61) P2I_W(62)
    default: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
  }
#undef P2I_W
}

// Pipeline (all global traffic is LDS-DMA issued from inline asm, so every vmcnt wait below is ours and counted):
//   stage      = TG taps of one 16-channel chunk; its split weights sit in one slot of an R-slot LDS ring and are
//                DMA'd D = R-1 stages ahead (global/L2 -> LDS latency is ~2 us under load, a stage 0.3-1 us);
//   staging    = the fp32 source patch of the NEXT chunk, DMA'd (4 B per lane, border zeros from the descriptor
//                range check) while the current chunk is multiplied;
//   split pass = in the last stage of a chunk every thread reads 8 channels of its patch pixels from staging,
//                splits them into hi/mid/lo bf16 and writes the three operand planes of the other patch buffer;
//   one s_barrier per stage, preceded by a COUNTED s_waitcnt that only waits for the DMAs the next stage needs.
template <int TM, int TN, int WAVES_M>
__global__ __launch_bounds__(256) void patch_gemm_x6_kernel(const PatchGeom g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int WAVES_N = 4 / WAVES_M;
  constexpr int MB = 32 * TM * WAVES_M;
  constexpr int KC = 2;                              // 8-channel groups per chunk (CK = 16)
  constexpr int SUBS = KC * MB / 64;                 // DMA wave-instructions per (plane, tap)
  const bool multi = g.nclass > 1;
  const ClassGeom& cg = g.cls[multi ? blockIdx.z : 0];
  const int c_nT = multi ? cg.nT : g.nT, c_nH = multi ? cg.nH : g.nH, c_nW = multi ? cg.nW : g.nW;
  const int c_pT = multi ? cg.pT : g.pT, c_pH = multi ? cg.pH : g.pH, c_pW = multi ? cg.pW : g.pW;
  const int c_bT = multi ? cg.bT : g.bT, c_bH = multi ? cg.bH : g.bH, c_bW = multi ? cg.bW : g.bW;
  const int c_ntaps = multi ? cg.ntaps : g.ntaps;
  const short* c_tap_w = multi ? cg.tap_w : g.tap_w;
  const int* c_tap_off = multi ? cg.tap_off : g.tap_off;
  const int TG = g.TG, NTP = g.NTP, CSl = g.CSl, R = g.R;

  // ---- LDS carve
  int* wtab = reinterpret_cast<int*>(smem);                       // [3][NTP][KC]
  const int wtab_sz = (3 * NTP * KC + 3) & ~3;
  int* ptab = wtab + wtab_sz;                                     // [CSl] byte offset of each patch pixel (channel 0) or -4
  const int ptab_sz = (CSl + 3) & ~3;
  u32x4* ws0 = reinterpret_cast<u32x4*>(smem + wtab_sz + ptab_sz);   // [R][3][TG][KC][MB]
  const int WS16 = 3 * TG * KC * MB;
  float* stg = reinterpret_cast<float*>(ws0 + R * WS16);          // [16][CSl] fp32 staging (padded to whole wave-instructions)
  const int STG = (16 * CSl + 63) & ~63;
  u32x4* pb0 = reinterpret_cast<u32x4*>(stg + STG);               // [2][3][KC][CSl]
  const int PB16 = 3 * KC * CSl;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int l31 = lane & 31, lhi = lane >> 5;

  int tile = blockIdx.x;
  const int tw = tile % g.ntw; tile /= g.ntw;
  const int th = tile % g.nth; tile /= g.nth;
  const int tt = tile % g.ntt;
  const int tb = tile / g.ntt;
  const int j0b = tb << g.ljb, j0t = tt << g.ljt, j0h = th << g.ljh, j0w = tw << g.ljw;
  const int o0 = blockIdx.y * MB;
  const int JWm = (1 << g.ljw) - 1, JHm = (1 << g.ljh) - 1, JTm = (1 << g.ljt) - 1;
  const int sHW = g.sH * g.sW;
  const int src_t0 = j0t * g.mT + c_bT, src_h0 = j0h * g.mH + c_bH, src_w0 = j0w * g.mW + c_bW;
  const int KCt = g.Ck >> 3;

  // ---- weight offset table (uint4 units, without the chunk and m terms; -1 = no such tap)
  for (int r = tid; r < 3 * NTP * KC; r += 256) {
    const int kc = r % KC;
    const int t = (r / KC) % NTP;
    const int p = r / (KC * NTP);
    wtab[r] = (t < c_ntaps) ? ((p * g.ntaps_w + c_tap_w[t]) * KCt + kc) * g.CmPad + o0 : -1;
  }
  // ---- patch pixel table: byte offset of (b, channel 0, t, h, w); -4 = outside the tensor (DMA writes 0)
  for (int e = tid; e < CSl; e += 256) {
    int off = -4;
    const int row = fast_div(e, g.mg_ew);
    const int ew = e - row * g.eW;
    const int jb = fast_div(row, g.mg_eth);
    const int r2 = row - jb * g.eth;
    const int et = fast_div(r2, g.mg_eh);
    const int eh = r2 - et * g.eH;
    const int b = j0b + jb, t = src_t0 + et, h = src_h0 + eh, w = src_w0 + ew;
    if (b < g.B && (unsigned)t < (unsigned)g.sT && (unsigned)h < (unsigned)g.sH && (unsigned)w < (unsigned)g.sW)
      off = 4 * (((b * g.Ck) * g.sT + t) * sHW + h * g.sW + w);
    ptab[e] = off;
  }
  const int nE = (CSl + 255) >> 8;

  int lane_base[TN];
#pragma unroll
  for (int f = 0; f < TN; ++f) {
    const int pix = (wn * TN + f) * 32 + l31;
    const int jw = pix & JWm;
    const int jh = (pix >> g.ljw) & JHm;
    const int jt = (pix >> (g.ljw + g.ljh)) & JTm;
    const int jb = pix >> (g.ljw + g.ljh + g.ljt);
    lane_base[f] = ((jb * g.eT + jt * g.mT) * g.eH + jh * g.mH) * g.eW + jw * g.mW + lhi * CSl;
  }
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int f = 0; f < TN; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][f][r] = 0.f;

  const v4i32 rs_src = make_rsrc(g.src, g.src_bytes);
  const v4i32 rs_w = make_rsrc(g.wb, g.wb_bytes);
  const unsigned smem_la = lds_base(smem);
  const unsigned ws_la = smem_la + 4u * (wtab_sz + ptab_sz);
  const unsigned stg_la = ws_la + 16u * (unsigned)(R * WS16);
  const int chan_bytes = g.sT * sHW * 4;

  const int ngroups = (c_ntaps + TG - 1) / TG;
  const int nchunks = g.Ck >> 4;
  const int nstages = nchunks * ngroups;
  const int D = R - 1;
  // per-wave DMA instruction counts (identical for the four waves: short waves repeat an instruction)
  const int w_units = 3 * TG * SUBS;
  const int NW = (w_units + 3) >> 2;
  const int p_total = STG >> 6;
  const int NP = (p_total + 3) >> 2;

  auto issue_w = [&](int k) {                        // split weights of stage k -> ring slot k % R
    const int c = k / ngroups, gi = k - c * ngroups, slot = k % R;
    const int soff = c * KC * g.CmPad * 16;
    int u = 0, mine = 0;
    auto one = [&](int p, int tl, int sub) {
      const int idx = sub * 64 + lane;                // (kc, m) inside this (plane, tap)
      const int kc = idx / MB, m = idx % MB;
      const int base = wtab[(p * NTP + gi * TG + tl) * KC + kc];
      const int voff = (base < 0 || o0 + m >= g.CmPad) ? -16 : (base + m) * 16;
      dma_b128(rs_w, ws_la + 16u * (unsigned)(slot * WS16 + ((p * TG + tl) * KC) * MB + sub * 64), voff, soff);
    };
    for (int p = 0; p < 3; ++p)
      for (int tl = 0; tl < TG; ++tl) {
#pragma unroll
        for (int sub = 0; sub < SUBS; ++sub, ++u) {
          if ((u & 3) != wave) continue;
          one(p, tl, sub);
          ++mine;
        }
      }
    if (mine < NW) one(0, 0, 0);                      // pad to NW (same bytes to the same place: harmless)
  };
  auto issue_p = [&](int c) {                        // fp32 patch of chunk c -> staging
    const int soff = c * 16 * chan_bytes;
    for (int i = 0; i < NP; ++i) {
      int q = wave + 4 * i;
      if (q >= p_total) q = p_total - 1;
      const int idx = q * 64 + lane;
      const int k = fast_div(idx, g.mg_csl);
      const int e = idx - k * CSl;
      const int pt = ptab[e];
      const int voff = (k < 16 && pt != -4) ? pt + k * chan_bytes : -4;
      dma_b32(rs_src, stg_la + 256u * (unsigned)q, voff, soff);
    }
  };
  auto split_pass = [&](u32x4* pb) {                 // staging -> three bf16 operand planes
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
#pragma unroll
      for (int it = 0; it < X6_MAXE; ++it)
        if (it < nE) {
          const int e = it * 256 + tid;
          if (e < CSl) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = stg[(kc * 8 + q) * CSl + e];
            u32x4 h, mi, lo;
            split8(v, h, mi, lo);
            pb[(0 * KC + kc) * CSl + e] = h;
            pb[(1 * KC + kc) * CSl + e] = mi;
            pb[(2 * KC + kc) * CSl + e] = lo;
          }
        }
  };
  // number of stages s' in [a, b] that issued a weight DMA batch (s' + D < nstages)
  auto w_issued_in = [&](int a, int b) {
    if (a < 0) a = 0;
    const int hi = min(b, nstages - 1 - D);
    return hi >= a ? hi - a + 1 : 0;
  };

  __syncthreads();                                   // tables visible
  if (nstages > 0) {
    issue_p(0);
    for (int k = 0; k < D && k < nstages; ++k) issue_w(k);
    wait_vm(0);
    __syncthreads();
    split_pass(pb0);
    __syncthreads();                                 // staging free again, patch buffer 0 visible
    if (nchunks > 1) issue_p(1);
  }
  // ---- this wave's DMA jobs, issued in the shadow of the MFMAs (slot = one group of TM*TN MFMAs):
  //      slots [0, NWMAX): one weight DMA each; slots [NWMAX, 18): PPS staging DMAs each
  constexpr int NWMAX = 9, NPMAX = 36, PPS = 4, NSL = 18;
  int wj_p[NWMAX], wj_tl[NWMAX], wj_sub[NWMAX];
#pragma unroll
  for (int jj = 0; jj < NWMAX; ++jj) {
    int u = wave + 4 * jj;
    if (u >= w_units) u = 0;                          // padding job (same bytes to the same place: harmless)
    wj_sub[jj] = u % SUBS;
    const int r = u / SUBS;
    wj_tl[jj] = r % TG;
    wj_p[jj] = r / TG;
  }
  int pv[NPMAX];
#pragma unroll
  for (int jj = 0; jj < NPMAX; ++jj) {
    int q = wave + 4 * jj;
    if (q >= p_total) q = p_total - 1;
    const int idx = q * 64 + lane;
    const int k = fast_div(idx, g.mg_csl);
    const int e = idx - k * CSl;
    const int pt = (jj < NP) ? ptab[e] : -4;
    pv[jj] = (k < 16 && pt != -4) ? pt + k * chan_bytes : -4;
  }
  int wvoff[NWMAX];

  int c = 0, gi = 0;
  for (int s = 0; s < nstages; ++s) {
    const int s0 = s - gi;
    const bool has_next_chunk = c + 1 < nchunks;
    const bool w_active = (s + D < nstages);
    const bool p_active = ngroups > 1 && gi == 0 && c > 0 && has_next_chunk;
    // target of this stage's weight DMA batch: stage s + D
    const int k_w = s + D;
    const int c_w = k_w / ngroups, gi_w = k_w - c_w * ngroups;
    const int w_soff = c_w * KC * g.CmPad * 16;
    const int w_slot16 = (k_w % R) * WS16;
    const int p_soff = (c + 1) * 16 * chan_bytes;
    if (w_active) {
#pragma unroll
      for (int jj = 0; jj < NWMAX; ++jj)
        if (jj < NW) {
          const int idx = wj_sub[jj] * 64 + lane;
          const int kc = idx / MB, m = idx % MB;
          const int base = wtab[(wj_p[jj] * NTP + gi_w * TG + wj_tl[jj]) * KC + kc];
          wvoff[jj] = (base < 0 || o0 + m >= g.CmPad) ? -16 : (base + m) * 16;
        }
    }
    auto do_slot = [&](int sl) {
      if (sl < NWMAX) {
        if (w_active && sl < NW)
          dma_b128(rs_w, ws_la + 16u * (unsigned)(w_slot16 + ((wj_p[sl] * TG + wj_tl[sl]) * KC) * MB + wj_sub[sl] * 64), wvoff[sl], w_soff);
      } else if (p_active) {
#pragma unroll
        for (int r = 0; r < PPS; ++r) {
          const int jj = (sl - NWMAX) * PPS + r;
          if (jj < NP) {
            int q = wave + 4 * jj;
            if (q >= p_total) q = p_total - 1;
            dma_b32(rs_src, stg_la + 256u * (unsigned)q, pv[jj], p_soff);
          }
        }
      }
    };
    if (ngroups > 1 && gi == ngroups - 1 && has_next_chunk) split_pass(pb0 + ((c + 1) & 1) * PB16);

    const u32x4* ws = ws0 + (s % R) * WS16 + lhi * MB + wm * TM * 32 + l31;
    const u32x4* pb = pb0 + (c & 1) * PB16;
    const int t0 = gi * TG;
    const int ntl = min(TG, c_ntaps - t0);
    u32x4 A0[TM][3], B0[TN][3], A1[TM][3], B1[TN][3];
    auto load_tap = [&](int tl, u32x4 (&Ad)[TM][3], u32x4 (&Bd)[TN][3]) {
      const int toff = c_tap_off[t0 + tl];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int i = 0; i < TM; ++i) Ad[i][p] = ws[((p * TG + tl) * KC) * MB + i * 32];
#pragma unroll
        for (int f = 0; f < TN; ++f) Bd[f][p] = pb[p * KC * CSl + lane_base[f] + toff];
      }
    };
    auto mfma_tap = [&](int tl, const u32x4 (&Ad)[TM][3], const u32x4 (&Bd)[TN][3]) {
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0};       // small terms first: lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi
      constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int q = 0; q < 6; ++q) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int f = 0; f < TN; ++f)
            acc[i][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Ad[i][PA[q]]),
                                                                __builtin_bit_cast(bf16x8, Bd[f][PB[q]]), acc[i][f], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        do_slot(tl * 6 + q);                          // DMA issue rides in the shadow of the MFMAs just queued
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    const int ntl_run = ntl;
    if (ntl_run > 0) load_tap(0, A0, B0);
#pragma unroll
    for (int tl = 0; tl < X6_MAXTG; tl += 2) {
      if (tl < ntl_run) {
        if (tl + 1 < ntl_run) load_tap(tl + 1, A1, B1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_tap(tl, A0, B0);
        if (tl + 1 < ntl_run) {
          if (tl + 2 < ntl_run) load_tap(tl + 2, A0, B0);
          __builtin_amdgcn_sched_barrier(0);
          mfma_tap(tl + 1, A1, B1);
        }
      }
    }
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl)
      if (sl >= ntl_run * 6) do_slot(sl);             // jobs whose slot did not run (short tap groups)

    // ---- end of stage: wait only for what stage s+1 needs, then hand over
    if (ngroups == 1) {
      // one stage per chunk (1-tap classes): no spare stage to hide the split pass behind
      wait_vm(0);
      __syncthreads();
      if (has_next_chunk) {
        split_pass(pb0 + ((c + 1) & 1) * PB16);
        __syncthreads();
        if (c + 2 < nchunks) issue_p(c + 2);
      }
    } else {
      const bool p_out = has_next_chunk && gi <= ngroups - 2;     // staging DMA of chunk c+1 issued and not yet waited for
      const int sp = (c == 0) ? -1 : s0;                          // stage that issued it (after that stage's weight batch)
      int n = 63;
      if (s + 1 < nstages && s + 1 >= D) {                        // weights of stage s+1 were issued in stage s+1-D
        const int sw = s + 1 - D;
        int m = NW * w_issued_in(sw + 1, s);
        if (p_out && sp >= sw) m += NP;
        n = min(n, m);
      }
      if (p_out && gi == ngroups - 2) n = min(n, NW * w_issued_in(sp + 1, s));   // staging must have landed for the split pass
      wait_vm(n);
      __syncthreads();
    }
    if (++gi == ngroups) { gi = 0; ++c; }
  }

  // ---- epilogue (C/D map of the 32x32 MFMA: col = lane&31 (pixel), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (dest channel))
  const int dHW = g.dH * g.dW;
#pragma unroll
  for (int f = 0; f < TN; ++f) {
    const int pix = (wn * TN + f) * 32 + l31;
    const int gw = j0w + (pix & JWm);
    const int gh = j0h + ((pix >> g.ljw) & JHm);
    const int gt = j0t + ((pix >> (g.ljw + g.ljh)) & JTm);
    const int gb = j0b + (pix >> (g.ljw + g.ljh + g.ljt));
    const bool pv = gb < g.B && gt < c_nT && gh < c_nH && gw < c_nW;
    const int sp = (gt * g.oT + c_pT) * dHW + (gh * g.oH + c_pH) * g.dW + gw * g.oW + c_pW;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = o0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (pv && o < g.Cm) {
          float v = acc[i][f][r];
          if (g.bias) v += g.bias[o];
          v = act_apply(v, g.act_epi);
          const size_t di = ((size_t)(gb * g.Cm + o)) * g.dT * dHW + sp;
          if (g.res) v += g.res[di];
          if (g.mask_y) v = act_grad(v, g.mask_y[di], g.mask_act);
          g.dst[di] = v;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------ host side
struct X6Tile { int TM, TN, WM; };
static const X6Tile kTiles[] = {{2, 4, 2}, {2, 2, 1}, {2, 2, 2}, {2, 1, 1}, {1, 2, 1}, {1, 1, 1}};
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);

template <int TM, int TN, int WM>
static int launch_x6(const PatchGeom& g, dim3 grid, size_t lds, hipStream_t s) {
  auto k = patch_gemm_x6_kernel<TM, TN, WM>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(k, grid, dim3(256), lds, s, g);
  return launch_status();
}

static int dispatch_x6(const X6Tile& t, const PatchGeom& g, dim3 grid, size_t lds, hipStream_t s) {
#define P2I_CASE(tm_, tn_, wm_) \
  if (t.TM == tm_ && t.TN == tn_ && t.WM == wm_) return launch_x6<tm_, tn_, wm_>(g, grid, lds, s);
  P2I_CASE(2, 4, 2)
  P2I_CASE(2, 2, 1)
  P2I_CASE(2, 2, 2)
  P2I_CASE(2, 1, 1)
  P2I_CASE(1, 2, 1)
  P2I_CASE(1, 1, 1)
#undef P2I_CASE
  set_error("no x6 kernel instance for tile %d %d %d", t.TM, t.TN, t.WM);
  return P2I_EINVAL;
}

X6Ctx& x6_ctx() {
  static thread_local X6Ctx c{nullptr, 0};
  return c;
}

int run_patch_gemm_x6(PatchGeom g, const ClassSpec* css, int ncls, const uint16_t* wb, int ntaps_w, int* plan6, hipStream_t s) {
  if (g.src_y != nullptr || ncls > MAX_CLASSES || (g.Ck & 15) != 0 || wb == nullptr) return 1;
  static const int disabled = getenv("P2I_CONV_X6") ? (atoi(getenv("P2I_CONV_X6")) == 0) : 0;
  if (disabled) return 1;
  const ClassSpec& c0s = css[0];
  g.mT = c0s.mT; g.mH = c0s.mH; g.mW = c0s.mW;
  g.oT = c0s.oT; g.oH = c0s.oH; g.oW = c0s.oW;
  int lo[MAX_CLASSES][3], rng[3] = {0, 0, 0}, mx[3] = {0, 0, 0}, max_taps = 0;
  long long total_pix = 0;
  for (int q = 0; q < ncls; ++q) {
    const ClassSpec& cs = css[q];
    int hi[3] = {0, 0, 0};
    lo[q][0] = lo[q][1] = lo[q][2] = 0;
    for (int i = 0; i < cs.ntaps; ++i) {
      const int d[3] = {cs.dt[i], cs.dh[i], cs.dw[i]};
      for (int k = 0; k < 3; ++k) {
        if (i == 0 || d[k] < lo[q][k]) lo[q][k] = d[k];
        if (i == 0 || d[k] > hi[k]) hi[k] = d[k];
      }
    }
    for (int k = 0; k < 3; ++k) if (hi[k] - lo[q][k] > rng[k]) rng[k] = hi[k] - lo[q][k];
    if (cs.nT > mx[0]) mx[0] = cs.nT;
    if (cs.nH > mx[1]) mx[1] = cs.nH;
    if (cs.nW > mx[2]) mx[2] = cs.nW;
    if (cs.ntaps > max_taps) max_taps = cs.ntaps;
    total_pix += (long long)g.B * (cs.nT > 0 ? cs.nT : 0) * (cs.nH > 0 ? cs.nH : 0) * (cs.nW > 0 ? cs.nW : 0);
  }
  if (mx[0] <= 0 || mx[1] <= 0 || mx[2] <= 0) return P2I_OK;
  if (max_taps == 0) return 1;
  static const int force_tg = getenv("P2I_X6_TG") ? atoi(getenv("P2I_X6_TG")) : 0;
  static const int force_r = getenv("P2I_X6_R") ? atoi(getenv("P2I_X6_R")) : 0;
  const int TG = (force_tg >= 1 && force_tg <= X6_MAXTG) ? force_tg : ((max_taps % 3 == 0) ? 3 : (max_taps == 1 ? 1 : 2));
  const int NTP = ceil_div(max_taps, TG) * TG;
  static const int force_tile = getenv("P2I_X6_TILE") ? atoi(getenv("P2I_X6_TILE")) : -1;

  int best = -1;
  long long best_score = -1;
  PatchGeom bg = g;
  size_t best_lds = 0;
  dim3 best_grid;
  for (int ci = 0; ci < kNumTiles; ++ci) {
    if (force_tile >= 0 && ci != force_tile) continue;
    const X6Tile& tl = kTiles[ci];
    const int MBc = 32 * tl.TM * tl.WM, NP = 32 * tl.TN * (4 / tl.WM);
    if (MBc > 32 && g.Cm <= MBc / 2) continue;            // more than half of the m-tile would be padding
    int jb, jt, jh, jw;
    pick_tile_dims(NP, g.B, mx[0], mx[1], mx[2], jb, jt, jh, jw);
    PatchGeom t = g;
    t.ljb = ilog2(jb); t.ljt = ilog2(jt); t.ljh = ilog2(jh); t.ljw = ilog2(jw);
    t.eT = (jt - 1) * g.mT + rng[0] + 1;
    t.eH = (jh - 1) * g.mH + rng[1] + 1;
    t.eW = (jw - 1) * g.mW + rng[2] + 1;
    t.eWp = t.eW;
    t.eth = t.eT * t.eH;
    t.rpc = jb * t.eth;
    t.CSl = t.rpc * t.eW;
    t.CS = t.CSl;
    if (t.CSl > 576) continue;                            // staging DMA jobs per wave: ceil(CSl/16) <= NPMAX
    const size_t fixed = (size_t)((3 * NTP * 2 + 3) & ~3) * 4 + (size_t)((t.CSl + 3) & ~3) * 4 + (size_t)((16 * t.CSl + 63) & ~63) * 4 +
                         2 * (size_t)(3 * 2 * t.CSl) * 16;
    const size_t slot = (size_t)(3 * TG * 2 * MBc) * 16;
    // ring depth: enough stages in flight to cover ~2.5 us of DMA latency (a stage is TG*TM*TN*6 MFMAs of 32 cycles), >= 3
    int Rc = force_r > 0 ? force_r : 1 + (6000 + TG * tl.TM * tl.TN * 192 - 1) / (TG * tl.TM * tl.TN * 192);
    if (Rc < 3) Rc = 3;
    if (Rc > X6_MAXR) Rc = X6_MAXR;
    while (Rc > 2 && fixed + Rc * slot > 160 * 1024) --Rc;
    const size_t lds = fixed + Rc * slot;
    if (lds > 160 * 1024) continue;
    t.R = Rc;
    const long long nb = (long long)ceil_div(g.Cm, MBc) * ((total_pix + NP - 1) / NP);
    // score: fill the chip first (>= 1 workgroup per CU; two when two fit), then MFMA work per staged byte (tile area)
    const long long want = (lds <= 80 * 1024) ? 512 : 256;
    long long score = (nb >= want ? 1000000000ll : (nb >= 256 ? 900000000ll + nb : nb * 1000000ll)) + (long long)MBc * NP;
    if (score > best_score) {
      best_score = score; best = ci; best_lds = lds;
      bg = t;
      bg.ntt = ceil_div(mx[0], jt); bg.nth = ceil_div(mx[1], jh); bg.ntw = ceil_div(mx[2], jw);
      best_grid = dim3((unsigned)(ceil_div(g.B, jb) * bg.ntt * bg.nth * bg.ntw), (unsigned)ceil_div(g.Cm, MBc), (unsigned)ncls);
    }
  }
  const unsigned long long sbytes = 4ull * g.B * g.Ck * g.sT * g.sH * g.sW;
  if (best < 0 || sbytes >= 0x7FFFFFF0ull) return 1;
  bg.mg_ew = magic_u16(bg.eW); bg.mg_eth = magic_u16(bg.eth); bg.mg_eh = magic_u16(bg.eH); bg.mg_csl = magic_u16(bg.CSl);
  bg.src_bytes = (unsigned)sbytes;
  bg.wb = wb;
  bg.ntaps_w = ntaps_w;
  bg.wb_bytes = 3u * (unsigned)ntaps_w * (unsigned)g.Ck * (unsigned)g.CmPad * 2u;
  bg.TG = TG; bg.NTP = NTP;
  bg.nclass = ncls;
  bg.ntaps = max_taps;
  for (int q = 0; q < ncls; ++q) {
    const ClassSpec& cs = css[q];
    ClassGeom& cgm = bg.cls[q];
    cgm.nT = cs.nT; cgm.nH = cs.nH; cgm.nW = cs.nW;
    cgm.pT = cs.pT; cgm.pH = cs.pH; cgm.pW = cs.pW;
    cgm.bT = lo[q][0]; cgm.bH = lo[q][1]; cgm.bW = lo[q][2];
    cgm.ntaps = cs.ntaps;
    for (int i = 0; i < cs.ntaps; ++i) {
      cgm.tap_w[i] = cs.tw[i];
      cgm.tap_off[i] = ((cs.dt[i] - lo[q][0]) * bg.eH + (cs.dh[i] - lo[q][1])) * bg.eW + (cs.dw[i] - lo[q][2]);
    }
  }
  bg.nT = css[0].nT; bg.nH = css[0].nH; bg.nW = css[0].nW;
  bg.pT = css[0].pT; bg.pH = css[0].pH; bg.pW = css[0].pW;
  bg.bT = lo[0][0]; bg.bH = lo[0][1]; bg.bW = lo[0][2];
  for (int i = 0; i < css[0].ntaps; ++i) { bg.tap_w[i] = bg.cls[0].tap_w[i]; bg.tap_off[i] = bg.cls[0].tap_off[i]; }
  if (ncls == 1) bg.ntaps = css[0].ntaps;
  const X6Tile& bt = kTiles[best];
  if (plan6) { plan6[0] = bt.TM; plan6[1] = bt.TN; plan6[2] = bt.WM; plan6[3] = TG; plan6[4] = ncls > 1 ? 0 : max_taps; plan6[5] = 6; }
  return dispatch_x6(bt, bg, best_grid, best_lds, s);
}

int x6_split_weights(const float* wp, uint16_t* wb, int ntaps, int Ck, int CmPad, hipStream_t s) {
  const int total = ntaps * (Ck >> 3) * CmPad;
  const int blocks = total > 0 ? (total + 255) / 256 : 1;
  hipLaunchKernelGGL(wsplit_kernel, dim3(blocks > 2048 ? 2048 : blocks), dim3(256), 0, s, wp, reinterpret_cast<u32x4*>(wb), ntaps, Ck, CmPad);
  return launch_status();
}

}  // namespace p2i
