// Strided-convolution data gradient with the input-parity classes FUSED in one workgroup.
//
// dgrad of a stride-s convolution splits the input positions into s_t*s_h*s_w parity classes; class p only sees the
// taps with (p + pad - tap) % s == 0 (1, 2, 2 and 4 of a 3x3 kernel at stride 2).  conv.hip runs the classes as
// blockIdx.z of one launch: every class stages its own dy patch, and the 1- and 2-tap classes are DMA-bound (one
// MFMA pass per staged element).  Here ONE workgroup stages the dy patch once per 16-channel chunk and multiplies it
// by the taps of ALL classes (9 of 9, 27 of 27) into one accumulator tile per class, i.e. the arithmetic intensity of
// a stride-1 3x3 convolution; the epilogue scatters the class tiles to their interleaved dx positions.
// Same LDS-DMA double-buffered pipeline, tables and epilogue as patch_gemm_dma_kernel (conv.hip).
#include "conv_common.h"

namespace p2i {

template <int MB, int NPIX, int WAVES_M, int CK, int NCLS>
__global__ __launch_bounds__(256) void patch_gemm_fused_kernel(const PatchGeom g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int WAVES_N = 4 / WAVES_M;
  constexpr int TM = MB / (32 * WAVES_M);
  constexpr int TN = NPIX / (32 * WAVES_N);
  constexpr int V = MB / 4;                          // float4 per weight row
  const int nwrows = g.ntaps * CK;                   // g.ntaps = taps of all classes, concatenated class by class
  const int WSZ = ((nwrows * V + 63) & ~63) * 4;
  const int PT4p = ((g.PT >> 2) + 63) & ~63;
  const int PTp = g.v4 ? PT4p * 4 : (g.PT + 63) & ~63;
  int* wtab = reinterpret_cast<int*>(smem);
  const int wtab_sz = (nwrows + 63) & ~63;
  int* ptab = wtab + wtab_sz;
  float* buf0 = smem + wtab_sz + PTp;
  const int BUFSZ = WSZ + PTp;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
  const int src_t0 = j0t * g.mT + g.bT, src_h0 = j0h * g.mH + g.bH, src_w0 = j0w * g.mW + g.bW;   // common patch origin

  for (int r = tid; r < wtab_sz; r += 256) {
    int off = -4;
    if (r < nwrows) {
      const int tap = r / CK, c = r - tap * CK;
      if (c < g.Ck && o0 < g.CmPad) off = ((g.tap_w[tap] * g.Ck + c) * g.CmPad + o0) * 4;
    }
    wtab[r] = off;
  }
  if (g.v4) {                                        // 16-B patch DMA: see patch_gemm_dma_kernel (conv.hip)
    for (int e = tid; e < PT4p; e += 256) {
      int off = -16;
      if (e < (g.PT >> 2)) {
        const int c = fast_div(e, g.mg_g4);
        int rem = e - c * g.G4;
        const int row = g.eW4 == 1 ? rem : fast_div(rem, g.mg_ew4);
        const int g4 = rem - row * g.eW4;
        const int jb = fast_div(row, g.mg_eth);
        int r2 = row - jb * g.eth;
        const int et = fast_div(r2, g.mg_eh);
        const int eh = r2 - et * g.eH;
        const int b = j0b + jb, t = src_t0 + et, h = src_h0 + eh, w = src_w0 + 4 * g4;
        if (b < g.B && c < g.Ck && (unsigned)t < (unsigned)g.sT && (unsigned)h < (unsigned)g.sH && (unsigned)w < (unsigned)g.sW)
          off = ((((b * g.Ck + c) * g.sT + t) * sHW) + h * g.sW + w) * 4;
      }
      ptab[e] = off;
    }
  } else
  for (int e = tid; e < PTp; e += 256) {
    int off = -4;
    if (e < g.PT) {
      const int c = fast_div(e, g.mg_csl);
      int rem = e - c * g.CSl;
      const int row = fast_div(rem, g.mg_ew);
      const int ew = rem - row * g.eW;
      const int jb = fast_div(row, g.mg_eth);
      int r2 = row - jb * g.eth;
      const int et = fast_div(r2, g.mg_eh);
      const int eh = r2 - et * g.eH;
      const int b = j0b + jb, t = src_t0 + et, h = src_h0 + eh, w = src_w0 + ew;
      if (b < g.B && c < g.Ck && (unsigned)t < (unsigned)g.sT && (unsigned)h < (unsigned)g.sH && (unsigned)w < (unsigned)g.sW)
        off = ((((b * g.Ck + c) * g.sT + t) * sHW) + h * g.sW + w) * 4;
    }
    ptab[e] = off;
  }

  int lane_base[TN];
#pragma unroll
  for (int f = 0; f < TN; ++f) {
    const int pix = (wn * TN + f) * 32 + l31;
    const int jw = pix & JWm;
    const int jh = (pix >> g.ljw) & JHm;
    const int jt = (pix >> (g.ljw + g.ljh)) & JTm;
    const int jb = pix >> (g.ljw + g.ljh + g.ljt);
    lane_base[f] = ((jb * g.eT + jt * g.mT) * g.eH + jh * g.mH) * g.eW + jw * g.mW + lhi * g.CSl;
  }
  f32x16 acc[NCLS][TM][TN];
#pragma unroll
  for (int q = 0; q < NCLS; ++q)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int f = 0; f < TN; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][i][f][r] = 0.f;

  const v4i32 rs_src = make_rsrc(g.src, g.src_bytes);
  const v4i32 rs_w = make_rsrc(g.wp, g.wp_bytes);
  const unsigned smem_la = lds_base(smem);
  const int buf0_off = (int)(buf0 - smem);
  const int chan_bytes = g.sT * sHW * 4;
  const int wbase = tid & ~63;
  const int nwv = nwrows * V;
  __syncthreads();

  // few DMA lanes per thread: offsets in registers, no LDS lookup chain in the issue (see patch_gemm_dma_kernel)
  constexpr int RW = 12, RP = 4;
  const int nwq = (((nwv + 63) & ~63) + 255) / 256;
  const int npq = g.v4 ? (PT4p + 255) / 256 : RP + 1;
  const bool reg_issue = nwq <= RW && npq <= RP;
  int wv[RW], pv[RP];
  if (reg_issue) {
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const int f = i * 256 + tid;
      int voff = -4;
      if (i < nwq && f < nwv) {
        const int row = f / V, col4 = f - row * V;
        const int base = wtab[row];
        voff = base < 0 ? -4 : base + col4 * 16;
      }
      wv[i] = voff;
    }
#pragma unroll
    for (int i = 0; i < RP; ++i) {
      const int e = i * 256 + tid;
      pv[i] = (i < npq && e < PT4p) ? ptab[e] : -16;
    }
  }
  auto issue = [&](int c0, int bufoff) {
    const int w_soff = c0 * g.CmPad * 4;
    if (reg_issue) {
      const int p_soff = c0 * chan_bytes, pbo = bufoff + WSZ;
#pragma unroll
      for (int i = 0; i < RW; ++i)
        if (i < nwq && i * 256 + wbase < ((nwv + 63) & ~63))
          dma_b128(rs_w, smem_la + 4u * (bufoff + (i * 256 + wbase) * 4), wv[i], w_soff);
#pragma unroll
      for (int i = 0; i < RP; ++i)
        if (i < npq && i * 256 + wbase < PT4p)
          dma_b128(rs_src, smem_la + 4u * (pbo + (i * 256 + wbase) * 4), pv[i], p_soff);
      return;
    }
    for (int f0 = 0; f0 < nwv; f0 += 256) {
      const int f = f0 + tid;
      int voff = -4;
      if (f < nwv) {
        const int row = f / V, col4 = f - row * V;
        const int base = wtab[row];
        voff = base < 0 ? -4 : base + col4 * 16;
      }
      if (f0 + wbase < ((nwv + 63) & ~63))
        dma_b128(rs_w, smem_la + 4u * (bufoff + (f0 + wbase) * 4), voff, w_soff);
    }
    const int pbo = bufoff + WSZ;
    const int p_soff = c0 * chan_bytes;
    if (g.v4) {
      for (int e0 = 0; e0 < PT4p; e0 += 256) {
        const int e = e0 + tid;
        const int voff = e < PT4p ? ptab[e] : -16;
        if (e0 + wbase < PT4p) dma_b128(rs_src, smem_la + 4u * (pbo + (e0 + wbase) * 4), voff, p_soff);
      }
    } else
    for (int e0 = 0; e0 < PTp; e0 += 256) {
      const int e = e0 + tid;
      const int voff = e < PTp ? ptab[e] : -4;
      if (e0 + wbase < PTp) dma_b32(rs_src, smem_la + 4u * (pbo + e0 + wbase), voff, p_soff);
    }
  };

  int cnt[NCLS];
#pragma unroll
  for (int q = 0; q < NCLS; ++q) cnt[q] = g.cls[q].ntaps;

  const int nchunks = (g.Ck + CK - 1) / CK;
  issue(0, buf0_off);
  for (int k = 0; k < nchunks; ++k) {
    dma_wait_all();
    __syncthreads();
    float* cur = buf0 + (k & 1) * BUFSZ;
    if (k + 1 < nchunks) issue((k + 1) * CK, buf0_off + ((k + 1) & 1) * BUFSZ);
    const float* lw = cur + lhi * MB + wm * TM * 32 + l31;
    const float* lp = cur + WSZ;
    constexpr int NCP = CK / 2;
    int tbase = 0;
#pragma unroll
    for (int q = 0; q < NCLS; ++q) {
      const int nt = cnt[q];
      float a[NCP][TM], bv[NCP][TN], an[NCP][TM], bn[NCP][TN];
      auto load_tap = [&](int tap, float (&aa)[NCP][TM], float (&bb)[NCP][TN]) {
        const int toff = g.tap_off[tap];
#pragma unroll
        for (int cp = 0; cp < NCP; ++cp) {
#pragma unroll
          for (int i = 0; i < TM; ++i) aa[cp][i] = lw[tap * CK * MB + cp * 2 * MB + i * 32];
#pragma unroll
          for (int f = 0; f < TN; ++f) bb[cp][f] = lp[lane_base[f] + toff + cp * 2 * g.CSl];
        }
      };
      auto mfma_tap = [&](const float (&aa)[NCP][TM], const float (&bb)[NCP][TN]) {
#pragma unroll
        for (int cp = 0; cp < NCP; ++cp)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int f = 0; f < TN; ++f)
              acc[q][i][f] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[cp][i], bb[cp][f], acc[q][i][f], 0, 0, 0);
      };
      if (nt > 0) load_tap(tbase, a, bv);
      for (int t = 0; t < nt; t += 2) {
        if (t + 1 < nt) load_tap(tbase + t + 1, an, bn);
        __builtin_amdgcn_sched_barrier(0);
        mfma_tap(a, bv);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < nt) {
          if (t + 2 < nt) load_tap(tbase + t + 2, a, bv);
          __builtin_amdgcn_sched_barrier(0);
          mfma_tap(an, bn);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      tbase += nt;
    }
  }

  const int dHW = g.dH * g.dW;
  // classes are ordered (pT, pH, pW) with pW fastest (conv.hip): q = 2k, 2k+1 differ in pW only.  With an even destination row they
  // cover the same (t, h, w/2) grid and interleave along w: stored together as float2 (g.pair_w, set by the host).
  if (NCLS == 4 && g.pair_w) {
#pragma unroll
    for (int q = 0; q < NCLS; q += 2) {
      const ClassGeom& cg = g.cls[q];
      const int c_nT = cg.nT, c_nH = cg.nH, c_nW = cg.nW, c_pT = cg.pT, c_pH = cg.pH, c_pW = cg.pW;
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
        for (int i = 0; i < TM; ++i)
          epilogue_pair16(acc[q][i][f], acc[q + 1][i][f], o0 + (wm * TM + i) * 32, lhi, g.Cm, pv, (size_t)gb * g.Cm * g.dT * dHW + sp,
                          (size_t)g.dT * dHW, g.res, g.mask_y, g.mask_act, g.dst);
      }
    }
    return;
  }
#pragma unroll
  for (int q = 0; q < NCLS; ++q) {
    const ClassGeom& cg = g.cls[q];
    const int c_nT = cg.nT, c_nH = cg.nH, c_nW = cg.nW, c_pT = cg.pT, c_pH = cg.pH, c_pW = cg.pW;
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
      for (int i = 0; i < TM; ++i)
        epilogue_tile16(acc[q][i][f], o0 + (wm * TM + i) * 32, lhi, g.Cm, pv, (size_t)gb * g.Cm * g.dT * dHW + sp, (size_t)g.dT * dHW,
                        nullptr, P2I_ACT_NONE, g.res, g.mask_y, g.mask_act, g.dst);
    }
  }
}

template <int MB, int NPIX, int WM, int CK, int NCLS>
static int launch_fused(const PatchGeom& g, dim3 grid, size_t lds, hipStream_t s) {
  auto k = patch_gemm_fused_kernel<MB, NPIX, WM, CK, NCLS>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  P2I_LAUNCH(k, grid, dim3(256), lds, s, g);
  return launch_status();
}

// returns 1 when the fused kernel does not apply (caller falls back to the per-class launch)
int run_patch_gemm_fused(PatchGeom g, const ClassSpec* css, int ncls, int* plan6, hipStream_t s) {
  static const int disabled = getenv("P2I_DGRAD_FUSED") ? (atoi(getenv("P2I_DGRAD_FUSED")) == 0) : 0;
  // (two-class launches, i.e. stride (2,1,1) with 27 taps, were measured slower fused: 61.6 vs 68.6 TF; they stay per class)
  if (disabled || ncls != 4 || g.src_y != nullptr || g.bias != nullptr || g.act_epi != P2I_ACT_NONE) return 1;
  const ClassSpec& c0s = css[0];
  int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0}, mx[3] = {0, 0, 0}, ntt = 0;
  bool first = true;
  for (int q = 0; q < ncls; ++q) {
    const ClassSpec& cs = css[q];
    if (cs.mT != 1 || cs.mH != 1 || cs.mW != 1 || cs.oT != c0s.oT || cs.oH != c0s.oH || cs.oW != c0s.oW) return 1;
    for (int i = 0; i < cs.ntaps; ++i) {
      const int d[3] = {cs.dt[i], cs.dh[i], cs.dw[i]};
      for (int k = 0; k < 3; ++k) {
        if (first || d[k] < lo[k]) lo[k] = d[k];
        if (first || d[k] > hi[k]) hi[k] = d[k];
      }
      first = false;
    }
    if (cs.nT > mx[0]) mx[0] = cs.nT;
    if (cs.nH > mx[1]) mx[1] = cs.nH;
    if (cs.nW > mx[2]) mx[2] = cs.nW;
    ntt += cs.ntaps;
  }
  if (mx[0] <= 0 || mx[1] <= 0 || mx[2] <= 0 || ntt == 0 || ntt > MAX_TAPS) return 1;
  {
    // float2 epilogue: classes (2k, 2k+1) = (.., pW 0), (.., pW 1) over the same grid, even destination row, 8-byte aligned tensors
    static const int pair_on = getenv("P2I_DGRAD_PAIR") ? atoi(getenv("P2I_DGRAD_PAIR")) : 1;
    bool pair = pair_on && c0s.oW == 2 && (g.dW & 1) == 0;
    for (int q = 0; q < ncls && pair; q += 2)
      pair = css[q].pW == 0 && css[q + 1].pW == 1 && css[q].pT == css[q + 1].pT && css[q].pH == css[q + 1].pH &&
             css[q].nT == css[q + 1].nT && css[q].nH == css[q + 1].nH && css[q].nW == css[q + 1].nW;
    const unsigned long long al = (unsigned long long)g.dst | (unsigned long long)g.res | (unsigned long long)g.mask_y;
    g.pair_w = (pair && (al & 7ull) == 0) ? 1 : 0;
  }
  g.mT = g.mH = g.mW = 1;
  g.oT = c0s.oT; g.oH = c0s.oH; g.oW = c0s.oW;
  // 64-channel m-tiles unless that leaves half of the CUs without a workgroup (deep, small layers)
  long long px = 0;
  for (int q = 0; q < ncls; ++q) px = px > (long long)g.B * css[q].nT * css[q].nH * css[q].nW ? px : (long long)g.B * css[q].nT * css[q].nH * css[q].nW;
  const bool few = g.Cm > 32 && (long long)ceil_div(g.Cm, 64) * ((px + 127) / 128) < 200;
  const int MBc = (g.Cm > 32 && !few) ? 64 : 32, NP = 128, WM = MBc == 64 ? 2 : 1;
  const int CKc = ntt <= 9 ? 16 : 4;
  if (g.Ck % CKc != 0) return 1;
  int jb, jt, jh, jw;
  pick_tile_dims(NP, g.B, mx[0], mx[1], mx[2], jb, jt, jh, jw);
  g.ljb = ilog2(jb); g.ljt = ilog2(jt); g.ljh = ilog2(jh); g.ljw = ilog2(jw);
  g.eT = (jt - 1) + hi[0] - lo[0] + 1;
  g.eH = (jh - 1) + hi[1] - lo[1] + 1;
  g.eW = (jw - 1) + hi[2] - lo[2] + 1;
  {
    static const int v4_off = getenv("P2I_CONV_V4") ? (atoi(getenv("P2I_CONV_V4")) == 0) : 0;
    g.v4 = (!v4_off && (g.sW & 3) == 0 && jw >= 4) ? 1 : 0;
    g.v4sh = 0;
    if (g.v4) {
      g.v4sh = ((lo[2] % 4) + 4) % 4;
      g.eW = (g.eW + g.v4sh + 3) & ~3;
      g.eW4 = g.eW >> 2;
    }
  }
  g.eWp = g.eW;
  g.eth = g.eT * g.eH;
  g.rpc = jb * g.eth;
  g.CSl = g.rpc * g.eW;
  g.CS = g.CSl;
  g.PT = CKc * g.CSl;
  if (g.PT >= 65536 || g.CSl >= 65536) return 1;
  const int PTp = g.v4 ? (((g.PT >> 2) + 63) & ~63) * 4 : (g.PT + 63) & ~63;
  const int nwrows = ntt * CKc;
  g.G4 = g.CSl >> 2;
  const size_t WSZ = (size_t)((nwrows * (MBc / 4) + 63) & ~63) * 4;
  const size_t lds = sizeof(float) * (((nwrows + 63) & ~63) + (size_t)PTp + 2 * (WSZ + PTp));
  const unsigned long long sbytes = 4ull * g.B * g.Ck * g.sT * g.sH * g.sW;
  if (lds > 160 * 1024 || sbytes >= 0xF0000000ull) return 1;
  g.mg_csl = magic_u16(g.CSl); g.mg_ew = magic_u16(g.eW);
  if (g.v4) { g.mg_g4 = magic_u16(g.G4); g.mg_ew4 = magic_u16(g.eW4); }
  g.mg_rpc = magic_u16(g.rpc); g.mg_eth = magic_u16(g.eth); g.mg_eh = magic_u16(g.eH);
  g.src_bytes = (unsigned)sbytes;
  g.nclass = ncls;
  g.ntaps = ntt;
  g.bT = lo[0]; g.bH = lo[1]; g.bW = lo[2] - g.v4sh;          // v4: aligned origin, v4sh columns further left
  g.nT = mx[0]; g.nH = mx[1]; g.nW = mx[2];
  int tix = 0;
  for (int q = 0; q < ncls; ++q) {
    const ClassSpec& cs = css[q];
    ClassGeom& cgm = g.cls[q];
    cgm.nT = cs.nT; cgm.nH = cs.nH; cgm.nW = cs.nW;
    cgm.pT = cs.pT; cgm.pH = cs.pH; cgm.pW = cs.pW;
    cgm.bT = lo[0]; cgm.bH = lo[1]; cgm.bW = lo[2];
    cgm.ntaps = cs.ntaps;
    for (int i = 0; i < cs.ntaps; ++i, ++tix) {
      g.tap_w[tix] = cs.tw[i];
      g.tap_off[tix] = ((cs.dt[i] - lo[0]) * g.eH + (cs.dh[i] - lo[1])) * g.eW + (cs.dw[i] - lo[2]) + g.v4sh;
    }
  }
  g.ntt = ceil_div(mx[0], jt); g.nth = ceil_div(mx[1], jh); g.ntw = ceil_div(mx[2], jw);
  const dim3 grid((unsigned)(ceil_div(g.B, jb) * g.ntt * g.nth * g.ntw), (unsigned)ceil_div(g.Cm, MBc), 1u);
  if (plan6) { plan6[0] = MBc; plan6[1] = NP; plan6[2] = WM; plan6[3] = CKc; plan6[4] = 0; plan6[5] = 10 + ncls; }
#define P2I_CASE(mb_, wm_, ck_, nc_) \
  if (MBc == mb_ && CKc == ck_ && ncls == nc_) return launch_fused<mb_, 128, wm_, ck_, nc_>(g, grid, lds, s);
  P2I_CASE(64, 2, 16, 4)
  P2I_CASE(32, 1, 16, 4)
  P2I_CASE(64, 2, 4, 4)
  P2I_CASE(32, 1, 4, 4)
#undef P2I_CASE
  return 1;
}

}  // namespace p2i
