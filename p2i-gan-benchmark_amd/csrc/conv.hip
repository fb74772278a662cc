// Convolution engine for gfx950: LDS-staged input patches + v_mfma_f32_32x32x2_f32.
//
// One "patch GEMM" kernel serves forward and data-gradient of every 2-D / 3-D convolution on the
// P2I-GAN hot path (DO-Conv 3x3 stack deconv_pytorch.py:103-109, UPPos 1x1 layer.py:390,
// spectral-norm Conv2d/Conv3d p2igan.py:120-142); a second kernel computes weight gradients.
//
// Formulation (NC(T)HW fp32, exact-f32 MFMA):
//   dest[b, m, j] = epilogue( sum_{tap, k} Wp[tap][k][m] * src[b, k, j*S + delta_tap] )
// j is a dest-local (t,h,w) index, S the source multiplier, delta_tap a per-tap source offset.
//   forward         : S = conv stride, delta = tap - pad, all taps, dest index = j
//   dgrad, class p  : S = 1, delta = (p + pad - tap)/stride for the taps with (p+pad-tap) % stride == 0,
//                     dest index = stride*j + p   (one launch per parity class; stride 1 => one class)
// A workgroup (4 waves) owns MB dest channels x NPIX dest positions (a (b,t,h,w) box with
// power-of-two sides).  Per CK source channels it stages the source patch (box*S + tap halo,
// zero-filled at borders, optionally multiplied by act'(y) for dgrad) and the CK*ntaps*MB packed
// weights into LDS; each MFMA B operand is then ONE ds_read_b32 at lane_base + tap_offset, each A
// operand one ds_read_b32 of the m-contiguous packed weights.
#include "conv_dma.h"

namespace p2i {


template <int MB, int NPIX, int WAVES_M, int CK>
__global__ __launch_bounds__(256) void patch_gemm_kernel(const PatchGeom g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int WAVES_N = 4 / WAVES_M;
  constexpr int TM = MB / (32 * WAVES_M);
  constexpr int TN = NPIX / (32 * WAVES_N);
  static_assert(TM >= 1 && TN >= 1, "tile too small");
  float* lw = smem;                                  // [ntaps][CK][MB]
  float* lp = smem + g.ntaps * CK * MB;              // [CK][CS]

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

  int lane_base[TN];
#pragma unroll
  for (int f = 0; f < TN; ++f) {
    const int pix = (wn * TN + f) * 32 + l31;
    const int jw = pix & JWm;
    const int jh = (pix >> g.ljw) & JHm;
    const int jt = (pix >> (g.ljw + g.ljh)) & JTm;
    const int jb = pix >> (g.ljw + g.ljh + g.ljt);
    lane_base[f] = ((jb * g.eT + jt * g.mT) * g.eH + jh * g.mH) * g.eWp + jw * g.mW + lhi * g.CS;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int f = 0; f < TN; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][f][r] = 0.f;

  const int src_t0 = j0t * g.mT + g.bT, src_h0 = j0h * g.mH + g.bH, src_w0 = j0w * g.mW + g.bW;
  const int sHW = g.sH * g.sW;
  const int rows = CK * g.rpc;
  const int hw = tid >> 5;           // 8 half-waves stage patch rows
  const int hl = tid & 31;

  for (int c0 = 0; c0 < g.Ck; c0 += CK) {
    __syncthreads();
    // ---- packed weights: ntaps*CK rows of MB floats (float4, m-contiguous)
    {
      constexpr int V = MB / 4;                 // float4 per row
      constexpr int RPP = 256 / V;              // rows per pass
      const int v = tid % V, r0 = tid / V;
      const int nrows = g.ntaps * CK;
      for (int r = r0; r < nrows; r += RPP) {
        const int tap = r / CK, c = r - tap * CK;
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c0 + c < g.Ck && o0 + v * 4 < g.CmPad)
          val = *reinterpret_cast<const float4*>(g.wp + ((size_t)(g.tap_w[tap] * g.Ck + c0 + c)) * g.CmPad + o0 + v * 4);
        *reinterpret_cast<float4*>(lw + r * MB + v * 4) = val;
      }
    }
    // ---- source patch rows (c, jb, et, eh) x eW, zero-filled outside the tensor
    for (int r = hw; r < rows; r += 8) {
      const int c = fast_div(r, g.mg_rpc);
      int rem = r - c * g.rpc;
      const int jb = fast_div(rem, g.mg_eth);
      rem -= jb * g.eth;
      const int et = fast_div(rem, g.mg_eh);
      const int eh = rem - et * g.eH;
      const int b = j0b + jb, t = src_t0 + et, h = src_h0 + eh, ch = c0 + c;
      const bool rv = (b < g.B) && (ch < g.Ck) && ((unsigned)t < (unsigned)g.sT) && ((unsigned)h < (unsigned)g.sH);
      const int sbase = rv ? (((b * g.Ck + ch) * g.sT + t) * sHW + h * g.sW) : 0;
      float* lrow = lp + c * g.CS + ((jb * g.eT + et) * g.eH + eh) * g.eWp;
      for (int ew = hl; ew < g.eW; ew += 32) {
        const int w = src_w0 + ew;
        float val = 0.f;
        if (rv && (unsigned)w < (unsigned)g.sW) {
          val = g.src[sbase + w];
          if (g.src_y) val = act_grad(val, g.src_y[sbase + w], g.act_pro);
        }
        lrow[ew] = val;
      }
    }
    __syncthreads();
    // ---- MFMA: per tap, per channel pair
    for (int tap = 0; tap < g.ntaps; ++tap) {
      const int toff = g.tap_off[tap];
      const float* wt = lw + tap * CK * MB + lhi * MB + wm * TM * 32 + l31;
#pragma unroll
      for (int cp = 0; cp < CK / 2; ++cp) {
        float a[TM], bv[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = wt[cp * 2 * MB + i * 32];
#pragma unroll
        for (int f = 0; f < TN; ++f) bv[f] = lp[lane_base[f] + toff + cp * 2 * g.CS];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int f = 0; f < TN; ++f)
            acc[i][f] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bv[f], acc[i][f], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: bias, activation, residual, store.  C/D map: col = lane&31 (pixel),
  //      row = (r&3) + 8*(r>>2) + 4*(lane>>5) (dest channel)
  const int dHW = g.dH * g.dW;
#pragma unroll
  for (int f = 0; f < TN; ++f) {
    const int pix = (wn * TN + f) * 32 + l31;
    const int gw = j0w + (pix & JWm);
    const int gh = j0h + ((pix >> g.ljw) & JHm);
    const int gt = j0t + ((pix >> (g.ljw + g.ljh)) & JTm);
    const int gb = j0b + (pix >> (g.ljw + g.ljh + g.ljt));
    const bool pv = gb < g.B && gt < g.nT && gh < g.nH && gw < g.nW;
    const int sp = (gt * g.oT + g.pT) * dHW + (gh * g.oH + g.pH) * g.dW + gw * g.oW + g.pW;
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

typedef void (*patch_fn)(const PatchGeom);
template <int MB, int NPIX, int WM, int CK>
static int launch_patch(const PatchGeom& g, dim3 grid, size_t lds, hipStream_t s) {
  auto k = patch_gemm_kernel<MB, NPIX, WM, CK>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  P2I_LAUNCH(k, grid, dim3(256), lds, s, g);
  return launch_status();
}

static int dispatch_patch_dma(const TileCfg& c, int KG, const PatchGeom& g, dim3 grid, size_t lds, hipStream_t s) {
  int rc = dispatch_patch_dma_g0(c, KG, g, grid, lds, s);
  if (rc == -1) rc = dispatch_patch_dma_g1(c, KG, g, grid, lds, s);
  if (rc == -1) rc = dispatch_patch_dma_g2(c, KG, g, grid, lds, s);
  if (rc == -1) rc = dispatch_patch_dma_g3(c, KG, g, grid, lds, s);
  if (rc != -1) return rc;
  set_error("no DMA kernel instance for tile cfg %d %d %d %d kg %d", c.MB, c.NPIX, c.WM, c.CK, KG);
  return P2I_EINVAL;
}

static int dispatch_patch(const TileCfg& c, const PatchGeom& g, dim3 grid, size_t lds, hipStream_t s) {
#define P2I_CASE(mb_, npix_, wm_, ck_) \
  if (c.MB == mb_ && c.NPIX == npix_ && c.WM == wm_ && c.CK == ck_) return launch_patch<mb_, npix_, wm_, ck_>(g, grid, lds, s);
  P2I_CASE(128, 256, 2, 8)
  P2I_CASE(64, 256, 1, 8)
  P2I_CASE(64, 128, 2, 8)
  P2I_CASE(32, 128, 1, 8)
  P2I_CASE(128, 256, 2, 4)
  P2I_CASE(64, 256, 1, 4)
  P2I_CASE(64, 128, 2, 4)
  P2I_CASE(32, 128, 1, 4)
  P2I_CASE(128, 256, 2, 2)
  P2I_CASE(64, 256, 1, 2)
  P2I_CASE(64, 128, 2, 2)
  P2I_CASE(32, 128, 1, 2)
#undef P2I_CASE
  set_error("no kernel instance for tile cfg %d %d %d %d", c.MB, c.NPIX, c.WM, c.CK);
  return P2I_EINVAL;
}


static thread_local int g_last_plan[6] = {0, 0, 0, 0, 0, 0};

static int run_patch_gemm(PatchGeom g, const ClassSpec& cs, hipStream_t s);

// One launch for `ncls` classes sharing source/dest tensors and multipliers (strided dgrad: blockIdx.z = class).
// Returns 1 if the merged DMA launch is not possible (caller falls back to one launch per class).
static int run_patch_gemm_classes(PatchGeom g, const ClassSpec* css, int ncls, hipStream_t s) {
  if (g.src_y != nullptr || ncls > MAX_CLASSES) return 1;
  if (const X6Ctx& xc = x6_ctx(); xc.wb != nullptr) {        // inside a p2i_conv_*_x6 call: bf16-split kernel first, one launch per class
    bool all = true;                                        // (all classes or none)
    for (int q = 0; q < ncls && all; ++q) all = run_patch_gemm_x6c(g, css[q], xc.wb, xc.ntaps_w, nullptr, s, true) == 0;
    if (all) {
      for (int q = 0; q < ncls; ++q)
        if (const int rc = run_patch_gemm_x6c(g, css[q], xc.wb, xc.ntaps_w, g_last_plan, s)) return rc == 1 ? P2I_EINVAL : rc;
      return P2I_OK;
    }
  }
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
  if (max_taps == 0) max_taps = 1;                       // a class without taps still has to write zeros
  g.ntaps = max_taps;
  static const int cand[5][3] = {{128, 256, 2}, {64, 256, 1}, {128, 128, 2}, {64, 128, 2}, {32, 128, 1}};
  int best = -1, best_ck = 0;
  long long best_score = -1;
  PatchGeom bg = g;
  size_t best_lds = 0;
  dim3 best_grid;
  for (int ci = 0; ci < 5; ++ci) {
    const int MBc = cand[ci][0], NP = cand[ci][1];
    if (MBc > 32 && g.Cm <= MBc / 2) continue;            // more than half of the m-tile would be padding
    int jb, jt, jh, jw;
    pick_tile_dims(NP, g.B, mx[0], mx[1], mx[2], jb, jt, jh, jw);
    PatchGeom t = g;
    t.ljb = ilog2(jb); t.ljt = ilog2(jt); t.ljh = ilog2(jh); t.ljw = ilog2(jw);
    t.eT = (jt - 1) * g.mT + rng[0] + 1;
    t.eH = (jh - 1) * g.mH + rng[1] + 1;
    t.eW = (jw - 1) * g.mW + rng[2] + 1;
    // 16-B patch DMA: rows and tile origins 16-B aligned
    static const int v4_off = getenv("P2I_CONV_V4") ? (atoi(getenv("P2I_CONV_V4")) == 0) : 0;   // read once per process
    bool same_lo_w = true;                                  // merged classes may share the aligned origin only if their w windows start alike
    for (int q = 1; q < ncls; ++q) same_lo_w = same_lo_w && lo[q][2] == lo[0][2];
    t.v4 = (!v4_off && same_lo_w && (g.sW & 3) == 0 && ((jw * g.mW) & 3) == 0) ? 1 : 0;   // any source multiplier: rows are contiguous
    t.v4sh = 0;
    if (t.v4) {
      t.v4sh = ((lo[0][2] % 4) + 4) % 4;                    // columns added on the left so that the row starts 16-B aligned
      t.eW = (t.eW + t.v4sh + 3) & ~3;
      t.eW4 = t.eW >> 2;
    }
    t.eWp = t.eW;
    t.eth = t.eT * t.eH;
    t.rpc = jb * t.eth;
    t.CSl = t.rpc * t.eW;
    t.CS = t.CSl;
    t.G4 = t.CSl >> 2;
    static const int force_ck16 = getenv("P2I_CONV_CK16") ? atoi(getenv("P2I_CONV_CK16")) : 0;
    for (int CKc = ((max_taps <= 4 || ((force_ck16 || g.Ck >= 256) && max_taps == 9 && MBc <= 64)) ? 16 : 8); CKc >= 2; CKc >>= 1) {
      if (g.Ck >= CKc ? (g.Ck % CKc != 0) : (CKc != 2 && g.Ck * 2 <= CKc)) continue;
      if (CKc == 2 && !(MBc == 64 && NP == 128) && !(MBc == 32)) continue;      // instantiated CK=2 tiles
      if ((CKc == 4 || CKc == 16) && MBc == 128 && NP == 128) continue;
      const int PT = CKc * t.CSl;
      if (PT >= 65536 || t.CSl >= 65536) continue;
      const int PTp = t.v4 ? (((PT >> 2) + 63) & ~63) * 4 : (PT + 63) & ~63;
      const int nwrows = max_taps * CKc;
      const size_t WSZ = (size_t)((nwrows * (MBc / 4) + 63) & ~63) * 4;
      const size_t lds = sizeof(float) * (((nwrows + 63) & ~63) + (size_t)PTp + 2 * (WSZ + PTp));
      if (lds > 160 * 1024) continue;
      const long long nb = (long long)ceil_div(g.Cm, MBc) * ((total_pix + NP - 1) / NP);
      // score: filling the chip first, then MFMA work per staged byte (tile area), then deeper chunks
      long long score = (nb >= 256 ? 1000000000ll : nb * 1000000ll) + (long long)MBc * NP * 10 + CKc + (lds <= 80 * 1024 ? 5 : 0);
      if (score > best_score) {
        best_score = score; best = ci; best_ck = CKc; best_lds = lds;
        bg = t; bg.PT = PT;
        bg.ntt = ceil_div(mx[0], jt); bg.nth = ceil_div(mx[1], jh); bg.ntw = ceil_div(mx[2], jw);
        best_grid = dim3((unsigned)(ceil_div(g.B, jb) * bg.ntt * bg.nth * bg.ntw), (unsigned)ceil_div(g.Cm, MBc), (unsigned)ncls);
      }
      break;   // largest feasible CK for this tile
    }
  }
  const unsigned long long sbytes = 4ull * g.B * g.Ck * g.sT * g.sH * g.sW;
  if (best < 0 || sbytes >= 0xF0000000ull) return 1;
  TileCfg c{cand[best][0], cand[best][1], cand[best][2], best_ck};
  bg.mg_csl = magic_u16(bg.CSl); bg.mg_ew = magic_u16(bg.eW);
  bg.mg_rpc = magic_u16(bg.rpc); bg.mg_eth = magic_u16(bg.eth); bg.mg_eh = magic_u16(bg.eH);
  if (bg.v4) { bg.mg_g4 = magic_u16(bg.G4); bg.mg_ew4 = magic_u16(bg.eW4); }
  bg.src_bytes = (unsigned)sbytes;
  bg.wp_bytes = g.wp_bytes;
  bg.nclass = ncls;
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
  // single class: the top-level fields are the class
  bg.nT = css[0].nT; bg.nH = css[0].nH; bg.nW = css[0].nW;
  bg.pT = css[0].pT; bg.pH = css[0].pH; bg.pW = css[0].pW;
  bg.bT = lo[0][0]; bg.bH = lo[0][1]; bg.bW = lo[0][2];
  for (int i = 0; i < css[0].ntaps; ++i) { bg.tap_w[i] = bg.cls[0].tap_w[i]; bg.tap_off[i] = bg.cls[0].tap_off[i]; }
  if (bg.v4) {                                               // aligned origin: v4sh columns further left
    bg.bW -= bg.v4sh;
    for (int i = 0; i < css[0].ntaps; ++i) bg.tap_off[i] += bg.v4sh;
    for (int q = 0; q < ncls; ++q) {
      bg.cls[q].bW -= bg.v4sh;
      for (int i = 0; i < css[q].ntaps; ++i) bg.cls[q].tap_off[i] += bg.v4sh;
    }
  }
  if (ncls == 1) bg.ntaps = css[0].ntaps;
  // ~1 workgroup per CU or fewer: 8-wave workgroups with intra-block split-K (small accumulator tiles only)
  const long long nb = (long long)best_grid.x * best_grid.y * best_grid.z;
  const int acc_regs = (c.MB / 32) * (c.NPIX / 32) / 4 * 16;
  const size_t red_bytes = (size_t)4 * acc_regs * 64 * 4;
  // 27-tap (3-D) layers stage 27 * CK weight rows per chunk, so CK stays 4 and one workgroup fills a CU's LDS: give it 8 waves
  // too (two per SIMD: one wave's DMA issue / LDS waits hide under the other's MFMAs), one channel pair per wave group
  static const int kg27 = getenv("P2I_CONV_KG27") ? atoi(getenv("P2I_CONV_KG27")) : 1;
  const bool kg_27 = kg27 && max_taps == 27 && c.CK == 4 && ncls == 1;
  const int KG = ((nb <= 320 || (c.CK == 16 && max_taps == 9) || kg_27) && (c.CK >= 8 || kg_27) && c.MB <= 64 && red_bytes <= best_lds) ? 2 : 1;
  g_last_plan[0] = c.MB; g_last_plan[1] = c.NPIX; g_last_plan[2] = c.WM; g_last_plan[3] = c.CK;
  g_last_plan[4] = ncls > 1 ? 0 : ((max_taps == 9 && (c.CK <= 8 || KG == 2)) ? 9 : (max_taps == 1 ? 1 : ((max_taps == 27 && (c.CK == 4 || (c.CK < 4 && KG == 1))) ? 27 : 0)));
  g_last_plan[5] = KG;
  return dispatch_patch_dma(c, KG, bg, best_grid, best_lds, s);
}

static int run_patch_gemm(PatchGeom g, const ClassSpec& cs, hipStream_t s) {
  if (cs.nT <= 0 || cs.nH <= 0 || cs.nW <= 0) return P2I_OK;
  // ---- DMA-pipelined path (no act'(y) prologue)
  {
    const int rc = run_patch_gemm_classes(g, &cs, 1, s);
    if (rc != 1) return rc;
  }
  g.nT = cs.nT; g.nH = cs.nH; g.nW = cs.nW;
  g.mT = cs.mT; g.mH = cs.mH; g.mW = cs.mW;
  g.oT = cs.oT; g.oH = cs.oH; g.oW = cs.oW; g.pT = cs.pT; g.pH = cs.pH; g.pW = cs.pW;
  g.ntaps = cs.ntaps;
  g.nclass = 1;
  int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  for (int i = 0; i < cs.ntaps; ++i) {
    const int d[3] = {cs.dt[i], cs.dh[i], cs.dw[i]};
    for (int k = 0; k < 3; ++k) {
      if (i == 0 || d[k] < lo[k]) lo[k] = d[k];
      if (i == 0 || d[k] > hi[k]) hi[k] = d[k];
    }
  }
  g.bT = lo[0]; g.bH = lo[1]; g.bW = lo[2];

  // ---- tile configuration
  TileCfg c;
  c.CK = (g.Ck == 1) ? 2 : (cs.ntaps > 9 ? 4 : 8);
  if (g.Ck < c.CK && g.Ck > 1) c.CK = (g.Ck >= 4) ? 4 : 2;
  c.MB = g.Cm > 64 ? 128 : (g.Cm > 32 ? 64 : 32);
  c.NPIX = 256;
  const long long total_pix = (long long)g.B * cs.nT * cs.nH * cs.nW;
  auto nblocks = [&](const TileCfg& t) { return (long long)ceil_div(g.Cm, t.MB) * ((total_pix + t.NPIX - 1) / t.NPIX); };
  const long long want = 512;
  while (nblocks(c) < want) {
    if (c.MB == 128) { c.MB = 64; continue; }          // more m-blocks first (keeps pixel reuse of weights)
    if (c.NPIX == 256) { c.NPIX = 128; continue; }
    if (c.MB == 64 && g.Cm > 32) { c.MB = 32; continue; }
    break;
  }
  if (c.MB == 128) { c.NPIX = 256; c.WM = 2; }
  else if (c.MB == 64) { c.WM = (c.NPIX == 256) ? 1 : 2; }
  else { c.NPIX = 128; c.WM = 1; }

  for (int attempt = 0; attempt < 4; ++attempt) {
    int jb, jt, jh, jw;
    pick_tile_dims(c.NPIX, g.B, cs.nT, cs.nH, cs.nW, jb, jt, jh, jw);
    g.ljb = ilog2(jb); g.ljt = ilog2(jt); g.ljh = ilog2(jh); g.ljw = ilog2(jw);
    g.eT = (jt - 1) * cs.mT + (hi[0] - lo[0]) + 1;
    g.eH = (jh - 1) * cs.mH + (hi[1] - lo[1]) + 1;
    g.eW = (jw - 1) * cs.mW + (hi[2] - lo[2]) + 1;
    g.eWp = g.eW | 1;
    g.eth = g.eT * g.eH;
    g.rpc = jb * g.eth;
    g.CS = g.rpc * g.eWp;
    g.mg_rpc = magic_u16(g.rpc); g.mg_eth = magic_u16(g.eth); g.mg_eh = magic_u16(g.eH);
    const size_t lds = sizeof(float) * ((size_t)cs.ntaps * c.CK * c.MB + (size_t)c.CK * g.CS);
    if (lds > 160 * 1024 || c.CK * g.rpc >= 65536) {
      if (c.CK > 2) { c.CK /= 2; continue; }
      if (c.NPIX == 256) { c.NPIX = 128; c.WM = (c.MB == 64) ? 2 : (c.MB == 128 ? 2 : 1); if (c.MB == 128) c.MB = 64; continue; }
      set_error("conv tile does not fit LDS (%zu bytes)", lds);
      return P2I_ELDS;
    }
    for (int i = 0; i < cs.ntaps; ++i) {
      g.tap_w[i] = cs.tw[i];
      g.tap_off[i] = ((cs.dt[i] - lo[0]) * g.eH + (cs.dh[i] - lo[1])) * g.eWp + (cs.dw[i] - lo[2]);
    }
    g.ntt = ceil_div(cs.nT, jt); g.nth = ceil_div(cs.nH, jh); g.ntw = ceil_div(cs.nW, jw);
    const int ntb = ceil_div(g.B, jb);
    dim3 grid((unsigned)(ntb * g.ntt * g.nth * g.ntw), (unsigned)ceil_div(g.Cm, c.MB));
    g_last_plan[0] = c.MB; g_last_plan[1] = c.NPIX; g_last_plan[2] = c.WM; g_last_plan[3] = c.CK; g_last_plan[4] = -1; g_last_plan[5] = 0;
    return dispatch_patch(c, g, grid, lds, s);
  }
  set_error("conv tile selection failed");
  return P2I_ELDS;
}

}  // namespace p2i

using namespace p2i;

#ifdef P2I_STAMP
namespace p2i { __device__ unsigned long long* p2i_stamp_buf = nullptr; }
extern "C" int p2i_debug_set_stamp(unsigned long long* buf) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(p2i::p2i_stamp_buf), &buf, sizeof(buf));
}
#endif

extern "C" int p2i_conv_last_plan(int* out6) {
  if (!out6) return P2I_EINVAL;
  for (int i = 0; i < 6; ++i) out6[i] = g_last_plan[i];
  return P2I_OK;
}

extern "C" int p2i_conv_fwd(const p2i_conv_desc* d, const float* x, const float* wp, const float* bias,
                            const float* residual, float* y, int act, void* stream) {
  if (int e = check_desc(d)) return e;
  P2I_REQUIRE(x && wp && y, "null pointer");
  if (d->Cin == 1 && residual == nullptr && x6_ctx().wb == nullptr) {         // single-channel input: dedicated kernel (conv_c1.hip)
    const int rc = c1_fwd(d, x, wp, bias, y, act, (hipStream_t)stream);
    if (rc != 1) { g_last_plan[0] = 32; g_last_plan[1] = 64; g_last_plan[2] = 1; g_last_plan[3] = 2; g_last_plan[4] = 27; g_last_plan[5] = 1; return rc; }
  }
  if (d->Cout == 1 && residual == nullptr) {                                  // single-channel output: bandwidth kernel (conv_c1.hip)
    const int rc = o1_fwd(d, x, wp, bias, y, act, (hipStream_t)stream);
    if (rc != 1) { g_last_plan[0] = 1; g_last_plan[1] = 64; g_last_plan[2] = 8; g_last_plan[3] = 1; g_last_plan[4] = 9; g_last_plan[5] = 3; return rc; }
  }
  PatchGeom g{};
  g.src = x; g.src_y = nullptr; g.wp = wp; g.bias = bias; g.res = residual; g.dst = y; g.act_epi = act; g.act_pro = P2I_ACT_NONE;
  g.B = d->B; g.Ck = d->Cin; g.Cm = d->Cout; g.CmPad = (d->Cout + 31) / 32 * 32;
  g.wp_bytes = 4u * (unsigned)(d->kt * d->kh * d->kw) * g.Ck * g.CmPad;
  g.sT = d->Ti; g.sH = d->Hi; g.sW = d->Wi; g.dT = d->To; g.dH = d->Ho; g.dW = d->Wo;
  ClassSpec cs{};
  cs.nT = d->To; cs.nH = d->Ho; cs.nW = d->Wo;
  cs.mT = d->st; cs.mH = d->sh; cs.mW = d->sw;
  cs.oT = cs.oH = cs.oW = 1; cs.pT = cs.pH = cs.pW = 0;
  int n = 0;
  for (int a = 0; a < d->kt; ++a)
    for (int b = 0; b < d->kh; ++b)
      for (int c = 0; c < d->kw; ++c) {
        cs.tw[n] = (short)n; cs.dt[n] = a - d->pt; cs.dh[n] = b - d->ph; cs.dw[n] = c - d->pw; ++n;
      }
  cs.ntaps = n;
  return run_patch_gemm(g, cs, (hipStream_t)stream);
}

extern "C" int p2i_conv_dgrad(const p2i_conv_desc* d, const float* dy, const float* y_act, int act,
                              const float* wp_d, const float* dx_add, const float* mask_y, int mask_act,
                              float* dx, void* stream) {
  if (int e = check_desc(d)) return e;
  P2I_REQUIRE(dy && wp_d && dx, "null pointer");
  PatchGeom g{};
  g.src = dy; g.src_y = y_act; g.wp = wp_d; g.bias = nullptr; g.res = dx_add; g.dst = dx; g.act_epi = P2I_ACT_NONE; g.act_pro = y_act ? act : P2I_ACT_NONE;
  g.mask_y = mask_y; g.mask_act = mask_y ? mask_act : P2I_ACT_NONE;
  P2I_REQUIRE(!(mask_y && y_act), "dgrad: use either the act'(y) prologue or the epilogue mask");
  if (d->Cin == 1 && !y_act && d->kt * d->kh * d->kw * d->Cout * 4 <= 64 * 1024)
    return c1_dgrad(d, dy, wp_d, dx_add, mask_y, mask_act, dx, (hipStream_t)stream);
  g.B = d->B; g.Ck = d->Cout; g.Cm = d->Cin; g.CmPad = (d->Cin + 31) / 32 * 32;
  g.wp_bytes = 4u * (unsigned)(d->kt * d->kh * d->kw) * g.Ck * g.CmPad;
  g.sT = d->To; g.sH = d->Ho; g.sW = d->Wo; g.dT = d->Ti; g.dH = d->Hi; g.dW = d->Wi;
  // input-parity classes modulo the stride: merged into ONE launch (blockIdx.z = class) when possible
  ClassSpec css[MAX_CLASSES];
  int ncls = 0;
  const bool mergeable = d->st * d->sh * d->sw <= MAX_CLASSES;
  for (int ct = 0; ct < d->st; ++ct)
    for (int chh = 0; chh < d->sh; ++chh)
      for (int cw = 0; cw < d->sw; ++cw) {
        ClassSpec cs{};
        cs.nT = (d->Ti - ct + d->st - 1) / d->st;
        cs.nH = (d->Hi - chh + d->sh - 1) / d->sh;
        cs.nW = (d->Wi - cw + d->sw - 1) / d->sw;
        cs.mT = cs.mH = cs.mW = 1;
        cs.oT = d->st; cs.oH = d->sh; cs.oW = d->sw; cs.pT = ct; cs.pH = chh; cs.pW = cw;
        int n = 0;
        for (int a = 0; a < d->kt; ++a) {
          if ((ct + d->pt - a) % d->st) continue;
          for (int b = 0; b < d->kh; ++b) {
            if ((chh + d->ph - b) % d->sh) continue;
            for (int c = 0; c < d->kw; ++c) {
              if ((cw + d->pw - c) % d->sw) continue;
              cs.tw[n] = (short)((a * d->kh + b) * d->kw + c);
              // exact division (divisible by construction)
              cs.dt[n] = (ct + d->pt - a) / d->st; cs.dh[n] = (chh + d->ph - b) / d->sh; cs.dw[n] = (cw + d->pw - c) / d->sw;
              ++n;
            }
          }
        }
        cs.ntaps = n;
        if (mergeable) css[ncls++] = cs;
        else if (int e = run_patch_gemm(g, cs, (hipStream_t)stream)) return e;
      }
  if (mergeable) {
    int rc = 1;
    if (const X6Ctx& xc = x6_ctx(); xc.wb != nullptr && ncls == 4)      // bf16-split kernel with the four parity classes fused
      rc = run_patch_gemm_x6c_fused(g, css, ncls, xc.wb, xc.ntaps_w, g_last_plan, (hipStream_t)stream);
    if (rc == 1) rc = (ncls > 1 && x6_ctx().wb == nullptr) ? run_patch_gemm_fused(g, css, ncls, g_last_plan, (hipStream_t)stream) : 1;
    if (rc == 1) rc = ncls > 1 ? run_patch_gemm_classes(g, css, ncls, (hipStream_t)stream) : 1;
    if (rc == 1) {
      for (int q = 0; q < ncls; ++q)
        if (int e = run_patch_gemm(g, css[q], (hipStream_t)stream)) return e;
    } else if (rc != P2I_OK) return rc;
  }
  return P2I_OK;
}


// ---- exact-fp32 convolution on the bf16 matrix pipe (conv_x6c.hip).  `wsplit` is caller-owned scratch of
// 3 * ntaps * K * pad32(M) bf16 (K = contraction channels, M = destination channels of the call) that receives the
// 3-plane split of `wp`; layers the bf16-split kernel does not take (x6c_would_take) run on the fp32-MFMA kernels above.
namespace p2i { int x6_split_weights(const float* wp, uint16_t* wb, int ntaps, int Ck, int CmPad, hipStream_t s); }


extern "C" int p2i_conv_fwd_x6(const p2i_conv_desc* d, const float* x, const float* wp, uint16_t* wsplit, const float* bias,
                               const float* residual, float* y, int act, void* stream) {
  if (int e = check_desc(d)) return e;
  if (wsplit == nullptr || (d->Cin & 15) != 0 || !x6c_would_take(d, false, act))
    return p2i_conv_fwd(d, x, wp, bias, residual, y, act, stream);
  P2I_REQUIRE(wp != nullptr, "null pointer");
  const int nt = d->kt * d->kh * d->kw;
  if (int e = x6_split_weights(wp, wsplit, nt, d->Cin, (d->Cout + 31) / 32 * 32, (hipStream_t)stream)) return e;
  x6_ctx() = X6Ctx{wsplit, nt};
  const int rc = p2i_conv_fwd(d, x, wp, bias, residual, y, act, stream);
  x6_ctx() = X6Ctx{nullptr, 0};
  return rc;
}

extern "C" int p2i_conv_dgrad_x6(const p2i_conv_desc* d, const float* dy, const float* wp_d, uint16_t* wsplit, const float* dx_add,
                                 const float* mask_y, int mask_act, float* dx, void* stream) {
  if (int e = check_desc(d)) return e;
  if (wsplit == nullptr || (d->Cout & 15) != 0 || d->Cin == 1 || !x6c_would_take(d, true, P2I_ACT_NONE))
    return p2i_conv_dgrad(d, dy, nullptr, P2I_ACT_NONE, wp_d, dx_add, mask_y, mask_act, dx, stream);
  P2I_REQUIRE(wp_d != nullptr, "null pointer");
  const int nt = d->kt * d->kh * d->kw;
  if (int e = x6_split_weights(wp_d, wsplit, nt, d->Cout, (d->Cin + 31) / 32 * 32, (hipStream_t)stream)) return e;
  x6_ctx() = X6Ctx{wsplit, nt};
  const int rc = p2i_conv_dgrad(d, dy, nullptr, P2I_ACT_NONE, wp_d, dx_add, mask_y, mask_act, dx, stream);
  x6_ctx() = X6Ctx{nullptr, 0};
  return rc;
}

// ---- pre-split weights: split once per weight update (a whole stack of same-shape packed tensors in one launch), then any number
// of conv calls.  `wb_layer` points at plane 0 of THIS layer inside the stack's split image Wb[plane][ntaps_w][K/8][pad32(M)][8].
extern "C" int p2i_x6_split(const float* wp, uint16_t* wb, int ntaps, int K, int Mpad, void* stream) {
  P2I_REQUIRE(wp && wb && ntaps > 0 && K > 0 && (K & 7) == 0 && Mpad > 0 && (Mpad & 31) == 0, "p2i_x6_split: bad arguments");
  return x6_split_weights(wp, wb, ntaps, K, Mpad, (hipStream_t)stream);
}

extern "C" int p2i_x6c_would_take(const p2i_conv_desc* d, int dgrad, int act) {
  if (check_desc(d)) return 0;
  if (dgrad ? ((d->Cout & 15) != 0 || d->Cin == 1) : (d->Cin & 15) != 0) return 0;
  return x6c_would_take(d, dgrad != 0, dgrad ? P2I_ACT_NONE : act) ? 1 : 0;
}

extern "C" int p2i_conv_fwd_x6s(const p2i_conv_desc* d, const float* x, const float* wp, const uint16_t* wb_layer, int ntaps_w,
                                const float* bias, const float* residual, float* y, int act, void* stream) {
  if (int e = check_desc(d)) return e;
  if (wb_layer == nullptr || (d->Cin & 15) != 0 || !x6c_would_take(d, false, act)) return p2i_conv_fwd(d, x, wp, bias, residual, y, act, stream);
  P2I_REQUIRE(ntaps_w >= d->kt * d->kh * d->kw, "ntaps_w smaller than the layer's tap count");
  x6_ctx() = X6Ctx{wb_layer, ntaps_w};
  const int rc = p2i_conv_fwd(d, x, wp, bias, residual, y, act, stream);
  x6_ctx() = X6Ctx{nullptr, 0};
  return rc;
}

extern "C" int p2i_conv_dgrad_x6s(const p2i_conv_desc* d, const float* dy, const float* wp_d, const uint16_t* wb_layer, int ntaps_w,
                                  const float* dx_add, const float* mask_y, int mask_act, float* dx, void* stream) {
  if (int e = check_desc(d)) return e;
  if (wb_layer == nullptr || (d->Cout & 15) != 0 || d->Cin == 1 || !x6c_would_take(d, true, P2I_ACT_NONE))
    return p2i_conv_dgrad(d, dy, nullptr, P2I_ACT_NONE, wp_d, dx_add, mask_y, mask_act, dx, stream);
  P2I_REQUIRE(ntaps_w >= d->kt * d->kh * d->kw, "ntaps_w smaller than the layer's tap count");
  x6_ctx() = X6Ctx{wb_layer, ntaps_w};
  const int rc = p2i_conv_dgrad(d, dy, nullptr, P2I_ACT_NONE, wp_d, dx_add, mask_y, mask_act, dx, stream);
  x6_ctx() = X6Ctx{nullptr, 0};
  return rc;
}
