// DMA-pipelined patch GEMM kernel of the conv engine (see conv.hip for the formulation) and its launchers.  The instances
// are spread over conv_dma_g*.hip (one group of tile configurations per translation unit) so that they compile in parallel.
#pragma once
#include "conv_common.h"

namespace p2i {

struct TileCfg { int MB, NPIX, WM, CK; };

#ifdef P2I_STAMP
// diagnostic build (tools/build_stamp.sh): per-wave cycle sums of the chunk loop's phases; never defined in the product build
extern __device__ unsigned long long* p2i_stamp_buf;
#endif

// ------------------------------------------------------------------------------------ DMA-pipelined patch GEMM
// Same tiling and MFMA loop as patch_gemm_kernel, but the K-chunks are double-buffered in LDS and
// filled by LDS-DMA (`buffer_load_dword{,x4} ... lds`): no staging VGPRs, no ds_write, and chunk k+1 is
// in flight while chunk k is multiplied (one barrier per chunk).  Border / channel-tail zero fill
// comes from the buffer descriptor's range check (invalid lanes carry an out-of-range voffset).
// The per-element source offsets are computed ONCE per workgroup into an LDS table; the chunk's
// channel base goes in the instruction's scalar offset.  LDS images are lane-linear (DMA writes
// wave-base + lane*size), hence the unpadded patch pitch eW and channel stride CSl.
// KG = 2: 8-wave workgroups; wave group kg multiplies half of each chunk's channel pairs (intra-block
// split-K, partial tiles combined through LDS before the epilogue) so that a launch with only ~1 workgroup
// per CU still has two waves per SIMD to hide DMA issue, LDS latency and barrier skew.
template <int MB, int NPIX, int WAVES_M, int CK, int NT, int KG>
__global__ __launch_bounds__(256 * KG) void patch_gemm_dma_kernel(const PatchGeom g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int WAVES_N = 4 / WAVES_M;
  constexpr int TM = MB / (32 * WAVES_M);
  constexpr int TN = NPIX / (32 * WAVES_N);
  constexpr int V = MB / 4;                          // float4 per weight row
  // class-local geometry (merged strided-dgrad launches pick theirs by blockIdx.z; scalar loads from kernarg)
  const bool multi = g.nclass > 1;
  const ClassGeom& cg = g.cls[multi ? blockIdx.z : 0];
  const int c_nT = multi ? cg.nT : g.nT, c_nH = multi ? cg.nH : g.nH, c_nW = multi ? cg.nW : g.nW;
  const int c_pT = multi ? cg.pT : g.pT, c_pH = multi ? cg.pH : g.pH, c_pW = multi ? cg.pW : g.pW;
  const int c_bT = multi ? cg.bT : g.bT, c_bH = multi ? cg.bH : g.bH, c_bW = multi ? cg.bW : g.bW;
  const int c_ntaps = multi ? cg.ntaps : g.ntaps;
  const short* c_tap_w = multi ? cg.tap_w : g.tap_w;
  const int* c_tap_off = multi ? cg.tap_off : g.tap_off;
  const int nwrows = g.ntaps * CK;
  const int WSZ = ((nwrows * V + 63) & ~63) * 4;     // weight floats per chunk (padded to whole wave-instructions)
  // patch dwords per chunk, padded to whole wave-instructions (v4: 64 lanes x 16 B)
  const int PT4p = ((g.PT >> 2) + 63) & ~63;
  const int PTp = g.v4 ? PT4p * 4 : (g.PT + 63) & ~63;
  int* wtab = reinterpret_cast<int*>(smem);          // [nwrows] (padded to 64)
  const int wtab_sz = (nwrows + 63) & ~63;
  int* ptab = wtab + wtab_sz;                        // [PTp]
  float* buf0 = smem + wtab_sz + PTp;                // [2][WSZ + PTp]
  const int BUFSZ = WSZ + PTp;

  constexpr int NTH = 256 * KG;
#ifdef P2I_STAMP
  unsigned long long st_entry;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_entry) :: "memory");
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, kg = tid >> 8;
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

  // ---- offset tables (bytes; 0xFFFFFFFC = out of range -> DMA writes 0)
  for (int r = tid; r < wtab_sz; r += NTH) {
    int off = -4;
    if (r < nwrows) {
      const int tap = r / CK, c = r - tap * CK;
      if (tap < c_ntaps && c < g.Ck && o0 < g.CmPad) off = ((c_tap_w[tap] * g.Ck + c) * g.CmPad + o0) * 4;
    }
    wtab[r] = off;
  }
  if (g.v4) {
    // 16 B per lane: the image rows start at a 16-B aligned source column (host shifted the origin left and padded the
    // row to a multiple of 4), so a 4-pixel group is inside the tensor or outside it as a whole (sW % 4 == 0)
    for (int e = tid; e < PT4p; e += NTH) {
      int off = -16;
      if (e < (g.PT >> 2)) {
        const int c = fast_div(e, g.mg_g4);
        int rem = e - c * g.G4;
        const int row = g.eW4 == 1 ? rem : fast_div(rem, g.mg_ew4);      // (the magic multiplier does not exist for d = 1)
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
  for (int e = tid; e < PTp; e += NTH) {
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
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int f = 0; f < TN; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][f][r] = 0.f;

  const v4i32 rs_src = make_rsrc(g.src, g.src_bytes);
  const v4i32 rs_w = make_rsrc(g.wp, g.wp_bytes);
  const unsigned smem_la = lds_base(smem);
  const int buf0_off = (int)(buf0 - smem);
  const int chan_bytes = g.sT * sHW * 4;             // source bytes per channel
  const int wbase = tid & ~63;                       // wave-uniform part of this thread's linear index
  const int nwv = nwrows * V;                        // float4 per chunk
  __syncthreads();                                   // tables visible

  // The per-lane source offsets are the same for every chunk (the chunk rides in the scalar offset): when a thread owns
  // only a few DMA lanes, keep them in registers so that the issue right after the barrier is not a chain of
  // ds_read -> wait -> DMA (the MFMAs of the chunk start behind it)
  constexpr int RW = 10, RP = 4;                                 // 10: 128-channel tiles at CK = 8 (9 taps) still qualify
  const int nwq = (((nwv + 63) & ~63) + NTH - 1) / NTH;          // weight DMA instructions of this thread
  const int npq = g.v4 ? (PT4p + NTH - 1) / NTH : RP + 1;        // patch DMA instructions (16-B mode only)
  const bool reg_issue = nwq <= RW && npq <= RP;
  int wv[RW], pv[RP];
  if (reg_issue) {
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const int f = i * NTH + tid;
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
      const int e = i * NTH + tid;
      pv[i] = (i < npq && e < PT4p) ? ptab[e] : -16;
    }
  }
  auto issue = [&](int c0, int bufoff) {                 // bufoff: float offset of the target buffer inside smem
    const int w_soff = c0 * g.CmPad * 4;
    if (reg_issue) {
      const int p_soff = c0 * chan_bytes, pbo = bufoff + WSZ;
#pragma unroll
      for (int i = 0; i < RW; ++i)
        if (i < nwq && i * NTH + wbase < ((nwv + 63) & ~63))
          dma_b128(rs_w, smem_la + 4u * (bufoff + (i * NTH + wbase) * 4), wv[i], w_soff);
#pragma unroll
      for (int i = 0; i < RP; ++i)
        if (i < npq && i * NTH + wbase < PT4p)
          dma_b128(rs_src, smem_la + 4u * (pbo + (i * NTH + wbase) * 4), pv[i], p_soff);
      return;
    }
    for (int f0 = 0; f0 < nwv; f0 += NTH) {          // weights: 16 B per lane
      const int f = f0 + tid;
      int voff = -4;
      if (f < nwv) {
        const int row = f / V, col4 = f - row * V;
        const int base = wtab[row];
        voff = base < 0 ? -4 : base + col4 * 16;
      }
      if (f0 + wbase < ((nwv + 63) & ~63))   // whole waves past the padded weight area must not write (they would zero the patch)
        dma_b128(rs_w, smem_la + 4u * (bufoff + (f0 + wbase) * 4), voff, w_soff);
    }
    const int pbo = bufoff + WSZ;
    const int p_soff = c0 * chan_bytes;
    if (g.v4) {
      for (int e0 = 0; e0 < PT4p; e0 += NTH) {       // patch: 16 B per lane (a quarter of the DMA instructions)
        const int e = e0 + tid;
        const int voff = e < PT4p ? ptab[e] : -16;
        if (e0 + wbase < PT4p)
          dma_b128(rs_src, smem_la + 4u * (pbo + (e0 + wbase) * 4), voff, p_soff);
      }
    } else
    for (int e0 = 0; e0 < PTp; e0 += NTH) {          // patch: 4 B per lane
      const int e = e0 + tid;
      const int voff = e < PTp ? ptab[e] : -4;
      if (e0 + wbase < PTp)
        dma_b32(rs_src, smem_la + 4u * (pbo + e0 + wbase), voff, p_soff);
    }
  };

  int toffs[NT > 0 ? NT : 1];
  if constexpr (NT > 0) {
#pragma unroll
    for (int t = 0; t < NT; ++t) toffs[t] = c_tap_off[t];
  }
  const int nchunks = (g.Ck + CK - 1) / CK;
#ifdef P2I_STAMP
  // diagnostic build only (tools/stamp_conv.py): where a chunk iteration spends its cycles, per wave
  unsigned long long st_wait = 0, st_bar = 0, st_issue = 0, st_mfma = 0, st_t0, st_t1;
#define P2I_STAMP_NOW(v) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
  unsigned long long st_begin; P2I_STAMP_NOW(st_begin);
#endif
  issue(0, buf0_off);
  for (int k = 0; k < nchunks; ++k) {
#ifdef P2I_STAMP
    P2I_STAMP_NOW(st_t0);
#endif
    dma_wait_all();                                  // this wave's DMA share of chunk k has landed ...
#ifdef P2I_STAMP
    P2I_STAMP_NOW(st_t1); st_wait += st_t1 - st_t0;
#endif
    __syncthreads();                                 // ... and everybody else's; buffer (k+1)&1 is free
#ifdef P2I_STAMP
    P2I_STAMP_NOW(st_t0); st_bar += st_t0 - st_t1;
#endif
    float* cur = buf0 + (k & 1) * BUFSZ;
    if (k + 1 < nchunks) issue((k + 1) * CK, buf0_off + ((k + 1) & 1) * BUFSZ);
#ifdef P2I_STAMP
    P2I_STAMP_NOW(st_t1); st_issue += st_t1 - st_t0;
#endif
    const float* lw = cur + lhi * MB + wm * TM * 32 + l31;
    const float* lp = cur + WSZ;
    // software pipeline over taps: tap t+1's operands (TM + TN ds_read_b32 per channel pair) are in flight while
    // tap t's MFMAs run; sched_barrier keeps hipcc from sinking the reads next to their uses
    constexpr int NCP = CK / 2 / KG;                 // channel pairs of this wave group
    float a[NCP][TM], bv[NCP][TN], an[NCP][TM], bn[NCP][TN];
    auto load_tap = [&](int tap, int toff, float (&aa)[NCP][TM], float (&bb)[NCP][TN]) {
#pragma unroll
      for (int cp = 0; cp < NCP; ++cp) {
#pragma unroll
        for (int i = 0; i < TM; ++i) aa[cp][i] = lw[tap * CK * MB + (kg * NCP + cp) * 2 * MB + i * 32];
#pragma unroll
        for (int f = 0; f < TN; ++f) bb[cp][f] = lp[lane_base[f] + toff + (kg * NCP + cp) * 2 * g.CSl];
      }
    };
    auto mfma_tap = [&](const float (&aa)[NCP][TM], const float (&bb)[NCP][TN]) {
#pragma unroll
      for (int cp = 0; cp < NCP; ++cp)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int f = 0; f < TN; ++f)
            acc[i][f] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[cp][i], bb[cp][f], acc[i][f], 0, 0, 0);
    };
    if constexpr (NT > 0) {
      load_tap(0, toffs[0], a, bv);
#pragma unroll
      for (int tap = 0; tap < NT; tap += 2) {
        if (tap + 1 < NT) load_tap(tap + 1, toffs[tap + 1 < NT ? tap + 1 : 0], an, bn);
        __builtin_amdgcn_sched_barrier(0);
        mfma_tap(a, bv);
        __builtin_amdgcn_sched_barrier(0);
        if (tap + 1 < NT) {
          if (tap + 2 < NT) load_tap(tap + 2, toffs[tap + 2 < NT ? tap + 2 : 0], a, bv);
          __builtin_amdgcn_sched_barrier(0);
          mfma_tap(an, bn);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
      const int nt = c_ntaps;
      if (nt > 0) load_tap(0, c_tap_off[0], a, bv);
      for (int tap = 0; tap < nt; tap += 2) {
        if (tap + 1 < nt) load_tap(tap + 1, c_tap_off[tap + 1], an, bn);
        __builtin_amdgcn_sched_barrier(0);
        mfma_tap(a, bv);
        __builtin_amdgcn_sched_barrier(0);
        if (tap + 1 < nt) {
          if (tap + 2 < nt) load_tap(tap + 2, c_tap_off[tap + 2], a, bv);
          __builtin_amdgcn_sched_barrier(0);
          mfma_tap(an, bn);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
#ifdef P2I_STAMP
    P2I_STAMP_NOW(st_t0); st_mfma += st_t0 - st_t1;
#endif
  }

#ifdef P2I_STAMP
  if (p2i_stamp_buf && (threadIdx.x & 63) == 0) {
    unsigned long long st_end; P2I_STAMP_NOW(st_end);
    unsigned long long* o = p2i_stamp_buf + ((size_t)(blockIdx.x + gridDim.x * blockIdx.y) * (NTH / 64) + (threadIdx.x >> 6)) * 8;
    o[0] = st_wait; o[1] = st_bar; o[2] = st_issue; o[3] = st_mfma; o[4] = st_end - st_begin; o[5] = st_begin;
    o[6] = st_begin - st_entry;
  }
  unsigned long long st_loop_end; P2I_STAMP_NOW(st_loop_end);
#endif
  if constexpr (KG == 2) {                           // combine the two wave groups' partial tiles through LDS
    __syncthreads();
    float* red = smem + ((wave * TM * TN * 16) << 6) + lane;
    if (kg == 1) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int f = 0; f < TN; ++f)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[((i * TN + f) * 16 + r) << 6] = acc[i][f][r];
    }
    __syncthreads();
    if (kg == 1) return;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int f = 0; f < TN; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][f][r] += red[((i * TN + f) * 16 + r) << 6];
  }
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
    for (int i = 0; i < TM; ++i)
      epilogue_tile16(acc[i][f], o0 + (wm * TM + i) * 32, lhi, g.Cm, pv, (size_t)gb * g.Cm * g.dT * dHW + sp, (size_t)g.dT * dHW,
                      g.bias, g.act_epi, g.res, g.mask_y, g.mask_act, g.dst);
  }
#ifdef P2I_STAMP
  if (p2i_stamp_buf && (threadIdx.x & 63) == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long st_x; P2I_STAMP_NOW(st_x);
    unsigned long long* o = p2i_stamp_buf + ((size_t)(blockIdx.x + gridDim.x * blockIdx.y) * (NTH / 64) + (threadIdx.x >> 6)) * 8;
    o[7] = st_x - st_loop_end;
  }
#endif
}

template <int MB, int NPIX, int WM, int CK, int NT, int KG>
static inline int launch_patch_dma_nt(const PatchGeom& g, dim3 grid, size_t lds, hipStream_t s) {
  auto k = patch_gemm_dma_kernel<MB, NPIX, WM, CK, NT, KG>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  P2I_LAUNCH(k, grid, dim3(256 * KG), lds, s, g);
  return launch_status();
}
template <int MB, int NPIX, int WM, int CK, int KG>
static inline int launch_patch_dma(const PatchGeom& g, dim3 grid, size_t lds, hipStream_t s) {
  if (g.nclass > 1) return launch_patch_dma_nt<MB, NPIX, WM, CK, 0, KG>(g, grid, lds, s);
  if constexpr (CK <= 8 || KG == 2) {
    if (g.ntaps == 9) return launch_patch_dma_nt<MB, NPIX, WM, CK, 9, KG>(g, grid, lds, s);
  }
  if (g.ntaps == 1) return launch_patch_dma_nt<MB, NPIX, WM, CK, 1, KG>(g, grid, lds, s);
  if constexpr (CK == 4 || (CK < 4 && KG == 1)) {
    if (g.ntaps == 27) return launch_patch_dma_nt<MB, NPIX, WM, CK, 27, KG>(g, grid, lds, s);
  }
  return launch_patch_dma_nt<MB, NPIX, WM, CK, 0, KG>(g, grid, lds, s);
}


// one instance group per translation unit; returns -1 when the group does not hold the configuration
int dispatch_patch_dma_g0(const TileCfg& c, int KG, const PatchGeom& g, dim3 grid, size_t lds, hipStream_t s);
int dispatch_patch_dma_g1(const TileCfg& c, int KG, const PatchGeom& g, dim3 grid, size_t lds, hipStream_t s);
int dispatch_patch_dma_g2(const TileCfg& c, int KG, const PatchGeom& g, dim3 grid, size_t lds, hipStream_t s);
int dispatch_patch_dma_g3(const TileCfg& c, int KG, const PatchGeom& g, dim3 grid, size_t lds, hipStream_t s);

#define P2I_DMA_CASE(mb_, npix_, wm_, ck_, kg_) \
  if (c.MB == mb_ && c.NPIX == npix_ && c.WM == wm_ && c.CK == ck_ && KG == kg_) return launch_patch_dma<mb_, npix_, wm_, ck_, kg_>(g, grid, lds, s);

}  // namespace p2i
