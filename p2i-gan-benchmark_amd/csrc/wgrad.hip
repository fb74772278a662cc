// Convolution weight gradients for gfx950 (see conv.hip for the engine's overall design).
#include "conv_common.h"

namespace p2i {

// ------------------------------------------------------------------------------------ wgrad
struct WgradGeom {
  const float* x;       // (B, Cx, sT, sH, sW)  conv input
  const float* dy;      // (B, Co, nT, nH, nW)  grad of conv output
  const float* y_act;   // saved activation output (may be null)
  float* dwp;           // packed grad [tapsTotal][Cx][CoPad], atomically accumulated
  int act;
  int B, Cx, Co, CoPad;
  int sT, sH, sW;
  int nT, nH, nW;
  int mT, mH, mW;       // conv stride
  int bT, bH, bW;       // -pad
  int eT, eH, eW, eWp;
  int ljb, ljt, ljh, ljw;
  int ntb, ntt, nth, ntw;
  int ntiles, nsplit;
  int tpg;              // taps per group (<= 9); group = blockIdx.z
  int ntaps;
  int CS, PP;           // patch channel stride (odd), dy row pitch (odd)
  int rpc, eth;
  unsigned mg_rpc, mg_eth, mg_eh;
  int tap_off[MAX_TAPS];
  int tap_dt[MAX_TAPS];  // tap's t delta relative to bT (patch staged per group with eT rows)
  // ---- DMA-pipelined variant
  int eWq, XSZ, YSZ;     // odd x-row pitch; x / dy dwords per stage
  int rowblk;            // dwords per patch row of the x image: CB * eWq rounded up to a whole wave-instruction
  int tap_offq[9];       // tap offsets in the [row][c][eWq] image
  unsigned mg_ewq, mg_pp, x_bytes, dy_bytes;
  int dbg;               // diagnostics only (P2I_WGRAD_DBG): 1 = skip MFMA, 2 = skip DMA after the first tile
  float* dbias;          // DMA variant, Y4: != null -> the c-block-0 workgroups also sum their dy tiles per channel (bias gradient)
  int x4, x4sh;          // x image filled by 16-B DMA: rows start x4sh columns left of the tap window (16-B aligned)
  unsigned mg_ewq4;
  float* partial;        // != null: workgroup blockIdx.x STORES its partial tile to partial + blockIdx.x * pstride (same
  long long pstride;     // [tap][c][o] indexing as dwp) instead of atomically adding it; wgrad_reduce_kernel sums the slices
  long long bias_off;    // slice mode with a fused bias gradient: the slice's bias row starts here (behind the tile)
};

// dwp[i] += sum_s partial[s][i] (o < Co; the padded columns stay untouched).  Replaces ns-way float atomics on every
// element (1.3 TB/s chip-wide, ~29 us for the 37.7 MB of a 3x3 C->C layer) by one coalesced write + read of the slices,
// and makes the weight gradient bit-reproducible from run to run.
// Round 4: a slice may carry a bias row behind its [tap][c][o] tile (floats [4 nmain4, 4 n4) of the slice: the per-channel sums of dy
// the kernel's bias-fusing workgroups stored); those elements are added to dbias instead of dwp -- the bias gradient is then summed
// in slice order like the weights (it was ns float atomics per channel).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int ns, long long pstride, int n4, int Co,
                                                           int CoPad, int lsg, float* __restrict__ dwp, int nmain4, float* __restrict__ dbias) {
  // 256 threads = (256 >> lsg) float4 columns x (1 << lsg) slice groups; group q sums slices q, q + SG, ... (4 loads in flight)
  __shared__ float4 red[256];
  const int SG = 1 << lsg, ncol = 256 >> lsg;
  const int col = threadIdx.x & (ncol - 1), q = threadIdx.x >> (8 - lsg);
  for (int i0 = blockIdx.x * ncol; i0 < n4; i0 += gridDim.x * ncol) {
    const int i = i0 + col;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4) {
      const float4* src = reinterpret_cast<const float4*>(partial) + i;
      const long long st4 = pstride / 4;
      int sidx = q;
      for (; sidx + 3 * SG < ns; sidx += 4 * SG) {
        const float4 v0 = src[(long long)sidx * st4], v1 = src[(long long)(sidx + SG) * st4];
        const float4 v2 = src[(long long)(sidx + 2 * SG) * st4], v3 = src[(long long)(sidx + 3 * SG) * st4];
        acc.x += (v0.x + v1.x) + (v2.x + v3.x); acc.y += (v0.y + v1.y) + (v2.y + v3.y);
        acc.z += (v0.z + v1.z) + (v2.z + v3.z); acc.w += (v0.w + v1.w) + (v2.w + v3.w);
      }
      for (; sidx < ns; sidx += SG) {
        const float4 v = src[(long long)sidx * st4];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
    __syncthreads();
    red[threadIdx.x] = acc;
    __syncthreads();
    if (q == 0 && i < n4) {
      for (int k = 1; k < SG; ++k) {
        const float4 v = red[k * ncol + col];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
      const int o = i < nmain4 ? (i * 4) % CoPad : (i - nmain4) * 4;      // CoPad % 4 == 0: a float4 never straddles rows
      if (o < Co) {
        float* d = i < nmain4 ? dwp + (size_t)i * 4 : dbias + (size_t)(i - nmain4) * 4;
        d[0] += acc.x;                                 // the padded columns stay as they are
        if (o + 1 < Co) d[1] += acc.y;
        if (o + 2 < Co) d[2] += acc.z;
        if (o + 3 < Co) d[3] += acc.w;
      }
    }
  }
}

// block: 64 x-channels (M) x 64 dy-channels (N); waves 2x2; each wave one 32x32 tile per tap (<=9)
template <int NPIX>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradGeom g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* lx = smem;                         // [64][CS]
  float* ly = smem + 64 * g.CS;             // [64][PP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int c0 = blockIdx.y * 64;           // x channel block
  const int nco = (g.Co + 63) / 64;
  const int zb = blockIdx.z;
  const int o0 = (zb % nco) * 64;
  const int grp = zb / nco;
  const int tap0 = grp * g.tpg;
  const int ntap = min(g.tpg, g.ntaps - tap0);

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int JW = 1 << g.ljw;
  const int JHm = (1 << g.ljh) - 1, JTm = (1 << g.ljt) - 1;
  const int sHW = g.sH * g.sW, nHW = g.nH * g.nW;
  const int rows = 64 * g.rpc;
  const int hw = tid >> 5, hl = tid & 31;
  const int nrow_pix = NPIX >> g.ljw;       // pixel rows per tile (power of two)
  const int lnrp = __builtin_ctz(nrow_pix);

  for (int tile = blockIdx.x; tile < g.ntiles; tile += g.nsplit) {
    int tl = tile;
    const int tw = tl % g.ntw; tl /= g.ntw;
    const int th = tl % g.nth; tl /= g.nth;
    const int tt = tl % g.ntt;
    const int tb = tl / g.ntt;
    const int j0b = tb << g.ljb, j0t = tt << g.ljt, j0h = th << g.ljh, j0w = tw << g.ljw;
    const int src_t0 = j0t * g.mT + g.bT, src_h0 = j0h * g.mH + g.bH, src_w0 = j0w * g.mW + g.bW;
    __syncthreads();
    // ---- x patch for 64 channels
    for (int r = hw; r < rows; r += 8) {
      const int c = fast_div(r, g.mg_rpc);
      int rem = r - c * g.rpc;
      const int jb = fast_div(rem, g.mg_eth);
      rem -= jb * g.eth;
      const int et = fast_div(rem, g.mg_eh);
      const int eh = rem - et * g.eH;
      const int b = j0b + jb, t = src_t0 + et, h = src_h0 + eh, ch = c0 + c;
      const bool rv = (b < g.B) && (ch < g.Cx) && ((unsigned)t < (unsigned)g.sT) && ((unsigned)h < (unsigned)g.sH);
      const int sbase = rv ? (((b * g.Cx + ch) * g.sT + t) * sHW + h * g.sW) : 0;
      float* lrow = lx + c * g.CS + ((jb * g.eT + et) * g.eH + eh) * g.eWp;
      for (int ew = hl; ew < g.eW; ew += 32) {
        const int w = src_w0 + ew;
        lrow[ew] = (rv && (unsigned)w < (unsigned)g.sW) ? g.x[sbase + w] : 0.f;
      }
    }
    // ---- dy tile [64 o][NPIX] (* act'(y)), zero outside
    for (int r = hw; r < 64 * nrow_pix; r += 8) {
      const int o = r >> lnrp, pr = r & (nrow_pix - 1);     // pr = pixel row inside tile
      const int jh = pr & JHm, jt = (pr >> g.ljh) & JTm, jb = pr >> (g.ljh + g.ljt);
      const int b = j0b + jb, t = j0t + jt, h = j0h + jh, oc = o0 + o;
      const bool rv = b < g.B && oc < g.Co && t < g.nT && h < g.nH;
      const int dbase = rv ? (((b * g.Co + oc) * g.nT + t) * nHW + h * g.nW) : 0;
      float* lrow = ly + o * g.PP + pr * JW;
      for (int jw = hl; jw < JW; jw += 32) {
        const int w = j0w + jw;
        float v = 0.f;
        if (rv && w < g.nW) {
          v = g.dy[dbase + w];
          if (g.y_act) v = act_grad(v, g.y_act[dbase + w], g.act);
        }
        lrow[jw] = v;
      }
    }
    __syncthreads();
    // ---- MFMA over pixels (K dim): A = x[c][pix + tap], B = dy[o][pix]
    const float* xa = lx + (wm * 32 + l31) * g.CS;
    const float* yb = ly + (wn * 32 + l31) * g.PP;
    for (int pr = 0; pr < nrow_pix; ++pr) {
      const int jh = pr & JHm, jt = (pr >> g.ljh) & JTm, jb = pr >> (g.ljh + g.ljt);
      const int xrow = ((jb * g.eT + jt * g.mT) * g.eH + jh * g.mH) * g.eWp;
      for (int s = 0; s < JW; s += 2) {
        const float bv = yb[pr * JW + s + lhi];
        const float* xp = xa + xrow + (s + lhi) * g.mW;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          if (t < ntap) {
            const float av = xp[g.tap_off[tap0 + t]];
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
          }
        }
      }
    }
  }
  // ---- epilogue: D[m = x channel][n = dy channel]; lanes -> n (contiguous in dwp)
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    if (t < ntap) {
      const int o = o0 + wn * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = c0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (c < g.Cx && o < g.Co) {
          const size_t e = ((size_t)((tap0 + t) * g.Cx + c)) * g.CoPad + o;
          if (g.partial != nullptr) g.partial[(long long)blockIdx.x * g.pstride + e] = acc[t][r];     // this pixel-split's slice (wgrad_reduce_kernel sums)
          else atomicAdd(g.dwp + e, acc[t][r]);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------ DMA-pipelined wgrad
// dWp[tap][c][o] += sum_pixels x[c][pix + tap] * dy[o][pix].  MFMA: A = x (m = c, lanes), B = dy (n = o,
// lanes), K = pixels (2 per v_mfma_f32_32x32x2).  Lanes index channels, so both LDS images give
// consecutive channels an ODD stride (x: [patch row][c][eWq], dy: [o][NPIX+1]) => conflict-free
// ds_read_b32, while staying lane-linear for LDS-DMA (each dword's source is a per-lane gather; border
// zeros come from the buffer range check).  Pixel tiles are double-buffered: tile k+1 streams in
// while tile k is multiplied.  Block = 64 c x 64 o, waves 2x2, <= 9 tap accumulators per wave.
// 512 threads: waves = 2 (c tiles) x 2 (o tiles) x 2 (halves of the tile's pixels: intra-block split-K), i.e.
// two waves per SIMD so that one wave's DMA issue / LDS waits hide under the other's MFMAs.
// Y4: dy rows are 16-B aligned (nW % 4 == 0): dy image pitch NPIX+4 filled by 16-B DMA and read with
// ds_read_b128 (conflict-free: 16-lane groups see 16 distinct residues of 4*o mod 64), one read per 2 k-steps.
// LJU >= 0 (3x3, stride 1, pad 1, Y4 + 16-B x image only): "window" inner loop.  A unit = 4 consecutive pixels of a tile row
// (LJU = log2 of the units per tile row).  The MFMA's two k slots are the pixel pairs (p, p+2) and (p+1, p+3): lanes 0-31 read the
// 8 bytes at pixel p, lanes 32-63 the 8 bytes at p + 2 (both 8-B aligned: the dx = 0 column of a unit start sits on a 16-B
// boundary of the image), so that register k of a unit's 4-float window is the operand (lo: x[p+k], hi: x[p+k+2]) as it stands.
// The three dx taps of a kernel row then take registers (k-1, k), i.e. the SAME loaded window, with k = -1 handed over from the
// previous unit: 6 ds_read_b64 (2-way conflicts) + 1 for dy per 18 MFMAs instead of 18 ds_read_b32 (4-way) + 1 ds_read_b128, every
// address a per-tile base register + an immediate, no address arithmetic in the loop.
// SX = 2 (stride 2 in h and w, pad 1): the x column of pixel p, tap c sits at image index 2p + 3 + c, so an 8-B read at 2p + 2 holds
// (-, c0) and the next one (c1, c2); the k slots are again the pixel pairs (p, p+2), (p+1, p+3): lanes 32-63 read 8 columns to
// the right.  12 ds_read_b64 + 1 for dy per 18 MFMAs, no operand is shared between units.
template <int NPIX, int NTAP, bool Y4, int CB, int LJU = -1, int SX = 1>
__global__ __launch_bounds__(512, 2) void wgrad_dma_kernel(const WgradGeom g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int PP = Y4 ? NPIX + 4 : NPIX + 1;
  constexpr int YSZp = ((64 * PP + 255) / 256) * 256;
  const int XSZp = g.XSZ;                                   // rows * CB * eWq
  const int BUFSZ = ((XSZp + 3) & ~3) + YSZp;
  const int YOFF = (XSZp + 3) & ~3;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int CT = CB / 32;                            // c tiles per block (1 or 2)
  constexpr int KS = 4 / CT;                             // pixel-range splits (intra-block split-K)
  const int wn = wave & 1, wm = (wave >> 1) & (CT - 1), kh = wave >> (CT == 2 ? 2 : 1);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int c0 = blockIdx.y * CB, o0 = blockIdx.z * 64;
  const int JW = 1 << g.ljw;
  const int JWm = JW - 1, JHm = (1 << g.ljh) - 1;
  const int wbase = tid & ~63;
  const int sHW = g.sH * g.sW, nHW = g.nH * g.nW;
  const int rowblk = g.rowblk;                              // dwords per patch row (CB channels x eWq, padded to 64)
  const int nprow = g.XSZ / rowblk;

  f32x16 acc[NTAP];
#pragma unroll
  for (int t = 0; t < NTAP; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const unsigned smem_la = lds_base(smem);
  const v4i32 rs_x = make_rsrc(g.x, g.x_bytes);
  const v4i32 rs_y = make_rsrc(g.dy, g.dy_bytes);
  // every wave issues its share of the next tile's DMA right after the barrier (measured: faster than a dedicated
  // issuer wave group, and faster than spreading the issue over the MFMA loop, whose kernarg/SGPR reloads
  // drain lgkmcnt and with it the pipelined LDS operand reads)
  auto issue = [&](int tile, int bufoff) {
    int tl = tile;
    const int tw = tl % g.ntw; tl /= g.ntw;
    const int th = tl % g.nth; tl /= g.nth;
    const int tt = tl % g.ntt;
    const int tb = tl / g.ntt;
    const int j0b = tb << g.ljb, j0h = th << g.ljh, j0w = tw << g.ljw;
    const int st = tt * g.mT + g.bT, sh0 = j0h * g.mH + g.bH, sw0 = j0w * g.mW + g.bW;
    const bool tvalid = (unsigned)st < (unsigned)g.sT;
    for (int prow = 0; prow < nprow; ++prow) {          // x image [prow][c][eWq]; prow-level math is scalar
      const int jb = prow / g.eH, eh = prow - jb * g.eH;
      const int b = j0b + jb, h = sh0 + eh;
      const bool rok = tvalid && b < g.B && (unsigned)h < (unsigned)g.sH;
      const int rbase = ((b * g.Cx + c0) * g.sT + st) * sHW + h * g.sW + sw0;
      if (g.x4) {
        // 16 B per lane: rows start at a 16-B aligned source column (sw0 % 4 == 0, sW % 4 == 0), pitch eWq = 4 * odd
        // (2-way bank conflict on the per-channel A reads, a quarter of the DMA instructions)
        const int rb4 = rowblk >> 2;
        for (int e0 = 0; e0 < rb4; e0 += 512) {
          const int e = e0 + tid;
          const int cc = fast_div(e, g.mg_ewq4);
          const int g4 = e - cc * (g.eWq >> 2);
          const bool ok = rok && cc < CB && c0 + cc < g.Cx && (unsigned)(sw0 + 4 * g4) < (unsigned)g.sW;
          const int voff = ok ? (rbase + cc * g.sT * sHW + 4 * g4) * 4 : -16;
          if (e0 + wbase < rb4) dma_b128(rs_x, smem_la + 4u * (bufoff + prow * rowblk + (e0 + wbase) * 4), voff, 0);
        }
      } else
      for (int e0 = 0; e0 < rowblk; e0 += 512) {
        const int e = e0 + tid;
        const int cc = fast_div(e, g.mg_ewq);
        const int xx = e - cc * g.eWq;
        const bool ok = rok && cc < CB && c0 + cc < g.Cx && (unsigned)(sw0 + xx) < (unsigned)g.sW;
        const int voff = ok ? (rbase + cc * g.sT * sHW + xx) * 4 : -4;
        if (e0 + wbase < rowblk) dma_b32(rs_x, smem_la + 4u * (bufoff + prow * rowblk + e0 + wbase), voff, 0);
      }
    }
    const int ybo = bufoff + YOFF;
    if constexpr (Y4) {                                   // dy image [o][NPIX + 4], 16 B per lane
      constexpr int NCH = 64 * PP / 4;
      for (int q0 = 0; q0 < NCH; q0 += 512) {
        const int q4 = q0 + tid;
        const int o = q4 / (PP / 4);
        const int pp = (q4 - o * (PP / 4)) * 4;
        const int jw = pp & JWm, r = pp >> g.ljw;
        const int jh = r & JHm, jb = r >> g.ljh;
        const int b = j0b + jb, h = j0h + jh, w = j0w + jw, oc = o0 + o;
        const bool ok = q4 < NCH && pp < NPIX && b < g.B && oc < g.Co && h < g.nH && w < g.nW;
        const int voff = ok ? ((((b * g.Co + oc) * g.nT + tt) * nHW) + h * g.nW + w) * 4 : -4;
        if (q0 + wbase < NCH) dma_b128(rs_y, smem_la + 4u * (ybo + (q0 + wbase) * 4), voff, 0);
      }
    } else {                                              // dy image [o][NPIX + 1], 4 B per lane
      for (int e0 = 0; e0 < 64 * PP; e0 += 512) {
        const int e = e0 + tid;
        const int o = fast_div(e, g.mg_pp);
        const int pp = e - o * PP;
        const int jw = pp & JWm, r = pp >> g.ljw;
        const int jh = r & JHm, jb = r >> g.ljh;
        const int b = j0b + jb, h = j0h + jh, w = j0w + jw, oc = o0 + o;
        const bool ok = e < 64 * PP && pp < NPIX && b < g.B && oc < g.Co && h < g.nH && w < g.nW;
        const int voff = ok ? ((((b * g.Co + oc) * g.nT + tt) * nHW) + h * g.nW + w) * 4 : -4;
        if (e0 + wbase < 64 * PP) dma_b32(rs_y, smem_la + 4u * (ybo + e0 + wbase), voff, 0);
      }
    }
  };

  // window kernels: the per-lane part of every DMA offset is tile-invariant -> computed once, kept in registers; a tile adds a
  // scalar base (soffset, >= 0) and the border tests.  (The generic `issue` above redoes ~30 VALU ops of index arithmetic per DMA
  // instruction, in both waves of every SIMD at once, right after the barrier: the matrix pipe idles behind it.)
  constexpr int NXI = 2, NYI = (64 * PP / 4 + 511) / 512;
  int xl_off[NXI], xl_col[NXI], yl_off[NYI], yl_pos[NYI];       // offsets < 0: lane never valid; yl_pos = jb << 16 | jh << 8 | jw
  if constexpr (LJU >= 0) {
    const int rb4 = rowblk >> 2;
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
      const int e = i * 512 + tid;
      const int cc = fast_div(e, g.mg_ewq4);
      const int g4 = e - cc * (g.eWq >> 2);
      const bool ok = e < rb4 && cc < CB && c0 + cc < g.Cx;
      xl_off[i] = ok ? (cc * g.sT * sHW + 4 * g4) * 4 : -1;
      xl_col[i] = 4 * g4;
    }
#pragma unroll
    for (int i = 0; i < NYI; ++i) {
      const int q4 = i * 512 + tid;
      const int o = q4 / (PP / 4);
      const int pp = (q4 - o * (PP / 4)) * 4;
      const int jw = pp & JWm, r = pp >> g.ljw;
      const int jh = r & JHm, jb = r >> g.ljh;
      const bool ok = q4 < 64 * PP / 4 && pp < NPIX && o0 + o < g.Co;
      yl_off[i] = ok ? (((jb * g.Co + o) * g.nT) * nHW + jh * g.nW + jw) * 4 : -1;
      yl_pos[i] = (jb << 16) | (jh << 8) | jw;
    }
  }
  auto issue_fast = [&](int tile, int bufoff) {
    int tl = tile;
    const int tw = tl % g.ntw; tl /= g.ntw;
    const int th = tl % g.nth; tl /= g.nth;
    const int tt = tl % g.ntt;
    const int tb = tl / g.ntt;
    const int j0b = tb << g.ljb, j0h = th << g.ljh, j0w = tw << g.ljw;
    const int st = tt * g.mT + g.bT, sh0 = j0h * g.mH + g.bH, sw0 = j0w * g.mW + g.bW;
    const bool tvalid = (unsigned)st < (unsigned)g.sT;
    const int rb4 = rowblk >> 2;
    int xv[NXI];
#pragma unroll
    for (int i = 0; i < NXI; ++i)
      xv[i] = (xl_off[i] >= 0 && (unsigned)(sw0 + xl_col[i]) < (unsigned)g.sW) ? xl_off[i] + sw0 * 4 : -16;
    for (int prow = 0; prow < nprow; ++prow) {
      const int jb = prow / g.eH, eh = prow - jb * g.eH;
      const int b = j0b + jb, h = sh0 + eh;
      const bool rok = tvalid && b < g.B && (unsigned)h < (unsigned)g.sH;
      const int soff = rok ? (((b * g.Cx + c0) * g.sT + st) * sHW + h * g.sW) * 4 : 0;
#pragma unroll
      for (int i = 0; i < NXI; ++i)
        if (i * 512 + wbase < rb4)
          dma_b128(rs_x, smem_la + 4u * (bufoff + prow * rowblk + (i * 512 + wbase) * 4), rok ? xv[i] : -16, soff);
    }
    const int ybo = bufoff + YOFF;
    const int ysoff = (((j0b * g.Co + o0) * g.nT + tt) * nHW + j0h * g.nW + j0w) * 4;
#pragma unroll
    for (int i = 0; i < NYI; ++i) {
      const bool ok = yl_off[i] >= 0 && j0b + (yl_pos[i] >> 16) < g.B && j0h + ((yl_pos[i] >> 8) & 255) < g.nH && j0w + (yl_pos[i] & 255) < g.nW;
      if (i * 512 + wbase < 64 * PP / 4) dma_b128(rs_y, smem_la + 4u * (ybo + (i * 512 + wbase) * 4), ok ? yl_off[i] : -16, ysoff);
    }
  };

  int toff[NTAP];
#pragma unroll
  for (int t = 0; t < NTAP; ++t) toff[t] = g.tap_offq[t];
  const int xlane = (wm * 32 + l31) * g.eWq + lhi * g.mW;
  const int ylane = (wn * 32 + l31) * PP + (Y4 ? 0 : lhi);
  constexpr int UPIX = Y4 ? 4 : 2;                        // pixels per loop unit
  constexpr int NU = NPIX / UPIX / KS;                    // units per k-split
  const int lju = g.ljw - (Y4 ? 2 : 1);                   // log2(units per pixel row)
  const int upr_m = (1 << lju) - 1;
  int k = 0;
  const bool fuse_bias = Y4 && g.dbias != nullptr && blockIdx.y == 0;   // bias gradient rides on the staged dy tiles
  float bsum = 0.f;
  if ((int)blockIdx.x < g.ntiles) {
    if constexpr (LJU >= 0) issue_fast(blockIdx.x, 0); else issue(blockIdx.x, 0);
  }
  for (int tile = blockIdx.x; tile < g.ntiles; tile += g.nsplit, ++k) {
    dma_wait_all();                                     // this wave's share of the tile has landed ...
    __syncthreads();                                    // ... and so has everybody else's; the other buffer is free
    float* cur = smem + (k & 1) * BUFSZ;
    if (tile + g.nsplit < g.ntiles && !(g.dbg & 2)) {
      if constexpr (LJU >= 0) issue_fast(tile + g.nsplit, ((k + 1) & 1) * BUFSZ); else issue(tile + g.nsplit, ((k + 1) & 1) * BUFSZ);
    }
    if constexpr (Y4) {
      if (fuse_bias) {                                      // 64 channels x 8 segments of NPIX/8 pixels; padding pixels are 0
        const float* yrow = cur + YOFF + (tid >> 3) * PP + (tid & 7) * (NPIX / 8);
#pragma unroll
        for (int q = 0; q < NPIX / 8; q += 4) {
          const float4 v = *reinterpret_cast<const float4*>(yrow + q);
          bsum += (v.x + v.y) + (v.z + v.w);
        }
      }
    }
    const float* xa = cur + xlane;
    const float* yb = cur + YOFF + ylane;
    auto xoff = [&](int uu) {                            // x-image offset of unit uu of this wave's k-range
      const int u = kh * NU + uu;
      const int pr = u >> lju, s0 = (u & upr_m) * UPIX;
      const int jh = pr & JHm, jb = pr >> g.ljh;
      return (jb * g.eH + jh * g.mH) * rowblk + s0 * g.mW;
    };
    auto yoff = [&](int uu) {
      const int u = kh * NU + uu;
      return (u >> lju) * JW + (u & upr_m) * UPIX;
    };
    const int nu = (g.dbg & 1) ? 0 : NU;
    if constexpr (LJU >= 0) {
      static_assert(Y4 && NTAP == 9, "window loop: 3x3");
      constexpr int UPR0 = 1 << LJU;                             // units per tile row
      constexpr int UPR = UPR0 < NU ? UPR0 : NU;                 // units per row SEGMENT of this wave (a wave may own part of a row)
      constexpr int NR = NU / UPR;                               // row segments of this wave
      static_assert(NU % UPR == 0 && (UPR0 % UPR) == 0, "window loop: whole segments");
      // per-tile bases (floats): x window of (segment rr, kernel row b) at the segment's first unit, this lane's channel
      const float* xb[NR][3];
#pragma unroll
      for (int rr = 0; rr < NR; ++rr) {
        const int u0 = __builtin_amdgcn_readfirstlane(kh * NU + rr * UPR);
        const int pr = u0 >> LJU, cu0 = u0 & (UPR0 - 1);
        const int jh = pr & JHm, jb = pr >> g.ljh;
#pragma unroll
        for (int b = 0; b < 3; ++b)
          xb[rr][b] = cur + (wm * 32 + l31) * g.eWq +
                      ((jb * g.eH + SX * jh + b) * rowblk + (SX == 1 ? 4 * cu0 + 4 + 2 * lhi : 8 * cu0 + 2 + 4 * lhi));
      }
      const float* ybp = cur + YOFF + (wn * 32 + l31) * PP + (kh * NU * 4 + 2 * lhi);
      if constexpr (SX == 1) {
        float2 c01[3], c23[3], n01[3], n23[3], yc, yn;
        float pr3[3], npr3[3];
        auto load_unit = [&](int uu, float2 (&w01)[3], float2 (&w23)[3], float (&wp)[3], float2& yy) {
          const int rr = uu / UPR, cu = uu % UPR;
#pragma unroll
          for (int b = 0; b < 3; ++b) {
            w01[b] = *reinterpret_cast<const float2*>(xb[rr][b] + 4 * cu);
            w23[b] = *reinterpret_cast<const float2*>(xb[rr][b] + 4 * cu + 2);
            if (cu == 0) wp[b] = xb[rr][b][-1];                    // (lo: x[p-1], hi: x[p+1]) at a row start
          }
          yy = *reinterpret_cast<const float2*>(ybp + 4 * uu);
        };
        if (nu > 0) {
          load_unit(0, c01, c23, pr3, yc);
#pragma unroll
          for (int uu = 0; uu < NU; ++uu) {
            if (uu + 1 < NU) load_unit(uu + 1, n01, n23, npr3, yn);
#pragma unroll
            for (int b = 0; b < 3; ++b) {
              acc[b * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(pr3[b], yc.x, acc[b * 3 + 0], 0, 0, 0);
              acc[b * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(c01[b].x, yc.x, acc[b * 3 + 1], 0, 0, 0);
              acc[b * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(c01[b].y, yc.x, acc[b * 3 + 2], 0, 0, 0);
              acc[b * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(c01[b].x, yc.y, acc[b * 3 + 0], 0, 0, 0);
              acc[b * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(c01[b].y, yc.y, acc[b * 3 + 1], 0, 0, 0);
              acc[b * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(c23[b].x, yc.y, acc[b * 3 + 2], 0, 0, 0);
            }
            // pin the interleave: the next unit's LDS reads ride in this unit's MFMA gaps
            if (uu + 1 < NU) {
              constexpr int NL = 7;
              const int extra = ((uu + 1) % UPR == 0) ? 3 : 0;
#pragma unroll
              for (int i = 0; i < NL; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              }
              if (extra) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
              } else {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
              }
            }
            if (uu + 1 < NU) {
#pragma unroll
              for (int b = 0; b < 3; ++b) {
                pr3[b] = ((uu + 1) % UPR == 0) ? npr3[b] : c23[b].y;
                c01[b] = n01[b]; c23[b] = n23[b];
              }
              yc = yn;
            }
          }
        }
      } else {
        // stride 2: group gq of a unit = pixels (4cu + gq) on lanes 0-31 and (4cu + gq + 2) on lanes 32-63
        float2 xa0[2][3], xa1[2][3], na0[2][3], na1[2][3], yc, yn;
        auto load_unit2 = [&](int uu, float2 (&w0)[2][3], float2 (&w1)[2][3], float2& yy) {
          const int rr = uu / UPR, cu = uu % UPR;
#pragma unroll
          for (int gq = 0; gq < 2; ++gq)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
              w0[gq][b] = *reinterpret_cast<const float2*>(xb[rr][b] + 8 * cu + 2 * gq);         // (-, c0)
              w1[gq][b] = *reinterpret_cast<const float2*>(xb[rr][b] + 8 * cu + 2 * gq + 2);     // (c1, c2)
            }
          yy = *reinterpret_cast<const float2*>(ybp + 4 * uu);
        };
        if (nu > 0) {
          load_unit2(0, xa0, xa1, yc);
#pragma unroll
          for (int uu = 0; uu < NU; ++uu) {
            if (uu + 1 < NU) load_unit2(uu + 1, na0, na1, yn);
#pragma unroll
            for (int gq = 0; gq < 2; ++gq) {
              const float yv = gq ? yc.y : yc.x;
#pragma unroll
              for (int b = 0; b < 3; ++b) {
                acc[b * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa0[gq][b].y, yv, acc[b * 3 + 0], 0, 0, 0);
                acc[b * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa1[gq][b].x, yv, acc[b * 3 + 1], 0, 0, 0);
                acc[b * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa1[gq][b].y, yv, acc[b * 3 + 2], 0, 0, 0);
              }
            }
            if (uu + 1 < NU) {
#pragma unroll
              for (int i = 0; i < 13; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              }
              __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
#pragma unroll
              for (int gq = 0; gq < 2; ++gq)
#pragma unroll
                for (int b = 0; b < 3; ++b) { xa0[gq][b] = na0[gq][b]; xa1[gq][b] = na1[gq][b]; }
              yc = yn;
            }
          }
        }
      }
    } else if constexpr (Y4) {
      // software pipeline: while the MFMAs of one half-unit run, the next half-unit's A operands are in flight
      float a0[NTAP], a1[NTAP];
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f), vn = v;
      if (nu > 0) {
        const float* xp = xa + xoff(0);
#pragma unroll
        for (int t = 0; t < NTAP; ++t) a0[t] = xp[toff[t]];
        v = *reinterpret_cast<const float4*>(yb + yoff(0));
      }
      for (int uu = 0; uu < nu; ++uu) {
        const float* xp = xa + xoff(uu);
#pragma unroll
        for (int t = 0; t < NTAP; ++t) a1[t] = xp[toff[t] + 2 * g.mW];
        __builtin_amdgcn_sched_barrier(0);
        const float b0 = lhi ? v.y : v.x, b1 = lhi ? v.w : v.z;
#pragma unroll
        for (int t = 0; t < NTAP; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b0, acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (uu + 1 < nu) {
          const float* xn = xa + xoff(uu + 1);
#pragma unroll
          for (int t = 0; t < NTAP; ++t) a0[t] = xn[toff[t]];
          vn = *reinterpret_cast<const float4*>(yb + yoff(uu + 1));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NTAP; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b1, acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        v = vn;
      }
    } else {
      for (int uu = 0; uu < nu; ++uu) {
        const float* xp = xa + xoff(uu);
        const float bv = yb[yoff(uu)];
        float av[NTAP];
#pragma unroll
        for (int t = 0; t < NTAP; ++t) av[t] = xp[toff[t]];
#pragma unroll
        for (int t = 0; t < NTAP; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv, acc[t], 0, 0, 0);
      }
    }
  }
  if (fuse_bias) {
    bsum += __shfl_xor(bsum, 1, 64);
    bsum += __shfl_xor(bsum, 2, 64);
    bsum += __shfl_xor(bsum, 4, 64);
    const int oc = o0 + (tid >> 3);
    if ((tid & 7) == 0 && oc < g.Co) {
      if (g.partial != nullptr) g.partial[(long long)blockIdx.x * g.pstride + g.bias_off + oc] = bsum;     // summed in slice order by wgrad_reduce_kernel
      else atomicAdd(g.dbias + oc, bsum);
    }
  }
  // combine the KS pixel-range partial sums inside the workgroup (through LDS, 4 taps at a time) so that only
  // one wave per (c tile, o tile) issues the global float atomics: they run at ~1.3 TB/s chip-wide and would
  // otherwise cost as much as a third of a small launch
  constexpr int TCH = 4;
  const int wtile = wm * 2 + wn;                           // (c tile, o tile) of this wave: 0 .. 2*CT-1
#pragma unroll
  for (int tc = 0; tc < NTAP; tc += TCH) {
    __syncthreads();
    if (kh > 0) {
      float* dst = smem + (((kh - 1) * 2 * CT + wtile) * TCH * 16 << 6) + lane;
#pragma unroll
      for (int t = 0; t < TCH; ++t)
        if (tc + t < NTAP) {
#pragma unroll
          for (int r = 0; r < 16; ++r) dst[(t * 16 + r) << 6] = acc[tc + t][r];
        }
    }
    __syncthreads();
    if (kh == 0) {
#pragma unroll 1                                           // (unrolled, KS = 4 keeps 3 x 64 partner values live and spills)
      for (int g2 = 0; g2 < KS - 1; ++g2) {
        const float* src = smem + ((g2 * 2 * CT + wtile) * TCH * 16 << 6) + lane;
#pragma unroll
        for (int t = 0; t < TCH; ++t)
          if (tc + t < NTAP) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tc + t][r] += src[(t * 16 + r) << 6];
          }
      }
    }
  }
  if (kh != 0) return;
  if (g.partial != nullptr) {                              // this workgroup's slice: plain coalesced stores
    float* slice = g.partial + (long long)blockIdx.x * g.pstride;
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
      const int o = o0 + wn * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = c0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (c < g.Cx && o < g.Co) slice[((size_t)(t * g.Cx + c)) * g.CoPad + o] = acc[t][r];
      }
    }
    return;
  }
#pragma unroll
  for (int t = 0; t < NTAP; ++t) {
    const int o = o0 + wn * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = c0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      if (c < g.Cx && o < g.Co) atomicAdd(g.dwp + ((size_t)(t * g.Cx + c)) * g.CoPad + o, acc[t][r]);
    }
  }
}

}  // namespace p2i

using namespace p2i;

static thread_local int g_wgrad_plan[4] = {0, 0, 0, 0};   // {kind 0 prologue kernel / 1 dma / 2 single-channel, NTAP, Y4, CB}

extern "C" int p2i_wgrad_last_plan(int* out4) {
  if (!out4) return P2I_EINVAL;
  for (int i = 0; i < 4; ++i) out4[i] = g_wgrad_plan[i];
  return P2I_OK;
}

// dwp[i] += sum over the ns slices of a launch
// slice: floats of one slice's [tap][c][o] tile; dbias != null: each slice holds CoPad more floats (its bias row), pstride = slice + CoPad
static int launch_wgrad_reduce(const float* ws, int ns, long long slice, int Co, int CoPad, float* dwp, hipStream_t s, float* dbias = nullptr,
                               long long pstride = 0) {
  if (pstride == 0) pstride = slice;
#ifdef P2I_STAMP
  { const char* e = getenv("P2I_DEBUG_SKIP_REDUCE"); if (e && atoi(e)) return P2I_OK; }      // diagnostic build only: what the slice reduces cost a step (wrong gradients)
#endif
  const int nmain4 = (int)(slice / 4);
  const int n4 = nmain4 + (dbias ? CoPad / 4 : 0);
  // enough threads to stream the ns * slice floats at HBM rate: split the slices over up to 8 groups while the
  // columns alone give fewer than ~2 blocks per CU
  int lsg = 0;
  while (lsg < 3 && (2 << lsg) <= ns && (n4 >> (8 - lsg)) < 512) ++lsg;
  const int ncol = 256 >> lsg;
  const int blocks = (n4 + ncol - 1) / ncol;
  P2I_LAUNCH(wgrad_reduce_kernel, dim3(blocks > 4096 ? 4096 : blocks), dim3(256), 0, s, ws, ns, pstride, n4, Co, CoPad, lsg, dwp, nmain4, dbias);
  return launch_status();
}

// wgrad_x6.hip: 3x3 stride-1 2-D layers with 64-multiple channel counts on the bf16 matrix pipe; 1 = not its case
namespace p2i {
int run_wgrad_x6(const p2i_conv_desc* d, const float* x, const float* dy, float* dwp, float* dbias, float* ws, long long ws_floats,
                 int* ns_out, long long* slice_out, hipStream_t s);
}

static thread_local float* g_wgrad_ws = nullptr;          // caller-owned slice scratch of the running p2i_conv_wgrad_ws call
static thread_local long long g_wgrad_ws_floats = 0;

extern "C" int p2i_conv_wgrad_ws(const p2i_conv_desc* d, const float* x, const float* dy, const float* y_act, int act, float* dwp,
                                 float* dbias, float* ws, int64_t ws_floats, void* stream) {
  g_wgrad_ws = ws; g_wgrad_ws_floats = ws ? (long long)ws_floats : 0;
  const int rc = p2i_conv_wgrad(d, x, dy, y_act, act, dwp, dbias, stream);
  g_wgrad_ws = nullptr; g_wgrad_ws_floats = 0;
  return rc;
}

extern "C" int p2i_conv_wgrad(const p2i_conv_desc* d, const float* x, const float* dy, const float* y_act,
                              int act, float* dwp, float* dbias, void* stream) {
  if (int e = check_desc(d)) return e;
  P2I_REQUIRE(x && dy && dwp, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (d->Cin == 1 && d->Cout <= 32 && d->kt * d->kh * d->kw <= 32 && !y_act) {
    g_wgrad_plan[0] = 2; g_wgrad_plan[1] = d->kt * d->kh * d->kw; g_wgrad_plan[2] = 0; g_wgrad_plan[3] = 1;
    return c1_wgrad(d, x, dy, dwp, dbias, s);
  }
  if (y_act == nullptr) {                                   // bf16-split kernel where it applies (spatial stride 1, 64-multiple channels)
    int ns6 = 0;
    long long slice6 = 0;
    const int rc = run_wgrad_x6(d, x, dy, dwp, dbias, g_wgrad_ws, g_wgrad_ws_floats, &ns6, &slice6, s);
    if (rc != 1) {
      if (rc) return rc;
      g_wgrad_plan[0] = 3; g_wgrad_plan[1] = 9 * d->kt; g_wgrad_plan[2] = 1; g_wgrad_plan[3] = 64;
      if (ns6 >= 2) return launch_wgrad_reduce(g_wgrad_ws, ns6, slice6, d->Cout, (d->Cout + 31) / 32 * 32, dwp, s, dbias,
                                               slice6 + (dbias ? (d->Cout + 31) / 32 * 32 : 0));
      return P2I_OK;
    }
  }
  g_wgrad_plan[0] = 0; g_wgrad_plan[1] = d->kh * d->kw; g_wgrad_plan[2] = 0; g_wgrad_plan[3] = 64;
  WgradGeom g{};
  g.x = x; g.dy = dy; g.y_act = y_act; g.dwp = dwp; g.act = act;
  g.B = d->B; g.Cx = d->Cin; g.Co = d->Cout; g.CoPad = (d->Cout + 31) / 32 * 32;
  g.sT = d->Ti; g.sH = d->Hi; g.sW = d->Wi; g.nT = d->To; g.nH = d->Ho; g.nW = d->Wo;
  g.mT = d->st; g.mH = d->sh; g.mW = d->sw;
  g.ntaps = d->kt * d->kh * d->kw;
  g.tpg = d->kh * d->kw;                      // one kt slice per group
  P2I_REQUIRE(g.tpg <= 9, "wgrad supports kh*kw <= 9");
  const int ngroups = d->kt;
  constexpr int NPIX = 64;
  // the staged patch covers ONE kt slice: group z uses t offset (a - pt) => separate patch per group
  int jb, jt, jh, jw;
  pick_tile_dims(NPIX, d->B, 1, d->Ho, d->Wo, jb, jt, jh, jw);   // jt = 1: one output frame per tile row group
  // allow several frames/batches in a tile when the frame is small
  g.ljb = ilog2(jb); g.ljt = 0; g.ljh = ilog2(jh); g.ljw = ilog2(jw);
  // tile covers jb "batch*time" slots: fold T into the batch-like dim by treating (b,t) pairs
  // -> keep it simple: jt = 1 and jb spans batches only; tiles iterate over t explicitly.
  g.eT = 1;
  g.eH = (jh - 1) * d->sh + d->kh;
  g.eW = (jw - 1) * d->sw + d->kw;
  g.eWp = g.eW | 1;
  g.eth = g.eT * g.eH;
  g.rpc = jb * g.eth;
  g.CS = (g.rpc * g.eWp) | 1;
  g.PP = NPIX | 1;
  P2I_REQUIRE(64 * g.rpc < 65536, "wgrad patch too large");
  g.mg_rpc = magic_u16(g.rpc); g.mg_eth = magic_u16(g.eth); g.mg_eh = magic_u16(g.eH);
  g.bH = -d->ph; g.bW = -d->pw;
  g.ntb = ceil_div(d->B, jb); g.ntt = d->To; g.nth = ceil_div(d->Ho, jh); g.ntw = ceil_div(d->Wo, jw);
  g.ntiles = g.ntb * g.ntt * g.nth * g.ntw;
  for (int b = 0; b < d->kh; ++b)
    for (int c = 0; c < d->kw; ++c)
      for (int a = 0; a < d->kt; ++a) g.tap_off[(a * d->kh + b) * d->kw + c] = b * g.eWp + c;
  const int ncx = ceil_div(d->Cin, 64), nco = ceil_div(d->Cout, 64);
  const size_t lds = sizeof(float) * (64 * (size_t)g.CS + 64 * (size_t)g.PP);
  P2I_REQUIRE(lds <= 160 * 1024, "wgrad tile does not fit LDS (%zu)", lds);
  int nsplit = 768 / (ncx * nco * ngroups);
  if (nsplit < 1) nsplit = 1;
  if (nsplit > g.ntiles) nsplit = g.ntiles;
  g.nsplit = nsplit;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)wgrad_kernel<NPIX>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  // ---- DMA-pipelined variant (no act'(y) prologue): [row][c][eWq] / [o][PP] images, double buffered.
  // CB = x channels per block: 64, or 32 when the patch is large (strided convs) or Cin is small.
  bool bias_fused = false;
  bool use_dma = (y_act == nullptr) && (g.tpg == 9 || g.tpg == 1) && jw >= 8;
  if (use_dma) {
    const bool y4 = (d->Wo % 4 == 0);
    const int PPh = y4 ? NPIX + 4 : NPIX + 1;
    g.eWq = g.eW | 1;
    {
      static const int x4_off = getenv("P2I_WGRAD_X4") ? (atoi(getenv("P2I_WGRAD_X4")) == 0) : 0;
      g.x4 = (!x4_off && (d->Wi & 3) == 0 && ((jw * d->sw) & 3) == 0) ? 1 : 0;
      g.x4sh = 0;
      if (g.x4) {
        g.x4sh = (((-d->pw) % 4) + 4) % 4;
        g.eWq = (g.eW + g.x4sh + 3) & ~3;
        if (((g.eWq >> 2) & 1) == 0) g.eWq += 4;          // pitch = 4 * odd: 16 distinct banks for 32 consecutive channels
        g.mg_ewq4 = (g.eWq >> 2) == 1 ? 0u : magic_u16(g.eWq >> 2);
      }
    }
    const unsigned long long xb = 4ull * d->B * d->Cin * d->Ti * d->Hi * d->Wi, yb = 4ull * d->B * d->Cout * d->To * d->Ho * d->Wo;
    int CBh = d->Cin > 32 ? 64 : 32;
    size_t lds2 = 0;
    for (;;) {
      g.rowblk = g.x4 ? (CBh * g.eWq + 255) & ~255 : (CBh * g.eWq + 63) & ~63;
      g.XSZ = jb * g.eH * g.rowblk;
      lds2 = sizeof(float) * 2 * ((size_t)((g.XSZ + 3) & ~3) + (size_t)((64 * PPh + 255) / 256) * 256);
      { const size_t red = sizeof(float) * 3 * 4 * 16 * 64 * 2; if (lds2 < red) lds2 = red; }   // k-split combine scratch
      if (lds2 <= 160 * 1024 || CBh == 32) break;
      CBh = 32;
    }
    g.YSZ = 64 * PPh;
    if (lds2 > 160 * 1024 || g.XSZ >= 65536 || xb >= 0xF0000000ull || yb >= 0xF0000000ull) use_dma = false;
    else {
      g.mg_ewq = magic_u16(g.eWq); g.mg_pp = magic_u16(PPh);
      g.x_bytes = (unsigned)xb; g.dy_bytes = (unsigned)yb;
      { static const int dbg_env = getenv("P2I_WGRAD_DBG") ? atoi(getenv("P2I_WGRAD_DBG")) : 0; g.dbg = dbg_env; }
      typedef void (*wk_t)(const WgradGeom);
      wk_t kern;
      // window inner loop (3x3, stride 1 in h and w, pad 1 in w, 16-B x image): P2I_WGRAD_WINDOW=0 keeps the per-tap reads
      static const int window_on = getenv("P2I_WGRAD_WINDOW") ? atoi(getenv("P2I_WGRAD_WINDOW")) : 1;
      const int lju = ilog2(jw) - 2;
      const bool win_shape = window_on && g.tpg == 9 && y4 && g.x4 && d->pw == 1 && d->kw == 3 && d->kh == 3 && lju >= 1 && lju <= 3 && g.rowblk <= 4096;
      const bool window = win_shape && d->sh == 1 && d->sw == 1;                    // stride 1: CB 64 or 32
      const bool window2 = win_shape && d->sh == 2 && d->sw == 2 && CBh == 32;      // stride 2 (its x image only fits with CB = 32)
      if (window && CBh == 64)
        kern = lju == 3 ? (wk_t)wgrad_dma_kernel<NPIX, 9, true, 64, 3> : (lju == 2 ? (wk_t)wgrad_dma_kernel<NPIX, 9, true, 64, 2> : (wk_t)wgrad_dma_kernel<NPIX, 9, true, 64, 1>);
      else if (window)
        kern = lju == 3 ? (wk_t)wgrad_dma_kernel<NPIX, 9, true, 32, 3> : (lju == 2 ? (wk_t)wgrad_dma_kernel<NPIX, 9, true, 32, 2> : (wk_t)wgrad_dma_kernel<NPIX, 9, true, 32, 1>);
      else if (window2)
        kern = lju == 3 ? (wk_t)wgrad_dma_kernel<NPIX, 9, true, 32, 3, 2> : (lju == 2 ? (wk_t)wgrad_dma_kernel<NPIX, 9, true, 32, 2, 2> : (wk_t)wgrad_dma_kernel<NPIX, 9, true, 32, 1, 2>);
      else if (CBh == 64) kern = g.tpg == 9 ? (y4 ? (wk_t)wgrad_dma_kernel<NPIX, 9, true, 64> : (wk_t)wgrad_dma_kernel<NPIX, 9, false, 64>)
                                       : (y4 ? (wk_t)wgrad_dma_kernel<NPIX, 1, true, 64> : (wk_t)wgrad_dma_kernel<NPIX, 1, false, 64>);
      else kern = g.tpg == 9 ? (y4 ? (wk_t)wgrad_dma_kernel<NPIX, 9, true, 32> : (wk_t)wgrad_dma_kernel<NPIX, 9, false, 32>)
                             : (y4 ? (wk_t)wgrad_dma_kernel<NPIX, 1, true, 32> : (wk_t)wgrad_dma_kernel<NPIX, 1, false, 32>);
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      g_wgrad_plan[0] = 1; g_wgrad_plan[1] = g.tpg; g_wgrad_plan[2] = y4 ? 1 : 0; g_wgrad_plan[3] = CBh;
      const int ncb = ceil_div(d->Cin, CBh);
      int ns = 256 / (ncb * nco);        // LDS admits one (8-wave) block per CU
      if (ns < 1) ns = 1;
      if (ns > g.ntiles) ns = g.ntiles;
      for (int a = 0; a < d->kt; ++a) {
        WgradGeom ga = g;
        ga.bT = a - d->pt;
        ga.ntaps = g.tpg;
        ga.nsplit = ns;
        ga.dwp = dwp + (size_t)a * g.tpg * d->Cin * g.CoPad;
        for (int b = 0; b < d->kh; ++b)
          for (int c = 0; c < d->kw; ++c) ga.tap_offq[b * d->kw + c] = b * g.rowblk + c + g.x4sh;
        if (g.x4) ga.bW = g.bW - g.x4sh;
        ga.dbias = (dbias != nullptr && y4 && a == 0) ? dbias : nullptr;   // one kt slice sums dy (every slice sees all of it)
        if (ga.dbias) bias_fused = true;
        // slice mode: ns >= 2 partial tiles per output element and a scratch that holds all ns slices of this launch
        const long long slice = (long long)g.tpg * d->Cin * g.CoPad;
        const long long pstr = slice + (ga.dbias ? g.CoPad : 0);          // (+ the slice's bias row)
        const bool sliced = ns >= 2 && g_wgrad_ws != nullptr && pstr * ns <= g_wgrad_ws_floats && slice < (1ll << 31);
        ga.partial = sliced ? g_wgrad_ws : nullptr;
        ga.pstride = pstr;
        ga.bias_off = slice;
        P2I_LAUNCH(kern, dim3(ns, ncb, nco), dim3(512), lds2, s, ga);
        if (int e = launch_status()) return e;
        if (sliced)
          if (int e = launch_wgrad_reduce(g_wgrad_ws, ns, slice, d->Cout, g.CoPad, ga.dwp, s, ga.dbias, pstr)) return e;
      }
    }
  }
  if (!use_dma) {
    // slice mode here too (round 4: maps narrower than 8 columns -- the deep levels of small crops -- took nsplit-way float atomics):
    // as many pixel splits as the scratch holds slices for, each storing its tile, summed in split order
    const long long slice = (long long)g.tpg * d->Cin * g.CoPad;
    if (g_wgrad_ws != nullptr && nsplit >= 2 && slice < (1ll << 31)) {
      const long long fit = g_wgrad_ws_floats / slice;
      if (fit < nsplit) nsplit = fit >= 2 ? (int)fit : 1;
    }
    const bool sliced = nsplit >= 2 && g_wgrad_ws != nullptr && slice * nsplit <= g_wgrad_ws_floats && slice < (1ll << 31);
    for (int a = 0; a < d->kt; ++a) {   // one launch per kt slice (the patch's t origin differs per slice)
      WgradGeom ga = g;
      ga.nsplit = nsplit;
      ga.bT = a - d->pt;
      ga.ntaps = g.tpg;
      ga.tpg = g.tpg;
      ga.dwp = dwp + (size_t)a * g.tpg * d->Cin * g.CoPad;
      ga.partial = sliced ? g_wgrad_ws : nullptr;
      ga.pstride = slice;
      for (int i = 0; i < g.tpg; ++i) ga.tap_off[i] = g.tap_off[a * g.tpg + i];
      P2I_LAUNCH(wgrad_kernel<NPIX>, dim3(nsplit, ncx, nco), dim3(256), lds, s, ga);
      if (int e = launch_status()) return e;
      if (sliced)
        if (int e = launch_wgrad_reduce(g_wgrad_ws, nsplit, slice, d->Cout, g.CoPad, ga.dwp, s)) return e;
    }
  }
  if (dbias && !bias_fused) return p2i_bias_grad(dy, y_act, act, dbias, d->B, d->Cout, (int64_t)d->To * d->Ho * d->Wo, stream);
  return P2I_OK;
}
