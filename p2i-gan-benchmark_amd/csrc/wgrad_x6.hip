// Weight gradient of 3x3 (and 3x3x3, as three t slices) layers with spatial stride 1 and C_in, C_out multiples of 64 -- the generator's
// DO-Conv stack, the discriminators' 256 -> 256 and 128 -> 128 stride-(2,1,1) layers, with their bias gradient -- on the bf16 matrix pipe with
// fp32 accuracy: both operands are activations (x and dy), each split exactly into three bf16 terms (hi + mid + lo, truncation
// split), six v_mfma_f32_32x32x16_bf16 products per fp32 product accumulated in fp32, small terms first (conv_x6c.hip has the
// numerics; the dropped products are <= 2^-23 |a b|).
//
//   dW[tap][c][o] = sum over pixels P of x[c][P + d_tap] * dy[o][P]
//
// The contraction runs over PIXELS, so an MFMA operand is "8 consecutive pixels of one channel" -- and a tap moves the x operand by
// +-1 pixel = 2 bytes, which no aligned 16-byte read of a [channel][pixel] image can follow.  The LDS images are therefore
// [pixel][64 channels] (a tap shift is a whole ROW of the image) and every operand is fetched with gfx950's transposing read
// ds_read_b64_tr_b16 (4 pixel rows x 16 channels per 16-lane group, delivered channel-per-lane): two reads per bf16x8 operand.
//   * workgroup = 64 x-channels x 64 dy-channels x 9 taps over a slice of the pixels (grid.x slices, summed by
//     wgrad_reduce_kernel exactly like the f32 kernel's slices); 8 waves = 2 tap groups x (2 x 2) channel quadrants: group 0 owns
//     taps 0-3, group 1 taps 5-8, the centre tap is shared (group 0 takes tile rows 0-1, group 1 rows 2-3, summed through LDS at
//     the end): 108 MFMAs per wave and tile in both groups, five 32x32 accumulators per wave;
//   * tile = 4 rows x 16 columns of output pixels (+ halo for x): both tiles go global -> registers -> split -> LDS as bf16 planes
//     [plane][pixel][64 ch] with a 144-byte pixel pitch (writes conflict-free; transposed reads conflict-free since round 3: each read
//     takes pixel rows r, r+4, r+8, r+12 -- wx_read_tr_s4), double
//     buffered, ONE barrier per tile; no LDS-DMA (nothing here is consumed in its memory layout), so hipcc counts the waits;
//   * a K-step is one tile row (16 pixels): 6 transposed reads for dy, 6 per tap for x, 24-30 MFMAs per wave; both tap groups run
//     one static schedule of 18 tap-steps per tile with the operand reads two steps ahead and the next tile's staging (24 buffer
//     loads with hardware zero for the halo, 9 split + ds_write_b128 chunks) spread over the steps.
// MFMA-bound time of a 9.1 GFLOP layer: 30.7 us at the 1.8 GHz the chip holds under this load; measured 49 us (f32 kernel: 84.5 us).
#include "conv_common.h"

namespace p2i {

#ifdef P2I_STAMP
// diagnostic build (tools/build_stamp.sh, tools/stamp_wgrad_x6.py): phase stamps and KNOCK-OUT variants of the tile loop (DIAG bits:
// 1 no global loads, 2 no split / LDS writes, 4 no MFMAs, 8 no operand reads) -- wrong results, read times only; never in the product
extern __device__ unsigned long long* p2i_stamp_buf;
#define WX_NOW(v) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define WX_NOW(v) do { } while (0)
#endif

typedef short s16x4w __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8w __attribute__((ext_vector_type(8)));
typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
typedef float f32x4w __attribute__((ext_vector_type(4)));

struct Wx6Geom {
  const float* x;        // (B, Cx, sT, H, W)
  const float* dy;       // (B, Co, nT, H, W)
  float* dwp;            // packed grad [kt * 9][Cx][CoPad]
  float* partial;        // != null: slice s stores to partial + s * pstride (same indexing), wgrad_reduce_kernel sums
  float* dbias;          // != null: db[o] += sum of dy (added atomically by the channel-block-0 / t-slice-0 workgroups)
  long long pstride;     // floats from one slice to the next (the tile, + CoPad for the slice's bias row when dbias is given)
  long long bias_off;    // where a slice's bias row starts
  int B, Cx, Co, CoPad, H, W;
  int nth, ntw, ntiles;
  unsigned x_bytes, dy_bytes;
  // time axis (3-D layers with spatial stride 1: the 27 taps are kt slices of 9, blockIdx.z = slice * nco + dy-channel block):
  // "images" of the tile loop are (b, to) pairs; slice a pairs dy frame to with x frame to * mT + dt[a] (zero outside [0, sT))
  int sT, nT, mT, nco;
  int dt[3];
};

constexpr int WX_TH = 4, WX_TW = 16, WX_EW = WX_TW + 2, WX_EH = WX_TH + 2;
constexpr int WX_XPX = WX_EH * WX_EW, WX_YPX = WX_TH * WX_TW;          // 108 patch pixels, 64 output pixels
constexpr int WX_ROW = 144;                                            // bytes per pixel row: 64 ch x 2 B + 16 (36 dwords: 4 * odd)
constexpr int WX_XPLANE = WX_XPX * WX_ROW, WX_YPLANE = WX_YPX * WX_ROW;
constexpr int WX_YOFF = 3 * WX_XPLANE, WX_BUF = WX_YOFF + 3 * WX_YPLANE;   // 74 304 B per buffer, two buffers
constexpr int WX_NXI = (8 * WX_XPX + 511) / 512;                       // x items (pixel, 8-channel chunk) per thread: 2

// (lo_elem >> 16) | (hi_elem & 0xFFFF0000) in one v_perm_b32: bytes {lo.2, lo.3, hi.2, hi.3}
__device__ __forceinline__ unsigned wx_pack(float lo_elem, float hi_elem) {
  return __builtin_amdgcn_perm(__float_as_uint(hi_elem), __float_as_uint(lo_elem), 0x07060302u);
}
__device__ __forceinline__ float wx_trunc(float v) { return __uint_as_float(__float_as_uint(v) & 0xFFFF0000u); }
// exact 3-way truncation split of 8 channel values into three packed bf16x8
__device__ __forceinline__ void wx_split8(const float (&v)[8], u32x4w& hi, u32x4w& mid, u32x4w& lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = v[2 * j], b = v[2 * j + 1];
    hi[j] = wx_pack(a, b);
    const float ra = a - wx_trunc(a), rb = b - wx_trunc(b);
    mid[j] = wx_pack(ra, rb);
    const float sa = ra - wx_trunc(ra), sb = rb - wx_trunc(rb);
    lo[j] = wx_pack(sa, sb);
  }
}

typedef __attribute__((address_space(3))) s16x4w* lds_s16x4_ptr;
// bf16x8 MFMA operand = 8 consecutive pixels of this lane's channel: two transposed reads of 4 pixel rows each
__device__ __forceinline__ bf16x8w wx_read_tr(const unsigned char* base, int byte_off) {
  const s16x4w a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + byte_off));
  const s16x4w b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + byte_off + 4 * WX_ROW));
  return __builtin_bit_cast(bf16x8w, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}
// The same with the 16 pixel rows of a K-step dealt out so that no two lanes of a 32-lane half meet on a bank: the contraction index
// of an MFMA has no order, only x and dy must agree on it.  A transposed read takes the four rows its lanes address (lane 4q+p: row
// q of the block); with rows r0, r0+4, r0+8, r0+12 and the 144-byte pitch (36 dwords) the eight (row, 16-channel group) windows of a
// half land on banks 16 q + 8 g .. + 7 -- all 64 banks once.  (Rows r0 .. r0+3, round 2: rows q and q+2 meet: 43 % of the LDS cycles
// were conflict cycles.)  Half lhi reads rows 2 lhi + 4 q and, second read, one row further: every row of the K-step exactly once.
__device__ __forceinline__ bf16x8w wx_read_tr_s4(const unsigned char* base, int byte_off) {
  const s16x4w a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + byte_off));
  const s16x4w b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + byte_off + WX_ROW));
  return __builtin_bit_cast(bf16x8w, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <bool S4, int DIAG = 0>
__global__ __launch_bounds__(512) void wgrad_x6_kernel(const Wx6Geom g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
#ifdef P2I_STAMP
  unsigned long long st_entry, st_loop0 = 0, st_loop1 = 0, st_bar = 0, st_t;
  WX_NOW(st_entry);
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tg = wave >> 2, kh = (wave >> 1) & 1, mh = wave & 1;        // tap group, x-channel half, dy-channel half
  const int i16 = lane & 15, g16 = (lane >> 4) & 1, lhi = lane >> 5, l31 = lane & 31;
  const int HW = g.H * g.W;
  const int cb = blockIdx.y, ob = blockIdx.z % g.nco, ta = blockIdx.z / g.nco;      // x-channel block, dy-channel block, t slice
  const int dta = ta == 2 ? g.dt[2] : (ta == 1 ? g.dt[1] : g.dt[0]);
  const int xcs = g.sT * HW, ycs = g.nT * HW;                                     // channel strides (elements)

  // ---- staging items of this thread: (patch pixel q, 8-channel chunk); lanes run along q (coalesced loads, conflict-free writes)
  int x_rel[WX_NXI], x_dst[WX_NXI], x_qr[WX_NXI], x_qc[WX_NXI];
#pragma unroll
  for (int it = 0; it < WX_NXI; ++it) {
    const int e = tid + 512 * it;
    const bool in = e < 8 * WX_XPX;
    const int chunk = in ? e / WX_XPX : 0, q = in ? e - chunk * WX_XPX : 0;
    const int qr = q / WX_EW, qc = q - qr * WX_EW;
    x_qr[it] = in ? qr - 1 : -(1 << 20);                               // row / column relative to the tile origin; "never valid" when idle
    x_qc[it] = qc - 1;
    x_rel[it] = (cb * 64 + chunk * 8) * xcs + (qr - 1) * g.W + (qc - 1);
    x_dst[it] = q * WX_ROW + chunk * 16;
  }
  const int y_chunk = tid >> 6, y_q = tid & 63;
  const int y_rel = (ob * 64 + y_chunk * 8) * ycs + (y_q >> 4) * g.W + (y_q & 15);
  const int y_dst = WX_YOFF + y_q * WX_ROW + y_chunk * 16;

  // The three staging items of a thread (two x items, one dy item; 8 channel values each) live in sv[3][8].  Next tile's 24 loads are
  // spread over tap-steps 0..7 of the current tile and its split + LDS writes over tap-steps 9..17 (one plane of one item per step),
  // so that this VALU / VMEM / DS-write work sits in the shadow of the MFMAs instead of between two tiles (15 of 62 us before).
  float sv[3][8];
  // Loads go through buffer descriptors: a pixel of the halo that lies outside the image gets an out-of-range offset and comes back
  // as 0 from the hardware range check (no per-element select, no 64-bit address arithmetic: the channel stride rides in the
  // scalar offset).  so[3]: byte offset of the item's first channel, or 0xFFFFFF00 when the pixel is outside.
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.x), 0, (int)g.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.dy), 0, (int)g.dy_bytes, 0x00020000);
  unsigned so[3];
  const bool x1_in = tid + 512 < 8 * WX_XPX;          // this thread has a second x item
  const int dummy_dst = 2 * WX_BUF + lane * 16;        // LDS slot for the writes of an idle second item (keeps the code branch-free:
                                                       // the transposed reads interleaved with it need EXEC all ones)
  auto tile_ptrs = [&](int t) {
    const int tw = t % g.ntw, r0 = t / g.ntw;
    const int th = r0 % g.nth, img = r0 / g.nth;
    const int b = img / g.nT, to = img - b * g.nT;
    const int ti = to * g.mT + dta;                     // x frame of this t slice (outside the clip: the whole x tile reads as zero)
    const bool tok = (unsigned)ti < (unsigned)g.sT;
    const int h0 = th * WX_TH, w0 = tw * WX_TW;
    const int xorg = (b * g.Cx * g.sT + ti) * HW + h0 * g.W + w0, yorg = (b * g.Co * g.nT + to) * HW + h0 * g.W + w0;
#pragma unroll
    for (int it = 0; it < WX_NXI; ++it) {
      const int h = h0 + x_qr[it], w = w0 + x_qc[it];
      const bool ok = tok && (unsigned)h < (unsigned)g.H && (unsigned)w < (unsigned)g.W;
      so[it] = ok ? 4u * (unsigned)(xorg + x_rel[it]) : 0xFFFFFF00u;
    }
    so[2] = 4u * (unsigned)(yorg + y_rel);
  };
  const int xchan_bytes = 4 * xcs, ychan_bytes = 4 * ycs;
  auto load_chunk = [&](int c) {                        // c = 0..7: loads 3c .. 3c+2 of the 24
#pragma unroll
    for (int f = 3 * c; f < 3 * c + 3; ++f)
      sv[f >> 3][f & 7] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32((f >> 3) == 2 ? rs_y : rs_x, so[f >> 3],
                                                                                      (f & 7) * ((f >> 3) == 2 ? ychan_bytes : xchan_bytes), 0));
  };
  // bias gradient: every dy value passes through sv[2] exactly once per (x-channel block, t slice) pass; the blocks with cb == 0
  // and t slice 0 sum theirs (wave = 8 channels, lane = pixel of the tile)
  const bool do_bias = g.dbias != nullptr && cb == 0 && ta == 0;
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float bias_w = 1.f;                                   // 0 while the loop re-stages its last tile (nothing new)
  // c = 0..8: plane c % 3 (hi, mid, lo) of item c / 3 -> packed bf16x8 to LDS; the item's values become their own remainder
  auto split_chunk = [&](int c, unsigned char* buf) {
    const int it = c / 3, pl = c % 3;
    float (&v)[8] = sv[it];
    if (c == 6 && do_bias) {
#pragma unroll
      for (int j = 0; j < 8; ++j) bsum[j] += bias_w * v[j];
    }
    u32x4w w;
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = wx_pack(v[2 * j], v[2 * j + 1]);
    const int dst = it == 2 ? y_dst + pl * WX_YPLANE : (it == 1 && !x1_in ? -1 : x_dst[it] + pl * WX_XPLANE);
    *reinterpret_cast<u32x4w*>(dst < 0 ? wsm + dummy_dst : buf + dst) = w;
    if (pl < 2) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = v[j] - wx_trunc(v[j]);         // exact
    }
  };

  // acc[0..3]: this group's own taps (tap = 5 * tg + k); acc[4]: its share of the centre tap
  f32x16 acc[5];
#pragma unroll
  for (int t = 0; t < 5; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // transposed-read lane bases (bytes inside a buffer): lane 4q+p of a 16-lane group addresses pixel row q, channels 4p .. 4p+3
  const int lrow = S4 ? 2 * lhi + 4 * (i16 >> 2) : 8 * lhi + (i16 >> 2);                               // (rows: see wx_read_tr_s4)
  const int a_lane = lrow * WX_ROW + (32 * kh + 16 * g16 + 4 * (i16 & 3)) * 2;
  const int b_lane = WX_YOFF + lrow * WX_ROW + (32 * mh + 16 * g16 + 4 * (i16 & 3)) * 2;
  int a_tap[5];                                       // + the tap's pixel shift (dh, dw) -> dh * 18 + dw patch pixels; [4] = centre
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int tap = 5 * tg + k;
    a_tap[k] = a_lane + ((tap / 3) * WX_EW + tap % 3) * WX_ROW;
  }
  a_tap[4] = a_lane + (WX_EW + 1) * WX_ROW;
  // Both groups run the same static schedule of 18 tap-steps over 4 K-steps (tile rows): K-steps 0 and 1 carry the centre tap too.
  // Group 1 starts at tile row 2, so "its" centre rows are 2 and 3: row of K-step s = (s + 2 tg) & 3 (a wave-uniform byte offset).
  int rowx[WX_TH], rowy[WX_TH];
#pragma unroll
  for (int ks = 0; ks < WX_TH; ++ks) {
    const int row = (ks + 2 * tg) & 3;
    rowx[ks] = row * WX_EW * WX_ROW;
    rowy[ks] = row * WX_TW * WX_ROW;
  }
  constexpr int NSTEP = 18;
  constexpr int ST_KS[NSTEP] = {0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3};
  constexpr int ST_K[NSTEP] = {0, 1, 2, 3, 4, 0, 1, 2, 3, 4, 0, 1, 2, 3, 0, 1, 2, 3};

  constexpr int PA[6] = {2, 0, 1, 1, 0, 0};            // small terms first: lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi
  constexpr int PB[6] = {0, 2, 1, 0, 1, 0};

  int t = blockIdx.x;
  if (t < g.ntiles) {
    tile_ptrs(t);
#pragma unroll
    for (int c = 0; c < 8; ++c) load_chunk(c);
#pragma unroll
    for (int c = 0; c < 9; ++c) split_chunk(c, wsm);
  }
  __syncthreads();
  int cur = 0;
  WX_NOW(st_loop0);
  for (; t < g.ntiles; t += gridDim.x) {
    const int tn = t + gridDim.x;
    const bool more = tn < g.ntiles;
    tile_ptrs(more ? tn : t);                          // (the last tile re-stages itself into the idle buffer: branch-free loop body)
    bias_w = more ? 1.f : 0.f;
    const unsigned char* buf = wsm + cur * WX_BUF;
    unsigned char* nbuf = wsm + (cur ^ 1) * WX_BUF;    // read last in the previous tile (all waves are past its barrier)
    // software pipeline over the 18 tap-steps: step i+1's six operand reads (and the next K-step's six dy reads) are issued under
    // step i's six MFMAs; two register sets each
    bf16x8w Av[3][3], Bv[2][3];
    auto load_a = [&](int set, int step) {
      const unsigned char* p = buf + a_tap[ST_K[step]] + rowx[ST_KS[step]];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) Av[set][pl] = S4 ? wx_read_tr_s4(p, pl * WX_XPLANE) : wx_read_tr(p, pl * WX_XPLANE);
    };
    auto load_b = [&](int set, int ks) {
      const unsigned char* p = buf + b_lane + rowy[ks];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) Bv[set][pl] = S4 ? wx_read_tr_s4(p, pl * WX_YPLANE) : wx_read_tr(p, pl * WX_YPLANE);
    };
    // x operands are fetched TWO steps ahead (three register sets), dy operands at the first step of the K-step before
    load_b(0, 0);
    load_a(0, 0);
    load_a(1, 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NSTEP; ++i) {
      const int ks = ST_KS[i], k = ST_K[i];
      const bool pre_a = i + 2 < NSTEP;
      const bool pre_b = ks + 1 < WX_TH && (i == 0 || ST_KS[i - 1] != ks);      // first step of a K-step: fetch the next K-step's dy
      if (i < 8 && !(DIAG & 1)) load_chunk(i);
      if (i >= 9 && !(DIAG & 2)) split_chunk(i - 9, nbuf);
      if (pre_a && !(DIAG & 8)) load_a((i + 2) % 3, i + 2);
      if (pre_b && !(DIAG & 8)) load_b((ks + 1) & 1, ks + 1);
      if (!(DIAG & 4)) {
#pragma unroll
        for (int q = 0; q < 6; ++q)
          acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Av[i % 3][PA[q]], Bv[ks & 1][PB[q]], acc[k], 0, 0, 0);
      } else {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) asm volatile("" :: "v"(Av[i % 3][pl]), "v"(Bv[ks & 1][pl]));      // keep the reads alive
      }
      // interleave: per MFMA one (two) operand reads and a share of the staging work
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (pre_a && pre_b) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        else if (pre_a || pre_b) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (i < 8 && q < 3) { __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
        if (i >= 9) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        if (i >= 9 && q == 5) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#ifdef P2I_STAMP
    WX_NOW(st_t);
#endif
    __syncthreads();
#ifdef P2I_STAMP
    { unsigned long long t2; WX_NOW(t2); st_bar += t2 - st_t; }
#endif
    cur ^= 1;
  }
  WX_NOW(st_loop1);

  // ---- store / accumulate the [tap][c][o] tile.  Own taps go straight out; group 1 parks its share of the centre tap in LDS
  // (the buffers are free now) and group 0 adds it to its own.
  f32x4w* red = reinterpret_cast<f32x4w*>(wsm);
  const int quad = kh * 2 + mh;
  if (tg == 1) {
#pragma unroll
    for (int j4 = 0; j4 < 4; ++j4) {
      f32x4w v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[4][4 * j4 + e];
      red[(quad * 4 + j4) * 64 + lane] = v;
    }
  }
  __syncthreads();
  if (tg == 0) {
#pragma unroll
    for (int j4 = 0; j4 < 4; ++j4) {
      const f32x4w v = red[(quad * 4 + j4) * 64 + lane];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[4][4 * j4 + e] += v[e];
    }
  }
  const size_t tap_stride = (size_t)g.Cx * g.CoPad;
  float* dst = (g.partial ? g.partial + (size_t)blockIdx.x * g.pstride : g.dwp) + (size_t)ta * 9 * tap_stride +
               ((size_t)cb * 64 + 32 * kh + 4 * lhi) * g.CoPad + ob * 64 + 32 * mh + l31;
  if (do_bias) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float sj = wave_sum(bsum[j]);
      if (lane == 0) {
        if (g.partial) g.partial[(size_t)blockIdx.x * g.pstride + g.bias_off + ob * 64 + y_chunk * 8 + j] = sj;   // summed in slice order by wgrad_reduce_kernel
        else atomicAdd(g.dbias + ob * 64 + y_chunk * 8 + j, sj);
      }
    }
  }
  auto emit = [&](auto&& put) {
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      if (k == 4 && tg == 1) break;
      float* dt = dst + (k == 4 ? 4 : 5 * tg + k) * tap_stride;
#pragma unroll
      for (int j = 0; j < 16; ++j) put(dt + ((j & 3) + 8 * (j >> 2)) * g.CoPad, acc[k][j]);
    }
  };
  if (g.partial) emit([](float* o, float v) { *o = v; });
  else emit([](float* o, float v) { atomicAdd(o, v); });
#ifdef P2I_STAMP
  if (p2i_stamp_buf && lane == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long st_end; WX_NOW(st_end);
    unsigned long long* o = p2i_stamp_buf + ((size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8 + wave) * 8;
    o[0] = st_loop0 - st_entry; o[1] = st_loop1 - st_loop0; o[2] = st_end - st_loop1; o[3] = st_bar; o[4] = st_entry; o[5] = st_end;
  }
#endif
}


// ---------------------------------------------------------------------------------------------------------------------------------
// "wgrad_x6p": the same LDS images, transposed operand reads, slices and numerics as wgrad_x6_kernel with the eight waves SPECIALISED
// (the scheme that took the convolution kernels from 63 % to 86 % of their MFMA-bound loop time, conv_x6c.hip "x6p"): waves 0-3 are
// CONSUMERS -- one per SIMD, channel quadrant (kh, mh), ALL nine taps (nine 32x32 accumulators), nothing but transposed reads and
// MFMAs: 216 MFMAs per 64-pixel tile -- and waves 4-7 are PRODUCERS: they load tile i+2 into registers (hardware zero for the halo),
// split and write tile i+1 into the other buffer, and sum the bias gradient on the way.  One barrier per tile.
__global__ __launch_bounds__(512) void wgrad_x6p_kernel(const Wx6Geom g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int HW = g.H * g.W;
  const int cb = blockIdx.y, ob = blockIdx.z % g.nco, ta = blockIdx.z / g.nco;
  const int dta = ta == 2 ? g.dt[2] : (ta == 1 ? g.dt[1] : g.dt[0]);
  const int xcs = g.sT * HW, ycs = g.nT * HW;
  const int nmine = blockIdx.x < g.ntiles ? (g.ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;   // tiles of this workgroup

  if (wave >= 4) {
    // =============================================================== producers
    const int p = tid - 256;
    constexpr int NX = (8 * WX_XPX + 255) / 256;       // x items (patch pixel, 8-channel chunk) per thread: 4
    constexpr int NY = (8 * WX_YPX) / 256;             // dy items per thread: 2
    int x_rel[NX], x_dst[NX], x_qr[NX], x_qc[NX];
    bool x_in[NX];
#pragma unroll
    for (int it = 0; it < NX; ++it) {
      const int e = p + 256 * it;
      x_in[it] = e < 8 * WX_XPX;
      const int chunk = x_in[it] ? e / WX_XPX : 0, q = x_in[it] ? e - chunk * WX_XPX : 0;
      const int qr = q / WX_EW, qc = q - qr * WX_EW;
      x_qr[it] = x_in[it] ? qr - 1 : -(1 << 20);
      x_qc[it] = qc - 1;
      x_rel[it] = (cb * 64 + chunk * 8) * xcs + (qr - 1) * g.W + (qc - 1);
      x_dst[it] = q * WX_ROW + chunk * 16;
    }
    int y_rel[NY], y_dst[NY];
#pragma unroll
    for (int it = 0; it < NY; ++it) {
      const int e = p + 256 * it;                      // = chunk * 64 + pixel: a wave holds the 64 pixels of one 8-channel chunk
      const int chunk = e >> 6, q = e & 63;
      y_rel[it] = (ob * 64 + chunk * 8) * ycs + (q >> 4) * g.W + (q & 15);
      y_dst[it] = WX_YOFF + q * WX_ROW + chunk * 16;
    }
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.x), 0, (int)g.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.dy), 0, (int)g.dy_bytes, 0x00020000);
    const int xchan_bytes = 4 * xcs, ychan_bytes = 4 * ycs;
    float xv[NX][8], yv[NY][8];
    auto load_tile = [&](int t) {
      const int tw = t % g.ntw, r0 = t / g.ntw;
      const int th = r0 % g.nth, img = r0 / g.nth;
      const int b = img / g.nT, to = img - b * g.nT;
      const int ti = to * g.mT + dta;
      const bool tok = (unsigned)ti < (unsigned)g.sT;
      const int h0 = th * WX_TH, w0 = tw * WX_TW;
      const int xorg = (b * g.Cx * g.sT + ti) * HW + h0 * g.W + w0, yorg = (b * g.Co * g.nT + to) * HW + h0 * g.W + w0;
#pragma unroll
      for (int it = 0; it < NX; ++it) {
        const int h = h0 + x_qr[it], w = w0 + x_qc[it];
        const bool ok = tok && (unsigned)h < (unsigned)g.H && (unsigned)w < (unsigned)g.W;
        const unsigned so = ok ? 4u * (unsigned)(xorg + x_rel[it]) : 0xFFFFFF00u;       // out of range: the hardware returns 0
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[it][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, so, j * xchan_bytes, 0));
      }
#pragma unroll
      for (int it = 0; it < NY; ++it) {
        const unsigned so = 4u * (unsigned)(yorg + y_rel[it]);
#pragma unroll
        for (int j = 0; j < 8; ++j) yv[it][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_y, so, j * ychan_bytes, 0));
      }
    };
    const bool do_bias = g.dbias != nullptr && cb == 0 && ta == 0;
    float bsum[NY][8];
#pragma unroll
    for (int it = 0; it < NY; ++it)
#pragma unroll
      for (int j = 0; j < 8; ++j) bsum[it][j] = 0.f;
    auto split3 = [&](float (&v)[8], unsigned char* dst, int plane_bytes) {      // three planes of one item, exact truncation split
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        u32x4w w;
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = wx_pack(v[2 * j], v[2 * j + 1]);
        *reinterpret_cast<u32x4w*>(dst + pl * plane_bytes) = w;
        if (pl < 2) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = v[j] - wx_trunc(v[j]);
        }
      }
    };
    auto write_tile = [&](unsigned char* buf) {
      if (do_bias) {
#pragma unroll
        for (int it = 0; it < NY; ++it)
#pragma unroll
          for (int j = 0; j < 8; ++j) bsum[it][j] += yv[it][j];
      }
#pragma unroll
      for (int it = 0; it < NX; ++it)
        if (x_in[it]) split3(xv[it], buf + x_dst[it], WX_XPLANE);
#pragma unroll
      for (int it = 0; it < NY; ++it) split3(yv[it], buf + y_dst[it], WX_YPLANE);
    };
    int t = blockIdx.x;
    if (nmine > 0) { load_tile(t); write_tile(wsm); }
    if (nmine > 1) load_tile(t + gridDim.x);
    __syncthreads();                                                     // tile 0 handed over
    for (int i = 0; i < nmine; ++i) {
      // consumers read buffer i & 1; tile i+1 (in registers since the previous trip) goes into the other one, tile i+2 into registers
      if (i + 1 < nmine) write_tile(wsm + ((i + 1) & 1) * WX_BUF);
      if (i + 2 < nmine) load_tile(t + (i + 2) * gridDim.x);
      __syncthreads();
    }
    if (do_bias) {
#pragma unroll
      for (int it = 0; it < NY; ++it)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float sj = wave_sum(bsum[it][j]);
          if (lane == 0) {
            if (g.partial) g.partial[(size_t)blockIdx.x * g.pstride + g.bias_off + ob * 64 + (((p + 256 * it) >> 6)) * 8 + j] = sj;
            else atomicAdd(g.dbias + ob * 64 + (((p + 256 * it) >> 6)) * 8 + j, sj);
          }
        }
    }
    return;
  }

  // ================================================================= consumers
  const int kh = wave >> 1, mh = wave & 1;                                // x-channel half, dy-channel half of the 64 x 64 block
  const int i16 = lane & 15, g16 = (lane >> 4) & 1, lhi = lane >> 5, l31 = lane & 31;
  f32x16 acc[9];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[tp][r] = 0.f;
  const int a_lane = (8 * lhi + (i16 >> 2)) * WX_ROW + (32 * kh + 16 * g16 + 4 * (i16 & 3)) * 2;
  const int b_lane = WX_YOFF + (8 * lhi + (i16 >> 2)) * WX_ROW + (32 * mh + 16 * g16 + 4 * (i16 & 3)) * 2;
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0};            // small terms first: lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi
  constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
  constexpr int NSTEP = 36;                            // (tile row ks, tap): 4 x 9
  __syncthreads();                                                       // tile 0 handed over
  for (int i = 0; i < nmine; ++i) {
    const unsigned char* buf = wsm + (i & 1) * WX_BUF;
    bf16x8w Av[3][3], Bv[2][3];
    auto load_a = [&](int set, int step) {
      const int ks = step / 9, tap = step % 9;
      const unsigned char* q = buf + a_lane + (ks * WX_EW + (tap / 3) * WX_EW + tap % 3) * WX_ROW;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) Av[set][pl] = wx_read_tr(q, pl * WX_XPLANE);
    };
    auto load_b = [&](int set, int ks) {
      const unsigned char* q = buf + b_lane + ks * WX_TW * WX_ROW;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) Bv[set][pl] = wx_read_tr(q, pl * WX_YPLANE);
    };
    load_b(0, 0);
    load_a(0, 0);
    load_a(1, 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
      const int ks = st / 9, tap = st % 9;
      const bool pre_a = st + 2 < NSTEP;
      const bool pre_b = tap == 0 && ks + 1 < WX_TH;                       // first tap of a tile row: fetch the next row's dy
      if (pre_a) load_a((st + 2) % 3, st + 2);
      if (pre_b) load_b((ks + 1) & 1, ks + 1);
#pragma unroll
      for (int q = 0; q < 6; ++q)
        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Av[st % 3][PA[q]], Bv[ks & 1][PB[q]], acc[tap], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (pre_a && pre_b) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        else if (pre_a || pre_b) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }
  // ---- store / accumulate the nine [c][o] tiles of this quadrant
  const size_t tap_stride = (size_t)g.Cx * g.CoPad;
  float* dst = (g.partial ? g.partial + (size_t)blockIdx.x * g.pstride : g.dwp) + (size_t)ta * 9 * tap_stride +
               ((size_t)cb * 64 + 32 * kh + 4 * lhi) * g.CoPad + ob * 64 + 32 * mh + l31;
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) {
    float* dt = dst + tp * tap_stride;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float* o = dt + ((j & 3) + 8 * (j >> 2)) * g.CoPad;
      if (g.partial) *o = acc[tp][j];
      else atomicAdd(o, acc[tp][j]);
    }
  }
}

// 0 = launched (plan filled), 1 = not a case of this kernel (caller continues with the f32 kernels)
int run_wgrad_x6(const p2i_conv_desc* d, const float* x, const float* dy, float* dwp, float* dbias, float* ws, long long ws_floats,
                 int* ns_out, long long* slice_out, hipStream_t s) {
  { const char* e = getenv("P2I_WGRAD_X6"); if (e && atoi(e) == 0) return 1; }     // read per call: tests run both engines in one process
  if (d->kh != 3 || d->kw != 3 || d->sh != 1 || d->sw != 1 || d->ph != 1 || d->pw != 1) return 1;
  const bool flat = d->kt == 1 && d->st == 1 && d->pt == 0;             // 2-D layer (or frames convolved independently)
  const bool vol = d->kt == 3 && d->pt == 1 && d->st <= 2;              // 3 x 3 x 3: three t slices of nine taps
  if (!flat && !vol) return 1;
  if ((d->Cin & 63) || (d->Cout & 63) || (d->Wo % WX_TW) || (d->Ho % WX_TH)) return 1;
  const long long nx = (long long)d->B * d->Cin * d->Ti * d->Hi * d->Wi, ny = (long long)d->B * d->Cout * d->To * d->Ho * d->Wo;
  if (nx >= (1ll << 30) - 64 || ny >= (1ll << 30) - 64) return 1;        // byte offsets in 32 bits, below the out-of-range marker
  Wx6Geom g{};
  g.x = x; g.dy = dy; g.dwp = dwp; g.dbias = dbias;
  g.B = d->B; g.Cx = d->Cin; g.Co = d->Cout; g.CoPad = (d->Cout + 31) / 32 * 32; g.H = d->Ho; g.W = d->Wo;
  g.x_bytes = (unsigned)(4 * nx); g.dy_bytes = (unsigned)(4 * ny);
  g.sT = d->Ti; g.nT = d->To; g.mT = d->st;
  for (int a = 0; a < d->kt; ++a) g.dt[a] = a - d->pt;
  g.nth = d->Ho / WX_TH; g.ntw = d->Wo / WX_TW; g.ntiles = d->B * d->To * g.nth * g.ntw;
  const int ncb = d->Cin / 64, nco = d->Cout / 64;
  g.nco = nco;
  int ns = 256 / (ncb * nco * d->kt);                  // LDS admits one workgroup per CU
  if (ns < 1) ns = 1;
  if (ns > g.ntiles) ns = g.ntiles;
  const long long slice = 9ll * d->kt * d->Cin * g.CoPad;
  const long long pstr = slice + (dbias ? g.CoPad : 0);                // (+ the slice's bias row)
  const bool sliced = ns >= 2 && ws != nullptr && pstr * ns <= ws_floats && slice < (1ll << 31);
  if (ns >= 2 && !sliced) return 1;                    // no scratch for the slices: the f32 kernel's atomic path
  g.partial = sliced ? ws : nullptr;
  g.pstride = pstr;
  g.bias_off = slice;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)wgrad_x6_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)wgrad_x6_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)wgrad_x6p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  // P2I_WGRAD_X6_PC=1: the producer / consumer variant.  Measured SLOWER than the symmetric kernel at B = 8 (gpurun_out/r03n: 64-channel
  // level 73.5 vs 66.2 us, 256: 68.8 vs 64.5, 3-D 128 -> 128: 93.8 vs 87.5): a consumer issues 1.1 transposed 8-byte reads per MFMA
  // (the convolution consumers 0.6-0.75 16-byte reads) and has no partner wave on its SIMD to cover the 2-way conflicted ones.  Kept as
  // an option (same results, tested); read per call.
  const char* pce = getenv("P2I_WGRAD_X6_PC");
  if (pce != nullptr && atoi(pce) != 0) P2I_LAUNCH(wgrad_x6p_kernel, dim3(ns, ncb, nco * d->kt), dim3(512), 2 * WX_BUF + 1024, s, g);
  // conflict-free row assignment of the transposed reads (wx_read_tr_s4): the default; P2I_WGRAD_X6_S4=0 keeps round 2's rows (A/B, read
  // per call).  Measured in one box (gpurun_out/r03x/wg_s0.log, wg_s4.log, wg_s0b.log; PMC: gpurun_out/r03x/pmc): conflict cycles 43 % ->
  // 0.5 % of the LDS cycles, launch time within 1 % (66.5-71.5 us either way): the transposed reads were never what the kernel waits for.
#ifdef P2I_STAMP
  else if (getenv("P2I_WGRAD_DIAG") && atoi(getenv("P2I_WGRAD_DIAG")) != 0) {
    const int dg = atoi(getenv("P2I_WGRAD_DIAG"));
    const dim3 gr(ns, ncb, nco * d->kt);
#define WX_DIAG_CASE(n) case n: { static bool a_ = false; if (!a_) { (void)hipFuncSetAttribute((const void*)wgrad_x6_kernel<true, n>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); a_ = true; } \
      P2I_LAUNCH((wgrad_x6_kernel<true, n>), gr, dim3(512), 2 * WX_BUF + 1024, s, g); break; }
    switch (dg) { WX_DIAG_CASE(1) WX_DIAG_CASE(2) WX_DIAG_CASE(3) WX_DIAG_CASE(4) WX_DIAG_CASE(8) WX_DIAG_CASE(12) WX_DIAG_CASE(11) WX_DIAG_CASE(7)
      default: P2I_LAUNCH(wgrad_x6_kernel<true>, gr, dim3(512), 2 * WX_BUF + 1024, s, g); }
#undef WX_DIAG_CASE
  }
#endif
  else if (!(getenv("P2I_WGRAD_X6_S4") && atoi(getenv("P2I_WGRAD_X6_S4")) == 0))
    P2I_LAUNCH(wgrad_x6_kernel<true>, dim3(ns, ncb, nco * d->kt), dim3(512), 2 * WX_BUF + 1024, s, g);
  else P2I_LAUNCH(wgrad_x6_kernel<false>, dim3(ns, ncb, nco * d->kt), dim3(512), 2 * WX_BUF + 1024, s, g);
  *ns_out = sliced ? ns : 0;
  *slice_out = slice;
  return launch_status();
}

}  // namespace p2i
