// Shared pieces of the conv engine (conv.hip: forward / dgrad patch GEMM, wgrad.hip: weight gradient).
#pragma once
#include "common.h"

namespace p2i {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int MAX_TAPS = 27;

typedef __attribute__((address_space(3))) void lds_void;
typedef int v4i32 __attribute__((ext_vector_type(4)));

// LDS-DMA issued from inline asm: hipcc does not count these in vmcnt, so it neither drains them before the
// next ds_read (it cannot prove the DMA's destination buffer differs from the one being read) nor before a
// barrier -- the kernels below wait `s_waitcnt vmcnt(0)` themselves right before the barrier that hands the
// buffer over.  M0 (LDS base of the wave-instruction) is saved/restored inside the same statement; the leading
// `s_nop 4` covers the v_readfirstlane -> VMEM-SGPR-operand hazard (soffset / descriptor words may be fresh),
// which hipcc does not pad inside an asm string.
__device__ __forceinline__ v4i32 make_rsrc(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  v4i32 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));     // stride 0, no swizzle
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}
// lds_addr: BYTE address inside the workgroup's LDS allocation (lds_base(smem) + 4 * float index), wave-uniform
__device__ __forceinline__ unsigned lds_base(const float* shared_array) {
  return (unsigned)(size_t)(const lds_void*)shared_array;
}
__device__ __forceinline__ void dma_b32(const v4i32 rsrc, unsigned lds_addr, int voff, int soff) {
  const unsigned la = __builtin_amdgcn_readfirstlane(lds_addr);
  const int so = __builtin_amdgcn_readfirstlane(soff);
  unsigned keep;
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(la), "s"(rsrc), "s"(so) : "memory");
}
__device__ __forceinline__ void dma_b128(const v4i32 rsrc, unsigned lds_addr, int voff, int soff) {
  const unsigned la = __builtin_amdgcn_readfirstlane(lds_addr);
  const int so = __builtin_amdgcn_readfirstlane(soff);
  unsigned keep;
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(la), "s"(rsrc), "s"(so) : "memory");
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }


// Epilogue of one 32x32 accumulator tile (16 registers per lane: dest channel o_r = o_base + (r&3) + 8*(r>>2) + 4*lhi, one dest
// position per lane): v = act(acc + bias); v += residual; v *= act'(mask).  The 16 residual / mask loads of the tile are issued
// TOGETHER from clamped addresses and masked afterwards: loads under a per-element condition (`if (res) v += res[di]`) make
// hipcc branch around each one and wait for it before the next -- 64 serialised HBM round trips per lane, measured at 50-60k
// cycles per workgroup on the generator's dgrad launches (a quarter of the kernel).
__device__ __forceinline__ void epilogue_tile16(const f32x16& acc, int o_base, int lhi, int Cm, bool pv, size_t pos_off, size_t chan_stride,
                                                const float* __restrict__ bias, int act_epi, const float* __restrict__ res,
                                                const float* __restrict__ mask_y, int mask_act, float* __restrict__ dst,
                                                bool atomic_out = false) {
  // atomic_out: the value is ADDED to a pre-zeroed destination (split-K partial sums; every term of the epilogue must be linear)
  // two batches of 8 with 32-bit element offsets (tensors are < 2^31 elements: check_desc) and a scheduling fence per batch: without
  // the fence hipcc hoists the loads of EVERY tile of the workgroup to the top and the register allocation grows by ~100
#pragma unroll
  for (int h8 = 0; h8 < 2; ++h8) {
    int di[8];
    bool ok[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int r = h8 * 8 + k;
      const int o = o_base + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      ok[k] = pv && o < Cm;
      di[k] = ok[k] ? (int)((size_t)o * chan_stride + pos_off) : 0;
    }
    float rv[8], mv[8];
    if (res) {
#pragma unroll
      for (int k = 0; k < 8; ++k) rv[k] = res[di[k]];
    }
    if (mask_y) {
#pragma unroll
      for (int k = 0; k < 8; ++k) mv[k] = mask_y[di[k]];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int r = h8 * 8 + k;
      const int o = o_base + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      float v = acc[r];
      if (bias) v += bias[o < Cm ? o : 0];
      v = act_apply(v, act_epi);
      if (res) v += rv[k];
      if (mask_y) v = act_grad(v, mv[k], mask_act);
      if (ok[k]) {
        if (atomic_out) atomicAdd(dst + di[k], v);
        else dst[di[k]] = v;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// epilogue_tile16 with 16-byte global accesses.  The 32x32 accumulator tile has one POSITION per lane and the 16 destination channels
// in the registers, so the plain epilogue stores one dword per lane and channel: 16 store instructions of 2 x 128 B per tile (and 16
// loads per residual / mask tensor).  At the end of the bf16-split kernels those instructions, not the bytes, bound the epilogue
// (stamped: 17 k of 93 k cycles of a 64 x 256 tile workgroup).  Here the tile goes through a 32 x 36-float LDS image (wave-private,
// any LDS is free after the main loop) and comes back with four consecutive positions of one channel per lane: 4 dwordx4 stores
// (and 4 dwordx4 loads per epilogue tensor) per tile.  Requirements, checked by the caller: the wave's 32 positions are 8 aligned
// groups of 4 consecutive destination elements (tile width >= 4 positions, destination row pitch and tile origin multiples of 4),
// validity is per group (nW % 4 == 0).  pos4_off: element offset of THIS lane's group (lane & 7) inside a channel; pv4: group valid.
typedef float f32x4e __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void epilogue_tile16_v4(const f32x16& acc, float* __restrict__ tile_lds, int o_base, int lane, int Cm,
                                                   size_t pos4_off, bool pv4, size_t chan_stride,
                                                   const float* __restrict__ bias, int act_epi, const float* __restrict__ res,
                                                   const float* __restrict__ mask_y, int mask_act, float* __restrict__ dst, bool atomic_out = false) {
  constexpr int PITCH = 36;
  const int l31 = lane & 31, lhi = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) tile_lds[((r & 3) + 8 * (r >> 2) + 4 * lhi) * PITCH + l31] = acc[r];
  // (wave-private image: the hardware orders a wave's own LDS accesses; the compiler's lgkmcnt wait covers the read-after-write)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int p4 = (lane & 7) * 4;
  int di[4];
  bool ok[4];
  f32x4e v[4], rv[4], mv[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int ch = k * 8 + (lane >> 3);                       // channel of this lane's k-th group
    const int o = o_base + ch;
    ok[k] = pv4 && o < Cm;
    di[k] = ok[k] ? (int)((size_t)o * chan_stride + pos4_off) : 0;
    v[k] = *reinterpret_cast<const f32x4e*>(tile_lds + ch * PITCH + p4);
  }
  if (res) {
#pragma unroll
    for (int k = 0; k < 4; ++k) rv[k] = *reinterpret_cast<const f32x4e*>(res + di[k]);
  }
  if (mask_y) {
#pragma unroll
    for (int k = 0; k < 4; ++k) mv[k] = *reinterpret_cast<const f32x4e*>(mask_y + di[k]);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int o = o_base + k * 8 + (lane >> 3);
    const float bv = bias ? bias[o < Cm ? o : 0] : 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float x = act_apply(v[k][e] + bv, act_epi);
      if (res) x += rv[k][e];
      if (mask_y) x = act_grad(x, mv[k][e], mask_act);
      v[k][e] = x;
    }
    if (ok[k]) {
      if (atomic_out) {
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(dst + di[k] + e, v[k][e]);
      } else {
        *reinterpret_cast<f32x4e*>(dst + di[k]) = v[k];
      }
    }
  }
  __builtin_amdgcn_wave_barrier();                            // the next tile of this wave reuses the image
}

// Epilogue of TWO accumulator tiles whose destinations interleave along w (input-parity classes (.., pW=0) and (.., pW=1) of a
// stride-2 data gradient, destination row pitch even): element r of both tiles belongs to the same lane and to adjacent
// addresses, so residual / mask are fetched and the result is stored as float2 -- full 128-B lines per 16 lanes instead of two
// half-used ones per class (WRITE_SIZE of the fused strided dgrad was 1.6x its algorithmic bytes).  pos_off: class pW=0, even.
__device__ __forceinline__ void epilogue_pair16(const f32x16& acc0, const f32x16& acc1, int o_base, int lhi, int Cm, bool pv, size_t pos_off,
                                                size_t chan_stride, const float* __restrict__ res, const float* __restrict__ mask_y,
                                                int mask_act, float* __restrict__ dst, bool atomic_out = false) {
  // atomic_out: both values are ADDED to a pre-zeroed destination (split-K partial sums: res rides with one split only, the mask
  // factor distributes over the sum)
  typedef float f32x2e __attribute__((ext_vector_type(2)));
  // batches of 4 (8 floats in flight per tensor): the <32,128,1,*,4> instances must stay below 256 VGPRs to keep two workgroups per CU
#pragma unroll
  for (int h4 = 0; h4 < 4; ++h4) {
    int di[4];
    bool ok[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = h4 * 4 + k;
      const int o = o_base + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      ok[k] = pv && o < Cm;
      di[k] = ok[k] ? (int)((size_t)o * chan_stride + pos_off) : 0;
    }
    f32x2e rv[4], mv[4];
    if (res) {
#pragma unroll
      for (int k = 0; k < 4; ++k) rv[k] = *reinterpret_cast<const f32x2e*>(res + di[k]);
    }
    if (mask_y) {
#pragma unroll
      for (int k = 0; k < 4; ++k) mv[k] = *reinterpret_cast<const f32x2e*>(mask_y + di[k]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = h4 * 4 + k;
      f32x2e v;
      v[0] = acc0[r]; v[1] = acc1[r];
      if (res) { v[0] += rv[k][0]; v[1] += rv[k][1]; }
      if (mask_y) { v[0] = act_grad(v[0], mv[k][0], mask_act); v[1] = act_grad(v[1], mv[k][1], mask_act); }
      if (ok[k]) {
        if (atomic_out) { atomicAdd(dst + di[k], v[0]); atomicAdd(dst + di[k] + 1, v[1]); }
        else *reinterpret_cast<f32x2e*>(dst + di[k]) = v;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

static inline void pick_tile_dims(int NPIX, int B, int nT, int nH, int nW, int& jb, int& jt, int& jh, int& jw) {
  jw = pow2_ceil(nW); if (jw > 32) jw = 32;
  int rem = NPIX / jw;
  jh = pow2_ceil(nH); if (jh > rem) jh = rem;
  rem /= jh;
  jt = pow2_ceil(nT); if (jt > rem) jt = rem;
  rem /= jt;
  jb = rem;
  (void)B;
}


static inline int check_desc(const p2i_conv_desc* d) {
  P2I_REQUIRE(d != nullptr, "null conv desc");
  P2I_REQUIRE(d->B > 0 && d->Cin > 0 && d->Cout > 0, "bad channel/batch dims");
  P2I_REQUIRE(d->kt >= 1 && d->kh >= 1 && d->kw >= 1 && d->kt * d->kh * d->kw <= MAX_TAPS, "kernel taps > %d", MAX_TAPS);
  P2I_REQUIRE(d->st >= 1 && d->sh >= 1 && d->sw >= 1, "bad stride");
  P2I_REQUIRE(d->To == (d->Ti + 2 * d->pt - d->kt) / d->st + 1, "To inconsistent");
  P2I_REQUIRE(d->Ho == (d->Hi + 2 * d->ph - d->kh) / d->sh + 1, "Ho inconsistent");
  P2I_REQUIRE(d->Wo == (d->Wi + 2 * d->pw - d->kw) / d->sw + 1, "Wo inconsistent");
  const long long nin = (long long)d->B * d->Cin * d->Ti * d->Hi * d->Wi;
  const long long nout = (long long)d->B * d->Cout * d->To * d->Ho * d->Wo;
  P2I_REQUIRE(nin < (1ll << 31) && nout < (1ll << 31), "tensor too large for 32-bit indexing");
  return P2I_OK;
}


// per-class fields of a merged multi-class launch (strided dgrad: blockIdx.z = input-parity class)
struct ClassGeom {
  int nT, nH, nW, pT, pH, pW, bT, bH, bW, ntaps;
  short tap_w[MAX_TAPS];
  int tap_off[MAX_TAPS];
};
constexpr int MAX_CLASSES = 8;

struct PatchGeom {
  const float* src;
  const float* src_y;   // dgrad: saved activation output for act'(y) (may be null)
  const float* wp;      // packed weights [tap][Ck][CmPad]
  const float* bias;    // fwd epilogue (may be null)
  const float* res;     // fwd epilogue residual, dest-shaped (may be null)
  float* dst;
  int act_epi;          // epilogue activation (forward)
  int act_pro;          // prologue act'(y) code (dgrad)
  int B, Ck, Cm, CmPad;
  int sT, sH, sW;       // source tensor dims
  int dT, dH, dW;       // dest tensor dims
  int nT, nH, nW;       // dest-local extents
  int mT, mH, mW;       // source multiplier S
  int oT, oH, oW;       // dest index multiplier
  int pT, pH, pW;       // dest index offset
  int bT, bH, bW;       // min tap delta (patch origin = j0*S + b)
  int eT, eH, eW, eWp;  // patch extents (eWp = row pitch)
  int ljb, ljt, ljh, ljw;
  int ntt, nth, ntw;    // tiles per dim (batch tiles = gridDim.x / (ntt*nth*ntw))
  int ntaps;
  int CS;               // patch channel stride (floats)
  int rpc, eth;         // rows per channel = JB*eT*eH ; eT*eH
  unsigned mg_rpc, mg_eth, mg_eh;
  short tap_w[MAX_TAPS];
  int tap_off[MAX_TAPS];
  // ---- DMA-pipelined variant only
  const float* mask_y;  // epilogue: dst *= act'(mask_y) (dest-shaped; the activation that PRODUCED dst's tensor)
  int mask_act;
  int CSl;              // linear (unpadded) patch channel stride = rpc * eW
  int PT;               // patch dwords per chunk = CK * CSl
  unsigned src_bytes, wp_bytes;
  unsigned mg_csl, mg_ew;
  int v4, v4sh;         // DMA variant: 16-B patch DMA (rows start v4sh columns left of the tap window, 16-B aligned)
  int eW4, G4;          // row pitch and channel stride in 4-pixel groups
  unsigned mg_g4, mg_ew4;
  // ---- x6 (bf16-split) variant only
  const uint16_t* wb;   // split weights [plane 3][ntaps_w][Ck/8][CmPad][8] bf16
  unsigned wb_bytes;
  int ntaps_w;          // taps of the full kernel (plane stride of wb)
  int TG, NTP, R;       // taps per pipeline stage; padded tap count of the LDS weight-offset table; weight ring slots
  int ksplit;           // x6c: > 1 = blockIdx.z takes 1/ksplit of the channel chunks and ADDS its partial sums to a zeroed dst
  int pair_w;           // fused strided dgrad: classes 2k / 2k+1 interleave along w and are stored together as float2
  int nclass;           // > 1: fields below override nT..ntaps / tap tables per blockIdx.z
  ClassGeom cls[MAX_CLASSES];
};

// one (class of a) patch GEMM.  taps: arrays of weight-tap index and per-dim source delta.
struct ClassSpec {
  int nT, nH, nW;      // dest-local extents
  int mT, mH, mW;      // source multiplier
  int oT, oH, oW, pT, pH, pW;
  int ntaps;
  short tw[MAX_TAPS];
  int dt[MAX_TAPS], dh[MAX_TAPS], dw[MAX_TAPS];
};

// exact-fp32 convolution on the bf16 matrix pipe (conv_x6c.hip): set for the duration of a p2i_conv_*_x6 call,
// points at the 3-plane bf16 split of the packed weights the call is about to use (nullptr: fp32 MFMA path)
struct X6Ctx { const uint16_t* wb; int ntaps_w; };
X6Ctx& x6_ctx();

// 3x3 stride-1 2-D layers on the bf16 matrix pipe, chunk/tap-row pipeline (conv_x6c.hip); returns 1 when it does not apply
int run_patch_gemm_x6c(PatchGeom g, const ClassSpec& cs, const uint16_t* wb, int ntaps_w, int* plan6, hipStream_t s, bool dry = false);
// data gradient of a 3x3(x3) stride-(1,2,2) pad-1 convolution: the four input-parity classes in one workgroup, bf16 matrix pipe
int run_patch_gemm_x6c_fused(PatchGeom g, const ClassSpec* css, int ncls, const uint16_t* wb, int ntaps_w, int* plan6, hipStream_t s, bool dry = false);
bool x6c_would_take(const p2i_conv_desc* d, bool dgrad, int act_epi);

// strided dgrad with the parity classes fused in one workgroup (conv_fused.hip); returns 1 when it does not apply
int run_patch_gemm_fused(PatchGeom g, const ClassSpec* css, int ncls, int* plan6, hipStream_t s);

// single-input-channel special cases (conv_c1.hip)
int c1_dgrad(const p2i_conv_desc* d, const float* dy, const float* wp_d, const float* add, const float* mask_y, int mask_act,
             float* dx, hipStream_t s);
int c1_wgrad(const p2i_conv_desc* d, const float* x, const float* dy, float* dwp, float* dbias, hipStream_t s);
// forward of the same layer; returns 1 when the shape is not the (3x3x3, stride (1,2,2), pad 1) one
int c1_fwd(const p2i_conv_desc* d, const float* x, const float* wp, const float* bias, float* y, int act, hipStream_t s);
// single-output-channel 3x3 stride-1 2-D forward (bandwidth-bound; conv_c1.hip); returns 1 when the shape is not that one
int o1_fwd(const p2i_conv_desc* d, const float* x, const float* wp, const float* bias, float* y, int act, hipStream_t s);

}  // namespace p2i
