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


// single-input-channel special cases (conv_c1.hip)
int c1_dgrad(const p2i_conv_desc* d, const float* dy, const float* wp_d, const float* add, const float* mask_y, int mask_act,
             float* dx, hipStream_t s);
int c1_wgrad(const p2i_conv_desc* d, const float* x, const float* dy, float* dwp, float* dbias, hipStream_t s);

}  // namespace p2i
