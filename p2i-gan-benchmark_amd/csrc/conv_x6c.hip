// "x6c": exact-fp32 3x3 stride-1 convolution (forward and data gradient of the generator's DO-Conv stack, 2-D) on the bf16 matrix
// pipe.  Successor of round 1's conv_x6.hip (removed: slower than the f32 engine on every layer, profiles/README.md "x6 decision"):
//   * numerics: x = hi + mid + lo (exact 3-way bf16 truncation split of every fp32 operand), six v_mfma_f32_32x32x16_bf16
//     products (lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi) accumulated in fp32; the three dropped products are <= 2^-23 |a b|,
//     i.e. below fp32 rounding of the term: results agree with the f32-MFMA engine to fp32 rounding (tests/test_ops_gpu.py);
//   * weights arrive pre-split (wsplit_kernel below) as Wb[plane][tap][k/8][m][8] bf16: one 16-B LDS-DMA element per (m, 8 k's);
//   * 8 waves (two per SIMD), one 64 x 256 output tile per workgroup, wave = 64 channels x 32 positions (TM 2, TN 1);
//   * a pipeline STAGE is one kernel row (3 taps) of one 16-channel chunk: its pre-split weights (18 KB) are LDS-DMA'd three
//     stages ahead into a 4-slot ring; the next chunk's fp32 patch is loaded straight into REGISTERS during the chunk's first
//     stage (no staging area, no DMA issue), split once per element and written as three bf16 operand planes
//     [plane][k/8][patch pixel][8 k] into the other plane buffer during the last stage;
//   * every MFMA operand is one ds_read_b128 at a per-tile base + tap offset; the tap loop is fully unrolled and runs as one
//     software pipeline across stages: tap t+1's nine operand reads are issued under tap t's twelve MFMAs;
//   * every vector-memory operation of the loop (weight DMAs and patch loads) is inline asm behind counted s_waitcnt vmcnt(n).
#include "conv_common.h"

namespace p2i {

#ifdef P2I_STAMP
// diagnostic build (tools/build_stamp.sh): per-wave cycle sums of the loop's phases; never defined in the product build
extern __device__ unsigned long long* p2i_stamp_buf;
#define X6C_NOW(v) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define X6C_ACC(sum, t_prev) do { unsigned long long t_; X6C_NOW(t_); sum += t_ - t_prev; t_prev = t_; } while (0)
#else
#define X6C_NOW(v) do { } while (0)
#define X6C_ACC(sum, t_prev) do { } while (0)
#endif

typedef __bf16 bf16x8c __attribute__((ext_vector_type(8)));
typedef unsigned u32x4c __attribute__((ext_vector_type(4)));

// (lo_elem >> 16) | (hi_elem & 0xFFFF0000) in one v_perm_b32: bytes {lo.2, lo.3, hi.2, hi.3}
__device__ __forceinline__ unsigned pack_hi16c(float lo_elem, float hi_elem) {
  return __builtin_amdgcn_perm(__float_as_uint(hi_elem), __float_as_uint(lo_elem), 0x07060302u);
}
__device__ __forceinline__ float trunc_bf16c(float v) { return __uint_as_float(__float_as_uint(v) & 0xFFFF0000u); }
__device__ __forceinline__ void split8c(const float (&v)[8], u32x4c& hi, u32x4c& mid, u32x4c& lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = v[2 * j], b = v[2 * j + 1];
    hi[j] = pack_hi16c(a, b);
    const float ra = a - trunc_bf16c(a), rb = b - trunc_bf16c(b);          // exact
    mid[j] = pack_hi16c(ra, rb);
    const float sa = ra - trunc_bf16c(ra), sb = rb - trunc_bf16c(rb);      // exact
    lo[j] = pack_hi16c(sa, sb);
  }
}

// packed fp32 weights wp[tap][k][CmPad] -> Wb[plane][tap][k/8][CmPad][8] bf16 (planes hi, mid, lo)
__global__ __launch_bounds__(256) void wsplit_kernel(const float* __restrict__ wp, u32x4c* __restrict__ wb, int ntaps, int Ck, int CmPad) {
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
    u32x4c h, mi, lo;
    split8c(v, h, mi, lo);
    wb[i] = h;
    wb[total + i] = mi;
    wb[2 * total + i] = lo;
  }
}

// wsplit_kernel for up to 24 packed tensors of DIFFERENT shapes in one launch (the discriminators' layers after a weight pack:
// blockIdx.y = tensor); same images as x6_split_weights
struct WsplitBatch { const float* wp[24]; u32x4c* wb[24]; int ntaps[24], Ck[24], CmPad[24]; };
__global__ __launch_bounds__(256) void wsplit_batched_kernel(const WsplitBatch b) {
  const int L = blockIdx.y;
  const float* wp = b.wp[L];
  u32x4c* wb = b.wb[L];
  const int Ck = b.Ck[L], CmPad = b.CmPad[L], KCt = Ck >> 3;
  const int total = b.ntaps[L] * KCt * CmPad;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int m = i % CmPad;
    const int r = i / CmPad;
    const int kc = r % KCt, tap = r / KCt;
    const float* p = wp + ((size_t)tap * Ck + kc * 8) * CmPad + m;
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = p[(size_t)q * CmPad];
    u32x4c h, mi, lo;
    split8c(v, h, mi, lo);
    wb[i] = h;
    wb[total + i] = mi;
    wb[2 * total + i] = lo;
  }
}

// fp32 activations (B, C, P) (P = frames x pixels, C % 8 == 0) -> bf16 planes [plane][b][k/8][P][8]: the exact 3-way split of
// split8c, made ONCE per tensor instead of once per (channel tile, position tile) of every kernel that consumes it
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, u32x4c* __restrict__ planes, int B, int KCt, long long P) {
  const long long total = (long long)B * KCt * P;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long pix = i % P, r = i / P;          // r = b * KCt + kg
    const float* px = x + (r * 8) * P + pix;
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = px[(long long)q * P];
    u32x4c h, mi, lo;
    split8c(v, h, mi, lo);
    planes[i] = h;
    planes[total + i] = mi;
    planes[2 * total + i] = lo;
  }
}

X6Ctx& x6_ctx() {
  static thread_local X6Ctx c{nullptr, 0};
  return c;
}

int x6_split_weights(const float* wp, uint16_t* wb, int ntaps, int Ck, int CmPad, hipStream_t s) {
  const int total = ntaps * (Ck >> 3) * CmPad;
  const int blocks = total > 0 ? (total + 255) / 256 : 1;
  P2I_LAUNCH(wsplit_kernel, dim3(blocks > 2048 ? 2048 : blocks), dim3(256), 0, s, wp, reinterpret_cast<u32x4c*>(wb), ntaps, Ck, CmPad);
  return launch_status();
}

// Kernel argument of patch_gemm_x6c_kernel: only what this kernel reads (~300 bytes).  The engine-wide PatchGeom is 2 KB (27-tap
// tables for four classes); passed by value it stayed in SGPRs only as long as the kernel was simple -- see DESIGN.md.
struct X6cGeom {
  const float* src; float* dst; const uint16_t* wb; const float* bias; const float* res; const float* mask_y;
  unsigned wb_bytes;
  int act_epi, mask_act;
  int B, Ck, Cm, CmPad;
  int sH, sW, dT, dH, dW, nH, nW;
  int ljb, ljh, ljw, eH, eW, CSl, bH, bW, nth, ntw;
  unsigned mg_ew, mg_eh;
  int ntaps_w, ksplit;
  // time axis: 'images' of the tile loop are (b, lt) pairs, lt in [0, nT); the taps form ns slices of 3x3 spatial taps, slice j reads
  // source frame lt * mT + sdt[j] (zero outside [0, sT)) and the weight taps swt[j] + tap_w[0..8]; destination frame lt * oT + pT
  int sT, nT, mT, oT, pT, ns, sdt0, sdt1, sdt2, swt0, swt1, swt2;
  int tap_off[9];
  int tap_w[9];
  int fused_atomic;     // FUSED kernels with split-K: the class pairs are ADDED to a zeroed destination
  int stagger;          // waves 4-7 (the SIMD partners of waves 0-3) run a stage's staging work AFTER its first taps instead of before
  int vec4_epi;         // epilogue through LDS with 16-byte global accesses (epilogue_tile16_v4): host-checked alignment / extents
  // PRE kernels (round 4 experiment): the source as pre-split bf16 planes [plane][b][k/8][frame, pixel][8] (split_planes_kernel):
  // 16-byte units; plane p starts plane_units units behind plane p - 1
  const u32x4c* splanes;
  unsigned plane_units;
  int cons_prio;        // x6p: the consumer waves raise their issue priority for the tap loop (s_setprio; P2I_X6P_PRIO, A/B)
};

// FUSED (data gradient of a stride-(.,2,2) 3x3 convolution: conv_fused.hip on the bf16 matrix pipe): the nine taps of a chunk belong
// to the FOUR input-parity classes (pH, pW) of the destination -- slot s feeds accumulator class X6C_CLS[s] = 2 pH + pW -- and all
// read ONE (jh+1) x (jw+1) patch of dy; the epilogue stores the pW pairs as float2 (rows 2 gh + pH, columns 2 gw, 2 gw + 1).
__device__ constexpr int X6C_CLS[9] = {3, 3, 3, 3, 2, 2, 1, 1, 0};

// Tile variants <NW waves, TM 32-channel tiles per wave>: a workgroup computes (32 TM) channels x (32 NW) positions, wave = 32 TM
// channels x 32 positions.  <8,2> (64 x 256) is the efficient one (0.75 operand reads per MFMA); <8,1> (32 x 256) gives the
// 256-channel level (32x32 maps, 128 tiles of 64 x 256 at B = 8) one workgroup per CU: 72 vs 103 us (<8,2>) vs 89 us (f32 engine).
// Four-wave variants (128 positions, one wave per SIMD) were measured and dropped: 112-230 us on the same layers.
template <int NW> struct X6cTile {
  static constexpr int MAXCSL = NW == 8 ? 384 : 256;     // patch pixels per tile (16x16 + halo = 324, 8x32 + halo = 340; 4x32: 204, 8x16: 180)
  static constexpr int NI = (2 * MAXCSL + 64 * NW - 1) / (64 * NW);
};

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the immediate must be a constant): a balanced tree of scalar compares, six
// deep.  (A `switch` over the 61 cases compiled to a LINEAR chain of compare + branch: ~3 n scalar instructions per wait, two or
// three waits per stage -- a third of a producer wave's instructions, tools: histogram of the loop in the .s.)  n >= 63: vmcnt(0).
__device__ __forceinline__ void x6c_wait_vm(int n) {
  if (n < 0 || n > 62) n = 0;
  if (n < 32) {
   if (n < 16) {
    if (n < 8) {
     if (n < 4) {
      if (n < 2) {
       if (n < 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
       } else {
        asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
       }
      } else {
       if (n < 3) {
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
       } else {
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
       }
      }
     } else {
      if (n < 6) {
       if (n < 5) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
       } else {
        asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
       }
      } else {
       if (n < 7) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
       } else {
        asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
       }
      }
     }
    } else {
     if (n < 12) {
      if (n < 10) {
       if (n < 9) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
       } else {
        asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
       }
      } else {
       if (n < 11) {
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
       } else {
        asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
       }
      }
     } else {
      if (n < 14) {
       if (n < 13) {
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
       } else {
        asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
       }
      } else {
       if (n < 15) {
        asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
       } else {
        asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
       }
      }
     }
    }
   } else {
    if (n < 24) {
     if (n < 20) {
      if (n < 18) {
       if (n < 17) {
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
       } else {
        asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
       }
      } else {
       if (n < 19) {
        asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
       } else {
        asm volatile("s_waitcnt vmcnt(19)" ::: "memory");
       }
      }
     } else {
      if (n < 22) {
       if (n < 21) {
        asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
       } else {
        asm volatile("s_waitcnt vmcnt(21)" ::: "memory");
       }
      } else {
       if (n < 23) {
        asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
       } else {
        asm volatile("s_waitcnt vmcnt(23)" ::: "memory");
       }
      }
     }
    } else {
     if (n < 28) {
      if (n < 26) {
       if (n < 25) {
        asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
       } else {
        asm volatile("s_waitcnt vmcnt(25)" ::: "memory");
       }
      } else {
       if (n < 27) {
        asm volatile("s_waitcnt vmcnt(26)" ::: "memory");
       } else {
        asm volatile("s_waitcnt vmcnt(27)" ::: "memory");
       }
      }
     } else {
      if (n < 30) {
       if (n < 29) {
        asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
       } else {
        asm volatile("s_waitcnt vmcnt(29)" ::: "memory");
       }
      } else {
       if (n < 31) {
        asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
       } else {
        asm volatile("s_waitcnt vmcnt(31)" ::: "memory");
       }
      }
     }
    }
   }
  } else {
   if (n < 48) {
    if (n < 40) {
     if (n < 36) {
      if (n < 34) {
       if (n < 33) {
        asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
       } else {
        asm volatile("s_waitcnt vmcnt(33)" ::: "memory");
       }
      } else {
       if (n < 35) {
        asm volatile("s_waitcnt vmcnt(34)" ::: "memory");
       } else {
        asm volatile("s_waitcnt vmcnt(35)" ::: "memory");
       }
      }
     } else {
      if (n < 38) {
       if (n < 37) {
        asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
       } else {
        asm volatile("s_waitcnt vmcnt(37)" ::: "memory");
       }
      } else {
       if (n < 39) {
        asm volatile("s_waitcnt vmcnt(38)" ::: "memory");
       } else {
        asm volatile("s_waitcnt vmcnt(39)" ::: "memory");
       }
      }
     }
    } else {
     if (n < 44) {
      if (n < 42) {
       if (n < 41) {
        asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
       } else {
        asm volatile("s_waitcnt vmcnt(41)" ::: "memory");
       }
      } else {
       if (n < 43) {
        asm volatile("s_waitcnt vmcnt(42)" ::: "memory");
       } else {
        asm volatile("s_waitcnt vmcnt(43)" ::: "memory");
       }
      }
     } else {
      if (n < 46) {
       if (n < 45) {
        asm volatile("s_waitcnt vmcnt(44)" ::: "memory");
       } else {
        asm volatile("s_waitcnt vmcnt(45)" ::: "memory");
       }
      } else {
       if (n < 47) {
        asm volatile("s_waitcnt vmcnt(46)" ::: "memory");
       } else {
        asm volatile("s_waitcnt vmcnt(47)" ::: "memory");
       }
      }
     }
    }
   } else {
    if (n < 56) {
     if (n < 52) {
      if (n < 50) {
       if (n < 49) {
        asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
       } else {
        asm volatile("s_waitcnt vmcnt(49)" ::: "memory");
       }
      } else {
       if (n < 51) {
        asm volatile("s_waitcnt vmcnt(50)" ::: "memory");
       } else {
        asm volatile("s_waitcnt vmcnt(51)" ::: "memory");
       }
      }
     } else {
      if (n < 54) {
       if (n < 53) {
        asm volatile("s_waitcnt vmcnt(52)" ::: "memory");
       } else {
        asm volatile("s_waitcnt vmcnt(53)" ::: "memory");
       }
      } else {
       if (n < 55) {
        asm volatile("s_waitcnt vmcnt(54)" ::: "memory");
       } else {
        asm volatile("s_waitcnt vmcnt(55)" ::: "memory");
       }
      }
     }
    } else {
     if (n < 60) {
      if (n < 58) {
       if (n < 57) {
        asm volatile("s_waitcnt vmcnt(56)" ::: "memory");
       } else {
        asm volatile("s_waitcnt vmcnt(57)" ::: "memory");
       }
      } else {
       if (n < 59) {
        asm volatile("s_waitcnt vmcnt(58)" ::: "memory");
       } else {
        asm volatile("s_waitcnt vmcnt(59)" ::: "memory");
       }
      }
     } else {
      if (n < 62) {
       if (n < 61) {
        asm volatile("s_waitcnt vmcnt(60)" ::: "memory");
       } else {
        asm volatile("s_waitcnt vmcnt(61)" ::: "memory");
       }
      } else {
       if (n < 63) {
        asm volatile("s_waitcnt vmcnt(62)" ::: "memory");
       } else {
        asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
       }
      }
     }
    }
   }
  }
}

// TPS = taps per pipeline stage (one hand-over barrier per stage): 3 = one kernel row, 9 = the whole 16-channel chunk.  A 32-channel
// tile (TM 1) has only 18 MFMAs per wave and kernel-row stage, against which the barrier, the counted waits and the DMA issue of a
// stage weigh twice as much as in the 64-channel tile (stamped: 42 % vs 63 % of the MFMA-bound time in the loop); with whole-chunk
// stages those costs are paid once per 54 MFMAs and the two waves of a SIMD drift apart inside the stage (one's staging work under
// the other's MFMAs).  The weight ring is then 3 slots of 27 KB (TM 1), filled two chunks ahead.
template <int NW, int TM, bool FUSED = false, int TPS = 3>
__global__ __launch_bounds__(64 * NW) void patch_gemm_x6c_kernel(const X6cGeom g) {
  constexpr int NCLS = FUSED ? 4 : 1;
  constexpr int NSTG = 9 / TPS;                           // stages per chunk
  static_assert(TPS == 3 || TPS == 9, "taps per stage");
  extern __shared__ __attribute__((aligned(16))) float smem[];
#ifdef P2I_STAMP
  unsigned long long st_entry, st_prev, st_wait = 0, st_bar = 0, st_issue = 0, st_mfma = 0, st_loop0;
  X6C_NOW(st_entry);
#endif
  constexpr int MB = 32 * TM, NTHR = 64 * NW, NI = X6cTile<NW>::NI;
  constexpr int RING = TPS == 9 ? 3 : (TM == 2 ? 4 : 6), LEAD = RING - 1;  // a kernel-row stage is ~1.2 us (TM 2) / ~0.6 us (TM 1) of MFMAs, an L2 -> LDS DMA ~2 us under load
  constexpr int WST = 6 * TPS * MB;                       // 16-B elements of one stage's weights: [3 planes][TPS taps][2 k-groups][MB m]
  constexpr int NWI = WST / 64;                           // ... = this many 64-lane DMA instructions (TM 1: one covers both k-groups of 32 m)
  const int CSl = g.CSl;
  int* ptab = reinterpret_cast<int*>(smem);                               // [CSl] element offset of patch pixel e, channel 0 (-1: outside)
  const int ptab_sz = (CSl + 3) & ~3;
  u32x4c* wbuf = reinterpret_cast<u32x4c*>(smem + ptab_sz);              // [RING][WST]: ring, filled LEAD stages ahead
  u32x4c* planes = wbuf + RING * WST;                                     // [2][3 planes][2 k-groups][CSl]
  const int PST = 6 * CSl;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  int tile = blockIdx.x;
  const int tw = tile % g.ntw; tile /= g.ntw;
  const int th = tile % g.nth;
  const int tb = tile / g.nth;
  const int j0b = tb << g.ljb, j0h = th << g.ljh, j0w = tw << g.ljw;
  const int o0 = blockIdx.y * MB;
  const int JWm = (1 << g.ljw) - 1, JHm = (1 << g.ljh) - 1;
  const int sHW = g.sH * g.sW;
  const int src_h0 = j0h + g.bH, src_w0 = j0w + g.bW;
  const int KCt = g.Ck >> 3;
  const int cps = g.Ck >> 4;                               // 16-channel chunks per tap slice
  const int cs = g.sT * sHW;                               // source channel stride
  const int nimg = g.B * g.nT;
  const int c0 = blockIdx.z * ((g.ns * cps) / g.ksplit);   // split-K: first chunk of this workgroup (chunks run slice-major)

  for (int e = tid; e < CSl; e += NTHR) {
    const int row = fast_div(e, g.mg_ew);
    const int ew = e - row * g.eW;
    const int jb = fast_div(row, g.mg_eh);
    const int eh = row - jb * g.eH;
    const int n = j0b + jb, h = src_h0 + eh, w = src_w0 + ew;          // image n = (b, lt)
    const int b = n / g.nT, lt = n - b * g.nT;
    ptab[e] = (n < nimg && (unsigned)h < (unsigned)g.sH && (unsigned)w < (unsigned)g.sW) ? ((b * g.Ck) * g.sT + lt * g.mT) * sHW + h * g.sW + w : -1;
  }
  // this wave's 32 output positions
  const int pix = wave * 32 + l31;
  const int pjw = pix & JWm, pjh = (pix >> g.ljw) & JHm, pjb = pix >> (g.ljw + g.ljh);
  const int lane_base = (pjb * g.eH + pjh) * g.eW + pjw + lhi * CSl;
  int toff[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) toff[t] = g.tap_off[t];

  f32x16 acc[NCLS][TM];
#pragma unroll
  for (int q = 0; q < NCLS; ++q)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][i][r] = 0.f;

  const v4i32 rs_w = make_rsrc(g.wb, g.wb_bytes);
  const unsigned wbuf_la = lds_base(smem) + 4u * ptab_sz;
  // lane -> (k-group, m) of a weight DMA instruction: TM 2: 64 m of one k-group; TM 1: 32 m of both k-groups
  const int wl_m = TM == 2 ? lane : l31, wl_kg = TM == 2 ? 0 : lhi;
  const int wvoff = (o0 + wl_m < g.CmPad) ? (wl_kg * g.CmPad + wl_m) * 16 : -16;
  __syncthreads();                                                        // ptab visible

  // patch items of this thread: (k-group kg, patch pixel e); 8 channel values each
  int it_off[NI], it_dst[NI], it_f0[NI];
#pragma unroll
  for (int it = 0; it < NI; ++it) {
    const int item = it * NTHR + tid;
    const int kg = item >= CSl ? 1 : 0, e = item - kg * CSl;
    const bool in = item < 2 * CSl;
    const int po = in ? ptab[e] : -1;
    it_off[it] = po < 0 ? -1 : po + kg * 8 * cs;
    it_dst[it] = in ? kg * CSl + e : -1;
    const int jbi = fast_div(fast_div(in ? e : 0, g.mg_ew), g.mg_eh);
    it_f0[it] = ((j0b + jbi) % g.nT) * g.mT;                       // source frame at slice offset 0
  }
  // The patch loads are issued from inline asm like the DMAs, so that EVERY vector-memory operation of the loop is counted by
  // hand: next to asm DMAs hipcc would wait vmcnt(0) before the first use of a plain load's result and drain the weight ring.
  // pv is not touched between load_patch and the counted wait in front of split_patch.
  float pv[NI][8];
  bool pok[NI];                                                   // the loaded chunk's slice frame exists for this item
  auto load_patch = [&](int c) {
    const int cg = c0 + c;
    const int j = cg >= 2 * cps ? 2 : (cg >= cps ? 1 : 0);            // tap slice (wave-uniform)
    const int dt = j == 2 ? g.sdt2 : (j == 1 ? g.sdt1 : g.sdt0);
    const float* sc = g.src + (size_t)(cg - j * cps) * 16 * cs;
    int voff[NI];
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      pok[it] = it_off[it] >= 0 && (unsigned)(it_f0[it] + dt) < (unsigned)g.sT;
      voff[it] = pok[it] ? (it_off[it] + dt * sHW) * 4 : 0;          // non-negative for valid items; idle items read element 0
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float* sq = sc + (size_t)q * cs;
      const unsigned long long sq_u = (unsigned long long)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)sq) & 0xffffffffull;
      const unsigned long long sq_hi = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)sq >> 32));
      const unsigned long long sbase = sq_u | (sq_hi << 32);
      // sbase is fresh from v_readfirstlane: a VMEM instruction that reads an SGPR written by the VALU needs 5 wait states, and hipcc
      // pads nothing for an asm statement (a build whose schedule put the two back to back faulted on a garbage base)
#pragma unroll
      for (int it = 0; it < NI; ++it) {
        if (it == 0) asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=v"(pv[it][q]) : "v"(voff[it]), "s"(sbase) : "memory");
        else asm volatile("global_load_dword %0, %1, %2" : "=v"(pv[it][q]) : "v"(voff[it]), "s"(sbase) : "memory");
      }
    }
  };
  auto split_patch = [&](u32x4c* pb) {
    // The loaded values become visible to the compiler HERE (the counted wait is in front of every call): an empty asm redefines
    // each register, so that nothing computed from pv can be hoisted above the wait.  Needed since the split pass exists at several
    // sites of the loop (staggered halves): hipcc merged the sites' common subexpressions and evaluated them right behind the
    // asm loads, before the data had landed (wrong results in the first staggered build).
#pragma unroll
    for (int it = 0; it < NI; ++it)
#pragma unroll
      for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(pv[it][q]));
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      if (it_dst[it] < 0) continue;
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = pok[it] ? pv[it][q] : 0.f;
      u32x4c h, mi, lo;
      split8c(v, h, mi, lo);
      pb[it_dst[it]] = h;
      pb[2 * CSl + it_dst[it]] = mi;
      pb[4 * CSl + it_dst[it]] = lo;
    }
  };
  // weights of stage (chunk c, kernel row b) -> wbuf[sb]: NWI wave-instructions of 64 x 16 B, dealt round-robin to the waves.
  // The byte offset of instruction r's source at chunk 0 is computed ONCE per kernel row (w_soff[r][b]): inside the loop the
  // look-up of g.tap_w[] is a scalar load + s_waitcnt lgkmcnt(0) per instruction, 1300-1500 cycles per stage (stamped).
  constexpr int NWR = (NWI + NW - 1) / NW;
  int w_soff[NWR][NSTG];
  unsigned w_dst[NWR];
#pragma unroll
  for (int r = 0; r < NWR; ++r) {
    const int u = wave + NW * r;
    // TM 2: u = (plane, tap, k-group); TM 1: u = (plane, tap), the two k-groups ride in the lane halves
    const int pt = TM == 2 ? u >> 1 : u, kg = TM == 2 ? u & 1 : 0;
    const int p = pt / TPS, tl = pt % TPS;
#pragma unroll
    for (int b = 0; b < NSTG; ++b)
      w_soff[r][b] = u < NWI ? __builtin_amdgcn_readfirstlane((((p * g.ntaps_w + g.tap_w[TPS * b + tl]) * KCt + kg) * g.CmPad + o0) * 16) : 0;
    w_dst[r] = wbuf_la + 16u * (unsigned)(u * 64);
  }
  const int w_cstep = 2 * g.CmPad * 16;                                // one 16-channel chunk further
  const int w_tstep = KCt * g.CmPad * 16;                              // one weight tap further
  auto issue_w = [&](int c, int b, int sb) {
    const int cg = c0 + c;
    const int j = cg >= 2 * cps ? 2 : (cg >= cps ? 1 : 0);
    const int soff = (cg - j * cps) * w_cstep + (j == 2 ? g.swt2 : (j == 1 ? g.swt1 : g.swt0)) * w_tstep;
#pragma unroll
    for (int r = 0; r < NWR; ++r)
      if (wave + NW * r < NWI) dma_b128(rs_w, w_dst[r] + 16u * (unsigned)(sb * WST), wvoff, w_soff[r][b] + soff);
  };

  const int nch = (g.ns * cps) / g.ksplit;            // this workgroup's chunks: [c0, c0 + nch)
  const int nst = NSTG * nch;
  const int nw_mine = (NWI / NW) + (wave < NWI % NW ? 1 : 0);          // this wave's DMA instructions per weight batch
  constexpr int NPL = 8 * NI;                                           // patch loads per thread and chunk
  auto nwb = [&](int st) { return st < nst ? nw_mine : 0; };
  // Vector-memory operations of this wave, in issue order: prologue [P(0)] [W(0)] .. [W(LEAD-1)], then per stage s, all at its
  // start, [P(chunk+1) if s % 3 == 0 and there is a next chunk] [W(s+LEAD)] (P = NPL patch loads, W = nw_mine weight DMAs).
  // prologue
  load_patch(0);
#pragma unroll
  for (int st = 0; st < LEAD; ++st)
    if (st < nst) issue_w(st / NSTG, st % NSTG, st % RING);
  {
    int n = 0;
#pragma unroll
    for (int st = 0; st < LEAD; ++st) n += nwb(st);
    x6c_wait_vm(n);                                                     // the patch loads (older than the weight batches) have landed
    __builtin_amdgcn_sched_barrier(0);
    split_patch(planes);
    x6c_wait_vm(n - nwb(0));                                            // weights of stage 0 too
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  constexpr int PA[6] = {2, 0, 1, 1, 0, 0};            // small terms first: lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi
  constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
  // One continuous software pipeline over all taps: tap t+1's operand reads are in flight under tap t's MFMAs, ACROSS stage
  // boundaries too.  The hand-over barrier of stage s+1 (its weights landed, the next chunk's planes written) therefore sits in
  // front of the LAST tap of stage s, and that tap's MFMAs cover the first reads of stage s+1.  Three operand register sets.
  u32x4c A[3][TM][3], Bv[3][3];
  auto load_tap = [&](const u32x4c* wsl, const u32x4c* pbp, int tap, int buf) {
    const int tl = tap % TPS;
    const int to = toff[tap];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int i = 0; i < TM; ++i) A[buf][i][p] = wsl[((p * TPS + tl) * 2) * MB + 32 * i];
      Bv[buf][p] = pbp[p * 2 * CSl + to];
    }
  };
  auto mfma_tap = [&](int buf, int slot) {              // slot = tap index inside the chunk (a constant after unrolling)
    const int cl = FUSED ? X6C_CLS[slot] : 0;
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int i = 0; i < TM; ++i)
        acc[cl][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8c, A[buf][i][PA[q]]), __builtin_bit_cast(bf16x8c, Bv[buf][PB[q]]),
                                                             acc[cl][i], 0, 0, 0);
  };
  auto ilv = [&]() {                                   // a tap's MFMAs with the next tap's operand reads in their gaps
    constexpr int NR = 3 * TM + 3, NM = 6 * TM;
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (i < NR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
  };
  const u32x4c* wlane = wbuf + lhi * MB + l31;
  load_tap(wlane, planes + lane_base, 0, 0);
#ifdef P2I_STAMP
  X6C_NOW(st_loop0); st_prev = st_loop0;
#endif
  const bool late = g.stagger > 0 && wave >= NW / 2;     // wave-uniform (wave is a readfirstlane value)
  const int split_t = g.stagger > 0 ? (late ? 7 : 4) : 6; // whole-chunk stages: tap in front of which this wave runs its split pass
  int s = 0;
  for (int c = 0; c < nch; ++c) {
    const u32x4c* pb = planes + (c & 1) * PST + lane_base;
    const u32x4c* pbn = planes + ((c + 1) & 1) * PST + lane_base;
    const bool more_c = c + 1 < nch;
#pragma unroll
    for (int b = 0; b < NSTG; ++b, ++s) {
      const int sa = s + LEAD;
      // Staging work of the stage (next chunk's patch loads, the split pass, the weight DMAs of stage s+LEAD).  The two waves of a SIMD
      // (w and w + 4) run the same program between the same barriers: in lockstep both stage first (matrix pipe idle) and then
      // both multiply (VALU idle).  With `stagger` waves 4-7 do their staging work AFTER the stage's first taps, so one wave's
      // VALU / VMEM / LDS-write work runs under its partner's MFMAs.  The ORDER of this wave's vector-memory operations and of its
      // counted waits is the same in both placements, so the vmcnt arithmetic below holds for both.
      auto staging = [&]() {
        if (b == 0 && more_c) load_patch(c + 1);
        if constexpr (TPS == 3) {
          if (b == 2 && more_c) {
            // patch loads were issued in stage s-2; younger: the weights issued in stages s-2 and s-1
            x6c_wait_vm(nwb(s - 2 + LEAD) + nwb(s - 1 + LEAD));
            __builtin_amdgcn_sched_barrier(0);
            split_patch(planes + ((c + 1) & 1) * PST);     // buffer read last in chunk c-1
          }
        }
        if (sa < nst) issue_w(c + (b + LEAD) / NSTG, (b + LEAD) % NSTG, sa % RING);   // slot read last in stage s-1 (complete before its barrier)
      };
      if (TPS == 9 || !late) staging();
      X6C_ACC(st_issue, st_prev);                                  // patch loads / split pass / weight DMA issue
      const u32x4c* wsl = wlane + (s % RING) * WST;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < TPS - 1; ++t) {
        if constexpr (TPS == 9) {
          // whole-chunk stage: the patch loads went out at the top of THIS stage; younger: only this stage's weight batch (the wait
          // also retires the weights of stage s+1, issued a stage earlier).  Split pass after four taps, or after seven (staggered half).
          if (more_c && (t == 4 || t == 6 || t == 7) && t == split_t) {
            x6c_wait_vm(nwb(sa));
            __builtin_amdgcn_sched_barrier(0);
            split_patch(planes + ((c + 1) & 1) * PST);   // buffer read last in chunk c-1
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        load_tap(wsl, pb, TPS * b + t + 1, (t + 1) % 3);
        mfma_tap(t % 3, TPS * b + t);
        ilv();
        __builtin_amdgcn_sched_barrier(0);
      }
      if (TPS == 3 && late) { staging(); __builtin_amdgcn_sched_barrier(0); }
      X6C_ACC(st_mfma, st_prev);
      // stage s+1 needs its weights W(s+1); everything issued after that batch may stay in flight: W(s+2) .. W(s+LEAD), and the
      // patch loads of the stages s+2-LEAD .. s that start a chunk (kernel-row stages only: a whole-chunk stage has split its
      // patch already)
      {
        int n = 0;
#pragma unroll
        for (int j = 2; j <= LEAD; ++j) n += nwb(s + j);
        if constexpr (TPS == 3) {
#pragma unroll
          for (int j = 0; j <= LEAD - 2; ++j) {
            const int bj = ((b - j) % 3 + 3) % 3;                     // kernel row of stage s-j
            if (bj == 0 && s - j >= 0 && (s - j) / 3 + 1 < nch) n += NPL;
          }
        }
        x6c_wait_vm(n);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // own tap reads and plane writes done
      X6C_ACC(st_wait, st_prev);
      __builtin_amdgcn_s_barrier();
      X6C_ACC(st_bar, st_prev);
      if (s + 1 < nst) load_tap(wlane + ((s + 1) % RING) * WST, b == NSTG - 1 ? pbn : pb, b == NSTG - 1 ? 0 : TPS * b + TPS, 0);
      mfma_tap(2, TPS * b + TPS - 1);
      ilv();
      __builtin_amdgcn_sched_barrier(0);
      X6C_ACC(st_mfma, st_prev);
    }
  }
#ifdef P2I_STAMP
  unsigned long long st_loop_end; X6C_NOW(st_loop_end);
#endif

  // ---- epilogue (shared with the f32 engine)
  const int gw = j0w + pjw, gh = j0h + pjh, gn = j0b + pjb;
  const bool pvld = gn < nimg && gh < g.nH && gw < g.nW;
  const int dHW = g.dT * g.dH * g.dW;                                   // destination channel stride
  const int gb_ = gn / g.nT, glt = gn - gb_ * g.nT;
  if constexpr (FUSED) {
    // class (pH, pW) of position (gh, gw) is destination pixel (2 gh + pH, 2 gw + pW): the pW pair is adjacent -> float2
    const size_t pos0 = (size_t)gb_ * g.Cm * dHW + (size_t)(glt * g.oT + g.pT) * g.dH * g.dW + (size_t)(2 * gh) * g.dW + 2 * gw;
#pragma unroll
    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
      for (int i = 0; i < TM; ++i)
        epilogue_pair16(acc[2 * ph][i], acc[2 * ph + 1][i], o0 + i * 32, lhi, g.Cm, pvld, pos0 + (size_t)ph * g.dW, (size_t)dHW,
                        (g.fused_atomic && blockIdx.z) ? nullptr : g.res, g.mask_y, g.mask_act, g.dst, g.fused_atomic != 0);
  } else {
    const size_t pos = (size_t)gb_ * g.Cm * dHW + (size_t)(glt * g.oT + g.pT) * g.dH * g.dW + (size_t)gh * g.dW + gw;
#pragma unroll
    for (int i = 0; i < TM; ++i)
      if (g.ksplit > 1)       // partial sum: bias / residual ride with split 0, the mask factor (0/1 for relu') distributes over the sum
        epilogue_tile16(acc[0][i], o0 + i * 32, lhi, g.Cm, pvld, pos, (size_t)dHW, blockIdx.z ? nullptr : g.bias, P2I_ACT_NONE,
                        blockIdx.z ? nullptr : g.res, g.mask_y, g.mask_act, g.dst, true);
      else
        epilogue_tile16(acc[0][i], o0 + i * 32, lhi, g.Cm, pvld, pos, (size_t)dHW, g.bias, g.act_epi, g.res, g.mask_y, g.mask_act, g.dst);
  }
#ifdef P2I_STAMP
  if (p2i_stamp_buf && lane == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long st_x; X6C_NOW(st_x);
    unsigned long long* o = p2i_stamp_buf + ((size_t)(blockIdx.x + gridDim.x * blockIdx.y) * NW + wave) * 8;
    o[0] = st_wait; o[1] = st_bar; o[2] = st_issue; o[3] = st_mfma; o[4] = st_loop_end - st_loop0; o[5] = st_loop0; o[6] = st_loop0 - st_entry;
    o[7] = st_x - st_loop_end;
  }
#endif
}


// ---------------------------------------------------------------------------------------------------------------------------------
// "x6p": the same tile, LDS images, weight ring and stage hand-over as patch_gemm_x6c_kernel, with the eight waves SPECIALISED:
// waves 0-3 (one per SIMD) are CONSUMERS -- operand reads and MFMAs only, 32 TM channels x 64 positions each (two position blocks
// share every weight operand: 0.625 / 0.75 reads per MFMA at TM 2 / 1) -- and waves 4-7 are PRODUCERS: the next chunk's patch loads,
// the split pass and all weight DMAs.  A consumer's instruction stream never contains a vector-memory wait, a split pass or a DMA
// issue; the producers' VALU / VMEM work runs on the same SIMDs beside the MFMAs and they park at the stage barrier when done.
// Same numerics, same epilogue, same X6cGeom.
// NPW = producer waves: 4 (one per SIMD beside its consumer) or 8 (two per SIMD: the 32-channel tiles' producers are otherwise as busy
// as their consumers -- stamped: split pass 33 %, load issue 23 %, DMA issue 20 % of their loop, consumers 16 % at the stage barrier;
// twelve waves leave 170 VGPRs per wave, which only the non-fused 32-channel kernel fits)
// TN = position blocks of 32 per consumer wave: 2 (256 positions per workgroup) or 1 (128: layers with so few positions that 32 x 256
// tiles would need split-K -- the 512-channel level at B = 8 -- get twice the workgroups instead of a zero-filled destination, atomic
// partial sums and a second activation pass; no epilogue exchange then: a consumer has one accumulator tile)
template <int TM, int TPS, bool FUSED = false, bool EPI4 = false, int NPW = 4, int TN = 2, bool PRE = false>
__global__ __launch_bounds__(64 * (4 + NPW)) void patch_gemm_x6p_kernel(const X6cGeom g) {
  constexpr int NSTG = 9 / TPS;
  constexpr int NCLS = FUSED ? 4 : 1;                     // FUSED: see X6C_CLS (strided data gradient, four parity classes)
  static_assert(TPS == 3 || TPS == 9, "taps per stage");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int MB = 32 * TM, NCW = 4, NTHR = 64 * (NCW + NPW), NPT = 64 * NPW;
  constexpr int NI = (2 * (TN == 2 ? X6cTile<8>::MAXCSL : X6cTile<4>::MAXCSL) + NPT - 1) / NPT;   // patch items per producer thread: 3, 2 or 1
  constexpr int RING = TPS == 9 ? 3 : (TM == 2 ? 4 : 6), LEAD = RING - 1;
  constexpr int WST = 6 * TPS * MB, NWI = WST / 64;
  const int CSl = g.CSl;
  int* ptab = reinterpret_cast<int*>(smem);
  const int ptab_sz = (CSl + 3) & ~3;
  u32x4c* wbuf = reinterpret_cast<u32x4c*>(smem + ptab_sz);
  u32x4c* planes = wbuf + RING * WST;
  const int PST = 6 * CSl;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  int tile = blockIdx.x;
  const int tw = tile % g.ntw; tile /= g.ntw;
  const int th = tile % g.nth;
  const int tb = tile / g.nth;
  const int j0b = tb << g.ljb, j0h = th << g.ljh, j0w = tw << g.ljw;
  const int o0 = blockIdx.y * MB;
  const int JWm = (1 << g.ljw) - 1, JHm = (1 << g.ljh) - 1;
  const int sHW = g.sH * g.sW;
  const int src_h0 = j0h + g.bH, src_w0 = j0w + g.bW;
  const int KCt = g.Ck >> 3;
  const int cps = g.Ck >> 4;
  const int cs = g.sT * sHW;
  const int nimg = g.B * g.nT;
  const int c0 = blockIdx.z * ((g.ns * cps) / g.ksplit);
  const int nch = (g.ns * cps) / g.ksplit;
  const int nst = NSTG * nch;

  for (int e = tid; e < CSl; e += NTHR) {
    const int row = fast_div(e, g.mg_ew);
    const int ew = e - row * g.eW;
    const int jb = fast_div(row, g.mg_eh);
    const int eh = row - jb * g.eH;
    const int n = j0b + jb, h = src_h0 + eh, w = src_w0 + ew;
    const int b = n / g.nT, lt = n - b * g.nT;
    // (PRE: offsets count 16-byte units of a plane -- 8 channels of one pixel -- instead of floats)
    ptab[e] = (n < nimg && (unsigned)h < (unsigned)g.sH && (unsigned)w < (unsigned)g.sW) ? ((b * (PRE ? KCt : g.Ck)) * g.sT + lt * g.mT) * sHW + h * g.sW + w : -1;
  }
  __syncthreads();                                                        // ptab visible

#ifdef P2I_STAMP
  unsigned long long st_a = 0, st_b = 0, st_c = 0, st_d = 0, st_e = 0, st_f = 0, st_t0, st_t1, st_l0 = 0, st_l1 = 0, st_entry;
  X6C_NOW(st_entry);
#endif
  if (wave >= NCW) {
    // =============================================================== producers
    const int ptid = tid - 64 * NCW, pw = wave - NCW;
    const v4i32 rs_w = make_rsrc(g.wb, g.wb_bytes);
    const unsigned wbuf_la = lds_base(smem) + 4u * ptab_sz;
    const int wl_m = TM == 2 ? lane : l31, wl_kg = TM == 2 ? 0 : lhi;
    const int wvoff = (o0 + wl_m < g.CmPad) ? (wl_kg * g.CmPad + wl_m) * 16 : -16;
    int it_off[NI], it_dst[NI], it_f0[NI];
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const int item = it * NPT + ptid;
      const int kg = item >= CSl ? 1 : 0, e = item - kg * CSl;
      const bool in = item < 2 * CSl;
      const int po = in ? ptab[e] : -1;
      it_off[it] = po < 0 ? -1 : po + kg * (PRE ? 1 : 8) * cs;
      it_dst[it] = in ? kg * CSl + e : -1;
      const int jbi = fast_div(fast_div(in ? e : 0, g.mg_ew), g.mg_eh);
      it_f0[it] = ((j0b + jbi) % g.nT) * g.mT;
    }
    float pv[PRE ? 1 : NI][8];
    u32x4c pq[PRE ? NI : 1][3];                                          // PRE: the item's three planes as they come from memory
    bool pok[NI];
    auto load_patch = [&](int c) {
      if constexpr (PRE) {
        const int cg = c0 + c;
        const int j = cg >= 2 * cps ? 2 : (cg >= cps ? 1 : 0);
        const int dt = j == 2 ? g.sdt2 : (j == 1 ? g.sdt1 : g.sdt0);
        int voff[NI];
#pragma unroll
        for (int it = 0; it < NI; ++it) {
          pok[it] = it_off[it] >= 0 && (unsigned)(it_f0[it] + dt) < (unsigned)g.sT;
          voff[it] = pok[it] ? (it_off[it] + dt * sHW) * 16 : 0;
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          const u32x4c* sq = g.splanes + (size_t)p * g.plane_units + (size_t)(cg - j * cps) * 2 * cs;
          const unsigned long long sq_u = (unsigned long long)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)sq) & 0xffffffffull;
          const unsigned long long sq_hi = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)sq >> 32));
          const unsigned long long sbase = sq_u | (sq_hi << 32);
#pragma unroll
          for (int it = 0; it < NI; ++it) {
            if (it == 0) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(pq[it][p]) : "v"(voff[it]), "s"(sbase) : "memory");
            else asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(pq[it][p]) : "v"(voff[it]), "s"(sbase) : "memory");
          }
        }
      } else {
      const int cg = c0 + c;
      const int j = cg >= 2 * cps ? 2 : (cg >= cps ? 1 : 0);
      const int dt = j == 2 ? g.sdt2 : (j == 1 ? g.sdt1 : g.sdt0);
      const float* sc = g.src + (size_t)(cg - j * cps) * 16 * cs;
      int voff[NI];
#pragma unroll
      for (int it = 0; it < NI; ++it) {
        pok[it] = it_off[it] >= 0 && (unsigned)(it_f0[it] + dt) < (unsigned)g.sT;
        voff[it] = pok[it] ? (it_off[it] + dt * sHW) * 4 : 0;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float* sq = sc + (size_t)q * cs;
        const unsigned long long sq_u = (unsigned long long)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)sq) & 0xffffffffull;
        const unsigned long long sq_hi = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)sq >> 32));
        const unsigned long long sbase = sq_u | (sq_hi << 32);
#pragma unroll
        for (int it = 0; it < NI; ++it) {
          if (it == 0) asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=v"(pv[it][q]) : "v"(voff[it]), "s"(sbase) : "memory");
          else asm volatile("global_load_dword %0, %1, %2" : "=v"(pv[it][q]) : "v"(voff[it]), "s"(sbase) : "memory");
        }
      }
      }
    };
    auto split_patch = [&](u32x4c* pb) {
      if constexpr (PRE) {                                                // nothing to split: three 16-byte LDS writes per item
#pragma unroll
        for (int it = 0; it < NI; ++it)
#pragma unroll
          for (int p = 0; p < 3; ++p) asm volatile("" : "+v"(pq[it][p]));    // values exist from here on
        const u32x4c zero = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int it = 0; it < NI; ++it) {
          if (it_dst[it] < 0) continue;
          pb[it_dst[it]] = pok[it] ? pq[it][0] : zero;
          pb[2 * CSl + it_dst[it]] = pok[it] ? pq[it][1] : zero;
          pb[4 * CSl + it_dst[it]] = pok[it] ? pq[it][2] : zero;
        }
      } else {
#pragma unroll
      for (int it = 0; it < NI; ++it)
#pragma unroll
        for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(pv[it][q]));     // values exist from here on (see patch_gemm_x6c_kernel)
#pragma unroll
      for (int it = 0; it < NI; ++it) {
        if (it_dst[it] < 0) continue;
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = pok[it] ? pv[it][q] : 0.f;
        u32x4c h, mi, lo;
        split8c(v, h, mi, lo);
        pb[it_dst[it]] = h;
        pb[2 * CSl + it_dst[it]] = mi;
        pb[4 * CSl + it_dst[it]] = lo;
      }
      }
    };
    constexpr int NWR = (NWI + NPW - 1) / NPW;
    int w_soff[NWR][NSTG];
    unsigned w_dst[NWR];
#pragma unroll
    for (int r = 0; r < NWR; ++r) {
      const int u = pw + NPW * r;
      const int pt = TM == 2 ? u >> 1 : u, kg = TM == 2 ? u & 1 : 0;
      const int p = pt / TPS, tl = pt % TPS;
#pragma unroll
      for (int b = 0; b < NSTG; ++b)
        w_soff[r][b] = u < NWI ? __builtin_amdgcn_readfirstlane((((p * g.ntaps_w + g.tap_w[TPS * b + tl]) * KCt + kg) * g.CmPad + o0) * 16) : 0;
      w_dst[r] = wbuf_la + 16u * (unsigned)(u * 64);
    }
    const int w_cstep = 2 * g.CmPad * 16;
    const int w_tstep = KCt * g.CmPad * 16;
    auto issue_w = [&](int c, int b, int sb) {
      const int cg = c0 + c;
      const int j = cg >= 2 * cps ? 2 : (cg >= cps ? 1 : 0);
      const int soff = (cg - j * cps) * w_cstep + (j == 2 ? g.swt2 : (j == 1 ? g.swt1 : g.swt0)) * w_tstep;
#pragma unroll
      for (int r = 0; r < NWR; ++r)
        if (pw + NPW * r < NWI) dma_b128(rs_w, w_dst[r] + 16u * (unsigned)(sb * WST), wvoff, w_soff[r][b] + soff);
    };
    const int nw_mine = (NWI / NPW) + (pw < NWI % NPW ? 1 : 0);
    constexpr int NPL = (PRE ? 3 : 8) * NI;                                 // patch loads per thread and chunk
    // (the weights of the first LEAD stages are DMA'd by the CONSUMER waves, which have nothing else to do during the prologue)
    auto nwb = [&](int st) { return (st >= LEAD && st < nst) ? nw_mine : 0; };
    // vector-memory operations of a producer wave, in issue order: prologue [P(0)]; per stage s:
    // [P(chunk+1) if the stage starts a chunk that has a successor] [W(s+LEAD)]
    load_patch(0);
    {
      int n = 0;
#pragma unroll
      for (int st = 0; st < LEAD; ++st) n += nwb(st);
      x6c_wait_vm(n);
      __builtin_amdgcn_sched_barrier(0);
      split_patch(planes);
      if constexpr (TPS == 9) {
        // whole-chunk stages: the patch of chunk c+1 is LOADED during stage c-1 and split at the top of stage c, so that a producer
        // never waits for loads it has just issued (a stage would otherwise be as long as a memory round trip); chunk 1 starts here
        if (nch > 1) load_patch(1);
        x6c_wait_vm(n - nwb(0) + (nch > 1 ? NPL : 0));                      // weights of stage 0: younger are W(1) .. and the loads of chunk 1
      } else {
        x6c_wait_vm(n - nwb(0));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                         // hand-over of stage 0
    int s = 0;
    for (int c = 0; c < nch; ++c) {
      const bool more_c = c + 1 < nch;
#pragma unroll
      for (int b = 0; b < NSTG; ++b, ++s) {
        const int sa = s + LEAD;
        if constexpr (TPS == 9) {
#ifdef P2I_STAMP
          X6C_NOW(st_t0);
          if (s == 0) st_l0 = st_t0;
#endif
          if (more_c) {
            // loads of chunk c+1: issued in the previous stage in front of its weight batch W(s-1+LEAD) (stage 0: in the prologue, last)
            x6c_wait_vm(s == 0 ? 0 : nwb(s - 1 + LEAD));
            X6C_ACC(st_a, st_t0);                                         // wait for the patch loads
            __builtin_amdgcn_sched_barrier(0);
            split_patch(planes + ((c + 1) & 1) * PST);                   // buffer read last in chunk c-1
            __builtin_amdgcn_sched_barrier(0);
            X6C_ACC(st_b, st_t0);                                         // split pass
            if (c + 2 < nch) load_patch(c + 2);                          // into the registers the split has just consumed
            X6C_ACC(st_e, st_t0);                                         // patch load issue
          }
          if (sa < nst) issue_w(c + LEAD, 0, sa % RING);                 // slot read last in stage s-1
          X6C_ACC(st_f, st_t0);                                           // weight DMA issue
          // stage s+1 needs W(s+1), issued at the top of stage s-1 (LEAD = 2); issued after it: the loads of chunk c+2 and W(s+2)
          x6c_wait_vm(nwb(s + 2) + ((more_c && c + 2 < nch) ? NPL : 0));
          X6C_ACC(st_c, st_t0);                                           // wait for the weights of the next stage
        } else {
          if (b == 0 && more_c) load_patch(c + 1);
          if (sa < nst) issue_w(c + (b + LEAD) / NSTG, (b + LEAD) % NSTG, sa % RING);   // slot read last in stage s-1
          if (b == NSTG - 1 && more_c) {
            // patch loads went out at the top of stage s - (NSTG - 1); younger: the weight batches of the stages since then
            int n = 0;
#pragma unroll
            for (int j = 0; j < NSTG; ++j) n += nwb(sa - j);
            x6c_wait_vm(n);
            __builtin_amdgcn_sched_barrier(0);
            split_patch(planes + ((c + 1) & 1) * PST);                   // buffer read last in chunk c-1
          }
          // stage s+1 needs W(s+1), issued at the top of stage s+1-LEAD.  Issued AFTER it (vmcnt counts in issue order, done or not):
          // W(s+2) .. W(s+LEAD) and the patch loads at the tops of the stages s+2-LEAD .. s that start a chunk with a successor
          int n = 0;
#pragma unroll
          for (int j = 2; j <= LEAD; ++j) n += nwb(s + j);
#pragma unroll
          for (int j = 0; j <= LEAD - 2; ++j) {
            const int sj = s - j;
            if (sj >= 0 && sj % NSTG == 0 && sj / NSTG + 1 < nch) n += NPL;
          }
          x6c_wait_vm(n);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // plane writes done
#ifdef P2I_STAMP
        X6C_NOW(st_t1);
#endif
        __builtin_amdgcn_s_barrier();                                     // hand-over of stage s+1
#ifdef P2I_STAMP
        X6C_ACC(st_d, st_t1);                                             // parked at the stage barrier
        st_l1 = st_t1;
#endif
      }
    }
  }

  // ================================================================= consumers (producers only set up the epilogue geometry here)
  const bool producer = wave >= NCW;
  const int cw = wave & (NCW - 1);                                       // positions [32 TN cw, 32 TN (cw + 1)); producer w helps consumer w - 4
  int lane_base[TN], pjw_[TN], pjh_[TN], pjb_[TN];
#pragma unroll
  for (int nb = 0; nb < TN; ++nb) {
    const int pix = cw * 32 * TN + nb * 32 + l31;
    pjw_[nb] = pix & JWm; pjh_[nb] = (pix >> g.ljw) & JHm; pjb_[nb] = pix >> (g.ljw + g.ljh);
    lane_base[nb] = (pjb_[nb] * g.eH + pjh_[nb]) * g.eW + pjw_[nb] + lhi * CSl;
  }
  int toff[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) toff[t] = g.tap_off[t];
  f32x16 acc[NCLS][TM][TN];
#pragma unroll
  for (int q = 0; q < NCLS; ++q)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int nb = 0; nb < TN; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][i][nb][r] = 0.f;
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
  constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
  u32x4c A[3][TM][3], Bv[3][TN][3];
  auto load_tap = [&](const u32x4c* wsl, const u32x4c* pbuf, int tap, int buf) {
    const int tl = tap % TPS;
    const int to = toff[tap];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int i = 0; i < TM; ++i) A[buf][i][p] = wsl[((p * TPS + tl) * 2) * MB + 32 * i];
#pragma unroll
      for (int nb = 0; nb < TN; ++nb) Bv[buf][nb][p] = pbuf[lane_base[nb] + p * 2 * CSl + to];
    }
  };
  auto mfma_tap = [&](int buf, int slot) {
    const int cl = FUSED ? X6C_CLS[slot] : 0;
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int nb = 0; nb < TN; ++nb)
          acc[cl][i][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8c, A[buf][i][PA[q]]), __builtin_bit_cast(bf16x8c, Bv[buf][nb][PB[q]]),
                                                                   acc[cl][i][nb], 0, 0, 0);
  };
  // (reads in consumption order, two per MFMA gap at the front of the tap: measured 2-4 % slower, gpurun_out/r03h/pc2.log)
  auto ilv = [&]() {
    constexpr int NR = 3 * TM + 3 * TN, NM = 6 * TM * TN;
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (i < NR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
  };
  const u32x4c* wlane = wbuf + lhi * MB + l31;
  if (!producer) {
    {
      // prologue: the weights of stages 0 .. LEAD-1 (the producers are busy with the first patch)
      const v4i32 rs_w = make_rsrc(g.wb, g.wb_bytes);
      const unsigned wbuf_la = lds_base(smem) + 4u * ptab_sz;
      const int wl_m = TM == 2 ? lane : l31, wl_kg = TM == 2 ? 0 : lhi;
      const int wvoff = (o0 + wl_m < g.CmPad) ? (wl_kg * g.CmPad + wl_m) * 16 : -16;
      const int w_cstep = 2 * g.CmPad * 16, w_tstep = KCt * g.CmPad * 16;
#pragma unroll
      for (int st = 0; st < LEAD; ++st) {
        if (st >= nst) break;
        const int cg = c0 + st / NSTG, b = st % NSTG;
        const int j = cg >= 2 * cps ? 2 : (cg >= cps ? 1 : 0);
        const int soff = (cg - j * cps) * w_cstep + (j == 2 ? g.swt2 : (j == 1 ? g.swt1 : g.swt0)) * w_tstep;
        for (int u = cw; u < NWI; u += NCW) {
          const int pt = TM == 2 ? u >> 1 : u, kg = TM == 2 ? u & 1 : 0;
          const int p = pt / TPS, tl = pt % TPS;
          const int so = (((p * g.ntaps_w + g.tap_w[TPS * b + tl]) * KCt + kg) * g.CmPad + o0) * 16 + soff;
          dma_b128(rs_w, wbuf_la + 16u * (unsigned)(u * 64 + (st % RING) * WST), wvoff, so);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                                         // stage 0 handed over by the producers
    // the consumers' instruction stream is operand reads + MFMAs only; the producers on the same SIMDs run VALU bursts (split pass) and
    // vector-memory issue: with a raised priority the arbiter prefers the consumer whenever both have an instruction ready
    if (g.cons_prio == 1) __builtin_amdgcn_s_setprio(1);
    else if (g.cons_prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (g.cons_prio == 3) __builtin_amdgcn_s_setprio(3);
    load_tap(wlane, planes, 0, 0);
    int s = 0;
    for (int c = 0; c < nch; ++c) {
      const u32x4c* pb = planes + (c & 1) * PST;
      const u32x4c* pbn = planes + ((c + 1) & 1) * PST;
#pragma unroll
      for (int b = 0; b < NSTG; ++b, ++s) {
        const u32x4c* wsl = wlane + (s % RING) * WST;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < TPS - 1; ++t) {
          load_tap(wsl, pb, TPS * b + t + 1, (t + 1) % 3);
          mfma_tap(t % 3, TPS * b + t);
          ilv();
          __builtin_amdgcn_sched_barrier(0);
        }
#ifdef P2I_STAMP
        X6C_NOW(st_t0);
        if (s == 0) st_l0 = st_t0;
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // own operand reads of this stage have returned
        X6C_ACC(st_a, st_t0);
        __builtin_amdgcn_s_barrier();                                     // stage s+1 handed over
#ifdef P2I_STAMP
        X6C_ACC(st_d, st_t0);                                             // parked at the stage barrier
        st_l1 = st_t0;
#endif
        if (s + 1 < nst) load_tap(wlane + ((s + 1) % RING) * WST, b == NSTG - 1 ? pbn : pb, b == NSTG - 1 ? 0 : TPS * b + TPS, 0);
        mfma_tap(2, TPS * b + TPS - 1);
        ilv();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if (!producer && g.cons_prio) __builtin_amdgcn_s_setprio(0);
  // ---- epilogue, shared: the consumer keeps its first position block and parks the second one in LDS (every LDS image is dead
  // now: all operand reads returned before the last stage barrier), its producer partner (same SIMD) picks it up, so that all
  // eight waves load residual / mask values and store, as in the symmetric kernel
  float* xch = reinterpret_cast<float*>(wbuf);                            // [4 consumers][NCLS][TM][16][64] floats <= 64 KB
  if constexpr (TN == 2) {
    if (!producer) {
#pragma unroll
      for (int q = 0; q < NCLS; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) xch[(((cw * NCLS + q) * TM + i) * 16 + r) * 64 + lane] = acc[q][i][1][r];
    }
    __syncthreads();
    if (wave >= 2 * NCW) return;                                          // (a second producer wave per SIMD has no share in the epilogue)
    if (producer) {
#pragma unroll
      for (int q = 0; q < NCLS; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[q][i][1][r] = xch[(((cw * NCLS + q) * TM + i) * 16 + r) * 64 + lane];
    }
  } else {
    if (producer) return;
  }
  const int dHW = g.dT * g.dH * g.dW;
#pragma unroll
  for (int nb = 0; nb < TN; ++nb) {
    if (TN == 2 && (nb == 1) != producer) continue;                       // consumer: block 0, producer: block 1
    const int gw = j0w + pjw_[nb], gh = j0h + pjh_[nb], gn = j0b + pjb_[nb];
    const bool pvld = gn < nimg && gh < g.nH && gw < g.nW;
    const int gb_ = gn / g.nT, glt = gn - gb_ * g.nT;
    if constexpr (FUSED) {
      const size_t pos0 = (size_t)gb_ * g.Cm * dHW + (size_t)(glt * g.oT + g.pT) * g.dH * g.dW + (size_t)(2 * gh) * g.dW + 2 * gw;
#pragma unroll
      for (int ph = 0; ph < 2; ++ph)
#pragma unroll
        for (int i = 0; i < TM; ++i)
          epilogue_pair16(acc[2 * ph][i][nb], acc[2 * ph + 1][i][nb], o0 + i * 32, lhi, g.Cm, pvld, pos0 + (size_t)ph * g.dW, (size_t)dHW,
                          (g.fused_atomic && blockIdx.z) ? nullptr : g.res, g.mask_y, g.mask_act, g.dst, g.fused_atomic != 0);
    } else if constexpr (EPI4) {
      // 16-byte accesses: this lane's group = positions 4 (lane & 7) .. + 3 of the block (one channel per lane and pass)
      const int pg = cw * 32 * TN + nb * 32 + 4 * (lane & 7);
      const int qw = pg & JWm, qh = (pg >> g.ljw) & JHm, qb = pg >> (g.ljw + g.ljh);
      const int hw = j0w + qw, hh = j0h + qh, hn = j0b + qb;
      const bool pv4 = hn < nimg && hh < g.nH && hw < g.nW;
      const int hb = hn / g.nT, hlt = hn - hb * g.nT;
      const size_t pos4 = (size_t)hb * g.Cm * dHW + (size_t)(hlt * g.oT + g.pT) * g.dH * g.dW + (size_t)hh * g.dW + hw;
      float* tl = xch + 4 * NCLS * TM * 16 * 64 + wave * (32 * 36);      // wave-private 32 x 36 image behind the exchange area
#pragma unroll
      for (int i = 0; i < TM; ++i)
        if (g.ksplit > 1)
          epilogue_tile16_v4(acc[0][i][nb], tl, o0 + i * 32, lane, g.Cm, pos4, pv4, (size_t)dHW, blockIdx.z ? nullptr : g.bias, P2I_ACT_NONE,
                             blockIdx.z ? nullptr : g.res, g.mask_y, g.mask_act, g.dst, true);
        else
          epilogue_tile16_v4(acc[0][i][nb], tl, o0 + i * 32, lane, g.Cm, pos4, pv4, (size_t)dHW, g.bias, g.act_epi, g.res, g.mask_y, g.mask_act, g.dst);
    } else {
      const size_t pos = (size_t)gb_ * g.Cm * dHW + (size_t)(glt * g.oT + g.pT) * g.dH * g.dW + (size_t)gh * g.dW + gw;
#pragma unroll
      for (int i = 0; i < TM; ++i)
        if (g.ksplit > 1)
          epilogue_tile16(acc[0][i][nb], o0 + i * 32, lhi, g.Cm, pvld, pos, (size_t)dHW, blockIdx.z ? nullptr : g.bias, P2I_ACT_NONE,
                          blockIdx.z ? nullptr : g.res, g.mask_y, g.mask_act, g.dst, true);
        else
          epilogue_tile16(acc[0][i][nb], o0 + i * 32, lhi, g.Cm, pvld, pos, (size_t)dHW, g.bias, g.act_epi, g.res, g.mask_y, g.mask_act, g.dst);
    }
  }
#ifdef P2I_STAMP
  if (p2i_stamp_buf && lane == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long st_end; X6C_NOW(st_end);
    unsigned long long* o = p2i_stamp_buf + ((size_t)(blockIdx.x + gridDim.x * blockIdx.y) * 8 + wave) * 8;
    o[0] = st_a; o[1] = st_d; o[2] = st_b; o[3] = st_c; o[4] = st_l1 - st_l0; o[5] = st_l0 - st_entry; o[6] = st_e; o[7] = st_end - st_l1;
  }
#endif
}

// Fewest workgroups for which a tile variant is used (256 CUs; below that the next smaller tile, or the f32 engine).  Read per call
// (not cached) so that the parity tests can send small layers through these kernels (P2I_X6C_MIN_WG=1).
static int x6c_min_wg() { const char* e = getenv("P2I_X6C_MIN_WG"); return e ? atoi(e) : 200; }
// When not even the smallest tile reaches that count, the smallest tile is still taken from this many workgroups on: a split-pipe
// workgroup does its tile ~2x faster than the f32 engine's, which has no more workgroups to offer for the same layer (its smallest tile
// is 32 x 128 too).  Small per-GPU batches live here: at B = 4 the 512-channel level has 128 tiles of 32 x 128.
static int x6c_min_wg_low() { const char* e = getenv("P2I_X6C_MIN_WG"); if (e) return atoi(e); e = getenv("P2I_X6C_MIN_WG_LOW"); return e ? atoi(e) : 64; }
static int x6c_pc();
static int x6p_tn1() { const char* e = getenv("P2I_X6P_TN1"); return e ? atoi(e) : 1; }      // read per call (A/B runs)
// the fused strided data gradient is taken from 128 workgroups on (half the chip, one round: the 128 -> 256 stride-2 layer at B = 8);
// its f32 alternative is no faster per workgroup.  P2I_X6C_MIN_WG overrides both thresholds (tests).
static int x6c_fused_min_wg() { const char* e = getenv("P2I_X6C_MIN_WG"); return e ? atoi(e) : 128; }
// P2I_X6C_TILE=<NW><TM> (82, 81, 41; 41 = the 32 x 128 tile, falling back to 81 where that has no kernel) forces one variant (tests, tuning)
static int x6c_forced() { const char* e = getenv("P2I_X6C_TILE"); return e ? atoi(e) : 0; }

// P2I_X6C_STAGGER=1: waves 4-7 run their staging work late in the stage (see `staging` in the kernel).  Measured in round 3 at B = 8
// (gpurun_out/r03d/s0.log vs s1.log): no gain -- 64-channel level 75.0 -> 74.3 / 76.1 -> 81.2 us, every other layer within 1 us -- so
// the lockstep schedule stays the default; read per call (A/B runs)
static int x6c_stagger() { const char* e = getenv("P2I_X6C_STAGGER"); return e ? atoi(e) : 0; }
static int x6c_fused_ksplit() { const char* e = getenv("P2I_X6C_FUSED_KSPLIT"); return e ? atoi(e) : 0; }   // read per call (tests)

struct X6cVariant { int NW, TM; };
static const X6cVariant kX6cVariants[] = {{8, 2}, {8, 1}, {4, 1}};      // in order of per-CU efficiency ({4, 1}: 32 x 128, x6p only)

struct X6cPick { int v, jb, jh, jw, csl, ksplit; long long wgs; };
// first variant whose grid fills the chip and whose patch fits; v = -1: none.  When even the 32 x 256 tiles are too few (the
// 512-channel level at B = 8: 128 tiles) and the epilogue is linear (no activation: split partial sums cannot pass through one),
// the channel chunks are split over two workgroups per tile that add into a zeroed destination (two addends: order-independent).
static X6cPick x6c_pick(int B, int nH, int nW, int Cm, int Ck, bool linear_epi, int Ck_slice) {   // Ck = channels x tap slices
  int forced = x6c_forced();
  const int min_wg = x6c_min_wg();
  const bool tn1_ok = x6c_pc() && Ck_slice >= 32;                // the 32 x 128 tile exists as a producer / consumer kernel only
  if (forced == 41 && !tn1_ok) forced = 81;
  static const int ksplit_on = getenv("P2I_X6C_KSPLIT") ? atoi(getenv("P2I_X6C_KSPLIT")) : 1;
  // candidate order: 64x256, 32x256, then split-K 32x256.  (P2I_X6C_KSPLIT=2 tries split-K 64x256 before plain 32x256: measured 70.8 vs
  // 73.5 us on the 256-channel level, inside the noise, not the default.)
  // (round 3: the 32 x 128 tile of the producer / consumer kernel comes before any split-K form; P2I_X6P_TN1=0 removes it)
  static const int order_a[5][2] = {{0, 0}, {1, 0}, {2, 0}, {1, 1}, {0, 1}}, order_b[5][2] = {{0, 0}, {0, 1}, {1, 0}, {2, 0}, {1, 1}};
  X6cPick low{-1, 0, 0, 0, 0, 1, 0};                       // the un-split variant with the most workgroups below the threshold
  for (int cand = 0; cand < 5; ++cand) {
      const int v = (ksplit_on == 2 ? order_b : order_a)[cand][0], pass = (ksplit_on == 2 ? order_b : order_a)[cand][1];
      const X6cVariant& t = kX6cVariants[v];
      if (v == 2 && (!tn1_ok || (forced ? forced != 41 : !x6p_tn1()))) continue;
      if (forced && forced != t.NW * 10 + t.TM) continue;
      if (t.TM == 2 && Cm <= 32 && !forced) continue;      // half of a 64-channel tile would be padding (16 -> 64 layers' data gradient)
      if (pass == 1 && ((ksplit_on != 2 && t.TM != 1) || !ksplit_on || !linear_epi || ((Ck >> 4) & 1) || forced)) continue;
      int jb, jt, jh, jw;
      pick_tile_dims(32 * t.NW, B, 1, nH, nW, jb, jt, jh, jw);
      const int csl = jb * (jh + 2) * (jw + 2);
      if (jt != 1 || csl > (t.NW == 8 ? X6cTile<8>::MAXCSL : X6cTile<4>::MAXCSL)) continue;
      const long long wgs = (long long)ceil_div(B, jb) * ceil_div(nH, jh) * ceil_div(nW, jw) * ceil_div(Cm, 32 * t.TM) * (pass + 1);
      if (wgs >= min_wg) return X6cPick{v, jb, jh, jw, csl, pass + 1, wgs};
      if (pass == 0 && !forced && wgs >= x6c_min_wg_low() && (low.v < 0 || wgs > low.wgs)) low = X6cPick{v, jb, jh, jw, csl, 1, wgs};
    }
  return low;          // (v = -1 when nothing qualified: f32 engine)
}

// cheap host-side test used before the weights are split: would run_patch_gemm_x6c take this layer?
bool x6c_would_take(const p2i_conv_desc* d, bool dgrad, int act_epi) {
  static const int on = getenv("P2I_CONV_X6C") ? atoi(getenv("P2I_CONV_X6C")) : 1;
  (void)act_epi;                                       // (an activation is applied by a second pass when the launch is split-K)
  if (on && dgrad && d->sh == 2 && d->sw == 2 && d->st == 1 && d->kh == 3 && d->kw == 3 && d->ph == 1 && d->pw == 1 &&
      ((d->kt == 1 && d->pt == 0) || (d->kt == 3 && d->pt == 1)) && (d->Cout & 15) == 0 && d->Cout >= 16 && !(d->Hi & 1) && !(d->Wi & 1)) {
    // fused strided data gradient (run_patch_gemm_x6c_fused): same tile / grid arithmetic as there
    static const int fused_on = getenv("P2I_CONV_X6C_FUSED") ? atoi(getenv("P2I_CONV_X6C_FUSED")) : 1;
    if (!fused_on) return false;
    const int nimg = d->B * d->Ti, nH = d->Hi / 2, nW = d->Wi / 2;
    int jb, jt, jh, jw;
    pick_tile_dims(256, nimg, 1, nH, nW, jb, jt, jh, jw);
    if (jt != 1 || jb * (jh + 1) * (jw + 1) > X6cTile<8>::MAXCSL) return false;
    const long long tiles = (long long)ceil_div(nimg, jb) * ceil_div(nH, jh) * ceil_div(nW, jw) * ceil_div(d->Cin, 32);
    const bool can_split = x6c_fused_ksplit() && (((d->kt * (d->Cout >> 4)) & 1) == 0);
    return tiles >= x6c_fused_min_wg() || (can_split && 2 * tiles >= x6c_fused_min_wg());
  }
  if (!on || d->kh != 3 || d->kw != 3 || d->sh != 1 || d->sw != 1 || d->ph != 1 || d->pw != 1) return false;
  const bool flat = d->kt == 1 && d->st == 1 && d->pt == 0;           // 2-D layer (or frames convolved independently)
  const bool vol = d->kt == 3 && d->pt == 1 && d->st <= 2;            // 3 x 3 x 3, t stride 1 or 2: tap slices along t
  if (!flat && !vol) return false;
  const int Ck = dgrad ? d->Cout : d->Cin, Cm = dgrad ? d->Cin : d->Cout;
  const int nH = dgrad ? d->Hi : d->Ho, nW = dgrad ? d->Wi : d->Wo;
  if ((Ck & 15) != 0 || Ck < 16) return false;
  // images of one launch: forward all output frames; data gradient one t-parity class of the input frames
  const int nT = dgrad ? d->Ti / d->st : d->To;
  if (nT < 1) return false;
  const int ns = flat ? 1 : (dgrad && d->st == 2 ? 1 : 3);           // fewest slices of a launch (conservative for the chunk-parity test)
  const bool whole = !dgrad || d->st == 1;
  return x6c_pick(d->B * nT, nH, nW, Cm, ns * Ck, whole, Ck).v >= 0;
}

// second pass of a split-K forward whose epilogue has an activation: y = act(sum of the two partial sums [+ bias, added by split 0])
// + residual, in place (4 MB at the 512-channel level: ~3 us)
__global__ __launch_bounds__(256) void x6c_post_act_kernel(float* __restrict__ y, const float* __restrict__ res, long long n4, int act) {
  typedef float f32x4p __attribute__((ext_vector_type(4)));
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    f32x4p v = reinterpret_cast<f32x4p*>(y)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], act);
    if (res) {
      const f32x4p r = reinterpret_cast<const f32x4p*>(res)[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += r[e];
    }
    reinterpret_cast<f32x4p*>(y)[i] = v;
  }
}

template <int NW, int TM, bool FUSED = false, int TPS = 3>
static void x6c_launch(const X6cGeom& g, dim3 grid, size_t lds, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)patch_gemm_x6c_kernel<NW, TM, FUSED, TPS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  P2I_LAUNCH((patch_gemm_x6c_kernel<NW, TM, FUSED, TPS>), grid, dim3(64 * NW), lds, s, g);
}
template <int TM, int TPS, bool FUSED = false, bool EPI4 = false, int NPW = 4, int TN = 2, bool PRE = false>
static void x6p_launch(const X6cGeom& g, dim3 grid, size_t lds, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)patch_gemm_x6p_kernel<TM, TPS, FUSED, EPI4, NPW, TN, PRE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  P2I_LAUNCH((patch_gemm_x6p_kernel<TM, TPS, FUSED, EPI4, NPW, TN, PRE>), grid, dim3(64 * (4 + NPW)), lds, s, g);
}
// experiment (round 4): the source planes of the NEXT split-pipe launch of this thread (tools/presplit_probe.py); consumed by it
static thread_local const void* g_next_splanes = nullptr;
// P2I_X6P_NPW=4: one producer wave per SIMD for the 32-channel tiles too (default 8); read per call (A/B runs)
static int x6p_npw() { const char* e = getenv("P2I_X6P_NPW"); return (e && atoi(e) == 4) ? 4 : 8; }
// Producer / consumer wave roles (patch_gemm_x6p_kernel): the default for every layer with more than one 16-channel chunk;
// P2I_X6C_PC=0 keeps the symmetric kernels; read per call (A/B runs).  Measured at B = 8, symmetric -> producer / consumer
// (gpurun_out/r03f/pc0.log, pc2.log): 64-channel level 75.4 -> 75.8 / 79.0 -> 73.9 us (fwd / dgrad), 128: 62.3 -> 57.4, 256: 64.3 ->
// 56.3, 512: 75.8 -> 68.6, 3-D 128 -> 128: 86.0 -> 76.5; K = 16 (16 -> 64 forward): 36.8 -> 39.9, hence the exception.
static int x6c_pc() { const char* e = getenv("P2I_X6C_PC"); return e ? atoi(e) : 1; }
// taps per stage of the 32-channel tiles: 9 (whole chunk) unless P2I_X6C_TPS=3 (read per call: A/B runs)
static int x6c_tps1() { const char* e = getenv("P2I_X6C_TPS"); return (e && atoi(e) == 3) ? 3 : 9; }
static size_t x6c_lds_bytes(int csl, int tm, int tps) {
  const int ring = tps == 9 ? 3 : (tm == 2 ? 4 : 6);
  return sizeof(float) * (size_t)((csl + 3) & ~3) + 16 * (size_t)(ring * 6 * tps * 32 * tm + 2 * 6 * csl);
}

// returns 1 when the layer is not an x6c case (caller continues with the other engines)
int run_patch_gemm_x6c(PatchGeom g, const ClassSpec& cs, const uint16_t* wb, int ntaps_w, int* plan6, hipStream_t s, bool dry) {
  static const int on = getenv("P2I_CONV_X6C") ? atoi(getenv("P2I_CONV_X6C")) : 1;
  if (!on || (!dry && wb == nullptr) || g.src_y != nullptr || (g.Ck & 15) != 0 || g.Ck < 16) return 1;
  // spatially a 3x3 stride-1 layer; along t: any source multiplier / destination stride / offset, taps in 1..3 slices of 9
  if (cs.mH != 1 || cs.mW != 1 || cs.oH != 1 || cs.oW != 1 || cs.pH || cs.pW) return 1;
  if (cs.ntaps != 9 && cs.ntaps != 18 && cs.ntaps != 27) return 1;
  const int ns = cs.ntaps / 9;
  int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  for (int i = 0; i < 9; ++i) {
    const int d[3] = {cs.dt[i], cs.dh[i], cs.dw[i]};
    for (int k = 0; k < 3; ++k) {
      if (i == 0 || d[k] < lo[k]) lo[k] = d[k];
      if (i == 0 || d[k] > hi[k]) hi[k] = d[k];
    }
  }
  if (lo[0] != hi[0] || hi[1] - lo[1] != 2 || hi[2] - lo[2] != 2) return 1;
  for (int j = 0; j < ns; ++j)                          // every slice: one source frame offset, slice 0's spatial taps, weight taps shifted as a block
    for (int i = 0; i < 9; ++i) {
      const int q = 9 * j + i;
      if (cs.dt[q] != cs.dt[9 * j] || cs.dh[q] != cs.dh[i] || cs.dw[q] != cs.dw[i] || cs.tw[q] - cs.tw[i] != cs.tw[9 * j] - cs.tw[0]) return 1;
    }
  const long long n_dst = (long long)g.B * g.Cm * g.dT * g.dH * g.dW;
  const long long nimg = (long long)g.B * cs.nT;
  if (nimg >= (1 << 24)) return 1;
  // split-K zeroes the WHOLE destination: only when this launch owns all of it (no destination stride / offset along t)
  // ... and only when the destination aliases neither epilogue operand: every other engine reads res[i] / mask_y[i] and then writes
  // dst[i] in the same thread, so in-place residual / mask works there; a zero-filled destination would be read back as zeros
  const bool alias = g.dst == g.res || g.dst == g.mask_y;
  const X6cPick pk = x6c_pick((int)nimg, cs.nH, cs.nW, g.Cm, ns * g.Ck,
                              (n_dst & 3) == 0 && cs.oT == 1 && cs.pT == 0 && cs.nT == g.dT && !alias, g.Ck);
  if (pk.v < 0) return 1;
  const unsigned long long sbytes = 4ull * g.B * g.Ck * g.sT * g.sH * g.sW;
  if (sbytes >= 0x7FFFFFF0ull || 4ull * (unsigned long long)n_dst >= 0x7FFFFFF0ull) return 1;
  if (dry) return 0;
  const X6cVariant& tv = kX6cVariants[pk.v];
  const int jb = pk.jb, jh = pk.jh, jw = pk.jw;
  g.nT = cs.nT; g.nH = cs.nH; g.nW = cs.nW;
  g.mT = cs.mT; g.mH = g.mW = 1; g.oT = cs.oT; g.oH = g.oW = 1; g.pT = cs.pT; g.pH = g.pW = 0;
  g.ljb = ilog2(jb); g.ljt = 0; g.ljh = ilog2(jh); g.ljw = ilog2(jw);
  g.eT = 1; g.eH = jh + 2; g.eW = jw + 2;
  g.CSl = pk.csl;
  g.bT = 0; g.bH = lo[1]; g.bW = lo[2];
  for (int i = 0; i < 9; ++i) {
    g.tap_w[i] = cs.tw[i];
    g.tap_off[i] = (cs.dh[i] - lo[1]) * g.eW + (cs.dw[i] - lo[2]);
  }
  g.ntaps = 9;
  g.nth = ceil_div(cs.nH, jh); g.ntw = ceil_div(cs.nW, jw); g.ntt = 1;
  const int ntb = ceil_div((int)nimg, jb);
  g.ksplit = pk.ksplit;
  const dim3 grid((unsigned)(ntb * g.nth * g.ntw), (unsigned)ceil_div(g.Cm, 32 * tv.TM), (unsigned)pk.ksplit);
  g.mg_ew = magic_u16(g.eW); g.mg_eh = magic_u16(g.eH);
  g.wb = wb; g.ntaps_w = ntaps_w;
  // planes 0..2 of this layer inside a stack of ntaps_w taps; the layer's own taps reach up to max(tw) + 1
  int tw_max = 0;
  for (int i = 0; i < cs.ntaps; ++i) tw_max = cs.tw[i] > tw_max ? cs.tw[i] : tw_max;
  g.wb_bytes = (2u * (unsigned)ntaps_w + (unsigned)tw_max + 1u) * (unsigned)g.Ck * (unsigned)g.CmPad * 2u;
  g.nclass = 1;
  const int tps = tv.NW == 4 ? 9 : (tv.TM == 1 ? x6c_tps1() : 3);
  const size_t lds = x6c_lds_bytes(g.CSl, tv.TM, tps);
  const int post_act = (pk.ksplit > 1 && g.act_epi != P2I_ACT_NONE) ? g.act_epi : P2I_ACT_NONE;
  const float* post_res = nullptr;
  if (pk.ksplit > 1) {       // partial sums are added: start from zero (stream-ordered in front of the kernel)
    if (p2i::memset_async(g.dst, 0, sizeof(float) * (size_t)n_dst, s) != hipSuccess) return P2I_EINVAL;
    if (post_act != P2I_ACT_NONE) { post_res = g.res; g.res = nullptr; g.act_epi = P2I_ACT_NONE; }   // act(sum + bias) + res: second pass
  }
  if (plan6) { plan6[0] = 32 * tv.TM; plan6[1] = 32 * tv.NW; plan6[2] = pk.ksplit; plan6[3] = (x6c_pc() && g.Ck >= 32) ? 4 : 8; plan6[4] = tps; plan6[5] = 7; }
  X6cGeom k{};
  k.src = g.src; k.dst = g.dst; k.wb = g.wb; k.bias = g.bias; k.res = g.res; k.mask_y = g.mask_y;
  k.wb_bytes = g.wb_bytes; k.act_epi = g.act_epi; k.mask_act = g.mask_act;
  k.B = g.B; k.Ck = g.Ck; k.Cm = g.Cm; k.CmPad = g.CmPad;
  k.sH = g.sH; k.sW = g.sW; k.dT = g.dT; k.dH = g.dH; k.dW = g.dW; k.nH = g.nH; k.nW = g.nW;
  k.ljb = g.ljb; k.ljh = g.ljh; k.ljw = g.ljw; k.eH = g.eH; k.eW = g.eW; k.CSl = g.CSl; k.bH = g.bH; k.bW = g.bW;
  k.nth = g.nth; k.ntw = g.ntw; k.mg_ew = g.mg_ew; k.mg_eh = g.mg_eh; k.ntaps_w = g.ntaps_w; k.ksplit = g.ksplit;
  for (int i = 0; i < 9; ++i) { k.tap_off[i] = g.tap_off[i]; k.tap_w[i] = g.tap_w[i]; }
  k.sT = g.sT; k.nT = g.nT; k.mT = g.mT; k.oT = g.oT; k.pT = g.pT; k.ns = ns;
  k.stagger = x6c_stagger();
  { const char* e = getenv("P2I_X6P_PRIO"); k.cons_prio = e ? atoi(e) : 0; }      // read per call (A/B runs)
  const int pc = x6c_pc();
  k.splanes = nullptr; k.plane_units = 0;
  if (g_next_splanes != nullptr) {                        // pre-split source (32-channel tiles with whole-chunk stages only)
    if (pc && g.Ck >= 32 && tv.TM == 1 && tps == 9 && pk.ksplit == 1) {
      k.splanes = static_cast<const u32x4c*>(g_next_splanes);
      k.plane_units = (unsigned)((long long)g.B * (g.Ck >> 3) * g.sT * g.sH * g.sW);
    }
    g_next_splanes = nullptr;
  }
  {
    static const int v4_on = getenv("P2I_X6C_EPI4") ? atoi(getenv("P2I_X6C_EPI4")) : 1;
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    // 64-channel tiles only (32-channel tiles: 57.9 vs 58.3 us, no gain) and never with split-K (its atomic adds want the plain
    // epilogue's 32 contiguous dwords per instruction: 92 vs 69 us on the 512-channel level)
    k.vec4_epi = v4_on && tv.TM == 2 && pk.ksplit == 1 && jw >= 4 && (cs.nW & 3) == 0 && (g.dW & 3) == 0 && al16(g.dst) && al16(g.res) && al16(g.mask_y) &&
                 (32 * 36 * 8 + 4 * tv.TM * 16 * 64) * 4 <= (int)lds - 4 * ((g.CSl + 3) & ~3);
  }
  k.sdt0 = cs.dt[0]; k.swt0 = 0;
  k.sdt1 = ns > 1 ? cs.dt[9] : 0; k.swt1 = ns > 1 ? cs.tw[9] - cs.tw[0] : 0;
  k.sdt2 = ns > 2 ? cs.dt[18] : 0; k.swt2 = ns > 2 ? cs.tw[18] - cs.tw[0] : 0;
  if (pc && g.Ck >= 32) {                                 // (a single 16-channel chunk is all prologue and epilogue: symmetric kernel)
    if (tv.TM == 2) { if (k.vec4_epi) x6p_launch<2, 3, false, true>(k, grid, lds, s); else x6p_launch<2, 3>(k, grid, lds, s); }
    else if (tv.NW == 4) { if (k.splanes) x6p_launch<1, 9, false, false, 8, 1, true>(k, grid, lds, s); else x6p_launch<1, 9, false, false, 8, 1>(k, grid, lds, s); }
    else if (tps == 9) {
      if (k.splanes && x6p_npw() == 8) x6p_launch<1, 9, false, false, 8, 2, true>(k, grid, lds, s);
      else if (x6p_npw() == 8) x6p_launch<1, 9, false, false, 8>(k, grid, lds, s);
      else x6p_launch<1, 9>(k, grid, lds, s);
    }
    else x6p_launch<1, 3>(k, grid, lds, s);
  } else if (tv.TM == 2) x6c_launch<8, 2>(k, grid, lds, s);
  else if (tps == 9) x6c_launch<8, 1, false, 9>(k, grid, lds, s);
  else x6c_launch<8, 1>(k, grid, lds, s);
  if (post_act != P2I_ACT_NONE) {
    const long long n4 = n_dst / 4;
    const long long blocks = (n4 + 255) / 256;
    P2I_LAUNCH(x6c_post_act_kernel, dim3((unsigned)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0, s, g.dst, post_res, n4, post_act);
  }
  return launch_status();
}

// ---- fused strided data gradient (stride (1,2,2), 3x3 or 3x3x3 taps, pad 1 in h and w; t stride 1): classes q = 2 pH + pW in the
// order p2i_conv_dgrad builds them; class q must hold ns * {1, 2, 2, 4}[q] taps (slice-major), every class the same extents.
// Returns 1 when the layer is not such a case.
int run_patch_gemm_x6c_fused(PatchGeom g, const ClassSpec* css, int ncls, const uint16_t* wb, int ntaps_w, int* plan6, hipStream_t s, bool dry) {
  static const int on = getenv("P2I_CONV_X6C_FUSED") ? atoi(getenv("P2I_CONV_X6C_FUSED")) : 1;
  if (!on || ncls != 4 || (!dry && wb == nullptr) || g.src_y != nullptr || (g.Ck & 15) != 0 || g.Ck < 16) return 1;
  static const int per[4] = {1, 2, 2, 4};
  const int ns = css[0].ntaps;                            // class (0,0) has one tap per t slice
  if (ns < 1 || ns > 3) return 1;
  for (int q = 0; q < 4; ++q) {
    const ClassSpec& c = css[q];
    if (c.ntaps != ns * per[q] || c.mT != 1 || c.mH != 1 || c.mW != 1 || c.oH != 2 || c.oW != 2 || c.pH != (q >> 1) || c.pW != (q & 1)) return 1;
    if (c.nT != css[0].nT || c.nH != css[0].nH || c.nW != css[0].nW || c.oT != css[0].oT || c.pT != css[0].pT) return 1;
  }
  if ((g.dH & 1) || (g.dW & 1) || css[0].nH * 2 != g.dH || css[0].nW * 2 != g.dW) return 1;
  // slot -> (class, tap within the class's slice): slots 0-3 class 3, 4-5 class 2, 6-7 class 1, 8 class 0 (X6C_CLS)
  static const int slot_cls[9] = {3, 3, 3, 3, 2, 2, 1, 1, 0}, slot_idx[9] = {0, 1, 2, 3, 0, 1, 0, 1, 0};
  int sdt[3] = {0, 0, 0}, swt[3] = {0, 0, 0};
  X6cGeom k{};
  for (int j = 0; j < ns; ++j)
    for (int sl = 0; sl < 9; ++sl) {
      const ClassSpec& c = css[slot_cls[sl]];
      const int i = j * per[slot_cls[sl]] + slot_idx[sl];      // class taps run slice-major (a outer)
      const int i0 = slot_idx[sl];
      if (c.dh[i] < 0 || c.dh[i] > 1 || c.dw[i] < 0 || c.dw[i] > 1) return 1;
      if (c.dh[i] != c.dh[i0] || c.dw[i] != c.dw[i0]) return 1;
      if (sl == 0) { sdt[j] = c.dt[i]; swt[j] = c.tw[i] - c.tw[i0]; }
      else if (c.dt[i] != sdt[j] || c.tw[i] - c.tw[i0] != swt[j]) return 1;
      if (j == 0) { k.tap_w[sl] = c.tw[i]; }
    }
  const ClassSpec& c0 = css[0];
  const long long n_dst = (long long)g.B * g.Cm * g.dT * g.dH * g.dW;
  const long long nimg = (long long)g.B * c0.nT;
  if (nimg >= (1 << 24)) return 1;
  // tile: 256 class-local positions of (b, t) images, 32 destination channels per workgroup (TM 1: four accumulator classes)
  int jb, jt, jh, jw;
  pick_tile_dims(256, (int)nimg, 1, c0.nH, c0.nW, jb, jt, jh, jw);
  const int csl = jb * (jh + 1) * (jw + 1);
  if (jt != 1 || csl > X6cTile<8>::MAXCSL) return 1;
  const bool alias = g.dst == g.res || g.dst == g.mask_y;
  const long long tiles = (long long)ceil_div((int)nimg, jb) * ceil_div(c0.nH, jh) * ceil_div(c0.nW, jw) * ceil_div(g.Cm, 32);
  const int cps = g.Ck >> 4;
  // split-K for layers with too few tiles: measured SLOWER than the f32 fused kernel on the one layer of the step it would take
  // (128 -> 256 stride 2 at B = 8: 90.6 vs 72 us, the float2 pairs become two atomics each), so it is off unless asked for
  const bool can_split = x6c_fused_ksplit() && !alias && ((ns * cps) & 1) == 0 && c0.oT == 1 && c0.pT == 0 && c0.nT == g.dT;
  const int min_wg = x6c_fused_min_wg();
  int ksplit = 1;
  if (tiles < min_wg) {
    if (can_split && 2 * tiles >= min_wg) ksplit = 2;
    else return 1;
  }
  const unsigned long long sbytes = 4ull * g.B * g.Ck * g.sT * g.sH * g.sW;
  if (sbytes >= 0x7FFFFFF0ull || 4ull * (unsigned long long)n_dst >= 0x7FFFFFF0ull) return 1;
  if (dry) return 0;
  k.src = g.src; k.dst = g.dst; k.wb = wb; k.bias = nullptr; k.res = g.res; k.mask_y = g.mask_y;
  int tw_max = 0;
  for (int q = 0; q < 4; ++q)
    for (int i = 0; i < css[q].ntaps; ++i) tw_max = css[q].tw[i] > tw_max ? css[q].tw[i] : tw_max;
  k.wb_bytes = (2u * (unsigned)ntaps_w + (unsigned)tw_max + 1u) * (unsigned)g.Ck * (unsigned)g.CmPad * 2u;
  k.act_epi = P2I_ACT_NONE; k.mask_act = g.mask_act;
  k.B = g.B; k.Ck = g.Ck; k.Cm = g.Cm; k.CmPad = g.CmPad;
  k.sH = g.sH; k.sW = g.sW; k.dT = g.dT; k.dH = g.dH; k.dW = g.dW; k.nH = c0.nH; k.nW = c0.nW;
  k.ljb = ilog2(jb); k.ljh = ilog2(jh); k.ljw = ilog2(jw);
  k.eH = jh + 1; k.eW = jw + 1; k.CSl = csl; k.bH = 0; k.bW = 0;
  k.nth = ceil_div(c0.nH, jh); k.ntw = ceil_div(c0.nW, jw);
  k.mg_ew = magic_u16(k.eW); k.mg_eh = magic_u16(k.eH);
  k.ntaps_w = ntaps_w; k.ksplit = ksplit; k.fused_atomic = ksplit > 1 ? 1 : 0; k.stagger = x6c_stagger();
  { const char* e = getenv("P2I_X6P_PRIO"); k.cons_prio = e ? atoi(e) : 0; }
  k.sT = g.sT; k.nT = c0.nT; k.mT = 1; k.oT = c0.oT; k.pT = c0.pT; k.ns = ns;
  k.sdt0 = sdt[0]; k.sdt1 = sdt[1]; k.sdt2 = sdt[2]; k.swt0 = swt[0]; k.swt1 = swt[1]; k.swt2 = swt[2];
  for (int sl = 0; sl < 9; ++sl) {
    const ClassSpec& c = css[slot_cls[sl]];
    k.tap_off[sl] = c.dh[slot_idx[sl]] * k.eW + c.dw[slot_idx[sl]];
  }
  if (ksplit > 1 && p2i::memset_async(g.dst, 0, sizeof(float) * (size_t)n_dst, s) != hipSuccess) return P2I_EINVAL;
  const dim3 grid((unsigned)(ceil_div((int)nimg, jb) * k.nth * k.ntw), (unsigned)ceil_div(g.Cm, 32), (unsigned)ksplit);
  const int tps = x6c_tps1();
  const size_t lds = x6c_lds_bytes(k.CSl, 1, tps);
  if (plan6) { plan6[0] = 32; plan6[1] = 256; plan6[2] = ksplit; plan6[3] = (x6c_pc() && tps == 9 && g.Ck >= 32) ? 4 : 8; plan6[4] = tps; plan6[5] = 8; }
  if (x6c_pc() && tps == 9 && g.Ck >= 32) x6p_launch<1, 9, true>(k, grid, lds, s);
  else if (tps == 9) x6c_launch<8, 1, true, 9>(k, grid, lds, s);
  else x6c_launch<8, 1, true>(k, grid, lds, s);
  return launch_status();
}

}  // namespace p2i

extern "C" int p2i_x6_split_batched(const float* const* wp, uint16_t* const* wb, const int* ntaps, const int* K, const int* Mpad, int n,
                                    void* stream) {
  using namespace p2i;
  P2I_REQUIRE(wp && wb && ntaps && K && Mpad && n >= 1 && n <= 24, "p2i_x6_split_batched: 1..24 tensors");
  WsplitBatch b{};
  int maxn = 0;
  for (int i = 0; i < n; ++i) {
    P2I_REQUIRE(wp[i] && wb[i] && ntaps[i] > 0 && K[i] > 0 && (K[i] & 7) == 0 && Mpad[i] > 0 && (Mpad[i] & 31) == 0, "p2i_x6_split_batched: bad tensor %d", i);
    b.wp[i] = wp[i]; b.wb[i] = reinterpret_cast<u32x4c*>(wb[i]); b.ntaps[i] = ntaps[i]; b.Ck[i] = K[i]; b.CmPad[i] = Mpad[i];
    const int tot = ntaps[i] * (K[i] >> 3) * Mpad[i];
    if (tot > maxn) maxn = tot;
  }
  const int blocks = (maxn + 255) / 256;
  P2I_LAUNCH(wsplit_batched_kernel, dim3(blocks > 1024 ? 1024 : blocks, n), dim3(256), 0, (hipStream_t)stream, b);
  return launch_status();
}

// ---- pre-split source planes (round 4 experiment; tools/presplit_probe.py).  p2i_x6_split_planes: fp32 (B, C, frames x pixels) ->
// planes; p2i_x6_next_source_planes: the next split-pipe forward / data-gradient call of this thread reads its source from them
// (32-channel tiles with whole-chunk stages; any other tile ignores them).
extern "C" int p2i_x6_split_planes(const float* x, void* planes, int B, int C, int64_t P, void* stream) {
  P2I_REQUIRE(x && planes && B > 0 && C > 0 && (C & 7) == 0 && P > 0, "split planes: C % 8 == 0");
  const long long total = (long long)B * (C >> 3) * P;
  const long long blocks = (total + 255) / 256;
  P2I_LAUNCH(p2i::split_planes_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, (hipStream_t)stream, x,
             static_cast<p2i::u32x4c*>(planes), B, C >> 3, (long long)P);
  return p2i::launch_status();
}
extern "C" int p2i_x6_next_source_planes(const void* planes) {
  p2i::g_next_splanes = planes;
  return P2I_OK;
}
