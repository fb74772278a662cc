// Gauge-point injection: nonzero(mask) compaction + brute-force 3-D 4-NN inverse-distance
// weighting (InputBlock.forward layer.py:324-361, idw_3d_knn layer.py:259-293, grid layer.py:246-256).
//
// Bit-level contract with the reference's CPU path (verified in oracle/idw_knn.c against torch):
//  * torch.cdist (r > 25) evaluates d^2 as the 5-term sgemm row  [-2x,-2y,-2z,|q|^2,1] . [px,py,pz,1,|p|^2]
//    which MKL accumulates as one k-ordered fmaf chain; sqrt(clamp_min(0)) follows.  The kernel runs
//    the identical chain (v_mul, v_fma, v_fma, v_add, v_add), so distances agree to the bit and the
//    (t-1)/(t+1) near-ties that a shared gauge mask produces resolve the same way.
//  * torch.topk (N >= 256 -> std::partial_sort) keeps a 4-entry max-heap and replaces the root
//    only on STRICTLY smaller d; the kernel replays libstdc++'s heap moves in registers.
#include "common.h"
#include <cstdlib>

namespace p2i {

// Heap entries live in scalar registers (d = distance, i = point index); SWAP/MOVE keep both in step.
#define HE_MOVE(a, b) do { a##d = b##d; a##i = b##i; } while (0)
#define HE_SET(a, vd, vi) do { a##d = (vd); a##i = (vi); } while (0)

// libstdc++ __adjust_heap(first, 0, 4, x) + __push_heap: x replaces the root of the max-heap h0..h3
#define HEAP4_REPLACE_ROOT(xd, xi)                                                      \
  do {                                                                                  \
    if (h2d < h1d) { /* larger child is 1 (ties pick 2): hole moves 0 -> 1 -> 3 */      \
      HE_MOVE(h0, h1); HE_MOVE(h1, h3);                                                 \
      if (h1d < (xd)) {                                                                 \
        HE_MOVE(h3, h1);                                                                \
        if (h0d < (xd)) { HE_MOVE(h1, h0); HE_SET(h0, xd, xi); } else HE_SET(h1, xd, xi); \
      } else HE_SET(h3, xd, xi);                                                        \
    } else {         /* hole moves 0 -> 2 */                                            \
      HE_MOVE(h0, h2);                                                                  \
      if (h0d < (xd)) { HE_MOVE(h2, h0); HE_SET(h0, xd, xi); } else HE_SET(h2, xd, xi); \
    }                                                                                   \
  } while (0)

// libstdc++ __sort_heap on the 4-entry max-heap -> ascending h0..h3
#define IDW_SORT_HEAP()                                                                                                          \
  do {                                                                                                                            \
    {  /* len 4 -> 3 */                                                                                                           \
      const float xd = h3d; const int xi = h3i; HE_MOVE(h3, h0);                                                                  \
      if (h2d < h1d) { HE_MOVE(h0, h1); if (h0d < xd) { HE_MOVE(h1, h0); HE_SET(h0, xd, xi); } else HE_SET(h1, xd, xi); }         \
      else { HE_MOVE(h0, h2); if (h0d < xd) { HE_MOVE(h2, h0); HE_SET(h0, xd, xi); } else HE_SET(h2, xd, xi); }                   \
    }                                                                                                                             \
    {  /* len 3 -> 2 */                                                                                                           \
      const float xd = h2d; const int xi = h2i; HE_MOVE(h2, h0); HE_MOVE(h0, h1);                                                 \
      if (h0d < xd) { HE_MOVE(h1, h0); HE_SET(h0, xd, xi); } else HE_SET(h1, xd, xi);                                             \
    }                                                                                                                             \
    {  /* len 2 -> 1 */                                                                                                           \
      const float xd = h1d; const int xi = h1i; HE_MOVE(h1, h0); HE_SET(h0, xd, xi);                                              \
    }                                                                                                                             \
  } while (0)

// Wave-uniform values, made so for the compiler: loop bounds and point indices built from them live in SGPRs, the points arrive
// by scalar loads (s_load_dwordx4 .. x16) and the loop is not exec-masked.  (Round 3: without these the scan loops issued one
// per-lane global_load_dwordx4 + s_waitcnt vmcnt(0) per point -- a memory round trip per distance.)
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float unif(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
  return uni(v);
}
__device__ __forceinline__ int wave_min_i(int v) { return -wave_max_i(-v); }
__device__ __forceinline__ float wave_max_u(float v) { return unif(wave_max(v)); }
__device__ __forceinline__ float wave_min_u(float v) { return -unif(wave_max(-v)); }

// ---- compaction of mask > 0 in (t, y, x) order
__global__ void idw_count_kernel(const float* __restrict__ mask, int32_t* frame_count, int HW) {
  __shared__ float red[16];
  const float* m = mask + (size_t)blockIdx.x * HW;
  float c = 0.f;
  if ((HW & 3) == 0 && (reinterpret_cast<uintptr_t>(mask) & 15) == 0) {       // (every frame is 16-byte aligned then)
    const float4* m4 = reinterpret_cast<const float4*>(m);
    for (int i = threadIdx.x; i < (HW >> 2); i += blockDim.x) {
      const float4 v = m4[i];
      c += (v.x > 0.f ? 1.f : 0.f) + (v.y > 0.f ? 1.f : 0.f) + (v.z > 0.f ? 1.f : 0.f) + (v.w > 0.f ? 1.f : 0.f);
    }
  } else {
    for (int i = threadIdx.x; i < HW; i += blockDim.x) c += (m[i] > 0.f) ? 1.f : 0.f;
  }
  c = block_sum(c, red);
  if (threadIdx.x == 0) frame_count[blockIdx.x] = (int)c;
}
constexpr int IDW_CT = 1024;                          // threads of the compaction kernels (a frame per workgroup: latency-bound)
__global__ __launch_bounds__(IDW_CT) void idw_compact_kernel(const float* __restrict__ mask, const int32_t* __restrict__ frame_count,
                                                         const float* __restrict__ gx, const float* __restrict__ gy,
                                                         const float* __restrict__ gz, int32_t* pt_pos, int32_t* pt_count,
                                                         int32_t* row_start, float* pt_xyzn, int32_t* amb, int T, int H, int W) {
  __shared__ int wsum[IDW_CT / 64];
  __shared__ int s_base;
  const int bt = blockIdx.x, b = bt / T, t = bt % T, HW = H * W, Q = T * HW;
  if (threadIdx.x == 0) {
    int off = 0;
    for (int k = 0; k < t; ++k) off += frame_count[b * T + k];
    s_base = off;
    if (t == T - 1) pt_count[b] = off + frame_count[bt];
    if (t == 0 && amb) {
      const size_t per = (size_t)T * H * W + 1 + (T * H * W + 255) / 256, nB = gridDim.x / T;
      amb[(size_t)b * per] = 0;                                        // the count of undecided voxels (idw_knn_kernel<1>)
      amb[nB * per + (size_t)b * ((size_t)T * H * W + 1)] = 0;         // ... and of those the fixed-bound phase of idw_knn_kernel<2> leaves to the replay
    }
  }
  __syncthreads();
  int base = s_base;
  const float* m = mask + (size_t)bt * HW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float Wd = (float)max(W - 1, 1), Hd = (float)max(H - 1, 1), Td = (float)max(T - 1, 1);
  for (int i0 = 0; i0 < HW; i0 += IDW_CT) {
    const int i = i0 + threadIdx.x;
    const bool on = i < HW && m[i] > 0.f;
    const unsigned long long bal = __ballot(on);
    const int rank = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int woff = 0, tot = 0;
    for (int k = 0; k < IDW_CT / 64; ++k) { if (k < wave) woff += wsum[k]; tot += wsum[k]; }
    // row_start[b][t][y] = index (within the sample's point list) of the first point of frame t in a row >= y
    if (i < HW && (i % W) == 0) row_start[(size_t)bt * (H + 1) + i / W] = base + woff + rank;
    if (on) {
      const int j = base + woff + rank;
      const int y = i / W, x = i - y * W;
      pt_pos[(size_t)b * Q + j] = t * HW + i;
      // points: tx/(W-1), ty/(H-1), tz/(D-1) (layer.py:335-342); |p|^2 = (x^2 + y^2) + z^2 unfused
      const float px = __fdiv_rn((float)x, Wd), py = __fdiv_rn((float)y, Hd), pz = __fdiv_rn((float)t, Td);
      const float pn = __fadd_rn(__fadd_rn(__fmul_rn(px, px), __fmul_rn(py, py)), __fmul_rn(pz, pz));
      *reinterpret_cast<float4*>(pt_xyzn + ((size_t)b * Q + j) * 4) = make_float4(px, py, pz, pn);
    }
    base += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) row_start[(size_t)bt * (H + 1) + H] = base;
  (void)gx; (void)gy; (void)gz;
}

// one point against a lane's selection.  (Macros on plain locals, not lambdas: with the heap captured by reference the compiler
// sinks the conditional heap moves into stores through selected ADDRESSES and the heap ends up in scratch memory.)
// MODE 0 / 2: the reference's heap on distances; r2 is the fast-reject bound on d^2: r2 = fl(fl(r*r)*(1+2^-20)) > r^2 exactly, so
// c2 >= r2 implies sqrt_rn(c2) >= r (no insert, as the strict comparison demands); below r2 the exact d-space test decides.
#define IDW_CONSIDER_HEAP(c2v, jv)                                                      \
  do {                                                                                  \
    if ((c2v) < r2) {                                                                   \
      const float dc_ = sqrtf(fmaxf((c2v), 0.f));                                       \
      if (dc_ < h0d) {                                                                  \
        HEAP4_REPLACE_ROOT(dc_, (jv));                                                  \
        r2 = h0d * h0d * 1.000001f + 1e-30f;                                            \
      }                                                                                 \
    }                                                                                   \
  } while (0)
// MODE 1: h0 <= h1 <= h2 <= h3 are the four smallest SQUARED distances so far and r2 (the bound a point must beat to matter) the
// fifth: d = sqrt_rn(max(d^2, 0)) is monotone, so the order by d^2 refines the order by d and the five roots are taken once, at
// the end.
#define IDW_CONSIDER_SORTED(c2v, jv)                                                    \
  do {                                                                                  \
    if ((c2v) < r2) {                                                                   \
      const float cc_ = (c2v);                                                          \
      const bool b3_ = cc_ < h3d, b2_ = cc_ < h2d, b1_ = cc_ < h1d, b0_ = cc_ < h0d;    \
      r2 = b3_ ? h3d : cc_;                                                             \
      h3d = b2_ ? h2d : (b3_ ? cc_ : h3d); h3i = b2_ ? h2i : (b3_ ? (jv) : h3i);        \
      h2d = b1_ ? h1d : (b2_ ? cc_ : h2d); h2i = b1_ ? h1i : (b2_ ? (jv) : h2i);        \
      h1d = b0_ ? h0d : (b1_ ? cc_ : h1d); h1i = b0_ ? h0i : (b1_ ? (jv) : h1i);        \
      h0d = b0_ ? cc_ : h0d; h0i = b0_ ? (jv) : h0i;                                    \
    }                                                                                   \
  } while (0)

// MODE 2, first phase (round 4): the voxel's 4th distance D is known from MODE 1, so the reference's result can be DECIDED from the points
// with d <= D alone, met in index order (S = that subsequence).  The reference's max-heap always holds the four smallest VALUES seen
// so far (a new point replaces the root on strictly smaller d), hence:
//   * the first four members of S are inserted when they arrive (fewer than four points <= D precede them: the root is a point > D)
//     and each evicts a point > D;
//   * a later member with d == D is never inserted (four points <= D are in the heap: root <= D, and the comparison is strict);
//   * a later member with d < D is inserted and evicts the root, a point with d == D.  With exactly ONE such point in the heap it is
//     that one; with two or more, which of them sits at the root depends on the heap's whole history -- only then is the voxel left
//     to the exact replay below (flag amb_; also when anything contradicts D being the 4th distance).
// Points with d > D never touch the outcome (they are evicted before any point <= D is, the root being the largest), so this scan
// runs with the FIXED bound D from its first point -- ~1/7 of the points the replay's shrinking root lets through, no heap moves.
// s0..s3: the current set (any order), ns_ members of S met, nd_ how many of the set have d == D.
#define IDW_CONSIDER_FIXED(c2v, jv)                                                     \
  do {                                                                                  \
    if ((c2v) < r2) {                                                                   \
      const float dc_ = sqrtf(fmaxf((c2v), 0.f));                                       \
      if (dc_ <= dq_) {                                                                 \
        const bool isd_ = dc_ == dq_;                                                   \
        if (ns_ < 4) {                                                                  \
          if (ns_ == 0) HE_SET(h0, dc_, (jv)); else if (ns_ == 1) HE_SET(h1, dc_, (jv)); \
          else if (ns_ == 2) HE_SET(h2, dc_, (jv)); else HE_SET(h3, dc_, (jv));         \
          ++ns_; nd_ += isd_ ? 1 : 0;                                                   \
        } else if (!isd_) {                                                             \
          if (nd_ == 1) {                                                               \
            if (h0d == dq_) HE_SET(h0, dc_, (jv)); else if (h1d == dq_) HE_SET(h1, dc_, (jv)); \
            else if (h2d == dq_) HE_SET(h2, dc_, (jv)); else HE_SET(h3, dc_, (jv));     \
            nd_ = 0;                                                                    \
          } else amb_ = true;                                                           \
        }                                                                               \
      }                                                                                 \
    }                                                                                   \
  } while (0)

// ---- 4-NN + IDW.  thread = one query voxel, workgroup = NT voxels.  The workgroup walks frames / row ranges together (bounds are
// workgroup-uniform: largest root, smallest |dz|, row span), keeps a window of the point list in LDS (loaded with all threads once
// per 4 NT points -- a gauge mask's frames t-2 .. t+2 are 400 points -- so it waits for global memory once or twice in all) and
// every lane reads the points back as broadcasts, four per trip.  (Scalar loads of the points, tried first in round 3, miss the
// 16 KB scalar cache -- eight samples' point lists are 160 KB -- and ran 3x slower than even per-lane global loads.)
//
// MODE 0 -- the reference's scan replayed for every voxel: points in index order (frame-major, then row-major), a 4-entry max-heap
//   whose root is replaced on STRICTLY smaller d (libstdc++ partial_sort as torch.topk uses it), exact under ties.
//   Exact pruning: whole frames / row ranges that provably cannot beat the CURRENT root are skipped: a skipped point has |dz| or
//   |dy| (hence its computed distance, up to the 2e-6 slack that covers the fp32 cancellation of the |a|^2+|b|^2-2ab chain) above
//   the root at its turn, so std::partial_sort would not have inserted it either and the heap evolves identically.  (A lane meeting
//   a point it alone could have skipped rejects it like any other.)
// MODE 1 -- fast path.  The reference's result depends on the ORDER of its scan only where distances tie: with d1 <= d2 <= d3 <= d4
//   the four smallest computed distances and d5 the smallest of all others,
//   * d4 < d5: the SET of selected points is the four nearest whatever the order (a max-heap that replaces its root on strictly
//     smaller d ends with exactly those); ties AMONG the four only permute equal weights in the four-term output sum (<= 1 ulp);
//   * d4 == d5: which of the tied points is kept depends on the heap's history -- such voxels (8 % with 79 gauges shared by all
//     frames: the same gauge in frames t-1 and t+1 is equidistant from every voxel of frame t, and the fp32 chain rounds both to
//     the same value often enough) are listed in `amb` and left to MODE 2.
//   Free of the scan order, this pass starts where the neighbours are -- own frame, rows around the voxels, then the rest of the
//   frame, then frames outwards in both directions until |dz| alone exceeds the workgroup's largest 4th distance -- with the same
//   pruning bounds; it evaluates ~1/7 of the points the index-order scan has to touch (there every frame in front of the voxel's
//   own improves the heap).  d5 is taken over evaluated points only: a pruned point is farther than the 4th at the time.
// MODE 2 -- the voxels MODE 1 listed, in the order (MODE 1 workgroup, rank) so that a workgroup's voxels are neighbours and its pruning
//   bounds stay tight: decided from the points within the known 4th distance (IDW_CONSIDER_FIXED above) where that is possible; the
//   rest go to a second, flat list and a second MODE 2 launch (flags 3) replays MODE 0's scan for those.  (flags 1: the replay for every
//   listed voxel, round 3's pass; A/B, tests.)
// amb2, behind the B samples' amb blocks, per sample: [0] number of voxels left to the replay; [1 ...] the voxels.
// amb, per sample (nblk1 = MODE 1's workgroups): [0] number of undecided voxels; [1 + 256 g ...] those of workgroup g;
//   [1 + Q + g] how many those are, turned into their exclusive prefix sums by idw_prefix_kernel.
constexpr int IDW_MAX_BLK = 1 << 20;                  // (prefix sums of more workgroups than this: single-pass scan instead)
template <int MODE, int NT>
__global__ __launch_bounds__(NT) void idw_knn_kernel(const float* __restrict__ vals, const float* __restrict__ gx,
                                                    const float* __restrict__ gy, const float* __restrict__ gz,
                                                    const int32_t* __restrict__ pt_pos, const int32_t* __restrict__ pt_count,
                                                    const int32_t* __restrict__ row_start, const float4* __restrict__ pt_xyzn, float* out, int32_t* sel_idx,
                                                    float* sel_w, int32_t* amb, int nblk1, int T, int H, int W, float tau, int flags) {
  constexpr int NW = NT / 64, WIN = 4 * NT;
  __shared__ float4 spts[WIN];
  __shared__ float sredf[2 * NW];
  __shared__ int sredi[3 * NW];
  const int b = blockIdx.y, HW = H * W, Q = T * HW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q_raw = blockIdx.x * NT + tid;
  int32_t* ambs = amb + (size_t)b * (Q + 1 + nblk1);
  // second list (MODE 2, flags & 2): the voxels the fixed-bound phase left undecided, flat: [0] count, [1 ..] voxels
  int32_t* amb2s = amb + (size_t)gridDim.y * (Q + 1 + nblk1) + (size_t)b * (Q + 1);
  const bool flat = MODE == 2 && (flags & 2);
  const int nq = MODE == 2 ? uni(flat ? amb2s[0] : ambs[0]) : Q;
  if (MODE == 2 && (int)(blockIdx.x * NT) >= nq) return;
  const bool active = q_raw < nq;                      // inactive lanes shadow the last voxel: the reductions below need every lane
  int q = active ? q_raw : Q - 1;                      // to hold defined values
  if constexpr (MODE == 2) {
    const int* pre = ambs + 1 + Q;                     // exclusive prefix sums of the per-workgroup counts
    const int d = active ? q_raw : nq - 1;
    if (flat) {
      q = amb2s[1 + d];
    } else {
      int lo = 0, hi = nblk1;                          // largest g with pre[g] <= d
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (pre[mid] <= d) lo = mid; else hi = mid; }
      q = ambs[1 + lo * 256 + (d - pre[lo])];
    }
  }
  const int N = uni(pt_count[b]);
  const size_t qo = (size_t)b * Q + q;
  if (N < 4) {       // N == 0: zeros (layer.py:330-332); 0 < N < 4: the reference raises in topk, here zeros as well (p2i_hip.h)
    if (!active) return;
    out[qo] = 0.f;
    if (sel_idx) {
      *reinterpret_cast<int4*>(sel_idx + qo * 4) = make_int4(0, 0, 0, 0);
      *reinterpret_cast<float4*>(sel_w + qo * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return;
  }
  const int t = q / HW, rem = q - t * HW, y = rem / W, x = rem - y * W;
  const float qx = gx[x], qy = gy[y], qz = gz[t];
  const float a0 = -2.f * qx, a1 = -2.f * qy, a2 = -2.f * qz;
  const float n1 = __fadd_rn(__fadd_rn(__fmul_rn(qx, qx), __fmul_rn(qy, qy)), __fmul_rn(qz, qz));
  const float4* pts = pt_xyzn + (size_t)b * Q;
  const float INF = __builtin_inff();

  auto dist2 = [&](const float4 p) __attribute__((always_inline)) {
    float acc = __fmul_rn(a0, p.x);
    acc = __fmaf_rn(a1, p.y, acc);
    acc = __fmaf_rn(a2, p.z, acc);
    acc = __fadd_rn(acc, n1);       // fma(n1, 1, acc)
    acc = __fadd_rn(acc, p.w);      // fma(1, |p|^2, acc)
    return acc;
  };
  // MODE 0 / 2: h0..h3 = libstdc++'s max-heap of distances (h0 the root); MODE 1: squared distances, ascending
  float h0d = INF, h1d = INF, h2d = INF, h3d = INF;
  int h0i = 0, h1i = 1, h2i = 2, h3i = 3;
  float r2 = INF;
  // MODE 2's first phase decides from the fixed bound D (stashed in out[] by MODE 1); the heap of MODE 0 / the fallback is seeded below
  const bool fixed_first = MODE == 2 && !(flags & 1);          // (flags & 1: the reference's scan replayed for every voxel of the list)
  const float dq_ = fixed_first ? out[qo] : 0.f;
  if (fixed_first) r2 = dq_ * dq_ * 1.000001f + 1e-30f;          // c2 >= r2 implies sqrt_rn(c2) > D (see IDW_CONSIDER_HEAP)
#define IDW_SEED_HEAP()                                                                                                   \
  do {                                                                                                                    \
    h0i = 0; h1i = 1; h2i = 2; h3i = 3;                                                                                   \
    h0d = sqrtf(fmaxf(dist2(pts[0]), 0.f)); h1d = sqrtf(fmaxf(dist2(pts[1]), 0.f));                                       \
    h2d = sqrtf(fmaxf(dist2(pts[2]), 0.f)); h3d = sqrtf(fmaxf(dist2(pts[3]), 0.f));                                       \
    /* __make_heap: parent = 1 (swap with child 3 unless child < parent), then parent = 0 */                              \
    if (!(h3d < h1d)) { const float td = h1d; const int ti = h1i; HE_MOVE(h1, h3); HE_SET(h3, td, ti); }                  \
    const float xd = h0d; const int xi = h0i;                                                                             \
    HEAP4_REPLACE_ROOT(xd, xi);                                                                                           \
    r2 = h0d * h0d * 1.000001f + 1e-30f;                                                                                  \
  } while (0)
  if (MODE != 1 && !fixed_first) IDW_SEED_HEAP();
  // workgroup-uniform reductions (results in SGPRs; two barriers each when the workgroup has more than one wave)
  auto wg_minmax_i = [&](int vmin, int vmax, int& omin, int& omax) __attribute__((always_inline)) {
    vmin = wave_min_i(vmin); vmax = wave_max_i(vmax);
    if (NW > 1) {
      __syncthreads();
      if (lane == 0) { sredi[wave] = vmin; sredi[NW + wave] = vmax; }
      __syncthreads();
      for (int k = 0; k < NW; ++k) { vmin = min(vmin, sredi[k]); vmax = max(vmax, sredi[NW + k]); }
    }
    omin = uni(vmin); omax = uni(vmax);
  };
  auto wg_max_min_f = [&](float vmax, float vmin, float& omax, float& omin) __attribute__((always_inline)) {
    vmax = wave_max_u(vmax); vmin = wave_min_u(vmin);
    if (NW > 1) {
      __syncthreads();
      if (lane == 0) { sredf[wave] = vmax; sredf[NW + wave] = vmin; }
      __syncthreads();
      for (int k = 0; k < NW; ++k) { vmax = fmaxf(vmax, sredf[k]); vmin = fminf(vmin, sredf[NW + k]); }
    }
    omax = unif(vmax); omin = unif(vmin);
  };
  const int* rs = row_start + (size_t)b * T * (H + 1);
  // points [lo, hi) of the list, in index order, read from the LDS window as broadcasts (IDW_SCAN, a macro for the reason above)
#define IDW_CONSIDER(c2v, jv) do { if (MODE == 1) IDW_CONSIDER_SORTED(c2v, jv); else IDW_CONSIDER_HEAP(c2v, jv); } while (0)
#define IDW_LOAD_WINDOW(start)                                                                                                  \
  do {                                                                                                                         \
    __syncthreads();                                   /* the previous window has been read */                                 \
    w_lo = (start); w_hi = min(N, w_lo + WIN);                                                                                 \
    for (int i_ = tid; i_ < w_hi - w_lo; i_ += NT) spts[i_] = pts[w_lo + i_];                                                  \
    __syncthreads();                                                                                                           \
  } while (0)
#define IDW_SCAN(lo_expr, hi_expr) IDW_SCAN_WITH(IDW_CONSIDER, lo_expr, hi_expr)
#define IDW_SCAN_WITH(CONS, lo_expr, hi_expr)                                                                                  \
  do {                                                                                                                         \
    const int hi_ = (hi_expr);                                                                                                 \
    int j_ = (lo_expr);                                                                                                        \
    while (j_ < hi_) {                                                                                                         \
      if (j_ < w_lo || j_ >= w_hi) IDW_LOAD_WINDOW(j_);                                                                        \
      const int e_ = min(hi_, w_hi);                                                                                           \
      const float4* sp_ = spts - w_lo;                                                                                         \
      for (; j_ + 4 <= e_; j_ += 4) {                                                                                          \
        const float4 p0_ = sp_[j_], p1_ = sp_[j_ + 1], p2_ = sp_[j_ + 2], p3_ = sp_[j_ + 3];                                   \
        const float c0_ = dist2(p0_), c1_ = dist2(p1_), c2_ = dist2(p2_), c3_ = dist2(p3_);                                    \
        if (fminf(fminf(c0_, c1_), fminf(c2_, c3_)) < r2) {                                                                    \
          CONS(c0_, j_); CONS(c1_, j_ + 1); CONS(c2_, j_ + 2); CONS(c3_, j_ + 3);                                              \
        }                                                                                                                      \
      }                                                                                                                        \
      for (; j_ < e_; ++j_) { const float c0_ = dist2(sp_[j_]); CONS(c0_, j_); }                                               \
    }                                                                                                                          \
  } while (0)
  int w_lo = 0, w_hi = 0;                              // points [w_lo, w_hi) of the list are in LDS (loaded at the first miss, kept across scans)
  int tlo, thi, ylo, yhi;
  wg_minmax_i(t, t, tlo, thi);
  wg_minmax_i(y, y, ylo, yhi);
  if (MODE == 1 && tlo != thi) { ylo = 0; yhi = H - 1; }     // (a workgroup across a frame end: whole frames)
  // rows of frame f that can hold a point within the largest root of the workgroup (false: none -- the frame is out of reach)
  auto reach = [&](const int f, int& ya, int& yb) __attribute__((always_inline)) {
    float r2w, dzf;
    // the bound on d^2 a point has to beat: MODE 0 / 2 the root's r2; MODE 1 the 4th squared distance (with the same relative slack)
    wg_max_min_f(MODE == 1 ? h3d * 1.000001f + 1e-30f : r2, fabsf(qz - gz[f]), r2w, dzf);     // gz[f] == f/(T-1) as the points carry it (to 1 ulp)
    r2w += 2e-6f; dzf -= 1e-6f;
    const float dz2 = dzf > 0.f ? dzf * dzf : 0.f;
    if (dz2 > r2w) return false;
    const float ry = sqrtf(r2w - dz2);
    int kk = ry < 2.f ? (int)(ry * (float)max(H - 1, 1)) + 2 : H;      // rows that can hold a point within ry (+ slack)
    if (kk > H) kk = H;
    ya = max(0, ylo - kk); yb = min(H, yhi + kk + 1);
    return true;
  };
  auto row0 = [&](const int f, const int yy) __attribute__((always_inline)) { return uni(rs[f * (H + 1) + yy]); };
  if (fixed_first) {
    int ns_ = 0, nd_ = 0;
    bool amb_ = false;
    for (int f = 0; f < T; ++f) {
      const int fs = row0(f, 0), fe = row0(f, H);
      if (fs == fe) continue;
      int ya, yb;
      if (!reach(f, ya, yb)) continue;                 // the whole frame is farther than every lane's D
      IDW_SCAN_WITH(IDW_CONSIDER_FIXED, row0(f, ya), row0(f, yb));
    }
    // (fewer than four points within D, or none AT D: D was not the 4th distance -- cannot happen; left to the replay if it does)
    amb_ = active && (amb_ || ns_ < 4 || nd_ < 1);
    // voxels the fixed bound cannot decide: appended to the sample's second list (a block per workgroup, reserved with one atomic: the
    // ORDER of the blocks varies from run to run, no voxel's result depends on the workgroup that replays it -- the pruning is exact)
    const unsigned long long bal = __ballot(amb_);
    __syncthreads();
    if (lane == 0) sredi[2 * NW + wave] = __popcll(bal);
    __syncthreads();
    if (tid == 0) {
      int tot = 0;
      for (int k = 0; k < NW; ++k) tot += sredi[2 * NW + k];
      sredi[0] = tot ? atomicAdd(amb2s, tot) : 0;
    }
    __syncthreads();
    if (amb_) {
      int off = sredi[0] + __popcll(bal & ((1ull << lane) - 1ull));
      for (int k = 0; k < wave; ++k) off += sredi[2 * NW + k];
      amb2s[1 + off] = q;
      return;
    }
    // ascending by distance (the order among equal distances permutes equal weights only): 5-exchange network
#define IDW_CSWAP(a, b) do { if (b##d < a##d) { const float td_ = a##d; const int ti_ = a##i; HE_MOVE(a, b); HE_SET(b, td_, ti_); } } while (0)
    IDW_CSWAP(h0, h1); IDW_CSWAP(h2, h3); IDW_CSWAP(h0, h2); IDW_CSWAP(h1, h3); IDW_CSWAP(h1, h2);
#undef IDW_CSWAP
  }
  if (MODE != 1 && !fixed_first) {
    for (int f = 0; f < T; ++f) {
      const int fs = row0(f, 0), fe = row0(f, H);
      if (fe <= 4 || fs == fe) continue;               // points 0..3 seeded the heap
      int ya, yb;
      if (!reach(f, ya, yb)) continue;                 // the whole frame is farther than every lane's root
      IDW_SCAN(max(row0(f, ya), 4), row0(f, yb));
    }
    IDW_SORT_HEAP();
  } else if (MODE == 1) {
    {   // frames tlo-2 .. thi+2 are one index range: in one window if they fit
      const int p0 = row0(max(tlo - 2, 0), 0), p1 = row0(min(thi + 2, T - 1), H);
      if (p1 - p0 <= WIN && p1 > p0) IDW_LOAD_WINDOW(p0);
    }
    for (int f = tlo; f <= thi; ++f) {
      // a band of rows around the voxels with at least 8 points seeds the selection; the rest of the frame is then pruned by it
      int band = 2, na, nb;
      for (;;) {
        na = max(0, ylo - band); nb = min(H, yhi + band + 1);
        if (row0(f, nb) - row0(f, na) >= 8 || (na == 0 && nb == H)) break;
        band *= 2;
      }
      IDW_SCAN(row0(f, na), row0(f, nb));
      int ya, yb;
      if (reach(f, ya, yb)) {
        if (ya < na) IDW_SCAN(row0(f, ya), row0(f, na));
        if (yb > nb) IDW_SCAN(row0(f, nb), row0(f, yb));
      }
    }
    bool dn = tlo > 0, up = thi < T - 1;
    for (int k = 1; dn || up; ++k) {
      int ya, yb;
      if (dn) {
        const int f = tlo - k;
        if (reach(f, ya, yb)) IDW_SCAN(row0(f, ya), row0(f, yb)); else dn = false;      // farther frames are farther still, and roots only shrink
        if (f == 0) dn = false;
      }
      if (up) {
        const int f = thi + k;
        if (reach(f, ya, yb)) IDW_SCAN(row0(f, ya), row0(f, yb)); else up = false;
        if (f == T - 1) up = false;
      }
    }
    // distances of the four (and of the fifth: r2); N >= 4, so every voxel has met at least four points and h3d is finite
    h0d = sqrtf(fmaxf(h0d, 0.f)); h1d = sqrtf(fmaxf(h1d, 0.f)); h2d = sqrtf(fmaxf(h2d, 0.f)); h3d = sqrtf(fmaxf(h3d, 0.f));
    const bool undecided = active && sqrtf(fmaxf(r2, 0.f)) == h3d;
    // undecided voxels: into this workgroup's 256 slots, in voxel order
    const unsigned long long bal = __ballot(undecided);
    __syncthreads();
    if (lane == 0) sredi[2 * NW + wave] = __popcll(bal);
    __syncthreads();
    if (tid == 0) {
      int tot = 0;
      for (int k = 0; k < NW; ++k) tot += sredi[2 * NW + k];
      ambs[1 + Q + blockIdx.x] = tot;
      if (tot) atomicAdd(ambs, tot);
    }
    if (undecided) {
      int off = __popcll(bal & ((1ull << lane) - 1ull));
      for (int k = 0; k < wave; ++k) off += sredi[2 * NW + k];
      ambs[1 + blockIdx.x * NT + off] = q;
      out[qo] = h3d;                                   // the 4th distance D: MODE 2 decides from the points within it
      return;
    }
  }
  // weights (layer.py:283-290): inv = 1/(d+tau); w = inv*inv; w /= (sum + 1e-12); out = sum(v*w)
  if (!active) return;
  const float i0 = __fdiv_rn(1.f, h0d + tau), i1 = __fdiv_rn(1.f, h1d + tau);
  const float i2 = __fdiv_rn(1.f, h2d + tau), i3 = __fdiv_rn(1.f, h3d + tau);
  float w0 = i0 * i0, w1 = i1 * i1, w2 = i2 * i2, w3 = i3 * i3;
  const float ws = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(w0, w1), w2), w3), 1e-12f);
  w0 = __fdiv_rn(w0, ws); w1 = __fdiv_rn(w1, ws); w2 = __fdiv_rn(w2, ws); w3 = __fdiv_rn(w3, ws);
  const int32_t* pp = pt_pos + (size_t)b * Q;
  const float* vb = vals + (size_t)b * Q;
  const float v0 = vb[pp[h0i]], v1 = vb[pp[h1i]], v2 = vb[pp[h2i]], v3 = vb[pp[h3i]];
  out[qo] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(v0, w0), __fmul_rn(v1, w1)), __fmul_rn(v2, w2)), __fmul_rn(v3, w3));
  if (sel_idx) {
    *reinterpret_cast<int4*>(sel_idx + qo * 4) = make_int4(h0i, h1i, h2i, h3i);      // POINT indices (pt_pos maps them to voxels)
    *reinterpret_cast<float4*>(sel_w + qo * 4) = make_float4(w0, w1, w2, w3);
  }
}
#undef IDW_SCAN
#undef IDW_SCAN_WITH
#undef IDW_SEED_HEAP
#undef IDW_LOAD_WINDOW
#undef IDW_CONSIDER

// The exact replay for the FEW voxels MODE 2's fixed-bound phase leaves undecided (~5 % of the listed ones: ~1 000 per sample, scattered
// over the volume).  The cooperative workgroup scan above walks the UNION of its lanes' frame / row ranges between barriers: for 256
// scattered voxels that is most of every frame, and the pass took as long for 1 000 voxels per sample as for 21 000 neighbouring ones
// (280-550 us, rocprofv3, round 4).  (A wave per voxel -- 64 points evaluated at once, candidates visited in index order through a
// ballot -- was tried first: the few hundred heap updates of a voxel form one dependent chain of ~100 cycles each, 250-400 us.)
// Here every LANE replays its own voxel over its OWN ranges (reach()'s bounds from its own root), reading the points from an LDS
// copy of the sample's list (when it fits: IDW_RW_PCAP points; 79 gauges x 16 frames are 1 264, an `sti` block-10 mask 2 704) at
// per-lane addresses: no barriers inside the scan, 64 independent chains per wave.  Same comparisons in the same order as MODE 0.
constexpr int IDW_RW_PCAP = 3072;
__global__ __launch_bounds__(64) void idw_replay_lane_kernel(const float* __restrict__ vals, const float* __restrict__ gx,
                                                            const float* __restrict__ gy, const float* __restrict__ gz,
                                                            const int32_t* __restrict__ pt_pos, const int32_t* __restrict__ pt_count,
                                                            const int32_t* __restrict__ row_start, const float4* __restrict__ pt_xyzn,
                                                            float* out, int32_t* sel_idx, float* sel_w, const int32_t* __restrict__ amb,
                                                            int nblk1, int T, int H, int W, float tau, int use_lds) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rwsm[];     // [IDW_RW_PCAP] points, then [T (H + 1)] row starts (0 bytes: global)
  const int b = blockIdx.y, HW = H * W, Q = T * HW;
  const int32_t* amb2s = amb + (size_t)gridDim.y * (Q + 1 + nblk1) + (size_t)b * (Q + 1);
  const int nq = uni(amb2s[0]);
  const int N = uni(pt_count[b]);
  if (N < 4 || (int)(blockIdx.x * 64) >= nq) return;
  const float4* gpts = pt_xyzn + (size_t)b * Q;
  const int* grs = row_start + (size_t)b * T * (H + 1);
  const bool in_lds = use_lds && N <= IDW_RW_PCAP;           // (workgroup-uniform)
  const float4* sp = reinterpret_cast<const float4*>(rwsm);
  const int* sr = reinterpret_cast<const int*>(rwsm + sizeof(float4) * IDW_RW_PCAP);
  if (in_lds) {
    float4* wp = reinterpret_cast<float4*>(rwsm);
    int* wr = reinterpret_cast<int*>(rwsm + sizeof(float4) * IDW_RW_PCAP);
    for (int i = threadIdx.x; i < N; i += 64) wp[i] = gpts[i];
    for (int i = threadIdx.x; i < T * (H + 1); i += 64) wr[i] = grs[i];
    __syncthreads();
  }
  // (grid-stride: the grid covers Q / 8 voxels per sample at once; a mask whose ties leave more than that loops)
  for (int v = blockIdx.x * 64 + threadIdx.x; v < nq; v += gridDim.x * 64) {
  const int q = amb2s[1 + v];
  const int t = q / HW, rem = q - t * HW, y = rem / W, x = rem - y * W;
  const float qx = gx[x], qy = gy[y], qz = gz[t];
  const float a0 = -2.f * qx, a1 = -2.f * qy, a2 = -2.f * qz;
  const float n1 = __fadd_rn(__fadd_rn(__fmul_rn(qx, qx), __fmul_rn(qy, qy)), __fmul_rn(qz, qz));
  auto dist2 = [&](const float4 p) __attribute__((always_inline)) {
    float acc = __fmul_rn(a0, p.x);
    acc = __fmaf_rn(a1, p.y, acc);
    acc = __fmaf_rn(a2, p.z, acc);
    acc = __fadd_rn(acc, n1);
    acc = __fadd_rn(acc, p.w);
    return acc;
  };
  auto point = [&](int j) __attribute__((always_inline)) { return in_lds ? sp[j] : gpts[j]; };
  auto rstart = [&](int i) __attribute__((always_inline)) { return in_lds ? sr[i] : grs[i]; };
  float h0d, h1d, h2d, h3d, r2;
  int h0i = 0, h1i = 1, h2i = 2, h3i = 3;
  h0d = sqrtf(fmaxf(dist2(point(0)), 0.f)); h1d = sqrtf(fmaxf(dist2(point(1)), 0.f));
  h2d = sqrtf(fmaxf(dist2(point(2)), 0.f)); h3d = sqrtf(fmaxf(dist2(point(3)), 0.f));
  if (!(h3d < h1d)) { const float td = h1d; const int ti = h1i; HE_MOVE(h1, h3); HE_SET(h3, td, ti); }     // __make_heap
  { const float xd = h0d; const int xi = h0i; HEAP4_REPLACE_ROOT(xd, xi); }
  r2 = h0d * h0d * 1.000001f + 1e-30f;
  for (int f = 0; f < T; ++f) {
    // frames / rows that cannot hold a point below this lane's CURRENT root (same bounds and slack as reach() in idw_knn_kernel)
    const float r2w = r2 + 2e-6f, dzf = fabsf(qz - gz[f]) - 1e-6f;
    const float dz2 = dzf > 0.f ? dzf * dzf : 0.f;
    int lo = 0, hi = 0;
    if (dz2 <= r2w) {
      const float ry = sqrtf(r2w - dz2);
      int kk = ry < 2.f ? (int)(ry * (float)max(H - 1, 1)) + 2 : H;
      if (kk > H) kk = H;
      lo = max(rstart(f * (H + 1) + max(0, y - kk)), 4);       // points 0..3 seeded the heap
      hi = rstart(f * (H + 1) + min(H, y + kk + 1));
    }
    // eight points per trip, their reads in flight together (one read per trip is one exposed LDS latency per point: 300 us)
    int j = lo;
    for (; j + 8 <= hi; j += 8) {
      float c[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) c[u] = dist2(point(j + u));
      const float m = fminf(fminf(fminf(c[0], c[1]), fminf(c[2], c[3])), fminf(fminf(c[4], c[5]), fminf(c[6], c[7])));
      if (m < r2) {
#pragma unroll
        for (int u = 0; u < 8; ++u) IDW_CONSIDER_HEAP(c[u], j + u);
      }
    }
    for (; j < hi; ++j) {
      const float c2 = dist2(point(j));
      IDW_CONSIDER_HEAP(c2, j);
    }
  }
  IDW_SORT_HEAP();
  const size_t qo = (size_t)b * Q + q;
  const float i0 = __fdiv_rn(1.f, h0d + tau), i1 = __fdiv_rn(1.f, h1d + tau);
  const float i2 = __fdiv_rn(1.f, h2d + tau), i3 = __fdiv_rn(1.f, h3d + tau);
  float w0 = i0 * i0, w1 = i1 * i1, w2 = i2 * i2, w3 = i3 * i3;
  const float ws = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(w0, w1), w2), w3), 1e-12f);
  w0 = __fdiv_rn(w0, ws); w1 = __fdiv_rn(w1, ws); w2 = __fdiv_rn(w2, ws); w3 = __fdiv_rn(w3, ws);
  const int32_t* pp = pt_pos + (size_t)b * Q;
  const float* vb = vals + (size_t)b * Q;
  const float v0 = vb[pp[h0i]], v1 = vb[pp[h1i]], v2 = vb[pp[h2i]], v3 = vb[pp[h3i]];
  out[qo] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(v0, w0), __fmul_rn(v1, w1)), __fmul_rn(v2, w2)), __fmul_rn(v3, w3));
  if (sel_idx) {
    *reinterpret_cast<int4*>(sel_idx + qo * 4) = make_int4(h0i, h1i, h2i, h3i);
    *reinterpret_cast<float4*>(sel_w + qo * 4) = make_float4(w0, w1, w2, w3);
  }
  }
}

// counts of undecided voxels per MODE-1 workgroup -> exclusive prefix sums, in place (one workgroup per sample)
__global__ __launch_bounds__(256) void idw_prefix_kernel(int32_t* amb, int Q, int nblk1) {
  __shared__ int wsum[4];
  int32_t* c = amb + (size_t)blockIdx.x * (Q + 1 + nblk1) + 1 + Q;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (nblk1 + 255) >> 8;
  int mine = 0;
  for (int i = 0; i < per; ++i) { const int g = tid * per + i; mine += g < nblk1 ? c[g] : 0; }
  int inc = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o, 64); if (lane >= o) inc += v; }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int run = inc - mine;
  for (int k = 0; k < wave; ++k) run += wsum[k];
  for (int i = 0; i < per; ++i) {
    const int g = tid * per + i;
    if (g < nblk1) { const int v = c[g]; c[g] = run; run += v; }
  }
}

// backward: d vals[pt_pos[j]] += w * dout[q]   (values enter the output linearly; weights are data).
// Every voxel adds into one of only N (gauge) addresses, so each block first accumulates in an LDS copy of the
// point list (N <= IDW_LDS_PTS) and then issues one global atomic per touched point.
constexpr int IDW_LDS_PTS = 8192;
__global__ __launch_bounds__(256) void idw_bwd_kernel(const float* __restrict__ dout, const int32_t* __restrict__ pt_pos,
                                                     const int32_t* __restrict__ pt_count, const int32_t* __restrict__ sel_idx,
                                                     const float* __restrict__ sel_w, float* dvals, int Q, int chunk) {
  __shared__ float accp[IDW_LDS_PTS];
  const int b = blockIdx.y;
  const int N = pt_count[b];
  if (N < 4) return;
  const bool use_lds = N <= IDW_LDS_PTS;
  if (use_lds) {
    for (int j = threadIdx.x; j < N; j += blockDim.x) accp[j] = 0.f;
    __syncthreads();
  }
  const int32_t* pp = pt_pos + (size_t)b * Q;
  float* dv = dvals + (size_t)b * Q;
  const int q0 = blockIdx.x * chunk, q1 = min(Q, q0 + chunk);
  for (int q = q0 + threadIdx.x; q < q1; q += blockDim.x) {
    const size_t i = (size_t)b * Q + q;
    const float g = dout[i];
    if (g == 0.f) continue;
    const int4 id = *reinterpret_cast<const int4*>(sel_idx + i * 4);
    const float4 w = *reinterpret_cast<const float4*>(sel_w + i * 4);
    if (use_lds) {
      atomicAdd(&accp[id.x], g * w.x); atomicAdd(&accp[id.y], g * w.y);
      atomicAdd(&accp[id.z], g * w.z); atomicAdd(&accp[id.w], g * w.w);
    } else {
      atomicAdd(dv + pp[id.x], g * w.x); atomicAdd(dv + pp[id.y], g * w.y);
      atomicAdd(dv + pp[id.z], g * w.z); atomicAdd(dv + pp[id.w], g * w.w);
    }
  }
  if (use_lds) {
    __syncthreads();
    for (int j = threadIdx.x; j < N; j += blockDim.x) {
      const float v = accp[j];
      if (v != 0.f) atomicAdd(dv + pp[j], v);
    }
  }
}

// ---- deterministic backward (a scratch registered with p2i_det_workspace): the same scatter in 64-bit FIXED POINT.  Integer addition
// is associative, so the order in which the atomics arrive no longer matters.  scale = 2^k with k chosen from max |dout| of the call
// (idw_absmax_kernel: an integer max over the float bit patterns, order-free too) so that the sum of all 4 Q contributions of a
// sample stays below 2^61: a contribution g * w (rounded to float as before) times 2^k is converted exactly unless it is smaller than
// 2^-41 of the largest -- more accurate than the float atomics it replaces.  ctl[0] = bits of max |dout|, ctl[1] = ticket of the
// last kernel.
__global__ __launch_bounds__(256) void idw_absmax_kernel(const float* __restrict__ dout, size_t n, unsigned* ctl) {
  __shared__ float red[16];
  float m = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(dout[i]));
  m = block_max(m, red);
  if (threadIdx.x == 0 && m > 0.f) atomicMax(ctl, __float_as_uint(m));
}
__device__ __forceinline__ int idw_fix_exp(unsigned maxbits, int Q) {
  if (maxbits == 0u) return 0;
  const int e = (int)((maxbits >> 23) & 0xffu) - 127;                // max |dout| < 2^(e + 1)
  int lq = 2;                                                        // ceil(log2(4 Q))
  while ((1ll << lq) < 4ll * Q) ++lq;
  int k = 61 - (e + 1) - lq;
  return k > 120 ? 120 : (k < -120 ? -120 : k);
}
__global__ __launch_bounds__(256) void idw_bwd_fix_kernel(const float* __restrict__ dout, const int32_t* __restrict__ pt_count,
                                                         const int32_t* __restrict__ sel_idx, const float* __restrict__ sel_w,
                                                         unsigned long long* acc64, const unsigned* __restrict__ ctl, int Q, int chunk) {
  extern __shared__ unsigned long long accl[];          // [IDW_LDS_PTS]
  const int b = blockIdx.y;
  const int N = pt_count[b];
  if (N < 4) return;
  const float S = ldexpf(1.f, idw_fix_exp(ctl[0], Q));
  const bool use_lds = N <= IDW_LDS_PTS;
  if (use_lds) {
    for (int j = threadIdx.x; j < N; j += blockDim.x) accl[j] = 0ull;
    __syncthreads();
  }
  unsigned long long* ga = acc64 + (size_t)b * Q;
  const int q0 = blockIdx.x * chunk, q1 = min(Q, q0 + chunk);
  for (int q = q0 + threadIdx.x; q < q1; q += blockDim.x) {
    const size_t i = (size_t)b * Q + q;
    const float g = dout[i];
    if (g == 0.f) continue;
    const int4 id = *reinterpret_cast<const int4*>(sel_idx + i * 4);
    const float4 w = *reinterpret_cast<const float4*>(sel_w + i * 4);
    const unsigned long long c0 = (unsigned long long)__float2ll_rn(g * w.x * S), c1 = (unsigned long long)__float2ll_rn(g * w.y * S);
    const unsigned long long c2 = (unsigned long long)__float2ll_rn(g * w.z * S), c3 = (unsigned long long)__float2ll_rn(g * w.w * S);
    if (use_lds) {
      atomicAdd(&accl[id.x], c0); atomicAdd(&accl[id.y], c1); atomicAdd(&accl[id.z], c2); atomicAdd(&accl[id.w], c3);
    } else {
      atomicAdd(ga + id.x, c0); atomicAdd(ga + id.y, c1); atomicAdd(ga + id.z, c2); atomicAdd(ga + id.w, c3);
    }
  }
  if (use_lds) {
    __syncthreads();
    for (int j = threadIdx.x; j < N; j += blockDim.x) {
      const unsigned long long v = accl[j];
      if (v != 0ull) atomicAdd(ga + j, v);
    }
  }
}
// d vals[pt_pos[j]] = fixed-point sum / scale
__global__ __launch_bounds__(256) void idw_bwd_fix_finish_kernel(const unsigned long long* __restrict__ acc64, const int32_t* __restrict__ pt_pos,
                                                                const int32_t* __restrict__ pt_count, float* dvals, unsigned* ctl, int Q) {
  const int b = blockIdx.y;
  const int N = pt_count[b];
  const int k = idw_fix_exp(ctl[0], Q);
  if (N >= 4) {
    const int32_t* pp = pt_pos + (size_t)b * Q;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < N; j += gridDim.x * blockDim.x) {
      const long long v = (long long)acc64[(size_t)b * Q + j];
      dvals[(size_t)b * Q + pp[j]] = (float)ldexp((double)v, -k);
    }
  }
  (void)ctl;
}

}  // namespace p2i
using namespace p2i;

static int idw_fwd_impl(const float* vals_src, const float* mask, const float* grid_x, const float* grid_y,
                        const float* grid_z, float* out, int32_t* pt_pos, int32_t* pt_count, int32_t* frame_count,
                        int32_t* row_start, float* pt_xyzn, int32_t* sel_idx, float* sel_w, int32_t* amb, int B, int T, int H, int W, float tau,
                        void* stream) {
  P2I_REQUIRE(vals_src && mask && grid_x && grid_y && grid_z && out && pt_pos && pt_count && frame_count && row_start && pt_xyzn,
              "null pointer");
  P2I_REQUIRE((sel_idx == nullptr) == (sel_w == nullptr), "sel_idx and sel_w go together");
  P2I_REQUIRE((long long)B * T * H * W < (1ll << 29), "IDW problem too large");
  hipStream_t s = (hipStream_t)stream;
  const int Q = T * H * W;
  P2I_LAUNCH(idw_count_kernel, dim3(B * T), dim3(IDW_CT), 0, s, mask, frame_count, H * W);
  P2I_LAUNCH(idw_compact_kernel, dim3(B * T), dim3(IDW_CT), 0, s, mask, frame_count, grid_x, grid_y, grid_z, pt_pos, pt_count,
                     row_start, pt_xyzn, amb, T, H, W);
  const dim3 grid(ceil_div(Q, 256), B);
  const int nblk1 = (int)grid.x;
  const float4* pts = reinterpret_cast<const float4*>(pt_xyzn);
  // P2I_IDW_REPLAY_ALL=1: MODE 2 replays the reference's scan for every listed voxel (round 3's pass) instead of deciding from the
  // points within the 4th distance first (read per call: A/B runs, tests)
  const char* rae = getenv("P2I_IDW_REPLAY_ALL");
  const int replay_all = (rae && atoi(rae) != 0) ? 1 : 0;
  if (amb && nblk1 <= IDW_MAX_BLK) {
    P2I_LAUNCH((idw_knn_kernel<1, 256>), grid, dim3(256), 0, s, vals_src, grid_x, grid_y, grid_z, pt_pos, pt_count, row_start, pts, out,
                       sel_idx, sel_w, amb, nblk1, T, H, W, tau, 0);
    P2I_LAUNCH(idw_prefix_kernel, dim3(B), dim3(256), 0, s, amb, Q, nblk1);
    // (the number of undecided voxels is known on the device only: a full grid whose surplus workgroups leave at once)
    // 256-thread replay workgroups: 288 us for 8 x 21 k voxels (79 gauges, B = 8); one wave per workgroup (4 x as many windows to
    // load, nothing to overlap them with): 656 us.  Either way the pass is a latency-bound chain per wave (sqrt + heap moves per
    // point some lane takes: ~500 cycles), with 2-3 waves per SIMD in all -- not an instruction-issue limit.
    P2I_LAUNCH((idw_knn_kernel<2, 256>), grid, dim3(256), 0, s, vals_src, grid_x, grid_y, grid_z, pt_pos, pt_count,
                       row_start, pts, out, sel_idx, sel_w, amb, nblk1, T, H, W, tau, replay_all);
    if (!replay_all) {   // what the fixed bound could not decide (~5 % of the listed voxels): the exact replay, a lane per voxel
      const char* we = getenv("P2I_IDW_LANE_REPLAY");               // 0: the cooperative replay over the second list (A/B; read per call)
      if (we && atoi(we) == 0)
        P2I_LAUNCH((idw_knn_kernel<2, 256>), grid, dim3(256), 0, s, vals_src, grid_x, grid_y, grid_z, pt_pos, pt_count,
                           row_start, pts, out, sel_idx, sel_w, amb, nblk1, T, H, W, tau, 3);
      else {
        // (the number of voxels left is known on the device only: a full grid whose surplus one-wave workgroups leave at once)
        const size_t lds = sizeof(float4) * IDW_RW_PCAP + sizeof(int) * (size_t)T * (H + 1);
        const int use_lds = lds <= 64 * 1024 ? 1 : 0;
        P2I_LAUNCH(idw_replay_lane_kernel, dim3(ceil_div(Q, 64 * 8), B), dim3(64), use_lds ? lds : 0, s, vals_src, grid_x, grid_y, grid_z,
                           pt_pos, pt_count, row_start, pts, out, sel_idx, sel_w, amb, nblk1, T, H, W, tau, use_lds);
      }
    }
  } else {
    P2I_LAUNCH((idw_knn_kernel<0, 256>), grid, dim3(256), 0, s, vals_src, grid_x, grid_y, grid_z, pt_pos, pt_count, row_start, pts, out,
                       sel_idx, sel_w, (int32_t*)nullptr, nblk1, T, H, W, tau, 0);
  }
  return launch_status();
}

extern "C" int p2i_idw_fwd(const float* vals_src, const float* mask, const float* grid_x, const float* grid_y,
                           const float* grid_z, float* out, int32_t* pt_pos, int32_t* pt_count, int32_t* frame_count,
                           int32_t* row_start, float* pt_xyzn, int32_t* sel_idx, float* sel_w, int B, int T, int H, int W, float tau, void* stream) {
  return idw_fwd_impl(vals_src, mask, grid_x, grid_y, grid_z, out, pt_pos, pt_count, frame_count, row_start, pt_xyzn, sel_idx, sel_w, nullptr,
                      B, T, H, W, tau, stream);
}

extern "C" int p2i_idw_fwd_ws(const float* vals_src, const float* mask, const float* grid_x, const float* grid_y,
                              const float* grid_z, float* out, int32_t* pt_pos, int32_t* pt_count, int32_t* frame_count,
                              int32_t* row_start, float* pt_xyzn, int32_t* sel_idx, float* sel_w, int32_t* amb, int B, int T, int H, int W,
                              float tau, void* stream) {
  P2I_REQUIRE(amb, "null workspace");
  return idw_fwd_impl(vals_src, mask, grid_x, grid_y, grid_z, out, pt_pos, pt_count, frame_count, row_start, pt_xyzn, sel_idx, sel_w, amb,
                      B, T, H, W, tau, stream);
}

extern "C" int p2i_idw_bwd(const float* dout, const int32_t* pt_pos, const int32_t* pt_count, const int32_t* sel_idx,
                           const float* sel_w, float* dvals_src, int B, int T, int H, int W, void* stream) {
  P2I_REQUIRE(dout && pt_pos && pt_count && sel_idx && sel_w && dvals_src, "null pointer");
  const int Q = T * H * W;
  const size_t total = (size_t)B * Q;
  (void)p2i::memset_async(dvals_src, 0, sizeof(float) * total, (hipStream_t)stream);
  const int chunk = 4096;
  hipStream_t s = (hipStream_t)stream;
  // deterministic mode: 8 bytes of fixed-point accumulator per voxel slot (a sample has at most Q points) from the registered scratch
  const DetWs ws = det_take(2 * total + 64, 0);
  if (ws.part != nullptr && (reinterpret_cast<uintptr_t>(ws.part) & 7) == 0) {
    unsigned long long* acc64 = reinterpret_cast<unsigned long long*>(ws.part);
    unsigned* ctl = reinterpret_cast<unsigned*>(ws.part + 2 * total);            // the max |dout| bits, behind the accumulators
    (void)p2i::memset_async(acc64, 0, sizeof(unsigned long long) * total + 64, s);
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void*)idw_bwd_fix_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
      attr_set = true;
    }
    P2I_LAUNCH(idw_absmax_kernel, dim3(256), dim3(256), 0, s, dout, total, ctl);
    P2I_LAUNCH(idw_bwd_fix_kernel, dim3(ceil_div(Q, chunk), B), dim3(256), sizeof(unsigned long long) * IDW_LDS_PTS, s, dout, pt_count, sel_idx,
               sel_w, acc64, ctl, Q, chunk);
    P2I_LAUNCH(idw_bwd_fix_finish_kernel, dim3(16, B), dim3(256), 0, s, acc64, pt_pos, pt_count, dvals_src, ctl, Q);
    return launch_status();
  }
  P2I_LAUNCH(idw_bwd_kernel, dim3(ceil_div(Q, chunk), B), dim3(256), 0, s, dout, pt_pos, pt_count, sel_idx, sel_w, dvals_src, Q, chunk);
  return launch_status();
}
