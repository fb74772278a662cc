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

// ---- compaction of mask > 0 in (t, y, x) order
__global__ void idw_count_kernel(const float* __restrict__ mask, int32_t* frame_count, int HW) {
  __shared__ float red[16];
  const float* m = mask + (size_t)blockIdx.x * HW;
  float c = 0.f;
  for (int i = threadIdx.x; i < HW; i += blockDim.x) c += (m[i] > 0.f) ? 1.f : 0.f;
  c = block_sum(c, red);
  if (threadIdx.x == 0) frame_count[blockIdx.x] = (int)c;
}
__global__ __launch_bounds__(256) void idw_compact_kernel(const float* __restrict__ mask, const int32_t* __restrict__ frame_count,
                                                         const float* __restrict__ gx, const float* __restrict__ gy,
                                                         const float* __restrict__ gz, int32_t* pt_pos, int32_t* pt_count,
                                                         int32_t* row_start, float* pt_xyzn, int T, int H, int W) {
  __shared__ int wsum[4];
  __shared__ int s_base;
  const int bt = blockIdx.x, b = bt / T, t = bt % T, HW = H * W, Q = T * HW;
  if (threadIdx.x == 0) {
    int off = 0;
    for (int k = 0; k < t; ++k) off += frame_count[b * T + k];
    s_base = off;
    if (t == T - 1) pt_count[b] = off + frame_count[bt];
  }
  __syncthreads();
  int base = s_base;
  const float* m = mask + (size_t)bt * HW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float Wd = (float)max(W - 1, 1), Hd = (float)max(H - 1, 1), Td = (float)max(T - 1, 1);
  for (int i0 = 0; i0 < HW; i0 += 256) {
    const int i = i0 + threadIdx.x;
    const bool on = i < HW && m[i] > 0.f;
    const unsigned long long bal = __ballot(on);
    const int rank = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int woff = 0, tot = 0;
    for (int k = 0; k < 4; ++k) { if (k < wave) woff += wsum[k]; tot += wsum[k]; }
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

// ---- 4-NN + IDW.  thread = one query voxel; points streamed through scalar loads (uniform index)
__global__ __launch_bounds__(256) void idw_knn_kernel(const float* __restrict__ vals, const float* __restrict__ gx,
                                                     const float* __restrict__ gy, const float* __restrict__ gz,
                                                     const int32_t* __restrict__ pt_pos, const int32_t* __restrict__ pt_count,
                                                     const int32_t* __restrict__ row_start, const float4* __restrict__ pt_xyzn, float* out, int32_t* sel_idx,
                                                     float* sel_w, int T, int H, int W, float tau) {
  const int b = blockIdx.y, HW = H * W, Q = T * HW;
  const int q_raw = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = q_raw < Q;                       // inactive lanes shadow the last voxel: wave reductions below
  const int q = active ? q_raw : Q - 1;                // need every lane to hold defined values
  const int N = pt_count[b];
  const size_t qo = (size_t)b * Q + q;
  if (N < 4) {       // N == 0: zeros (layer.py:330-332); 0 < N < 4: reference raises in topk
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

  auto dist2 = [&](const float4 p) {
    float acc = __fmul_rn(a0, p.x);
    acc = __fmaf_rn(a1, p.y, acc);
    acc = __fmaf_rn(a2, p.z, acc);
    acc = __fadd_rn(acc, n1);       // fma(n1, 1, acc)
    acc = __fadd_rn(acc, p.w);      // fma(1, |p|^2, acc)
    return acc;
  };
  float h0d, h1d, h2d, h3d;
  int h0i = 0, h1i = 1, h2i = 2, h3i = 3;
  float r2;                         // fast-reject bound on d^2 (see below)
  {
    h0d = sqrtf(fmaxf(dist2(pts[0]), 0.f)); h1d = sqrtf(fmaxf(dist2(pts[1]), 0.f));
    h2d = sqrtf(fmaxf(dist2(pts[2]), 0.f)); h3d = sqrtf(fmaxf(dist2(pts[3]), 0.f));
    // __make_heap: parent = 1 (swap with child 3 unless child < parent), then parent = 0
    if (!(h3d < h1d)) { const float td = h1d; const int ti = h1i; HE_MOVE(h1, h3); HE_SET(h3, td, ti); }
    const float xd = h0d; const int xi = h0i;
    HEAP4_REPLACE_ROOT(xd, xi);
    r2 = h0d * h0d * 1.000001f + 1e-30f;
  }
  // fast reject: r2 = fl(fl(r*r)*(1+2^-20)) > r^2 exactly, so c2 >= r2 implies sqrt_rn(c2) >= r (no insert,
  // as std::partial_sort's strict comparison demands); below r2 the exact d-space test decides.
  //
  // Exact pruning.  Points are scanned in index order (frame-major, then row-major) exactly like the reference,
  // but whole frames / row ranges that provably cannot beat the CURRENT root are skipped: a skipped point has
  // |dz| or |dy| (hence its computed distance, up to the 2e-6 slack that covers the fp32 cancellation of the
  // |a|^2+|b|^2-2ab chain) above the root at its turn, so std::partial_sort would not have inserted it either
  // and the heap evolves identically.  Bounds are wave-uniform (max root / row span over the lanes).
  const int* rs = row_start + (size_t)b * T * (H + 1);
  const float inv_h = 1.f / (float)max(H - 1, 1);
  for (int f = 0; f < T; ++f) {
    const int fs = rs[f * (H + 1)], fe = rs[f * (H + 1) + H];
    if (fe <= 4 || fs == fe) continue;                 // points 0..3 seeded the heap
    const float r2w = wave_max(r2) + 2e-6f;
    const float dzf = fabsf(qz - gz[f]) - 1e-6f;       // gz[f] == f/(T-1) as the points carry it (to 1 ulp)
    const float dz2 = dzf > 0.f ? dzf * dzf : 0.f;
    if (dz2 > r2w) continue;                           // the whole frame is farther than every lane's root
    const float ry = sqrtf(r2w - dz2);
    int kk = (int)(ry * (float)max(H - 1, 1)) + 2;     // rows that can hold a point within ry (+ slack)
    if (kk > H) kk = H;
    const int ymin = -(int)wave_max((float)(-y)), ymax = (int)wave_max((float)y);
    const int ya = max(0, ymin - kk), yb = min(H, ymax + kk + 1);
    (void)inv_h;
    int lo = rs[f * (H + 1) + ya];
    const int hi = rs[f * (H + 1) + yb];
    if (lo < 4) lo = 4;
    for (int j = lo; j < hi; ++j) {
      const float c2 = dist2(pts[j]);
      if (c2 < r2) {
        const float dc = sqrtf(fmaxf(c2, 0.f));
        if (dc < h0d) {
          HEAP4_REPLACE_ROOT(dc, j);
          r2 = h0d * h0d * 1.000001f + 1e-30f;
        }
      }
    }
  }
  // __sort_heap -> ascending h0..h3
  {  // len 4 -> 3
    const float xd = h3d; const int xi = h3i; HE_MOVE(h3, h0);
    if (h2d < h1d) { HE_MOVE(h0, h1); if (h0d < xd) { HE_MOVE(h1, h0); HE_SET(h0, xd, xi); } else HE_SET(h1, xd, xi); }
    else { HE_MOVE(h0, h2); if (h0d < xd) { HE_MOVE(h2, h0); HE_SET(h0, xd, xi); } else HE_SET(h2, xd, xi); }
  }
  {  // len 3 -> 2
    const float xd = h2d; const int xi = h2i; HE_MOVE(h2, h0); HE_MOVE(h0, h1);
    if (h0d < xd) { HE_MOVE(h1, h0); HE_SET(h0, xd, xi); } else HE_SET(h1, xd, xi);
  }
  {  // len 2 -> 1
    const float xd = h1d; const int xi = h1i; HE_MOVE(h1, h0); HE_SET(h0, xd, xi);
  }
  // weights (layer.py:283-290): inv = 1/(d+tau); w = inv*inv; w /= (sum + 1e-12); out = sum(v*w)
  const float i0 = __fdiv_rn(1.f, h0d + tau), i1 = __fdiv_rn(1.f, h1d + tau);
  const float i2 = __fdiv_rn(1.f, h2d + tau), i3 = __fdiv_rn(1.f, h3d + tau);
  float w0 = i0 * i0, w1 = i1 * i1, w2 = i2 * i2, w3 = i3 * i3;
  const float ws = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(w0, w1), w2), w3), 1e-12f);
  w0 = __fdiv_rn(w0, ws); w1 = __fdiv_rn(w1, ws); w2 = __fdiv_rn(w2, ws); w3 = __fdiv_rn(w3, ws);
  const int32_t* pp = pt_pos + (size_t)b * Q;
  const float* vb = vals + (size_t)b * Q;
  if (!active) return;
  const float v0 = vb[pp[h0i]], v1 = vb[pp[h1i]], v2 = vb[pp[h2i]], v3 = vb[pp[h3i]];
  out[qo] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(v0, w0), __fmul_rn(v1, w1)), __fmul_rn(v2, w2)), __fmul_rn(v3, w3));
  if (sel_idx) {
    *reinterpret_cast<int4*>(sel_idx + qo * 4) = make_int4(h0i, h1i, h2i, h3i);      // POINT indices (pt_pos maps them to voxels)
    *reinterpret_cast<float4*>(sel_w + qo * 4) = make_float4(w0, w1, w2, w3);
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

}  // namespace p2i
using namespace p2i;

extern "C" int p2i_idw_fwd(const float* vals_src, const float* mask, const float* grid_x, const float* grid_y,
                           const float* grid_z, float* out, int32_t* pt_pos, int32_t* pt_count, int32_t* frame_count,
                           int32_t* row_start, float* pt_xyzn, int32_t* sel_idx, float* sel_w, int B, int T, int H, int W, float tau, void* stream) {
  P2I_REQUIRE(vals_src && mask && grid_x && grid_y && grid_z && out && pt_pos && pt_count && frame_count && row_start && pt_xyzn,
              "null pointer");
  P2I_REQUIRE((sel_idx == nullptr) == (sel_w == nullptr), "sel_idx and sel_w go together");
  P2I_REQUIRE((long long)B * T * H * W < (1ll << 29), "IDW problem too large");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(idw_count_kernel, dim3(B * T), dim3(256), 0, s, mask, frame_count, H * W);
  hipLaunchKernelGGL(idw_compact_kernel, dim3(B * T), dim3(256), 0, s, mask, frame_count, grid_x, grid_y, grid_z, pt_pos, pt_count,
                     row_start, pt_xyzn, T, H, W);
  hipLaunchKernelGGL(idw_knn_kernel, dim3(ceil_div(T * H * W, 256), B), dim3(256), 0, s, vals_src, grid_x, grid_y, grid_z, pt_pos,
                     pt_count, row_start, reinterpret_cast<const float4*>(pt_xyzn), out, sel_idx, sel_w, T, H, W, tau);
  return launch_status();
}

extern "C" int p2i_idw_bwd(const float* dout, const int32_t* pt_pos, const int32_t* pt_count, const int32_t* sel_idx,
                           const float* sel_w, float* dvals_src, int B, int T, int H, int W, void* stream) {
  P2I_REQUIRE(dout && pt_pos && pt_count && sel_idx && sel_w && dvals_src, "null pointer");
  const int Q = T * H * W;
  const size_t total = (size_t)B * Q;
  (void)hipMemsetAsync(dvals_src, 0, sizeof(float) * total, (hipStream_t)stream);
  const int chunk = 4096;
  hipLaunchKernelGGL(idw_bwd_kernel, dim3(ceil_div(Q, chunk), B), dim3(256), 0, (hipStream_t)stream, dout, pt_pos, pt_count, sel_idx,
                     sel_w, dvals_src, Q, chunk);
  return launch_status();
}
