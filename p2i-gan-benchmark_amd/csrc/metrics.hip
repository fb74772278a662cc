// Rainfall evaluation metrics on the device (reference: p2igan_bench/metrics/metric.py).
//   p2i_metrics_pointwise : one pass over (pred, target): |d| and d^2 sums (RegressionMetrics.update, :42-52) and the
//                           2x2 contingency counts per threshold (CategoricalMetrics.update, :92-111), after the
//                           rain-rate transform 10^(x/16) * 0.036 (:16-20)
//   p2i_metrics_fss       : FractionalSkillScoreMetric.update (:152-170): for every threshold and scale the sums of
//                           (fp-ft)^2 and fp^2+ft^2 over the avg_pool2d(kernel=s, stride=1, padding=s//2,
//                           count_include_pad) fraction fields -- output extent H+1 for even s, as in the reference
// HBM-bound: pred/target are read once per kernel; the FSS pass first packs the 2*nt threshold bits of a pixel into a
// byte plane (1/8 of the fp32 bytes) and box-sums that plane out of L2.
#include "common.h"

namespace p2i {

constexpr int MAX_THR = 4, MAX_SCALE = 4;
struct MetricArgs {
  float thr[MAX_THR];
  int scale[MAX_SCALE];
  int nt, ns, apply_transform;
};

__device__ __forceinline__ float rain_rate(float x, int apply) { return apply ? powf(10.0f, x * 0.0625f) * 0.036f : x; }

__global__ __launch_bounds__(256) void metrics_pointwise_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                                long long n, MetricArgs a, float* __restrict__ sums,
                                                                unsigned long long* __restrict__ counts, uint8_t* __restrict__ bits, DetWs ws) {
  __shared__ float red[16];
  float s_abs = 0.f, s_sq = 0.f;
  unsigned c[MAX_THR][4];
#pragma unroll
  for (int t = 0; t < MAX_THR; ++t) c[t][0] = c[t][1] = c[t][2] = c[t][3] = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float p0 = pred[i], t0 = target[i];
    const float d = rain_rate(p0, a.apply_transform) - rain_rate(t0, a.apply_transform);   // RegressionMetrics honours apply_transform
    s_abs += fabsf(d);
    s_sq += d * d;
    const float p = rain_rate(p0, 1), t = rain_rate(t0, 1);                                 // categorical / FSS always transform
    unsigned b = 0;
#pragma unroll
    for (int k = 0; k < MAX_THR; ++k)
      if (k < a.nt) {
        const bool pp = p >= a.thr[k], tt = t >= a.thr[k];
        c[k][0] += (pp && tt);      // hits
        c[k][1] += (!pp && tt);     // misses
        c[k][2] += (pp && !tt);     // false alarms
        c[k][3] += (!pp && !tt);    // correct negatives
        b |= (pp ? 1u : 0u) << k;
        b |= (tt ? 1u : 0u) << (4 + k);
      }
    if (bits) bits[i] = (uint8_t)b;
  }
  s_abs = block_sum(s_abs, red);
  s_sq = block_sum(s_sq, red);
  if (threadIdx.x == 0) {
    if (ws.part) { ws.part[2 * blockIdx.x] = s_abs; ws.part[2 * blockIdx.x + 1] = s_sq; }     // added in workgroup order by det_reduce
    else { atomicAdd(&sums[0], s_abs); atomicAdd(&sums[1], s_sq); }
  }
#pragma unroll
  for (int k = 0; k < MAX_THR; ++k)
    if (k < a.nt) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned v = c[k][j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&counts[k * 4 + j], (unsigned long long)v);
      }
    }
}

// one thread per output position (i, j) of the (H+1) x (W+1) frame; scales whose output is H x W mask the last row/col
__global__ __launch_bounds__(256) void metrics_fss_kernel(const uint8_t* __restrict__ bits, int N, int H, int W, MetricArgs a,
                                                          float* __restrict__ num, float* __restrict__ den, DetWs ws) {
  __shared__ float red[16];
  const int Ho = H + 1, Wo = W + 1;
  const long long total = (long long)N * Ho * Wo;
  float ln[MAX_THR][MAX_SCALE], ld[MAX_THR][MAX_SCALE];
#pragma unroll
  for (int t = 0; t < MAX_THR; ++t)
#pragma unroll
    for (int s = 0; s < MAX_SCALE; ++s) ln[t][s] = ld[t][s] = 0.f;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int j = (int)(idx % Wo);
    const int i = (int)((idx / Wo) % Ho);
    const int n = (int)(idx / ((long long)Wo * Ho));
    const uint8_t* img = bits + (long long)n * H * W;
#pragma unroll
    for (int si = 0; si < MAX_SCALE; ++si)
      if (si < a.ns) {
        const int s = a.scale[si], pad = s / 2;
        const int ho = H + 2 * pad - s + 1, wo = W + 2 * pad - s + 1;
        if (i >= ho || j >= wo) continue;
        unsigned cp[MAX_THR] = {0, 0, 0, 0}, ct[MAX_THR] = {0, 0, 0, 0};
        for (int y = i - pad; y < i - pad + s; ++y) {
          if ((unsigned)y >= (unsigned)H) continue;
          for (int x = j - pad; x < j - pad + s; ++x) {
            if ((unsigned)x >= (unsigned)W) continue;
            const unsigned b = img[y * W + x];
#pragma unroll
            for (int k = 0; k < MAX_THR; ++k) { cp[k] += (b >> k) & 1u; ct[k] += (b >> (4 + k)) & 1u; }
          }
        }
        const float inv = 1.0f / (float)(s * s);                // count_include_pad=True: always s*s
#pragma unroll
        for (int k = 0; k < MAX_THR; ++k)
          if (k < a.nt) {
            const float fp = (float)cp[k] * inv, ft = (float)ct[k] * inv;
            ln[k][si] += (fp - ft) * (fp - ft);
            ld[k][si] += fp * fp + ft * ft;
          }
      }
  }
#pragma unroll
  for (int k = 0; k < MAX_THR; ++k)
#pragma unroll
    for (int si = 0; si < MAX_SCALE; ++si)
      if (k < a.nt && si < a.ns) {
        const float vn = block_sum(ln[k][si], red), vd = block_sum(ld[k][si], red);
        if (threadIdx.x == 0) {
          if (ws.part) {       // [workgroups][2][nt * ns]: added in workgroup order by det_reduce
            ws.part[((size_t)blockIdx.x * 2 + 0) * (a.nt * a.ns) + k * a.ns + si] = vn;
            ws.part[((size_t)blockIdx.x * 2 + 1) * (a.nt * a.ns) + k * a.ns + si] = vd;
          } else {
            atomicAdd(&num[k * a.ns + si], vn);
            atomicAdd(&den[k * a.ns + si], vd);
          }
        }
      }
}

static int fill_args(MetricArgs& a, const float* thr, int nt, const int* scales, int ns, int apply_transform) {
  P2I_REQUIRE(nt >= 0 && nt <= MAX_THR && ns >= 0 && ns <= MAX_SCALE, "at most %d thresholds and %d scales", MAX_THR, MAX_SCALE);
  a.nt = nt; a.ns = ns; a.apply_transform = apply_transform;
  for (int i = 0; i < MAX_THR; ++i) a.thr[i] = i < nt ? thr[i] : 0.f;
  for (int i = 0; i < MAX_SCALE; ++i) {
    a.scale[i] = i < ns ? scales[i] : 1;
    P2I_REQUIRE(a.scale[i] >= 1 && a.scale[i] <= 64, "bad FSS scale");
  }
  return P2I_OK;
}

}  // namespace p2i

using namespace p2i;

extern "C" int p2i_metrics_pointwise(const float* pred, const float* target, int64_t n, const float* thresholds_host, int nt,
                                     int apply_transform, float* sums2, unsigned long long* counts, uint8_t* bits, void* stream) {
  P2I_REQUIRE(pred && target && sums2 && (counts || nt == 0) && n > 0, "bad arguments");
  MetricArgs a;
  if (int e = fill_args(a, thresholds_host, nt, nullptr, 0, apply_transform)) return e;
  const long long blocks = (n + 256 * 8 - 1) / (256 * 8);
  const int nb = (int)(blocks > 4096 ? 4096 : blocks);
  const DetWs ws = det_take((size_t)2 * nb, 0);
  P2I_LAUNCH(metrics_pointwise_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, pred, target, (long long)n, a, sums2, counts, bits, ws);
  if (ws.part) return det_reduce(ws.part, 2, nb, 1, 2, DetSegs{{sums2, nullptr, nullptr, nullptr}, {2, 0, 0, 0}}, (hipStream_t)stream);
  return launch_status();
}

extern "C" int p2i_metrics_fss(const uint8_t* bits, int N, int H, int W, int nt, const int* scales_host, int ns, float* num, float* den,
                               void* stream) {
  P2I_REQUIRE(bits && num && den && N > 0 && H > 0 && W > 0, "bad arguments");
  MetricArgs a;
  float dummy[MAX_THR] = {0, 0, 0, 0};
  if (int e = fill_args(a, dummy, nt, scales_host, ns, 1)) return e;
  const long long total = (long long)N * (H + 1) * (W + 1);
  const long long blocks = (total + 255) / 256;
  const int nb = (int)(blocks > 8192 ? 8192 : blocks);
  const int nv = nt * ns;
  const DetWs ws = nv > 0 ? det_take((size_t)nb * 2 * nv, 0) : DetWs{nullptr, nullptr};
  P2I_LAUNCH(metrics_fss_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, bits, N, H, W, a, num, den, ws);
  if (ws.part) return det_reduce(ws.part, 2 * nv, nb, 1, 2 * nv, DetSegs{{num, den, nullptr, nullptr}, {nv, nv, 0, 0}}, (hipStream_t)stream);
  return launch_status();
}
