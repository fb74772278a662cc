// Loss + gradient fusions: ReconstructionLoss (losses.py:38-48: weighted L1 :56-65, temporal
// forward difference :83-85, softmax(./0.1) :68-73, KL batchmean :76-80) and the hinge / lsgan
// adversarial terms (losses.py:210-226, train.py:266-283,301-308).
#include "common.h"

namespace p2i {

// one block per (b, t') row of the temporal difference; HW elements per row
__global__ __launch_bounds__(1024) void kl_row_kernel(const float* __restrict__ pred, const float* __restrict__ tgt, float* G,
                                                      float* rowkl, int T, int HW, float gscale) {
  __shared__ float red[16];
  const int row = blockIdx.x, b = row / (T - 1), tp = row % (T - 1);
  const float* p0 = pred + ((size_t)b * T + tp) * HW;
  const float* p1 = p0 + HW;
  const float* q0 = tgt + ((size_t)b * T + tp) * HW;
  const float* q1 = q0 + HW;
  const float temp = 0.1f;
  float mp = -INFINITY, mq = -INFINITY;
  for (int i = threadIdx.x; i < HW; i += blockDim.x) {
    mp = fmaxf(mp, (p1[i] - p0[i]) / temp);
    mq = fmaxf(mq, (q1[i] - q0[i]) / temp);
  }
  mp = block_max(mp, red);
  mq = block_max(mq, red);
  float sp = 0.f, sq = 0.f;
  for (int i = threadIdx.x; i < HW; i += blockDim.x) {
    sp += expf((p1[i] - p0[i]) / temp - mp);
    sq += expf((q1[i] - q0[i]) / temp - mq);
  }
  sp = block_sum(sp, red);
  sq = block_sum(sq, red);
  float kl = 0.f;
  float* g = G + (size_t)row * HW;
  for (int i = threadIdx.x; i < HW; i += blockDim.x) {
    const float ph = expf((p1[i] - p0[i]) / temp - mp) / sp;     // softmax(pred diff)
    const float qq = expf((q1[i] - q0[i]) / temp - mq) / sq;     // softmax(true diff)
    if (qq > 0.f) kl += qq * (logf(qq) - logf(ph));              // F.kl_div(log p, q): q*(log q - log p)
    g[i] = (ph - qq) * gscale;                                   // d/d(pred diff) of k1 * reg
  }
  kl = block_sum(kl, red);
  if (threadIdx.x == 0) rowkl[row] = kl;
}

// weighted-L1 term + assembly of dpred; block partial sums of w*|d| in partial[blockIdx]
__global__ __launch_bounds__(256) void l1_grad_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                      const float* __restrict__ G, float* dpred, float* partial, int T, int HW,
                                                      size_t total, float inv_total) {
  __shared__ float red[16];
  const float a = 0.50f, bb = 5.14f, c = 0.12f, xmax = 0.70f;
  const float wmax = a * expf(bb * xmax) + c;
  float acc = 0.f;
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int i = idx % HW;
    const size_t bt = idx / HW;
    const int t = bt % T;
    const size_t b = bt / T;
    const float y = tgt[idx], p = pred[idx];
    const float w = (y > xmax) ? wmax : (a * expf(bb * y) + c);
    const float d = p - y;
    acc += w * fabsf(d);
    float g = (d > 0.f ? w : (d < 0.f ? -w : 0.f)) * inv_total;
    if (G) {
      const size_t r = b * (T - 1) + t;
      if (t >= 1) g += G[(r - 1) * HW + i];
      if (t <= T - 2) g -= G[r * HW + i];
    }
    dpred[idx] = g;
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

__global__ void recloss_final_kernel(const float* __restrict__ partial, int np, const float* __restrict__ rowkl, int nr, float inv_total,
                                     float inv_b, float k1, float* out3) {
  __shared__ float red[16];
  float a = 0.f, r = 0.f;
  for (int i = threadIdx.x; i < np; i += blockDim.x) a += partial[i];
  for (int i = threadIdx.x; i < nr; i += blockDim.x) r += rowkl[i];
  a = block_sum(a, red);
  r = block_sum(r, red);
  if (threadIdx.x == 0) {
    const float pool = a * inv_total, reg = r * inv_b;
    out3[0] = pool; out3[1] = reg; out3[2] = pool + k1 * reg;
  }
}

// nn.BCELoss term of the reference's 'nsgan' (losses.py:201-202: BCELoss on the RAW logits): -(y log x + (1-y) log(1-x)) with
// torch's clamp of the logs at -100; x outside [0,1] is an error in torch (flagged, the loss comes back NaN and the wrapper raises)
__device__ __forceinline__ float bce_term(float x, float y, bool& bad) {
  if (!(x >= 0.f && x <= 1.f)) { bad = true; return 0.f; }
  return -(y * fmaxf(logf(x), -100.f) + (1.f - y) * fmaxf(logf(1.f - x), -100.f));
}
__device__ __forceinline__ float bce_grad(float x, float y) {     // d/dx with the same clamps (grad 0 where the log is clamped)
  const float gl = (logf(x) > -100.f) ? 1.f / x : 0.f, g1 = (logf(1.f - x) > -100.f) ? 1.f / (1.f - x) : 0.f;
  return -(y * gl - (1.f - y) * g1);
}
// adversarial losses, single block.  loss_type 0 hinge, 1 lsgan, 2 nsgan.
__global__ __launch_bounds__(1024) void gan_loss_kernel(const float* __restrict__ la, const float* __restrict__ lb, int n, int loss_type,
                                                        int mode, float weight, float real_label, float fake_label, float* loss,
                                                        float* da, float* db) {
  __shared__ float red[16];
  const float inv_n = 1.f / (float)n;
  float acc = 0.f;
  bool bad = false;                                     // nsgan: an input outside [0, 1] (torch's BCELoss raises)
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float a = la[i];
    if (mode == 0) {                        // discriminator: 0.5*(L(real=a) + L(fake=b))
      const float b = lb[i];
      if (loss_type == 0) {
        acc += fmaxf(1.f - a, 0.f) + fmaxf(1.f + b, 0.f);
        if (da) da[i] = (1.f - a > 0.f) ? -0.5f * inv_n : 0.f;
        if (db) db[i] = (1.f + b > 0.f) ? 0.5f * inv_n : 0.f;
      } else if (loss_type == 2) {
        acc += bce_term(a, real_label, bad) + bce_term(b, fake_label, bad);
        if (da) da[i] = 0.5f * inv_n * bce_grad(a, real_label);
        if (db) db[i] = 0.5f * inv_n * bce_grad(b, fake_label);
      } else {
        acc += (a - real_label) * (a - real_label) + (b - fake_label) * (b - fake_label);
        if (da) da[i] = (a - real_label) * inv_n;
        if (db) db[i] = (b - fake_label) * inv_n;
      }
    } else {                                // generator
      if (loss_type == 0) {
        acc += -a;
        if (da) da[i] = -weight * inv_n;
      } else if (loss_type == 2) {
        acc += bce_term(a, real_label, bad);
        if (da) da[i] = weight * inv_n * bce_grad(a, real_label);
      } else {
        acc += (a - real_label) * (a - real_label);
        if (da) da[i] = 2.f * (a - real_label) * inv_n * weight;
      }
    }
  }
  acc = block_sum(acc, red);
  const float nbad = block_sum(bad ? 1.f : 0.f, red);
  if (threadIdx.x == 0) *loss = nbad > 0.f ? NAN : ((mode == 0) ? 0.5f * acc * inv_n : weight * acc * inv_n);
}

}  // namespace p2i
using namespace p2i;

extern "C" int p2i_recloss(const float* pred, const float* target, float k1_alpha, float* out3, float* dpred, float* scratch,
                           int B, int T, int HW, void* stream) {
  P2I_REQUIRE(pred && target && out3 && dpred && scratch, "null pointer");
  P2I_REQUIRE(T >= 2, "need T >= 2 for the temporal difference");
  hipStream_t s = (hipStream_t)stream;
  const int nrows = B * (T - 1);
  const size_t total = (size_t)B * T * HW;
  float* G = scratch;                       // nrows*HW
  float* rowkl = scratch + (size_t)nrows * HW;
  float* partial = rowkl + nrows;
  const int nblk = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
  P2I_REQUIRE(nrows + nblk <= 4096, "recloss scratch tail too small (B*(T-1) + blocks <= 4096)");
  // d(k1*reg)/d diff = k1 * (p_hat - q) / (temp * B)
  P2I_LAUNCH(kl_row_kernel, dim3(nrows), dim3(1024), 0, s, pred, target, G, rowkl, T, HW, k1_alpha / (0.1f * (float)B));
  P2I_LAUNCH(l1_grad_kernel, dim3(nblk), dim3(256), 0, s, pred, target, G, dpred, partial, T, HW, total, 1.f / (float)total);
  P2I_LAUNCH(recloss_final_kernel, dim3(1), dim3(256), 0, s, partial, nblk, rowkl, nrows, 1.f / (float)total, 1.f / (float)B,
                     k1_alpha, out3);
  return launch_status();
}

extern "C" int p2i_gan_loss(const float* logits_a, const float* logits_b, int n, int loss_type, int mode, float weight,
                            float real_label, float fake_label, float* loss, float* dlogits_a, float* dlogits_b, void* stream) {
  P2I_REQUIRE(logits_a && loss && n > 0, "null pointer");
  P2I_REQUIRE(loss_type >= 0 && loss_type <= 2, "loss_type: 0 hinge, 1 lsgan, 2 nsgan");
  P2I_REQUIRE(mode == 1 || logits_b, "discriminator mode needs both logit tensors");
  P2I_LAUNCH(gan_loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, logits_a, logits_b, n, loss_type, mode, weight,
                     real_label, fake_label, loss, dlogits_a, dlogits_b);
  return launch_status();
}
