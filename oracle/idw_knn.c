/* CPU restatement in C of the k=4 IDW of the reference's CPU path — TEST INFRASTRUCTURE ONLY
 * (see oracle/__init__.py).  Documents, and lets tests verify bit-for-bit, how the ATen
 * primitives called by idw_3d_knn (p2igan_bench/modules/layer.py:259-293) behave on fp32 CPU:
 *   torch.cdist (rows > 25 -> "mm" Euclidean path): d2 = [-2x,-2y,-2z,|q|^2,1].[px,py,pz,1,|p|^2]
 *     accumulated by MKL sgemm as one k-ordered fmaf chain, then sqrt(clamp_min(., 0));
 *   torch.topk(k=4, largest=False) with N >= 256 (k*64 <= N): std::partial_sort over
 *     (value, index) pairs = libstdc++ __heap_select (max-heap of 4, replace root on strictly
 *     smaller value) + __sort_heap.
 * Validated against torch itself in tests/test_oracle_idw_c.py.  Build: make -C oracle. */
#include <math.h>
#include <stdint.h>

typedef struct { float d; int i; } he;

static void adjust_heap(he* f, int hole, int len, he value) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (f[child].d < f[child - 1].d) child--;
    f[hole] = f[child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    f[hole] = f[child - 1];
    hole = child - 1;
  }
  int parent = (hole - 1) / 2;
  while (hole > top && f[parent].d < value.d) {
    f[hole] = f[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  f[hole] = value;
}

/* gx,gy,gz: linspace tables; pts: N x 3 (x,y,z); vals: N.  out: Q floats; sel: Q x 4 indices and
 * seld: Q x 4 distances, ascending (either may be NULL) */
void idw_knn4(const float* gx, const float* gy, const float* gz, int T, int H, int W, const float* pts,
              const float* vals, int N, float tau, float* out, int32_t* sel, float* seld) {
  /* voxels are independent: threads change nothing of any voxel's arithmetic (the tests' dense masks are 4e9 pair evaluations) */
#pragma omp parallel for collapse(2) schedule(static)
  for (int t = 0; t < T; ++t)
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        const long q = ((long)t * H + y) * W + x;
        const float qx = gx[x], qy = gy[y], qz = gz[t];
        const float a0 = -2.f * qx, a1 = -2.f * qy, a2 = -2.f * qz;
        const float n1 = (qx * qx + qy * qy) + qz * qz;
        he h[4];
        for (int j = 0; j < N; ++j) {
          const float px = pts[3 * j], py = pts[3 * j + 1], pz = pts[3 * j + 2];
          const float n2 = (px * px + py * py) + pz * pz;
          float acc = a0 * px;
          acc = fmaf(a1, py, acc);
          acc = fmaf(a2, pz, acc);
          acc = fmaf(n1, 1.f, acc);
          acc = fmaf(1.f, n2, acc);
          const float d = sqrtf(acc > 0.f ? acc : 0.f);
          if (j < 4) {
            h[j].d = d; h[j].i = j;
            if (j == 3) {                       /* __make_heap */
              for (int parent = 1; parent >= 0; --parent) adjust_heap(h, parent, 4, h[parent]);
            }
          } else if (d < h[0].d) {               /* __pop_heap(first, middle, i) */
            he v = {d, j};
            adjust_heap(h, 0, 4, v);
          }
        }
        for (int last = 3; last >= 1; --last) {  /* __sort_heap */
          he v = h[last];
          h[last] = h[0];
          adjust_heap(h, 0, last, v);
        }
        float w[4], ws = 0.f;
        for (int k = 0; k < 4; ++k) { const float inv = 1.0f / (h[k].d + tau); w[k] = inv * inv; }
        ws = ((w[0] + w[1]) + w[2]) + w[3];
        ws = ws + 1e-12f;
        float o = 0.f;
        o = ((vals[h[0].i] * (w[0] / ws) + vals[h[1].i] * (w[1] / ws)) + vals[h[2].i] * (w[2] / ws)) + vals[h[3].i] * (w[3] / ws);
        out[q] = o;
        if (sel) for (int k = 0; k < 4; ++k) sel[4 * q + k] = h[k].i;
        if (seld) for (int k = 0; k < 4; ++k) seld[4 * q + k] = h[k].d;
      }
}
