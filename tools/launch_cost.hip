// Host cost of the HIP runtime calls a train step is made of (tools/tape_probe.py: ~500 calls cost ~5 ms whoever issues them).
// build: hipcc --offload-arch=gfx950 -O2 tools/launch_cost.hip -o tools/bin/launch_cost ; run on the GPU box
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { int v[512]; };      // 2 KB by value (PatchGeom-sized)
struct Mid { int v[75]; };       // 300 B (X6cGeom-sized)
__global__ void k_small(float* p) { if (p && threadIdx.x == 9999) p[0] = 1.f; }
__global__ void k_mid(Mid m, float* p) { if (p && threadIdx.x == 9999) p[0] = m.v[3]; }
__global__ void k_big(Big b, float* p) { if (p && threadIdx.x == 9999) p[0] = b.v[3]; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const int N = 2000;
  hipStream_t s0, s1, s2;
  hipStreamCreateWithFlags(&s0, hipStreamNonBlocking); hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  hipEvent_t ev[64];
  for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
  float* d; hipMalloc(&d, 1 << 20);
  Mid m{}; Big b{};
  auto run = [&](const char* name, auto&& body) {
    for (int i = 0; i < 50; ++i) body(i);
    hipDeviceSynchronize();
    const double t0 = now();
    for (int i = 0; i < N; ++i) body(i);
    const double t1 = now();
    hipDeviceSynchronize();
    const double t2 = now();
    printf("%-58s host %6.2f us/iter, with drain %6.2f us/iter\n", name, 1e6 * (t1 - t0) / N, 1e6 * (t2 - t0) / N);
  };
  run("small kernel, one stream", [&](int) { hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s0, d); });
  run("300-B-argument kernel, one stream", [&](int) { hipLaunchKernelGGL(k_mid, dim3(1), dim3(64), 0, s0, m, d); });
  run("2-KB-argument kernel, one stream", [&](int) { hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, s0, b, d); });
  run("small kernel, 256 workgroups x 512 threads, 148 KB LDS", [&](int) { hipLaunchKernelGGL(k_small, dim3(256), dim3(512), 148 * 1024, s0, d); });
  run("memset 4 KB", [&](int) { hipMemsetAsync(d, 0, 4096, s0); });
  run("memset 4 B", [&](int) { hipMemsetAsync(d, 0, 4, s0); });
  run("event record + cross-stream wait (no kernels)", [&](int i) { hipEventRecord(ev[i & 63], s0); hipStreamWaitEvent(s1, ev[i & 63], 0); });
  run("kernel on s0; fork to s1; kernel on s1", [&](int i) {
    hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s0, d);
    hipEventRecord(ev[i & 63], s0); hipStreamWaitEvent(s1, ev[i & 63], 0);
    hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s1, d);
  });
  run("two kernels on s0 (same work, no fork)", [&](int) {
    hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s0, d);
    hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s0, d);
  });
  run("kernel s0; fork s1; kernel s1; join s0 (a side-stream wgrad)", [&](int i) {
    hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s0, d);
    hipEventRecord(ev[i & 63], s0); hipStreamWaitEvent(s1, ev[i & 63], 0);
    hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s1, d);
    hipEventRecord(ev[(i + 32) & 63], s1); hipStreamWaitEvent(s0, ev[(i + 32) & 63], 0);
  });
  return 0;
}
