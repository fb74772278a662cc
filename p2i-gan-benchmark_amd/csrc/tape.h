// Launch tape: the library's kernel launches, memsets and stream dependencies of ONE pass over the hot path, recorded while they
// execute and re-enqueued later by a single C call (p2i_tape_replay) -- the native step sequencer of include/p2i_hip.h.  Every
// launch site goes through P2I_LAUNCH (a typed wrapper of hipLaunchKernel), which is where the recording happens: no entry point
// knows about tapes.  Host code only.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <tuple>
#include <utility>

namespace p2i {

struct Tape;
Tape* tape_current();      // the tape this thread is recording into, or nullptr
void tape_add_kernel(Tape* t, const void* fn, dim3 grid, dim3 block, size_t shmem, hipStream_t s, void* const* argv, const size_t* sizes,
                     const size_t* aligns, int nargs);
void tape_add_memset(Tape* t, void* p, int value, size_t bytes, hipStream_t s);

// hipMemsetAsync that a recording tape sees
hipError_t memset_async(void* p, int value, size_t bytes, hipStream_t s);

template <typename... KArgs, size_t... I>
inline void launch_impl(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t shmem, hipStream_t s, std::tuple<KArgs...>& vals,
                        std::index_sequence<I...>) {
  void* argv[sizeof...(KArgs) ? sizeof...(KArgs) : 1] = {const_cast<void*>(static_cast<const void*>(&std::get<I>(vals)))...};
  if (Tape* t = tape_current()) {
    const size_t sizes[sizeof...(KArgs) ? sizeof...(KArgs) : 1] = {sizeof(KArgs)...};
    const size_t aligns[sizeof...(KArgs) ? sizeof...(KArgs) : 1] = {alignof(KArgs)...};
    tape_add_kernel(t, reinterpret_cast<const void*>(kernel), grid, block, shmem, s, argv, sizes, aligns, (int)sizeof...(KArgs));
  }
  (void)hipLaunchKernel(reinterpret_cast<const void*>(kernel), grid, block, argv, shmem, s);
}

// launch(kernel, grid, block, dynamic LDS bytes, stream, kernel arguments...): the arguments are converted to the kernel's parameter
// types first (as a <<<>>> launch would), then passed by address to hipLaunchKernel
template <typename... KArgs, typename... Args>
inline void launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t shmem, hipStream_t s, Args&&... args) {
  static_assert(sizeof...(KArgs) == sizeof...(Args), "kernel argument count");
  std::tuple<KArgs...> vals{static_cast<KArgs>(std::forward<Args>(args))...};
  launch_impl(kernel, grid, block, shmem, s, vals, std::index_sequence_for<KArgs...>{});
}

}  // namespace p2i

#define P2I_LAUNCH(kernel, grid, block, shmem, stream, ...) p2i::launch(kernel, grid, block, shmem, stream, ##__VA_ARGS__)
