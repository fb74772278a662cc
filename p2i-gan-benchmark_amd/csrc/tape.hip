// Launch tape (tape.h): recording, replay and the stream-dependency primitives of the C ABI.  Host code only.
#include "common.h"
#include <string.h>
#include <mutex>
#include <vector>

namespace p2i {

enum { OP_KERNEL = 0, OP_MEMSET = 1, OP_RECORD = 2, OP_WAIT = 3 };

struct TapeOp {
  int kind;
  hipStream_t s;
  // kernel
  const void* fn;
  dim3 grid, block;
  size_t shmem;
  int nargs;
  size_t blob_off;           // into Tape::blob: the argument values, each at its own alignment
  size_t off_off;            // into Tape::offs: nargs offsets relative to blob_off
  // memset
  void* p;
  int value;
  size_t bytes;
  // event record / wait
  int slot;
};

struct Tape {
  hipStream_t origin;
  std::vector<TapeOp> ops;
  std::vector<unsigned char> blob;
  std::vector<size_t> offs;
  int nk = 0, nm = 0, ne = 0;
};

static thread_local Tape* g_rec = nullptr;
Tape* tape_current() { return g_rec; }

void tape_add_kernel(Tape* t, const void* fn, dim3 grid, dim3 block, size_t shmem, hipStream_t s, void* const* argv, const size_t* sizes,
                     const size_t* aligns, int nargs) {
  TapeOp op{};
  op.kind = OP_KERNEL; op.s = s; op.fn = fn; op.grid = grid; op.block = block; op.shmem = shmem; op.nargs = nargs;
  size_t base = (t->blob.size() + 63) & ~(size_t)63;
  op.blob_off = base;
  op.off_off = t->offs.size();
  size_t cur = 0;
  for (int i = 0; i < nargs; ++i) {
    const size_t a = aligns[i] ? aligns[i] : 1;
    cur = (cur + a - 1) / a * a;
    t->offs.push_back(cur);
    cur += sizes[i];
  }
  t->blob.resize(base + cur + 64);
  for (int i = 0; i < nargs; ++i) memcpy(t->blob.data() + base + t->offs[op.off_off + i], argv[i], sizes[i]);
  t->ops.push_back(op);
  ++t->nk;
}

void tape_add_memset(Tape* t, void* p, int value, size_t bytes, hipStream_t s) {
  TapeOp op{};
  op.kind = OP_MEMSET; op.s = s; op.p = p; op.value = value; op.bytes = bytes;
  t->ops.push_back(op);
  ++t->nm;
}

hipError_t memset_async(void* p, int value, size_t bytes, hipStream_t s) {
  if (g_rec) tape_add_memset(g_rec, p, value, bytes, s);
  return hipMemsetAsync(p, value, bytes, s);
}

// ---- stream dependencies: a table of lazily created events (no timing), addressed by slot.  An event may be recorded again while
// an earlier wait on it is still pending: a wait depends on the record that was current when the wait was enqueued.
constexpr int N_SLOTS = 256;
static thread_local hipEvent_t g_events[N_SLOTS];
static thread_local bool g_event_ok[N_SLOTS];

static hipError_t event_of(int slot, hipEvent_t* ev) {
  if (slot < 0 || slot >= N_SLOTS) return hipErrorInvalidValue;
  if (!g_event_ok[slot]) {
    hipError_t e = hipEventCreateWithFlags(&g_events[slot], hipEventDisableTiming);
    if (e != hipSuccess) return e;
    g_event_ok[slot] = true;
  }
  *ev = g_events[slot];
  return hipSuccess;
}

static int event_op(int kind, int slot, hipStream_t s, bool record_it) {
  hipEvent_t ev;
  hipError_t e = event_of(slot, &ev);
  if (e == hipSuccess) e = kind == OP_RECORD ? hipEventRecord(ev, s) : hipStreamWaitEvent(s, ev, 0);
  if (e != hipSuccess) { set_error("stream event %s failed: %s", kind == OP_RECORD ? "record" : "wait", hipGetErrorString(e)); return (int)e; }
  if (record_it && g_rec) {
    TapeOp op{};
    op.kind = kind; op.s = s; op.slot = slot;
    g_rec->ops.push_back(op);
    ++g_rec->ne;
  }
  return P2I_OK;
}

// ---- scratch of the deterministic reductions (common.h: det_take / det_reduce).  Registered once by the caller (caller-owned device
// memory); every call that needs scratch takes the next piece of the ring.  Pieces are reused only after the ring has wrapped:
// with the ring sized for many calls (the Python binding registers 64 MB for ~5 MB per train step) the kernels that used a piece
// before have long finished -- the steps of a training run are chained through the weights.
// (process-wide, one GPU per process: autograd runs a user's loss.backward() on its own thread, which must find the scratch too)
static float* g_det_part = nullptr;
static size_t g_det_floats = 0, g_det_pos = 0;
static unsigned* g_det_cnt = nullptr;
static int g_det_ncnt = 0, g_det_cpos = 0;
static std::mutex g_det_mu;

DetWs det_take(size_t floats, int counters) {
  DetWs w{nullptr, nullptr};
  std::lock_guard<std::mutex> lock(g_det_mu);
  floats = (floats + 63) & ~(size_t)63;
  if (!g_det_part || floats > g_det_floats || counters > g_det_ncnt) return w;
  if (g_det_pos + floats > g_det_floats) g_det_pos = 0;
  if (g_det_cpos + counters > g_det_ncnt) g_det_cpos = 0;
  w.part = g_det_part + g_det_pos;
  w.counter = g_det_cnt + g_det_cpos;
  g_det_pos += floats;
  g_det_cpos += counters;
  return w;
}

}  // namespace p2i
using namespace p2i;

extern "C" int p2i_det_workspace(float* part, int64_t part_floats, unsigned* counters, int n_counters) {
  std::lock_guard<std::mutex> lock(g_det_mu);
  if (part == nullptr) { g_det_part = nullptr; g_det_floats = 0; g_det_cnt = nullptr; g_det_ncnt = 0; g_det_pos = 0; g_det_cpos = 0; return P2I_OK; }
  P2I_REQUIRE(part_floats >= (1 << 16) && counters && n_counters >= 1024, "deterministic-reduction scratch too small");
  g_det_part = part; g_det_floats = (size_t)part_floats; g_det_cnt = counters; g_det_ncnt = n_counters; g_det_pos = 0; g_det_cpos = 0;
  return P2I_OK;
}

extern "C" int p2i_event_record(int slot, void* stream) { return event_op(OP_RECORD, slot, (hipStream_t)stream, true); }
extern "C" int p2i_event_wait(int slot, void* stream) { return event_op(OP_WAIT, slot, (hipStream_t)stream, true); }

extern "C" int p2i_tape_begin(void* origin_stream) {
  P2I_REQUIRE(g_rec == nullptr, "a tape is already being recorded on this thread");
  g_rec = new Tape();
  g_rec->origin = (hipStream_t)origin_stream;
  return P2I_OK;
}

extern "C" int p2i_tape_end(void** tape_out) {
  P2I_REQUIRE(g_rec != nullptr && tape_out != nullptr, "no tape is being recorded");
  *tape_out = g_rec;
  g_rec = nullptr;
  return P2I_OK;
}

extern "C" int p2i_tape_info(const void* tape, int* counts4) {
  P2I_REQUIRE(tape && counts4, "null pointer");
  const Tape* t = static_cast<const Tape*>(tape);
  std::vector<hipStream_t> seen;
  for (const TapeOp& op : t->ops) {
    bool f = false;
    for (hipStream_t s : seen) f = f || s == op.s;
    if (!f) seen.push_back(op.s);
  }
  counts4[0] = t->nk; counts4[1] = t->nm; counts4[2] = t->ne; counts4[3] = (int)seen.size();
  return P2I_OK;
}

extern "C" int p2i_tape_replay(const void* tape, void* origin_stream) {
  P2I_REQUIRE(tape != nullptr, "null tape");
  P2I_REQUIRE(g_rec == nullptr, "replay while recording");
  const Tape* t = static_cast<const Tape*>(tape);
  hipStream_t now = (hipStream_t)origin_stream;
  void* argv[64];
  for (const TapeOp& op : t->ops) {
    hipStream_t s = op.s == t->origin ? now : op.s;       // work recorded on the origin stream follows the caller's current stream
    hipError_t e = hipSuccess;
    switch (op.kind) {
      case OP_KERNEL: {
        if (op.nargs > 64) { set_error("tape: kernel with %d arguments", op.nargs); return P2I_EINVAL; }
        unsigned char* base = const_cast<unsigned char*>(t->blob.data()) + op.blob_off;
        for (int i = 0; i < op.nargs; ++i) argv[i] = base + t->offs[op.off_off + i];
        e = hipLaunchKernel(op.fn, op.grid, op.block, argv, op.shmem, s);
        break;
      }
      case OP_MEMSET: e = hipMemsetAsync(op.p, op.value, op.bytes, s); break;
      case OP_RECORD: {
        int rc = event_op(OP_RECORD, op.slot, s, false);
        if (rc != P2I_OK) return rc;
        break;
      }
      case OP_WAIT: {
        int rc = event_op(OP_WAIT, op.slot, s, false);
        if (rc != P2I_OK) return rc;
        break;
      }
    }
    if (e != hipSuccess) { set_error("tape replay failed: %s", hipGetErrorString(e)); return (int)e; }
  }
  return launch_status();
}

extern "C" int p2i_tape_free(void* tape) {
  P2I_REQUIRE(tape != nullptr && tape != g_rec, "bad tape");
  delete static_cast<Tape*>(tape);
  return P2I_OK;
}
