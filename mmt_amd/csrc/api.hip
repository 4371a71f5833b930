// ABI bookkeeping for libmmt_hip.so
#include "../../include/mmt_hip.h"

extern "C" int mmt_abi_version(void) { return MMT_ABI_VERSION; }
extern "C" const char* mmt_build_info(void) {
  return "libmmt_hip gfx950 (CDNA4) bf16-MFMA " __DATE__ " " __TIME__;
}

// Stream ordering helper: an event of a small ring is recorded on `from` and waited for on `to`.  Reusing ring entries is
// safe because record and wait are issued back to back by the same host thread (a wait refers to the record that
// precedes it), eagerly as well as under stream capture (where the pair becomes a graph edge).
#include <hip/hip_runtime.h>
#include <mutex>
namespace {
constexpr int RING = 64;
hipEvent_t g_ring[RING];
int g_ring_n = 0, g_ring_i = 0;
std::mutex g_ring_mu;
}  // namespace
extern "C" int mmt_stream_fork(void* from, void* to) {
  if (from == to) return 0;
  std::lock_guard<std::mutex> lock(g_ring_mu);
  if (g_ring_n < RING && g_ring_i == g_ring_n) {
    hipError_t rc = hipEventCreateWithFlags(&g_ring[g_ring_n], hipEventDisableTiming);
    if (rc != hipSuccess) return (int)rc;
    ++g_ring_n;
  }
  hipEvent_t ev = g_ring[g_ring_i];
  g_ring_i = (g_ring_i + 1) % RING;
  hipError_t rc = hipEventRecord(ev, (hipStream_t)from);
  if (rc != hipSuccess) return (int)rc;
  return (int)hipStreamWaitEvent((hipStream_t)to, ev, 0);
}
