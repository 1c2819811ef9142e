// How a host thread waits for the device (shared by every translation unit of the library).
#pragma once
#include <hip/hip_runtime.h>
#include <sys/prctl.h>
#include <time.h>

#include <atomic>

namespace sg {

// 0 (default): the HIP runtime's own wait -- hipEventSynchronize / hipStreamSynchronize, which on ROCm poll and yield:
// the lowest latency, and one busy CPU per waiting thread.  > 0 ("host.wait_sleep_us"): the thread polls the event /
// stream every that many microseconds and SLEEPS in between.  A lone proof wants the first (a dozen waits per proof, each
// on its critical path); a batch with sixteen to thirty-two proofs in flight wants the second: its threads wait most of
// the time, thirty yielding pollers eat the process's CPU quota (on the boxes of this pool: 16 cores under a 256-CPU
// mask, and eight ranks of one node share such a quota), and a few tens of microseconds of wake-up latency cost a proof
// that is in flight for a hundred milliseconds nothing (DESIGN.md sections 4.4 and 5; bench.py --cpu-share).
inline std::atomic<int>& host_wait_sleep_us() {
  static std::atomic<int> v{0};
  return v;
}
// The naps want a timer slack of 1 us (the default of 50 us would be added to every one of them).  The waiting threads are the
// CALLER'S -- halo2's / rayon's workers calling through the shim -- so the slack is the caller's property: it is lowered for the
// length of ONE wait and put back before the wait returns (two prctl calls per wait that naps at all, none for a wait that
// finds its event complete).
struct HostNapSlack {
  long saved = -1;
  void nap(int us) {
    if (saved < 0) {
      saved = prctl(PR_GET_TIMERSLACK, 0UL, 0UL, 0UL, 0UL);
      if (saved < 0) saved = 0;      // (0 = "the thread's default" when it is written back)
      (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0UL, 0UL, 0UL);
    }
    struct timespec ts = {0, (long)us * 1000L};
    (void)nanosleep(&ts, nullptr);
  }
  ~HostNapSlack() {
    if (saved >= 0) (void)prctl(PR_SET_TIMERSLACK, (unsigned long)saved, 0UL, 0UL, 0UL);
  }
};
inline hipError_t host_wait_event(hipEvent_t e) {
  const int us = host_wait_sleep_us().load(std::memory_order_relaxed);
  if (us <= 0) return hipEventSynchronize(e);
  HostNapSlack slack;
  for (;;) {
    const hipError_t q = hipEventQuery(e);
    if (q != hipErrorNotReady) return q;
    slack.nap(us);
  }
}
inline hipError_t host_wait_stream(hipStream_t s) {
  const int us = host_wait_sleep_us().load(std::memory_order_relaxed);
  if (us <= 0) return hipStreamSynchronize(s);
  HostNapSlack slack;
  for (;;) {
    const hipError_t q = hipStreamQuery(s);
    if (q != hipErrorNotReady) return q;
    slack.nap(us);
  }
}

// Device -> host memory that may be pageable, complete on return.  A copy into pageable memory makes the runtime wait for
// everything ahead of it on the stream INSIDE hipMemcpyAsync, busily (measured in the proof batch: 240 us per call, nine
// calls per proof, 2 ms of CPU per proof -- tools/batch_cpu_profile.py, tools/hip_api_counts.py); with sleeping waits on,
// the stream is drained asleep first and the copy finds it empty.  With the runtime's own wait (a lone proof) nothing
// changes: one call, one synchronisation.
inline hipError_t host_copy_d2h(void* host, const void* dev, size_t bytes, hipStream_t s) {
  hipError_t e = hipSuccess;
  if (host_wait_sleep_us().load(std::memory_order_relaxed) > 0) e = host_wait_stream(s);
  if (e == hipSuccess) e = hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = host_wait_stream(s);
  return e;
}

}  // namespace sg
