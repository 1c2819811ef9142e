// Retired device memory (shared by the engines of every lane).
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <vector>

namespace sg {

// Device memory that is outgrown or no longer needed WHILE THE LIBRARY IS RUNNING is retired, not freed: hipFree waits for
// the whole device to go idle, and with several host threads feeding the GPU (lanes, proofs in flight) that wait stalls
// the caller for as long as the others keep it busy (measured: a lane warming up under load dropped a 100 proofs/s
// batch to 5 proofs/s).  Work spaces grow geometrically, so the retired memory is bounded by the final sizes; it is
// returned by retired_device_memory_collect() (sg_shutdown, or any moment the caller knows the device is idle).
inline std::mutex& retired_mu() {
  static std::mutex m;
  return m;
}
inline std::vector<void*>& retired_list() {
  static std::vector<void*>* v = new std::vector<void*>();   // never destroyed: used from destructors at exit
  return *v;
}
inline void retire_device_memory(void* p) {
  if (!p) return;
  std::lock_guard<std::mutex> lk(retired_mu());
  retired_list().push_back(p);
}
inline void retired_device_memory_collect() {
  std::vector<void*> mine;
  {
    std::lock_guard<std::mutex> lk(retired_mu());
    mine.swap(retired_list());
  }
  for (void* p : mine) (void)hipFree(p);
}

}  // namespace sg
