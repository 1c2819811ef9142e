// Wave priority of every kernel that is not msm_accumulate.
//
// A SIMD's issue arbiter serves the oldest wave first.  msm_accumulate's waves are persistent (old) and each is a dependent
// chain of multiply-adds that nearly saturates the issue port on its own, so a wave of any other kernel that lands beside
// them -- the next MSM's sort, the previous one's bucket reduction, a proof's transforms under its commitments -- gets a
// tenth of the issue slots and its whole stream stalls behind it (profiles/r03_sweeps/persistent_accumulate.txt).
// s_setprio ranks before age: with every other kernel at priority 3 those kernels run at their own speed and the
// accumulation takes the slots they leave.  Without an accumulation on the device all waves are equal and nothing changes.
// One copy of the switch per translation unit (no relocatable device code): sg_set_param("side_prio", 0 | 1) sets them all.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace sg {
static __device__ uint32_t g_side_prio = 1;
static __device__ __forceinline__ void side_kernel_prio() {
  if (g_side_prio) __builtin_amdgcn_s_setprio(3);
}
hipError_t msm_set_side_prio(uint32_t on);
hipError_t ntt_set_side_prio(uint32_t on);
hipError_t poly_set_side_prio(uint32_t on);
hipError_t quotient_set_side_prio(uint32_t on);
hipError_t gates_set_side_prio(uint32_t on);
hipError_t witness_set_side_prio(uint32_t on);
hipError_t abi_set_side_prio(uint32_t on);
}  // namespace sg
#define SG_DEFINE_SIDE_PRIO_SETTER(name) \
  hipError_t name(uint32_t on) { return hipMemcpyToSymbol(HIP_SYMBOL(g_side_prio), &on, sizeof on); }
