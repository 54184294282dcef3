// lane_swap.h - v_permlane16_swap / v_permlane32_swap of a value WITH ITSELF (the cross-row step of a wave reduction).
// Trap met in round 4 (hipcc, ROCm 7.2): read the two results with __uint_as_float(a[i]) or through unsigned locals, never
// with __builtin_bit_cast(float, a[1]) - bit_cast of an ext-vector ELEMENT lvalue reads element 0 whatever the index
// (the front end emits `extractelement 0` twice), which turns a row-pair sum into twice the even row's value.
// Internal to libcqs_hip.so.
#pragma once
#if defined(__HIPCC__)
namespace cqs {
typedef unsigned lane_u2 __attribute__((ext_vector_type(2)));
// a[0] = the value of the EVEN 16-lane row of this lane's row pair, a[1] = of the ODD one, in both rows
__device__ __forceinline__ lane_u2 swap16_self(unsigned x) { return __builtin_amdgcn_permlane16_swap(x, x, false, false); }
// a[0] = the value of lanes 0-31 (same lane & 31), a[1] = of lanes 32-63, in both halves
__device__ __forceinline__ lane_u2 swap32_self(unsigned x) { return __builtin_amdgcn_permlane32_swap(x, x, false, false); }
}  // namespace cqs
#endif
