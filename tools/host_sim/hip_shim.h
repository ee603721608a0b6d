// Dev-only: lets pk_device.hpp compile with g++ as a ONE-LANE "wavefront" so the step machine can be diffed against
// the oracle on the build container (no GPU).  Not part of the product, not used by tests' parity claims.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#define PK_WAVE 1
#define __device__
#define __constant__
struct uint4 { unsigned x, y, z, w; };
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(x)
struct dim3_ { unsigned x = 0, y = 0, z = 0; };
static dim3_ threadIdx, blockIdx, blockDim;
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffs(unsigned x) { return __builtin_ffs((int)x); }
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }  // device: v_ffbh_u32 -> 32 (as -1 -> clamped) for 0
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xffffffu) * (b & 0xffffffu); }
static inline unsigned long long __ballot(int p) { return p ? 1ull : 0ull; }
static inline int __any(int p) { return p; }
static inline void __syncthreads() {}
static inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned, unsigned v) { return v; }
static inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned, unsigned v) { return v; }
using std::min;
using std::max;
