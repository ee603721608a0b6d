// Diagnostic (GPU box): issue cost of the VALU instructions the step machine is made of, per wave-instruction, for one
// wave per SIMD (the headline configuration: 65 536 tables = 1 024 waves on 1 024 SIMDs) and for 2 / 4 / 8 waves per SIMD,
// as dependent chains and as 8 independent chains.  Prints cycles per wave-instruction at the 2.4 GHz nominal clock and
// the chip-wide rate; the roofline `peak` of bench.py's VALU block is taken from the best plain-ALU row.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates tools/microbench/valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

#define REP8(x) x x x x x x x x
#define OP32(name, ins)                                                                         \
    struct name { static constexpr const char *label = #ins;                                    \
        static __device__ __forceinline__ void dep(unsigned &a, unsigned b) { asm volatile(#ins " %0, %0, %1" : "+v"(a) : "v"(b)); } };
OP32(AddU32, v_add_u32)
OP32(XorB32, v_xor_b32)
OP32(AndB32, v_and_b32)
OP32(MulLo, v_mul_lo_u32)
OP32(MulHi, v_mul_hi_u32)
OP32(MulU24, v_mul_u32_u24)
OP32(LshlRev, v_lshlrev_b32)
struct Cndmask { static constexpr const char *label = "v_cndmask_b32";
    static __device__ __forceinline__ void dep(unsigned &a, unsigned b) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b)); } };
struct CndmaskSgpr { static constexpr const char *label = "v_cndmask_b32 (sgpr pair)";
    static __device__ __forceinline__ void dep(unsigned &a, unsigned b) {
        unsigned long long m = 0x5555555555555555ull;
        asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "s"(m)); } };
struct CmpCnd { static constexpr const char *label = "v_cmp_lt_u32 + v_cndmask";
    static __device__ __forceinline__ void dep(unsigned &a, unsigned b) {
        asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc"); } };
struct CmpCndS { static constexpr const char *label = "v_cmp_lt_u32 s[] + v_cndmask";
    static __device__ __forceinline__ void dep(unsigned &a, unsigned b) {
        unsigned long long m;
        asm volatile("v_cmp_lt_u32_e64 %2, %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a), "+v"(b), "=&s"(m)); } };
OP32(BfeU32_, v_lshrrev_b32)
struct Bfe { static constexpr const char *label = "v_bfe_u32";
    static __device__ __forceinline__ void dep(unsigned &a, unsigned b) { asm volatile("v_bfe_u32 %0, %0, %1, 13" : "+v"(a) : "v"(b)); } };
struct Perm { static constexpr const char *label = "v_perm_b32";
    static __device__ __forceinline__ void dep(unsigned &a, unsigned b) { asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a) : "v"(b)); } };
struct Or3 { static constexpr const char *label = "v_or3_b32";
    static __device__ __forceinline__ void dep(unsigned &a, unsigned b) { asm volatile("v_or3_b32 %0, %0, %1, %1" : "+v"(a) : "v"(b)); } };
struct Max3 { static constexpr const char *label = "v_max3_u32";
    static __device__ __forceinline__ void dep(unsigned &a, unsigned b) { asm volatile("v_max3_u32 %0, %0, %1, %1" : "+v"(a) : "v"(b)); } };
struct Bitop3 { static constexpr const char *label = "v_bitop3_b32";
    static __device__ __forceinline__ void dep(unsigned &a, unsigned b) { asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x6c" : "+v"(a) : "v"(b)); } };
struct LshlOr { static constexpr const char *label = "v_lshl_or_b32";
    static __device__ __forceinline__ void dep(unsigned &a, unsigned b) { asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a) : "v"(b)); } };
struct Ffbh { static constexpr const char *label = "v_ffbh_u32";
    static __device__ __forceinline__ void dep(unsigned &a, unsigned b) { asm volatile("v_ffbh_u32 %0, %0" : "+v"(a)); } };
struct MaxU32 { static constexpr const char *label = "v_max_u32";
    static __device__ __forceinline__ void dep(unsigned &a, unsigned b) { asm volatile("v_max_u32 %0, %0, %1" : "+v"(a) : "v"(b)); } };
struct Bcnt { static constexpr const char *label = "v_bcnt_u32_b32";
    static __device__ __forceinline__ void dep(unsigned &a, unsigned b) { asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a) : "v"(b)); } };

template <typename OP, int ILP>
__global__ void __launch_bounds__(256) k32(int iters, unsigned *sink) {
    unsigned x[8], b = threadIdx.x | 1;
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
        if (ILP == 1) { REP8(REP8(OP::dep(x[0], b);)) }
        else { REP8(OP::dep(x[0], b); OP::dep(x[1], b); OP::dep(x[2], b); OP::dep(x[3], b); OP::dep(x[4], b); OP::dep(x[5], b); OP::dep(x[6], b); OP::dep(x[7], b);) }
    }
    unsigned r = 0;
    for (int i = 0; i < 8; ++i) r ^= x[i];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

#define OP64(name, ins)                                                                       \
    struct name { static constexpr const char *label = #ins;                                  \
        static __device__ __forceinline__ void dep(double &a, double b) { asm volatile(#ins " %0, %0, %1" : "+v"(a) : "v"(b)); } };
OP64(AddF64, v_add_f64)
OP64(MulF64, v_mul_f64)
struct FmaF64 { static constexpr const char *label = "v_fma_f64";
    static __device__ __forceinline__ void dep(double &a, double b) { asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a) : "v"(b)); } };
struct CmpF64 { static constexpr const char *label = "v_cmp_gt_f64 (to vcc)";
    static __device__ __forceinline__ void dep(double &a, double b) { asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc"); } };
struct Lshl64 { static constexpr const char *label = "v_lshlrev_b64";
    static __device__ __forceinline__ void dep(double &a, double b) {
        unsigned long long &x = reinterpret_cast<unsigned long long &>(a);
        asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(x)); } };
struct LshlAdd64 { static constexpr const char *label = "v_lshl_add_u64";
    static __device__ __forceinline__ void dep(double &a, double b) {
        unsigned long long &x = reinterpret_cast<unsigned long long &>(a);
        asm volatile("v_lshl_add_u64 %0, %0, 0, %0" : "+v"(x)); } };
struct MaxF64 { static constexpr const char *label = "v_max_f64";
    static __device__ __forceinline__ void dep(double &a, double b) { asm volatile("v_max_f64 %0, %0, %1" : "+v"(a) : "v"(b)); } };
struct Mad64 { static constexpr const char *label = "v_mad_u64_u32";
    static __device__ __forceinline__ void dep(double &a, double b) {
        unsigned long long &x = reinterpret_cast<unsigned long long &>(a);
        unsigned lo = (unsigned)x;
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x) : "v"(lo), "v"(0x9E3779B9u) : "vcc"); } };

template <typename OP, int ILP>
__global__ void __launch_bounds__(256) k64(int iters, double *sink) {
    double x[8], b = 1.0000001;
    for (int i = 0; i < 8; ++i) x[i] = 1.0 + threadIdx.x * 1e-9 + i;
    for (int it = 0; it < iters; ++it) {
        if (ILP == 1) { REP8(REP8(OP::dep(x[0], b);)) }
        else { REP8(OP::dep(x[0], b); OP::dep(x[1], b); OP::dep(x[2], b); OP::dep(x[3], b); OP::dep(x[4], b); OP::dep(x[5], b); OP::dep(x[6], b); OP::dep(x[7], b);) }
    }
    double r = 0;
    for (int i = 0; i < 8; ++i) r += x[i];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// What a "cycle" above is worth: the rows divide elapsed TIME by the 2.4 GHz nominal clock.  This kernel reads the shader-clock counter
// (s_memtime) and the constant 100 MHz counter (s_memrealtime) around the same independent v_add_u32 stream, one wave per SIMD: the
// ratio is the clock the SIMDs really ran at, and counter ticks / instructions = issue cost in REAL shader cycles.
__global__ void __launch_bounds__(256) k_clock(int iters, unsigned long long *out, unsigned *sink) {
    unsigned x[8], b = threadIdx.x | 1;
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        REP8(AddU32::dep(x[0], b); AddU32::dep(x[1], b); AddU32::dep(x[2], b); AddU32::dep(x[3], b); AddU32::dep(x[4], b); AddU32::dep(x[5], b); AddU32::dep(x[6], b); AddU32::dep(x[7], b);)
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    unsigned r = 0;
    for (int i = 0; i < 8; ++i) r ^= x[i];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = r1 - r0; }
}

template <typename K, typename S>
static void row(const char *label, K kern, S *sink, int ilp) {
    const int iters = 4000;
    printf("%-24s ilp%d ", label, ilp);
    for (int wps : {1, 2, 4, 8}) {  // waves per SIMD: 256 CUs x 4 SIMDs x wps waves = grid of 256-thread (4-wave) blocks
        int grid = 256 * wps;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, iters, sink);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, iters, sink);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        double insts = (double)grid * 4 * iters * 64;              // wave-instructions
        double per_simd = insts / 1024.0;                          // issued on each SIMD
        double cyc = ms * 1e-3 * 2.4e9 / per_simd;                 // cycles per wave-instruction per SIMD at 2.4 GHz
        printf(" | %dw/SIMD %5.2f cyc %6.1f Ginst/s", wps, cyc, insts / (ms * 1e-3) / 1e9);
    }
    printf("\n");
}

int main() {
    unsigned *s32; double *s64;
    (void)hipMalloc(&s32, 2048 * 256 * 4); (void)hipMalloc(&s64, 2048 * 256 * 8);
#define R32(OP) row(OP::label, k32<OP, 1>, s32, 1); row(OP::label, k32<OP, 8>, s32, 8);
#define R64(OP) row(OP::label, k64<OP, 1>, s64, 1); row(OP::label, k64<OP, 8>, s64, 8);
    R32(AddU32) R32(XorB32) R32(AndB32) R32(Cndmask) R32(CndmaskSgpr) R32(CmpCnd) R32(CmpCndS) R32(LshlRev) R32(Bcnt) R32(MulU24) R32(MulLo) R32(MulHi)
    R32(Bfe) R32(Perm) R32(Or3) R32(Max3) R32(MaxU32) R32(Bitop3) R32(LshlOr) R32(Ffbh)
    R64(AddF64) R64(MulF64) R64(FmaF64) R64(CmpF64) R64(Mad64) R64(Lshl64) R64(LshlAdd64) R64(MaxF64)
    {   // the real clock, and the lone wave's issue cost in real shader cycles
        unsigned long long *d, h[512];
        (void)hipMalloc(&d, sizeof(h));
        const int iters = 20000;
        for (int wps : {1}) {   // (one wave per SIMD: with more, a wave's own elapsed ticks no longer say what the SIMD issued)
            hipLaunchKernelGGL(k_clock, dim3(256 * wps), dim3(256), 0, 0, iters, d, s32);
            hipLaunchKernelGGL(k_clock, dim3(256 * wps), dim3(256), 0, 0, iters, d, s32);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            double ticks = 0, real = 0;
            for (int i = 0; i < 256; ++i) { ticks += (double)h[2 * i]; real += (double)h[2 * i + 1]; }
            const double mhz = ticks / real * 100.0;
            printf("clock probe, %d wave(s) per SIMD, independent v_add_u32 stream: shader-clock counter / 100 MHz counter = %.0f MHz;  %.2f counter ticks per "
                   "wave-instruction (%.2f ns -> %.2f cycles at the nominal 2.4 GHz)\n", wps, mhz, ticks / 256 / ((double)iters * 64 * wps),
                   real / 256 * 10.0 / ((double)iters * 64 * wps), real / 256 * 10.0 * 2.4 / ((double)iters * 64 * wps));
        }
    }
    return 0;
}
