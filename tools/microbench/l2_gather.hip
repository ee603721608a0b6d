// Diagnostic (GPU box): can a streaming evaluator afford ONE gather from an L2-resident table per hand?  (Candidate of the round-4 verdict:
// the non-flush ranking of a 7-card hand from a table indexed by a key of its rank multiset -- 49 205 multisets, so the table lives in
// L2, not LDS -- instead of the pairs / trips / quads extraction through five LDS lookups.)
// The kernel streams what pk_eval7_d streams (8 bytes in, 4 bytes out per hand, two hands per lane per iteration, 16-byte loads), and per
// hand makes ONE 4-byte gather from a table of TABLE_WORDS words at an index mixed from the hand, plus ALU dependent-chain instructions
// of the half-rate kind as a stand-in for the rest of the evaluation.  Reports hands/s; compare with k_eval7_tab_stream's ~290 G (125 VALU
// + 6 LDS lookups per hand).
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/l2_gather tools/microbench/l2_gather.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int ALU>
__device__ __forceinline__ uint32_t filler(uint32_t x) {
#pragma unroll
    for (int i = 0; i < ALU / 2; ++i) x = ((x << 7) | (x >> 25)) * 0x9E3779B1u;   // v_alignbit + v_mul_lo: two half-rate instructions, a bijection (nothing folds)
    return x;
}

template <int ALU, bool GATHER>
__global__ void __launch_bounds__(512, 8) k(const uint4 *__restrict__ hands, size_t pairs, uint2 *__restrict__ out, const uint32_t *__restrict__ tab, uint32_t mask) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += stride) {
        const uint4 w = hands[i];
        uint32_t k0 = (w.x * 0x9E3779B1u) ^ (w.y * 0x85EBCA77u), k1 = (w.z * 0x9E3779B1u) ^ (w.w * 0x85EBCA77u);
        k0 = (k0 >> 9) & mask; k1 = (k1 >> 9) & mask;
        uint32_t a = GATHER ? tab[k0] : k0, b = GATHER ? tab[k1] : k1;
        a = filler<ALU>(a ^ w.x); b = filler<ALU>(b ^ w.z);
        out[i] = make_uint2(a, b);
    }
}

template <int ALU, bool GATHER>
static void run(const uint4 *hands, size_t pairs, uint2 *out, const uint32_t *tab, uint32_t words, const char *what) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int grid = 256 * 32, reps = 10;
    hipLaunchKernelGGL((k<ALU, GATHER>), dim3(grid), dim3(512), 0, 0, hands, pairs, out, tab, words - 1);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<ALU, GATHER>), dim3(grid), dim3(512), 0, 0, hands, pairs, out, tab, words - 1);
    CHK(hipEventRecord(e1, 0));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    const double hps = 2.0 * pairs * reps / (ms * 1e-3);
    printf("%-44s ALU %3d  table %8u B : %7.1f G hands/s  (12 B/hand = %.2f TB/s)\n", what, ALU, GATHER ? words * 4 : 0, hps / 1e9, hps * 12 / 1e12);
}

int main() {
    const size_t m = (size_t)1 << 28, pairs = m / 2;
    uint4 *hands; uint2 *out; uint32_t *tab;
    CHK(hipMalloc(&hands, m * 8)); CHK(hipMalloc(&out, m * 4)); CHK(hipMalloc(&tab, 16u << 20));
    {   // pseudo-random hands: a cheap fill kernel would do; hipMemset pattern is enough for the stream, the index is mixed anyway
        uint32_t *h = (uint32_t *)malloc(64u << 20);
        uint32_t s = 12345;
        for (size_t i = 0; i < (64u << 20) / 4; ++i) { s = s * 1664525u + 1013904223u; h[i] = s; }
        for (size_t off = 0; off < m * 8; off += (64u << 20)) CHK(hipMemcpy((char *)hands + off, h, 64u << 20, hipMemcpyHostToDevice));
        CHK(hipMemcpy(tab, h, 16u << 20, hipMemcpyHostToDevice));
        free(h);
    }
    run<0, false>(hands, pairs, out, tab, 1, "stream only (no gather, no filler)");
    run<40, false>(hands, pairs, out, tab, 1, "stream + filler");
    run<90, false>(hands, pairs, out, tab, 1, "stream + filler");
    for (uint32_t words : {8192u, 65536u, 131072u, 262144u, 1048576u, 4194304u}) {
        run<0, true>(hands, pairs, out, tab, words, "stream + 1 gather/hand");
        run<40, true>(hands, pairs, out, tab, words, "stream + 1 gather/hand + filler");
        run<90, true>(hands, pairs, out, tab, words, "stream + 1 gather/hand + filler");
    }
    return 0;
}
