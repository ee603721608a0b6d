// Diagnostic (GPU box): (1) which SIMD each wave of a 512-thread workgroup lands on; (2) what a co-resident helper wave
// costs an owner wave that issues a dependent VALU stream (one wave alone issues one VALU per ~4 cycles, the SIMD can
// issue one per 2: is the other slot usable by a second wave for free?).
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/wave_pairing tools/microbench/wave_pairing.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(512) k_census(unsigned *out) {
    unsigned hw = __builtin_amdgcn_s_getreg(4 | (31 << 11));      // HW_REG_HW_ID
    unsigned xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));    // HW_REG_XCC_ID
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 8 + threadIdx.x / 64) * 2 + 0] = hw;
        out[(blockIdx.x * 8 + threadIdx.x / 64) * 2 + 1] = xcc;
    }
}

__device__ __forceinline__ unsigned philox_round_mix(unsigned x, unsigned k) {
    return __umulhi(0xD2511F53u, x) ^ (0xCD9E8D57u * x) ^ k;
}

// owners (waves 0..3 of the workgroup): dependent chain mixing f64 and integer ops (like the step machine)
// helpers (waves 4..7): mode 0 exit, 1 integer multiply chain (Philox-like), 2 cheap ALU chain, 3 sleep loop
template <int WG>
__global__ void __launch_bounds__(WG) k_pair(int iters, int helper_mode, int helper_iters, double *sink, unsigned *sink2) {
    const int wave = threadIdx.x / 64;
    const bool owner = (WG == 64) || wave < WG / 128;
    if (owner) {
        double a = 1.0 + threadIdx.x * 1e-9, b = 0.5;
        unsigned m = threadIdx.x;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                a = a * 1.0000001 + b;
                m = (m ^ (m >> 3)) + 0x9E3779B9u;
                b = (m & 1) ? b : a * 0.25;
            }
        }
        sink[blockIdx.x * WG + threadIdx.x] = a + b + m;
    } else {
        if (helper_mode == 0) return;
        unsigned x = threadIdx.x, k = 12345;
        if (helper_mode == 1) {
            for (int i = 0; i < helper_iters; ++i) {
#pragma unroll
                for (int j = 0; j < 10; ++j) { x = philox_round_mix(x, k); k += 0x9E3779B9u; }
            }
        } else if (helper_mode == 2) {
            for (int i = 0; i < helper_iters; ++i) {
#pragma unroll
                for (int j = 0; j < 40; ++j) { x = (x ^ (x >> 5)) + k; k += 0x9E3779B9u; }
            }
        } else {
            for (int i = 0; i < helper_iters; ++i) __builtin_amdgcn_s_sleep(32);
        }
        sink2[blockIdx.x * WG + threadIdx.x] = x;
    }
}

template <int WG>
float run(int grid, int iters, int mode, int hiters, double *sink, unsigned *sink2) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_pair<WG>, dim3(grid), dim3(WG), 0, 0, iters, mode, hiters, sink, sink2);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_pair<WG>, dim3(grid), dim3(WG), 0, 0, iters, mode, hiters, sink, sink2);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 3;
}

int main() {
    unsigned *d_out;
    hipMalloc(&d_out, 256 * 8 * 2 * 4);
    hipLaunchKernelGGL(k_census, dim3(256), dim3(512), 0, 0, d_out);
    std::vector<unsigned> h(256 * 8 * 2);
    hipMemcpy(h.data(), d_out, h.size() * 4, hipMemcpyDeviceToHost);
    for (int b = 0; b < 3; ++b) {
        printf("block %d:", b);
        for (int w = 0; w < 8; ++w) {
            unsigned hw = h[(b * 8 + w) * 2], xcc = h[(b * 8 + w) * 2 + 1];
            printf("  w%d[hw=%08x simd=%u cu=%u se=%u xcc=%u]", w, hw, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 13) & 7, xcc & 15);
        }
        printf("\n");
    }
    int same = 0, total = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 4; ++w) {
        unsigned a = h[(b * 8 + w) * 2], c = h[(b * 8 + w + 4) * 2];
        same += ((a >> 4) & 3) == ((c >> 4) & 3); ++total;
    }
    printf("waves w and w+4 on the same SIMD: %d of %d\n", same, total);
    int distinct_ok = 0;
    for (int b = 0; b < 256; ++b) {
        unsigned mask = 0;
        for (int w = 0; w < 4; ++w) mask |= 1u << ((h[(b * 8 + w) * 2] >> 4) & 3);
        distinct_ok += mask == 15;
    }
    printf("waves 0..3 on four distinct SIMDs: %d of 256 blocks\n", distinct_ok);

    double *sink; unsigned *sink2;
    hipMalloc(&sink, 1024 * 512 * 8); hipMalloc(&sink2, 1024 * 512 * 4);
    const int iters = 20000;
    printf("owner only, WG=64 x1024:                    %.3f ms\n", run<64>(1024, iters, 0, 0, sink, sink2));
    printf("owner only, WG=512 x256 (helpers exit):      %.3f ms\n", run<512>(256, iters, 0, 0, sink, sink2));
    for (int hi : {2000, 8000, 20000, 40000}) {
        printf("helper philox-like  x%-6d:                 %.3f ms\n", hi, run<512>(256, iters, 1, hi, sink, sink2));
        printf("helper cheap-alu    x%-6d:                 %.3f ms\n", hi, run<512>(256, iters, 2, hi, sink, sink2));
    }
    printf("helper sleep loop x20000:                    %.3f ms\n", run<512>(256, iters, 3, 20000, sink, sink2));
    printf("helper alone philox-like x20000 (owner 1 it): %.3f ms\n", run<512>(256, 1, 1, 20000, sink, sink2));
    printf("helper alone cheap-alu x20000 (owner 1 it):   %.3f ms\n", run<512>(256, 1, 2, 20000, sink, sink2));
    return 0;
}
