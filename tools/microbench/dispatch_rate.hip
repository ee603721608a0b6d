// Diagnostic (GPU box): what a kernel LAUNCH over 65 536 one-lane-per-table items costs as a function of the workgroup shape and of the
// resources a workgroup claims -- the fixed part of every table-kernel launch (k_step / k_pick / k_reset / the env kernels run 1 024
// one-wave workgroups).  Per variant: average time of back-to-back launches on one stream (HIP events), i.e. launch + dispatch + drain.
//   tiny     : 1 byte in, 4 bytes out per lane, nothing else           (k_pick's shape)
//   fat      : + LDS_PER_WAVE bytes of LDS per wave and ~168 VGPRs     (k_step's footprint), a short dependent ALU chain
// block 64 x 1024 workgroups  vs  block 256 x 256  vs  block 1024 x 64.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/dispatch_rate tools/microbench/dispatch_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_tiny(const unsigned char *in, int *out, int n) {
    int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t < n) out[t] = in[t] * 3 + 1;
}

template <int BLOCK, int LDS_PER_WAVE, int CHAIN>
__global__ void __launch_bounds__(BLOCK, 1) k_fat(const double *in, double *out, int n) {
    __shared__ unsigned lds[(BLOCK / 64) * (LDS_PER_WAVE / 4)];
    int t = blockIdx.x * BLOCK + threadIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double r[40];                                   // ~80 VGPRs of live state + the chain's temporaries
    if (t < n) {
#pragma unroll
        for (int i = 0; i < 40; ++i) r[i] = in[(size_t)i * n + t];
    } else {
#pragma unroll
        for (int i = 0; i < 40; ++i) r[i] = 0.0;
    }
    lds[wave * (LDS_PER_WAVE / 4) + lane] = (unsigned)r[0];
    __builtin_amdgcn_wave_barrier();
    unsigned x = lds[wave * (LDS_PER_WAVE / 4) + (lane ^ 1)];
#pragma unroll 1
    for (int c = 0; c < CHAIN; ++c) {               // dependent ALU chain through all registers
#pragma unroll
        for (int i = 0; i < 40; ++i) r[i] = r[i] * 1.0000001 + (double)(x & 1);
        x = x * 1664525u + 1013904223u;
    }
    if (t < n) {
#pragma unroll
        for (int i = 0; i < 40; ++i) out[(size_t)i * n + t] = r[i];
    }
}

template <typename F>
static double time_launches(F launch, int reps) {
    hipEvent_t a, b;
    CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    for (int i = 0; i < 50; ++i) launch();
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) launch();
    CHK(hipEventRecord(b, 0));
    CHK(hipEventSynchronize(b));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3 / reps;
}

int main() {
    const int n = 65536, reps = 3000;
    unsigned char *in8; int *out32; double *din, *dout;
    CHK(hipMalloc(&in8, n)); CHK(hipMalloc(&out32, n * 4));
    CHK(hipMalloc(&din, (size_t)40 * n * 8)); CHK(hipMalloc(&dout, (size_t)40 * n * 8));
    CHK(hipMemset(in8, 1, n)); CHK(hipMemset(din, 0, (size_t)40 * n * 8));
    printf("n = %d items, %d back-to-back launches per variant, us per launch\n", n, reps);
    printf("tiny  block   64 x %4d workgroups: %7.2f us\n", n / 64, time_launches([&] { hipLaunchKernelGGL(k_tiny<64>, dim3(n / 64), dim3(64), 0, 0, in8, out32, n); }, reps));
    printf("tiny  block  256 x %4d workgroups: %7.2f us\n", n / 256, time_launches([&] { hipLaunchKernelGGL(k_tiny<256>, dim3(n / 256), dim3(256), 0, 0, in8, out32, n); }, reps));
    printf("tiny  block 1024 x %4d workgroups: %7.2f us\n", n / 1024, time_launches([&] { hipLaunchKernelGGL(k_tiny<1024>, dim3(n / 1024), dim3(1024), 0, 0, in8, out32, n); }, reps));
#define FAT(BLOCK, LDS, CHAIN) printf("fat   block %4d x %4d workgroups, %5d B LDS per wave, chain %3d: %7.2f us\n", BLOCK, n / BLOCK, LDS, CHAIN, \
        time_launches([&] { hipLaunchKernelGGL((k_fat<BLOCK, LDS, CHAIN>), dim3(n / BLOCK), dim3(BLOCK), 0, 0, din, dout, n); }, reps));
    FAT(64, 10240, 0) FAT(256, 10240, 0) FAT(1024, 10240, 0)
    FAT(64, 1024, 0) FAT(256, 1024, 0)
    FAT(64, 10240, 20) FAT(256, 10240, 20) FAT(1024, 10240, 20)
    FAT(64, 10240, 100) FAT(256, 10240, 100)
    return 0;
}
