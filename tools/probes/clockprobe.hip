// Measures the shader clock a kernel actually runs at, for a given number of
// workgroups: clock = d(s_memtime) / d(s_memrealtime) * 100 MHz (guide: DVFS item 6).
// Also times a dependent v_pk/v_cndmask chain to get cycles per dependent op.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned long long* out, int iters, float a, float b) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    float x = threadIdx.x * 1e-3f, y = 1.0f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) { float t = y * a; t = t + x * b; y = x > y ? x : t; }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[blockIdx.x * 3] = t1 - t0; out[blockIdx.x * 3 + 1] = r1 - r0; }
    if (y == 12345.678f) out[blockIdx.x * 3 + 2] = 1;
}
int main() {
    for (int blocks : {1, 8, 64, 256, 1024, 4096}) {
        unsigned long long* d; hipMalloc(&d, blocks * 3 * 8);
        int iters = 20000;
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(probe, dim3(blocks), dim3(64), 0, 0, d, iters, 0.99999f, 1e-5f);
            hipDeviceSynchronize();
        }
        std::vector<unsigned long long> h(blocks * 3); hipMemcpy(h.data(), d, blocks * 3 * 8, hipMemcpyDeviceToHost);
        double cyc = h[0], real = h[1];
        printf("blocks %5d: shader clock %.0f MHz, %.1f cycles per step (3 dependent ops + cmp), %.1f ns/step\n", blocks,
               cyc / real * 100.0, cyc / (16.0 * iters), real * 10.0 / (16.0 * iters));
        hipFree(d);
    }
    return 0;
}
