// Isolates the inner loop of the detector kernels: each lane walks `steps` consecutive
// floats of its own region applying the max-tracker step; reports ns per step for
// different lane counts and with/without memory.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../include/ofp_math.h"

template <int PB4, bool MEM>
__global__ __launch_bounds__(64) void probe(const float* __restrict__ buf, int64_t stride, int64_t steps, float* out, int64_t n_lanes) {
    int64_t id = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (id >= n_lanes) return;
    const float4* q = reinterpret_cast<const float4*>(buf + id * stride);
    float mx = 0.0f;
    const float ia = 0.99999f, al = 1e-5f;
    float4 A[PB4], B[PB4];
    int64_t nb = steps / 4;
    if (MEM) {
#pragma unroll
        for (int i = 0; i < PB4; ++i) A[i] = q[i];
        q += PB4; nb -= PB4;
        while (nb >= 2 * PB4) {
#pragma unroll
            for (int i = 0; i < PB4; ++i) B[i] = q[i];
            q += PB4;
#pragma unroll
            for (int i = 0; i < PB4; ++i) { mx = ofp_max_step(A[i].x, mx, ia, al); mx = ofp_max_step(A[i].y, mx, ia, al); mx = ofp_max_step(A[i].z, mx, ia, al); mx = ofp_max_step(A[i].w, mx, ia, al); }
#pragma unroll
            for (int i = 0; i < PB4; ++i) A[i] = q[i];
            q += PB4;
#pragma unroll
            for (int i = 0; i < PB4; ++i) { mx = ofp_max_step(B[i].x, mx, ia, al); mx = ofp_max_step(B[i].y, mx, ia, al); mx = ofp_max_step(B[i].z, mx, ia, al); mx = ofp_max_step(B[i].w, mx, ia, al); }
            nb -= 2 * PB4;
        }
    } else {
        float x = threadIdx.x * 0.01f;
        for (int64_t i = 0; i < steps; i += 4) { mx = ofp_max_step(x, mx, ia, al); mx = ofp_max_step(x + 1, mx, ia, al); mx = ofp_max_step(x + 2, mx, ia, al); mx = ofp_max_step(x + 3, mx, ia, al); }
    }
    out[id] = mx;
}

int main() {
    const int64_t total = 24 * 1024 * 1024;  // floats (96 MB), like the C2 relative envelope
    float* buf; hipMalloc(&buf, total * 4 + 4096); hipMemset(buf, 0, total * 4);
    std::vector<float> h(total); for (int64_t i = 0; i < total; ++i) h[i] = (float)((i * 2654435761u) % 1000) * 0.001f;
    hipMemcpy(buf, h.data(), total * 4, hipMemcpyHostToDevice);
    float* out; hipMalloc(&out, 1 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto kern, int64_t lanes, int64_t stride, int64_t steps) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3((lanes + 63) / 64), dim3(64), 0, 0, buf, stride, steps, out, lanes);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s lanes %6lld stride %6lld steps %6lld: %8.3f ms  %6.1f ns/step\n", name, (long long)lanes, (long long)stride, (long long)steps, ms, ms * 1e6 / steps);
    };
    for (int64_t lanes : {64, 512, 2816}) {
        int64_t stride = 8192, steps = 8192 * 6;  // overlapping regions: lanes read 6 chunks ahead (like the warm-up)
        if ((lanes - 1) * 8256 + steps + 64 > total) { printf("skip lanes %lld\n", (long long)lanes); continue; }  // stay inside the buffer
        run("nomem", probe<16, false>, lanes, stride, steps);
        run("mem PB4=16", probe<16, true>, lanes, stride, steps);
        run("mem PB4=8", probe<8, true>, lanes, stride, steps);
        run("mem PB4=16 stride 8256", probe<16, true>, lanes, 8256, steps);
    }
    return 0;
}
