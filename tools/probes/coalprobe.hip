// Does a wave whose 64 lanes each stream their OWN contiguous series read faster when the wave loads the next
// [64 rows x 128 B] tile cooperatively (8 lanes x 16 B per row: full 128-byte lines per request) and hands the rows to
// their lanes through LDS, than when every lane issues private 16-byte loads (64 different lines per instruction)?
// Chunk-pass geometry: lane i walks `steps` floats from i * steps; light (max tracker) and heavy (IIR-like) steps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../include/ofp_math.h"

template <bool HEAVY>
__device__ __forceinline__ void step4(const float4& v, float& s0, float& s1) {
    const float ia = 0.99999f, al = 1e-5f;
    if (!HEAVY) {
        s0 = ofp_max_step(v.x, s0, ia, al); s0 = ofp_max_step(v.y, s0, ia, al);
        s0 = ofp_max_step(v.z, s0, ia, al); s0 = ofp_max_step(v.w, s0, ia, al);
    } else {  // ten dependent-ish operations per sample, two state words
        const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float y = s0 + 0.3f * x[i];
            const float a = (s1 + 0.2f * x[i]) - 0.5f * y;
            const float b = (0.1f * x[i]) - 0.25f * y;
            s0 = a * 0.999f + b * 0.001f;
            s1 = b * 0.9f + a * 0.01f;
        }
    }
}

template <bool HEAVY>
__global__ __launch_bounds__(64) void k_private(const float* __restrict__ buf, int64_t steps, float* out, int64_t n_lanes) {
    const int64_t id = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (id >= n_lanes) return;
    const float4* q = reinterpret_cast<const float4*>(buf + id * steps);
    float s0 = 0.0f, s1 = 0.0f;
    float4 A[8], B[8];
    int64_t nb = steps / 32;
#pragma unroll
    for (int i = 0; i < 8; ++i) A[i] = q[i];
    q += 8;
    for (int64_t b = 0; b + 2 < nb; b += 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) B[i] = q[i];
        q += 8;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) step4<HEAVY>(A[i], s0, s1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) A[i] = q[i];
        q += 8;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) step4<HEAVY>(B[i], s0, s1);
        __builtin_amdgcn_sched_barrier(0);
    }
    out[id] = s0 + s1;
}

template <bool HEAVY>
__global__ __launch_bounds__(64) void k_coop(const float* __restrict__ buf, int64_t steps, float* out, int64_t n_lanes) {
    __shared__ float4 tile[64][9];  // one batch: 8 float4 per row + one pad
    const int lane = threadIdx.x;
    const int64_t id0 = (int64_t)blockIdx.x * 64;
    const int64_t id = id0 + lane;
    const int rsub = lane >> 3, chunk = lane & 7;
    float s0 = 0.0f, s1 = 0.0f;
    float4 A[8], S[8];
    const int64_t nb = steps / 32;
    auto coop_load = [&](int64_t b) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int64_t row = id0 + 8 * k + rsub;
            const float4* p = reinterpret_cast<const float4*>(buf + (row < n_lanes ? row : n_lanes - 1) * steps) + b * 8 + chunk;
            S[k] = *p;
        }
    };
    auto hand_over = [&]() {
#pragma unroll
        for (int k = 0; k < 8; ++k) tile[8 * k + rsub][chunk] = S[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int i = 0; i < 8; ++i) A[i] = tile[lane][i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    coop_load(0);
    hand_over();
    for (int64_t b = 0; b + 1 < nb; ++b) {
        coop_load(b + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) step4<HEAVY>(A[i], s0, s1);
        __builtin_amdgcn_sched_barrier(0);
        hand_over();
    }
    if (id < n_lanes) out[id] = s0 + s1;
}

int main() {
    const int64_t total = 640ll * 1024 * 1024;  // floats: 2.5 GB, beyond every cache
    float* buf; hipMalloc(&buf, total * 4 + 4096);
    hipMemset(buf, 0, total * 4);
    float* out; hipMalloc(&out, 64 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto kern, int64_t lanes, int64_t steps) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, 0, buf, steps, out, lanes);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        printf("%-16s lanes %7lld (%4.1f waves/SIMD) steps %6lld: %8.3f ms  %7.1f GB/s  %6.1f ns/step\n", name, (long long)lanes,
               lanes / 64.0 / 1024.0, (long long)steps, best, lanes * steps * 4.0 / best / 1e6, best * 1e6 / steps);
    };
    for (int64_t steps : {4096, 16384}) {
        for (int64_t lanes : {65536, 131072, 262144, 524288}) {
            if (lanes * steps > total) continue;
            run("private light", k_private<false>, lanes, steps);
            run("coop    light", k_coop<false>, lanes, steps);
            run("private heavy", k_private<true>, lanes, steps);
            run("coop    heavy", k_coop<true>, lanes, steps);
        }
    }
    return 0;
}
