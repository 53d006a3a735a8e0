// What do a walker's OUTPUT stores cost?  64 lanes each stream their own contiguous series (16-byte loads a batch
// ahead, an IIR-like dependent step per sample, as the chunk passes do) and write a series of the same length:
//   private: every lane stores its own 16 bytes (a wave's store instruction touches 64 different lines, 16 B each;
//            a lane completes a 128-byte line with 8 consecutive instructions) -- what walk<.., OUT = 4> does;
//   lines  : the wave hands the batch over through LDS and every store instruction writes 8 COMPLETE 128-byte lines
//            (8 lanes x 16 B per line).
//   none   : no output (the cost of the walk itself).
// hipcc --offload-arch=gfx950 -O3 -o storeprobe storeprobe.hip;  ./storeprobe [lanes] [steps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ float step1(float x, float& s0, float& s1) {
    const float y = s0 + 0.3f * x;
    const float a = (s1 + 0.2f * x) - 0.5f * y;
    const float b = (0.1f * x) - 0.25f * y;
    s0 = a * 0.999f + b * 0.001f;
    s1 = b * 0.9f + a * 0.01f;
    return y;
}

constexpr int PITCH = 36;  // floats per lane row in LDS (32 + 4)

template <int MODE>  // 0 none, 1 private, 2 lines
__global__ __launch_bounds__(64) void k_walk(const float* __restrict__ in, float* __restrict__ outp, int64_t steps, float* sink,
                                             int64_t n_lanes) {
    __shared__ float tile[64 * PITCH];
    const int lane = threadIdx.x;
    const int64_t id = (int64_t)blockIdx.x * 64 + lane;  // (n_lanes is a multiple of 64)
    const float4* q = reinterpret_cast<const float4*>(in + id * steps);
    float* op = outp + id * steps;
    float s0 = 0.0f, s1 = 0.0f;
    float4 A[8], B[8];
    const int64_t nb = steps / 32;
    auto load = [&](float4 (&v)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = q[i];
        q += 8;
        __builtin_amdgcn_sched_barrier(0);
    };
    auto run = [&](const float4 (&v)[8]) {
        float4 o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            o[i].x = step1(v[i].x, s0, s1); o[i].y = step1(v[i].y, s0, s1);
            o[i].z = step1(v[i].z, s0, s1); o[i].w = step1(v[i].w, s0, s1);
        }
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) reinterpret_cast<float4*>(op)[i] = o[i];
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(&tile[lane * PITCH + 4 * i]) = o[i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            // store j: lines of lanes 8 j .. 8 j + 7; this lane writes piece lane & 7 of lane 8 j + (lane >> 3)
            const int64_t my = reinterpret_cast<int64_t>(op);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int owner = 8 * j + (lane >> 3);
                const int lo = __shfl((int)my, owner), hi = __shfl((int)(my >> 32), owner);
                float* dst = reinterpret_cast<float*>(((int64_t)hi << 32) | (uint32_t)lo) + 4 * (lane & 7);
                *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(&tile[owner * PITCH + 4 * (lane & 7)]);
            }
            __builtin_amdgcn_wave_barrier();
        }
        op += 32;
        __builtin_amdgcn_sched_barrier(0);
    };
    load(A);
    int64_t b = 0;
    for (; b + 2 < nb; b += 2) {
        load(B); run(A);
        load(A); run(B);
    }
    if (sink && s0 + s1 == 123.456f) sink[id] = s0;
}

int main(int argc, char** argv) {
    const int64_t n_lanes = argc > 1 ? atoll(argv[1]) : 65536;
    const int64_t steps = argc > 2 ? atoll(argv[2]) : 16384;
    const int64_t n = n_lanes * steps;
    float *in, *out, *sink;
    hipMalloc(&in, n * 4); hipMalloc(&out, n * 4); hipMalloc(&sink, n_lanes * 4);
    hipMemset(in, 0, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char* name, auto kern, double bytes) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3((unsigned)(n_lanes / 64)), dim3(64), 0, 0, (const float*)in, out, steps, sink, n_lanes);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("%-8s lanes %lld steps %lld: %.3f ms  %.1f ns/step/lane  %.2f TB/s\n", name, (long long)n_lanes, (long long)steps, ms,
                                 ms * 1e6 / steps, bytes / ms / 1e9);
        }
    };
    time("none", k_walk<0>, n * 4.0);
    time("private", k_walk<1>, n * 8.0);
    time("lines", k_walk<2>, n * 8.0);
    return 0;
}
