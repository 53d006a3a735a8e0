// Times the library's own tracker kernels (k_mm_max / k_mm_warm / k_mm_chunk pass 0) on a
// synthetic planar stream.  Historical note: with all three loops in ONE kernel the max-only
// loop cost 35-50 ns/step; as lean kernels it costs 11 ns/step (see README.md).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Wno-unused \
//        -o mmprobe mmprobe.hip ../../onset_fingerprinting_amd/csrc/ofp_core.hip
#include "../../onset_fingerprinting_amd/csrc/ofp_detect.hip"
#include <cstdio>
int main() {
    Geom g;
    g.C = 8; g.B = 256; g.N = 2880000; g.Nm = 2880000; g.n_w = 24000; g.n_wb = 23808;
    g.U = g.n_wb + g.Nm; g.V = g.n_w + g.Nm;
    const int64_t chains = 8;
    float* rel;
    if (hipMalloc(&rel, chains * g.U * 4 + 256) != hipSuccess) return 1;
    std::vector<float> h(chains * g.U);
    for (size_t i = 0; i < h.size(); ++i)
        h[i] = 1.0f + (float)((i * 2654435761u) % 1000) * 0.002f + ((i % 24000) < 40 ? 20.0f : 0.0f);
    hipMemcpy(rel, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int64_t nb = g.Nm / g.B;
    float *tmn, *tmx;
    hipMalloc(&tmn, nb * 8 * 4);
    hipMalloc(&tmx, nb * 8 * 4);
    hipEvent_t e[4];
    for (auto& x : e) hipEventCreate(&x);
    for (int64_t L : {8192, 65536})
        for (int64_t W : {16384, 65536}) {
            MmArgs a;
            a.g = g; a.rel = rel; a.thr_mn = tmn; a.thr_mx = tmx;
            a.alpha_min = 1e-4f; a.alpha_max = 1e-5f;
            a.ialpha_min = ofp_ialpha(1e-4f); a.ialpha_max = ofp_ialpha(1e-5f);
            a.minmin = 2.0f; a.min0 = 0; a.max0 = 10; a.nb = nb; a.L = L; a.W = W;
            a.n_chunks = (g.U + L - 1) / L;
            const int64_t nt = chains * a.n_chunks;  // lanes; every lane stays inside its chain's U floats
            uint32_t* st;
            hipMalloc(&st, nt * 2 * 4 * 3);
            int* chg;
            hipMalloc(&chg, 64);
            const unsigned grid = (unsigned)((nt + 63) / 64);
            float ms[3] = {0, 0, 0};
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e[0]);
                hipLaunchKernelGGL(k_mm_max, dim3(grid), dim3(64), 0, 0, a, nt, st);
                hipEventRecord(e[1]);
                hipLaunchKernelGGL(k_mm_warm, dim3(grid), dim3(64), 0, 0, a, nt, st);
                hipEventRecord(e[2]);
                hipLaunchKernelGGL(k_mm_chunk, dim3(grid), dim3(64), 0, 0, a, 0, nt, (const uint32_t*)(st + nt * 4),
                                   st + nt * 2, st, chg);
                hipEventRecord(e[3]);
                hipEventSynchronize(e[3]);
                for (int k = 0; k < 3; ++k) hipEventElapsedTime(&ms[k], e[k], e[k + 1]);
            }
            printf("L=%lld W=%lld lanes=%lld: k_mm_max %.3f ms (%.1f ns/step)  k_mm_warm %.3f ms  k_mm_chunk %.3f ms (%.1f ns/step)\n",
                   (long long)L, (long long)W, (long long)nt, ms[0], ms[0] * 1e6 / (W - 4096), ms[1], ms[2],
                   ms[2] * 1e6 / L);
            hipFree(st);
            hipFree(chg);
        }
    return 0;
}
