// Times the library's own k_mm (pass 0) on a synthetic planar stream, to bisect its per-step cost.
#include "../../onset_fingerprinting_amd/csrc/ofp_detect.hip"
#include <cstdio>
// the same walk + MaxStep, alone in a lean kernel
template <bool DEEP>
__global__ __launch_bounds__(64) void k_maxonly(const float* rel, int64_t U, int64_t L, int64_t W, int64_t n_chunks, int64_t n_threads, float* out) {
    const int64_t id = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (id >= n_threads) return;
    const int64_t k = id % n_chunks, chain = id / n_chunks;
    const int64_t start = k * L;
    int64_t ws = start - W; if (ws < 0) ws = 0;
    MaxStep mo{0.0f, 0.99999f, 1e-5f};
    int norem = -1;
    walk<16, 0, false, DEEP>(rel + chain * U + ws, nullptr, start - ws, norem, mo);
    out[id] = mo.mx;
}
int main() {
    Geom g; g.C = 8; g.B = 256; g.N = 2880000; g.Nm = 2880000; g.n_w = 24000; g.n_wb = 23808; g.U = g.n_wb + g.Nm; g.V = g.n_w + g.Nm;
    const int64_t chains = 8;
    float* rel; hipMalloc(&rel, chains * g.U * 4 + 256);
    std::vector<float> h(chains * g.U);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 1.0f + (float)((i * 2654435761u) % 1000) * 0.002f + ((i % 24000) < 40 ? 20.0f : 0.0f);
    hipMemcpy(rel, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    int64_t nb = g.Nm / g.B;
    float *tmn, *tmx; hipMalloc(&tmn, nb * 8 * 4); hipMalloc(&tmx, nb * 8 * 4);
    for (int64_t L : {8192, 65536}) for (int64_t W : {16384, 65536}) {
        MmArgs a; a.g = g; a.rel = rel; a.thr_mn = tmn; a.thr_mx = tmx;
        a.alpha_min = 1e-4f; a.alpha_max = 1e-5f; a.ialpha_min = ofp_ialpha(1e-4f); a.ialpha_max = ofp_ialpha(1e-5f);
        a.minmin = 2.0f; a.min0 = 0; a.max0 = 10; a.nb = nb; a.L = L; a.W = W; a.n_chunks = (g.U + L - 1) / L;
        int64_t nt = chains * a.n_chunks;
        uint32_t* st; hipMalloc(&st, nt * 2 * 4 * 3); int* chg; hipMalloc(&chg, 64);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_mm, dim3((unsigned)((nt + 63) / 64)), dim3(64), 0, 0, a, 0, nt, (const uint32_t*)st, st + nt * 2, st + nt * 4, chg);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        printf("k_mm pass0 L=%lld W=%lld lanes=%lld: %.3f ms\n", (long long)L, (long long)W, (long long)nt, ms);
        float* o2; hipMalloc(&o2, nt * 4);
        for (int deep = 0; deep < 2; ++deep) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (deep) hipLaunchKernelGGL(k_maxonly<true>, dim3((unsigned)((nt + 63) / 64)), dim3(64), 0, 0, rel, g.U, L, W, a.n_chunks, nt, o2);
                else hipLaunchKernelGGL(k_maxonly<false>, dim3((unsigned)((nt + 63) / 64)), dim3(64), 0, 0, rel, g.U, L, W, a.n_chunks, nt, o2);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            printf("   lean max-only deep=%d: %.3f ms = %.1f ns/step\n", deep, ms, ms * 1e6 / W);
        }
        hipFree(o2);
        hipFree(st); hipFree(chg);
    }
    return 0;
}
