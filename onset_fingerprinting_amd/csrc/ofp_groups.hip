// Onset grouping and group windows on the device:
// find_onset_groups (reference detection.py:131-189) straight from the detector's onset
// records, and the FrameExtractor gather (data.py:90-120) over the resulting rows, so that
// detect -> group -> window -> classifier needs no host round trip.
//
// Integer work only; results are bit-identical to the reference (tests/golden/g5_groups.npz).
#include <algorithm>

#include "ofp_common.h"

namespace {

using ofp::cdiv;

constexpr int GW = 64;        // wavefront
constexpr int G_THREADS = 256;
constexpr int G_WAVES = G_THREADS / GW;

__device__ __forceinline__ int64_t iabs64(int64_t v) { return v < 0 ? -v : v; }

// Row of one group, one channel per lane (strided over C): the LAST record of channel c in
// [a, e) wins (detection.py:170-172).  Returns the value for channel `c` (-1 when absent).
__device__ __forceinline__ int64_t group_entry(const ofp_onset* __restrict__ r, int a, int e, int c, bool* present) {
    int64_t v = -1;
    bool p = false;
    for (int i = a; i < e; ++i) {
        ofp_onset o = r[i];
        if (o.channel == c) {
            v = o.sample;
            p = true;
        }
    }
    *present = p;
    return v;
}

// One workgroup per clip.
//   1. wave 0 walks the anchors: a group is anchored at its first record and ends at the first
//      later record farther than max_dist from the anchor (detection.py:160-174); 64 records
//      are tested per step.
//   2. every wave takes groups round-robin, builds the row and decides whether it is kept
//      (distinct channels >= min_ch, detection.py:168-169; close_channel test, :185).
//   3. a block-wide scan of the keep flags gives each kept group its output row (order kept).
__global__ __launch_bounds__(G_THREADS) void k_group_onsets(const ofp_onset* __restrict__ rec, int64_t cap,
                                                           const int64_t* __restrict__ counts, int C,
                                                           int64_t max_dist, int min_ch, int close_ch,
                                                           int64_t* __restrict__ groups, int64_t cap_groups,
                                                           int64_t* __restrict__ n_groups,
                                                           int32_t* __restrict__ ws_anchor,
                                                           int32_t* __restrict__ ws_keep) {
    const int clip = blockIdx.x;
    const ofp_onset* r = rec + (int64_t)clip * cap;
    const int n = (int)min(counts[clip], cap);
    int32_t* anchor = ws_anchor + (int64_t)clip * (cap + 1);
    int32_t* keep = ws_keep + (int64_t)clip * cap;
    int64_t* out = groups + (int64_t)clip * cap_groups * C;
    const int lane = threadIdx.x & (GW - 1), wave = threadIdx.x / GW;
    __shared__ int s_ng;
    __shared__ int s_scan[G_WAVES];

    if (wave == 0) {
        int ng = 0, a = 0;
        while (a < n) {
            if (lane == 0) anchor[ng] = a;
            ++ng;
            const int64_t sa = r[a].sample;
            int j = a + 1;
            for (;;) {  // first record beyond the anchor's reach, 64 candidates per step
                int i = j + lane;
                bool far = i < n && iabs64(r[i].sample - sa) > max_dist;
                unsigned long long m = __ballot(far);
                if (m) {
                    j += __ffsll((long long)m) - 1;
                    break;
                }
                j += GW;
                if (j >= n) {
                    j = n;
                    break;
                }
            }
            a = j;
        }
        if (lane == 0) {
            anchor[ng] = n;
            s_ng = ng;
        }
    }
    __syncthreads();
    const int ng = s_ng;

    for (int g = wave; g < ng; g += G_WAVES) {
        const int a = anchor[g], e = anchor[g + 1];
        int distinct = 0;
        int64_t mn = INT64_MAX, vclose = -1;
        for (int c0 = 0; c0 < C; c0 += GW) {
            int c = c0 + lane;
            bool p = false;
            int64_t v = -1;
            if (c < C) {
                v = group_entry(r, a, e, c, &p);
                mn = min(mn, v);
            }
            distinct += __popcll(__ballot(p));
            if (close_ch >= c0 && close_ch < c0 + GW) vclose = __shfl(v, close_ch - c0);
        }
        for (int o = GW / 2; o > 0; o >>= 1) mn = min(mn, __shfl_xor(mn, o));
        bool k = distinct >= min_ch && (close_ch < 0 || vclose <= mn);
        if (lane == 0) keep[g] = k ? 1 : 0;
    }
    __syncthreads();

    // exclusive scan of keep[0..ng) in tiles of 256; keep[g] becomes (pos+1) for kept groups, 0 otherwise
    int base = 0;
    for (int g0 = 0; g0 < ng; g0 += G_THREADS) {
        int g = g0 + threadIdx.x;
        int k = g < ng ? keep[g] : 0;
        int incl = k;
        for (int o = 1; o < GW; o <<= 1) {
            int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == GW - 1) s_scan[wave] = incl;
        __syncthreads();
        int woff = 0, tot = 0;
        for (int w = 0; w < G_WAVES; ++w) {
            if (w < wave) woff += s_scan[w];
            tot += s_scan[w];
        }
        if (g < ng) keep[g] = k ? base + woff + incl : 0;
        base += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) n_groups[clip] = base;

    for (int g = wave; g < ng; g += G_WAVES) {
        const int pos = keep[g] - 1;
        if (pos < 0 || pos >= cap_groups) continue;
        const int a = anchor[g], e = anchor[g + 1];
        for (int c = lane; c < C; c += GW) {
            bool p;
            out[(int64_t)pos * C + c] = group_entry(r, a, e, c, &p);
        }
    }
}

// offsets[clip] = sum over earlier clips of min(n_groups, cap_groups); offsets[n_clips] = total
__global__ __launch_bounds__(G_THREADS) void k_group_offsets(const int64_t* __restrict__ n_groups, int64_t cap_groups,
                                                            int64_t n_clips, int64_t* __restrict__ offsets) {
    __shared__ int64_t s_scan[G_WAVES];
    const int lane = threadIdx.x & (GW - 1), wave = threadIdx.x / GW;
    int64_t base = 0;
    for (int64_t c0 = 0; c0 < n_clips; c0 += G_THREADS) {
        int64_t c = c0 + threadIdx.x;
        int64_t k = c < n_clips ? min(n_groups[c], cap_groups) : 0;
        int64_t incl = k;
        for (int o = 1; o < GW; o <<= 1) {
            int64_t t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == GW - 1) s_scan[wave] = incl;
        __syncthreads();
        int64_t woff = 0, tot = 0;
        for (int w = 0; w < G_WAVES; ++w) {
            if (w < wave) woff += s_scan[w];
            tot += s_scan[w];
        }
        if (c < n_clips) offsets[c] = base + woff + incl - k;
        base += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) offsets[n_clips] = base;
}

// FrameExtractor over group rows: one workgroup per (clip, group) pair, grid-strided.
// Window c of a group starts at min_c(row) - pre (use_min) or row[c] - pre (data.py:104-118).
// Samples outside the clip read as 0.
__global__ __launch_bounds__(G_THREADS) void k_group_windows(const float* __restrict__ x, int64_t n_clips,
                                                            int64_t n_samples, int C,
                                                            const int64_t* __restrict__ groups, int64_t cap_groups,
                                                            const int64_t* __restrict__ n_groups, int pre, int use_min,
                                                            int width, float* __restrict__ out, int64_t cap_total,
                                                            const int64_t* __restrict__ offsets) {
    __shared__ int64_t s_min;
    __shared__ int64_t s_red[G_WAVES];
    const int lane = threadIdx.x & (GW - 1), wave = threadIdx.x / GW;
    const int64_t pairs = n_clips * cap_groups;
    for (int64_t p = blockIdx.x; p < pairs; p += gridDim.x) {
        const int64_t clip = p / cap_groups, g = p % cap_groups;
        if (g >= min(n_groups[clip], cap_groups)) continue;  // uniform over the workgroup
        const int64_t row_out = offsets[clip] + g;
        if (row_out >= cap_total) continue;
        const int64_t* row = groups + (clip * cap_groups + g) * C;
        if (use_min) {
            int64_t mn = INT64_MAX;
            for (int c = threadIdx.x; c < C; c += G_THREADS) mn = min(mn, row[c]);
            for (int o = GW / 2; o > 0; o >>= 1) mn = min(mn, __shfl_xor(mn, o));
            if (lane == 0) s_red[wave] = mn;
            __syncthreads();
            if (threadIdx.x == 0) {
                int64_t m = s_red[0];
                for (int w = 1; w < G_WAVES; ++w) m = min(m, s_red[w]);
                s_min = m;
            }
            __syncthreads();
        }
        const float* xc = x + clip * n_samples * C;
        float* o = out + row_out * C * width;
        for (int i = threadIdx.x; i < C * width; i += G_THREADS) {
            int c = i / width, w = i - c * width;
            int64_t t = (use_min ? s_min : row[c]) - pre + w;
            o[i] = (t >= 0 && t < n_samples) ? xc[t * C + c] : 0.0f;
        }
        __syncthreads();  // s_min is rewritten by the next pair
    }
}


// ---- collation block: the records of all clips of a rank, compacted in clip order behind a header.
// One workgroup per clip: its first output row = 1 + the (clamped) counts of the clips before it, summed by the
// workgroup itself (at most 65535 clips: a few hundred loads per lane); workgroup 0 also writes the header.
__global__ __launch_bounds__(256) void k_pack_records(const ofp_onset* __restrict__ rec, const int64_t* __restrict__ counts,
                                                      int64_t n_clips, int64_t cap, int64_t cap_total, int32_t clip_offset,
                                                      ofp_onset* __restrict__ block) {
    __shared__ long long s_part[G_WAVES];
    __shared__ int s_over[G_WAVES];
    const int64_t clip = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t upto = clip == 0 ? n_clips : clip;  // workgroup 0 sums everything (the total)
    long long acc = 0;
    int over = 0;
    for (int64_t i = threadIdx.x; i < upto; i += G_THREADS) {
        const int64_t c = counts[i];
        acc += (long long)min<int64_t>(max<int64_t>(c, 0), cap);
        over |= c > cap;
    }
    for (int o = 32; o > 0; o >>= 1) {
        acc += __shfl_xor(acc, o);
        over |= __shfl_xor(over, o);
    }
    if (lane == 0) {
        s_part[wave] = acc;
        s_over[wave] = over;
    }
    __syncthreads();
    acc = 0;
    over = 0;
    for (int w = 0; w < G_WAVES; ++w) {
        acc += s_part[w];
        over |= s_over[w];
    }
    int64_t first = 1 + acc;
    if (clip == 0) {
        first = 1;
        if (threadIdx.x == 0) {
            ofp_onset h;
            h.clip = 0;
            h.channel = over ? 1 : 0;
            h.sample = acc;
            block[0] = h;
        }
    }
    const int64_t n = min<int64_t>(max<int64_t>(counts[clip], 0), cap);
    const ofp_onset* src = rec + clip * cap;
    for (int64_t k = threadIdx.x; k < n; k += G_THREADS) {
        const int64_t row = first + k;
        if (row > cap_total) break;  // does not fit: dropped, the header tells
        ofp_onset o = src[k];
        o.clip += clip_offset;
        block[row] = o;
    }
}

}  // namespace

extern "C" {

int64_t ofp_group_workspace_bytes(int64_t n_clips, int64_t cap_per_clip) {
    if (n_clips < 0 || cap_per_clip < 0) return -1;
    return ofp::align_up(n_clips * (cap_per_clip + 1) * 4, 256) + ofp::align_up(n_clips * cap_per_clip * 4, 256);
}

int ofp_group_onsets(const ofp_onset* d_records, int64_t cap_per_clip, const int64_t* d_counts, int64_t n_clips,
                     int32_t n_channels, int64_t max_distance, int32_t min_channels, int32_t close_channel,
                     int64_t* d_groups, int64_t cap_groups, int64_t* d_n_groups, void* d_ws, int64_t ws_bytes,
                     void* stream) {
    if (n_clips == 0) return OFP_OK;
    OFP_REQUIRE(d_records && d_counts && d_groups && d_n_groups && d_ws, "ofp_group_onsets: NULL argument");
    OFP_REQUIRE(n_clips > 0 && n_clips < (1 << 30) && cap_per_clip >= 1 && cap_per_clip < (1ll << 30) &&
                    cap_groups >= 1 && n_channels >= 1,
                "ofp_group_onsets: bad size (n_clips=%lld cap_per_clip=%lld cap_groups=%lld n_channels=%d)",
                (long long)n_clips, (long long)cap_per_clip, (long long)cap_groups, n_channels);
    OFP_REQUIRE(close_channel < n_channels, "ofp_group_onsets: close_channel %d outside [0, %d)", close_channel,
                n_channels);
    OFP_REQUIRE(ws_bytes >= ofp_group_workspace_bytes(n_clips, cap_per_clip),
                "ofp_group_onsets: work space too small (%lld < %lld)", (long long)ws_bytes,
                (long long)ofp_group_workspace_bytes(n_clips, cap_per_clip));
    int32_t* anchor = (int32_t*)d_ws;
    int32_t* keep = (int32_t*)((char*)d_ws + ofp::align_up(n_clips * (cap_per_clip + 1) * 4, 256));
    hipLaunchKernelGGL(k_group_onsets, dim3((unsigned)n_clips), dim3(G_THREADS), 0, (hipStream_t)stream, d_records,
                       cap_per_clip, d_counts, n_channels, max_distance, min_channels,
                       close_channel < 0 ? -1 : close_channel, d_groups, cap_groups, d_n_groups, anchor, keep);
    OFP_LAUNCH_CHECK("k_group_onsets");
    return OFP_OK;
}

int ofp_group_windows(const float* d_x, int64_t n_clips, int64_t n_samples, int32_t n_channels,
                      const int64_t* d_groups, int64_t cap_groups, const int64_t* d_n_groups, int32_t pre_samples,
                      int32_t use_min_onset, int32_t width, float* d_out, int64_t cap_total, int64_t* d_offsets,
                      void* stream_) {
    if (n_clips == 0) return OFP_OK;
    OFP_REQUIRE(d_x && d_groups && d_n_groups && d_out && d_offsets, "ofp_group_windows: NULL argument");
    OFP_REQUIRE(n_clips > 0 && n_samples >= 0 && n_channels >= 1 && cap_groups >= 1 && width >= 1 && cap_total >= 0 &&
                    (int64_t)n_channels * width < (1ll << 31),
                "ofp_group_windows: bad size");
    hipStream_t stream = (hipStream_t)stream_;
    hipLaunchKernelGGL(k_group_offsets, dim3(1), dim3(G_THREADS), 0, stream, d_n_groups, cap_groups, n_clips,
                       d_offsets);
    OFP_LAUNCH_CHECK("k_group_offsets");
    unsigned grid = (unsigned)std::min<int64_t>(n_clips * cap_groups, 256 * 16);
    hipLaunchKernelGGL(k_group_windows, dim3(grid), dim3(G_THREADS), 0, stream, d_x, n_clips, n_samples, n_channels,
                       d_groups, cap_groups, d_n_groups, pre_samples, use_min_onset, width, d_out, cap_total,
                       (const int64_t*)d_offsets);
    OFP_LAUNCH_CHECK("k_group_windows");
    return OFP_OK;
}

int ofp_pack_records(const ofp_onset* d_records, const int64_t* d_counts, int64_t n_clips, int64_t cap_per_clip,
                     int64_t cap_total, int32_t clip_offset, ofp_onset* d_block, void* stream) {
    OFP_REQUIRE(d_counts && d_block && n_clips >= 1 && n_clips <= 65535 && cap_per_clip >= 0 && cap_total >= 0,
                "ofp_pack_records: bad argument");
    OFP_REQUIRE(d_records || cap_per_clip == 0, "ofp_pack_records: d_records is NULL");
    hipLaunchKernelGGL(k_pack_records, dim3((unsigned)n_clips), dim3(G_THREADS), 0, (hipStream_t)stream, d_records, d_counts,
                       n_clips, cap_per_clip, cap_total, clip_offset, d_block);
    OFP_LAUNCH_CHECK("k_pack_records");
    return OFP_OK;
}

}  // extern "C"
