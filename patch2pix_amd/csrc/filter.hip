// filter_coarse on the device -- reference networks/utils.py:38-72 without the ptmax sampling (that one draws from
// the host's numpy RNG and stays on the host): per batch item, the lexicographically sorted distinct rows of the
// [n,4] int64 match list with the score of their first occurrence; `mutual` keeps rows that occur more than once;
// if nothing is kept the list stays as it was; then rows with score > ncn_thres, again "or everything".
// One work-group per item: packed 64-bit keys (4 x 16 bits, pixel coordinates are < 2^15) + original index,
// bitonic sort, run detection, two order-preserving compactions.  Up to 8192 rows (padded to a power of two) the whole
// item lives in LDS.  Longer lists (a 960x1280 pair has 9600 rows, a 1600-pixel image 15000) are sorted in a
// caller-provided workspace: chunks of 8192 rows are sorted in LDS, the network steps whose stride spans chunks run on
// global memory (the work-group's own stores, ordered by its barriers), the rest of every merge again chunk-wise in LDS.
#include "p2p_common.h"

namespace p2p {

constexpr int FT = 1024;                      // threads per work-group
constexpr int FILTER_LDS_ROWS = 8192;         // rows held in LDS at a time
constexpr int FILTER_MAX_ROWS = 1 << 20;      // workspace path: 16 bytes of workspace per padded row

struct FilterArgs {
    const long long *matches;                 // [B][n][4]
    const float *scores;                      // [B][n]
    int n, npad;                              // rows per item, next power of two
    float thres;
    int mutual;
    long long *out_matches;                   // [B][n][4]
    float *out_scores;                        // [B][n]
    int *out_counts;                          // [B]; -1 = a coordinate did not fit the packed key (caller falls back)
    unsigned char *ws;                        // BIG only: npad * 16 bytes per item (keys, indices, kept-row list)
};

// exclusive prefix sum of one int per thread over the work-group; returns the total through *total
__device__ __forceinline__ int block_exclusive_scan(int v, int *wave_sums, int *total) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wave_sums[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        int s = (lane < FT / 64) ? wave_sums[lane] : 0;
#pragma unroll
        for (int d = 1; d < FT / 64; d <<= 1) {
            const int o = __shfl_up(s, d);
            if (lane >= d) s += o;
        }
        if (lane < FT / 64) wave_sums[lane] = s;      // inclusive over waves
    }
    __syncthreads();
    const int base = wave ? wave_sums[wave - 1] : 0;
    *total = wave_sums[FT / 64 - 1];
    __syncthreads();                                  // wave_sums may be reused by the caller
    return base + incl - v;
}

// one compare-exchange of the bitonic network on (key, original index): equal rows stay in input order, so a run of
// equal keys starts with its first occurrence
__device__ __forceinline__ void cmp_xchg(unsigned long long *key, int *idx, int lo, int hi, bool up) {
    const unsigned long long kl = key[lo], kh = key[hi];
    const int il = idx[lo], ih = idx[hi];
    const bool greater = kl > kh || (kl == kh && il > ih);
    if (greater == up) { key[lo] = kh; key[hi] = kl; idx[lo] = ih; idx[hi] = il; }
}

// the steps stride_first, stride_first / 2, ..., 1 of the merge of width `size` on `count` rows that start at global
// position `base` of the item (the direction of a pair depends on its global position)
__device__ __forceinline__ void bitonic_steps(unsigned long long *key, int *idx, int count, int base, int size, int stride_first) {
    for (int stride = stride_first; stride > 0; stride >>= 1) {
        for (int t = threadIdx.x; t < count / 2; t += FT) {
            const int lo = 2 * t - (t & (stride - 1));
            cmp_xchg(key, idx, lo, lo + stride, ((base + lo) & size) == 0);
        }
        __syncthreads();
    }
}

template <bool BIG>
__global__ __launch_bounds__(FT) void filter_coarse_kernel(FilterArgs a) {
    P2P_DYN_SHARED(unsigned char, fsm);
    constexpr int CH = FILTER_LDS_ROWS;
    const int lrows = BIG ? CH : a.npad;                                // rows in LDS
    unsigned long long *lkey = (unsigned long long *)fsm;              // [lrows]
    int *lidx = (int *)(fsm + (size_t)lrows * 8);                      // [lrows]  original row of a sorted position
    __shared__ int wave_sums[FT / 64];
    __shared__ int flag_bad;
    const int tid = threadIdx.x, item = blockIdx.x;
    unsigned long long *key = lkey;
    int *idx = lidx, *pos = lidx + a.npad;                             // pos [npad]: list of kept source rows
    if (BIG) {
        unsigned char *w = a.ws + (size_t)item * a.npad * 16;
        key = (unsigned long long *)w;
        idx = (int *)(w + (size_t)a.npad * 8);
        pos = idx + a.npad;
    }
    const long long *rows = a.matches + (size_t)item * a.n * 4;
    const float *sc = a.scores + (size_t)item * a.n;
    long long *orow = a.out_matches + (size_t)item * a.n * 4;
    float *osc = a.out_scores + (size_t)item * a.n;
    if (tid == 0) flag_bad = 0;
    __syncthreads();
    for (int i = tid; i < a.npad; i += FT) {
        unsigned long long k = ~0ull;
        if (i < a.n) {
            const long long x0 = rows[i * 4], x1 = rows[i * 4 + 1], x2 = rows[i * 4 + 2], x3 = rows[i * 4 + 3];
            if ((unsigned long long)(x0 | x1 | x2 | x3) >= (1ull << 15)) flag_bad = 1;      // also catches negatives
            k = ((unsigned long long)x0 << 48) | ((unsigned long long)x1 << 32) | ((unsigned long long)x2 << 16) |
                (unsigned long long)x3;
        }
        key[i] = k;
        idx[i] = i;
    }
    __syncthreads();
    if (flag_bad) {
        if (tid == 0) a.out_counts[item] = -1;
        return;
    }
    if (!BIG) {
        for (int size = 2; size <= a.npad; size <<= 1) bitonic_steps(key, idx, a.npad, 0, size, size >> 1);
    } else {
        auto load = [&](int c0) {
            for (int i = tid; i < CH; i += FT) { lkey[i] = key[c0 + i]; lidx[i] = idx[c0 + i]; }
            __syncthreads();
        };
        auto store = [&](int c0) {
            for (int i = tid; i < CH; i += FT) { key[c0 + i] = lkey[i]; idx[c0 + i] = lidx[i]; }
            __syncthreads();
        };
        for (int c0 = 0; c0 < a.npad; c0 += CH) {
            load(c0);
            for (int size = 2; size <= CH; size <<= 1) bitonic_steps(lkey, lidx, CH, c0, size, size >> 1);
            store(c0);
        }
        for (int size = 2 * CH; size <= a.npad; size <<= 1) {
            for (int stride = size >> 1; stride >= CH; stride >>= 1) {
                for (int t = tid; t < a.npad / 2; t += FT) {
                    const int lo = 2 * t - (t & (stride - 1));
                    cmp_xchg(key, idx, lo, lo + stride, (lo & size) == 0);
                }
                __syncthreads();
            }
            for (int c0 = 0; c0 < a.npad; c0 += CH) {
                load(c0);
                bitonic_steps(lkey, lidx, CH, c0, size, CH >> 1);
                store(c0);
            }
        }
    }
    // stage 1: run starts (first occurrences in sorted order), with `mutual` only runs longer than one
    const int per = (a.n + FT - 1) / FT, i0 = min(tid * per, a.n), i1 = min(i0 + per, a.n);
    int mine = 0;
    for (int i = i0; i < i1; ++i) {
        const bool start = (i == 0) || key[i] != key[i - 1];
        const bool more = (i + 1 < a.n) && key[i + 1] == key[i];
        mine += start && (!a.mutual || more);
    }
    int nsel;
    int at = block_exclusive_scan(mine, wave_sums, &nsel);
    if (nsel > 0) {
        for (int i = i0; i < i1; ++i) {
            const bool start = (i == 0) || key[i] != key[i - 1];
            const bool more = (i + 1 < a.n) && key[i + 1] == key[i];
            if (start && (!a.mutual || more)) pos[at++] = idx[i];
        }
    } else {
        nsel = a.n;                                  // nothing selected: the list stays as it was
        for (int i = i0; i < i1; ++i) pos[i] = i;
    }
    __syncthreads();
    // stage 2: score threshold on that list, "or everything"
    const int per2 = (nsel + FT - 1) / FT, j0 = min(tid * per2, nsel), j1 = min(j0 + per2, nsel);
    int pass = 0;
    for (int j = j0; j < j1; ++j) pass += sc[pos[j]] > a.thres;
    int npass;
    int at2 = block_exclusive_scan(pass, wave_sums, &npass);
    const bool all = (npass == 0);
    if (all) at2 = j0;
    for (int j = j0; j < j1; ++j) {
        const int src = pos[j];
        const float s = sc[src];
        if (all || s > a.thres) {
#pragma unroll
            for (int q = 0; q < 4; ++q) orow[(size_t)at2 * 4 + q] = rows[(size_t)src * 4 + q];
            osc[at2] = s;
            ++at2;
        }
    }
    if (tid == 0) a.out_counts[item] = all ? nsel : npass;
}

// The tail of estimate_matches (utils/eval/model_helper.py:92-109) for a batch of items with device-side counts:
// keep the rows whose fine score exceeds io_thres -- all rows if none does --, in order, and scale the refined and the
// coarse coordinates to original-image pixels in float64 (scale = [w1o/w1, h1o/h1, w2o/w2, h2o/h2] per item).
struct TailArgs {
    const float *fine;                        // [B][stride][4]
    const float *scores;                      // [B][stride]
    const long long *coarse;                  // [B][stride][4]
    const int *counts;                        // [B] valid rows per item (device); < 0 passes through as -1
    const double *scale;                      // [B][4]
    int stride;
    float thres;
    double *out_matches;                      // [B][stride][4]
    float *out_scores;                        // [B][stride]
    double *out_coarse;                       // [B][stride][4]
    int *out_counts;                          // [B]
};

__global__ __launch_bounds__(FT) void match_tail_kernel(TailArgs a) {
    __shared__ int wave_sums[FT / 64];
    const int tid = threadIdx.x, item = blockIdx.x;
    const int n = a.counts[item];
    if (n < 0) {
        if (tid == 0) a.out_counts[item] = -1;
        return;
    }
    const float *fine = a.fine + (size_t)item * a.stride * 4;
    const float *sc = a.scores + (size_t)item * a.stride;
    const long long *co = a.coarse + (size_t)item * a.stride * 4;
    double *om = a.out_matches + (size_t)item * a.stride * 4, *oc = a.out_coarse + (size_t)item * a.stride * 4;
    float *os = a.out_scores + (size_t)item * a.stride;
    const double s0 = a.scale[item * 4], s1 = a.scale[item * 4 + 1], s2 = a.scale[item * 4 + 2], s3 = a.scale[item * 4 + 3];
    const int per = (n + FT - 1) / FT, j0 = min(tid * per, n), j1 = min(j0 + per, n);
    int pass = 0;
    for (int j = j0; j < j1; ++j) pass += sc[j] > a.thres;
    int npass;
    int at = block_exclusive_scan(pass, wave_sums, &npass);
    const bool all = (npass == 0);
    if (all) at = j0;
    for (int j = j0; j < j1; ++j) {
        const float s = sc[j];
        if (all || s > a.thres) {
            om[(size_t)at * 4 + 0] = s0 * (double)fine[(size_t)j * 4 + 0];
            om[(size_t)at * 4 + 1] = s1 * (double)fine[(size_t)j * 4 + 1];
            om[(size_t)at * 4 + 2] = s2 * (double)fine[(size_t)j * 4 + 2];
            om[(size_t)at * 4 + 3] = s3 * (double)fine[(size_t)j * 4 + 3];
            oc[(size_t)at * 4 + 0] = s0 * (double)co[(size_t)j * 4 + 0];
            oc[(size_t)at * 4 + 1] = s1 * (double)co[(size_t)j * 4 + 1];
            oc[(size_t)at * 4 + 2] = s2 * (double)co[(size_t)j * 4 + 2];
            oc[(size_t)at * 4 + 3] = s3 * (double)co[(size_t)j * 4 + 3];
            os[at] = s;
            ++at;
        }
    }
    if (tid == 0) a.out_counts[item] = all ? n : npass;
}

}  // namespace p2p

using namespace p2p;

extern "C" int p2p_match_tail_batch(const float *fine, const float *scores, const int64_t *coarse, const int *counts,
                                    const double *scale, int batch, int stride, float io_thres, double *out_matches,
                                    float *out_scores, double *out_coarse, int *out_counts, p2p_stream_t stream) {
    P2P_REQUIRE(fine && scores && coarse && counts && scale && out_matches && out_scores && out_coarse && out_counts,
                P2P_EINVAL, "p2p_match_tail: null argument");
    P2P_REQUIRE(batch >= 1 && batch <= 65535 && stride >= 1, P2P_EINVAL, "p2p_match_tail: bad sizes");
    TailArgs a{fine, scores, (const long long *)coarse, counts, scale, stride, io_thres, out_matches, out_scores, out_coarse,
               out_counts};
    hipLaunchKernelGGL(match_tail_kernel, dim3(batch), dim3(FT), 0, (hipStream_t)stream, a);
    return check_launch("match_tail_kernel");
}

extern "C" size_t p2p_filter_coarse_workspace_bytes(int batch, int n) {
    if (batch < 1 || n < 1 || n > FILTER_MAX_ROWS) return 0;
    if (n <= FILTER_LDS_ROWS) return 0;
    size_t npad = 2;
    while (npad < (size_t)n) npad <<= 1;
    return (size_t)batch * npad * 16;
}

extern "C" int p2p_filter_coarse_batch(const int64_t *matches, const float *scores, int batch, int n, float ncn_thres,
                                       int mutual, int64_t *out_matches, float *out_scores, int *out_counts,
                                       void *workspace, size_t workspace_bytes, p2p_stream_t stream) {
    P2P_REQUIRE(matches && scores && out_matches && out_scores && out_counts, P2P_EINVAL, "p2p_filter_coarse: null argument");
    P2P_REQUIRE(batch >= 1 && batch <= 65535 && n >= 1, P2P_EINVAL, "p2p_filter_coarse: bad sizes");
    P2P_REQUIRE(n <= FILTER_MAX_ROWS, P2P_EUNSUPPORTED, "p2p_filter_coarse: %d rows per item (at most %d)", n, FILTER_MAX_ROWS);
    int npad = 2;
    while (npad < n) npad <<= 1;
    const bool big = n > FILTER_LDS_ROWS;
    const size_t need = p2p_filter_coarse_workspace_bytes(batch, n);
    P2P_REQUIRE(!big || (workspace && workspace_bytes >= need), P2P_ENOMEM,
                "p2p_filter_coarse: %d rows per item need a workspace of %zu bytes (p2p_filter_coarse_workspace_bytes), got %zu",
                n, need, workspace ? workspace_bytes : (size_t)0);
    const size_t lds = big ? (size_t)FILTER_LDS_ROWS * 12 : (size_t)npad * 16;
    int dev = 0;
    P2P_HIP_CHECK(hipGetDevice(&dev));
    static DeviceOnce attr_set;      // per device: a process may drive several GPUs
    if (!attr_set.done(dev)) {
        P2P_HIP_CHECK(hipFuncSetAttribute((const void *)filter_coarse_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          FILTER_LDS_ROWS * 16));
        P2P_HIP_CHECK(hipFuncSetAttribute((const void *)filter_coarse_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          FILTER_LDS_ROWS * 12));
        attr_set.set(dev);
    }
    FilterArgs a{(const long long *)matches, scores, n, npad, ncn_thres, mutual, (long long *)out_matches, out_scores, out_counts,
                 (unsigned char *)workspace};
    if (big) hipLaunchKernelGGL(filter_coarse_kernel<true>, dim3(batch), dim3(FT), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(filter_coarse_kernel<false>, dim3(batch), dim3(FT), lds, (hipStream_t)stream, a);
    return check_launch("filter_coarse_kernel");
}
