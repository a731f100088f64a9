// Neighbourhood consensus (reference networks/ncn/model.py:145-155, conv4d.py:12-74; 1 -> 16 -> 1 channels, kernel 3^4, both
// symmetric branches) as ONE kernel on the fp16 matrix cores: the 16-channel hidden volume never leaves the compute unit.
//
// Work-group = one branch x one output tile [TA a][TB b][TC c][TD d] of the volume Y[a][b][c][d].  It walks the hidden
// "strips" (a', b') that feed the tile, a' outer.  Per strip:
//   S1  the 9 planes (da, db) of the input X (first MutualMatching applied) around the strip are staged in LDS as two fp16
//       planes of X * 2^12 (|X| <= 1), rows c0-2 ... c0+TC+1, columns dt0-2 ... dt0+TD+1;
//   S2  layer 1 on v_mfma_f32_32x32x16_f16: the hidden positions of the strip are FLAT (q = row * P + column, pitch P), an
//       m-tile = 32 even positions, N = 16 channels x the position's two parities, K = 27 taps (da, db, dc) x a 4-wide
//       window along d (the 3 taps dd of both parities): the A fragment of a lane is two aligned 8-byte reads of X, the
//       weights (7 K-slabs, zero where a tap does not apply) live in registers.  bias + ReLU, zero outside the volume,
//       hidden values as two fp16 planes [plane][8-channel half][position][8 channels] in LDS;
//   S3  layer 2 on v_mfma_f32_16x16x32_f16 in "gather" form over the B taps: an m-tile = 16 flat OUTPUT positions, K = 9 taps
//       (dc, dd) x 16 channels (every A fragment is one aligned 16-byte read of the hidden planes), N = the 9 taps (da, db):
//       column n is this strip's contribution to the output plane (a' - da + 1, b' - db + 1), added to the tile's
//       accumulators in LDS with ds_add_f32.  Inside a strip no two adds meet (different positions or different planes),
//       strips are separated by barriers: the summation order of every output is fixed.
// An output slice a is complete once hidden slice a + 1 is done: relu(sum + b2) is added to Y (zeroed beforehand; the two
// branches are its two addends, so the result does not depend on which arrives first) and its accumulator slot recycled.
//
// Arithmetic: fp32-equivalent like the other 16-bit paths -- operands scaled by exact powers of two into the normal range of
// fp16 and split into two planes (2^-24 relative), three MFMA products per fp32 product, fp32 accumulation; the scales
// (2^12 on X, per-channel on the layer-1 weights, one per branch on the hidden planes and the layer-2 weights) are undone
// exactly.  The halo of the tile is recomputed ((TA+2)(TB+2)(TC+2)/(TA TB TC) of layer 1); HBM sees X once per tile
// neighbourhood (L2 hits) and Y once.
#include "p2p_common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace p2p {

typedef _Float16 nh8 __attribute__((ext_vector_type(8)));
typedef float nf4 __attribute__((ext_vector_type(4)));

constexpr int NCF_THREADS = 256, NCF_WAVES = 4;
constexpr int NCF_W1_BYTES = 7 * 2 * 64 * 16;      // layer-1 B fragments [slab][plane][lane][8 fp16]
constexpr int NCF_W2_BYTES = 5 * 2 * 64 * 16;      // layer-2 B fragments [K step][plane][lane][8 fp16]
constexpr int NCF_C_FLOATS = 64;                   // s1[16], b1[16], hscale, yunscale, padding
constexpr int NCF_BRANCH_BYTES = NCF_W1_BYTES + NCF_W2_BYTES + NCF_C_FLOATS * 4;

struct NcFusedArgs {
    const float *X;          // [nA][nB] per pair
    float *Y;                // [nA][nB] per pair; Y2 == null: zero on entry, both branches are ADDED to it (two addends per cell)
    float *Y2;               // optional [nA][nB] per pair: the transposed branch is stored here, the direct one in Y (plain stores)
    size_t stride;           // floats between pairs (both arrays)
    int d0, d1, d2, d3;
    int ta, tb, tc, td, P;   // tile, flat pitch (even, >= td + 4, <= 64)
    int na, nb, nc, nd;      // tiles per axis
    const unsigned char *w;  // [2 branches][NCF_BRANCH_BYTES]
    float b2;
    const int *xmax;         // per pair: float bits of max |X| (stride xmax_stride ints)
    size_t xmax_stride;
};

__host__ __device__ __forceinline__ int nc_yrow(int tc, int P) {
    int r = (tc * P + 3) & ~3;            // 16-byte rows
    while ((r & 63) != 4 && (r & 63) != 12 && (r & 63) != 20 && (r & 63) != 28 && (r & 63) != 36 && (r & 63) != 44 && (r & 63) != 52 && (r & 63) != 60) r += 4;
    return r;
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ unsigned short nf2h(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }
__device__ __forceinline__ float nh2f(unsigned short b) { return (float)__builtin_bit_cast(_Float16, b); }
#define NCF_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(nh8, (a)), __builtin_bit_cast(nh8, (b)), (c), 0, 0, 0)
#define NCF_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(nh8, (a)), __builtin_bit_cast(nh8, (b)), (c), 0, 0, 0)

__global__ __launch_bounds__(NCF_THREADS, 2) void nc_fused_kernel(NcFusedArgs a) {
    P2P_DYN_SHARED(unsigned char, sm);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int br = blockIdx.y;
    const float *X = a.X + (size_t)blockIdx.z * a.stride;
    float *Y = ((br && a.Y2) ? a.Y2 : a.Y) + (size_t)blockIdx.z * a.stride;
    const bool plain = a.Y2 != nullptr;
    int g = blockIdx.x;
    const int dt0 = (g % a.nd) * a.td; g /= a.nd;
    const int c0 = (g % a.nc) * a.tc; g /= a.nc;
    const int b0 = (g % a.nb) * a.tb; g /= a.nb;
    const int a0 = g * a.ta;
    const int TA = a.ta, TB = a.tb, TC = a.tc, TD = a.td, P = a.P;
    const int XROWS = TC + 4, HROWS = TC + 2;
    const int HN = max(HROWS * P, ((HROWS * P + 63) >> 6) * 64) + 2;      // whole layer-1 tiles fit (their tail rows store zeros)
    const int XPLANE = (9 * XROWS * P * 2 + 8 + 15) & ~15;   // bytes of one fp16 plane of the staged input (+ the window overrun of its
                                                             // last position; the hidden planes behind it need 16-byte alignment)
    const int HKH = HN * 16, HPLANE = 2 * HKH;               // hidden: [plane][channel half][position][8 x fp16]
    // accumulators: [3 slots][TB][TC * P (+ pad)] floats; the pad (stride = 4 mod 8 floats... = 16 B mod 32 B, and never a
    // multiple of 64 floats) spreads the nine planes a layer-2 tile adds into over the LDS banks
    const int YROW = nc_yrow(TC, P), YSLOT = TB * YROW;
    unsigned char *Xs = sm;
    unsigned char *Hs = sm + 2 * XPLANE;
    unsigned char *W2s = Hs + 2 * HPLANE;                    // layer-2 B fragments
    float *Ya = (float *)(W2s + NCF_W2_BYTES);
    const size_t nB = (size_t)a.d2 * a.d3;

    // weights of this branch -> registers (they are the B operands of every MFMA of the kernel)
    const unsigned char *wb = a.w + (size_t)br * NCF_BRANCH_BYTES;
    nf4 w1[7][2];
#pragma unroll
    for (int s = 0; s < 7; ++s)
#pragma unroll
        for (int p = 0; p < 2; ++p) w1[s][p] = *(const nf4 *)(wb + ((s * 2 + p) * 64 + lane) * 16);
    // (the 10 layer-2 fragments go to LDS: 40 more registers would not fit beside the prefetched input rows)
    for (int i = tid; i < NCF_W2_BYTES / 16; i += NCF_THREADS) *(nf4 *)(W2s + i * 16) = *(const nf4 *)(wb + NCF_W1_BYTES + i * 16);
    // (made opaque so that their loads are waited for HERE: a wait inside the strip loop would also drain the input rows
    // prefetched for the next strip)
#pragma unroll
    for (int s = 0; s < 7; ++s)
#pragma unroll
        for (int p = 0; p < 2; ++p) P2P_OPAQUE_V4(w1[s][p]);
    const float *cf = (const float *)(wb + NCF_W1_BYTES + NCF_W2_BYTES);
    // The fp16 planes are laid out for |X| <= 1 (what MutualMatching makes of non-negative correlations).  A volume with
    // larger values (negative correlations can do that) is scaled down by the power of two 2^E that brings its largest
    // magnitude below 1; the hidden planes and the two un-scalings follow (all exact).
    const int xeb = (a.xmax[(size_t)blockIdx.z * a.xmax_stride] >> 23) & 0xff;
    const int E = min(max(xeb - 126, 0), 100);
    const float up = __int_as_float((127 + E) << 23), down = __int_as_float((127 - E) << 23);
    const float xscale = 4096.0f * down;
    const float s1o = cf[lane & 15] * up, b1o = cf[16 + (lane & 15)], hscale = cf[32] * down, yun = cf[33] * up;

    for (int i = tid; i < 3 * YSLOT; i += NCF_THREADS) Ya[i] = 0.f;
    for (int i = tid; i < 2 * 2 * 2; i += NCF_THREADS) {     // the pad slots before and after the hidden positions stay zero
        const int pl = i >> 2, kh = (i >> 1) & 1, end = i & 1;
        *(nf4 *)(Hs + pl * HPLANE + kh * HKH + (end ? (HN - 1) * 16 : 0)) = (nf4){0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();

    const int a_hi = min(a0 + TA, a.d0);                     // outputs of the tile: [a0, a_hi)
    const int ap_first = max(a0 - 1, 0), ap_last = min(a0 + TA, a.d0 - 1);
    const int bp_first = max(b0 - 1, 0), bp_last = min(b0 + TB, a.d1 - 1);

    auto flush = [&](int aout) {
        // relu(sum + b2) of output slice aout -> Y; the slot becomes zero again
        __syncthreads();
        float *slot = Ya + (aout % 3) * YSLOT;
        for (int r = wave; r < TB * TC; r += NCF_WAVES) {
            const int bb = r / TC, ro = r - bb * TC;
            const int ib = b0 + bb, ic = c0 + ro, id = dt0 + lane;
            if (lane < TD && ib < a.d1 && ic < a.d2 && id < a.d3) {
                const float v = fmaxf(slot[bb * YROW + ro * P + lane + 1] + a.b2, 0.f);
                float *dst = Y + ((size_t)aout * a.d1 + ib) * nB + (size_t)ic * a.d3 + id;
                if (plain) *dst = v; else unsafeAtomicAdd(dst, v);
            }
        }
        __syncthreads();
        for (int i = tid; i < YSLOT; i += NCF_THREADS) slot[i] = 0.f;
    };

    // The strips (a', b') in order, a' outer.  The input rows of the NEXT strip are fetched into registers while the current
    // one is computed (their latency is a microsecond).  Row addresses are wave-uniform (scalar base + the lane's clamped
    // column), everything that does not depend on the strip is worked out once, here.
    constexpr int NCF_XJ = 3;                                // input rows per wave and plane: (TC + 4) / 4 <= 3
    const int nbp = bp_last - bp_first + 1, nstrips = (ap_last - ap_first + 1) * nbp;
    float xv[9 * NCF_XJ];
    unsigned xok_lo = 0, xok_hi = 0;                         // bit (pl9 * 4 + j): that row of the fetched strip is inside the volume
    const bool okd = lane < P && dt0 - 2 + lane >= 0 && dt0 - 2 + lane < a.d3;
    // 32-bit element offsets (a volume has < 2^31 cells): row part = clamped c row * d3 + the lane's clamped column (does
    // not depend on the strip), plane part = clamped (a, b) * strides (three values each per strip, scalar)
    const int sB = (int)nB, sA = a.d1 * sB;
    int rowoff[NCF_XJ];
    unsigned okc = 0;
#pragma unroll
    for (int j = 0; j < NCF_XJ; ++j) {
        const int xr = wave + NCF_WAVES * j, ic = c0 - 2 + xr;
        rowoff[j] = clampi(ic, 0, a.d2 - 1) * a.d3 + clampi(dt0 - 2 + lane, 0, a.d3 - 1);
        okc |= (unsigned)(xr < XROWS && ic >= 0 && ic < a.d2) << j;
    }
    auto fetch = [&](int strip) {
        const int ap = ap_first + strip / nbp, bp = bp_first + strip % nbp;
        int pa[3], pb[3];
        unsigned oka = 0, okb = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            pa[i] = clampi(ap + i - 1, 0, a.d0 - 1) * sA;
            pb[i] = clampi(bp + i - 1, 0, a.d1 - 1) * sB;
            oka |= (unsigned)(ap + i - 1 >= 0 && ap + i - 1 < a.d0) << i;
            okb |= (unsigned)(bp + i - 1 >= 0 && bp + i - 1 < a.d1) << i;
        }
        unsigned lo = 0, hi = 0;
#pragma unroll
        for (int pl9 = 0; pl9 < 9; ++pl9) {
            const int base = pa[pl9 / 3] + pb[pl9 % 3];
            const unsigned okab = (oka >> (pl9 / 3)) & (okb >> (pl9 % 3)) & 1u;
#pragma unroll
            for (int j = 0; j < NCF_XJ; ++j) {
                // unconditional load from a clamped address; the zero padding is selected when the value is USED (a load in a
                // branch, or a select right behind it, is waited for on the spot: dozens of round trips one after the other)
                xv[pl9 * NCF_XJ + j] = X[base + rowoff[j]];
                const int bit = pl9 * NCF_XJ + j;
                const unsigned ok = okab & (okc >> j);
                if (bit < 32) lo |= ok << bit; else hi |= ok << (bit - 32);
            }
        }
        xok_lo = okd ? lo : 0u;
        xok_hi = okd ? hi : 0u;
    };

    // layer 1: this wave's m-tiles (<= 2): store offsets, validity of the 16 accumulator rows of the lane
    const int l31 = lane & 31, kb5 = lane >> 5, ch = lane & 15, par = (lane >> 4) & 1;
    const int nt1 = (HROWS * P + 63) >> 6;
    unsigned s2ok[2] = {0, 0};
    int s2dst[2] = {0, 0}, s2qh[2] = {0, 0};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = wave + NCF_WAVES * u, q0 = t * 64;
        s2qh[u] = min(q0 + 2 * l31, HROWS * P - 2);           // rows past the strip repeat its last row (never stored)
        const int fl0 = q0 + 8 * kb5 + par;                   // D: row i = (r & 3) + 8 (r >> 2) + 4 kb -> position q0 + 2 i + par
        s2dst[u] = (ch >> 3) * HKH + (ch & 7) * 2 + 16 + fl0 * 16;     // + 16: position -1 is slot 0
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int fl = fl0 + 2 * (r & 3) + 16 * (r >> 2), rowh = fl / P, col = fl - rowh * P;
            const int ic = c0 - 1 + rowh, id = dt0 + col - 1;
            const bool ok = t < nt1 && rowh < HROWS && col <= TD + 1 && ic >= 0 && ic < a.d2 && id >= 0 && id < a.d3;
            s2ok[u] |= (unsigned)ok << r;
        }
    }
    // layer 2: hidden offsets of the lane's K block per step
    const int row16 = lane & 15, kb4 = lane >> 4;
    int s3off[5];
#pragma unroll
    for (int st = 0; st < 5; ++st) {
        const int tap = min(2 * st + (kb4 >> 1), 8);          // tap 9 does not exist (zero weights)
        s3off[st] = (kb4 & 1) * HKH + ((tap / 3) * P + tap % 3) * 16;      // hidden position + dc P + dd - 1, slot + 1
    }
    const int nt2 = (TC * P + 15) >> 4;

    if (nstrips > 0) fetch(0);
    for (int strip = 0; strip < nstrips; ++strip) {
        const int ap = ap_first + strip / nbp, bp = bp_first + strip % nbp;
        // ---------------- S1: the nine input planes around the strip -> two fp16 planes of X * 2^12 in LDS
#pragma unroll
        for (int pl9 = 0; pl9 < 9; ++pl9)
#pragma unroll
            for (int j = 0; j < NCF_XJ; ++j) {
                const int xr = wave + NCF_WAVES * j, bit = pl9 * NCF_XJ + j;
                const bool okv = (bit < 32) ? (xok_lo >> bit) & 1u : (xok_hi >> (bit - 32)) & 1u;
                const float v = okv ? xv[bit] * xscale : 0.f;
                const unsigned short h0 = nf2h(v), h1 = nf2h(v - nh2f(h0));
                // columns (c, c + 1) sit in adjacent lanes: the even lane stores the pair's first plane, the odd lane its second
                const bool oddl = lane & 1;
                const unsigned mine = oddl ? h1 : h0, give = oddl ? h0 : h1;
                const unsigned got = P2P_SWAP_ADJACENT(give);
                if (xr < XROWS && lane < P)
                    *(unsigned *)(Xs + (oddl ? XPLANE : 0) + ((pl9 * XROWS + xr) * P + (lane & ~1)) * 2) = oddl ? (got | mine << 16) : (mine | got << 16);
            }
        __syncthreads();
        if (strip + 1 < nstrips) fetch(strip + 1);
        // ---------------- S2: layer 1 -> hidden planes
#ifndef NCF_SKIP_S2                     // timing experiments (wrong results): NCF_SKIP_S2 / _S3 / _S2EPI drop one part
        {
            // both m-tiles of the wave together (independent accumulators, the weights are shared); a wave with one tile
            // multiplies a clamped copy of it -- it would wait at the barrier for the others anyway
            f32x16 acc[2] = {{0}, {0}};
#pragma unroll
            for (int sl = 0; sl < 7; ++sl) {
                // the lane's two tap groups of this slab: g = 4 * sl + 2 * kb + {0, 1}; group 27 does not exist (zero weights)
                const int g0 = 4 * sl + 2 * kb5, g1 = min(g0 + 1, 26);
                const int o0 = ((g0 / 3) * XROWS + g0 % 3) * P, o1 = ((g1 / 3) * XROWS + g1 % 3) * P;
                nf4 av[2][2];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const unsigned *x0 = (const unsigned *)(Xs + p * XPLANE + (o0 + s2qh[u]) * 2);
                        const unsigned *x1 = (const unsigned *)(Xs + p * XPLANE + (o1 + s2qh[u]) * 2);
                        av[u][p] = (nf4){__uint_as_float(x0[0]), __uint_as_float(x0[1]), __uint_as_float(x1[0]), __uint_as_float(x1[1])};
                    }
                acc[0] = NCF_MFMA32(av[0][1], w1[sl][0], acc[0]); acc[1] = NCF_MFMA32(av[1][1], w1[sl][0], acc[1]);
                acc[0] = NCF_MFMA32(av[0][0], w1[sl][1], acc[0]); acc[1] = NCF_MFMA32(av[1][0], w1[sl][1], acc[1]);
                acc[0] = NCF_MFMA32(av[0][0], w1[sl][0], acc[0]); acc[1] = NCF_MFMA32(av[1][0], w1[sl][0], acc[1]);
            }
#ifndef NCF_SKIP_S2EPI
            // bias, ReLU, zero outside the volume, scale, split.  Channel pairs (o, o + 1) sit in adjacent lanes: the even lane
            // stores the pair's first plane, the odd lane its second plane -- one 4-byte store per lane and row instead of two
            // 2-byte ones
            const bool odd = ch & 1;
            const float s1h = s1o * hscale, b1h = b1o * hscale;      // relu(x) * 2^k = relu(x * 2^k)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (wave + NCF_WAVES * u >= nt1) break;               // (wave-uniform) the clamped copy is not stored
                unsigned char *hdst = Hs + s2dst[u] + (odd ? HPLANE - 2 : 0);      // the pair's first channel in this lane's plane
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float h = ((s2ok[u] >> r) & 1u) ? fmaxf(fmaf(acc[u][r], s1h, b1h), 0.f) : 0.f;
                    const unsigned short h0 = nf2h(h), h1 = nf2h(h - nh2f(h0));
                    const unsigned mine = odd ? h1 : h0, give = odd ? h0 : h1;       // keep the half of my plane, hand the other to the partner
                    const unsigned got = P2P_SWAP_ADJACENT(give);
                    const unsigned word = odd ? (got | mine << 16) : (mine | got << 16);
                    *(unsigned *)(hdst + (2 * (r & 3) + 16 * (r >> 2)) * 16) = word;
                }
            }
#else
            if (acc[0][0] == 12345.f || acc[1][0] == 12345.f) Hs[tid] = 1;
#endif
        }
#endif
        __syncthreads();
        // ---------------- S3: layer 2, contributions of the strip to the 3 x 3 output planes around it
#ifndef NCF_SKIP_S3
        {
            const int n = lane & 15, da = n / 3, db = n - 3 * da;
            const int aout = ap - da + 1, bout = bp - db + 1;
            const bool lane_ok = n < 9 && aout >= a0 && aout < a_hi && bout >= b0 && bout < min(b0 + TB, a.d1);
            float *ydst = Ya + ((aout + 3) % 3) * YSLOT + (bout - b0) * YROW;
            // two m-tiles at a time (independent accumulators), the fragments of step st + 1 in flight during the MFMAs of step st
            const bool yalign = (YROW & 3) == 0;               // every (slot, plane) row of the accumulators starts 16-byte aligned
            for (int t = wave; t < nt2; t += 2 * NCF_WAVES) {
                const int qa = t * 16, qb = qa + NCF_WAVES * 16;
                const unsigned char *ha = Hs + min(qa + row16, TC * P - 1) * 16, *hb = Hs + min(qb + row16, TC * P - 1) * 16;
                const unsigned char *wl = W2s + lane * 16;
                nf4 accA = {0.f, 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f};
                nf4 a0v = *(const nf4 *)(ha + s3off[0]), a1v = *(const nf4 *)(ha + s3off[0] + HPLANE);
                nf4 b0v = *(const nf4 *)(hb + s3off[0]), b1v = *(const nf4 *)(hb + s3off[0] + HPLANE);
                nf4 w0v = *(const nf4 *)wl, w1v = *(const nf4 *)(wl + 1024);
#pragma unroll
                for (int st = 0; st < 5; ++st) {
                    nf4 na0 = a0v, na1 = a1v, nb0 = b0v, nb1 = b1v, nw0 = w0v, nw1 = w1v;
                    if (st < 4) {
                        na0 = *(const nf4 *)(ha + s3off[st + 1]); na1 = *(const nf4 *)(ha + s3off[st + 1] + HPLANE);
                        nb0 = *(const nf4 *)(hb + s3off[st + 1]); nb1 = *(const nf4 *)(hb + s3off[st + 1] + HPLANE);
                        nw0 = *(const nf4 *)(wl + (st + 1) * 2048); nw1 = *(const nf4 *)(wl + (st + 1) * 2048 + 1024);
                    }
                    accA = NCF_MFMA16(a1v, w0v, accA); accB = NCF_MFMA16(b1v, w0v, accB);
                    accA = NCF_MFMA16(a0v, w1v, accA); accB = NCF_MFMA16(b0v, w1v, accB);
                    accA = NCF_MFMA16(a0v, w0v, accA); accB = NCF_MFMA16(b0v, w0v, accB);
                    a0v = na0; a1v = na1; b0v = nb0; b1v = nb1; w0v = nw0; w1v = nw1;
                }
                // D: row 4 kb + r = output position q + 4 kb + r, column n = plane (da, db).  Plain read-add-write: inside a
                // strip every accumulator word is touched by exactly one lane (other tiles = other positions, other lanes =
                // other planes or rows), strips are separated by barriers.  (ds_add_f32 cost ~600 cycles per wave instruction.)
                if (lane_ok) {
                    float *ya = ydst + qa + 4 * kb4, *yb = ydst + qb + 4 * kb4;
                    if (yalign) {
                        if (qa + 4 * kb4 + 3 < TC * P) {
                            nf4 v = *(nf4 *)ya;
                            v += accA * yun;
                            *(nf4 *)ya = v;
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) if (qa + 4 * kb4 + r < TC * P) ya[r] += accA[r] * yun;
                        }
                        if (qb + 4 * kb4 + 3 < TC * P) {
                            nf4 v = *(nf4 *)yb;
                            v += accB * yun;
                            *(nf4 *)yb = v;
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) if (qb + 4 * kb4 + r < TC * P) yb[r] += accB[r] * yun;
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (qa + 4 * kb4 + r < TC * P) ya[r] += accA[r] * yun;
                            if (qb + 4 * kb4 + r < TC * P) yb[r] += accB[r] * yun;
                        }
                    }
                }
            }
        }
#endif
        if (bp == bp_last && ap - 1 >= a0) flush(ap - 1);
    }
    if (nstrips > 0 && ap_last < a_hi && ap_last >= a0) flush(ap_last);      // the volume ends inside the tile: its last slice has no slice above
}

// ---- host side ------------------------------------------------------------------------------------------------
static uint16_t h_f2h(float v) { return __builtin_bit_cast(uint16_t, (_Float16)v); }
static float h_h2f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
static void put2(unsigned char *frag, int lane, int j, float v) {      // frag: [plane 2][lane 64][8 fp16]
    uint16_t *d = (uint16_t *)frag;
    const uint16_t h0 = h_f2h(v);
    d[lane * 8 + j] = h0;
    d[(64 + lane) * 8 + j] = h_f2h(v - h_h2f(h0));
}
static int pow2_exponent_to(float mx, int target) {      // t with mx * 2^t in [2^(target-1), 2^target)
    if (!(mx > 0.f) || !std::isfinite(mx)) return 0;
    int e;
    std::frexp(mx, &e);
    return target - e;
}

// w1 / w2 in the reference's stored layout (conv4d.py:119-120): w1s[da][o][0][db][dc][dd], w2s[da][0][c][db][dc][dd]
void pack_nc_fused(const float *w1, const float *b1, const float *w2, std::vector<unsigned char> &out) {
    out.assign(2 * NCF_BRANCH_BYTES, 0);
    auto W1 = [&](int o, int da, int db, int dc, int dd) { return w1[(((da * 16 + o) * 3 + db) * 3 + dc) * 3 + dd]; };
    auto W2 = [&](int c, int da, int db, int dc, int dd) { return w2[(((da * 16 + c) * 3 + db) * 3 + dc) * 3 + dd]; };
    for (int br = 0; br < 2; ++br) {
        // the transposed branch evaluates conv(x^T)^T: the same volume with A/B-swapped taps
        auto W1b = [&](int o, int da, int db, int dc, int dd) { return br ? W1(o, dc, dd, da, db) : W1(o, da, db, dc, dd); };
        auto W2b = [&](int c, int da, int db, int dc, int dd) { return br ? W2(c, dc, dd, da, db) : W2(c, da, db, dc, dd); };
        unsigned char *base = out.data() + (size_t)br * NCF_BRANCH_BYTES;
        int t1[16];
        float hmax = 0.f, w2max = 0.f;
        for (int o = 0; o < 16; ++o) {
            float mx = 0.f, sum = std::fabs(b1[o]);
            for (int t = 0; t < 81; ++t) {
                const float v = W1b(o, t / 27, (t / 9) % 3, (t / 3) % 3, t % 3);
                mx = std::max(mx, std::fabs(v));
                sum += std::fabs(v);
            }
            t1[o] = pow2_exponent_to(mx, 12);
            hmax = std::max(hmax, sum);                 // |hidden| <= sum |w| + |b| for |X| <= 1
            for (int t = 0; t < 81; ++t) w2max = std::max(w2max, std::fabs(W2b(o, t / 27, (t / 9) % 3, (t / 3) % 3, t % 3)));
        }
        const int sh = pow2_exponent_to(2.f * hmax, 13), t2 = pow2_exponent_to(w2max, 12);
        // layer 1: k = slab * 16 + 8 * (lane >> 5) + j = 4 g + e, g = (da, db, dc), e = window column; n = lane & 31 = (s, o)
        for (int sl = 0; sl < 7; ++sl)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int k = sl * 16 + 8 * (lane >> 5) + j, gq = k >> 2, e = k & 3;
                    const int n = lane & 31, o = n & 15, s = n >> 4, dd = e - s;
                    float v = 0.f;
                    if (gq < 27 && dd >= 0 && dd <= 2) v = std::ldexp(W1b(o, gq / 9, (gq / 3) % 3, gq % 3, dd), t1[o]);
                    put2(base + sl * 2 * 64 * 16, lane, j, v);
                }
        // layer 2: k = step * 32 + 8 * (lane >> 4) + j: tap (dc, dd) = 2 step + (lane >> 5), channel 8 ((lane >> 4) & 1) + j; n = lane & 15 = (da, db)
        for (int st = 0; st < 5; ++st)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int kb = lane >> 4, tap = 2 * st + (kb >> 1), ch = 8 * (kb & 1) + j, n = lane & 15;
                    float v = 0.f;
                    if (tap < 9 && n < 9) v = std::ldexp(W2b(ch, n / 3, n % 3, tap / 3, tap % 3), t2);
                    put2(base + NCF_W1_BYTES + st * 2 * 64 * 16, lane, j, v);
                }
        float *cf = (float *)(base + NCF_W1_BYTES + NCF_W2_BYTES);
        for (int o = 0; o < 16; ++o) {
            cf[o] = std::ldexp(1.f, -12 - t1[o]);      // accumulators of layer 1 carry 2^12 (X) x 2^t1[o] (weights)
            cf[16 + o] = b1[o];
        }
        cf[32] = std::ldexp(1.f, sh);                  // hidden values are stored times 2^sh
        cf[33] = std::ldexp(1.f, -sh - t2);            // accumulators of layer 2 carry 2^sh x 2^t2
    }
}

size_t nc_fused_lds_bytes(int tb, int tc, int P) {
    return (size_t)2 * ((9 * (tc + 4) * P * 2 + 8 + 15) & ~15) + (size_t)2 * 2 * (std::max((tc + 2) * P, (((tc + 2) * P + 63) >> 6) * 64) + 2) * 16 + (size_t)3 * tb * nc_yrow(tc, P) * 4 + 256 + NCF_W2_BYTES;
}

// float bits of max |x| over n values per pair -> out[pair * out_stride] (zero beforehand); one atomic per wave
__global__ __launch_bounds__(256) void absmax_kernel(const float *__restrict__ x, size_t n, size_t stride, int *out, size_t out_stride) {
    x += (size_t)blockIdx.z * stride;
    float m = 0.f;
    const size_t n4 = (((size_t)x & 15) == 0) ? n >> 2 : 0;          // 16-byte loads where the item starts aligned
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const nf4 v = ((const nf4 *)x)[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    for (size_t i = 4 * n4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    int *dst = out + (size_t)blockIdx.z * out_stride;
    if ((threadIdx.x & 63) == 0 && __float_as_int(m) > *(volatile int *)dst) atomicMax(dst, __float_as_int(m));
}

int launch_absmax(const float *x, size_t n, size_t stride, int pairs, int *out, size_t out_stride, hipStream_t stream) {
    const unsigned blocks = (unsigned)std::min<size_t>((n + 4095) / 4096, 1024);
    hipLaunchKernelGGL(absmax_kernel, dim3(blocks, 1, pairs), dim3(256), 0, stream, x, n, stride, out, out_stride);
    return check_launch("absmax_kernel");
}

// Tile of the fused kernel for a volume and batch: the B row in pieces of <= 60 columns; (TB, TC) = (5, 8) keeps two
// work-groups per compute unit (78 KB of LDS each) and gives 7 + 22 well-balanced MFMA tiles per strip; the march along a is
// split only until the launch has >= 1024 work-groups (two per compute unit, twice over).  Returns the number of work-groups.
static long nc_fused_tile(NcFusedArgs &a, int pairs, const int *forced) {
    const int d0 = a.d0, d1 = a.d1, d2 = a.d2, d3 = a.d3;
    a.nd = ceil_div(d3, 60);
    a.td = ceil_div(d3, a.nd);
    a.P = (a.td + 4 + 1) & ~1;
    a.tb = 5; a.tc = 8; a.ta = 0;
    if (forced && forced[1] > 0 && forced[2] > 0) { a.ta = forced[0]; a.tb = forced[1]; a.tc = forced[2]; }      // (ta = 0: pick)
    a.tb = std::min(a.tb, d1); a.tc = std::min(a.tc, d2);
    // kernel limits: a row of <= 64 columns per wave instruction, <= 3 input rows per wave and plane, <= 2 layer-1 tiles per wave
    while (a.tc > 1 && (a.tc + 4 > 12 || (a.tc + 2) * a.P > 512)) --a.tc;
    a.nb = ceil_div(d1, a.tb); a.nc = ceil_div(d2, a.tc);
    if (a.ta <= 0) {
        a.ta = d0;
        while (a.ta > 4 && (long)ceil_div(d0, a.ta) * a.nb * a.nc * a.nd * 2 * pairs < 1024) a.ta = (a.ta + 1) / 2;
    }
    a.ta = std::min(a.ta, d0);
    a.na = ceil_div(d0, a.ta);
    return (long)a.na * a.nb * a.nc * a.nd * 2 * pairs;
}

int launch_nc_fused(const float *X, float *Y, float *Y2, size_t stride, int pairs, int d0, int d1, int d2, int d3,
                    const unsigned char *w_dev, float b2, const int *xmax, size_t xmax_stride, const int *forced_tile,
                    hipStream_t stream) {
    NcFusedArgs a{};
    a.X = X; a.Y = Y; a.Y2 = Y2; a.stride = stride; a.d0 = d0; a.d1 = d1; a.d2 = d2; a.d3 = d3; a.w = w_dev; a.b2 = b2;
    a.xmax = xmax; a.xmax_stride = xmax_stride;
    nc_fused_tile(a, pairs, forced_tile);
    const size_t lds = nc_fused_lds_bytes(a.tb, a.tc, a.P);
    P2P_REQUIRE(a.P <= 64 && lds <= 160 * 1024 && a.tc + 4 <= 12 && (a.tc + 2) * a.P <= 512, P2P_EUNSUPPORTED,
                "consensus tile does not fit (P %d, tc %d, LDS %zu)", a.P, a.tc, lds);
    int dev = 0;
    P2P_HIP_CHECK(hipGetDevice(&dev));
    static bool attr_set[64] = {false};
    if (dev >= 64 || !attr_set[dev]) {
        P2P_HIP_CHECK(hipFuncSetAttribute((const void *)nc_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if (dev < 64) attr_set[dev] = true;
    }
    hipLaunchKernelGGL(nc_fused_kernel, dim3(a.na * a.nb * a.nc * a.nd, 2, pairs), dim3(NCF_THREADS), lds, stream, a);
    return check_launch("nc_fused_kernel");
}

}  // namespace p2p
