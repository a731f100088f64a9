// Neighbourhood consensus (reference networks/ncn/model.py:145-155, conv4d.py:12-74; 1 -> 16 -> 1 channels, kernel 3^4, both
// symmetric branches) as ONE kernel on the fp16 matrix cores: the 16-channel hidden volume never leaves the compute unit.
//
// Work-group = one branch x one output tile [TA a][TB b][TC c][TD d] of the volume Y[a][b][c][d].  It walks the hidden
// "strips" (a', b') that feed the tile in SERPENTINE order (a' ascending; b' ascending in even rows of the GLOBAL index a',
// descending in odd ones), so that every step brings exactly ONE new triple of input planes.  Per strip:
//   S1  the 9 planes (da, db) of the input X (first MutualMatching applied) around the strip live in LDS as two fp16 planes
//       of X * 2^12 (|X| <= 1), rows c0-2 ... c0+TC+1, columns dt0-2 ... dt0+TD+1, in a 2-D RING: plane (a, b) sits in slot
//       (a mod 3, b mod 3).  The new triple of the NEXT strip is loaded into registers at the top of this strip (one set of
//       load destinations) and converted / stored behind this strip's layer 2: two barriers per strip, none for the staging;
//   S2  layer 1 on v_mfma_f32_32x32x16_f16 with the CHANNELS as MFMA rows: A = the 7 weight fragments (registers, zero where
//       a tap does not apply), B = 32 even flat positions (q = row * P + column, pitch P) x K = 27 taps (da, db, dc) x a
//       4-wide window along d (the 3 taps dd of both parities; two aligned 8-byte reads of X per lane and tap group).  The A
//       rows are ordered so that the 16 D registers of a lane are the 16 channels of ONE position (parity = lane >> 5):
//       bias + ReLU, zero outside the volume, split -> one 16-byte store per channel half and plane,
//       hidden planes [plane][8-channel half][position][8 x fp16]; a channel half is a multiple of 256 bytes long, so both
//       halves sit on the same bank slots and layer 2's fragment reads are conflict-free;
//   S3  layer 2 on v_mfma_f32_16x16x32_f16 in "gather" form over the B taps: an m-tile = 16 flat OUTPUT positions, K = 9 taps
//       (dc, dd) x 16 channels (every A fragment is one aligned 16-byte read of the hidden planes; the 10 B fragments of
//       the weights live in registers), N = the 9 taps (da, db): column n is this strip's contribution to the output plane
//       (a' - da + 1, b' - db + 1), added to the tile's accumulators in LDS by plain read-add-write (inside a strip no two
//       lanes touch the same word; ds_add_f32 cost ~600 cycles per wave instruction).  Strips are separated by barriers and
//       their order depends on the global a' only: the summation order of every output cell is fixed, whatever the tile.
// An output slice a is complete once hidden slice a + 1 is done: relu(sum + b2) is written out (the two branches into two
// arrays, or added into one that was zeroed beforehand) at the top of the next strip and its accumulator slot recycled.
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

// Slots (positions of 16 bytes) per channel half of a hidden plane: whole layer-1 tiles (their tail rows store zeros) + the
// two zero slots around them, rounded up to a multiple of 16 slots = 256 bytes: the channel halves of a plane then sit on
// the same LDS bank slots, and the 16 lanes of a ds_read_b128 service group in layer 2 -- rows {0-3, 12-15} of one half and
// rows {4-11} of the other -- read 16 different slots (with 450 slots per half the halves were two slots apart: every
// A-fragment read of layer 2 took two LDS cycles per group instead of one).
__host__ __device__ __forceinline__ int nc_hidden_used(int tc, int P) {
    const int n = (tc + 2) * P, r = ((n + 63) >> 6) * 64;
    return (n > r ? n : r) + 2;
}
__host__ __device__ __forceinline__ int nc_hidden_slots(int tc, int P) { return (nc_hidden_used(tc, P) + 15) & ~15; }
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

// two fp32 -> one dword of two fp16 (round to nearest even): v_cvt_pk_f16_f32 (the bit_cast of a _Float16 pair)
typedef _Float16 nh2 __attribute__((ext_vector_type(2)));
typedef float nf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned npk(float x, float y) {
    const nf2 v = {x, y};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, nh2));
}
__device__ __forceinline__ float npk_lo(unsigned h) { return (float)__builtin_bit_cast(nh2, h)[0]; }
__device__ __forceinline__ float npk_hi(unsigned h) { return (float)__builtin_bit_cast(nh2, h)[1]; }

// Staging of the input planes (S1).  One wave instruction moves two rows (lane >> 5) x 32 column PAIRS (lane & 31) of one
// plane: an item = (plane, pair of rows), six row pairs per plane (XROWS <= 12).  The planes of a strip live in a RING over
// b: slot (da, b mod 3), so a step along b restages only the three planes (da, b' + 1) -- NCF_KRING items per wave,
// prefetched into registers a phase ahead -- and only the first strip of an a' row all nine (three such triples).
#ifdef NCF_TIMING                       // phase lengths in s_memtime ticks, summed over the strips of one work-group (tools/ncf_timing.py)
#define NT_DECL unsigned nt_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned nt_last_ = (unsigned)__builtin_amdgcn_s_memtime();
#define NT(i) { const unsigned n_ = (unsigned)__builtin_amdgcn_s_memtime(); nt_[i] += n_ - nt_last_; nt_last_ = n_; }
#else
#define NT_DECL
#define NT(i)
#endif
constexpr int NCF_KRING = 5;     // items per wave and triple of planes: ceil(3 planes x 6 row pairs / 4 waves)

// FIXED: the tile (tb, tc, td, P) = (5, 8, 40, 44) of every 480x640 / 960x1280 pair (pooled rows of 40 or 80 cells) as
// compile-time constants -- strides, divisions and the LDS carve-up fold away (the generic body spends more instructions on
// scalar bookkeeping than on the matrix pipe: ~2100 per strip and wave for 72 MFMAs); any other shape runs the same body
// with the runtime values.
template <bool FIXED>
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
    const int TA = a.ta, TB = FIXED ? 5 : a.tb, TC = FIXED ? 8 : a.tc, TD = FIXED ? 40 : a.td, P = FIXED ? 44 : a.P;
    const int XROWS = TC + 4, HROWS = TC + 2;
    const int HNU = FIXED ? nc_hidden_used(8, 44) : nc_hidden_used(TC, P), HN = FIXED ? nc_hidden_slots(8, 44) : nc_hidden_slots(TC, P);
    const int XPLANE = (9 * XROWS * P * 2 + 8 + 15) & ~15;   // bytes of one fp16 plane of the staged input (+ the window overrun of its
                                                             // last position; the hidden planes behind it need 16-byte alignment)
    const int HKH = HN * 16, HPLANE = 2 * HKH;               // hidden: [plane][channel half][position][8 x fp16]
    // accumulators: [3 slots][TB][TC * P (+ pad)] floats; the pad (stride = 4 mod 8 floats... = 16 B mod 32 B, and never a
    // multiple of 64 floats) spreads the nine planes a layer-2 tile adds into over the LDS banks
    const int YROW = nc_yrow(TC, P), YSLOT = TB * YROW;
    unsigned char *Xs = sm;
    unsigned char *Hs = sm + 2 * XPLANE;
    float *Ya = (float *)(Hs + 2 * HPLANE);
    const size_t nB = (size_t)a.d2 * a.d3;

    // weights of this branch -> registers (layer 1: the A operands of its MFMAs)
    const unsigned char *wb = a.w + (size_t)br * NCF_BRANCH_BYTES;
    nf4 w1[7][2];
#pragma unroll
    for (int s = 0; s < 7; ++s)
#pragma unroll
        for (int p = 0; p < 2; ++p) w1[s][p] = *(const nf4 *)(wb + ((s * 2 + p) * 64 + lane) * 16);
    // ... and so do the 10 layer-2 B fragments (two waves per SIMD leave a wave 256 registers; re-reading them from LDS for
    // every pair of m-tiles was a third of layer 2's ds_read_b128 traffic)
    nf4 w2[5][2];
#pragma unroll
    for (int s = 0; s < 5; ++s)
#pragma unroll
        for (int p = 0; p < 2; ++p) w2[s][p] = *(const nf4 *)(wb + NCF_W1_BYTES + ((s * 2 + p) * 64 + lane) * 16);
    // (made opaque so that their loads are waited for HERE: a wait inside the strip loop would also drain the input rows
    // prefetched for the next strip)
#pragma unroll
    for (int s = 0; s < 7; ++s)
#pragma unroll
        for (int p = 0; p < 2; ++p) P2P_OPAQUE_V4(w1[s][p]);
#pragma unroll
    for (int s = 0; s < 5; ++s)
#pragma unroll
        for (int p = 0; p < 2; ++p) P2P_OPAQUE_V4(w2[s][p]);
    const float *cf = (const float *)(wb + NCF_W1_BYTES + NCF_W2_BYTES);
    // The fp16 planes are laid out for |X| <= 1 (what MutualMatching makes of non-negative correlations).  A volume with
    // larger values (negative correlations can do that) is scaled down by the power of two 2^E that brings its largest
    // magnitude below 1; the hidden planes and the two un-scalings follow (all exact).
    const int xeb = (a.xmax[(size_t)blockIdx.z * a.xmax_stride] >> 23) & 0xff;
    const int E = min(max(xeb - 126, 0), 100);
    const float up = __int_as_float((127 + E) << 23), down = __int_as_float((127 - E) << 23);
    const float xscale = 4096.0f * down;
    const float yun = cf[33] * up;
    // relu(acc * s1 + b1) * hscale = relu(acc * (s1 hscale) + b1 hscale): layer 1's epilogue constants per channel, in
    // registers for the whole kernel (they were re-read from LDS for every tile of every strip)
    float e1s[16], e1b[16];
    {
        const float hscale = cf[32] * down;
#pragma unroll
        for (int c = 0; c < 16; ++c) { e1s[c] = cf[c] * up * hscale; e1b[c] = cf[16 + c] * hscale; }
    }

    for (int i = tid; i < 3 * YSLOT; i += NCF_THREADS) Ya[i] = 0.f;
    // The hidden planes start as zeros: the pad slots around the positions and every position outside the volume stay zero
    // for the whole kernel (which positions those are depends on (c, d) only, not on the strip) -- their lanes store
    // to the spare slot HNU instead (HN - HNU = 14 spare slots per channel half).
    for (int i = tid; i < 2 * HPLANE / 16; i += NCF_THREADS) *(nf4 *)(Hs + i * 16) = (nf4){0.f, 0.f, 0.f, 0.f};

    const int a_hi = min(a0 + TA, a.d0);                     // outputs of the tile: [a0, a_hi)
    const int ap_first = max(a0 - 1, 0), ap_last = min(a0 + TA, a.d0 - 1);
    const int bp_first = max(b0 - 1, 0), bp_last = min(b0 + TB, a.d1 - 1);

    // relu(sum + b2) of the finished output slice aout -> Y; every cell read is reset for the slice that re-uses the slot (the
    // pad cells of a row are never read, so they need no reset).  No barrier of its own: it runs at the top of the strip after
    // the slice's last contribution (behind that strip's closing barrier), and the next layer-2 pass that touches the slot
    // again is behind the barrier between this strip's two layers.
    auto flush = [&](int aout) {
        float *slot = Ya + (aout % 3) * YSLOT;
        // 32-bit element offsets (a volume has < 2^31 cells): this lane's cell of the tile's row (b0, c0) of slice aout
        const int ybase = (aout * a.d1 + b0) * (int)nB + c0 * a.d3 + dt0 + lane;
        const bool lane_in = lane < TD && dt0 + lane < a.d3;
        auto row = [&](int r) {
            const int bb = r / TC, ro = r - bb * TC;
            if (lane < TD) {
                float *cell = slot + bb * YROW + ro * P + lane + 1;
                const float v = fmaxf(*cell + a.b2, 0.f);
                *cell = 0.f;
                if (lane_in && b0 + bb < a.d1 && c0 + ro < a.d2) {
                    float *dst = Y + (unsigned)(ybase + bb * (int)nB + ro * a.d3);
                    if (plain) *dst = v; else unsafeAtomicAdd(dst, v);
                }
            }
        };
        if (FIXED) {
#pragma unroll 2
            for (int i = 0; i < 10; ++i) row(wave + NCF_WAVES * i);
        } else {
            for (int r = wave; r < TB * TC; r += NCF_WAVES) row(r);
        }
    };

    // ---- S1: this lane's pair of columns of a staged row (strip-independent)
    const int nbp = bp_last - bp_first + 1, nstrips = (ap_last - ap_first + 1) * nbp;
    const int sB = (int)nB, sA = a.d1 * sB;                  // 32-bit element offsets (a volume has < 2^31 cells)
    // Lanes and items beyond the layout (column pairs >= P / 2, rows >= XROWS, items 18 and 19) are made exact DUPLICATES of the
    // last valid column pair / row: same source, same destination, same value -- so the staging needs no validity branches.
    const int cp = min(lane & 31, P / 2 - 1), rh = lane >> 5;
    const int dcol = dt0 - 2 + 2 * cp;
    const int coff0 = clampi(dcol, 0, a.d3 - 1), coff1 = clampi(dcol + 1, 0, a.d3 - 1);
    const bool okd0 = dcol >= 0 && dcol < a.d3, okd1 = dcol + 1 >= 0 && dcol + 1 < a.d3;
    // item k of this wave = (plane da of the triple (., db), row pair rp); loads are unconditional from clamped addresses, the
    // zero padding is applied when the value is USED (a load in a branch, or a select right behind it, is waited for on
    // the spot).  q0v / q1v are the ONLY load destinations inside the strip loop (a second set -- e.g. for the nine planes
    // of a row start -- makes the compiler wait for every outstanding load, the prefetched ones included, before the first
    // MFMA of a strip).
    float q0v[NCF_KRING], q1v[NCF_KRING];
    // strip-independent parts of this lane's items: source row offset (clamped c row), byte offset of the row pair inside an
    // LDS plane, and the factor of each of its two values: 2^12 / 2^E inside the volume, 0 in the zero padding
    int crow[NCF_KRING], lrow[NCF_KRING];
    float m0v[NCF_KRING], m1v[NCF_KRING];
#pragma unroll
    for (int k = 0; k < NCF_KRING; ++k) {
        const int it = wave + NCF_WAVES * k, i3 = min(it / 6, 2), rp = it - 6 * i3;
        const int xr = min(2 * rp + rh, XROWS - 1), ic = c0 - 2 + xr;
        crow[k] = clampi(ic, 0, a.d2 - 1) * a.d3;
        lrow[k] = (xr * P + 2 * cp) * 2;
        const bool rowok = ic >= 0 && ic < a.d2;
        m0v[k] = (rowok && okd0) ? xscale : 0.f;
        m1v[k] = (rowok && okd1) ? xscale : 0.f;
    }
    // a triple = the three planes (pa + i, pb) (along a) or (pa, pb + i) (along b); plane (ia, ib) sits in slot (ia mod 3,
    // ib mod 3) of the 2-D ring.  Per item (scalars, once per strip): clamped source offset of its plane, inside the volume?,
    // byte offset of the plane's slot.
    int isrc[NCF_KRING], idst[NCF_KRING];
    bool iin[NCF_KRING];
    // (three plane descriptors per strip; item k of wave w belongs to plane min((w + 4 k) / 6, 2): 0, w >= 2, 1, 2, 2)
    const bool k1_second = wave >= 2;
    auto rot3 = [](int m, int i) { return m + i >= 3 ? m + i - 3 : m + i; };      // (m + i) mod 3 for m in 0..2, i in 0..2
    // ma, mb = ring slot (pa mod 3, pb mod 3) of the triple's first plane; the others follow cyclically
    auto triple = [&](int pa, int pb, bool along_a, int ma, int mb) {
        int src3[3], dst3[3];
        bool in3[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int ia = along_a ? pa + i : pa, ib = along_a ? pb : pb + i;
            const int sa = along_a ? rot3(ma, i) : ma, sb = along_a ? mb : rot3(mb, i);
            src3[i] = clampi(ia, 0, a.d0 - 1) * sA + clampi(ib, 0, a.d1 - 1) * sB;
            in3[i] = ia >= 0 && ia < a.d0 && ib >= 0 && ib < a.d1;
            dst3[i] = (sa * 3 + sb) * XROWS * P * 2;
        }
        isrc[0] = src3[0]; iin[0] = in3[0]; idst[0] = dst3[0];
        isrc[1] = k1_second ? src3[1] : src3[0]; iin[1] = k1_second ? in3[1] : in3[0]; idst[1] = k1_second ? dst3[1] : dst3[0];
        isrc[2] = src3[1]; iin[2] = in3[1]; idst[2] = dst3[1];
        isrc[3] = src3[2]; iin[3] = in3[2]; idst[3] = dst3[2];
        isrc[4] = src3[2]; iin[4] = in3[2]; idst[4] = dst3[2];
    };
    static_assert(NCF_KRING == 5 && NCF_WAVES == 4, "the item -> plane table above");
    // the current triple: loads (unconditional, from clamped addresses; the zero padding is selected when the value is used) ...
    auto ring_load = [&]() {
#pragma unroll
        for (int k = 0; k < NCF_KRING; ++k) {
            const int src = isrc[k] + crow[k];                 // (non-negative, < 2^31: unsigned offsets need no sign extension)
            q0v[k] = X[(unsigned)(src + coff0)]; q1v[k] = X[(unsigned)(src + coff1)];
        }
    };
    // ... and their conversion to two fp16 planes of X * 2^12 (adjacent columns packed: one 4-byte store per plane)
    auto ring_store = [&]() {
#pragma unroll
        for (int k = 0; k < NCF_KRING; ++k) {
            const float f = iin[k] ? 1.f : 0.f;                // (scalar) the whole plane lies outside the volume: zeros
            const int dst = idst[k] + lrow[k];
            const float x0 = q0v[k] * m0v[k] * f, x1 = q1v[k] * m1v[k] * f;
            const unsigned h = npk(x0, x1);
            *(unsigned *)(Xs + dst) = h;
            *(unsigned *)(Xs + XPLANE + dst) = npk(x0 - npk_lo(h), x1 - npk_hi(h));
        }
    };

    // ---- S2 (layer 1): D[row <-> (parity s, channel o)][column = position q0 + 2 * (lane & 31) + s], with the rows ordered
    // (pack_nc_fused) so that register r of this lane is channel r of the position of parity s = lane >> 5.
    const int l31 = lane & 31, kb5 = lane >> 5;
    const int nt1 = (HROWS * P + 63) >> 6;
    int s2qh[2] = {0, 0};
    int s2dst[2] = {0, 0};                                    // byte offset of this lane's hidden slot (position + 1), or of the spare slot
    unsigned s2ok = 0;                                        // bit 2 u + s: position q0(u) + 2 l31 + s is a cell of the volume
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = wave + NCF_WAVES * u, q0 = t * 64;
        s2qh[u] = min(q0 + 2 * l31, HROWS * P - 2);           // columns past the strip repeat its last pair (never stored)
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
            const int fl = q0 + 2 * l31 + sp, rowh = fl / P, col = fl - rowh * P;
            const int ic = c0 - 1 + rowh, id = dt0 + col - 1;
            const bool ok = t < nt1 && rowh < HROWS && col <= TD + 1 && ic >= 0 && ic < a.d2 && id >= 0 && id < a.d3;
            s2ok |= (unsigned)ok << (2 * u + sp);
        }
        s2dst[u] = (((s2ok >> (2 * u + kb5)) & 1u) ? q0 + 2 * l31 + kb5 + 1 : HNU) * 16;
    }
    const unsigned char *s2x[2] = {Xs + s2qh[0] * 2, Xs + s2qh[1] * 2};      // this lane's window origin inside a staged plane
    // layer 2: hidden offsets of the lane's K block per step
    const int row16 = lane & 15, kb4 = lane >> 4;
    const int l2da3 = (lane & 15) / 3 % 3, l2da = (lane & 15) < 9 ? (lane & 15) / 3 : 64, l2db = (lane & 15) < 9 ? (lane & 15) % 3 : 64;
    int s3off[5];
#pragma unroll
    for (int st = 0; st < 5; ++st) {
        const int tap = min(2 * st + (kb4 >> 1), 8);          // tap 9 does not exist (zero weights)
        s3off[st] = (kb4 & 1) * HKH + ((tap / 3) * P + tap % 3) * 16;      // hidden position + dc P + dd - 1, slot + 1
    }
    const int nt2 = (TC * P + 15) >> 4;

    // The strips of the tile in SERPENTINE order: a' rows ascending, b' ascending in even rows (global index a') and descending
    // in odd ones, so that EVERY step -- along b' inside a row, along a' at a row's end -- brings exactly one new triple of
    // planes into the 2-D ring (prefetched a phase ahead; a raster order restaged all nine planes at every row start, with
    // their latency exposed).  The direction depends on the global a' only: an output cell sums its nine contributions in an
    // order that does not depend on the tile.
    auto strip_bp = [&](int ap, int j) { return (ap & 1) ? bp_last - j : bp_first + j; };
    __syncthreads();
    NT_DECL
    {
        const int bq = strip_bp(ap_first, 0);
        for (int i = 0; i < 3; ++i) {         // all nine planes of the first strip
            triple(ap_first - 1, bq - 1 + i, true, (ap_first + 2) % 3, (bq + 2 + i) % 3);
            ring_load();
            ring_store();
        }
    }
    __syncthreads();
    NT(0)
    int pending = -1;                         // output slice whose last contribution the previous strip added
    for (int strip = 0, row = 0, j = 0; strip < nstrips; ++strip, (j + 1 < nbp ? ++j : (j = 0, ++row))) {
        const int ap = ap_first + row, bp = strip_bp(ap, j);
        const bool more = strip + 1 < nstrips, same_row = j + 1 < nbp;
        const int dir = (ap & 1) ? -1 : 1;
        // the new triple of the next strip: planes (ap - 1 .. ap + 1, b' + 2 dir) inside a row, (ap + 2, b' - 1 .. b' + 1) at its end
        const int npa = same_row ? ap - 1 : ap + 2, npb = same_row ? bp + 2 * dir : bp - 1;
        if (pending >= 0) flush(pending);
        pending = -1;
        NT(5)
        // ring slots by rotation from the two of this strip: am = (a' - 1) mod 3, bm = (b' - 1) mod 3.  The next triple starts at
        // a' - 1 (same slot as am) or a' + 2 (= a' - 1 mod 3), and at b' + 2 dir or b' - 1
        const int am = (ap + 2) % 3, bm = (bp + 2) % 3;
        triple(npa, npb, same_row, am, same_row ? rot3(bm, dir > 0 ? 0 : 2) : bm);
        if (more) ring_load();                                // in flight during both layers of this strip
        // ---------------- S2: layer 1 -> hidden planes
        {
            // ring slots of the planes a' - 1 .. a' + 1 and b' - 1 .. b' + 1 (element offsets)
            const int sa0 = am * 3 * XROWS * P, sa1 = rot3(am, 1) * 3 * XROWS * P, sa2 = rot3(am, 2) * 3 * XROWS * P;
            const int sl0 = bm * XROWS * P, sl1 = rot3(bm, 1) * XROWS * P, sl2 = rot3(bm, 2) * XROWS * P;
            auto xoff = [&](int gq) {                         // tap group (da, db, dc) -> element offset of its plane row
                const int da = gq / 9, db = (gq / 3) % 3, dc = gq % 3;
                return (da == 0 ? sa0 : da == 1 ? sa1 : sa2) + (db == 0 ? sl0 : db == 1 ? sl1 : sl2) + dc * P;
            };
            // both position tiles of the wave together (independent accumulators, the weights are shared); a wave with one
            // tile multiplies a clamped copy of it -- it would wait at the barrier for the others anyway
            f32x16 acc[2] = {{0}, {0}};
            // the X fragments of slab sl + 1 are read while the six MFMAs of slab sl issue; sched_barrier pins that order (left to
            // itself the compiler re-uses one set of fragment registers and waits for every read right in front of its MFMA:
            // ~4000 of a strip's ~5800 layer-1 cycles were LDS latency)
            nf4 xv[2][2][2];                                   // [buffer][tile][plane]
            auto frags = [&](int sl, nf4 (&d)[2][2]) {
                // the lane's two tap groups of this slab: g = 4 * sl + 2 * kb + {0, 1}; group 27 does not exist (zero weights)
                // (the four offsets are made opaque scalars: otherwise the select over the lane half becomes a lane-indexed lookup
                // in a private array = scratch memory)
                int oa0 = xoff(4 * sl), oa1 = xoff(4 * sl + 1), ob0 = xoff(4 * sl + 2), ob1 = xoff(min(4 * sl + 3, 26));
                P2P_OPAQUE_S(oa0); P2P_OPAQUE_S(oa1); P2P_OPAQUE_S(ob0); P2P_OPAQUE_S(ob1);
                const int o0 = kb5 ? ob0 : oa0, o1 = kb5 ? ob1 : oa1;
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const unsigned *x0 = (const unsigned *)(s2x[u] + p * XPLANE + o0 * 2);      // one shift-add per address
                        const unsigned *x1 = (const unsigned *)(s2x[u] + p * XPLANE + o1 * 2);
                        d[u][p] = (nf4){__uint_as_float(x0[0]), __uint_as_float(x0[1]), __uint_as_float(x1[0]), __uint_as_float(x1[1])};
                    }
            };
            frags(0, xv[0]);
#ifndef NCF_NOPRIO                      // (A/B switch of tools/ab_variants.sh; experiment builds only)
            // layer 1 is the MFMA-dense phase of a strip: its wave outranks the other wave of the SIMD (which is in an epilogue,
            // a flush or layer 2's read-add-writes) for the issue slots; back to normal for the epilogue below (round 5: -2 %)
            __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
            for (int sl = 0; sl < 7; ++sl) {
                nf4 (&c)[2][2] = xv[sl & 1];
                if (sl < 6) frags(sl + 1, xv[(sl + 1) & 1]);
                acc[0] = NCF_MFMA32(w1[sl][0], c[0][1], acc[0]); acc[1] = NCF_MFMA32(w1[sl][0], c[1][1], acc[1]);
                acc[0] = NCF_MFMA32(w1[sl][1], c[0][0], acc[0]); acc[1] = NCF_MFMA32(w1[sl][1], c[1][0], acc[1]);
                acc[0] = NCF_MFMA32(w1[sl][0], c[0][0], acc[0]); acc[1] = NCF_MFMA32(w1[sl][0], c[1][0], acc[1]);
                // order inside the slab: its first MFMA leads, the address arithmetic and the eight reads of the next slab's
                // fragments follow in the shadow of the MFMAs (a matrix instruction occupies its pipe for 32 cycles, the
                // vector ALU is free meanwhile)
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_setprio(0);
            // bias, ReLU, zero outside the volume, scale, split: register r of a lane is channel r of its position
            // q0 + 2 (lane & 31) + (lane >> 5) -> one 16-byte store per channel half and plane
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (wave + NCF_WAVES * u >= nt1) break;               // (wave-uniform) the clamped copy is not stored
                unsigned char *dst = Hs + s2dst[u];                  // (a position outside the volume: the spare slot)
#pragma unroll
                for (int kh = 0; kh < 2; ++kh) {
                    float h[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const nf2 av = {acc[u][8 * kh + 2 * j], acc[u][8 * kh + 2 * j + 1]};
                        const nf2 sv = {e1s[8 * kh + 2 * j], e1s[8 * kh + 2 * j + 1]}, bv = {e1b[8 * kh + 2 * j], e1b[8 * kh + 2 * j + 1]};
                        const nf2 hv = __builtin_elementwise_fma(av, sv, bv);
                        h[2 * j] = fmaxf(hv[0], 0.f); h[2 * j + 1] = fmaxf(hv[1], 0.f);
                    }
                    unsigned p0[4], p1[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        p0[j] = npk(h[2 * j], h[2 * j + 1]);
                        p1[j] = npk(h[2 * j] - npk_lo(p0[j]), h[2 * j + 1] - npk_hi(p0[j]));
                    }
                    *(nf4 *)(dst + kh * HKH) = (nf4){__uint_as_float(p0[0]), __uint_as_float(p0[1]), __uint_as_float(p0[2]), __uint_as_float(p0[3])};
                    *(nf4 *)(dst + kh * HKH + HPLANE) = (nf4){__uint_as_float(p1[0]), __uint_as_float(p1[1]), __uint_as_float(p1[2]), __uint_as_float(p1[3])};
                }
            }
        }
        NT(1)
        __syncthreads();
        NT(2)
        // ---------------- S3: layer 2, contributions of the strip to the 3 x 3 output planes around it
        {
            // lane n = lane & 15 adds into output plane (a' - da + 1, b' - db + 1), (da, db) = (n / 3, n % 3) (l2da, l2db: per-lane
            // constants, 64 for the unused columns n >= 9); accumulator slot (a' + 4 - da) mod 3 from the scalar (a' + 4) mod 3
            const int aout = ap + 1 - l2da, bout = bp + 1 - l2db;
            const bool lane_ok = aout >= a0 && aout < a_hi && bout >= b0 && bout < min(b0 + TB, a.d1);
            const int apm = rot3(am, 2), sl3 = apm - l2da3;
            float *ydst = Ya + (sl3 < 0 ? sl3 + 3 : sl3) * YSLOT + (bout - b0) * YROW;
            // two m-tiles at a time (independent accumulators), the fragments of step st + 1 in flight during the MFMAs of step st
            const bool yalign = (YROW & 3) == 0;               // every (slot, plane) row of the accumulators starts 16-byte aligned
            if (FIXED) {
                // TC * P = 352 = 22 whole m-tiles: wave w takes the pairs (w + 8 i, w + 8 i + 4), i = 0 .. 2; the second tile of
                // the last pair does not exist for waves 2 and 3 (they multiply the first one twice and drop the copy)
                const unsigned char *hl = Hs + row16 * 16;
                float *yl = ydst + 4 * kb4;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int qa = (wave + 8 * i) * 16;
                    const bool has_b = i < 2 || wave < 2;
                    const int qb = has_b ? qa + 64 : qa;
                    const unsigned char *ha = hl + qa * 16, *hb = hl + qb * 16;
                    nf4 accA = {0.f, 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f};
                    nf4 fr[2][4];
                    auto frags3 = [&](int st, nf4 (&d)[4]) {
                        d[0] = *(const nf4 *)(ha + s3off[st]); d[1] = *(const nf4 *)(ha + s3off[st] + HPLANE);
                        d[2] = *(const nf4 *)(hb + s3off[st]); d[3] = *(const nf4 *)(hb + s3off[st] + HPLANE);
                    };
                    frags3(0, fr[0]);
#pragma unroll
                    for (int st = 0; st < 5; ++st) {
                        nf4 (&c)[4] = fr[st & 1];
                        if (st < 4) frags3(st + 1, fr[(st + 1) & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                        accA = NCF_MFMA16(c[1], w2[st][0], accA); accB = NCF_MFMA16(c[3], w2[st][0], accB);
                        accA = NCF_MFMA16(c[0], w2[st][1], accA); accB = NCF_MFMA16(c[2], w2[st][1], accB);
                        accA = NCF_MFMA16(c[0], w2[st][0], accA); accB = NCF_MFMA16(c[2], w2[st][0], accB);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (lane_ok) {
                        nf4 va = *(nf4 *)(yl + qa);
                        va += accA * yun;
                        *(nf4 *)(yl + qa) = va;
                        if (has_b) {
                            nf4 vb = *(nf4 *)(yl + qb);
                            vb += accB * yun;
                            *(nf4 *)(yl + qb) = vb;
                        }
                    }
                }
            } else
            for (int t = wave; t < nt2; t += 2 * NCF_WAVES) {
                const int qa = t * 16, qb = qa + NCF_WAVES * 16;
                const unsigned char *ha = Hs + min(qa + row16, TC * P - 1) * 16, *hb = Hs + min(qb + row16, TC * P - 1) * 16;
                nf4 accA = {0.f, 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f};
                // the fragments of step st + 1 are read while the six MFMAs of step st issue (order pinned by sched_barrier)
                nf4 fr[2][4];                                  // [buffer][a plane 0, a plane 1, b plane 0, b plane 1]
                auto frags3 = [&](int st, nf4 (&d)[4]) {
                    d[0] = *(const nf4 *)(ha + s3off[st]); d[1] = *(const nf4 *)(ha + s3off[st] + HPLANE);
                    d[2] = *(const nf4 *)(hb + s3off[st]); d[3] = *(const nf4 *)(hb + s3off[st] + HPLANE);
                };
                frags3(0, fr[0]);
#pragma unroll
                for (int st = 0; st < 5; ++st) {
                    nf4 (&c)[4] = fr[st & 1];
                    if (st < 4) frags3(st + 1, fr[(st + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    accA = NCF_MFMA16(c[1], w2[st][0], accA); accB = NCF_MFMA16(c[3], w2[st][0], accB);
                    accA = NCF_MFMA16(c[0], w2[st][1], accA); accB = NCF_MFMA16(c[2], w2[st][1], accB);
                    accA = NCF_MFMA16(c[0], w2[st][0], accA); accB = NCF_MFMA16(c[2], w2[st][0], accB);
                    __builtin_amdgcn_sched_barrier(0);
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
        NT(3)
        // ---------------- S1 of the next strip, behind this strip's layer 2 (layer 1 of this strip is done with the planes)
        if (more) ring_store();
        NT(4)
        if (!same_row && ap - 1 >= a0) pending = ap - 1;       // output slice ap - 1 is complete once hidden slice ap is done
        __syncthreads();
        NT(6)
    }
    if (pending >= 0) flush(pending);
    if (nstrips > 0 && ap_last < a_hi && ap_last >= a0) flush(ap_last);      // the volume ends inside the tile: its last slice has no slice above
#ifdef NCF_TIMING
    if (blockIdx.x == 5 && blockIdx.y == 0 && lane == 0) {       // (overwrites a few output cells: timing builds only)
        __syncthreads();
        for (int i = 0; i < 8; ++i) Y[wave * 8 + i] = (float)nt_[i];
        Y[32] = (float)nstrips;
    }
#endif
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
                    // A row n -> (parity s, channel o) such that the D registers of a lane (rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5))
                    // are the 16 channels r of ONE position of parity lane >> 5
                    const int n = lane & 31, s = (n >> 2) & 1, o = (n & 3) + 4 * (n >> 3), dd = e - s;
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
    return (size_t)2 * ((9 * (tc + 4) * P * 2 + 8 + 15) & ~15) + (size_t)2 * 2 * nc_hidden_slots(tc, P) * 16 + (size_t)3 * tb * nc_yrow(tc, P) * 4 + 256;
}

// float bits of max |x| over n values per pair -> out[pair * out_stride] (zero beforehand); one atomic per work-group, <= 256
// work-groups per pair with four 16-byte loads in flight per thread (the per-wave probes of thousands of small work-groups
// queued on the one cache line that holds a batch's results)
__global__ __launch_bounds__(256) void absmax_kernel(const float *__restrict__ x, size_t n, size_t stride, int *out, size_t out_stride) {
    __shared__ float part[4];
    x += (size_t)blockIdx.z * stride;
    float m = 0.f;
    const size_t n4 = (((size_t)x & 15) == 0) ? n >> 2 : 0;          // 16-byte loads where the item starts aligned
    const size_t step = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * step < n4; i += 4 * step) {
        const nf4 v0 = ((const nf4 *)x)[i], v1 = ((const nf4 *)x)[i + step], v2 = ((const nf4 *)x)[i + 2 * step], v3 = ((const nf4 *)x)[i + 3 * step];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v0[0]), fabsf(v0[1]))), fmaxf(fabsf(v0[2]), fabsf(v0[3])));
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v1[0]), fabsf(v1[1]))), fmaxf(fabsf(v1[2]), fabsf(v1[3])));
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v2[0]), fabsf(v2[1]))), fmaxf(fabsf(v2[2]), fabsf(v2[3])));
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v3[0]), fabsf(v3[1]))), fmaxf(fabsf(v3[2]), fabsf(v3[3])));
    }
    for (; i < n4; i += step) {
        const nf4 v = ((const nf4 *)x)[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    for (size_t j = 4 * n4 + (size_t)blockIdx.x * 256 + threadIdx.x; j < n; j += step) m = fmaxf(m, fabsf(x[j]));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float mm = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
        int *dst = out + (size_t)blockIdx.z * out_stride;
        if (__float_as_int(mm) > *(volatile int *)dst) atomicMax(dst, __float_as_int(mm));
    }
}

int launch_absmax(const float *x, size_t n, size_t stride, int pairs, int *out, size_t out_stride, hipStream_t stream) {
    const unsigned blocks = (unsigned)std::min<size_t>((n + 16383) / 16384, 256);
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
    // the flush computes output offsets in 32 bits
    P2P_REQUIRE((unsigned long long)d0 * d1 * d2 * d3 < (1ull << 31), P2P_EUNSUPPORTED,
                "consensus volume %d x %d x %d x %d has 2^31 cells or more", d0, d1, d2, d3);
    NcFusedArgs a{};
    a.X = X; a.Y = Y; a.Y2 = Y2; a.stride = stride; a.d0 = d0; a.d1 = d1; a.d2 = d2; a.d3 = d3; a.w = w_dev; a.b2 = b2;
    a.xmax = xmax; a.xmax_stride = xmax_stride;
    nc_fused_tile(a, pairs, forced_tile);
    const size_t lds = nc_fused_lds_bytes(a.tb, a.tc, a.P);
    P2P_REQUIRE(a.P <= 64 && lds <= 160 * 1024 && a.tc + 4 <= 12 && (a.tc + 2) * a.P <= 512, P2P_EUNSUPPORTED,
                "consensus tile does not fit (P %d, tc %d, LDS %zu)", a.P, a.tc, lds);
    int dev = 0;
    P2P_HIP_CHECK(hipGetDevice(&dev));
    static DeviceOnce attr_set;
    if (!attr_set.done(dev)) {
        P2P_HIP_CHECK(hipFuncSetAttribute((const void *)nc_fused_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        P2P_HIP_CHECK(hipFuncSetAttribute((const void *)nc_fused_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set.set(dev);
    }
    if (a.tb == 5 && a.tc == 8 && a.td == 40 && a.P == 44)
        hipLaunchKernelGGL(nc_fused_kernel<true>, dim3(a.na * a.nb * a.nc * a.nd, 2, pairs), dim3(NCF_THREADS), lds, stream, a);
    else
        hipLaunchKernelGGL(nc_fused_kernel<false>, dim3(a.na * a.nb * a.nc * a.nd, 2, pairs), dim3(NCF_THREADS), lds, stream, a);
    return check_launch("nc_fused_kernel");
}

}  // namespace p2p
