// Coarse stage of the Patch2Pix matching path on gfx950 (reference networks/patch2pix.py:120-136,
// :340-375): L2-normalise -> 4-D correlation -> 4-D max-pool -> mutual matching -> neighbourhood
// consensus (two 3^4 convolutions, symmetric) -> mutual matching -> soft mutual-NN matches.
//
// Data layout in HBM (A = image 1 cells, B = image 2 cells):
//   Fn      [row block of 128 pos'][K chunk of 32 ch][plane 2][row 128][4 pieces of 8 ch] fp16: L2-normalised layer-3
//                         features x 2^12 as two fp16 planes, in the 16 KB blocks the correlation GEMM copies into LDS by
//                         LDS-DMA (piece q of row r at slot q ^ ((r >> 2) & 3): conflict-free fragment reads); positions
//                         re-ordered cell-major so that the k^2 positions of one pooling cell are adjacent:
//                         pos' = cell*k^2 + (i%k)*k + (j%k)
//   P, Y, Y2 [nA'][nB']   fp32 pooled correlation volume viewed as a matrix (row = A cell, col = B cell); Y / Y2 =
//                         the two branches of the consensus net (consensus.hip), summed by the kernels that read them
//   delta   [nA'][nB']    uint8 argmax code s = ((di*k+dj)*k+dk)*k+dl
// The full-resolution correlation (92 MB at 480x640, 1.5 GB at 960x1280) is never written: the
// pooling runs on the MFMA accumulators of the correlation GEMM; the 16-channel hidden volume of the
// consensus net never leaves LDS.  ONE arithmetic whatever the batch (fp32-equivalent fp16x2 on the matrix
// cores, fp32 elsewhere): a pair's results do not depend on the pairs it shares a launch with.
//
// Batches: the reference's tensors carry a batch axis ([B,C,h,w] features of B equally sized pairs).  Every
// kernel takes the pair from blockIdx.z and a per-pair stride for each of its pointers, so a batch is
// ONE launch per kernel (B x the work-groups: the 30x40x30x40 volume of a single 480x640 pair does not
// fill 256 CUs in the consensus layers).
#include "p2p_common.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace p2p {

constexpr float MM_EPS = 1e-5f;

// monotone float <-> int key so that atomicMax(int) implements a float max (order independent)
__device__ __forceinline__ int f2key(float f) {
    int b = __float_as_int(f);
    return b >= 0 ? b : b ^ 0x7fffffff;
}
__device__ __forceinline__ float key2f(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7fffffff); }
constexpr int KEY_NEG_INF = (int)0xff800000 ^ 0x7fffffff;

// ------------------------------------------------------------------------------------------------
// 1. L2 normalise + transpose to position-major with cell-major position order (modules.py:6)
// ------------------------------------------------------------------------------------------------
constexpr int PREP_P = 16;    // positions per work-group (300 groups at 60x80: fills the chip)
// Writes the normalised features x 2^12 as two fp16 planes (|v| <= 1, so both planes stay in the normal range of fp16 and
// v * 2^12 = p0 + p1 to within 2^-24 |v * 2^12|) in the block layout of the correlation GEMM (CX_BLK, below).
constexpr float CORR_FP16_SCALE = 4096.0f;
// One launch for both images of every pair (blockIdx.y = image); the same launch resets the maxima keys of the mutual
// matchings (nkeys words = -inf, then one word = 0: the float bits of max |X|, see mm_apply_kernel).
struct PrepArgs {
    const float *F[2];
    unsigned short *Fn[2];
    int h[2], w[2];
    size_t sF[2];
    int C, k;
    size_t sFn;
    int *keys;
    int nkeys;
    size_t sKeys;
};
__global__ __launch_bounds__(256) void prep_kernel(PrepArgs a) {
    const int img = blockIdx.y;
    if (img == 0) {
        const int i = blockIdx.x * 256 + threadIdx.x;
        if (i <= a.nkeys) a.keys[blockIdx.z * a.sKeys + i] = (i < a.nkeys) ? KEY_NEG_INF : 0;
    }
    const int h = a.h[img], w = a.w[img], C = a.C, k = a.k;
    if ((int)blockIdx.x * PREP_P >= h * w) return;
    const float *__restrict__ F = a.F[img] + blockIdx.z * a.sF[img];
    unsigned short *__restrict__ Fn = a.Fn[img] + blockIdx.z * a.sFn;
    // The GEMM copies whole 128-row blocks: the rows between h * w and the next multiple of 128 are zeros, never uninitialised
    // memory (its stores are masked, so their products were never used -- but NaN bit patterns went through the matrix cores).
    // The work-group of the last positions writes them: <= 127 rows x C channels x 2 planes, 16 bytes per store.
    if (((int)blockIdx.x + 1) * PREP_P >= h * w) {
        const int pad0 = h * w, pad1 = (pad0 + 127) & ~127, pieces = (C >> 5) * 2 * 4;      // 16-byte pieces per row
        for (int e = threadIdx.x; e < (pad1 - pad0) * pieces; e += 256) {
            const int r = (pad0 + e / pieces) & 127, q = e % pieces, kc = q >> 3, pl = (q >> 2) & 1, piece = q & 3;
            unsigned short *d = Fn + ((size_t)(pad0 >> 7) * (C >> 5) + kc) * (2 * 128 * 32) + pl * (128 * 32) + (r * 4 + piece) * 8;
            *(f32x4 *)d = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    __shared__ float tile[256 * (PREP_P + 1)];   // [C <= 256][17]
    __shared__ float part[16][PREP_P];
    __shared__ float inv[PREP_P];
    const int hw = h * w;
    const int p0 = blockIdx.x * PREP_P;
    const int tid = threadIdx.x;
    // 4 threads x float4 cover the 16 positions of one channel row (64 contiguous bytes)
    for (int e = tid; e < C * 4; e += 256) {
        const int c = e >> 2, q = e & 3;
        const int pos = p0 + 4 * q;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (pos + 3 < hw && (hw & 3) == 0) v = *(const f32x4 *)(F + (size_t)c * hw + pos);
        else
            for (int i = 0; i < 4; ++i) if (pos + i < hw) v[i] = F[(size_t)c * hw + pos + i];
        float *t = tile + c * (PREP_P + 1) + 4 * q;
        t[0] = v[0]; t[1] = v[1]; t[2] = v[2]; t[3] = v[3];
    }
    __syncthreads();
    {   // sum of squares: 16 channel groups x 16 positions, ascending channel order within a group
        const int p = tid & 15, g = tid >> 4;
        float ss = 0.f;
        for (int c = g; c < C; c += 16) { const float v = tile[c * (PREP_P + 1) + p]; ss = fmaf(v, v, ss); }
        part[g][p] = ss;
    }
    __syncthreads();
    if (tid < PREP_P) {
        float ss = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) ss += part[g][tid];
        inv[tid] = 1.0f / sqrtf(ss + 1e-6f);
    }
    __syncthreads();
    const int wc = w / k;
    for (int p = 0; p < PREP_P; ++p) {
        const int pos = p0 + p;
        if (pos >= hw) break;
        const int i = pos / w, j = pos - i * w;
        const int pp = ((i / k) * wc + (j / k)) * (k * k) + (i % k) * k + (j % k);
        for (int c = tid; c < C; c += 256) {
            const float x = tile[c * (PREP_P + 1) + p] * inv[p] * CORR_FP16_SCALE;
            // block (pp / 128, c / 32): [plane][row pp % 128][piece (c / 8) % 4 at slot piece ^ ((row / 4) % 4)][c % 8]
            const int r = pp & 127;
            unsigned short *d = Fn + ((size_t)(pp >> 7) * (C >> 5) + (c >> 5)) * (2 * 128 * 32) +
                                (r * 4 + (((c >> 3) & 3) ^ ((r >> 2) & 3))) * 8 + (c & 7);
            const _Float16 h0 = (_Float16)x;
            d[0] = __builtin_bit_cast(unsigned short, h0);
            d[128 * 32] = __builtin_bit_cast(unsigned short, (_Float16)(x - (float)h0));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 2. correlation GEMM (modules.py:41-53) with the 4-D max-pool (modules.py:11-34) in the epilogue
//    C[pA'][pB'] = sum_c FnA[pA'][c] * FnB[pB'][c];  128x128 tile, 4 waves x (2x2) tiles of 32x32
// ------------------------------------------------------------------------------------------------
// fp32-equivalent arithmetic on the fp16 matrix cores: both operands arrive as two fp16 planes of the features times 2^12
// (prep_kernel), three v_mfma_f32_32x32x16_f16 per product (a0 b0 + a0 b1 + a1 b0; the dropped a1 b1 is <= 2^-24 of the
// product), the accumulators are scaled back by 2^-24 (exact) in the epilogue.  fp32 accumulation, no VALU in the loop: per
// K = 16 slab and wave 8 ds_read_b128 and 12 MFMAs.
// Staging (round 5): the operands of a K = 32 stage are two 16 KB blocks that prep_kernel wrote as the LDS image, copied by
// LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write_b128 -- the store path of the register-staged version
// took more LDS cycles than the fragment reads) into a double buffer: the DMA of stage i + 1 is in flight while stage i is
// multiplied, one vmcnt(0) + raw s_barrier per stage; the fragments of slab s + 1 are read behind the MFMAs of slab s.
// LDS: [stage 2][A|B][plane 2][128 rows][4 pieces of 16 B, XOR-swizzled] = 64 KB, two work-groups per compute unit.
//
// Tile order.  The hardware deals consecutive work-group ids to the 8 XCDs round-robin, and every XCD has its own 4 MB L2.
// A row-major raster therefore makes every XCD stream ALL B panels for every row of tiles (960x1280: 3.1 GB fetched per
// pair for 39 MB of operands).  Here work-group id g belongs to XCD g % 8, and XCD x walks the x-th eighth of a
// "block-major" order: blocks of CX_AB tile rows (their A panels, CX_AB x 128 KB, stay in the L2), inside a block column
// after column of B tiles -- each B panel is fetched once per block instead of once per tile row.
constexpr int CT = 128;      // tile edge
constexpr int CX_AB = 16;    // tile rows per L2-resident block of A panels (16 x 128 KB = 2 MB of the 4 MB L2; measured at
                             // 960x1280: 16 rows fetch 224 MB per pair, 24 rows 246 MB, 28 rows 682 MB -- the streaming B
                             // panels and the output need the other half)
typedef _Float16 cf16x8 __attribute__((ext_vector_type(8)));
constexpr int CX_BLK = 2 * CT * 64;          // one operand block of a stage: [plane 2][row 128][32 K fp16] = 16 KB
constexpr int CX_STAGE = 2 * CX_BLK, CX_LDS = 2 * CX_STAGE;
__device__ __forceinline__ f32x16 cx_mfma(const f32x4 &a, const f32x4 &b, const f32x16 &c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(cf16x8, a), __builtin_bit_cast(cf16x8, b), c, 0, 0, 0);
}
// stage IT of the tile -> ring slot SLOT: wave w copies bytes [4096 w, 4096 w + 4096) of the A and of the B block
#define CX_ISSUE(IT, SLOT)                                                                                             \
    {                                                                                                                  \
        const unsigned char *ga_ = Ab + (size_t)(IT) * CX_BLK + lane16;                                                \
        const unsigned char *gb_ = Bb + (size_t)(IT) * CX_BLK + lane16;                                                \
        unsigned char *la_ = cx + (SLOT) * CX_STAGE + wave * 4096;                                                     \
        P2P_GLOBAL_LOAD_LDS16(ga_, la_, 0); P2P_GLOBAL_LOAD_LDS16(ga_, la_, 1024);                                     \
        P2P_GLOBAL_LOAD_LDS16(ga_, la_, 2048); P2P_GLOBAL_LOAD_LDS16(ga_, la_, 3072);                                  \
        P2P_GLOBAL_LOAD_LDS16(gb_, la_ + CX_BLK, 0); P2P_GLOBAL_LOAD_LDS16(gb_, la_ + CX_BLK, 1024);                   \
        P2P_GLOBAL_LOAD_LDS16(gb_, la_ + CX_BLK, 2048); P2P_GLOBAL_LOAD_LDS16(gb_, la_ + CX_BLK, 3072);                \
    }
// fragments of slab S of ring slot SLOT: a[m-tile][plane], b[n-tile][plane]
#define CX_READ(FA, FB, SLOT, S)                                                                                       \
    {                                                                                                                  \
        const unsigned char *pa_ = cx + (SLOT) * CX_STAGE + ((S) ? aoff1 : aoff0);                                     \
        const unsigned char *pb_ = cx + (SLOT) * CX_STAGE + CX_BLK + ((S) ? boff1 : boff0);                            \
        _Pragma("unroll") for (int t_ = 0; t_ < 2; ++t_)                                                               \
            _Pragma("unroll") for (int q_ = 0; q_ < 2; ++q_) {                                                         \
                FA[t_][q_] = *(const f32x4 *)(pa_ + t_ * 2048 + q_ * 8192);                                            \
                FB[t_][q_] = *(const f32x4 *)(pb_ + t_ * 2048 + q_ * 8192);                                            \
            }                                                                                                          \
    }
// smallest terms first (a1 b0, a0 b1, a0 b0); the four accumulators rotate
#define CX_SLAB(FA, FB)                                                                                                \
    _Pragma("unroll") for (int t_ = 0; t_ < 3; ++t_) {                                                                 \
        const int pa_ = (t_ == 0) ? 1 : 0, pb_ = (t_ == 1) ? 1 : 0;                                                    \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                                               \
            _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) acc[i_][j_] = cx_mfma(FA[i_][pa_], FB[j_][pb_], acc[i_][j_]); \
    }
// one LDS read behind each of the first eight MFMAs of a slab (left alone the compiler builds read -> wait -> MFMA chains)
#define CX_PIPE()                                                                                                      \
    _Pragma("unroll") for (int g_ = 0; g_ < 8; ++g_) {                                                                 \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }        \
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_barrier(0);

template <int KS>
__global__ __launch_bounds__(256, 2) void corr_pool_kernel(const unsigned short *__restrict__ A, const unsigned short *__restrict__ B,
                                                           int nA, int nB, int C, float *__restrict__ P,
                                                           uint8_t *__restrict__ delta, size_t sAB, size_t sP, size_t sDelta,
                                                           int gx, int gy, int ntiles, int per_xcd) {
    // Persistent work-groups: XCD x = blockIdx.x % 8 owns the tiles [x * per_xcd, (x + 1) * per_xcd) of the order described
    // under "Tile order", its work-group j = blockIdx.x / 8 walks j, j + G, j + 2G, ... of them (G work-groups per XCD);
    // stage 0 of the next tile is in flight during the pooling epilogue of the current one.
    const int Gx = gridDim.x >> 3, xcd = blockIdx.x & 7;
    int tj = blockIdx.x >> 3;
    // tile number -> (pair, A tile row, B tile column)
    int rowA0 = 0, rowB0 = 0, zp = 0;
    const unsigned char *Ab = nullptr, *Bb = nullptr;             // the tile's first blocks: stage it = block it of the row block
    auto locate = [&](int j, int &ra, int &rb, int &z, const unsigned char *&pa, const unsigned char *&pb) -> bool {
        const int L = xcd * per_xcd + j;
        if (j >= per_xcd || L >= ntiles) return false;
        const int tp = gx * gy;
        z = L / tp;
        const int r = L - z * tp;
        const int blk = r / (CX_AB * gx), rr = r - blk * (CX_AB * gx);
        const int hb = min(CX_AB, gy - blk * CX_AB);
        const int bcol = rr / hb, arow = blk * CX_AB + rr - bcol * hb;
        ra = arow * CT; rb = bcol * CT;
        pa = (const unsigned char *)(A + (size_t)z * sAB) + (size_t)arow * (C >> 5) * CX_BLK;
        pb = (const unsigned char *)(B + (size_t)z * sAB) + (size_t)bcol * (C >> 5) * CX_BLK;
        return true;
    };
    if (!locate(tj, rowA0, rowB0, zp, Ab, Bb)) return;
    P2P_DYN_SHARED(unsigned char, cx);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;
    const unsigned lane16 = wave * 4096 + lane * 16;
    // fragment addresses inside a block: row r, piece (2 slab + half) ^ ((r >> 2) & 3)
    const int sw = (l31 >> 2) & 3;
    const unsigned aoff0 = ((wr * 64 + l31) * 4 + ((0 + half) ^ sw)) * 16, aoff1 = ((wr * 64 + l31) * 4 + ((2 + half) ^ sw)) * 16;
    const unsigned boff0 = ((wc * 64 + l31) * 4 + ((0 + half) ^ sw)) * 16, boff1 = ((wc * 64 + l31) * 4 + ((2 + half) ^ sw)) * 16;

    const int nk = C / 32;
    CX_ISSUE(0, 0)
#pragma unroll 1
  for (;;) {
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16){0};

    f32x4 xa[2][2], xb[2][2], ya[2][2], yb[2][2];
#pragma unroll 1
    for (int it = 0; it < nk; it += 2) {
        // even stage (slot 0): its pieces have landed for everybody behind the barrier, and everybody is done with slot 1
        P2P_WAIT_VMCNT(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (it + 1 < nk) CX_ISSUE(it + 1, 1)
        CX_READ(xa, xb, 0, 0)
        __builtin_amdgcn_sched_barrier(0);
        CX_READ(ya, yb, 0, 1)
        CX_SLAB(xa, xb)
        CX_PIPE()
        CX_SLAB(ya, yb)
        __builtin_amdgcn_sched_barrier(0);
        if (it + 1 >= nk) break;
        // odd stage (slot 1)
        P2P_WAIT_VMCNT(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (it + 2 < nk) CX_ISSUE(it + 2, 0)
        CX_READ(xa, xb, 1, 0)
        __builtin_amdgcn_sched_barrier(0);
        CX_READ(ya, yb, 1, 1)
        CX_SLAB(xa, xb)
        CX_PIPE()
        CX_SLAB(ya, yb)
        __builtin_amdgcn_sched_barrier(0);
    }
    // the next tile of this work-group: its stage 0 goes to slot 0 behind a barrier (everybody is done reading the ring)
    const int rowA0c = rowA0, rowB0c = rowB0;
    float *Pz = P + (size_t)zp * sP;
    uint8_t *dz = delta ? delta + (size_t)zp * sDelta : nullptr;
    tj += Gx;
    const bool more = locate(tj, rowA0, rowB0, zp, Ab, Bb);
    __builtin_amdgcn_s_barrier();
    if (more) CX_ISSUE(0, 0)
    __builtin_amdgcn_sched_barrier(0);
    // the planes carried 2^12 each: the accumulators hold 2^24 x the correlation (undone exactly at the stores; a maximum
    // commutes with the positive power of two)
    constexpr float inv = 1.0f / (CORR_FP16_SCALE * CORR_FP16_SCALE);

    // accumulator element r of lane: row = (r&3) + 8*(r>>2) + 4*half, col = l31
    if (KS == 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int tr = rowA0c + wr * 64 + i * 32, tc = rowB0c + wc * 64 + j * 32;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = tr + (r & 3) + 8 * (r >> 2) + 4 * half, col = tc + l31;
                    if (row < nA && col < nB) Pz[(size_t)row * nB + col] = acc[i][j][r] * inv;
                }
            }
    } else
    // k = 2: a pooling cell is a 4x4 block: rows = (i,j) of A in regs 4g..4g+3, cols = (k,l) of B in 4 adjacent lanes.
    // First maximum in the order s = row_in_cell*4 + col_in_cell.  The exchange inside a quad of lanes is two DPP moves per
    // value (quad_perm; __shfl_xor goes through the LDS crossbar and is waited for: 64 of them were a third of a tile's
    // time), the pooled offsets are 32-bit (a pooled volume has < 2^31 cells), the bounds of a tile are checked once.
    {
        const int nAc = nA >> 2, nBc = nB >> 2;
        const int crow0 = ((rowA0c + wr * 64) >> 2) + half, ccol0 = (rowB0c + wc * 64 + l31) >> 2;
        const bool writer = (lane & 3) == 0;
        const bool full = ((rowA0c + CT) >> 2) <= nAc && ((rowB0c + CT) >> 2) <= nBc;      // (wave-uniform) no cell of the tile is outside
        // all sixteen cells of the lane's quad first (the exchanges need every lane), then ONE masked block of stores
        float pbest[2][2][4];
        int ps[2][2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float best = acc[i][j][4 * g];
                    int s = 0;
#pragma unroll
                    for (int r = 1; r < 4; ++r) {
                        const float v = acc[i][j][4 * g + r];
                        if (v > best) { best = v; s = r; }
                    }
                    s = s * 4 + (lane & 3);
                    {
                        const float ov = __uint_as_float(P2P_SWAP_ADJACENT(__float_as_uint(best)));
                        const int os = (int)P2P_SWAP_ADJACENT((unsigned)s);
                        if (ov > best || (ov == best && os < s)) { best = ov; s = os; }
                    }
                    {
                        const float ov = __uint_as_float(P2P_SWAP_PAIRS(__float_as_uint(best)));
                        const int os = (int)P2P_SWAP_PAIRS((unsigned)s);
                        if (ov > best || (ov == best && os < s)) { best = ov; s = os; }
                    }
                    pbest[i][j][g] = best * inv;
                    ps[i][j][g] = s;
                }
        if (writer) {
            float *Pl = Pz + (unsigned)(crow0 * nBc + ccol0);
            uint8_t *Dl = dz ? dz + (unsigned)(crow0 * nBc + ccol0) : nullptr;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int dr = 8 * i + 2 * g, dc = 8 * j;
                        if (full || (crow0 + dr < nAc && ccol0 + dc < nBc)) {
                            const unsigned at = (unsigned)(dr * nBc + dc);
                            Pl[at] = pbest[i][j][g];
                            if (Dl) Dl[at] = (uint8_t)ps[i][j][g];
                        }
                    }
        }
    }
    if (!more) break;
  }
}

// ------------------------------------------------------------------------------------------------
// 3. row / column maxima of an [nA][nB] matrix (the two torch.max of ncn/model.py:165-166)
// ------------------------------------------------------------------------------------------------
// One launch for both: the first gridDim.x - ceil(nA / 4) work-groups take the column maxima (a thread owns one column
// over a 64-row chunk; chunks meet in an order independent atomicMax), the others the row maxima (one wave per row).
// (X2: optional second addend of every value -- the fused consensus kernel leaves its two branches in two arrays)
__global__ __launch_bounds__(256) void maxima_kernel(const float *__restrict__ X, int nA, int nB, int *rkey, int *ckey, size_t sX,
                                                     size_t sKeys, const float *__restrict__ X2, int col_groups) {
    X += blockIdx.z * sX;
    if (X2) X2 += blockIdx.z * sX;
    rkey += blockIdx.z * sKeys;
    ckey += blockIdx.z * sKeys;
    const int gcol = (nB + 255) / 256;
    if ((int)blockIdx.x < col_groups) {
        const int col = ((int)blockIdx.x % gcol) * 256 + threadIdx.x;
        if (col >= nB) return;
        const int r0 = ((int)blockIdx.x / gcol) * 64, r1 = min(r0 + 64, nA);
        float cm = -INFINITY;
        if (X2) {
#pragma unroll 8
            for (int r = r0; r < r1; ++r) cm = fmaxf(cm, X[(size_t)r * nB + col] + X2[(size_t)r * nB + col]);
        } else {
#pragma unroll 8
            for (int r = r0; r < r1; ++r) cm = fmaxf(cm, X[(size_t)r * nB + col]);
        }
        atomicMax(&ckey[col], f2key(cm));
        return;
    }
    const int row = ((int)blockIdx.x - col_groups) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= nA) return;
    const float *x = X + (size_t)row * nB;
    float rm = -INFINITY;
    if (X2) {
        const float *x2 = X2 + (size_t)row * nB;
        for (int c = lane; c < nB; c += 64) rm = fmaxf(rm, x[c] + x2[c]);
    } else {
        for (int c = lane; c < nB; c += 64) rm = fmaxf(rm, x[c]);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) rm = fmaxf(rm, __shfl_xor(rm, m));
    if (lane == 0) rkey[row] = f2key(rm);
}

// MutualMatching value (ncn/model.py:168-175): x * ((x / (max_over_B + eps)) * (x / (max_over_A + eps)))
__device__ __forceinline__ float mm_value(float x, float max_over_b, float max_over_a) {
    const float xa = x / (max_over_b + MM_EPS);
    const float xb = x / (max_over_a + MM_EPS);
    return x * (xa * xb);
}

// (in place when out == X).  VEC: a thread owns four adjacent columns of a row (one 16-byte load and store per array); the
// scalar form is for volumes whose rows are not a multiple of four cells.  The divisions stay IEEE divisions (the values
// must round like the reference's); the index split is one 32-bit division per thread.
typedef int mm_i32x4 __attribute__((ext_vector_type(4)));
template <bool VEC>
__global__ __launch_bounds__(256) void mm_apply_kernel(const float *X, int nA, int nB, const int *__restrict__ rkey,
                                                       const int *__restrict__ ckey, float *out, size_t sX, size_t sKeys,
                                                       size_t sOut, int *amax, const float *X2) {
    X += blockIdx.z * sX;
    if (X2) X2 += blockIdx.z * sX;
    rkey += blockIdx.z * sKeys;
    ckey += blockIdx.z * sKeys;
    out += blockIdx.z * sOut;
    constexpr int V = VEC ? 4 : 1;
    const unsigned per_row = (unsigned)nB / V;
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    float m = 0.f;
    if (i < (unsigned)nA * per_row) {
        const unsigned r = i / per_row, c = (i - r * per_row) * V;
        const size_t at = (size_t)r * nB + c;
        const float mb = key2f(rkey[r]);
        if constexpr (VEC) {
            f32x4 x = *(const f32x4 *)(X + at);
            if (X2) {
                const f32x4 y = *(const f32x4 *)(X2 + at);
#pragma unroll
                for (int j = 0; j < 4; ++j) x[j] += y[j];
            }
            const mm_i32x4 ck = *(const mm_i32x4 *)(ckey + c);
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = mm_value(x[j], mb, key2f(ck[j]));
                m = fmaxf(m, fabsf(v[j]));
            }
            *(f32x4 *)(out + at) = v;
        } else {
            const float v = mm_value(X2 ? X[at] + X2[at] : X[at], mb, key2f(ckey[c]));
            out[at] = v;
            m = fabsf(v);
        }
    }
    if (amax) {                                     // largest magnitude of the volume (the fused consensus kernel scales its fp16 planes by it)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
        // the word only grows: a (possibly stale) plain read first keeps nearly every wave off the atomic unit
        int *dst = amax + blockIdx.z * sKeys;
        if ((threadIdx.x & 63) == 0 && __float_as_int(m) > *(volatile int *)dst) atomicMax(dst, __float_as_int(m));
    }
}

static void launch_mm_apply(const float *X, int nA, int nB, const int *rkey, const int *ckey, float *out, size_t sX,
                            size_t sKeys, size_t sOut, int *amax, const float *X2, size_t nz, hipStream_t stream) {
    const bool vec = nB % 4 == 0 && sX % 4 == 0 && sOut % 4 == 0 && sKeys % 4 == 0 && ((uintptr_t)X & 15) == 0 &&
                     ((uintptr_t)out & 15) == 0 && ((uintptr_t)X2 & 15) == 0 && ((uintptr_t)ckey & 15) == 0;
    const size_t items = (size_t)nA * (nB / (vec ? 4 : 1));
    const dim3 grid((unsigned)((items + 255) / 256), 1, (unsigned)nz);
    if (vec) hipLaunchKernelGGL(mm_apply_kernel<true>, grid, dim3(256), 0, stream, X, nA, nB, rkey, ckey, out, sX, sKeys, sOut, amax, X2);
    else hipLaunchKernelGGL(mm_apply_kernel<false>, grid, dim3(256), 0, stream, X, nA, nB, rkey, ckey, out, sX, sKeys, sOut, amax, X2);
}

// ------------------------------------------------------------------------------------------------
// 4. neighbourhood consensus (ncn/model.py:145-155; conv4d.py:12-74): consensus.hip, one fused kernel for every
//    volume and batch size
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// 5. matches (ncn/extract_ncmatches.py:6-94 twice; patch2pix.py:340-375)
// ------------------------------------------------------------------------------------------------
struct MatchArgs {
    const float *X;
    const uint8_t *delta;
    int hA, wA, hB, wB, ksize, upsample, center;
    long long *matches;
    float *scores;
    size_t sX, sM;      // per-pair strides: cells of the volume, rows of the match list
};
__device__ __forceinline__ MatchArgs match_args_of_pair(MatchArgs m, size_t z) {
    m.X += z * m.sX;
    if (m.delta) m.delta += z * m.sX;
    m.matches += z * m.sM * 4;
    m.scores += z * m.sM;
    return m;
}

__device__ __forceinline__ void emit_match(const MatchArgs &m, int out_row, int ra, int cb, float sum_exp) {
    int ia = ra / m.wA, ja = ra - ia * m.wA, ib = cb / m.wB, jb = cb - ib * m.wB;
    if (m.ksize > 1 && m.delta) {
        const int k = m.ksize;
        int s = m.delta[(size_t)ra * (m.hB * m.wB) + cb];
        const int dl = s % k; s /= k;
        const int dk = s % k; s /= k;
        const int dj = s % k; s /= k;
        ia = ia * k + s; ja = ja * k + dj; ib = ib * k + dk; jb = jb * k + dl;
    } else if (m.ksize > 1) {
        ia *= m.ksize; ja *= m.ksize; ib *= m.ksize; jb *= m.ksize;
    }
    const long long up = m.upsample, off = m.center ? m.upsample / 2 : 0;
    long long *o = m.matches + (size_t)out_row * 4;
    o[0] = up * ja + off; o[1] = up * ia + off; o[2] = up * jb + off; o[3] = up * ib + off;
    m.scores[out_row] = 1.0f / sum_exp;       // max of softmax = exp(0) / sum exp(x - max)
}

// direction B->A: one block per 16 columns (a row of the block = half a 128-byte line: 8-column blocks fetched every line of
// the volume four times; 32 columns leave too few blocks per pair), 16 interleaved row slices
constexpr int MC_COLS = 16, MC_SLICES = 16;
__global__ __launch_bounds__(256) void match_cols_kernel(MatchArgs m_) {
    const MatchArgs m = match_args_of_pair(m_, blockIdx.z);
    __shared__ float smax[MC_SLICES][MC_COLS];
    __shared__ int sarg[MC_SLICES][MC_COLS];
    __shared__ float ssum[MC_SLICES][MC_COLS];
    const int nA = m.hA * m.wA, nB = m.hB * m.wB;
    const int cs = threadIdx.x & (MC_COLS - 1), rs = threadIdx.x / MC_COLS;
    const int col = blockIdx.x * MC_COLS + cs;
    const bool ok = col < nB;
    float best = -INFINITY;
    int arg = 0x7fffffff;
    if (ok)
        for (int r = rs; r < nA; r += MC_SLICES) {
            const float v = m.X[(size_t)r * nB + col];
            if (v > best) { best = v; arg = r; }
        }
    smax[rs][cs] = best; sarg[rs][cs] = arg;
    __syncthreads();
    float gb = smax[0][cs];
    int ga = sarg[0][cs];
#pragma unroll
    for (int s = 1; s < MC_SLICES; ++s) {
        const float v = smax[s][cs];
        const int a = sarg[s][cs];
        if (v > gb || (v == gb && a < ga)) { gb = v; ga = a; }
    }
    float sum = 0.f;
    if (ok)
        for (int r = rs; r < nA; r += MC_SLICES) sum += expf(m.X[(size_t)r * nB + col] - gb);
    ssum[rs][cs] = sum;
    __syncthreads();
    if (rs == 0 && ok) {
        float t = 0.f;
#pragma unroll
        for (int s = 0; s < MC_SLICES; ++s) t += ssum[s][cs];
        emit_match(m, col, ga, col, t);
    }
}

// direction A->B: one wave per row
__global__ __launch_bounds__(256) void match_rows_kernel(MatchArgs m_) {
    const MatchArgs m = match_args_of_pair(m_, blockIdx.z);
    const int nA = m.hA * m.wA, nB = m.hB * m.wB;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= nA) return;
    const float *x = m.X + (size_t)row * nB;
    float best = -INFINITY;
    int arg = 0x7fffffff;
    for (int c = lane; c < nB; c += 64) {
        const float v = x[c];
        if (v > best) { best = v; arg = c; }
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const float ov = __shfl_xor(best, s);
        const int oa = __shfl_xor(arg, s);
        if (ov > best || (ov == best && oa < arg)) { best = ov; arg = oa; }
    }
    float sum = 0.f;
    for (int c = lane; c < nB; c += 64) sum += expf(x[c] - best);
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) sum += __shfl_xor(sum, s);
    if (lane == 0) emit_match(m, nB + row, row, arg, sum);
}

__global__ void delta_unpack_kernel(const uint8_t *__restrict__ delta, size_t n, int k, long long *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int s = delta[i];
    out[3 * n + i] = s % k; s /= k;
    out[2 * n + i] = s % k; s /= k;
    out[1 * n + i] = s % k; s /= k;
    out[i] = s;
}

// workspace carve-up shared by the size query and the launcher
struct CoarseWs {
    size_t fnA, fnB, P, Y, Y2, keys, total;   // byte offsets
};
static CoarseWs coarse_ws(int C, int hA, int wA, int hB, int wB, int k) {
    auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
    const size_t nA = (size_t)hA * wA, nB = (size_t)hB * wB;
    const size_t nAc = nA / (k * k), nBc = nB / (k * k);
    CoarseWs w;
    size_t off = 0;
    w.fnA = off; off += al(((nA + 127) / 128) * 128 * C * 4);      // two fp16 planes in blocks of 128 positions x 32 channels
    w.fnB = off; off += al(((nB + 127) / 128) * 128 * C * 4);
    w.P = off; off += al(nAc * nBc * 4);
    w.Y = off; off += al(nAc * nBc * 4);     // the two branches of the consensus net
    w.Y2 = off; off += al(nAc * nBc * 4);
    w.keys = off; off += al((2 * (nAc + nBc) + 1) * 4);      // row / column maxima of both mutual matchings + max |X|
    w.total = off;
    return w;
}

// consensus.hip
void pack_nc_fused(const float *w1, const float *b1, const float *w2, std::vector<unsigned char> &out);
int launch_nc_fused(const float *X, float *Y, float *Y2, size_t stride, int pairs, int d0, int d1, int d2, int d3,
                    const unsigned char *w_dev, float b2, const int *xmax, size_t xmax_stride, const int *forced_tile,
                    hipStream_t stream);
int launch_absmax(const float *x, size_t n, size_t stride, int pairs, int *out, size_t out_stride, hipStream_t stream);

}  // namespace p2p

using namespace p2p;

extern "C" int p2p_ncn_create(const float *w1, const float *b1, const float *w2, const float *b2, p2p_ncn **out) {
    P2P_REQUIRE(w1 && b1 && w2 && b2 && out, P2P_EINVAL, "p2p_ncn_create: null argument");
    // stored layout (conv4d.py:119-120): w1s[da][o][ci=0][db][dc][dd], w2s[da][o=0][ci][db][dc][dd] -> MFMA fragments of
    // both branches (consensus.hip)
    std::vector<unsigned char> wf;
    pack_nc_fused(w1, b1, w2, wf);
    unsigned char *wfd = nullptr;
    hipError_t e = hipMalloc(&wfd, wf.size());
    if (e == hipSuccess) e = hipMemcpy(wfd, wf.data(), wf.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (wfd) (void)hipFree(wfd);
        set_error("upload of the consensus weights failed: %s", hipGetErrorString(e));
        return P2P_EHIP;
    }
    p2p_ncn *n = new p2p_ncn();
    n->b2 = b2[0];
    n->wfused = wfd;
    n->tile[0] = n->tile[1] = n->tile[2] = 0;
    *out = n;
    return P2P_OK;
}

extern "C" int p2p_ncn_set_tile(p2p_ncn *ncn, int ta, int tb, int tc) {
    P2P_REQUIRE(ncn && ta >= 0 && tb >= 0 && tc >= 0, P2P_EINVAL, "p2p_ncn_set_tile: bad argument");
    // (0, 0, 0) = automatic; (ta, tb, tc) with tb, tc > 0 = forced (ta = 0: only the march length is picked); anything else
    // would be ignored silently
    P2P_REQUIRE((tb > 0 && tc > 0) || (ta == 0 && tb == 0 && tc == 0), P2P_EINVAL,
                "p2p_ncn_set_tile: (%d, %d, %d) is neither (0, 0, 0) nor a tile with tb, tc > 0", ta, tb, tc);
    ncn->tile[0] = ta; ncn->tile[1] = tb; ncn->tile[2] = tc;
    return P2P_OK;
}

extern "C" void p2p_ncn_destroy(p2p_ncn *ncn) {
    if (!ncn) return;
    (void)hipFree(ncn->wfused);
    delete ncn;
}

extern "C" size_t p2p_coarse_workspace_bytes(int channels, int hA, int wA, int hB, int wB, int ksize) {
    if (channels <= 0 || hA <= 0 || wA <= 0 || hB <= 0 || wB <= 0 || ksize < 1) return 0;
    return coarse_ws(channels, hA, wA, hB, wB, ksize).total;
}

extern "C" int p2p_coarse_forward_batch(const float *featA, const float *featB, int batch, int C, int hA, int wA, int hB,
                                        int wB, int ksize, const p2p_ncn *ncn, float *corr4d_out, uint8_t *delta_out,
                                        void *workspace, size_t workspace_bytes, p2p_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    P2P_REQUIRE(featA && featB && ncn && corr4d_out && workspace, P2P_EINVAL, "p2p_coarse_forward: null argument");
    P2P_REQUIRE(batch >= 1 && batch <= 65535, P2P_EINVAL, "p2p_coarse_forward: batch %d out of range", batch);
    P2P_REQUIRE(ksize == 1 || ksize == 2, P2P_EUNSUPPORTED, "p2p_coarse_forward: ksize %d not supported (1 or 2)", ksize);
    P2P_REQUIRE(C > 0 && C % 32 == 0 && C <= 256, P2P_EUNSUPPORTED, "p2p_coarse_forward: channels %d (multiple of 32, <= 256)", C);
    P2P_REQUIRE(hA > 0 && wA > 0 && hB > 0 && wB > 0 && hA % ksize == 0 && wA % ksize == 0 && hB % ksize == 0 &&
                    wB % ksize == 0, P2P_EINVAL, "p2p_coarse_forward: feature map sizes must be positive multiples of ksize");
    const CoarseWs ws = coarse_ws(C, hA, wA, hB, wB, ksize);
    P2P_REQUIRE(workspace_bytes >= ws.total, P2P_ENOMEM, "p2p_coarse_forward: workspace %zu < %zu bytes (one pair)", workspace_bytes,
                ws.total);
    const int nA = hA * wA, nB = hB * wB, kk = ksize * ksize;
    const int nAc = nA / kk, nBc = nB / kk;
    const size_t nel = (size_t)nAc * nBc;
    const size_t sWs = ws.total / 4;        // every workspace buffer of pair z sits z * ws.total bytes further on
    const int per_launch = (int)std::min<size_t>(batch, workspace_bytes / ws.total);   // pairs the workspace holds at once
    const int d0 = hA / ksize, d1 = wA / ksize, d2 = hB / ksize, d3 = wB / ksize;
    int dev = 0;
    P2P_HIP_CHECK(hipGetDevice(&dev));
    static DeviceOnce attr_set_dev;      // per device: a process may drive several GPUs
    if (!attr_set_dev.done(dev)) {
        P2P_HIP_CHECK(hipFuncSetAttribute((const void *)corr_pool_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, CX_LDS));
        P2P_HIP_CHECK(hipFuncSetAttribute((const void *)corr_pool_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, CX_LDS));
        attr_set_dev.set(dev);
    }

    for (int z0 = 0; z0 < batch; z0 += per_launch) {
        const unsigned nz = (unsigned)std::min(per_launch, batch - z0);
        const float *fA = featA + (size_t)z0 * C * nA, *fB = featB + (size_t)z0 * C * nB;
        float *out = corr4d_out + (size_t)z0 * nel;
        uint8_t *dout = delta_out ? delta_out + (size_t)z0 * nel : nullptr;
        char *base = (char *)workspace;
        unsigned short *fnA = (unsigned short *)(base + ws.fnA), *fnB = (unsigned short *)(base + ws.fnB);
        float *P = (float *)(base + ws.P), *Y = (float *)(base + ws.Y), *Y2 = (float *)(base + ws.Y2);
        int *rkey1 = (int *)(base + ws.keys), *ckey1 = rkey1 + nAc, *rkey2 = ckey1 + nBc, *ckey2 = rkey2 + nAc;

        const int nkeys = 2 * (nAc + nBc);
        int *xmax = rkey1 + nkeys;
        {
            PrepArgs pa;
            pa.F[0] = fA; pa.F[1] = fB; pa.Fn[0] = fnA; pa.Fn[1] = fnB;
            pa.h[0] = hA; pa.w[0] = wA; pa.h[1] = hB; pa.w[1] = wB;
            pa.sF[0] = (size_t)C * nA; pa.sF[1] = (size_t)C * nB;
            pa.C = C; pa.k = ksize; pa.sFn = 2 * sWs;
            pa.keys = rkey1; pa.nkeys = nkeys; pa.sKeys = sWs;
            const int gp = std::max(ceil_div(std::max(nA, nB), PREP_P), ceil_div(nkeys + 1, 256));
            hipLaunchKernelGGL(prep_kernel, dim3(gp, 2, nz), dim3(256), 0, stream, pa);
        }
        {
            const int gx = ceil_div(nB, CT), gy = ceil_div(nA, CT);
            const long long ntiles = (long long)gx * gy * nz;
            P2P_REQUIRE(ntiles < (1ll << 30), P2P_EUNSUPPORTED, "p2p_coarse_forward: %lld correlation tiles in one launch", ntiles);
            // persistent work-groups: two fit a compute unit, 32 compute units per XCD
            const int per_xcd = (int)((ntiles + 7) / 8);
            const dim3 cgrid((unsigned)(8 * std::min(per_xcd, 64)));
            if (ksize == 1)
                hipLaunchKernelGGL(corr_pool_kernel<1>, cgrid, dim3(256), CX_LDS, stream, fnA, fnB, nA, nB, C, P, (uint8_t *)nullptr,
                                   2 * sWs, sWs, (size_t)0, gx, gy, (int)ntiles, per_xcd);
            else
                hipLaunchKernelGGL(corr_pool_kernel<2>, cgrid, dim3(256), CX_LDS, stream, fnA, fnB, nA, nB, C, P, dout, 2 * sWs, sWs,
                                   nel, gx, gy, (int)ntiles, per_xcd);
        }

        const int col_groups = ceil_div(nBc, 256) * ceil_div(nAc, 64);
        const dim3 mgrid(col_groups + ceil_div(nAc, 4), 1, nz);
        hipLaunchKernelGGL(maxima_kernel, mgrid, dim3(256), 0, stream, P, nAc, nBc, rkey1, ckey1, sWs, sWs, (const float *)nullptr, col_groups);

        // first mutual matching, in place on the pooled volume (+ max |X| for the consensus kernel's operand scale)
        launch_mm_apply(P, nAc, nBc, rkey1, ckey1, P, sWs, sWs, sWs, xmax, nullptr, nz, stream);
        {   // both consensus layers, both branches: relu(.) of the direct branch into Y, of the transposed one into Y2
            const int st = launch_nc_fused(P, Y, Y2, sWs, (int)nz, d0, d1, d2, d3, ncn->wfused, ncn->b2, xmax, sWs, ncn->tile, stream);
            if (st != P2P_OK) return st;
        }
        hipLaunchKernelGGL(maxima_kernel, mgrid, dim3(256), 0, stream, Y, nAc, nBc, rkey2, ckey2, sWs, sWs, Y2, col_groups);
        launch_mm_apply(Y, nAc, nBc, rkey2, ckey2, out, sWs, sWs, nel, nullptr, Y2, nz, stream);
    }
    return check_launch("coarse_forward kernels");
}

extern "C" int p2p_coarse_forward(const float *featA, const float *featB, int C, int hA, int wA, int hB, int wB,
                                  int ksize, const p2p_ncn *ncn, float *corr4d_out, uint8_t *delta_out,
                                  void *workspace, size_t workspace_bytes, p2p_stream_t stream) {
    return p2p_coarse_forward_batch(featA, featB, 1, C, hA, wA, hB, wB, ksize, ncn, corr4d_out, delta_out, workspace,
                                    workspace_bytes, stream);
}

extern "C" int p2p_neigh_consensus_batch(const float *x, int batch, int hA, int wA, int hB, int wB, const p2p_ncn *ncn, float *y_out,
                                         void *workspace, size_t workspace_bytes, p2p_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    P2P_REQUIRE(x && ncn && y_out && workspace, P2P_EINVAL, "p2p_neigh_consensus: null argument");
    P2P_REQUIRE(batch >= 1 && batch <= 65535 && hA > 0 && wA > 0 && hB > 0 && wB > 0, P2P_EINVAL, "p2p_neigh_consensus: bad sizes");
    P2P_REQUIRE(workspace_bytes >= (size_t)batch * sizeof(int), P2P_ENOMEM, "p2p_neigh_consensus: workspace of %zu bytes needed (4 per volume)",
                (size_t)batch * sizeof(int));
    const size_t nel = (size_t)hA * wA * hB * wB;
    int *xmax = (int *)workspace;
    P2P_HIP_CHECK(hipMemsetAsync(y_out, 0, (size_t)batch * nel * sizeof(float), stream));
    P2P_HIP_CHECK(hipMemsetAsync(xmax, 0, (size_t)batch * sizeof(int), stream));
    const int st = launch_absmax(x, nel, nel, batch, xmax, 1, stream);
    if (st != P2P_OK) return st;
    return launch_nc_fused(x, y_out, nullptr, nel, batch, hA, wA, hB, wB, ncn->wfused, ncn->b2, xmax, 1, ncn->tile, stream);
}

extern "C" int p2p_delta_unpack(const uint8_t *delta, size_t n, int ksize, int64_t *out, p2p_stream_t stream) {
    P2P_REQUIRE(delta && out && ksize >= 1, P2P_EINVAL, "p2p_delta_unpack: bad argument");
    if (n == 0) return P2P_OK;
    hipLaunchKernelGGL(delta_unpack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, delta, n,
                       ksize, (long long *)out);
    return check_launch("delta_unpack_kernel");
}

extern "C" int p2p_coarse_matches_batch(const float *corr4d, const uint8_t *delta, int batch, int hA, int wA, int hB, int wB,
                                        int ksize, int upsample, int center, int64_t *matches_out, float *scores_out,
                                        p2p_stream_t stream) {
    P2P_REQUIRE(corr4d && matches_out && scores_out, P2P_EINVAL, "p2p_coarse_matches: null argument");
    P2P_REQUIRE(batch >= 1 && batch <= 65535, P2P_EINVAL, "p2p_coarse_matches: batch %d out of range", batch);
    P2P_REQUIRE(hA > 0 && wA > 0 && hB > 0 && wB > 0 && ksize >= 1, P2P_EINVAL, "p2p_coarse_matches: bad sizes");
    P2P_REQUIRE(ksize == 1 || delta, P2P_EINVAL, "p2p_coarse_matches: delta required when ksize > 1");
    const int nA = hA * wA, nB = hB * wB;
    MatchArgs m{corr4d, delta, hA, wA, hB, wB, ksize, upsample, center, (long long *)matches_out, scores_out,
                (size_t)nA * nB, (size_t)nA + nB};
    hipLaunchKernelGGL(match_cols_kernel, dim3(ceil_div(nB, MC_COLS), 1, batch), dim3(256), 0, (hipStream_t)stream, m);
    hipLaunchKernelGGL(match_rows_kernel, dim3(ceil_div(nA, 4), 1, batch), dim3(256), 0, (hipStream_t)stream, m);
    return check_launch("match kernels");
}

extern "C" int p2p_coarse_matches(const float *corr4d, const uint8_t *delta, int hA, int wA, int hB, int wB, int ksize,
                                  int upsample, int center, int64_t *matches_out, float *scores_out, p2p_stream_t stream) {
    return p2p_coarse_matches_batch(corr4d, delta, 1, hA, wA, hB, wB, ksize, upsample, center, matches_out, scores_out, stream);
}
