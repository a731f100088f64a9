/* The coarse stage of the Patch2Pix matching path driven from plain C through the C ABI of libp2p_hip.so
 * (include/p2p_hip.h) -- no Python, no torch: what a non-Python host of the reference's
 * Patch2Pix.forward_coarse_match + cal_coarse_matches (networks/patch2pix.py:120-136, :340-375) would do.
 *
 *   cabi_coarse IN.bin OUT.bin
 *
 * IN.bin : int32 header {B, C, hA, wA, hB, wB, ksize, upsample}, then fp32 arrays
 *          w1[3*16*1*27], b1[16], w2[3*1*16*27], b2[1]   (NeighConsensus filters in the reference's stored layout,
 *                                                          networks/ncn/conv4d.py:119-120)
 *          featA[B*C*hA*wA], featB[B*C*hB*wB]            (layer-3 feature maps)
 * OUT.bin: fp32 corr4d[B*nAc*nBc], uint8 delta[B*nAc*nBc] (only when ksize > 1), int64 matches[B*(nAc+nBc)*4],
 *          fp32 scores[B*(nAc+nBc)]
 * Exit codes: 0 ok, 2 usage / file error, 3 no HIP device, 4 library error.
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "p2p_hip.h"

#define HIP_OK(call)                                                                     \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));                   \
            return 4;                                                                    \
        }                                                                                \
    } while (0)
#define P2P_OK_(call)                                                                    \
    do {                                                                                 \
        int s_ = (call);                                                                 \
        if (s_ != 0) {                                                                   \
            fprintf(stderr, "%s -> %d: %s\n", #call, s_, p2p_last_error());              \
            return 4;                                                                    \
        }                                                                                \
    } while (0)

static float *read_floats(FILE *f, size_t n) {
    float *p = (float *)malloc(n * sizeof(float));
    if (!p || fread(p, sizeof(float), n, f) != n) { free(p); return NULL; }
    return p;
}

int main(int argc, char **argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s IN.bin OUT.bin\n", argv[0]); return 2; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { fprintf(stderr, "no HIP device\n"); return 3; }

    FILE *in = fopen(argv[1], "rb");
    int32_t hd[8];
    if (!in || fread(hd, sizeof(int32_t), 8, in) != 8) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    const int B = hd[0], C = hd[1], hA = hd[2], wA = hd[3], hB = hd[4], wB = hd[5], ksize = hd[6], upsample = hd[7];
    const size_t nfa = (size_t)B * C * hA * wA, nfb = (size_t)B * C * hB * wB;
    float *w1 = read_floats(in, 3 * 16 * 27), *b1 = read_floats(in, 16), *w2 = read_floats(in, 3 * 16 * 27),
          *b2 = read_floats(in, 1), *fa = read_floats(in, nfa), *fb = read_floats(in, nfb);
    fclose(in);
    if (!w1 || !b1 || !w2 || !b2 || !fa || !fb) { fprintf(stderr, "short input file\n"); return 2; }

    const int k = ksize > 1 ? ksize : 1;
    const size_t nAc = (size_t)(hA / k) * (wA / k), nBc = (size_t)(hB / k) * (wB / k);
    const size_t ncell = (size_t)B * nAc * nBc, nmatch = (size_t)B * (nAc + nBc);

    p2p_ncn *ncn = NULL;
    P2P_OK_(p2p_ncn_create(w1, b1, w2, b2, &ncn));          /* host pointers: packed and uploaded once */

    /* device buffers are the caller's; the workspace holds all B pairs here (one pair's worth is the minimum) */
    const size_t ws_bytes = (size_t)B * p2p_coarse_workspace_bytes(C, hA, wA, hB, wB, ksize);
    float *d_fa, *d_fb, *d_corr, *d_scores;
    uint8_t *d_delta = NULL;
    int64_t *d_matches;
    void *d_ws;
    HIP_OK(hipMalloc((void **)&d_fa, nfa * 4));
    HIP_OK(hipMalloc((void **)&d_fb, nfb * 4));
    HIP_OK(hipMalloc((void **)&d_corr, ncell * 4));
    if (ksize > 1) HIP_OK(hipMalloc((void **)&d_delta, ncell));
    HIP_OK(hipMalloc((void **)&d_matches, nmatch * 4 * sizeof(int64_t)));
    HIP_OK(hipMalloc((void **)&d_scores, nmatch * 4));
    HIP_OK(hipMalloc(&d_ws, ws_bytes));
    HIP_OK(hipMemcpy(d_fa, fa, nfa * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_fb, fb, nfb * 4, hipMemcpyHostToDevice));

    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    P2P_OK_(p2p_coarse_forward_batch(d_fa, d_fb, B, C, hA, wA, hB, wB, ksize, ncn, d_corr, d_delta, d_ws, ws_bytes,
                                     (p2p_stream_t)stream));
    P2P_OK_(p2p_coarse_matches_batch(d_corr, d_delta, B, hA / k, wA / k, hB / k, wB / k, ksize, upsample, 1, d_matches,
                                     d_scores, (p2p_stream_t)stream));
    HIP_OK(hipStreamSynchronize(stream));

    float *corr = (float *)malloc(ncell * 4), *scores = (float *)malloc(nmatch * 4);
    uint8_t *delta = (uint8_t *)malloc(ncell);
    int64_t *matches = (int64_t *)malloc(nmatch * 4 * sizeof(int64_t));
    HIP_OK(hipMemcpy(corr, d_corr, ncell * 4, hipMemcpyDeviceToHost));
    if (ksize > 1) HIP_OK(hipMemcpy(delta, d_delta, ncell, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(matches, d_matches, nmatch * 4 * sizeof(int64_t), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(scores, d_scores, nmatch * 4, hipMemcpyDeviceToHost));

    FILE *out = fopen(argv[2], "wb");
    if (!out) { fprintf(stderr, "cannot write %s\n", argv[2]); return 2; }
    fwrite(corr, 4, ncell, out);
    if (ksize > 1) fwrite(delta, 1, ncell, out);
    fwrite(matches, sizeof(int64_t), nmatch * 4, out);
    fwrite(scores, 4, nmatch, out);
    fclose(out);
    printf("cabi_coarse: %d pair(s), volume %zu x %zu, %zu matches, first match (%lld, %lld, %lld, %lld) score %g\n", B, nAc,
           nBc, nmatch, (long long)matches[0], (long long)matches[1], (long long)matches[2], (long long)matches[3],
           scores[0]);

    p2p_ncn_destroy(ncn);
    (void)hipFree(d_fa); (void)hipFree(d_fb); (void)hipFree(d_corr); (void)hipFree(d_delta);
    (void)hipFree(d_matches); (void)hipFree(d_scores); (void)hipFree(d_ws);
    (void)hipStreamDestroy(stream);
    free(w1); free(b1); free(w2); free(b2); free(fa); free(fb); free(corr); free(delta); free(matches); free(scores);
    return 0;
}
