// Cosine k-NN scoring in ONE kernel: row L2-normalisation of the queries, the similarity GEMM against the (normalised) bank on the
// fp32 matrix cores and the mean of the k smallest clip(1 - sim, 0, 2) per query -- the N x R similarity matrix never exists in HBM.
//
// Replaces AnomalyDetector.predict of the reference (src/self_supervised/models.py:363-370: sklearn NearestNeighbors(metric='cosine')
// .kneighbors + torch.mean over the 3 distances) and, before this kernel, the launch chain l2norm_rows -> conv_igemm (sim) ->
// knn_mean: the chain wrote and re-read N x R floats (2.4 GB for the 1 M pixels of a WideResNet-50 layer1 scale).  Every
// intermediate value is formed by the same expression in the same order as in that chain (x / ||x|| per element; the k-order of the
// MFMA chain; the three smallest distances added smallest first), so the scores are bit-identical to it.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32, LDK = BK + 4, TM = 2, TN = 2, NT = 256;
constexpr int STAGE = (BM + BN) * LDK;          // floats
constexpr int LDC = BN + 4;

struct KnnParams {
    const float* x;       // [N][D] queries (not normalised)
    const float* bank;    // [R][D] bank rows, L2-normalised (ssad_l2_normalize_rows)
    float* out;           // [N]
    int64_t N;
    int D, R, k;
};

__device__ __forceinline__ void keep3(float v, float& a, float& b, float& c) {      // a <= b <= c: the three smallest so far
    if (v < c) {
        if (v < b) {
            c = b;
            if (v < a) { b = a; a = v; } else b = v;
        } else c = v;
    }
}

__global__ __launch_bounds__(NT, 2) void cosine_knn_fused_kernel(KnnParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* nrm_s = lds + 2 * STAGE;             // [BM]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int sc = tid & 7, sr = tid >> 3;      // staging: 16-byte chunk sc of rows sr + 32 i

    // ---- row norms: one wave per row, lane-strided squares + xor butterfly (l2norm_rows_kernel's order) ----
    for (int lr = wave; lr < BM; lr += 4) {
        const int64_t row = m0 + lr;
        float s = 0.f;
        if (row < p.N) {
            const float* q = p.x + row * p.D;
            for (int k = lane; k < p.D; k += 64) s += q[k] * q[k];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) nrm_s[lr] = sqrtf(s);
    }
    __syncthreads();
    float nrm[4];
    const float* aptr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t row = m0 + sr + 32 * i;
        nrm[i] = nrm_s[sr + 32 * i];
        aptr[i] = row < p.N ? p.x + row * p.D + sc * 4 : nullptr;
    }

    // running three smallest distances of the two rows this thread scans (one per epilogue pass), over its 32-column quarter
    float best[TM][3];
#pragma unroll
    for (int i = 0; i < TM; ++i) best[i][0] = best[i][1] = best[i][2] = INFINITY;
    const int nks = p.D / BK;

    for (int n0 = 0; n0 < p.R; n0 += BN) {
        const float* bptr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int col = n0 + sr + 32 * i;
            bptr[i] = col < p.R ? p.bank + (int64_t)col * p.D + sc * 4 : nullptr;
        }
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        f32x4 ra[4], rb[4];
        auto load = [&](int ks) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (aptr[i]) v = *(const f32x4*)(aptr[i] + ks * BK);
                ra[i] = v;
                f32x4 w = {0.f, 0.f, 0.f, 0.f};
                if (bptr[i]) w = *(const f32x4*)(bptr[i] + ks * BK);
                rb[i] = w;
            }
        };
        auto store = [&](float* st) {       // the loads were issued a whole K-step of MFMAs ago; normalise while staging
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (aptr[i]) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) ra[i][k] = ra[i][k] / nrm[i];  // l2norm_rows_kernel's expression
                }
                *(f32x4*)(st + (sr + 32 * i) * LDK + sc * 4) = ra[i];
                *(f32x4*)(st + BM * LDK + (sr + 32 * i) * LDK + sc * 4) = rb[i];
            }
        };
        __syncthreads();                        // the previous column tile's epilogue is done with the stage memory
        load(0);
        store(lds);
        __syncthreads();
        for (int ks = 0; ks < nks; ++ks) {
            const float* cur = lds + (ks & 1) * STAGE;
            if (ks + 1 < nks) load(ks + 1);
            const float* As = cur + (wm * 32 * TM + r) * LDK + h * 4;
            const float* Bs = cur + BM * LDK + (wn * 32 * TN + r) * LDK + h * 4;
#pragma unroll
            for (int kk = 0; kk < BK / 8; ++kk) {
                f32x4 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = *(const f32x4*)(As + i * 32 * LDK + kk * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = *(const f32x4*)(Bs + j * 32 * LDK + kk * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(a[i][e], b[j][e], acc[i][j]);
            }
            if (ks + 1 < nks) store(lds + ((ks + 1) & 1) * STAGE);
            __syncthreads();
        }
        // ---- epilogue: the tile goes through LDS one row-tile pass at a time; thread = (row lr of the pass, 32-column quarter) ----
        float* C = lds;
        const int lr = tid >> 2, q = tid & 3;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (i) __syncthreads();
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    C[(wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * LDC + (wn * TN + j) * 32 + r] = acc[i][j][e];
            __syncthreads();
            const float* crow = C + lr * LDC + q * 32;
#pragma unroll 8
            for (int c = 0; c < 32; ++c) {
                if (n0 + q * 32 + c < p.R) {
                    float d = 1.f - crow[c];
                    d = fminf(fmaxf(d, 0.f), 2.f);
                    keep3(d, best[i][0], best[i][1], best[i][2]);
                }
            }
        }
    }
    // ---- the four quarters of a row sit in four adjacent lanes ----
    const int lr = tid >> 2, q = tid & 3;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        float a = best[i][0], b = best[i][1], c = best[i][2];
#pragma unroll
        for (int o = 1; o <= 2; o <<= 1) {
            const float oa = __shfl_xor(a, o), ob = __shfl_xor(b, o), oc = __shfl_xor(c, o);
            keep3(oa, a, b, c);
            keep3(ob, a, b, c);
            keep3(oc, a, b, c);
        }
        const int64_t row = m0 + ((lr >> 5) * TM + i) * 32 + (lr & 31);      // lr >> 5 = the wave-row that produced the pass row
        if (q == 0 && row < p.N) {
            float s = a;
            if (p.k > 1) s += b;
            if (p.k > 2) s += c;
            p.out[row] = s / (float)p.k;
        }
    }
}

}  // namespace

// out[n] = mean of the k (1..3) smallest clip(1 - <x_n / ||x_n||, bank_r>, 0, 2) over the R bank rows; bank rows are L2-normalised
// (ssad_l2_normalize_rows).  D must be a multiple of 32.
extern "C" int ssad_cosine_knn_fused(const float* x, const float* bank_normalized, float* out, int64_t N, int D, int R, int k,
                                     void* stream) {
    SSAD_CHECK_ARG(x && bank_normalized && out && N > 0 && D > 0 && R > 0, "bad argument");
    SSAD_CHECK_ARG(D % BK == 0, "D must be a multiple of 32");
    SSAD_CHECK_ARG(k >= 1 && k <= 3 && k <= R, "k in 1..3 and <= bank rows");
    SSAD_CHECK_ARG(cdiv64(N, BM) < (int64_t)2147483647, "too many rows for one launch");
    constexpr int lds_bytes = (2 * STAGE + BM) * 4;
    static_assert((BM / TM) * LDC <= 2 * STAGE, "epilogue tile must fit the stages");
    static bool attr_set = false;
    if (!attr_set) {
        SSAD_SET_DYN_LDS(cosine_knn_fused_kernel, lds_bytes);
        attr_set = true;
    }
    KnnParams p{x, bank_normalized, out, N, D, R, k};
    hipLaunchKernelGGL(cosine_knn_fused_kernel, dim3((unsigned)cdiv64(N, BM)), dim3(NT), lds_bytes, (hipStream_t)stream, p);
    SSAD_CHECK_LAUNCH();
    return 0;
}
