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

// Orientation: the BANK rows are the M side of the MFMA tile and the QUERIES its N side, so that a lane's 16 accumulator registers are
// 16 bank rows of ONE query (column r) and the three smallest distances are kept straight from the accumulators, branch-free -- no
// trip of every tile through LDS and no scalar scan (round 4: the LDS epilogue was 2.4 of the 6.3 ms of the 1 M-query WideResNet-50
// layer1 call, the one-row-at-a-time norm prologue another ~2 ms; tools/knn_probe.py).
constexpr int BB = 128, BQ = 128, BK = 32, LDK = BK + 4, TB = 2, TQ = 2, NT = 256;      // bank rows x queries per workgroup tile
constexpr int STAGE = (BB + BQ) * LDK;          // floats

struct KnnParams {
    const float* x;       // [N][D] queries (not normalised)
    const float* bank;    // [R][D] bank rows, L2-normalised (ssad_l2_normalize_rows)
    float* out;           // [N]
    int64_t N;
    int D, R, k;
};

// a <= b <= c are the three smallest so far; v joins them (no branches: min / max only)
__device__ __forceinline__ void keep3(float v, float& a, float& b, float& c) {
    c = fminf(c, fmaxf(b, v));
    b = fminf(b, fmaxf(a, v));
    a = fminf(a, v);
}

__global__ __launch_bounds__(NT, 2) void cosine_knn_fused_kernel(KnnParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* nrm_s = lds + 2 * STAGE;             // [BQ]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wb = wave >> 1, wq = wave & 1;    // 64-row bank block / 64-query block of this wave
    const int64_t m0 = (int64_t)blockIdx.x * BQ;
    const int sc = tid & 7, sr = tid >> 3;      // staging: 16-byte chunk sc of rows sr + 32 i

    // ---- query norms: one wave per row, lane-strided squares + xor butterfly (l2norm_rows_kernel's order), eight rows in flight
    // per wave (one row at a time was a chain of 32 memory latencies per workgroup) ----
    for (int base = wave; base < BQ; base += 32) {
        float s[8];
        const float* q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t row = m0 + base + 4 * u;
            q[u] = row < p.N ? p.x + row * p.D : nullptr;
            s[u] = 0.f;
        }
        for (int k = lane; k < p.D; k += 64) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = q[u] ? q[u][k] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] += v[u] * v[u];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float t = s[u];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
            if (lane == 0) nrm_s[base + 4 * u] = sqrtf(t);
        }
    }
    __syncthreads();
    float nrm[4];
    const float* qptr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t row = m0 + sr + 32 * i;
        nrm[i] = nrm_s[sr + 32 * i];
        qptr[i] = row < p.N ? p.x + row * p.D + sc * 4 : nullptr;
    }

    // running three smallest distances of this lane's queries (column r of its TQ query blocks) over the bank rows it has seen
    float best[TQ][3];
#pragma unroll
    for (int j = 0; j < TQ; ++j) best[j][0] = best[j][1] = best[j][2] = INFINITY;
    const int nks = p.D / BK;

    for (int n0 = 0; n0 < p.R; n0 += BB) {
        const float* bptr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = n0 + sr + 32 * i;
            bptr[i] = row < p.R ? p.bank + (int64_t)row * p.D + sc * 4 : nullptr;
        }
        f32x16 acc[TB][TQ];
#pragma unroll
        for (int i = 0; i < TB; ++i)
#pragma unroll
            for (int j = 0; j < TQ; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        f32x4 rq[4], rb[4];
        auto load = [&](int ks) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (qptr[i]) v = *(const f32x4*)(qptr[i] + ks * BK);
                rq[i] = v;
                f32x4 w = {0.f, 0.f, 0.f, 0.f};
                if (bptr[i]) w = *(const f32x4*)(bptr[i] + ks * BK);
                rb[i] = w;
            }
        };
        auto store = [&](float* st) {       // the loads were issued a whole K-step of MFMAs ago; normalise while staging
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (qptr[i]) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) rq[i][k] = rq[i][k] / nrm[i];  // l2norm_rows_kernel's expression
                }
                *(f32x4*)(st + (sr + 32 * i) * LDK + sc * 4) = rb[i];                   // bank rows: the tile's M side
                *(f32x4*)(st + BB * LDK + (sr + 32 * i) * LDK + sc * 4) = rq[i];        // queries: its N side
            }
        };
        __syncthreads();                        // every wave has left the previous bank tile's last stage
        load(0);
        store(lds);
        __syncthreads();
        for (int ks = 0; ks < nks; ++ks) {
            const float* cur = lds + (ks & 1) * STAGE;
            if (ks + 1 < nks) load(ks + 1);
            const float* As = cur + (wb * 32 * TB + r) * LDK + h * 4;
            const float* Bs = cur + BB * LDK + (wq * 32 * TQ + r) * LDK + h * 4;
#pragma unroll
            for (int kk = 0; kk < BK / 8; ++kk) {
                f32x4 a[TB], b[TQ];
#pragma unroll
                for (int i = 0; i < TB; ++i) a[i] = *(const f32x4*)(As + i * 32 * LDK + kk * 8);
#pragma unroll
                for (int j = 0; j < TQ; ++j) b[j] = *(const f32x4*)(Bs + j * 32 * LDK + kk * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TB; ++i)
#pragma unroll
                        for (int j = 0; j < TQ; ++j) acc[i][j] = mfma32(a[i][e], b[j][e], acc[i][j]);
            }
            if (ks + 1 < nks) store(lds + ((ks + 1) & 1) * STAGE);
            __syncthreads();
        }
        // ---- register e of lane (r, h) in block (i, j): bank row n0 + (wb TB + i) 32 + (e & 3) + 8 (e >> 2) + 4 h, query column r ----
#pragma unroll
        for (int i = 0; i < TB; ++i) {
            const int row0 = n0 + (wb * TB + i) * 32 + 4 * h;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const bool ok = row0 + (e & 3) + 8 * (e >> 2) < p.R;
#pragma unroll
                for (int j = 0; j < TQ; ++j) {
                    float d = 1.f - acc[i][j][e];
                    d = fminf(fmaxf(d, 0.f), 2.f);
                    keep3(ok ? d : INFINITY, best[j][0], best[j][1], best[j][2]);
                }
            }
        }
    }
    // ---- a query's candidates sit in the two lane halves of two waves (wb = 0, 1): halves by shuffle, waves through LDS ----
    __syncthreads();                            // the stages are dead
    float* M = lds;                             // [2 wq][TQ][32][3]
#pragma unroll
    for (int j = 0; j < TQ; ++j) {
        const float oa = __shfl_xor(best[j][0], 32), ob = __shfl_xor(best[j][1], 32), oc = __shfl_xor(best[j][2], 32);
        keep3(oa, best[j][0], best[j][1], best[j][2]);
        keep3(ob, best[j][0], best[j][1], best[j][2]);
        keep3(oc, best[j][0], best[j][1], best[j][2]);
        if (wb == 1 && h == 0) {
            float* m = M + ((wq * TQ + j) * 32 + r) * 3;
            m[0] = best[j][0]; m[1] = best[j][1]; m[2] = best[j][2];
        }
    }
    __syncthreads();
    if (wb == 0 && h == 0) {
#pragma unroll
        for (int j = 0; j < TQ; ++j) {
            const float* m = M + ((wq * TQ + j) * 32 + r) * 3;
            float a = best[j][0], b = best[j][1], c = best[j][2];
            keep3(m[0], a, b, c);
            keep3(m[1], a, b, c);
            keep3(m[2], a, b, c);
            const int64_t row = m0 + (wq * TQ + j) * 32 + r;
            if (row < p.N) {
                float s = a;                    // the k smallest, smallest first
                if (p.k > 1) s += b;
                if (p.k > 2) s += c;
                p.out[row] = s / (float)p.k;
            }
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
    SSAD_CHECK_ARG(cdiv64(N, BQ) < (int64_t)2147483647, "too many rows for one launch");
    constexpr int lds_bytes = (2 * STAGE + BQ) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        SSAD_SET_DYN_LDS(cosine_knn_fused_kernel, lds_bytes);
        attr_set = true;
    }
    KnnParams p{x, bank_normalized, out, N, D, R, k};
    hipLaunchKernelGGL(cosine_knn_fused_kernel, dim3((unsigned)cdiv64(N, BQ)), dim3(NT), lds_bytes, (hipStream_t)stream, p);
    SSAD_CHECK_LAUNCH();
    return 0;
}
