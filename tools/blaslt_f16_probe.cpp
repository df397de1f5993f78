// Probe (not part of the product; compiled ON the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/blaslt_f16_probe.cpp -lhipblaslt -o /tmp/blaslt_probe):
// the split GEMM's product called on hipBLASLt directly -- fp16 A/B, fp32 C/D, fp32 compute, alpha, bias epilogue -- in case
// torch.addmm(..., out_dtype=float32) is not available on this build (tools/split_gemm_probe.py says so first).
// y[M,N] (row-major) = alpha * A[M,K'] . W[N,K']^T + bias[N]   ==   column-major  D[N x M] = alpha * op_T(W[K' x N]) . A[K' x M] + bias (per row of D)
// Checks a small case against the host, then times the shapes of a decode step / the rescoring forward for every algorithm the heuristic returns.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hipblaslt/hipblaslt.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#define HCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define BCHK(x) do { hipblasStatus_t s_ = (x); if (s_ != HIPBLAS_STATUS_SUCCESS) { fprintf(stderr, "%s: hipblas status %d (line %d)\n", #x, (int)s_, __LINE__); exit(1); } } while (0)

struct Gemm {
    hipblasLtMatmulDesc_t desc; hipblasLtMatrixLayout_t la, lb, lc; hipblasLtMatmulPreference_t pref;
    std::vector<hipblasLtMatmulHeuristicResult_t> algos;
};

static Gemm make(hipblasLtHandle_t h, int64_t M, int64_t N, int64_t K, const float *d_bias, uint64_t ws_bytes)
{
    Gemm g;
    BCHK(hipblasLtMatmulDescCreate(&g.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
    hipblasOperation_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
    BCHK(hipblasLtMatmulDescSetAttribute(g.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta)));
    BCHK(hipblasLtMatmulDescSetAttribute(g.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb)));
    if (d_bias) {
        hipblasLtEpilogue_t epi = HIPBLASLT_EPILOGUE_BIAS;
        hipDataType bt = HIP_R_32F;
        BCHK(hipblasLtMatmulDescSetAttribute(g.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi)));
        BCHK(hipblasLtMatmulDescSetAttribute(g.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &d_bias, sizeof(d_bias)));
        BCHK(hipblasLtMatmulDescSetAttribute(g.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt)));
    }
    BCHK(hipblasLtMatrixLayoutCreate(&g.la, HIP_R_16F, K, N, K));      // W: [N, K] row-major = K x N column-major
    BCHK(hipblasLtMatrixLayoutCreate(&g.lb, HIP_R_16F, K, M, K));      // A: [M, K] row-major = K x M column-major
    BCHK(hipblasLtMatrixLayoutCreate(&g.lc, HIP_R_32F, N, M, N));      // y: [M, N] row-major = N x M column-major
    BCHK(hipblasLtMatmulPreferenceCreate(&g.pref));
    BCHK(hipblasLtMatmulPreferenceSetAttribute(g.pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws_bytes, sizeof(ws_bytes)));
    g.algos.resize(16);
    int n = 0;
    BCHK(hipblasLtMatmulAlgoGetHeuristic(h, g.desc, g.la, g.lb, g.lc, g.lc, g.pref, (int)g.algos.size(), g.algos.data(), &n));
    g.algos.resize(n);
    return g;
}

static float urand() { return (float)rand() / RAND_MAX * 2.f - 1.f; }

int main()
{
    hipblasLtHandle_t h;
    BCHK(hipblasLtCreate(&h));
    const uint64_t ws_bytes = 64ull << 20;
    void *ws; HCHK(hipMalloc(&ws, ws_bytes));
    hipStream_t st; HCHK(hipStreamCreate(&st));
    // ---- a small case against the host ----
    {
        const int M = 40, N = 96, K = 192;
        std::vector<__half> A(M * K), W(N * K); std::vector<float> b(N), ref(M * N), got(M * N);
        for (auto &v : A) v = __float2half(urand());
        for (auto &v : W) v = __float2half(urand());
        for (auto &v : b) v = urand();
        const float alpha = 0.25f, beta = 0.f;
        for (int m = 0; m < M; m++) for (int n = 0; n < N; n++) {
            double s = 0; for (int k = 0; k < K; k++) s += (double)__half2float(A[m * K + k]) * (double)__half2float(W[n * K + k]);
            ref[m * N + n] = (float)(alpha * s + b[n]);
        }
        __half *dA, *dW; float *db, *dy;
        HCHK(hipMalloc(&dA, A.size() * 2)); HCHK(hipMalloc(&dW, W.size() * 2)); HCHK(hipMalloc(&db, N * 4)); HCHK(hipMalloc(&dy, M * N * 4));
        HCHK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); HCHK(hipMemcpy(dW, W.data(), W.size() * 2, hipMemcpyHostToDevice));
        HCHK(hipMemcpy(db, b.data(), N * 4, hipMemcpyHostToDevice));
        Gemm g = make(h, M, N, K, db, ws_bytes);
        printf("small case: %zu algorithms\n", g.algos.size());
        if (g.algos.empty()) { printf("NO algorithm for fp16 x fp16 -> fp32 with a bias epilogue\n"); return 2; }
        BCHK(hipblasLtMatmul(h, g.desc, &alpha, dW, g.la, dA, g.lb, &beta, dy, g.lc, dy, g.lc, &g.algos[0].algo, ws, ws_bytes, st));
        HCHK(hipStreamSynchronize(st));
        HCHK(hipMemcpy(got.data(), dy, M * N * 4, hipMemcpyDeviceToHost));
        double worst = 0; for (int i = 0; i < M * N; i++) worst = fmax(worst, fabs((double)got[i] - ref[i]));
        printf("small case: max abs err vs host float64 %.3e (alpha 0.25, bias per output column)\n", worst);
    }
    // ---- the shapes (K' = 3K) ----
    const int Ms[] = {600, 300, 40, 3200};
    const int shapes[][2] = {{3072, 1024}, {1024, 1024}, {4096, 1024}, {1024, 4096}, {50265, 1024}};
    hipEvent_t e0, e1; HCHK(hipEventCreate(&e0)); HCHK(hipEventCreate(&e1));
    for (int M : Ms) for (auto &s : shapes) {
        const int64_t N = s[0], K3 = 3ll * s[1];
        __half *dA, *dW; float *db, *dy;
        HCHK(hipMalloc(&dA, (size_t)M * K3 * 2)); HCHK(hipMalloc(&dW, (size_t)N * K3 * 2)); HCHK(hipMalloc(&db, N * 4)); HCHK(hipMalloc(&dy, (size_t)M * N * 4));
        HCHK(hipMemset(dA, 0x3c, (size_t)M * K3 * 2)); HCHK(hipMemset(dW, 0x2c, (size_t)N * K3 * 2)); HCHK(hipMemset(db, 0, N * 4));
        Gemm g = make(h, M, N, K3, db, ws_bytes);
        const float alpha = 1.f / 16384.f, beta = 0.f;
        double best = 1e30; int best_i = -1;
        for (size_t i = 0; i < g.algos.size(); i++) {
            if (g.algos[i].workspaceSize > ws_bytes) continue;
            for (int w = 0; w < 3; w++) BCHK(hipblasLtMatmul(h, g.desc, &alpha, dW, g.la, dA, g.lb, &beta, dy, g.lc, dy, g.lc, &g.algos[i].algo, ws, ws_bytes, st));
            HCHK(hipEventRecord(e0, st));
            for (int w = 0; w < 20; w++) BCHK(hipblasLtMatmul(h, g.desc, &alpha, dW, g.la, dA, g.lb, &beta, dy, g.lc, dy, g.lc, &g.algos[i].algo, ws, ws_bytes, st));
            HCHK(hipEventRecord(e1, st)); HCHK(hipEventSynchronize(e1));
            float ms; HCHK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / 20;
            if (us < best) { best = us; best_i = (int)i; }
        }
        printf("{\"M\": %d, \"N\": %lld, \"K3\": %lld, \"algorithms\": %zu, \"best_us\": %.1f, \"best_index\": %d, \"fp16_TFs\": %.1f}\n", M, (long long)N, (long long)K3,
               g.algos.size(), best, best_i, 2.0 * M * N * K3 / best / 1e6);
        fflush(stdout);
        HCHK(hipFree(dA)); HCHK(hipFree(dW)); HCHK(hipFree(db)); HCHK(hipFree(dy));
    }
    return 0;
}
