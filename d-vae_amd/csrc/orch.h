// Helpers shared by the orchestration files (train.hip, decode.hip): the three products of a
// Linear layer on top of the GEMM entry points, the matrix-unit input type of the call in
// progress, and the workspace bump allocator.  Header-only; every translation unit gets its own
// copy of the (thread-local) mode flag and sets it at its entry points.
#pragma once
#include "common.h"
#include "model.h"

namespace dvae {

// Matrix-unit input type of the call in progress (dvae_model_desc.mfma_bf16), set by every
// entry point before it enqueues anything.
static thread_local int tl_bf16 = 0;
static inline int run_gemm(const dvae_gemm_desc& g, hipStream_t s, int max_split = 0) {
    return tl_bf16 ? gemm_bf16(g, s) : gemm(g, s, max_split);
}

// Y[M x N] (+)= act(X[M x K] W[N x K]^T + bias)       (forward of a Linear)
static int lin_fwd(const float* X, int ldx, const int* xrows, const float* W, int ldw, const float* bias, float* Y,
                   int ldy, int M, int N, int K, int act, int acc, hipStream_t s) {
    dvae_gemm_desc g = {};
    g.A = X; g.lda = ldx; g.a_rows = xrows; g.B = W; g.ldb = ldw; g.C = Y; g.ldc = ldy; g.bias = bias;
    g.M = M; g.N = N; g.K = K; g.act = act; g.accumulate = acc;
    return run_gemm(g, s);
}
// dX[M x K] (+)= dY[M x N] W[N x K]                    (input gradient of a Linear)
// relu_out (nullable, laid out like dX): the layer's input was relu_out = ReLU(.), and dX is masked by relu_out > 0
static int lin_dgrad(const float* dY, int ldy, const float* W, int ldw, float* dX, int ldx, const int* dxrows, int M,
                     int N, int K, int acc, hipStream_t s, const float* relu_out = nullptr) {
    dvae_gemm_desc g = {};
    g.A = dY; g.lda = ldy; g.B = W; g.ldb = ldw; g.b_kmajor = 1; g.C = dX; g.ldc = ldx; g.c_rows = dxrows;
    g.M = M; g.N = K; g.K = N; g.accumulate = acc; g.relu_mask = relu_out;
    return run_gemm(g, s);
}
// dW[N x K] += dY[M x N]^T X[M x K]                    (weight gradient; dW must hold zeros / running sum)
// xrows (nullable) gathers the rows of X; xrows_bound = rows of the stored X the list may name
static int lin_wgrad(const float* dY, int ldy, const float* X, int ldx, const int* xrows, float* dW, int ldw, int M,
                     int N, int K, hipStream_t s, int xrows_bound = 0, int max_split = 0) {
    dvae_gemm_desc g = {};
    g.A = dY; g.lda = ldy; g.a_kmajor = 1; g.B = X; g.ldb = ldx; g.b_rows = xrows; g.b_kmajor = 1; g.C = dW; g.ldc = ldw;
    g.M = N; g.N = K; g.K = M; g.accumulate = 1; g.b_rows_bound = xrows_bound;
    return run_gemm(g, s, max_split);
}

struct Bump {
    float* base;
    size_t off;
    float* take(size_t n) {
        float* p = base ? base + off : nullptr;
        off += (n + 63) & ~(size_t)63;
        return p;
    }
};

}  // namespace dvae
