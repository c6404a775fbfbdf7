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
// bf16 mode: the packed weights of the call in progress and their layout (the bf16 copies behind the fp32 part)
static thread_local const float* tl_pw = nullptr;
static thread_local PackedLayout tl_layout;
static inline void set_mode(const Dims& d, const PackedLayout& L, const float* pw) {
    tl_bf16 = d.bf16;
    tl_pw = d.bf16 ? pw : nullptr;
    if (d.bf16) tl_layout = L;
    chain_bf16_context(d.bf16 ? pw : nullptr, &L);  // the fused level kernels find their bf16 weight copies through it
}
// Scope of an entry point's mode: set at entry, CLEARED on every exit, so that the thread-local pointer to the packed
// weights never outlives the call (a later direct dvae_chain_level_batch / dvae_gru_level_fwd call from this thread
// must not be switched to bf16 inputs because its fp32 weight happens to lie where an old packed buffer was).
struct ModeGuard {
    ModeGuard(const Dims& d, const PackedLayout& L, const float* pw) { set_mode(d, L, pw); }
    ~ModeGuard() {
        tl_bf16 = 0;
        tl_pw = nullptr;
        chain_bf16_context(nullptr, nullptr);
    }
    ModeGuard(const ModeGuard&) = delete;
    ModeGuard& operator=(const ModeGuard&) = delete;
};
// A product whose B operand is a packed weight matrix reads the bf16 copy instead (half the bytes, no conversion in the
// K loop), and a data gradient (B K-major = the weight as stored) reads the TRANSPOSED bf16 copy, which makes it a
// product of two K-contiguous operands: both then run on gemm_bf16_nn.hip.  Anything the copies cannot serve
// (alignment, a sub-block that does not start on 8 elements, a gathered B) stays as it is.
static inline bool bf16_weight_copies_on() {
    static const int knob = getenv("DVAE_BF16_WEIGHTS") ? atoi(getenv("DVAE_BF16_WEIGHTS")) : 1;
    return knob != 0;
}
// always: also where a small accumulating product would rather split K over the fp32 weight (the caller has no fp32 A)
static inline void use_bf16_weight_copy(dvae_gemm_desc& g, bool always = false) {
    static const int knob = getenv("DVAE_BF16_WEIGHTS") ? atoi(getenv("DVAE_BF16_WEIGHTS")) : 1;  // 0: off (A/B runs)
    const PackedLayout& L = tl_layout;
    if (!knob || !tl_pw || g.a_kmajor || g.b_rows || g.B < tl_pw || g.B >= tl_pw + L.total) return;
    if (g.N % 4 || g.ldc % 4 || (uintptr_t)g.C % 16 || (uintptr_t)g.bias % 16 || (uintptr_t)g.relu_mask % 16 || g.K % 8) return;
    const size_t off = (size_t)(g.B - tl_pw);
    const uint16_t* half = reinterpret_cast<const uint16_t*>(tl_pw + L.half);
    const uint16_t* halfT = reinterpret_cast<const uint16_t*>(tl_pw + L.halfT);
    if (!g.b_kmajor) {
        if (off % 8 || g.ldb % 8) return;
        g.B = reinterpret_cast<const float*>(half + off);
        g.b_bf16 = 1;
        return;
    }
    // (a small accumulating data gradient is better off with split-K over the fp32 weight, gemm_bf16.hip: the
    // K-contiguous kernels do not split K)
    if (!always && g.accumulate && (long)cdiv(g.M, 64) * cdiv(g.N, 64) < 256) return;
    for (int i = 0; i < L.nmats; ++i) {
        const PackedLayout::Mat& m = L.mats[i];
        if (off < m.off || off >= m.off + (size_t)m.rows * m.cols) continue;
        if (g.ldb != m.cols) return;
        const size_t r0 = (off - m.off) / m.cols, c0 = (off - m.off) % m.cols;
        // the block is rows r0 .. r0 + K - 1 (the contraction), columns c0 .. c0 + N - 1 (the output columns) of W
        if (r0 + g.K > (size_t)m.rows || c0 + g.N > (size_t)m.cols || m.rows % 8 || (c0 * m.rows + r0) % 8) return;
        g.B = reinterpret_cast<const float*>(halfT + m.toff + c0 * m.rows + r0);
        g.ldb = m.rows;
        g.b_kmajor = 0;
        g.b_bf16 = 1;
        return;
    }
}
static inline int run_gemm(const dvae_gemm_desc& g, hipStream_t s, int max_split = 0) {
    if (!tl_bf16) return gemm(g, s, max_split);
    {  // products of few rows run in exact fp32 in either mode (gemm_skinny.hip), as the levels of at most 256 rows do
        const int rc = gemm_skinny(g, s, 0);
        if (rc != 1) return rc;
    }
    dvae_gemm_desc h = g;
    use_bf16_weight_copy(h);
    if (h.a_bf16 && !h.b_bf16) return set_err(DVAE_ERR_ARG, "run_gemm: a bf16 activation shadow needs the bf16 weight copy");
    return gemm_bf16(h, s);
}

// Y[M x N] (+)= act(X[M x K] W[N x K]^T + bias)       (forward of a Linear)
// y_bf16 (bf16 mode only): Y is stored as bf16 (Y then points at bf16 words, ldy counts bf16 elements)
// x_h (bf16 mode only, nullable): the bf16 shadow of X (same row pitch in elements): the product reads it instead
static int lin_fwd(const float* X, int ldx, const int* xrows, const float* W, int ldw, const float* bias, float* Y,
                   int ldy, int M, int N, int K, int act, int acc, hipStream_t s, int y_bf16 = 0,
                   const uint16_t* x_h = nullptr) {
    dvae_gemm_desc g = {};
    g.A = x_h ? reinterpret_cast<const float*>(x_h) : X; g.a_bf16 = x_h ? 1 : 0;
    g.lda = ldx; g.a_rows = xrows; g.B = W; g.ldb = ldw; g.C = Y; g.ldc = ldy; g.bias = bias;
    g.M = M; g.N = N; g.K = K; g.act = act; g.accumulate = acc; g.c_bf16 = y_bf16;
    return run_gemm(g, s);
}
// dX[M x K] (+)= dY[M x N] W[N x K]                    (input gradient of a Linear)
// relu_out (nullable, laid out like dX): the layer's input was relu_out = ReLU(.), and dX is masked by relu_out > 0
// dy_h (bf16 mode only, nullable): the bf16 shadow of dY (same row pitch in elements).  Used only when the product
// will run on the transposed bf16 weight copy (two K-contiguous operands); otherwise the fp32 dY is read as before.
// dy_h_only: there IS no complete fp32 dY (its producer stored bf16): the bf16 product or an error.
static int lin_dgrad(const float* dY, int ldy, const float* W, int ldw, float* dX, int ldx, const int* dxrows, int M,
                     int N, int K, int acc, hipStream_t s, const float* relu_out = nullptr, const uint16_t* dy_h = nullptr,
                     bool dy_h_only = false) {
    dvae_gemm_desc g = {};
    g.A = dY; g.lda = ldy; g.B = W; g.ldb = ldw; g.b_kmajor = 1; g.C = dX; g.ldc = ldx; g.c_rows = dxrows;
    g.M = M; g.N = K; g.K = N; g.accumulate = acc; g.relu_mask = relu_out;
    if (dy_h && tl_bf16) {
        if (!dy_h_only) {  // small products stay exact fp32, as in run_gemm
            const int rc = gemm_skinny(g, s, 0);
            if (rc != 1) return rc;
        }
        dvae_gemm_desc h = g;
        use_bf16_weight_copy(h, dy_h_only);
        if (h.b_bf16 && !h.b_kmajor) {  // the transposed copy serves it: A may be the shadow
            h.A = reinterpret_cast<const float*>(dy_h);
            h.a_bf16 = 1;
            return gemm_bf16(h, s);
        }
    }
    if (dy_h_only) return set_err(DVAE_ERR_ARG, "lin_dgrad: a bf16-stored gradient needs the transposed bf16 weight copy");
    return run_gemm(g, s);
}
// dW[N x K] += dY[M x N]^T X[M x K]                    (weight gradient; dW must hold zeros / running sum)
// xrows (nullable) gathers the rows of X; xrows_bound = rows of the stored X the list may name
// dy_h (bf16 mode only, nullable): dY as its producer stored it in bf16 (same pitch in elements) -- the fp32 dY is then NOT
// read (it may be incomplete); x_h (nullable, used only beside dy_h): the bf16 shadow of X.  Large weight-gradient shapes only
// (gemm_bf16.hip's 256 x 256 tile): the caller passes them where the shapes are known to be.
static int lin_wgrad(const float* dY, int ldy, const float* X, int ldx, const int* xrows, float* dW, int ldw, int M,
                     int N, int K, hipStream_t s, int xrows_bound = 0, int max_split = 0, const uint16_t* dy_h = nullptr,
                     const uint16_t* x_h = nullptr) {
    dvae_gemm_desc g = {};
    g.A = dY; g.lda = ldy; g.a_kmajor = 1; g.B = X; g.ldb = ldx; g.b_rows = xrows; g.b_kmajor = 1; g.C = dW; g.ldc = ldw;
    g.M = N; g.N = K; g.K = M; g.accumulate = 1; g.b_rows_bound = xrows_bound;
    if (dy_h) {
        if (!tl_bf16) return set_err(DVAE_ERR_ARG, "lin_wgrad: a bf16-stored gradient outside the bf16 mode");
        g.A = reinterpret_cast<const float*>(dy_h); g.a_bf16 = 1;
        if (x_h) { g.B = reinterpret_cast<const float*>(x_h); g.b_bf16 = 1; }
        return gemm_bf16(g, s);
    }
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
