// fp32 MFMA GEMM for gfx950 (v_mfma_f32_32x32x2_f32: exact f32, k-ordered fma chain).
//
//   C[M x N] (+)= act( A'[M x K] * B'[K x N] + bias[N] )
//
// Each operand is stored either "K-contiguous" (row = m or n index, K along the
// row: activations [rows][feat], nn.Linear weights [out][in]) or "K-major"
// (row = k index: the operand of a weight-gradient product dW = dY^T X, or a
// weight used for dX = dY W).  The three products of a Linear layer map to
//   forward   Y  = X W^T      A K-contig, B K-contig
//   dgrad     dX = dY W       A K-contig, B K-major
//   wgrad     dW = dY^T X     A K-major,  B K-major   (split-K, atomic accumulate)
// Row gather (a_rows / b_rows index the stored matrix's leading dimension) and
// row scatter (c_rows) let callers run a product over an index list of states
// without materialising the gathered copy.
//
// Tile: BM x BN x 32 per 256-thread workgroup (4 waves as WM x WN), each wave
// owns (BM/WM/32) x (BN/WN/32) MFMA tiles of 32x32.  LDS images:
//   K-contig operand: [rows][36]  (32 + 4 pad floats): one ds_read_b128 per lane
//                     gives 4 consecutive k; lane half h takes k = 8g+4h..+3, so
//                     MFMA i of group g contracts k in {8g+i, 8g+4+i}.  The row
//                     stride 36 makes every 16-lane group of the b128 read hit 64
//                     distinct banks.
//   K-major operand:  [32][rows]: ds_read_b32 of 32 consecutive floats per half.
// Both operands present the same k to the same (lane half, MFMA) so the k
// permutation inside a group is consistent.  Global->LDS is register staged
// (next tile's loads in flight during the current tile's MFMAs), two LDS
// buffers, one barrier per K step.
#include "common.h"
#include "dvae_hip.h"

namespace dvae {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmParams {
    const float* A; const float* B; float* C;
    const int* a_rows; const int* b_rows; const int* c_rows;
    const float* bias;
    int lda, ldb, ldc;
    int M, N, K;
    int act;         // 0 none, 1 relu, 2 tanh
    int accumulate;  // C += result
    int split_k;     // >1: atomicAdd into C (C must hold the running sum / zeros)
    int k_per_split; // multiple of 32
};

constexpr int BK = 32;
constexpr int KC_LD = BK + 4;  // K-contiguous LDS row stride (floats)

template <int ROWS, bool KMAJOR>
struct TileLoader {
    // number of float4 units per thread
    static constexpr int UNITS = ROWS * BK / 4 / 256;
    static constexpr int LDS_ELEMS = KMAJOR ? BK * ROWS : ROWS * KC_LD;
    float4 regs[UNITS];

    // K-contig: unit u -> row u/8, c4 u%8 ; global addr = base + ridx(row)*ld + k0 + c4*4
    // K-major : unit u -> k u/(ROWS/4), m4 u%(ROWS/4); addr = base + ridx(k0+k)*ld + r0 + m4*4
    __device__ __forceinline__ void load(const float* __restrict__ base, const int* __restrict__ rows, int ld,
                                         int r0, int rdim, int k0, int kend, int tid) {
#pragma unroll
        for (int i = 0; i < UNITS; ++i) {
            int u = tid + 256 * i;
            if (!KMAJOR) {
                int row = u >> 3, c4 = u & 7;
                int r = r0 + row;
                r = r < rdim ? r : rdim - 1;
                long ridx = rows ? (long)rows[r] : (long)r;
                regs[i] = *reinterpret_cast<const float4*>(base + ridx * ld + k0 + c4 * 4);
            } else {
                int k = u / (ROWS / 4), m4 = u % (ROWS / 4);
                int kk = k0 + k;
                int r = r0 + m4 * 4;
                if (kk < kend && r < rdim) {
                    long ridx = rows ? (long)rows[kk] : (long)kk;
                    regs[i] = *reinterpret_cast<const float4*>(base + ridx * ld + r);
                } else {
                    regs[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
    }
    __device__ __forceinline__ void store(float* lds, int tid) {
#pragma unroll
        for (int i = 0; i < UNITS; ++i) {
            int u = tid + 256 * i;
            if (!KMAJOR) {
                int row = u >> 3, c4 = u & 7;
                *reinterpret_cast<float4*>(lds + row * KC_LD + c4 * 4) = regs[i];
            } else {
                int k = u / (ROWS / 4), m4 = u % (ROWS / 4);
                *reinterpret_cast<float4*>(lds + k * ROWS + m4 * 4) = regs[i];
            }
        }
    }
};

// fragment fetch for one 8-k group g: 4 values (MFMA i = 0..3) for the 32-row
// sub-tile starting at row `roff`
template <int ROWS, bool KMAJOR>
__device__ __forceinline__ float4 frag(const float* lds, int roff, int g, int lane) {
    int h = lane >> 5, r = roff + (lane & 31);
    if (!KMAJOR) {
        return *reinterpret_cast<const float4*>(lds + r * KC_LD + g * 8 + 4 * h);
    } else {
        const float* p = lds + (g * 8 + 4 * h) * ROWS + r;
        return make_float4(p[0], p[ROWS], p[2 * ROWS], p[3 * ROWS]);
    }
}

template <int BM, int BN, int WM, int WN, bool AKM, bool BKM>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmParams p) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert(WM * WN == 4, "4 waves");
    using LA = TileLoader<BM, AKM>;
    using LB = TileLoader<BN, BKM>;
    __shared__ __attribute__((aligned(16))) float lds[2 * (LA::LDS_ELEMS + LB::LDS_ELEMS)];
    constexpr int STAGE = LA::LDS_ELEMS + LB::LDS_ELEMS;  // one stage: A image then B image

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    int kbeg = 0, kend = p.K;
    if (p.split_k > 1) {
        kbeg = blockIdx.z * p.k_per_split;
        kend = min(p.K, kbeg + p.k_per_split);
        if (kbeg >= kend) return;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    LA la; LB lb;
    la.load(p.A, p.a_rows, p.lda, m0, p.M, kbeg, kend, tid);
    lb.load(p.B, p.b_rows, p.ldb, n0, p.N, kbeg, kend, tid);
    la.store(lds, tid);
    lb.store(lds + LA::LDS_ELEMS, tid);
    __syncthreads();

    int cur = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        const bool more = (k0 + BK) < kend;
        if (more) {
            la.load(p.A, p.a_rows, p.lda, m0, p.M, k0 + BK, kend, tid);
            lb.load(p.B, p.b_rows, p.ldb, n0, p.N, k0 + BK, kend, tid);
        }
        const float* a_l = lds + cur * STAGE;
        const float* b_l = a_l + LA::LDS_ELEMS;
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            float4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = frag<BM, AKM>(a_l, (wm * TM + i) * 32, g, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = frag<BN, BKM>(b_l, (wn * TN + j) * 32, g, lane);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (more) {
            la.store(lds + (cur ^ 1) * STAGE, tid);
            lb.store(lds + (cur ^ 1) * STAGE + LA::LDS_ELEMS, tid);
        }
        __syncthreads();
        cur ^= 1;
    }

    // epilogue: D[row][col], col = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int h = lane >> 5, cl = lane & 31;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        int col = n0 + (wn * TN + j) * 32 + cl;
        if (col >= p.N) continue;
        float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row >= p.M) continue;
                long crow = p.c_rows ? (long)p.c_rows[row] : (long)row;
                float* dst = p.C + crow * p.ldc + col;
                float v = acc[i][j][r];
                if (p.split_k > 1) {
                    atomicAdd(dst, v);
                } else {
                    v += bv;
                    if (p.act == 1) v = fmaxf(v, 0.f);
                    else if (p.act == 2) v = tanhf(v);
                    if (p.accumulate) v += *dst;
                    *dst = v;
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN>
static int launch_cfg(const GemmParams& p, int akm, int bkm, hipStream_t s) {
    dim3 grid(cdiv(p.N, BN), cdiv(p.M, BM), p.split_k > 1 ? p.split_k : 1);
    dim3 block(256);
    if (!akm && !bkm) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, false, false>), grid, block, 0, s, p);
    else if (!akm && bkm) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, false, true>), grid, block, 0, s, p);
    else if (akm && bkm) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, true, true>), grid, block, 0, s, p);
    else return set_err(DVAE_ERR_ARG, "gemm: A K-major with B K-contiguous is not built");
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}

int gemm(const dvae_gemm_desc& d, hipStream_t s) {
    DVAE_REQUIRE(d.M >= 0 && d.N >= 0 && d.K >= 0, "gemm: negative size");
    if (d.M == 0 || d.N == 0) return DVAE_OK;
    DVAE_REQUIRE(d.A && d.B && d.C, "gemm: null operand");
    DVAE_REQUIRE(((uintptr_t)d.A % 16) == 0 && ((uintptr_t)d.B % 16) == 0, "gemm: operands must be 16-byte aligned");
    DVAE_REQUIRE(d.lda % 4 == 0 && d.ldb % 4 == 0, "gemm: leading dimensions must be multiples of 4 floats");
    if (!d.a_kmajor) DVAE_REQUIRE(d.K % 32 == 0 && d.lda >= d.K, "gemm: K-contiguous A needs K %% 32 == 0 (K=%d lda=%d)", d.K, d.lda);
    else DVAE_REQUIRE(d.M % 4 == 0 && d.lda >= d.M, "gemm: K-major A needs M %% 4 == 0 (M=%d lda=%d)", d.M, d.lda);
    if (!d.b_kmajor) DVAE_REQUIRE(d.K % 32 == 0 && d.ldb >= d.K, "gemm: K-contiguous B needs K %% 32 == 0 (K=%d ldb=%d)", d.K, d.ldb);
    else DVAE_REQUIRE(d.N % 4 == 0 && d.ldb >= d.N, "gemm: K-major B needs N %% 4 == 0 (N=%d ldb=%d)", d.N, d.ldb);
    DVAE_REQUIRE(d.ldc >= d.N, "gemm: ldc < N");
    DVAE_REQUIRE(d.act >= 0 && d.act <= 2, "gemm: unknown activation %d", d.act);

    GemmParams p;
    p.A = d.A; p.B = d.B; p.C = d.C;
    p.a_rows = d.a_rows; p.b_rows = d.b_rows; p.c_rows = d.c_rows;
    p.bias = d.bias; p.lda = d.lda; p.ldb = d.ldb; p.ldc = d.ldc;
    p.M = d.M; p.N = d.N; p.K = d.K; p.act = d.act; p.accumulate = d.accumulate;
    p.split_k = 1; p.k_per_split = 0;
    if (d.K == 0) {
        // empty contraction: result is bias/act only; run with kend == 0
    }
    // tile choice: big tiles when they fill the chip, small ones otherwise
    long t128 = (long)cdiv(d.M, 128) * cdiv(d.N, 128);
    long t64 = (long)cdiv(d.M, 64) * cdiv(d.N, 64);
    bool use128 = t128 >= 192 || (d.a_kmajor && t128 >= 24);
    int ksteps = cdiv(d.K, BK);
    if (d.a_kmajor && d.b_kmajor && ksteps > 8 && d.accumulate && d.bias == nullptr && d.act == 0) {
        // weight-gradient shape: long contraction, few output tiles -> split K
        long tiles = use128 ? t128 : t64;
        int want = (int)((1024 + tiles - 1) / tiles);
        int maxsplit = ksteps / 4 > 0 ? ksteps / 4 : 1;
        int sk = want < maxsplit ? want : maxsplit;
        if (sk > 1) {
            p.k_per_split = cdiv(ksteps, sk) * BK;
            p.split_k = cdiv(d.K, p.k_per_split);
        }
    }
    if (d.M <= 32 && d.a_kmajor) return launch_cfg<32, 128, 1, 4>(p, d.a_kmajor, d.b_kmajor, s);
    if (use128) return launch_cfg<128, 128, 2, 2>(p, d.a_kmajor, d.b_kmajor, s);
    return launch_cfg<64, 64, 2, 2>(p, d.a_kmajor, d.b_kmajor, s);
}

}  // namespace dvae

extern "C" int dvae_gemm_f32(const dvae_gemm_desc* d, void* stream) {
    if (!d) return dvae::set_err(DVAE_ERR_ARG, "dvae_gemm_f32: null descriptor");
    return dvae::gemm(*d, (hipStream_t)stream);
}
