// fp32 products with FEW ROWS (M up to a few hundred): every forward product and data gradient of a batch-32 train
// step (the reference's own recipe, train.py --batch-size 32: 224-298 state rows, 32 latent rows).
//
//   C[M x N] (+)= act( A[M x K] * B'[K x N] + bias[N] )       A K-contiguous (optionally row-gathered),
//                                                              B K-contiguous ([N][K], a weight) or K-major ([K][N])
//
// gemm_f32_kernel's 64 x 64 tile gives such a product a handful of workgroups (M = 32, N = 128: two), each walking
// K as 16-64 DEPENDENT steps of stage -> barrier -> 16 MFMAs on ONE compute unit: 12-24 us for 4-600 MFLOP, where
// the chip's fp32 matrix rate would need 0.03-4 us (profiles/r03_gemm_skinny.txt).  Here a workgroup owns 16 rows x
// 16 * NT columns and its EIGHT waves are eight interleaved K slices (wave q contracts the 16-k steps q, q + 8, ...),
// operands go from global memory / L2 straight into v_mfma_f32_16x16x4_f32 fragments (lane (i, kg) loads 4
// consecutive k of row i; MFMA j of a step contracts k = 16 s + 4 kg + j; no LDS image, no barrier in the K loop; the
// operands of these products are a few MB and L2-resident), the slice sums meet in LDS in slice order, and NT (or 4)
// waves finish a 16 x 16 block each with 16-byte accesses.  M = 32 x N = 128 x K = 512 is 16 workgroups of 4 steps;
// M = 224, N = 32, K = 1024 (the vertex predictor's second layer: 4 workgroups x 32 steps before) 28 x 8.
//
// Column maps.  K-contiguous B: block t of a tile holds columns n0 + 16 t + i, lane (i, kg) loads B[col][k..k+3] as one
// float4.  K-major B (NT = 4): block t holds the INTERLEAVED columns n0 + 4 i + t, so that lane (i, kg) loads
// B[k][n0 + 4 i .. + 3] -- one float4 = one k of all four blocks -- and after the MFMAs holds 16 consecutive columns
// of its row: 16 lanes read 256 contiguous bytes per k instead of four 64-byte pieces.
//
// Another K order than gemm_f32_kernel's (agrees to rounding, not bit for bit), the same order whatever M and N are.
#include "gemm_common.h"

namespace dvae {

typedef float f32x4s __attribute__((ext_vector_type(4)));
constexpr int SKINNY_KS = 8;  // waves = K slices

__device__ __forceinline__ float4 sk_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int NT, bool BKM>
__global__ __launch_bounds__(512) void gemm_skinny_kernel(GemmParams p) {
    static_assert(!BKM || NT == 4 || NT == 1, "K-major B: one block (dword loads) or four interleaved blocks (float4 loads)");
    __shared__ float4 xch[SKINNY_KS * NT * 64];
    const int tid = threadIdx.x, lane = tid & 63, kq = tid >> 6;
    const int li = lane & 15, kg = lane >> 4;
    // XCD-contiguous runs of tiles (workgroup ids equal mod 8 share an L2); inside a run the row tiles of one column
    // tile follow each other, so an XCD reads 1/8 of B and all of the (small) A
    const int ntiles = p.tiles_x * p.tiles_y;
    int id;
    {
        const int q = ntiles >> 3, r = ntiles & 7, xcd = blockIdx.x & 7;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int m0 = (id % p.tiles_y) * 16, n0 = (id / p.tiles_y) * 16 * NT;

    int row = m0 + li;
    row = row < p.M ? row : p.M - 1;  // clamped rows are computed, never stored
    const float* pa = p.A + (p.a_rows ? (long)p.a_rows[row] : (long)row) * p.lda + 4 * kg;
    // B: K-contiguous: one row pointer per block; K-major: the lane's column (4 consecutive ones for NT = 4)
    const float* pb[NT];
    if constexpr (!BKM) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            int col = n0 + 16 * t + li;
            col = col < p.N ? col : p.N - 1;
            pb[t] = p.B + (long)col * p.ldb + 4 * kg;
        }
    } else {
        int col = NT == 4 ? n0 + 4 * li : n0 + li;
        col = col < p.N ? col : (NT == 4 ? p.N - 4 : p.N - 1);  // (N % 4 == 0: checked by the launcher)
        pb[0] = p.B + (long)(4 * kg) * p.ldb + col;
    }

    f32x4s acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4s{0.f, 0.f, 0.f, 0.f};
    struct Frag { float4 a; float4 b[BKM ? (NT == 4 ? 4 : 1) : NT]; };
    const int ns = p.K / 16;
    auto load = [&](Frag& f, int st) {
        const bool ok = st < ns;
        const int k = ok ? 16 * st : 0;  // past the end: a valid address, the A part is zero
        const float4 a = sk_ld4(pa + k);
        f.a = ok ? a : make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (!BKM) {
#pragma unroll
            for (int t = 0; t < NT; ++t) f.b[t] = sk_ld4(pb[t] + k);
        } else if constexpr (NT == 4) {
            const float* b = pb[0] + (long)k * p.ldb;
#pragma unroll
            for (int j = 0; j < 4; ++j) f.b[j] = sk_ld4(b + (long)j * p.ldb);  // k = 16 s + 4 kg + j, four blocks' columns
        } else {
            const float* b = pb[0] + (long)k * p.ldb;
            f.b[0] = make_float4(b[0], b[p.ldb], b[2 * (long)p.ldb], b[3 * (long)p.ldb]);
        }
    };
    auto mma = [&](const Frag& f) {
        if constexpr (!BKM || NT == 1) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.b[t].x, f.a.x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.b[t].y, f.a.y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.b[t].z, f.a.z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.b[t].w, f.a.w, acc[t], 0, 0, 0);
            }
        } else {
            const float av[4] = {f.a.x, f.a.y, f.a.z, f.a.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.b[j].x, av[j], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.b[j].y, av[j], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.b[j].z, av[j], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.b[j].w, av[j], acc[3], 0, 0, 0);
            }
        }
    };
    // two steps in flight ahead of the one being contracted (fixed register sets)
    Frag f0, f1, f2;
    load(f0, kq);
    load(f1, kq + SKINNY_KS);
    for (int st = kq; st < ns; st += 3 * SKINNY_KS) {
        load(f2, st + 2 * SKINNY_KS);
        mma(f0);
        if (st + SKINNY_KS < ns) {
            load(f0, st + 3 * SKINNY_KS);
            mma(f1);
        }
        if (st + 2 * SKINNY_KS < ns) {
            load(f1, st + 4 * SKINNY_KS);
            mma(f2);
        }
    }
    // slice sums through LDS.  acc[t][r] = C[row li][block t, column index 4 kg + r]:
    //   K-contiguous B (and one block): the quad over r is 4 consecutive columns n0 + 16 t + 4 kg ..; wave t finishes block t
    //   K-major B, NT = 4: the quad over t is 4 consecutive columns n0 + 16 kg + 4 r ..; stored transposed, wave r finishes it
    constexpr bool INTERLEAVED = BKM && NT == 4;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if constexpr (!INTERLEAVED) xch[(kq * NT + t) * 64 + lane] = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
        else xch[(kq * NT + t) * 64 + lane] = make_float4(acc[0][t], acc[1][t], acc[2][t], acc[3][t]);
    }
    __syncthreads();
    const int orow = m0 + li;
    if (kq < NT && orow < p.M) {
        const int col = INTERLEAVED ? n0 + 16 * kg + 4 * kq : n0 + 16 * kq + 4 * kg;
        if (col < p.N) {  // N % 4 == 0: a quad is inside or outside as a whole
            float4 v = xch[kq * 64 + lane];
#pragma unroll
            for (int k = 1; k < SKINNY_KS; ++k) {  // K slices in order
                const float4 t = xch[(k * NT + kq) * 64 + lane];
                v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            }
            const long crow = p.c_rows ? (long)p.c_rows[orow] : (long)orow;
            if (p.bias) {
                const float4 b = sk_ld4(p.bias + col);
                v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
            }
            if (p.act == 1) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            } else if (p.act == 2) {
                v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w);
            }
            if (p.relu_mask) {
                const float4 m = sk_ld4(p.relu_mask + crow * p.ldc + col);
                v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f;
                v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
            }
            float4* dst = reinterpret_cast<float4*>(p.C + crow * p.ldc + col);
            if (p.accumulate) {
                const float4 o = *dst;
                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
            *dst = v;
        }
    }
}

template <int NT, bool BKM>
static int launch_skinny(GemmParams p, hipStream_t s) {
    p.tiles_x = cdiv(p.N, 16 * NT);
    p.tiles_y = cdiv(p.M, 16);
    hipLaunchKernelGGL((gemm_skinny_kernel<NT, BKM>), dim3((unsigned)(p.tiles_x * p.tiles_y)), dim3(512), 0, s, p);
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}

// Serves d if it is a product this kernel is built for and faster at; returns 1 when it is not (the caller then runs
// gemm_f32_kernel), DVAE_OK or an error otherwise.  Faster at = SMALL products: measured on the shapes of a batch-32
// step (profiles/r03_gemm_skinny.txt) it wins up to ~1.2e8 multiply-adds (32 x 128 x 512: 11.8 -> 4.7 us; 224 x 32 x
// 1024: 20.9 -> 6.3; 224 x 1024 x 512: 12.1 -> 8.8) and loses from ~3e8 on (298 x 2048 x 512: 13 -> 22 us: a 16-row
// tile re-reads B M / 16 times from L2 where the 64 x 64 LDS tile re-reads it M / 64 times).
// DVAE_GEMM_SKINNY_MMAC: limit in millions of multiply-adds (default 160, 0 = never);
// DVAE_GEMM_SKINNY_NT = 1 / 2 / 4 forces the blocks per tile.
static int skinny_mmac_default() {
    const char* v = getenv("DVAE_GEMM_SKINNY_MMAC");
    return v ? atoi(v) : 160;
}
static int g_skinny_mmac = -1;  // -1: not read yet (dvae_debug_set_gemm_skinny_mmac overrides the environment)

int gemm_skinny(const dvae_gemm_desc& d, hipStream_t s, int force_nt) {
    if (g_skinny_mmac < 0) g_skinny_mmac = skinny_mmac_default();
    const int knob_m = g_skinny_mmac;
    static const int knob_nt = [] { const char* v = getenv("DVAE_GEMM_SKINNY_NT"); return v ? atoi(v) : 0; }();
    static const int knob_log = [] { const char* v = getenv("DVAE_GEMM_LOG"); return v ? atoi(v) : 0; }();
    if ((!force_nt && (double)d.M * d.N * d.K > 1e6 * knob_m) || d.a_kmajor || d.b_rows || d.a_bf16 || d.b_bf16 || d.c_bf16 || d.M <= 0 || d.N <= 0) return 1;
    if (d.K < 16 || d.K % 16 || d.N % 4 || d.ldc % 4 || d.lda % 4 || d.ldb % 4) return 1;
    if ((uintptr_t)d.A % 16 || (uintptr_t)d.B % 16 || (uintptr_t)d.C % 16 || (uintptr_t)d.bias % 16 || (uintptr_t)d.relu_mask % 16) return 1;
    if (d.b_kmajor && d.N < 4) return 1;
    GemmParams p;
    p.A = d.A; p.B = d.B; p.C = d.C;
    p.a_rows = d.a_rows; p.b_rows = nullptr; p.c_rows = d.c_rows;
    p.bias = d.bias; p.relu_mask = d.relu_mask; p.lda = d.lda; p.ldb = d.ldb; p.ldc = d.ldc;
    p.M = d.M; p.N = d.N; p.K = d.K; p.act = d.act; p.accumulate = d.accumulate;
    p.split_k = 1; p.k_per_split = 0; p.c_bf16 = 0;
    // blocks per tile: as many workgroups as the product has 16 x 16 blocks until that is more than the chip holds at
    // once (a tile of NT blocks re-reads its A rows once instead of NT times)
    const long blocks = (long)cdiv(d.M, 16) * cdiv(d.N, 16);
    int nt = d.b_kmajor ? (blocks <= 512 ? 1 : 4) : (blocks <= 256 ? 1 : blocks <= 640 ? 2 : 4);
    const int want = force_nt ? force_nt : knob_nt;
    if (want == 1 || want == 4 || (want == 2 && !d.b_kmajor)) nt = want;
    if (knob_log)
        fprintf(stderr, "GEMMSEL skinny16x%d %d %d %d akm=0 bkm=%d split=1\n", 16 * nt, d.M, d.N, d.K, d.b_kmajor);
    if (d.b_kmajor) return nt == 1 ? launch_skinny<1, true>(p, s) : launch_skinny<4, true>(p, s);
    return nt == 1 ? launch_skinny<1, false>(p, s) : nt == 2 ? launch_skinny<2, false>(p, s) : launch_skinny<4, false>(p, s);
}

}  // namespace dvae

extern "C" int dvae_debug_set_gemm_skinny_mmac(int mmac) {
    dvae::g_skinny_mmac = mmac < 0 ? dvae::skinny_mmac_default() : mmac;
    return DVAE_OK;
}

extern "C" int dvae_gemm_skinny_f32(const dvae_gemm_desc* d, int nt, void* stream) {
    if (!d) return dvae::set_err(DVAE_ERR_ARG, "dvae_gemm_skinny_f32: null descriptor");
    DVAE_REQUIRE(nt == 0 || nt == 1 || nt == 2 || nt == 4, "dvae_gemm_skinny_f32: nt must be 0 (auto), 1, 2 or 4");
    DVAE_TRY(dvae::gemm_check(*d, 16));
    const int rc = dvae::gemm_skinny(*d, (hipStream_t)stream, nt ? nt : -1);
    if (rc == 1) return dvae::set_err(DVAE_ERR_ARG, "dvae_gemm_skinny_f32: not a product this kernel serves (A K-contiguous, no B gather, K %% 16 == 0, N %% 4 == 0, 16-byte aligned operands)");
    return rc;
}
