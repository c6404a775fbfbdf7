// Level kernels of the state chain (fp32 mode): the hidden product of a level AND its pointwise
// tail in ONE launch, so the [rows x 3Hp] / [rows x 2Hp] pre-activations never exist in HBM and a
// level of the forward chain is 2 launches instead of 4.
//
//   chain_level_kernel<3>:  (r, z, n, W_hn h + b_hn, h') = GRUCell tail of  h_prev W_hh^T
//   chain_level_kernel<2>:  (g, m, msg = g*m, next level's predecessor sum) of  h [Wg | Wm]^T
//
// Why a kernel of its own instead of an epilogue of gemm_f32_kernel: the tail needs the r, z and n
// (or g and m) pre-activations of one (row, hidden unit) in ONE lane, i.e. columns j, Hp + j,
// 2Hp + j of the product in one wave.  A workgroup therefore owns 64 rows x 32 hidden units x all
// NGATE gates (a 64 x 96 or 64 x 64 tile whose B rows come from NGATE row blocks of the weight), and
// its 4 waves are (K half) x (row half): each wave contracts HALF of every K step for its 32 rows
// and all gates, the two K halves meet through LDS in the epilogue, and each wave then finishes
// half of the columns.  At the chain's shape (1024 rows, Hp = 512) that is 256 workgroups = one per
// CU with all four SIMDs busy for exactly K/2 MFMAs per gate -- where the 64 x 64 tiles of the
// generic kernel need 1.5 (GRU) rounds of the chip.
//
// MFMA: v_mfma_f32_32x32x2_f32 (exact fp32), accumulated transposed (operands swapped) so a lane
// ends up with 4 consecutive hidden units of one row per register quad -> 16-byte loads / stores in
// the tail.  LDS images as in gemm.hip: [rows][36] floats, ds_read_b128 fragments, two stages,
// register-staged global loads one tile ahead, one barrier per K step.
#include "gemm_common.h"

namespace dvae {

constexpr int CBM = 64;        // rows per workgroup
constexpr int CBH = 32;        // hidden units per workgroup (times NGATE product columns)
constexpr int CBK = 32;        // K step
constexpr int CLD = CBK + 4;   // LDS row stride (floats)

struct ChainParams {
    const float* A; int lda;   // [M][K] states (h_prev / h of the level)
    const float* W; int ldw;   // [NGATE * Hp][K] weight, K contiguous
    int M, Hp, K;
    int tiles_x, tiles_y;      // Hp / 32, ceil(M / 64)
    // GRU tail
    const float* gi_tab; const int* types; const float* bhn; const float* hprev; int ld_hprev;
    float* gates; float* ghn; float* hout; int ld_hout;
    // gate / mapper tail
    const float* gv; float* gm; float* msg; int ld_msg; int cur_row0;
    float* next_agg; int ld_agg; const int* next_off; const int* idx;
};

__device__ __forceinline__ float4 cld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void cst4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// GRUCell tail for 4 consecutive hidden units of one row (same arithmetic as gru_fwd_kernel)
__device__ __forceinline__ void gru_tail(const ChainParams& p, int row, int col, float4 ghr, float4 ghz, float4 ghn) {
    const float* gi = p.gi_tab + (long)p.types[row] * 3 * p.Hp;
    const float4 gir = cld4(gi + col), giz = cld4(gi + p.Hp + col), gin = cld4(gi + 2 * p.Hp + col);
    const float4 bn = cld4(p.bhn + col);
    const float4 hp = cld4(p.hprev + (long)row * p.ld_hprev + col);
    float4 R, Z, N, Hn, GN;
#define GRU1(f)                                          \
    {                                                    \
        float gn = ghn.f + bn.f;                         \
        float rr = 1.f / (1.f + expf(-(gir.f + ghr.f))); \
        float zz = 1.f / (1.f + expf(-(giz.f + ghz.f))); \
        float nn = tanhf(gin.f + rr * gn);               \
        R.f = rr; Z.f = zz; N.f = nn; GN.f = gn;         \
        Hn.f = (1.f - zz) * nn + zz * hp.f;              \
    }
    GRU1(x) GRU1(y) GRU1(z) GRU1(w)
#undef GRU1
    float* g = p.gates + (long)row * 3 * p.Hp;
    cst4(g + col, R);
    cst4(g + p.Hp + col, Z);
    cst4(g + 2 * p.Hp + col, N);
    cst4(p.ghn + (long)row * p.Hp + col, GN);
    cst4(p.hout + (long)row * p.ld_hout + col, Hn);
}

// gate/mapper tail + the next level's predecessor sum (same arithmetic and summation order as
// gated_fwd_gather_kernel)
__device__ __forceinline__ void gate_tail(const ChainParams& p, int row, int col, float4 pg, float4 pm) {
    const float4 vg = cld4(p.gv + col), vm = cld4(p.gv + p.Hp + col);
    const float4 a = make_float4(pg.x + vg.x, pg.y + vg.y, pg.z + vg.z, pg.w + vg.w);
    const float4 m = make_float4(pm.x + vm.x, pm.y + vm.y, pm.z + vm.z, pm.w + vm.w);
    const float4 g = make_float4(1.f / (1.f + expf(-a.x)), 1.f / (1.f + expf(-a.y)), 1.f / (1.f + expf(-a.z)),
                                 1.f / (1.f + expf(-a.w)));
    const float4 own = make_float4(g.x * m.x, g.y * m.y, g.z * m.z, g.w * m.w);
    float* o = p.gm + (long)row * 2 * p.Hp;
    cst4(o + col, g);
    cst4(o + p.Hp + col, m);
    cst4(p.msg + (long)(p.cur_row0 + row) * p.ld_msg + col, own);
    if (p.next_agg) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = p.next_off[row]; k < p.next_off[row + 1]; ++k) {
            const int s = p.idx[k];
            const float4 v = s == p.cur_row0 + row ? own : cld4(p.msg + (long)s * p.ld_msg + col);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        cst4(p.next_agg + (long)row * p.ld_agg + col, acc);
    }
}

// Epilogue of the wave with K half KH: hand the partner wave (other K half, same rows) the two
// register quads it finishes, take its partial sums of quads 2KH, 2KH+1, run the tail on them.
template <int NGATE, int KH>
__device__ __forceinline__ void chain_epilogue(const ChainParams& p, f32x16 (&acc)[NGATE], float* lds, int m0, int n0,
                                               int wm, int lane) {
    float4* xch = reinterpret_cast<float4*>(lds);
    constexpr int MINE = 2 * KH, THEIRS = 2 * (1 - KH);
#pragma unroll
    for (int i = 0; i < NGATE; ++i)
#pragma unroll
        for (int ql = 0; ql < 2; ++ql) {
            const int q = THEIRS + ql;
            xch[(((KH * 2 + wm) * NGATE + i) * 2 + ql) * 64 + lane] =
                make_float4(acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]);
        }
    __syncthreads();
    const int row = m0 + wm * 32 + (lane & 31), h = lane >> 5;
    float4 v[NGATE][2];
#pragma unroll
    for (int i = 0; i < NGATE; ++i)
#pragma unroll
        for (int ql = 0; ql < 2; ++ql) {
            const int q = MINE + ql;
            const float4 o = xch[((((1 - KH) * 2 + wm) * NGATE + i) * 2 + ql) * 64 + lane];
            // (K half 0) + (K half 1): fp32 addition commutes, so both waves form the same sum
            v[i][ql] = make_float4(acc[i][4 * q] + o.x, acc[i][4 * q + 1] + o.y, acc[i][4 * q + 2] + o.z,
                                   acc[i][4 * q + 3] + o.w);
        }
    if (row >= p.M) return;
#pragma unroll
    for (int ql = 0; ql < 2; ++ql) {
        const int col = n0 + 8 * (MINE + ql) + 4 * h;
        if (NGATE == 3) gru_tail(p, row, col, v[0][ql], v[1][ql], v[NGATE - 1][ql]);
        else gate_tail(p, row, col, v[0][ql], v[1][ql]);
    }
}

template <int NGATE>
__global__ __launch_bounds__(256) void chain_level_kernel(ChainParams p) {
    constexpr int STAGE = (CBM + NGATE * 32) * CLD;  // A image then the NGATE weight images
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];
    static_assert(2 * STAGE * 4 >= 4 * NGATE * 2 * 64 * 16, "the K-half exchange reuses the stage buffers");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kh = wave >> 1, wm = wave & 1;
    int m0, n0;
    tile_origin<CBM, CBH>(blockIdx.x, p.tiles_x, p.tiles_x * p.tiles_y, m0, n0);

    // register-staged loaders: thread -> row tid/8 (+32 for the second A unit / per gate), 16 bytes at (tid%8)*4
    const int lr = tid >> 3, lc = (tid & 7) * 4;
    const float *pa0, *pa1, *pw0, *pw1, *pw2;
    {
        int r = m0 + lr;
        r = r < p.M ? r : p.M - 1;  // clamped rows are computed but never stored
        pa0 = p.A + (long)r * p.lda + lc;
        r = m0 + lr + 32;
        r = r < p.M ? r : p.M - 1;
        pa1 = p.A + (long)r * p.lda + lc;
        pw0 = p.W + (long)(n0 + lr) * p.ldw + lc;
        pw1 = pw0 + (long)p.Hp * p.ldw;
        pw2 = pw1 + (long)p.Hp * p.ldw;
    }
    const int so = lr * CLD + lc;
    float4 ra0, ra1, rw0, rw1, rw2;
    auto gload = [&](int k0) {
        ra0 = cld4(pa0 + k0);
        ra1 = cld4(pa1 + k0);
        rw0 = cld4(pw0 + k0);
        rw1 = cld4(pw1 + k0);
        if (NGATE == 3) rw2 = cld4(pw2 + k0);
    };
    auto lstore = [&](float* st) {
        cst4(st + so, ra0);
        cst4(st + so + 32 * CLD, ra1);
        cst4(st + so + CBM * CLD, rw0);
        cst4(st + so + (CBM + 32) * CLD, rw1);
        if (NGATE == 3) cst4(st + so + (CBM + 64) * CLD, rw2);
    };

    f32x16 acc[NGATE];
#pragma unroll
    for (int i = 0; i < NGATE; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const int nk = p.K / CBK;
    gload(0);
    lstore(lds);
    if (nk > 1) gload(CBK);
    __syncthreads();
    // fragment base of this wave: row (lane & 31) of its row half / of each gate image, k offset of
    // its K half (groups 2kh, 2kh+1 of 8 k each) and lane half
    const int fo = (lane & 31) * CLD + kh * 16 + 4 * (lane >> 5);
    for (int t = 0; t < nk; ++t) {
        const float* cs = lds + (t & 1) * STAGE;
        float* ns = lds + ((t & 1) ^ 1) * STAGE;
        float4 fa[2], fw[2][NGATE];
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
            fa[gg] = cld4(cs + wm * 32 * CLD + fo + gg * 8);
#pragma unroll
            for (int i = 0; i < NGATE; ++i) fw[gg][i] = cld4(cs + (CBM + i * 32) * CLD + fo + gg * 8);
        }
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
#pragma unroll
            for (int i = 0; i < NGATE; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fw[gg][i].x, fa[gg].x, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NGATE; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fw[gg][i].y, fa[gg].y, acc[i], 0, 0, 0);
            if (gg == 0 && t + 1 < nk) {
                // tile t+1 (in flight since the previous step): registers -> the other stage; reload from tile t+2
                lstore(ns);
                if (t + 2 < nk) gload((t + 2) * CBK);
            }
#pragma unroll
            for (int i = 0; i < NGATE; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fw[gg][i].z, fa[gg].z, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NGATE; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fw[gg][i].w, fa[gg].w, acc[i], 0, 0, 0);
        }
        __syncthreads();  // stage t^1 complete; every wave holds its fragments of stage t in registers
    }
    if (kh == 0) chain_epilogue<NGATE, 0>(p, acc, lds, m0, n0, wm, lane);
    else chain_epilogue<NGATE, 1>(p, acc, lds, m0, n0, wm, lane);
}

static int chain_check(const float* A, int lda, const float* W, int rows, int Hp) {
    DVAE_REQUIRE(rows >= 0 && Hp > 0 && Hp % 32 == 0, "chain level: Hp must be a positive multiple of 32 (Hp=%d)", Hp);
    DVAE_REQUIRE(A && W, "chain level: null operand");
    DVAE_REQUIRE(lda >= Hp && lda % 4 == 0, "chain level: bad leading dimension %d", lda);
    DVAE_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0, "chain level: operands must be 16-byte aligned");
    return DVAE_OK;
}

template <int NGATE>
static int chain_launch(ChainParams& p, hipStream_t s) {
    p.K = p.Hp;
    p.ldw = p.Hp;
    p.tiles_x = p.Hp / CBH;
    p.tiles_y = cdiv(p.M, CBM);
    hipLaunchKernelGGL((chain_level_kernel<NGATE>), dim3((unsigned)(p.tiles_x * p.tiles_y)), dim3(256), 0, s, p);
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}


// ------------------------------------------------------------------------------------------
// Backward twin: the input gradient of a level's product, dX[M x Hp] += dY[M x K] W[K x Hp]
// (W = the nn.Linear weight [out = K][in = Hp], i.e. K-major for this contraction), with the same
// balance -- 64 rows x 32 columns per workgroup, the two wave pairs each contract half of every
// 64-deep K step, so M = 1024, Hp = 512 is again 256 workgroups with all four SIMDs busy -- where
// the generic kernel runs these K = 2Hp / 3Hp, N = Hp products as split-K with fp32 atomics.
//   TAIL 0: dX += product                                   (d agg += d gh W_hh)
//   TAIL 1: d h = dX + product, then the GRUCell pointwise backward of d h in the same launch
//           (d h += d gm [Wg | Wm]; gru_bwd_kernel's arithmetic)
// The result is deterministic (no atomics).
constexpr int DBK = 64;        // K step (two halves of 32)
constexpr int DLA = DBK + 4;   // LDS row stride of the dY image [64 rows][64 k]
constexpr int DLW = 40;        // LDS row stride of the W image [64 k][32 cols]: lane halves 4 k rows apart hit disjoint banks

struct ChainBwdParams {
    const float* A; int lda;   // dY [M][K]
    const float* W; int ldw;   // [K][Hp]
    float* C; int ldc;         // dX [M][Hp], read and written
    int M, Hp, K;
    int tiles_x, tiles_y;
    // GRU backward tail
    const float* gates; const float* ghn; const float* hprev; int ld_hprev;
    float* dgh; float* dgin; float* dhprev; int ld_dhprev;
};

// same arithmetic as gru_bwd_kernel for 4 consecutive hidden units of one row
__device__ __forceinline__ void gru_bwd_tail(const ChainBwdParams& p, int row, int col, float4 D) {
    const float* g = p.gates + (long)row * 3 * p.Hp;
    const float4 R = cld4(g + col), Z = cld4(g + p.Hp + col), N = cld4(g + 2 * p.Hp + col);
    const float4 GN = cld4(p.ghn + (long)row * p.Hp + col);
    const float4 hp = cld4(p.hprev + (long)row * p.ld_hprev + col);
    float4 dR, dZ, dGN, dPN, dHP;
#define GRUB(f)                             \
    {                                       \
        float d = D.f;                      \
        float dn = d * (1.f - Z.f);         \
        float dz = d * (hp.f - N.f);        \
        float dpn = dn * (1.f - N.f * N.f); \
        float dr = dpn * GN.f;              \
        dGN.f = dpn * R.f;                  \
        dPN.f = dpn;                        \
        dR.f = dr * R.f * (1.f - R.f);      \
        dZ.f = dz * Z.f * (1.f - Z.f);      \
        dHP.f = d * Z.f;                    \
    }
    GRUB(x) GRUB(y) GRUB(z) GRUB(w)
#undef GRUB
    float* o = p.dgh + (long)row * 3 * p.Hp;
    cst4(o + col, dR);
    cst4(o + p.Hp + col, dZ);
    cst4(o + 2 * p.Hp + col, dGN);
    cst4(p.dgin + (long)row * p.Hp + col, dPN);
    cst4(p.dhprev + (long)row * p.ld_dhprev + col, dHP);
}

template <int TAIL, int KH>
__device__ __forceinline__ void chain_bwd_epilogue(const ChainBwdParams& p, f32x16& acc, float* lds, int m0, int n0,
                                                   int wm, int lane) {
    float4* xch = reinterpret_cast<float4*>(lds);
    constexpr int MINE = 2 * KH, THEIRS = 2 * (1 - KH);
#pragma unroll
    for (int ql = 0; ql < 2; ++ql) {
        const int q = THEIRS + ql;
        xch[((KH * 2 + wm) * 2 + ql) * 64 + lane] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    }
    __syncthreads();
    const int row = m0 + wm * 32 + (lane & 31), h = lane >> 5;
    if (row >= p.M) return;
#pragma unroll
    for (int ql = 0; ql < 2; ++ql) {
        const int q = MINE + ql;
        const float4 o = xch[(((1 - KH) * 2 + wm) * 2 + ql) * 64 + lane];
        const int col = n0 + 8 * q + 4 * h;
        float* c = p.C + (long)row * p.ldc + col;
        const float4 c0 = cld4(c);
        // (K half 0 + K half 1) + dX: the same value whichever wave forms it
        const float4 v = make_float4(c0.x + (acc[4 * q] + o.x), c0.y + (acc[4 * q + 1] + o.y), c0.z + (acc[4 * q + 2] + o.z),
                                     c0.w + (acc[4 * q + 3] + o.w));
        cst4(c, v);
        if (TAIL == 1) gru_bwd_tail(p, row, col, v);
    }
}

template <int TAIL>
__global__ __launch_bounds__(256) void chain_dgrad_kernel(ChainBwdParams p) {
    constexpr int STAGE = CBM * DLA + DBK * DLW;  // dY image then W image
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kh = wave >> 1, wm = wave & 1;
    int m0, n0;
    tile_origin<CBM, CBH>(blockIdx.x, p.tiles_x, p.tiles_x * p.tiles_y, m0, n0);

    // dY: thread -> rows tid/16 + 16 i (i = 0..3), 16 bytes at k (tid%16)*4;  W: k rows tid/8 + 32 i, cols (tid%8)*4
    const int ar = tid >> 4, ak = (tid & 15) * 4;
    const int wk = tid >> 3, wc = (tid & 7) * 4;
    const float *pa0, *pa1, *pa2, *pa3;
    {
        auto rowp = [&](int i) {
            int r = m0 + ar + 16 * i;
            r = r < p.M ? r : p.M - 1;  // clamped rows are computed but never stored
            return p.A + (long)r * p.lda;
        };
        pa0 = rowp(0); pa1 = rowp(1); pa2 = rowp(2); pa3 = rowp(3);
    }
    const float* pwc = p.W + n0 + wc;
    const int sa = ar * DLA + ak, sw = CBM * DLA + wk * DLW + wc;
    float4 ra0, ra1, ra2, ra3, rw0, rw1;
    // K % 64 may be 32: the second half of the last step then lies past K; its loads are redirected
    // to valid addresses and the K-half-1 waves skip that half (below)
    auto gload = [&](int k0) {
        int ka = k0 + ak;
        ka = ka < p.K ? ka : ka - 32;
        ra0 = cld4(pa0 + ka);
        ra1 = cld4(pa1 + ka);
        ra2 = cld4(pa2 + ka);
        ra3 = cld4(pa3 + ka);
        const int k1 = k0 + wk + 32;
        rw0 = cld4(pwc + (long)(k0 + wk) * p.ldw);
        rw1 = cld4(pwc + (long)(k1 < p.K ? k1 : k1 - 32) * p.ldw);
    };
    auto lstore = [&](float* st) {
        cst4(st + sa, ra0);
        cst4(st + sa + 16 * DLA, ra1);
        cst4(st + sa + 32 * DLA, ra2);
        cst4(st + sa + 48 * DLA, ra3);
        cst4(st + sw, rw0);
        cst4(st + sw + 32 * DLW, rw1);
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    const int nk = (p.K + DBK - 1) / DBK;
    gload(0);
    lstore(lds);
    if (nk > 1) gload(DBK);
    __syncthreads();
    const int foa = (wm * 32 + (lane & 31)) * DLA + kh * 32 + 4 * (lane >> 5);
    const int fow = CBM * DLA + (kh * 32 + 4 * (lane >> 5)) * DLW + (lane & 31);
    for (int t = 0; t < nk; ++t) {
        const float* cs = lds + (t & 1) * STAGE;
        float* ns = lds + ((t & 1) ^ 1) * STAGE;
        const bool active = t * DBK + kh * 32 < p.K;  // wave-uniform
        float4 fa[4], fw[4];
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
            fa[gg] = cld4(cs + foa + gg * 8);
            const float* w = cs + fow + gg * 8 * DLW;
            fw[gg] = make_float4(w[0], w[DLW], w[2 * DLW], w[3 * DLW]);
        }
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
            if (active) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fw[gg].x, fa[gg].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fw[gg].y, fa[gg].y, acc, 0, 0, 0);
            }
            if (gg == 0 && t + 1 < nk) {
                lstore(ns);
                if (t + 2 < nk) gload((t + 2) * DBK);
            }
            if (active) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fw[gg].z, fa[gg].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fw[gg].w, fa[gg].w, acc, 0, 0, 0);
            }
        }
        __syncthreads();
    }
    if (kh == 0) chain_bwd_epilogue<TAIL, 0>(p, acc, lds, m0, n0, wm, lane);
    else chain_bwd_epilogue<TAIL, 1>(p, acc, lds, m0, n0, wm, lane);
}

template <int TAIL>
static int chain_bwd_launch(ChainBwdParams& p, hipStream_t s) {
    p.ldw = p.Hp;
    p.tiles_x = p.Hp / CBH;
    p.tiles_y = cdiv(p.M, CBM);
    hipLaunchKernelGGL((chain_dgrad_kernel<TAIL>), dim3((unsigned)(p.tiles_x * p.tiles_y)), dim3(256), 0, s, p);
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}

static int chain_bwd_check(const float* dy, int ld_dy, int K, const float* w, const float* dx, int ld_dx, int rows, int Hp) {
    DVAE_REQUIRE(rows >= 0 && Hp > 0 && Hp % 32 == 0, "level dgrad: Hp must be a positive multiple of 32 (Hp=%d)", Hp);
    DVAE_REQUIRE(K >= 32 && K % 32 == 0, "level dgrad: K must be a positive multiple of 32 (K=%d)", K);
    DVAE_REQUIRE(dy && w && dx, "level dgrad: null operand");
    DVAE_REQUIRE(ld_dy >= K && ld_dy % 4 == 0 && ld_dx >= Hp && ld_dx % 4 == 0, "level dgrad: bad leading dimension");
    DVAE_REQUIRE(((uintptr_t)dy % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)dx % 16) == 0,
                 "level dgrad: operands must be 16-byte aligned");
    return DVAE_OK;
}

}  // namespace dvae

using namespace dvae;

extern "C" int dvae_gru_level_fwd(const float* hprev, int32_t ld_hprev, const float* whh, const float* gi_tab,
                                  const int32_t* types, const float* bhn, float* gates, float* ghn_out, float* hout,
                                  int32_t ld_hout, int32_t rows, int32_t Hp, void* stream) {
    if (rows == 0) return DVAE_OK;
    DVAE_TRY(chain_check(hprev, ld_hprev, whh, rows, Hp));
    DVAE_REQUIRE(gi_tab && types && bhn && gates && ghn_out && hout, "gru_level_fwd: null pointer");
    DVAE_REQUIRE(ld_hout >= Hp && ld_hout % 4 == 0, "gru_level_fwd: bad ld_hout %d", ld_hout);
    DVAE_REQUIRE(((uintptr_t)gi_tab % 16) == 0 && ((uintptr_t)bhn % 16) == 0 && ((uintptr_t)gates % 16) == 0 &&
                     ((uintptr_t)ghn_out % 16) == 0 && ((uintptr_t)hout % 16) == 0,
                 "gru_level_fwd: buffers must be 16-byte aligned");
    ChainParams p = {};
    p.A = hprev; p.lda = ld_hprev; p.W = whh; p.M = rows; p.Hp = Hp;
    p.gi_tab = gi_tab; p.types = types; p.bhn = bhn; p.hprev = hprev; p.ld_hprev = ld_hprev;
    p.gates = gates; p.ghn = ghn_out; p.hout = hout; p.ld_hout = ld_hout;
    return chain_launch<3>(p, (hipStream_t)stream);
}

extern "C" int dvae_gated_level_fwd(const float* h, int32_t ld_h, const float* wgm, const float* gv_row, float* gm,
                                    float* msg, int32_t ld_msg, int32_t cur_row0, float* next_agg, int32_t ld_agg,
                                    const int32_t* next_off, const int32_t* idx, int32_t rows, int32_t Hp,
                                    void* stream) {
    if (rows == 0) return DVAE_OK;
    DVAE_TRY(chain_check(h, ld_h, wgm, rows, Hp));
    DVAE_REQUIRE(gv_row && gm && msg, "gated_level_fwd: null pointer");
    DVAE_REQUIRE(!next_agg || (next_off && idx), "gated_level_fwd: next_agg needs the CSR lists");
    DVAE_REQUIRE(ld_msg >= Hp && ld_msg % 4 == 0 && (!next_agg || (ld_agg >= Hp && ld_agg % 4 == 0)),
                 "gated_level_fwd: bad leading dimension");
    DVAE_REQUIRE(((uintptr_t)gv_row % 16) == 0 && ((uintptr_t)gm % 16) == 0 && ((uintptr_t)msg % 16) == 0 &&
                     ((uintptr_t)next_agg % 16) == 0,
                 "gated_level_fwd: buffers must be 16-byte aligned");
    ChainParams p = {};
    p.A = h; p.lda = ld_h; p.W = wgm; p.M = rows; p.Hp = Hp;
    p.gv = gv_row; p.gm = gm; p.msg = msg; p.ld_msg = ld_msg; p.cur_row0 = cur_row0;
    p.next_agg = next_agg; p.ld_agg = ld_agg; p.next_off = next_off; p.idx = idx;
    return chain_launch<2>(p, (hipStream_t)stream);
}

extern "C" int dvae_level_dgrad_acc(const float* dy, int32_t ld_dy, int32_t K, const float* w, float* dx, int32_t ld_dx,
                                    int32_t rows, int32_t Hp, void* stream) {
    if (rows == 0) return DVAE_OK;
    DVAE_TRY(chain_bwd_check(dy, ld_dy, K, w, dx, ld_dx, rows, Hp));
    ChainBwdParams p = {};
    p.A = dy; p.lda = ld_dy; p.W = w; p.C = dx; p.ldc = ld_dx; p.M = rows; p.Hp = Hp; p.K = K;
    return chain_bwd_launch<0>(p, (hipStream_t)stream);
}

extern "C" int dvae_gru_level_bwd(const float* dy, int32_t ld_dy, int32_t K, const float* w, float* dh, int32_t ld_dh,
                                  const float* gates, const float* ghn, const float* hprev, int32_t ld_hprev,
                                  float* dgh, float* dgin, float* dhprev, int32_t ld_dhprev, int32_t rows, int32_t Hp,
                                  void* stream) {
    if (rows == 0) return DVAE_OK;
    DVAE_TRY(chain_bwd_check(dy, ld_dy, K, w, dh, ld_dh, rows, Hp));
    DVAE_REQUIRE(gates && ghn && hprev && dgh && dgin && dhprev, "gru_level_bwd: null pointer");
    DVAE_REQUIRE(ld_hprev >= Hp && ld_hprev % 4 == 0 && ld_dhprev >= Hp && ld_dhprev % 4 == 0,
                 "gru_level_bwd: bad leading dimension");
    DVAE_REQUIRE(((uintptr_t)gates % 16) == 0 && ((uintptr_t)ghn % 16) == 0 && ((uintptr_t)hprev % 16) == 0 &&
                     ((uintptr_t)dgh % 16) == 0 && ((uintptr_t)dgin % 16) == 0 && ((uintptr_t)dhprev % 16) == 0,
                 "gru_level_bwd: buffers must be 16-byte aligned");
    ChainBwdParams p = {};
    p.A = dy; p.lda = ld_dy; p.W = w; p.C = dh; p.ldc = ld_dh; p.M = rows; p.Hp = Hp; p.K = K;
    p.gates = gates; p.ghn = ghn; p.hprev = hprev; p.ld_hprev = ld_hprev;
    p.dgh = dgh; p.dgin = dgin; p.dhprev = dhprev; p.ld_dhprev = ld_dhprev;
    return chain_bwd_launch<1>(p, (hipStream_t)stream);
}
