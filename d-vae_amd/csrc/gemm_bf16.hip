// bf16-input MFMA GEMM for gfx950 (v_mfma_f32_32x32x16_bf16, fp32 accumulate).
//
// Same contract as gemm.hip: fp32 operands and fp32 result in HBM, the three products of a
// Linear layer (forward / dgrad / wgrad), row gather and scatter, split-K.  The operands are
// rounded to bf16 (round-to-nearest-even) on their way from registers to LDS; everything
// after the MFMA (bias, activation, accumulate, atomics) is fp32.  Master weights, activations
// and gradients stay fp32 in HBM -- this is the "bf16 MFMA inputs, fp32 accumulate, fp32 master
// weights" mode of SURVEY.md section 8d, reported separately from the fp32 parity mode.
//
// Tile: BM x BN x 64 per 256-thread workgroup.  LDS images (bf16):
//   K-contiguous operand: [rows][64 + 8]: a lane's MFMA fragment (8 consecutive k of one row)
//                         is one ds_read_b128; the 144-byte row stride spreads a 16-lane group
//                         over all 64 banks.
//   K-major operand:      [64][rows + pad], rows contiguous exactly as in HBM, so the global
//                         loads stay coalesced and the LDS writes contiguous; the fragment is
//                         fetched with the transposing read ds_read_b64_tr_b16 (two per
//                         fragment: 4 k each).  pad makes the row stride = 64 (mod 256) bytes:
//                         the four k rows of a transposed read fall into four disjoint 16-bank
//                         windows.
// Pipeline as in gemm.hip: register staged, the next tile is written to the other LDS buffer
// in the middle of a step and the loads of the tile after it are issued right behind.
#include "gemm_common.h"

namespace dvae {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int BKH = 64;          // K step
constexpr int KC_LDH = BKH + 8;  // K-contiguous image row stride, bf16 elements (144 B)
template <int ROWS>
struct KmLd {
    static constexpr int value = ROWS == 32 ? 32 : ROWS + 32;  // K-major image row stride (64 / 192 / 320 B)
};

__device__ __forceinline__ uint2 pack_bf16x4(float4 v) {
    bf16x4 b = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    return __builtin_bit_cast(uint2, b);
}

// HALF (K-major operands only): the operand is STORED as bf16 (a gradient table or an activation shadow written in
// bf16 by its producer): a unit is 8 bytes of HBM instead of 16 and goes to LDS as loaded.
template <int ROWS, bool KMAJOR, bool KGATHER, int NT = 256, bool HALF = false>
struct TileLoaderH {
    static_assert(!HALF || KMAJOR, "bf16-stored K-contiguous operands: gemm_bf16_nn.hip");
    static constexpr int UNITS = ROWS * 16 / NT;  // float4 units per thread: 2, 4 or 8
    static constexpr int KM_LD = KmLd<ROWS>::value;
    static constexpr int LDS_ELEMS = KMAJOR ? BKH * KM_LD : ROWS * KC_LDH;  // bf16 elements
    static constexpr int KROWS = NT / (ROWS / 4);                           // K-major: k rows covered by one unit round
    static_assert(UNITS == 2 || UNITS == 4 || UNITS == 8, "tile rows must be 32, 64 or 128 (256 threads), 256 (512 threads)");

    float4 r[HALF ? 1 : UNITS];
    uint2 rh[HALF ? UNITS : 1];
    const float* rowp[UNITS];  // K-contig: start of the (gathered, clamped) row
    int nidx[UNITS];           // K-major gather: row indices of the NEXT tile to load
    const int* rows_;
    int ld_;
    unsigned col_;             // K-major: column of this thread's float4 (elements) or OOB
    __amdgpu_buffer_rsrc_t rsrc_;

    __device__ __forceinline__ void init(const float* __restrict__ base, const int* __restrict__ rows, int ld, int r0,
                                         int rdim, int tid) {
        rows_ = rows;
        ld_ = ld;
        if (!KMAJOR) {
#pragma unroll
            for (int i = 0; i < UNITS; ++i) {
                int row = (tid + NT * i) >> 4;
                int rr = r0 + row;
                rr = rr < rdim ? rr : rdim - 1;  // clamped rows are computed but never stored
                long ridx = rows ? (long)rows[rr] : (long)rr;
                rowp[i] = base + ridx * ld;
            }
        } else {
            rsrc_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, OOB, 0x00020000);
            int m4 = tid % (ROWS / 4);
            int c = r0 + m4 * 4;
            col_ = c < rdim ? (unsigned)c : OOB;
        }
    }
    // K-major + gather: fetch the row indices of tile `knext` (one tile ahead of the data)
    __device__ __forceinline__ void next_idx(int knext, int kend, int tid) {
        if (KMAJOR && KGATHER) {
#pragma unroll
            for (int i = 0; i < UNITS; ++i) {
                int kk = knext + tid / (ROWS / 4) + KROWS * i;
                nidx[i] = rows_[kk < kend ? kk : kend - 1];
            }
        }
    }
    __device__ __forceinline__ void load(int k0, int kend, int tid) {
        if (!KMAJOR) {
            const int c4 = tid & 15;
            int kk = k0 + c4 * 4;
            kk = kk < kend ? kk : kend - 4;  // past the end: a valid address; the value is zeroed at store()
#pragma unroll
            for (int i = 0; i < UNITS; ++i) r[i] = *reinterpret_cast<const float4*>(rowp[i] + kk);
        } else {
#pragma unroll
            for (int i = 0; i < UNITS; ++i) {
                int kk = k0 + tid / (ROWS / 4) + KROWS * i;
                bool ok = kk < kend && col_ != OOB;
                unsigned ridx = KGATHER ? (unsigned)nidx[i] : (unsigned)kk;
                if constexpr (HALF) {
                    unsigned off = ok ? (ridx * (unsigned)ld_ + col_) * 2u : OOB;
                    u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc_, off, 0, 0);
                    rh[i] = make_uint2(v.x, v.y);
                } else {
                    unsigned off = ok ? (ridx * (unsigned)ld_ + col_) * 4u : OOB;
                    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_, off, 0, 0);
                    r[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
                }
            }
        }
    }
    // registers (the tile that was loaded for k0) -> bf16 LDS image
    __device__ __forceinline__ void store(__bf16* lds, int k0, int kend, int tid) {
        if (!KMAJOR) {
            const int c4 = tid & 15;
            const bool ok = k0 + c4 * 4 < kend;
#pragma unroll
            for (int i = 0; i < UNITS; ++i) {
                int row = (tid + NT * i) >> 4;
                float4 v = ok ? r[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<uint2*>(lds + row * KC_LDH + c4 * 4) = pack_bf16x4(v);
            }
        } else {
            const int m4 = tid % (ROWS / 4);
#pragma unroll
            for (int i = 0; i < UNITS; ++i) {
                int k = tid / (ROWS / 4) + KROWS * i;
                if constexpr (HALF) *reinterpret_cast<uint2*>(lds + k * KM_LD + m4 * 4) = rh[i];
                else *reinterpret_cast<uint2*>(lds + k * KM_LD + m4 * 4) = pack_bf16x4(r[i]);
            }
        }
    }
};

// MFMA fragment of k-step s (16 k) for the 32-row sub-tile at `roff`: lane l holds row
// roff + (l & 31), k = 16 s + 8 (l >> 5) + 0..7.
template <int ROWS, bool KMAJOR>
__device__ __forceinline__ bf16x8 frag_h(const __bf16* lds, int roff, int s, int lane) {
    const int h = lane >> 5;
    if (!KMAJOR) {
        return *reinterpret_cast<const bf16x8*>(lds + (roff + (lane & 31)) * KC_LDH + (2 * s + h) * 8);
    } else {
        constexpr int LD = KmLd<ROWS>::value;
        // transposed read: lane 4q+p of a 16-lane group addresses k row q, columns 4p..4p+3 of the
        // group's 4 x 16 block and receives column (lane & 15), k rows 0..3
        const int li = lane & 15, q = li >> 2, pcol = li & 3;
        const __bf16* p = lds + (16 * s + 8 * h + q) * LD + roff + 16 * ((lane >> 4) & 1) + 4 * pcol;
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * LD));
        s16x8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        return __builtin_bit_cast(bf16x8, v);
    }
}

template <int BM, int BN, int WM, int WN, bool AKM, bool BKM, bool KGATHER, bool SWAP, bool AH = false, bool BH = false>
__global__ __launch_bounds__(64 * WM * WN) void gemm_bf16_kernel(GemmParams p) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32, NT = 64 * WM * WN;
    static_assert(NT == 256 || NT == 512, "4 or 8 waves");
    using LA = TileLoaderH<BM, AKM, false, NT, AH>;
    using LB = TileLoaderH<BN, BKM, KGATHER, NT, BH>;
    constexpr int STAGE = LA::LDS_ELEMS + LB::LDS_ELEMS;
    __shared__ __attribute__((aligned(16))) __bf16 lds[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    constexpr bool XMAP = BM == 256;  // the 256 x 256 weight-gradient tile: one (K range, tile) item per workgroup, see launch
    const int ntiles = p.tiles_x * p.tiles_y;  // persistent over output tiles, as in gemm.hip
    int kbeg = 0, kend = p.K, first = blockIdx.x, stride = gridDim.x;
    if (XMAP) {
        // Items are numbered K range by K range and dealt to the XCDs in contiguous runs (workgroup ids equal
        // mod 8 share an XCD and its L2), so the tiles that contract one K range -- and re-read the same rows of
        // both operands, N/256 and M/256 times -- meet in one L2 and HBM delivers each operand about once.
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
        const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
        const int ks = id / ntiles;
        first = id - ks * ntiles;
        stride = ntiles;  // one tile per workgroup
        kbeg = ks * p.k_per_split;
        kend = min(p.K, kbeg + p.k_per_split);
    } else if (p.split_k > 1) {
        kbeg = blockIdx.z * p.k_per_split;
        kend = min(p.K, kbeg + p.k_per_split);
        if (kbeg >= kend) return;
    }
    for (int tl = first; tl < ntiles; tl += stride) {
    int m0, n0;
    if (XMAP) {
        m0 = (tl % p.tiles_y) * BM;
        n0 = (tl / p.tiles_y) * BN;
    } else {
        tile_origin<BM, BN>(tl, p.tiles_x, ntiles, m0, n0);
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (kend > kbeg) {
        LA la; LB lb;
        la.init(p.A, p.a_rows, p.lda, m0, p.M, tid);
        lb.init(p.B, p.b_rows, p.ldb, n0, p.N, tid);
        // tile(t) = kbeg + t*BKH, clamped to a valid tile past the end (loaded, never used)
        auto tile = [&](int t) { int k = kbeg + t * BKH; return k < kend ? k : kbeg; };
        lb.next_idx(tile(0), kend, tid);
        la.load(tile(0), kend, tid);
        lb.load(tile(0), kend, tid);
        lb.next_idx(tile(1), kend, tid);
        la.store(lds, tile(0), kend, tid);
        lb.store(lds + LA::LDS_ELEMS, tile(0), kend, tid);
        la.load(tile(1), kend, tid);
        lb.load(tile(1), kend, tid);
        lb.next_idx(tile(2), kend, tid);
        __syncthreads();

        int cur = 0, t = 0;
        for (int k0 = kbeg; k0 < kend; k0 += BKH, ++t) {
            const __bf16* a_l = lds + cur * STAGE;
            const __bf16* b_l = a_l + LA::LDS_ELEMS;
#pragma unroll
            for (int s = 0; s < BKH / 16; ++s) {
                bf16x8 af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = frag_h<BM, AKM>(a_l, (wm * TM + i) * 32, s, lane);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = frag_h<BN, BKM>(b_l, (wn * TN + j) * 32, s, lane);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j], af[i], acc[i][j], 0, 0, 0)
                                         : __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
                if (s == 1) {
                    // mid-step: staged registers (tile t+1) -> LDS[cur^1], then issue the loads of tile t+2
                    la.store(lds + (cur ^ 1) * STAGE, tile(t + 1), kend, tid);
                    lb.store(lds + (cur ^ 1) * STAGE + LA::LDS_ELEMS, tile(t + 1), kend, tid);
                    __builtin_amdgcn_sched_barrier(0);
                    la.load(tile(t + 2), kend, tid);
                    lb.load(tile(t + 2), kend, tid);
                    lb.next_idx(tile(t + 3), kend, tid);
                }
            }
            __syncthreads();
            cur ^= 1;
        }
    }
    store_acc<TM, TN, SWAP>(p, acc, m0, n0, wm, wn, lane, BM, BN);  // SWAP: transposed accumulation, 16-byte epilogue (gemm_common.h)
    }  // tiles
}

template <int BM, int BN, int WM, int WN>
static int launch_cfg_h(GemmParams p, int akm, int bkm, hipStream_t s, int ah = 0, int bh = 0) {
    p.tiles_x = cdiv(p.N, BN);
    p.tiles_y = cdiv(p.M, BM);
    const long ntiles = (long)p.tiles_x * p.tiles_y;
    const long cap = (BM * BN >= 128 * 128 ? 512 : 1024);
    dim3 grid((unsigned)(ntiles < cap ? ntiles : cap), 1, p.split_k > 1 ? p.split_k : 1);
    if (BM == 256) {  // one workgroup per (K range, tile)
        if (p.split_k < 1 || p.k_per_split <= 0) { p.split_k = 1; p.k_per_split = cdiv(p.K, BKH) * BKH; }
        grid = dim3((unsigned)(ntiles * p.split_k), 1, 1);
    }
    dim3 block(64 * WM * WN);
    const bool kg = bkm && p.b_rows != nullptr;
    if (akm && p.a_rows) return set_err(DVAE_ERR_ARG, "gemm: row gather on a K-major A operand is not built");
#define DVAE_GEMM_LAUNCH(A_, B_, G_)                                                                              \
    do {                                                                                                          \
        if (vec) hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WM, WN, A_, B_, G_, true>), grid, block, 0, s, p);  \
        else hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WM, WN, A_, B_, G_, false>), grid, block, 0, s, p);     \
    } while (0)
    // bf16-STORED K-major operands (weight-gradient form: the edge predictor's gradient tables, d gh / d gm and the
    // state / predecessor-sum shadows): A alone or both
#define DVAE_GEMM_LAUNCH_H(G_, BH_)                                                                                       \
    do {                                                                                                                  \
        if (vec) hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WM, WN, true, true, G_, true, true, BH_>), grid, block, 0, s, p);  \
        else hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WM, WN, true, true, G_, false, true, BH_>), grid, block, 0, s, p);     \
    } while (0)
    const bool vec = gemm_vec_epilogue_ok(p);
    if (bh && !ah) return set_err(DVAE_ERR_ARG, "gemm: a bf16-stored K-major B operand is built beside a bf16-stored A only");
    if (ah) {
        if (!(akm && bkm)) return set_err(DVAE_ERR_ARG, "gemm: bf16-stored operands here are K-major (the weight-gradient form)");
        if (kg && bh) DVAE_GEMM_LAUNCH_H(true, true);
        else if (kg) DVAE_GEMM_LAUNCH_H(true, false);
        else if (bh) DVAE_GEMM_LAUNCH_H(false, true);
        else DVAE_GEMM_LAUNCH_H(false, false);
    } else
    if constexpr (BM == 256) {  // built for the weight-gradient form only
        if (!(akm && bkm)) return set_err(DVAE_ERR_ARG, "gemm: the 256 x 256 tile is built for two K-major operands");
        if (kg) DVAE_GEMM_LAUNCH(true, true, true);
        else DVAE_GEMM_LAUNCH(true, true, false);
    } else
    if (!akm && !bkm) DVAE_GEMM_LAUNCH(false, false, false);
    else if (!akm && bkm && !kg) DVAE_GEMM_LAUNCH(false, true, false);
    else if (akm && bkm && !kg) DVAE_GEMM_LAUNCH(true, true, false);
    else if (akm && bkm && kg) DVAE_GEMM_LAUNCH(true, true, true);
    else return set_err(DVAE_ERR_ARG, "gemm: this operand layout / gather combination is not built");
#undef DVAE_GEMM_LAUNCH_H
#undef DVAE_GEMM_LAUNCH
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}

int gemm_bf16(const dvae_gemm_desc& d, hipStream_t s) {
    DVAE_TRY(gemm_check(d, 4));
    if (d.M == 0 || d.N == 0) return DVAE_OK;
    {   // two K-contiguous operands, bf16- or fp32-stored: the large-tile kernels of gemm_bf16_nn.hip
        const int rc = gemm_bf16_nn(d, s);
        if (rc <= 0) return rc;
        // (bf16-stored K-major operands: the large weight-gradient form below, A or A and B)
        if (d.c_bf16 || ((d.a_bf16 || d.b_bf16) && !(d.a_kmajor && d.b_kmajor && d.a_bf16)))
            return set_err(DVAE_ERR_ARG, "gemm_bf16: bf16-stored operands / outputs need two K-contiguous operands");
    }
    dvae_gemm_desc sub;
    int k0 = 0;
    // a K-major operand is addressed with 32-bit byte offsets: contractions over more than 4 GiB
    // of it run as accumulating K chunks
    const int kc = gemm_k_chunk(d, BKH);
    if (kc < 0) return kc;
    if (kc < d.K) {
        while (gemm_next_chunk(d, kc, k0, sub)) DVAE_TRY(gemm_bf16(sub, s));
        return DVAE_OK;
    }
    GemmParams p;
    p.A = d.A; p.B = d.B; p.C = d.C;
    p.a_rows = d.a_rows; p.b_rows = d.b_rows; p.c_rows = d.c_rows;
    p.bias = d.bias; p.relu_mask = d.relu_mask; p.lda = d.lda; p.ldb = d.ldb; p.ldc = d.ldc;
    p.M = d.M; p.N = d.N; p.K = d.K; p.act = d.act; p.accumulate = d.accumulate;
    p.split_k = 1; p.k_per_split = 0; p.c_bf16 = 0;
    // Tile / split-K choice: as for the fp32 kernel.  A bf16 K step is 16x cheaper on the matrix
    // pipe, so these products are bound by operand traffic and launch latency: prefer many
    // workgroups (two or more per CU) over long K loops.
    static const int knob_tile = getenv("DVAE_GEMM_TILE") ? atoi(getenv("DVAE_GEMM_TILE")) : 0;    // tools/gemm_tune.py
    static const int knob_split = getenv("DVAE_GEMM_SPLIT") ? atoi(getenv("DVAE_GEMM_SPLIT")) : 0;
    static const int knob_log = getenv("DVAE_GEMM_LOG") ? atoi(getenv("DVAE_GEMM_LOG")) : 0;
    long t128 = (long)cdiv(d.M, 128) * cdiv(d.N, 128);
    long t64 = (long)cdiv(d.M, 64) * cdiv(d.N, 64);
    bool use128 = t128 >= 384 || (d.a_kmajor && t128 >= 64);
    int ksteps = cdiv(d.K, BKH);
    bool small_m = d.M <= 32 && d.a_kmajor;
    // Large weight gradients (cfg 5: 4096 x 1024 and 3072 x 1024 outputs contracted over 1e5 .. 6e5 rows): 256 x 256
    // tiles halve the operand bytes each MFMA flop pulls through L2 -> LDS, which is what bounds these products
    // (fp32-stored operands, 8 bytes per element pair); the K ranges fill the chip, one workgroup per CU.
    const bool can_split = d.accumulate && d.bias == nullptr && d.act == 0;
    const long t256 = (long)cdiv(d.M, 256) * cdiv(d.N, 256);
    bool use256 = d.a_kmajor && d.b_kmajor && can_split && t256 >= 16 && ksteps >= 256 && d.a_rows == nullptr;
    static const int knob_256 = getenv("DVAE_WGRAD_256") ? atoi(getenv("DVAE_WGRAD_256")) : 1;  // 0: the 128 x 128 tile as before
    if (!knob_256) use256 = false;
    if (knob_tile == 256) use256 = d.a_kmajor && d.b_kmajor && d.a_rows == nullptr;
    else if (knob_tile) use256 = false;
    if (d.a_bf16) {  // bf16-stored K-major operands (every tile's loaders take them)
        if (!can_split || d.a_rows) return set_err(DVAE_ERR_ARG, "gemm_bf16: bf16-stored K-major operands serve accumulating weight-gradient products only");
        if ((d.lda % 4 || (uintptr_t)d.A % 8) || (d.b_bf16 && (d.ldb % 4 || (uintptr_t)d.B % 8)))
            return set_err(DVAE_ERR_ARG, "gemm_bf16: bf16-stored K-major operands need 8-byte aligned rows");
    }
    if (use256) {
        static const int cus = [] {
            int dev = 0, n = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
            return n;
        }();
        int sk = can_split ? (int)(cus / t256) : 1;
        if (sk > ksteps / 8) sk = ksteps / 8;
        if (knob_split > 0 && can_split) sk = knob_split < ksteps ? knob_split : ksteps;
        if (sk < 1) sk = 1;
        p.k_per_split = cdiv(ksteps, sk) * BKH;
        p.split_k = cdiv(d.K, p.k_per_split);
        if (knob_log)
            fprintf(stderr, "GEMMSELH 256x256 %d %d %d akm=1 bkm=1 split=%d br=%d\n", d.M, d.N, d.K, p.split_k, d.b_rows != nullptr);
        return launch_cfg_h<256, 256, 2, 4>(p, d.a_kmajor, d.b_kmajor, s, d.a_bf16, d.b_bf16);
    }
    if (knob_tile == 128) { use128 = true; small_m = false; }
    else if (knob_tile == 64) { use128 = false; small_m = false; }
    else if (knob_tile == 32) small_m = true;
    if (d.accumulate && d.bias == nullptr && d.act == 0 && ksteps >= 4) {
        long tiles = small_m ? (long)cdiv(d.N, 128) : (use128 ? t128 : t64);
        long target = use128 && !small_m ? 512 : 256;  // small tiles: measured best at one workgroup per CU
        int want = (int)((target + tiles - 1) / tiles);
        int maxsplit = ksteps / 2;
        int sk = want < maxsplit ? want : maxsplit;
        if (!use128 && !small_m && d.a_kmajor && d.b_kmajor) {
            // weight gradients on 64x64 tiles (tools/gemm_tune.py, DVAE_GEMM_BF16=1): whole rounds of
            // 256 workgroups, at least two per CU
            double best_cost = -1.0;
            for (int k = 1; k <= 8 && k <= maxsplit; ++k) {
                const long wgs = tiles * k;
                const double cost = (double)((wgs + 255) / 256) * (cdiv(ksteps, k) + 6) * (wgs < 512 ? 1.25 : 1.0);
                if (best_cost < 0.0 || cost < best_cost) { best_cost = cost; sk = k; }
            }
        }
        if (knob_split > 0) sk = knob_split < ksteps ? knob_split : ksteps;
        if (sk > 1) {
            p.k_per_split = cdiv(ksteps, sk) * BKH;
            p.split_k = cdiv(d.K, p.k_per_split);
        }
    }
    if (knob_log)
        fprintf(stderr, "GEMMSELH %s %d %d %d akm=%d bkm=%d split=%d br=%d\n", small_m ? "32x128" : use128 ? "128x128" : "64x64", d.M,
                d.N, d.K, d.a_kmajor, d.b_kmajor, p.split_k, d.b_rows != nullptr);
    if (small_m) return launch_cfg_h<32, 128, 1, 4>(p, d.a_kmajor, d.b_kmajor, s, d.a_bf16, d.b_bf16);
    if (use128) return launch_cfg_h<128, 128, 2, 2>(p, d.a_kmajor, d.b_kmajor, s, d.a_bf16, d.b_bf16);
    return launch_cfg_h<64, 64, 2, 2>(p, d.a_kmajor, d.b_kmajor, s, d.a_bf16, d.b_bf16);
}

}  // namespace dvae

extern "C" int dvae_gemm_bf16(const dvae_gemm_desc* d, void* stream) {
    if (!d) return dvae::set_err(DVAE_ERR_ARG, "dvae_gemm_bf16: null descriptor");
    return dvae::gemm_bf16(*d, (hipStream_t)stream);
}
