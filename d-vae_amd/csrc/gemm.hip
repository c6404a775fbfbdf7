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
#include "gemm_common.h"

namespace dvae {

#ifndef DVAE_GEMM_BK
#define DVAE_GEMM_BK 32
#endif
constexpr int BK = DVAE_GEMM_BK;
constexpr int C4R = BK / 4;  // float4 chunks per K-contiguous tile row
constexpr int KC_LD = BK + 4;  // K-contiguous LDS row stride (floats)
#ifndef DVAE_GEMM_MIDG
#define DVAE_GEMM_MIDG (DVAE_GEMM_BK / 16 - 1)
#endif
constexpr int MIDG = DVAE_GEMM_MIDG;  // k-group of a step after which the next tile is staged

// Register-staged tile loader.  Everything that does not depend on the K step
// (row gather, row clamping, column guards) is resolved ONCE in init(); the K loop
// then issues UNITS independent 16-byte loads back to back with no branch between
// them.  The staging registers and row pointers are NAMED scalars (r0..r3, p0..p3),
// not arrays: as arrays hipcc left them in scratch memory across the loop back-edge
// (scratch_store / scratch_load of every staged tile, ScratchSize 144 B/lane).
template <int ROWS, bool KMAJOR, bool KGATHER>
struct TileLoader {
    static constexpr int UNITS = ROWS * BK / 4 / 256;  // float4 units per thread: 1, 2 or 4
    static constexpr int LDS_ELEMS = KMAJOR ? BK * ROWS : ROWS * KC_LD;
    static_assert(UNITS == 1 || UNITS == 2 || UNITS == 4, "tile rows must be 32, 64 or 128");
    float4 r0, r1, r2, r3;
    const float *p0, *p1, *p2, *p3;  // K-contig: row base (+ c4*4); K-major: base + column
    bool c0, c1, c2, c3;             // K-major: column inside the matrix
    const int* rows_;
    int ld_;

    // K-contig: unit u -> row u/8, c4 u%8.   K-major: unit u -> k u/(ROWS/4), m4 u%(ROWS/4)
    __device__ __forceinline__ void init1(const float*& ptr, bool& colok, unsigned& col, int i,
                                          const float* __restrict__ base, const int* __restrict__ rows, int ld,
                                          int r0_, int rdim, int tid) {
        int u = tid + 256 * i;
        if (!KMAJOR) {
            int row = u / C4R, c4 = u % C4R;
            int r = r0_ + row;
            r = r < rdim ? r : rdim - 1;  // clamped rows are computed but never stored
            long ridx = rows ? (long)rows[r] : (long)r;
            ptr = base + ridx * ld + c4 * 4;
            colok = true;
            col = 0;
        } else {
            int m4 = u % (ROWS / 4);
            int r = r0_ + m4 * 4;
            colok = r < rdim;
            col = (unsigned)r;
            ptr = base;
        }
    }
    __device__ __forceinline__ void init(const float* __restrict__ base, const int* __restrict__ rows, int ld, int r0_,
                                         int rdim, int tid) {
        rows_ = rows;
        ld_ = ld;
        if (KMAJOR) rsrc_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, OOB, 0x00020000);
        init1(p0, c0, col0_, 0, base, rows, ld, r0_, rdim, tid);
        if (UNITS > 1) init1(p1, c1, col1_, 1, base, rows, ld, r0_, rdim, tid);
        if (UNITS > 2) {
            init1(p2, c2, col2_, 2, base, rows, ld, r0_, rdim, tid);
            init1(p3, c3, col3_, 3, base, rows, ld, r0_, rdim, tid);
        }
    }
    // K-major operands are read with buffer loads through a raw descriptor whose range
    // check supplies the zeros: an element outside the matrix (k >= kend in the last tile
    // of a split, column >= rdim at the tile edge) gets the byte offset OOB and the hardware
    // returns 0 -- no branch, and no instruction depends on the loaded data before the LDS
    // store (a select on the data is hoisted to the top of the next step by hipcc and exposes
    // the load latency; a select between pointers of two address spaces becomes branches).
    // Consequence: a K-major operand must lie within 4 GiB of its base pointer.
    // With KGATHER the row indices are themselves loaded one tile ahead of the data
    // (next_idx), so the data loads never wait on an index load.
    __amdgpu_buffer_rsrc_t rsrc_;
    unsigned col0_, col1_, col2_, col3_;  // column (elements) of each unit, or OOB
    int n0_, n1_, n2_, n3_;               // gathered row indices of the NEXT tile to load
    __device__ __forceinline__ int kclamp(int i, int k0, int kend, int tid) {
        int kk = k0 + (tid + 256 * i) / (ROWS / 4);
        return kk < kend ? kk : kend - 1;
    }
    __device__ __forceinline__ float4 kload(int i, int k0, int kend, int tid, bool colok, unsigned col, int nidx) {
        int kk = k0 + (tid + 256 * i) / (ROWS / 4);
        bool ok = kk < kend && colok;
        unsigned ridx = KGATHER ? (unsigned)nidx : (unsigned)kk;
        unsigned off = ok ? (ridx * (unsigned)ld_ + col) * 4u : OOB;
        u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_, off, 0, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
    // call once before the first load(k0) and after every load: fetches indices for tile knext
    __device__ __forceinline__ void next_idx(int knext, int kend, int tid) {
        if (KMAJOR && KGATHER) {
            n0_ = rows_[kclamp(0, knext, kend, tid)];
            if (UNITS > 1) n1_ = rows_[kclamp(1, knext, kend, tid)];
            if (UNITS > 2) {
                n2_ = rows_[kclamp(2, knext, kend, tid)];
                n3_ = rows_[kclamp(3, knext, kend, tid)];
            }
        }
    }
    // loads the tile at k0 (for KGATHER: k0 must be the tile next_idx() was last called for)
    __device__ __forceinline__ void load(int k0, int kend, int tid) {
        if (!KMAJOR) {
            r0 = *reinterpret_cast<const float4*>(p0 + k0);
            if (UNITS > 1) r1 = *reinterpret_cast<const float4*>(p1 + k0);
            if (UNITS > 2) {
                r2 = *reinterpret_cast<const float4*>(p2 + k0);
                r3 = *reinterpret_cast<const float4*>(p3 + k0);
            }
        } else {
            r0 = kload(0, k0, kend, tid, c0, col0_, n0_);
            if (UNITS > 1) r1 = kload(1, k0, kend, tid, c1, col1_, n1_);
            if (UNITS > 2) {
                r2 = kload(2, k0, kend, tid, c2, col2_, n2_);
                r3 = kload(3, k0, kend, tid, c3, col3_, n3_);
            }
        }
    }
    __device__ __forceinline__ static void store1(float* lds, int i, int tid, float4 v) {
        int u = tid + 256 * i;
        if (!KMAJOR) {
            int row = u / C4R, c4 = u % C4R;
            *reinterpret_cast<float4*>(lds + row * KC_LD + c4 * 4) = v;
        } else {
            int k = u / (ROWS / 4), m4 = u % (ROWS / 4);
            *reinterpret_cast<float4*>(lds + k * ROWS + m4 * 4) = v;
        }
    }
    __device__ __forceinline__ void store(float* lds, int tid) {
        store1(lds, 0, tid, r0);
        if (UNITS > 1) store1(lds, 1, tid, r1);
        if (UNITS > 2) {
            store1(lds, 2, tid, r2);
            store1(lds, 3, tid, r3);
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

template <int BM, int BN, int WM, int WN, bool AKM, bool BKM, bool KGATHER>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmParams p) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert(WM * WN == 4, "4 waves");
    using LA = TileLoader<BM, AKM, false>;  // only the B operand of a weight-gradient product is row-gathered along K
    using LB = TileLoader<BN, BKM, KGATHER>;
    __shared__ __attribute__((aligned(16))) float lds[2 * (LA::LDS_ELEMS + LB::LDS_ELEMS)];
    constexpr int STAGE = LA::LDS_ELEMS + LB::LDS_ELEMS;  // one stage: A image then B image

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    int kbeg = 0, kend = p.K;
    if (p.split_k > 1) {
        kbeg = blockIdx.z * p.k_per_split;
        kend = min(p.K, kbeg + p.k_per_split);
        if (kbeg >= kend) return;
    }
    // Persistent over output tiles: the launch has at most as many workgroups as fit on the chip
    // at once; each walks its tiles, so a tile's stores drain under the next tile's prologue
    // instead of holding the workgroup's slot until they are acknowledged.
    const int ntiles = p.tiles_x * p.tiles_y;
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
    int m0, n0;
    tile_origin<BM, BN>(tl, p.tiles_x, ntiles, m0, n0);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    LA la; LB lb;
    la.init(p.A, p.a_rows, p.lda, m0, p.M, tid);
    lb.init(p.B, p.b_rows, p.ldb, n0, p.N, tid);
    // tile(t) = kbeg + t*BK, clamped to a valid tile past the end (loaded, never used)
    auto tile = [&](int t) { int k = kbeg + t * BK; return k < kend ? k : kbeg; };
    lb.next_idx(tile(0), kend, tid);
    la.load(tile(0), kend, tid);
    lb.load(tile(0), kend, tid);
    lb.next_idx(tile(1), kend, tid);
    la.store(lds, tid);
    lb.store(lds + LA::LDS_ELEMS, tid);
    la.load(tile(1), kend, tid);
    lb.load(tile(1), kend, tid);
    lb.next_idx(tile(2), kend, tid);
    __syncthreads();

    // Steady state of step k: MFMAs on LDS[cur] (tile k) | registers hold tile k+1, in flight
    // since the END of step k-1 | after the MFMAs: registers -> LDS[cur^1], then the loads of
    // tile k+2 are issued.  Issuing at the block end keeps the loads a whole step ahead of
    // their consumer (hipcc sinks loads placed at the top of the step down to the LDS stores;
    // a scheduling fence there demotes the staging registers to scratch).
    int cur = 0, t = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BK, ++t) {
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
            if (g == MIDG) {
                // mid-step: registers (tile k+1, in flight since the middle of step k-1) -> LDS[cur^1],
                // then the loads of tile k+2 are issued: a whole step of MFMAs hides their latency.
                la.store(lds + (cur ^ 1) * STAGE, tid);
                lb.store(lds + (cur ^ 1) * STAGE + LA::LDS_ELEMS, tid);
                // fence: the loads below must not be hoisted above the LDS stores that read the same
                // staging registers (hipcc then renames them and waits vmcnt(0) on copies at the block end)
                __builtin_amdgcn_sched_barrier(0);
                la.load(tile(t + 2), kend, tid);
                lb.load(tile(t + 2), kend, tid);
                lb.next_idx(tile(t + 3), kend, tid);
            }
        }
        __syncthreads();
        cur ^= 1;
    }

    store_acc<TM, TN>(p, acc, m0, n0, wm, wn, lane);
    }  // tiles
}

template <int BM, int BN, int WM, int WN>
static int launch_cfg(GemmParams p, int akm, int bkm, hipStream_t s) {
    p.tiles_x = cdiv(p.N, BN);
    p.tiles_y = cdiv(p.M, BM);
    const long ntiles = (long)p.tiles_x * p.tiles_y;
    const long cap = (BM * BN >= 128 * 128 ? 512 : 1024);  // co-resident workgroups the chip holds (multiple of 8: XCD runs)
    dim3 grid((unsigned)(ntiles < cap ? ntiles : cap), 1, p.split_k > 1 ? p.split_k : 1);
    dim3 block(256);
    const bool kg = bkm && p.b_rows != nullptr;
    if (akm && p.a_rows) return set_err(DVAE_ERR_ARG, "gemm: row gather on a K-major A operand is not built");
    if (!akm && !bkm) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, false, false, false>), grid, block, 0, s, p);
    else if (!akm && bkm && !kg) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, false, true, false>), grid, block, 0, s, p);
    else if (akm && bkm && !kg) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, true, true, false>), grid, block, 0, s, p);
    else if (akm && bkm && kg) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, true, true, true>), grid, block, 0, s, p);
    else return set_err(DVAE_ERR_ARG, "gemm: this operand layout / gather combination is not built");
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}

// argument checks common to both kernels; kmult = granularity of K for K-contiguous operands
int gemm_check(const dvae_gemm_desc& d, int kmult) {
    DVAE_REQUIRE(d.M >= 0 && d.N >= 0 && d.K >= 0, "gemm: negative size");
    if (d.M == 0 || d.N == 0) return DVAE_OK;
    DVAE_REQUIRE(d.A && d.B && d.C, "gemm: null operand");
    DVAE_REQUIRE(((uintptr_t)d.A % 16) == 0 && ((uintptr_t)d.B % 16) == 0, "gemm: operands must be 16-byte aligned");
    DVAE_REQUIRE(d.lda % 4 == 0 && d.ldb % 4 == 0, "gemm: leading dimensions must be multiples of 4 floats");
    if (!d.a_kmajor) DVAE_REQUIRE(d.K % kmult == 0 && d.lda >= d.K, "gemm: K-contiguous A needs K %% %d == 0 (K=%d lda=%d)", kmult, d.K, d.lda);
    else DVAE_REQUIRE(d.M % 4 == 0 && d.lda >= d.M, "gemm: K-major A needs M %% 4 == 0 (M=%d lda=%d)", d.M, d.lda);
    if (!d.b_kmajor) DVAE_REQUIRE(d.K % kmult == 0 && d.ldb >= d.K, "gemm: K-contiguous B needs K %% %d == 0 (K=%d ldb=%d)", kmult, d.K, d.ldb);
    else DVAE_REQUIRE(d.N % 4 == 0 && d.ldb >= d.N, "gemm: K-major B needs N %% 4 == 0 (N=%d ldb=%d)", d.N, d.ldb);
    DVAE_REQUIRE(d.ldc >= d.N, "gemm: ldc < N");
    DVAE_REQUIRE(d.act >= 0 && d.act <= 2, "gemm: unknown activation %d", d.act);
    return DVAE_OK;
}

// K-major operands are addressed with 32-bit byte offsets (buffer loads): a contraction whose
// K-major operand spans more than 4 GiB runs as a sequence of accumulating K chunks.  Returns
// the chunk length (a multiple of `bk`; >= K when no chunking is needed) or a negative error.
int gemm_k_chunk(const dvae_gemm_desc& d, int bk) {
    const double lim = 4294967000.0;
    double rowb = 0.0;
    if (d.a_kmajor) rowb = d.lda * 4.0;
    if (d.b_kmajor && !d.b_rows && d.ldb * 4.0 > rowb) rowb = d.ldb * 4.0;
    if (rowb == 0.0 || (double)d.K * rowb < lim) return d.K > 0 ? d.K : 1;
    DVAE_REQUIRE(d.act == 0, "gemm: a chunked contraction cannot apply an activation");
    const int kc = (int)(lim / rowb) / bk * bk;
    DVAE_REQUIRE(kc >= bk, "gemm: leading dimension too large for a K-major operand");
    return kc;
}
// next chunk [k0, k0 + kc) of d -> sub; advances k0; false when done
bool gemm_next_chunk(const dvae_gemm_desc& d, int kc, int& k0, dvae_gemm_desc& sub) {
    if (k0 >= d.K) return false;
    sub = d;
    sub.K = d.K - k0 < kc ? d.K - k0 : kc;
    sub.A = d.a_kmajor ? d.A + (size_t)k0 * d.lda : d.A + k0;
    if (d.b_kmajor) {
        if (d.b_rows) sub.b_rows = d.b_rows + k0;
        else sub.B = d.B + (size_t)k0 * d.ldb;
    } else {
        sub.B = d.B + k0;
    }
    if (k0 > 0) { sub.accumulate = 1; sub.bias = nullptr; }
    k0 += sub.K;
    return true;
}

int gemm(const dvae_gemm_desc& d, hipStream_t s) {
    DVAE_TRY(gemm_check(d, 32));
    if (d.M == 0 || d.N == 0) return DVAE_OK;
    {
        const int kc = gemm_k_chunk(d, BK);
        if (kc < 0) return kc;
        if (kc < d.K) {
            dvae_gemm_desc sub;
            int k0 = 0;
            while (gemm_next_chunk(d, kc, k0, sub)) DVAE_TRY(gemm(sub, s));
            return DVAE_OK;
        }
    }

    GemmParams p;
    p.A = d.A; p.B = d.B; p.C = d.C;
    p.a_rows = d.a_rows; p.b_rows = d.b_rows; p.c_rows = d.c_rows;
    p.bias = d.bias; p.lda = d.lda; p.ldb = d.ldb; p.ldc = d.ldc;
    p.M = d.M; p.N = d.N; p.K = d.K; p.act = d.act; p.accumulate = d.accumulate;
    p.split_k = 1; p.k_per_split = 0;
    if (d.K == 0) {
        // empty contraction: result is bias/act only; run with kend == 0
    }
    // Tile and split-K choice.  The chip has 256 CUs x 4 SIMDs; a 128x128 tile keeps 4 waves
    // busy for K/32 steps of 64 MFMAs, a 64x64 tile for K/32 steps of 16 MFMAs.  Products that
    // accumulate into C (every dgrad / wgrad of the step) may split K across workgroups and
    // combine with fp32 atomics; the split is chosen to put >= ~2 workgroups on every CU
    // while keeping >= 4 K steps per workgroup.
    long t128 = (long)cdiv(d.M, 128) * cdiv(d.N, 128);
    long t64 = (long)cdiv(d.M, 64) * cdiv(d.N, 64);
    bool use128 = t128 >= 192 || (d.a_kmajor && t128 >= 24);
    int ksteps = cdiv(d.K, BK);
    const bool small_m = d.M <= 32 && d.a_kmajor;
    if (d.accumulate && d.bias == nullptr && d.act == 0 && ksteps >= 8) {
        // Split choice, fitted to measurements on MI355X (tools/gemm_quant.py, tools/gemm_bench.py):
        // run time advances in whole ROUNDS of 256 workgroups (a launch of 528 costs as much as one
        // of 768).  128x128 tiles: two workgroups per CU hide each other's load latency, so fill
        // 512 slots without overshooting; with more than one round of tiles already, pick the split
        // that minimises rounds x (K steps + ~5 steps of prologue and atomic epilogue).  64x64 tiles
        // (four fit on a CU): three per CU.
        const long tiles = small_m ? (long)cdiv(d.N, 128) : (use128 ? t128 : t64);
        const int maxsplit = ksteps / 4;
        int best = 1;
        if (use128 && !small_m && tiles <= 256) {
            best = (int)(512 / tiles);
        } else if (use128 && !small_m) {
            long best_cost = -1;
            for (int sk = 1; sk <= 16; ++sk) {
                const long rounds = (tiles * sk + 255) / 256;
                const long cost = rounds * (cdiv(ksteps, sk) + 5);
                if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = sk; }
            }
        } else {
            best = (int)((768 + tiles - 1) / tiles);
        }
        if (best > maxsplit) best = maxsplit;
        if (best < 1) best = 1;
        if (best > 1) {
            p.k_per_split = cdiv(ksteps, best) * BK;
            p.split_k = cdiv(d.K, p.k_per_split);
        }
    }
    if (small_m) return launch_cfg<32, 128, 1, 4>(p, d.a_kmajor, d.b_kmajor, s);
    if (use128) return launch_cfg<128, 128, 2, 2>(p, d.a_kmajor, d.b_kmajor, s);
    return launch_cfg<64, 64, 2, 2>(p, d.a_kmajor, d.b_kmajor, s);
}

}  // namespace dvae

extern "C" int dvae_gemm_f32(const dvae_gemm_desc* d, void* stream) {
    if (!d) return dvae::set_err(DVAE_ERR_ARG, "dvae_gemm_f32: null descriptor");
    return dvae::gemm(*d, (hipStream_t)stream);
}
