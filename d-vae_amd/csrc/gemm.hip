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
// buffers, one barrier per K step.  The K step is software pipelined by hand
// (scheduling fences pin the order): fragments are read one 8-k group ahead of
// their MFMAs, and the staging (16-byte unit: registers -> LDS, reload from
// tile k+2) is spread one unit per MFMA quad instead of in one burst.
#include "gemm_common.h"

namespace dvae {

constexpr int BK0 = 32;  // K step of the default kernels; the unit of the host-side K arithmetic

// Register-staged tile loader.  Everything that does not depend on the K step
// (row gather, row clamping, column guards) is resolved ONCE in init(); the K loop
// then issues UNITS independent 16-byte loads back to back with no branch between
// them.  The staging registers and row pointers are NAMED scalars (r0..r3, p0..p3),
// not arrays: as arrays hipcc left them in scratch memory across the loop back-edge
// (scratch_store / scratch_load of every staged tile, ScratchSize 144 B/lane).
template <int ROWS, bool KMAJOR, bool KGATHER, int BK>
struct TileLoader {
    static constexpr int C4R = BK / 4;    // float4 chunks per K-contiguous tile row
    static constexpr int KC_LD = BK + 4;  // K-contiguous LDS row stride (floats)
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
        lo_ = KMAJOR ? (tid / (ROWS / 4)) * ROWS + (tid % (ROWS / 4)) * 4 : (tid / C4R) * KC_LD + (tid % C4R) * 4;
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
    // LDS position of unit i = this thread's base (computed once in init) + a compile-time step,
    // so a store is one ds_write_b128 with an immediate offset and no address arithmetic in the loop
    int lo_;
    __device__ __forceinline__ void store1(float* lds, int i, int tid, float4 v) const {
        constexpr int STEP = KMAJOR ? 1024 : (256 / C4R) * KC_LD;  // floats between consecutive units
        // (the hint makes this one ds_write_b128; hipcc cannot prove the alignment and emits two b64)
        *reinterpret_cast<float4*>(__builtin_assume_aligned(lds + lo_ + i * STEP, 16)) = v;
    }
    // unit i alone: registers -> LDS image (stage_store), then the same registers are reloaded
    // from tile k0 (stage_load).  i is a constant after unrolling; a switch over named registers
    // keeps them out of scratch.
    __device__ __forceinline__ void stage_store(int i, float* lds, int tid) {
        switch (i) {
        case 0: store1(lds, 0, tid, r0); break;
        case 1: store1(lds, 1, tid, r1); break;
        case 2: store1(lds, 2, tid, r2); break;
        default: store1(lds, 3, tid, r3); break;
        }
    }
    __device__ __forceinline__ void stage_load(int i, int k0, int kend, int tid) {
        switch (i) {
        case 0: r0 = KMAJOR ? kload(0, k0, kend, tid, c0, col0_, n0_) : *reinterpret_cast<const float4*>(p0 + k0); break;
        case 1: r1 = KMAJOR ? kload(1, k0, kend, tid, c1, col1_, n1_) : *reinterpret_cast<const float4*>(p1 + k0); break;
        case 2: r2 = KMAJOR ? kload(2, k0, kend, tid, c2, col2_, n2_) : *reinterpret_cast<const float4*>(p2 + k0); break;
        default: r3 = KMAJOR ? kload(3, k0, kend, tid, c3, col3_, n3_) : *reinterpret_cast<const float4*>(p3 + k0); break;
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
template <int ROWS, bool KMAJOR, int BK>
__device__ __forceinline__ float4 frag(const float* lds, int roff, int g, int lane) {
    constexpr int KC_LD = BK + 4;
    int h = lane >> 5, r = roff + (lane & 31);
    if (!KMAJOR) {
        return *reinterpret_cast<const float4*>(lds + r * KC_LD + g * 8 + 4 * h);
    } else {
        const float* p = lds + (g * 8 + 4 * h) * ROWS + r;
        return make_float4(p[0], p[ROWS], p[2 * ROWS], p[3 * ROWS]);
    }
}

template <int BM, int BN, int WM, int WN, bool AKM, bool BKM, bool KGATHER, bool SWAP, int BK>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmParams p) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert(WM * WN == 4, "4 waves");
    using LA = TileLoader<BM, AKM, false, BK>;  // only the B operand of a weight-gradient product is row-gathered along K
    using LB = TileLoader<BN, BKM, KGATHER, BK>;
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
    // The first K tile of a workgroup's NEXT output tile is requested BEFORE the current tile's epilogue: vector
    // memory operations complete in issue order, so loads issued behind the epilogue's 64 stores per lane would
    // wait for all of them; issued ahead, they land while the stores drain and the next tile starts at once.
    LA la; LB lb;
    int m0 = 0, n0 = 0;
    auto tile = [&](int t) { int k = kbeg + t * BK; return k < kend ? k : kbeg; };  // clamped past the end (loaded, never used)
    auto begin_tile = [&](int tl) {
        tile_origin<BM, BN>(tl, p.tiles_x, ntiles, m0, n0);
        la.init(p.A, p.a_rows, p.lda, m0, p.M, tid);
        lb.init(p.B, p.b_rows, p.ldb, n0, p.N, tid);
        lb.next_idx(tile(0), kend, tid);
        la.load(tile(0), kend, tid);
        lb.load(tile(0), kend, tid);
        lb.next_idx(tile(1), kend, tid);
    };
    if ((int)blockIdx.x < ntiles) begin_tile(blockIdx.x);
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
    const int cm0 = m0, cn0 = n0;  // this tile's origin (begin_tile for the next one moves m0 / n0)

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    la.store(lds, tid);
    lb.store(lds + LA::LDS_ELEMS, tid);
    la.load(tile(1), kend, tid);
    lb.load(tile(1), kend, tid);
    lb.next_idx(tile(2), kend, tid);
    __syncthreads();

    // Steady state of step t (tile t in LDS[cur], tile t+1 in the staging registers, in flight
    // for a whole step), groups g = 0..NG-1 of 8 k each:
    //   * the fragments of group g+1 are read while the MFMAs of group g run; the last group reads
    //     group 0 of LDS[cur^1], i.e. of the next step;
    //   * after each MFMA quad of groups 0..NG-2 one staging unit goes registers -> LDS[cur^1] and
    //     is reloaded from tile t+2 (a burst of 16 LDS stores + 8 loads here cost 10 % of the MFMA
    //     rate at 4096^3; hipcc re-clusters them unless fenced);
    //   * ONE barrier, before the last group: by then every wave holds its last fragments of
    //     LDS[cur] in registers (so step t+1 may overwrite it) and LDS[cur^1] is complete.
    constexpr int NG = BK / 8;
    constexpr int NU = LA::UNITS + LB::UNITS;
    constexpr int SLOTS = (NG - 1) * TM * TN;
    constexpr int UPS = (NU + SLOTS - 1) / SLOTS;  // staging units per slot
    static_assert(UPS <= 2, "a slot stages at most two units");
    static_assert(NG % 2 == 0, "fragment double buffer assumes an even number of groups per step");
    float4 fa[2][TM], fb[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[0][i] = frag<BM, AKM, BK>(lds, (wm * TM + i) * 32, 0, lane);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[0][j] = frag<BN, BKM, BK>(lds + LA::LDS_ELEMS, (wn * TN + j) * 32, 0, lane);
    int cur = 0, t = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BK, ++t) {
        float* nl = lds + (cur ^ 1) * STAGE;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const float* src = (g < NG - 1) ? lds + cur * STAGE : nl;
            const int gn = (g + 1) % NG;
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[(g + 1) & 1][i] = frag<BM, AKM, BK>(src, (wm * TM + i) * 32, gn, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[(g + 1) & 1][j] = frag<BN, BKM, BK>(src + LA::LDS_ELEMS, (wn * TN + j) * 32, gn, lane);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    // SWAP: the block is accumulated transposed (same products, same k order), which
                    // gives each lane 4 consecutive columns of one row for the epilogue
                    const float4 a4 = SWAP ? fb[g & 1][j] : fa[g & 1][i], b4 = SWAP ? fa[g & 1][i] : fb[g & 1][j];
                    // one staging slot per MFMA quad of groups 0..NG-2: up to two units, each
                    // "registers -> LDS[cur^1]" behind one MFMA and "reload from tile k+2" behind the
                    // next, so that a wave alone on its SIMD issues them in the MFMAs' shadow
                    const int slot = (g * TM + i) * TN + j;
                    const int u0 = slot * UPS, u1 = slot * UPS + 1;
                    const bool on0 = g < NG - 1 && u0 < NU, on1 = g < NG - 1 && UPS > 1 && u1 < NU;
                    auto st = [&](int u) {
                        if (u < LA::UNITS) la.stage_store(u, nl, tid);
                        else lb.stage_store(u - LA::UNITS, nl + LA::LDS_ELEMS, tid);
                    };
                    auto ld = [&](int u) {
                        if (u < LA::UNITS) la.stage_load(u, tile(t + 2), kend, tid);
                        else lb.stage_load(u - LA::UNITS, tile(t + 2), kend, tid);
                        if (u == NU - 1) lb.next_idx(tile(t + 3), kend, tid);
                    };
                    // K-major operand pairs (weight gradients: buffer loads, b32 fragment reads) gain
                    // 10-15 % from placing each store / reload behind its own MFMA; K-contiguous
                    // ones run 1 % faster with the quad kept whole (measured both ways)
                    constexpr bool FINE = AKM && BKM;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc[i][j], 0, 0, 0);
                    if (FINE) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (on0) st(u0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc[i][j], 0, 0, 0);
                    if (FINE) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (on0) ld(u0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc[i][j], 0, 0, 0);
                    if (FINE) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (on1) st(u1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc[i][j], 0, 0, 0);
                    if (FINE) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (on1) ld(u1);
                    } else {
                        if (on0) { st(u0); ld(u0); }
                        if (on1) { st(u1); ld(u1); }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            if (g == NG - 2) __syncthreads();
        }
        cur ^= 1;
    }

    if (tl + (int)gridDim.x < ntiles) begin_tile(tl + gridDim.x);
    store_acc<TM, TN, SWAP>(p, acc, cm0, cn0, wm, wn, lane, BM, BN);
    }  // tiles
}

template <int BM, int BN, int WM, int WN, int BK>
static int launch_cfg(GemmParams p, int akm, int bkm, hipStream_t s) {
    p.tiles_x = cdiv(p.N, BN);
    p.tiles_y = cdiv(p.M, BM);
    const long ntiles = (long)p.tiles_x * p.tiles_y;
    const long cap = (BM * BN * BK >= 64 * 64 * 64 ? 512 : 1024);  // co-resident workgroups the chip holds (multiple of 8: XCD runs)
    dim3 grid((unsigned)(ntiles < cap ? ntiles : cap), 1, p.split_k > 1 ? p.split_k : 1);
    dim3 block(256);
    const bool kg = bkm && p.b_rows != nullptr;
    if (akm && p.a_rows) return set_err(DVAE_ERR_ARG, "gemm: row gather on a K-major A operand is not built");
#define DVAE_GEMM_LAUNCH(A_, B_, G_)                                                                             \
    do {                                                                                                         \
        if (vec) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, A_, B_, G_, true, BK>), grid, block, 0, s, p);  \
        else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, A_, B_, G_, false, BK>), grid, block, 0, s, p);     \
    } while (0)
    const bool vec = gemm_vec_epilogue_ok(p);
    if (!akm && !bkm) DVAE_GEMM_LAUNCH(false, false, false);
    else if (!akm && bkm && !kg) DVAE_GEMM_LAUNCH(false, true, false);
    else if (akm && bkm && !kg) DVAE_GEMM_LAUNCH(true, true, false);
    else if (akm && bkm && kg) DVAE_GEMM_LAUNCH(true, true, true);
    else return set_err(DVAE_ERR_ARG, "gemm: this operand layout / gather combination is not built");
#undef DVAE_GEMM_LAUNCH
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}

// Span (bytes from the base pointer) a K-major operand may cover: 32-bit buffer offsets.  Tests lower it
// through dvae_debug_set_kmajor_span_limit to run the chunked path and the gather guard at small sizes.
static double g_kmajor_span_limit = 4294967000.0;
double kmajor_span_limit() { return g_kmajor_span_limit; }

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
    DVAE_REQUIRE(!d.relu_mask || !d.accumulate, "gemm: relu_mask cannot be combined with accumulate");
    if (d.b_kmajor && d.b_rows) {
        // a row-gathered K-major operand is addressed as (row index * ldb + column) * 4 in 32 bits: the rows the
        // index list may name must all lie within the addressable span (chunking over K cannot rebase a gather)
        DVAE_REQUIRE(d.b_rows_bound > 0, "gemm: a row-gathered K-major B needs b_rows_bound (rows of the stored matrix)");
        DVAE_REQUIRE((double)d.b_rows_bound * d.ldb * (d.b_bf16 ? 2.0 : 4.0) < kmajor_span_limit(),
                     "gemm: row-gathered K-major B spans %.0f bytes (b_rows_bound %d x ldb %d floats), limit %.0f",
                     (double)d.b_rows_bound * d.ldb * 4.0, d.b_rows_bound, d.ldb, kmajor_span_limit());
    }
    return DVAE_OK;
}

// K-major operands are addressed with 32-bit byte offsets (buffer loads): a contraction whose
// K-major operand spans more than 4 GiB runs as a sequence of accumulating K chunks.  Returns
// the chunk length (a multiple of `bk`; >= K when no chunking is needed) or a negative error.
int gemm_k_chunk(const dvae_gemm_desc& d, int bk) {
    const double lim = kmajor_span_limit();
    double rowb = 0.0;
    const double ea = d.a_bf16 ? 2.0 : 4.0, eb = d.b_bf16 ? 2.0 : 4.0;  // bytes per stored element
    if (d.a_kmajor) rowb = d.lda * ea;
    if (d.b_kmajor && !d.b_rows && d.ldb * eb > rowb) rowb = d.ldb * eb;
    if (rowb == 0.0 || (double)d.K * rowb < lim) return d.K > 0 ? d.K : 1;
    DVAE_REQUIRE(d.act == 0 && !d.relu_mask, "gemm: a chunked contraction cannot apply an activation or a mask");
    const int kc = (int)(lim / rowb) / bk * bk;
    DVAE_REQUIRE(kc >= bk, "gemm: leading dimension too large for a K-major operand");
    return kc;
}
// next chunk [k0, k0 + kc) of d -> sub; advances k0; false when done
bool gemm_next_chunk(const dvae_gemm_desc& d, int kc, int& k0, dvae_gemm_desc& sub) {
    if (k0 >= d.K) return false;
    sub = d;
    sub.K = d.K - k0 < kc ? d.K - k0 : kc;
    // (bf16-stored operands: the pointers are typed float, the elements are 2 bytes)
    auto adv = [](const float* p, size_t elems, int half) {
        return half ? reinterpret_cast<const float*>(reinterpret_cast<const uint16_t*>(p) + elems) : p + elems;
    };
    sub.A = adv(d.A, d.a_kmajor ? (size_t)k0 * d.lda : (size_t)k0, d.a_bf16);
    if (d.b_kmajor) {
        if (d.b_rows) sub.b_rows = d.b_rows + k0;
        else sub.B = adv(d.B, (size_t)k0 * d.ldb, d.b_bf16);
    } else {
        sub.B = adv(d.B, (size_t)k0, d.b_bf16);
    }
    if (k0 > 0) { sub.accumulate = 1; sub.bias = nullptr; }
    // (a masked product is never chunked: gemm_k_chunk refuses an activation, and the mask is applied with it)
    k0 += sub.K;
    return true;
}

// tuning / tracing knobs (tools/gemm_tune.py): DVAE_GEMM_LOG=1 prints every product's shape to
// stderr, DVAE_GEMM_TILE={128,64,32} and DVAE_GEMM_SPLIT=n override the tile and split-K choice
static int env_int(const char* name) {
    const char* v = getenv(name);
    return v ? atoi(v) : 0;
}

int gemm(const dvae_gemm_desc& d, hipStream_t s, int max_split) {
    static const int knob_log = env_int("DVAE_GEMM_LOG"), knob_tile = env_int("DVAE_GEMM_TILE"),
                     knob_split = env_int("DVAE_GEMM_SPLIT");
    DVAE_TRY(gemm_check(d, 32));
    DVAE_REQUIRE(!d.a_bf16 && !d.b_bf16 && !d.c_bf16, "gemm_f32: bf16-stored operands / outputs belong to dvae_gemm_bf16");
    if (knob_log)
        fprintf(stderr, "GEMM %d %d %d akm=%d bkm=%d acc=%d bias=%d act=%d ar=%d br=%d cr=%d\n", d.M, d.N, d.K, d.a_kmajor,
                d.b_kmajor, d.accumulate, d.bias != nullptr, d.act, d.a_rows != nullptr, d.b_rows != nullptr,
                d.c_rows != nullptr);
    if (d.M == 0 || d.N == 0) return DVAE_OK;
    {
        const int kc = gemm_k_chunk(d, BK0);
        if (kc < 0) return kc;
        if (kc < d.K) {
            dvae_gemm_desc sub;
            int k0 = 0;
            while (gemm_next_chunk(d, kc, k0, sub)) DVAE_TRY(gemm(sub, s, max_split));
            return DVAE_OK;
        }
    }

    if (knob_tile == 0 && knob_split == 0) {  // products of few rows: 16-row tiles, the waves as K slices (gemm_skinny.hip)
        const int rc = gemm_skinny(d, s, 0);
        if (rc != 1) return rc;
    }

    GemmParams p;
    p.A = d.A; p.B = d.B; p.C = d.C;
    p.a_rows = d.a_rows; p.b_rows = d.b_rows; p.c_rows = d.c_rows;
    p.bias = d.bias; p.relu_mask = d.relu_mask; p.lda = d.lda; p.ldb = d.ldb; p.ldc = d.ldc;
    p.M = d.M; p.N = d.N; p.K = d.K; p.act = d.act; p.accumulate = d.accumulate;
    p.split_k = 1; p.k_per_split = 0; p.c_bf16 = 0;
    if (d.K == 0) {
        // empty contraction: result is bias/act only; run with kend == 0
    }
    // Tile and split-K choice: a small cost model fitted to sweeps of every product of the train
    // step under forced tiles / splits on MI355X (tools/gemm_tune.py; cfg 2 and cfg 5 shapes).
    // Unit = one K step of a 64x64 tile (16 MFMAs per wave, 0.43 us).
    //   * run time advances in ROUNDS of 256 workgroups (one per CU; co-resident workgroups share
    //     the MFMA pipe), so cost = rounds x (K steps per workgroup x step cost + epilogue);
    //   * a 128x128 step costs 4 units, a 64x64 step 1.2 (half the operand reuse: more L2 traffic);
    //     a tile kind running with fewer co-resident workgroups than hide its latencies is slower;
    //   * epilogue + prologue: 16-byte stores 1.5 / 6 units (64 / 128 tile), split-K atomics 3 / 38.
    // Products that accumulate into C with no bias / activation (every dgrad / wgrad of the step)
    // may split K over workgroups that combine with fp32 atomics.
    long t128 = (long)cdiv(d.M, 128) * cdiv(d.N, 128);
    long t64 = (long)cdiv(d.M, 64) * cdiv(d.N, 64);
    int ksteps = cdiv(d.K, BK0);
    bool small_m = d.M <= 32 && d.a_kmajor;
    const bool may_split = d.accumulate && d.bias == nullptr && d.act == 0 && ksteps >= 8;
    bool use128 = false;
    int best = 1;
    {
        double best_cost = -1.0;
        int maxsplit = may_split ? (ksteps / 4 < 16 ? ksteps / 4 : 16) : 1;
        if (max_split > 0 && maxsplit > max_split) maxsplit = max_split;
        for (int big = 0; big < 2; ++big) {
            const long tiles = big ? t128 : t64;
            for (int sk = 1; sk <= maxsplit; ++sk) {
                const long wgs = tiles * sk;
                const long rounds = (wgs + 255) / 256;
                const double occ = (double)wgs / 256.0;
                const double steps = (double)cdiv(ksteps, sk);
                double step_cost, epi;
                if (big) {
                    step_cost = 4.0 * (occ < 1.5 ? 1.12 : 1.0);
                    epi = sk > 1 ? 38.0 : 6.0;
                } else {
                    step_cost = 1.2 * (occ < 1.5 ? 1.3 : (occ < 2.5 ? 1.1 : 1.0));
                    epi = sk > 1 ? 3.0 : 1.5;
                }
                const double cost = (double)rounds * (steps * step_cost + epi);
                if (best_cost < 0.0 || cost < best_cost) { best_cost = cost; best = sk; use128 = big != 0; }
            }
        }
    }
    if (knob_tile == 128) { use128 = true; small_m = false; }
    else if (knob_tile == 64) { use128 = false; small_m = false; }
    else if (knob_tile == 32) small_m = true;
    if (small_m && may_split) {
        // one row block (M <= 32), 32x128 tiles: three workgroups per CU
        const long tiles = (long)cdiv(d.N, 128) * cdiv(d.M, 32);
        best = (int)((768 + tiles - 1) / tiles);
        if (best > ksteps / 4) best = ksteps / 4;
    } else if (small_m) {
        best = 1;
    }
    if (knob_split > 0 && may_split) best = knob_split < ksteps ? knob_split : ksteps;
    if (best < 1) best = 1;
    if (best > 1) {
        p.k_per_split = cdiv(ksteps, best) * BK0;
        p.split_k = cdiv(d.K, p.k_per_split);
    }
    // (a K step of 64 for the 64x64 tile was measured too: 6 % on the 1.5-tiles-per-CU GRU product,
    // nothing elsewhere, 1 % slower train step; only the 32-k kernels are built)
    if (knob_log)  // one line per LAUNCH (tools/roofline_by_kernel.py maps launches to kernel families with it)
        fprintf(stderr, "GEMMSEL %s %d %d %d akm=%d bkm=%d split=%d\n", small_m ? "32x128" : use128 ? "128x128" : "64x64", d.M, d.N,
                d.K, d.a_kmajor, d.b_kmajor, p.split_k);
    if (small_m) return launch_cfg<32, 128, 1, 4, 32>(p, d.a_kmajor, d.b_kmajor, s);
    if (use128) return launch_cfg<128, 128, 2, 2, 32>(p, d.a_kmajor, d.b_kmajor, s);
    return launch_cfg<64, 64, 2, 2, 32>(p, d.a_kmajor, d.b_kmajor, s);
}

}  // namespace dvae

extern "C" int dvae_debug_set_kmajor_span_limit(int64_t bytes) {
    DVAE_REQUIRE(bytes >= 0 && (double)bytes <= 4294967000.0, "span limit must be in 0 .. 4294967000 (0 restores the default)");
    dvae::g_kmajor_span_limit = bytes == 0 ? 4294967000.0 : (double)bytes;
    return DVAE_OK;
}

extern "C" int dvae_gemm_f32(const dvae_gemm_desc* d, void* stream) {
    if (!d) return dvae::set_err(DVAE_ERR_ARG, "dvae_gemm_f32: null descriptor");
    return dvae::gemm(*d, (hipStream_t)stream, 0);
}
