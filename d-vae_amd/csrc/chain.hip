// Level kernels of the state chain (fp32 mode): a level's product AND its pointwise tail in ONE
// launch, so the [rows x 3Hp] / [rows x 2Hp] pre-activations never exist in HBM, a level of the
// forward chain is 2 launches instead of 4, and the backward products need no split-K atomics.
//
//   chain_kernel<3, false, 0>:  (r, z, n, W_hn h + b_hn, h') = GRUCell tail of  h_prev W_hh^T
//   chain_kernel<2, false, 1>:  (g, m, msg = g*m, next level's predecessor sum) of  h [Wg | Wm]^T
//   chain_kernel<1, true, 2>:   dX += dY W   (d agg += d gh W_hh), then optionally the gate/mapper pointwise
//                               backward of the level processed next (its d msg gathers dX rows)
//   chain_kernel<1, true, 3>:   d h = dX + dY W, then the GRUCell pointwise backward of d h
// A forward level is 2 launches (GEMM + pointwise + GEMM + pointwise/gather before), a backward level 2
// (gather + split-K GEMM + pointwise + split-K GEMM before).
//
// Why kernels of their own instead of epilogues of gemm_f32_kernel: the forward tails need the r, z
// and n (or g and m) pre-activations of one (row, hidden unit) in ONE lane, i.e. columns j, Hp + j,
// 2Hp + j of the product in one wave; and the chain's shape (1024 rows, Hp = 512) is 1.5 rounds of
// 64 x 64 tiles (GRU) or needs split-K (backward: K = 2Hp / 3Hp, N = Hp).  Here a workgroup owns
// 64 rows x 32 hidden units x all NGATE gates (B rows from NGATE row blocks of the weight) -> 256
// workgroups = one per CU at that shape, and its 8 waves are (K quarter) x (row half): every wave
// contracts a quarter of each 64-deep K step for its 32 rows and all gates, so each SIMD holds two
// waves whose MFMA streams fill each other's waits (a lone wave per SIMD ran the same loop at 60 %
// of the MFMA rate).  The four K quarters meet through LDS in the epilogue and each wave finishes
// one register quad (8 columns x 32 rows as 16-byte accesses).  Deterministic: no atomics, fixed
// summation order.
//
// MFMA: v_mfma_f32_32x32x2_f32 (exact fp32), accumulated transposed (operands swapped) so a lane
// holds 4 consecutive hidden units of one row per register quad.  LDS images as in gemm.hip:
// K-contiguous operands [rows][68] floats read with ds_read_b128, the K-major weight of the
// backward products [64 k][40] read with ds_read_b32; two stages, register-staged global loads,
// one barrier per K step.
#include "chain_common.h"

namespace dvae {

// Two shapes of workgroup, template parameter KS = number of K slices of a step:
//   KS = 4: 512 threads, 64-deep K steps, 87 KB LDS (GRU) -- two waves per SIMD inside ONE workgroup: the default
//   KS = 2: 256 threads, 32-deep K steps, 46 KB LDS -- two or three workgroups can share a CU; measured
//           slower (forward level 42 vs 36 us alone, and no gain with two directions co-resident:
//           workgroups launched together run in lockstep); kept behind DVAE_CHAIN_KS=2 and tested
// Either way a wave contracts 16 k (two groups of 8) of each step for its 32 rows and all gates.
constexpr int CBM = 64;        // rows per workgroup
constexpr int CBH = 32;        // hidden units (output columns per gate) per workgroup
constexpr int CLW = 40;        // LDS row stride of the K-major W image [BK k][32 cols]: lane halves 4 k rows apart hit disjoint banks


template <int V>
struct IC { static constexpr int value = V; };  // compile-time register-set index for the staging lambdas
// (The K loop has NO branch between its steps: with conditional loads hipcc's wait-count insertion
// falls back to s_waitcnt vmcnt(0) before every LDS store, which drains all the tiles in flight.)


// ---------------------------------------------------------------------------------- epilogue
// TAIL: 0 GRU forward (NGATE 3), 1 gate/mapper forward (NGATE 2), 2 dX += (NGATE 1), 3 dX += then GRU backward.
// The wave of K slice KQ finishes 4 / KS register quads (quad q = columns 8 q + 4 (lane / 32) .. + 3):
// it hands the other quads to their owners through LDS, adds the partial sums it receives and
// its own IN K ORDER, and runs the tail.
template <int NGATE, int TAIL, int KS, int KQ, bool SC1>
__device__ __forceinline__ void chain_epilogue(const ChainParams& p, f32x16 (&acc)[NGATE], float* lds, int m0, int n0,
                                               int wm, int lane) {
    constexpr int QPW = 4 / KS;  // register quads a wave finishes: quads KQ * QPW .. + QPW - 1
    float4* xch = reinterpret_cast<float4*>(lds);
    // slot of (destination quad q, source slice k != owner of q): ((q * (KS-1) + (k < owner ? k : k - 1)) * 2 + wm) * NGATE + gate
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int owner = q / QPW;
        if (owner == KQ) continue;
        const int src = KQ < owner ? KQ : KQ - 1;
#pragma unroll
        for (int i = 0; i < NGATE; ++i)
            xch[(((q * (KS - 1) + src) * 2 + wm) * NGATE + i) * 64 + lane] =
                make_float4(acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]);
    }
    __syncthreads();
    const int row = m0 + wm * 32 + (lane & 31), h = lane >> 5;
    if (row >= p.M) return;
#pragma unroll
    for (int ql = 0; ql < QPW; ++ql) {
        const int q = KQ * QPW + ql;
        float4 v[NGATE];
#pragma unroll
        for (int i = 0; i < NGATE; ++i) {
            const float4 own = make_float4(acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]);
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < KS; ++k) {  // K slices 0 .. KS-1 in order
                float4 t;
                if (k == KQ) t = own;
                else t = xch[(((q * (KS - 1) + (k < KQ ? k : k - 1)) * 2 + wm) * NGATE + i) * 64 + lane];
                if (k == 0) s = t;
                else { s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
            }
            v[i] = s;
        }
        const int col = n0 + 8 * q + 4 * h;
        if (CHAIN_ABL & 1) {  // keep the sums alive, touch no tail operand
            if (v[0].x == 123.456f) p.C[0] = v[NGATE - 1].y;
            continue;
        }
        if (TAIL == 0) {
            gru_tail<SC1>(p, row, col, v[0], v[NGATE > 1 ? 1 : 0], v[NGATE - 1]);
        } else if (TAIL == 1) {
            gate_tail<SC1>(p, row, col, v[0], v[NGATE - 1]);
        } else {
            float* c = p.C + (long)row * p.ldc + col;
            const float4 c0 = cld4(c);
            const float4 d = make_float4(c0.x + v[0].x, c0.y + v[0].y, c0.z + v[0].z, c0.w + v[0].w);
            cst4(c, d);
            if (TAIL == 3) gru_bwd_tail(p, row, col, d);
            if (TAIL == 2 && p.dgm_out) gather_bwd_tail(p, row, col, d);
        }
    }
}

// ---------------------------------------------------------------------------------- kernel
// NK > 0: the number of K steps, known at compile time -- the K loop is then unrolled completely
// (around a loop back-edge hipcc rotates the staging registers with moves that wait for every
// outstanding load, s_waitcnt vmcnt(0) once per trip).  NK == 0: any K, runtime loop, PF = 1.
// K % 64 may be 32: the second half of the last step then lies past K; its loads are redirected
// to valid addresses and the A half is stored to LDS as zeros (the weight half stays finite).
//
// WKM = false: forward product, W [NGATE * Hp][K] K-contiguous (TAIL 0 / 1)
// WKM = true:  backward product, W [K][Hp] (NGATE = 1, TAIL 2 / 3)
// ND = 2: TWO independent levels of the same shape per workgroup (the two directions of the
// bidirectional encoder): the K loop alternates between them step by step, each with its own
// accumulators, so the per-workgroup latencies (cold first tile, epilogue) are paid once for both
// while the MFMA work per workgroup doubles.  (As two grids -- blockIdx.y, or two streams -- the
// directions only serialize: the level kernels of the two cannot share a CU, and workgroups
// launched together run in lockstep.)
struct ChainBatch { ChainParams d[2]; };

// The loop runs over steps s = 0 .. nk * ND - 1: level s % ND, tile s / ND, LDS stage s & 1; the data of
// step s + 1 sit in register set (s + 1) % PF.  Unrolled in groups of U = lcm(PF, ND) steps so that
// sets and levels are compile-time constants (J = s % U).
template <int J, int U, class F>
__device__ __forceinline__ void chain_steps(F& step, int s) {
    if constexpr (J < U) {
        step(IC<J>{}, s + J);
        chain_steps<J + 1, U>(step, s);
    }
}
template <int J, int PF, class F>
__device__ __forceinline__ void chain_prefetch(F& gload) {
    if constexpr (J < PF) {
        gload(IC<J>{}, J);
        chain_prefetch<J + 1, PF>(gload);
    }
}

template <int NGATE, bool WKM, int KS>
struct ChainLds {
    static constexpr int CBK = 16 * KS;
    static constexpr int CLD = CBK + 4;
    static constexpr int A_ELEMS = CBM * CLD;
    static constexpr int W_ELEMS = WKM ? CBK * CLW : NGATE * 32 * CLD;
    static constexpr int STAGE = A_ELEMS + W_ELEMS;
    static constexpr int FLOATS = 2 * STAGE;
};

// One level operation of one output tile (workgroup `bid` of tiles_x * tiles_y): the body of chain_kernel, also
// called phase by phase by the persistent kernel below (SC1: the A operand and the tail's hand-off arrays are
// accessed write-through, see ld4_sc1).  `lds` holds ChainLds<NGATE, WKM, KS>::FLOATS floats.
template <int NGATE, bool WKM, int TAIL, int KS, int PF, int NK, int ND, bool SC1>
__device__ __forceinline__ void chain_body(const ChainBatch& batch, float* lds, int bid) {
    const ChainParams& p = batch.d[0];  // shape (M, Hp, K, tiles) is common to the levels of a launch
    constexpr int CBK = 16 * KS;   // K step: KS slices of 16, one per wave pair
    constexpr int CLD = CBK + 4;   // LDS row stride of a K-contiguous image (floats): ds_read_b128 conflict-free for 36 and 68
    constexpr int C4R = CBK / 4;   // 16-byte units per K-contiguous tile row
    constexpr int A_ELEMS = CBM * CLD;
    constexpr int W_ELEMS = WKM ? CBK * CLW : NGATE * 32 * CLD;
    constexpr int STAGE = A_ELEMS + W_ELEMS;
    constexpr int WU = WKM ? 1 : NGATE;  // weight staging units (16 bytes) per thread; the A operand has 2
    constexpr int U = (PF % ND == 0) ? PF : PF * ND;  // lcm(PF, ND) for ND in {1, 2}
    static_assert(!SC1 || ND == 1, "the write-through form runs one level per call");
    static_assert(KS == 2 || KS == 4, "two or four K slices");
    static_assert(ND == 1 || ND == 2, "one or two levels per workgroup");
    static_assert(2 * STAGE * 4 >= 4 * (KS - 1) * 2 * NGATE * 64 * 16, "the K-slice exchange reuses the stage buffers");
    static_assert(ChainLds<NGATE, WKM, KS>::FLOATS == 2 * STAGE, "LDS size");
    static_assert((NK * ND) % U == 0 && (NK > 0 || PF == 1), "the unroll group must divide the step count");
    static_assert(!WKM || NGATE == 1, "backward products have one output block per wave");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kq = wave >> 1, wm = wave & 1;
    int m0, n0;
    tile_origin<CBM, CBH>(bid, p.tiles_x, p.tiles_x * p.tiles_y, m0, n0);

    // A: thread -> rows tid/C4R and tid/C4R + 32, 16 bytes at k (tid%C4R)*4
    // W K-contiguous: rows tid/C4R of each gate block, the same k;  W K-major: k row tid/8, columns (tid%8)*4
    const int ar = tid / C4R, ak = (tid % C4R) * 4;
    const float *pa0[ND], *pa1[ND], *pw0[ND], *pw1[ND], *pw2[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        const ChainParams& q = batch.d[d];
        int r = m0 + ar;
        r = r < p.M ? r : p.M - 1;  // clamped rows are computed but never stored
        pa0[d] = q.A + (long)r * q.lda;
        r = m0 + ar + 32;
        r = r < p.M ? r : p.M - 1;
        pa1[d] = q.A + (long)r * q.lda;
        if (!WKM) {
            pw0[d] = q.W + (long)(n0 + ar) * q.ldw;
            pw1[d] = pw0[d] + (long)p.Hp * q.ldw;
            pw2[d] = pw1[d] + (long)p.Hp * q.ldw;
        } else {
            pw0[d] = q.W + n0 + (tid & 7) * 4;
            pw1[d] = pw2[d] = pw0[d];
        }
    }
    // SC1: the A operand (states written by other workgroups of this launch) is read through a buffer descriptor with
    // the sc1 bit; the byte offsets of the thread's two rows replace the row pointers
    const __amdgpu_buffer_rsrc_t rsa = rsrc_of(batch.d[0].A);
    unsigned oa0 = 0, oa1 = 0;
    if (SC1) {
        int r = m0 + ar;
        r = r < p.M ? r : p.M - 1;
        oa0 = (unsigned)(r * p.lda * 4);
        r = m0 + ar + 32;
        r = r < p.M ? r : p.M - 1;
        oa1 = (unsigned)(r * p.lda * 4);
    }
    const int sa = ar * CLD + ak;
    const int sw = A_ELEMS + (WKM ? (tid >> 3) * CLW + (tid & 7) * 4 : ar * CLD + ak);
    const int nk = NK > 0 ? NK : (p.K + CBK - 1) / CBK;
    const int ns_total = nk * ND;
    // (Rotating each workgroup's K order by its column tile, so that workgroups walking K in step do not
    // read the same k-slab at the same moment, was measured: no change -- L2 hit rate 82 %, see profiles/.)
    auto tile_of = [&](int s) { const int t = s / ND; return t < nk ? t : nk - 1; };  // steps past the end reload the last tile
    float4 rs[PF][2 + WU];
    // loads the data of step s into register set (s % PF); s = J (mod U) gives set and level statically
    auto gload = [&](auto Jc, int s) {
        constexpr int s_ = decltype(Jc)::value % PF, d_ = decltype(Jc)::value % ND;
        const int k0 = tile_of(s) * CBK;
        int ka = k0 + ak;
        ka = ka < p.K ? ka : ka - 32;
        if (SC1) {
            rs[s_][0] = ld4_sc1(rsa, oa0 + (unsigned)ka * 4u);
            rs[s_][1] = ld4_sc1(rsa, oa1 + (unsigned)ka * 4u);
        } else {
            rs[s_][0] = cld4(pa0[d_] + ka);
            rs[s_][1] = cld4(pa1[d_] + ka);
        }
        if (!WKM) {
            rs[s_][2] = cld4(pw0[d_] + ka);
            if (NGATE > 1) rs[s_][WU > 1 ? 3 : 2] = cld4(pw1[d_] + ka);
            if (NGATE > 2) rs[s_][WU > 2 ? 4 : 2] = cld4(pw2[d_] + ka);
        } else {
            int kw = k0 + (tid >> 3);
            kw = kw < p.K ? kw : kw - 32;
            rs[s_][2] = cld4(pw0[d_] + (long)kw * batch.d[d_].ldw);
        }
    };
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto lstore = [&](auto Jc, float* st, int s) {  // register set (s % PF) -> LDS stage
        constexpr int s_ = decltype(Jc)::value % PF;
        const bool past = tile_of(s) * CBK + ak >= p.K;  // only in the last tile of a K with K % 64 == 32 (KS = 4)
        cst4(st + sa, past ? zero4 : rs[s_][0]);
        cst4(st + sa + 32 * CLD, past ? zero4 : rs[s_][1]);
        cst4(st + sw, rs[s_][2]);
        if (!WKM && NGATE > 1) cst4(st + sw + 32 * CLD, rs[s_][WU > 1 ? 3 : 2]);
        if (!WKM && NGATE > 2) cst4(st + sw + 64 * CLD, rs[s_][WU > 2 ? 4 : 2]);
    };

    f32x16 acc[ND][NGATE];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int i = 0; i < NGATE; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[d][i][r] = 0.f;

    // steps 0 .. PF-1 -> sets 0 .. PF-1, step 0 -> LDS, step PF -> set 0
    gload(IC<0>{}, 0);
    chain_prefetch<1, PF>(gload);
    lstore(IC<0>{}, lds, 0);
    gload(IC<PF>{}, PF);
    __syncthreads();
    // fragment bases of this wave: row (lane & 31) of its row half / of each gate image (or column
    // lane & 31 of the K-major image), k offset of its K slice (2 groups of 8 k) and lane half
    const int foa = (wm * 32 + (lane & 31)) * CLD + kq * 16 + 4 * (lane >> 5);
    const int fow = A_ELEMS + (WKM ? (kq * 16 + 4 * (lane >> 5)) * CLW + (lane & 31)
                                   : (lane & 31) * CLD + kq * 16 + 4 * (lane >> 5));
    auto step = [&](auto Jc, int s) {
        constexpr int J = decltype(Jc)::value;
        constexpr int D = J % ND;  // level of the tile being contracted
        const float* cs = lds + (s & 1) * STAGE;
        float* ns = lds + ((s & 1) ^ 1) * STAGE;
        float4 fa[2], fw[2][NGATE];
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
            fa[gg] = cld4(cs + foa + gg * 8);
#pragma unroll
            for (int i = 0; i < NGATE; ++i) {
                if (!WKM) {
                    fw[gg][i] = cld4(cs + fow + i * 32 * CLD + gg * 8);
                } else {
                    const float* w = cs + fow + gg * 8 * CLW;
                    fw[gg][i] = make_float4(w[0], w[CLW], w[2 * CLW], w[3 * CLW]);
                }
            }
        }
#if CHAIN_ABL & 2
#define CHAIN_MFMA(a_, b_, c_) ({ f32x16 r_ = (c_); r_[0] += (a_) * (b_); r_; })
#else
#define CHAIN_MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x2f32((a_), (b_), (c_), 0, 0, 0)
#endif
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
#pragma unroll
            for (int i = 0; i < NGATE; ++i) acc[D][i] = CHAIN_MFMA(fw[gg][i].x, fa[gg].x, acc[D][i]);
#pragma unroll
            for (int i = 0; i < NGATE; ++i) acc[D][i] = CHAIN_MFMA(fw[gg][i].y, fa[gg].y, acc[D][i]);
            if (gg == 0) {
                // step s+1 (in flight for PF steps): registers -> the other stage; reload from step s+1+PF
                lstore(IC<J + 1>{}, ns, s + 1);
                gload(IC<J + 1 + PF>{}, s + 1 + PF);
            }
#pragma unroll
            for (int i = 0; i < NGATE; ++i) acc[D][i] = CHAIN_MFMA(fw[gg][i].z, fa[gg].z, acc[D][i]);
#pragma unroll
            for (int i = 0; i < NGATE; ++i) acc[D][i] = CHAIN_MFMA(fw[gg][i].w, fa[gg].w, acc[D][i]);
        }
        __syncthreads();  // stage s^1 complete; every wave holds its fragments of stage s in registers
    };
    if constexpr (NK > 0) {
#pragma unroll
        for (int s = 0; s < ((CHAIN_ABL & 4) ? U : NK * ND); s += U) chain_steps<0, U>(step, s);  // s % U == 0: sets and levels are constants
    } else {
        for (int s = 0; s < ns_total; s += U) chain_steps<0, U>(step, s);
    }
    // (wave-uniform branches: the quads a wave finishes must be compile-time register indices)
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        if (d > 0) __syncthreads();  // the exchange buffer is reused
        const ChainParams& q = batch.d[d];
        if (kq == 0) chain_epilogue<NGATE, TAIL, KS, 0, SC1>(q, acc[d], lds, m0, n0, wm, lane);
        else if (kq == 1) chain_epilogue<NGATE, TAIL, KS, 1, SC1>(q, acc[d], lds, m0, n0, wm, lane);
        else if constexpr (KS == 4) {
            if (kq == 2) chain_epilogue<NGATE, TAIL, KS, 2, SC1>(q, acc[d], lds, m0, n0, wm, lane);
            else chain_epilogue<NGATE, TAIL, KS, 3, SC1>(q, acc[d], lds, m0, n0, wm, lane);
        }
    }
}

template <int NGATE, bool WKM, int TAIL, int KS, int PF, int NK, int ND>
__global__ __launch_bounds__(128 * KS) void chain_kernel(ChainBatch batch) {
    __shared__ __attribute__((aligned(16))) float lds[ChainLds<NGATE, WKM, KS>::FLOATS];
    chain_body<NGATE, WKM, TAIL, KS, PF, NK, ND, false>(batch, lds, blockIdx.x);
}

// ---------------------------------------------------------------------------------- small levels
// The same level operations for FEW rows (batch 32: the reference's own training recipe, train.py: --batch-size 32).
// chain_kernel's workgroup owns 64 rows and walks all of K: at 32 rows half of its MFMA work is padding, only Hp / 32 =
// 16 workgroups exist, and a level costs the same 10 us of MFMA issue per workgroup as at 1024 rows.  Here a workgroup
// owns 32 rows x 32 hidden units x all gates, its 4 waves a quarter of K each, and the operands go from global memory
// STRAIGHT into MFMA fragments (lane (row | column, half h) loads 8 consecutive k of its row: no LDS image, no barrier
// in the K loop; the whole operand set of a level is a few MB and L2-resident).  The quarter sums meet through LDS in
// K order and run the same tails (chain_epilogue with one row block).  Deterministic; same arithmetic per element as
// chain_kernel up to the order of the K partial sums.
template <int NGATE, bool WKM, int TAIL>
__global__ __launch_bounds__(256) void chain_small_kernel(ChainParams p) {
    constexpr int KS = 4;
    __shared__ __attribute__((aligned(16))) float lds[4 * (KS - 1) * 2 * NGATE * 64 * 4];  // the K-slice exchange of chain_epilogue
    const int tid = threadIdx.x, lane = tid & 63, kq = tid >> 6;
    const int n0 = (blockIdx.x % p.tiles_x) * CBH, m0 = (blockIdx.x / p.tiles_x) * 32;
    const int h = lane >> 5;
    int row = m0 + (lane & 31);
    row = row < p.M ? row : p.M - 1;  // clamped rows are computed, never stored
    const float* pa = p.A + (long)row * p.lda;
    const int col = n0 + (lane & 31);
    f32x16 acc[NGATE];
#pragma unroll
    for (int i = 0; i < NGATE; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // K is walked EXACTLY as chain_kernel walks it, so that the two kernels give the same bits: steps of 64 k, wave kq
    // the 16 k at 64 s + 16 kq of every step s; within them lane half h holds k + 8 c + 4 h + {0..3} (c = 0, 1) and MFMA
    // (c, j) contracts the pair {8 c + j, 8 c + 4 + j}; a K with K % 64 == 32 ends in half a step (zeros in chain_kernel's
    // LDS image, skipped here).
    struct Frag { float4 a[2]; float4 w[NGATE][2]; };
    const int nk = (p.K + 63) / 64;
    auto load = [&](Frag& f, int st) {
        const int k0 = 64 * st + 16 * kq;
        const bool ok = st < nk && k0 < p.K;
        const int k = (ok ? k0 : 0) + 4 * h;  // past the end: a valid address (K >= 32), the A half is zero
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            f.a[c] = ok ? cld4(pa + k + 8 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < NGATE; ++i) {
                if (!WKM) {
                    f.w[i][c] = cld4(p.W + (long)(i * p.Hp + col) * p.ldw + k + 8 * c);
                } else {
                    const float* w = p.W + (long)(k + 8 * c) * p.ldw + col;
                    f.w[i][c] = make_float4(w[0], w[p.ldw], w[2 * (long)p.ldw], w[3 * (long)p.ldw]);
                }
            }
        }
    };
    auto mma = [&](const Frag& f) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int i = 0; i < NGATE; ++i) acc[i] = CHAIN_MFMA(f.w[i][c].x, f.a[c].x, acc[i]);
#pragma unroll
            for (int i = 0; i < NGATE; ++i) acc[i] = CHAIN_MFMA(f.w[i][c].y, f.a[c].y, acc[i]);
#pragma unroll
            for (int i = 0; i < NGATE; ++i) acc[i] = CHAIN_MFMA(f.w[i][c].z, f.a[c].z, acc[i]);
#pragma unroll
            for (int i = 0; i < NGATE; ++i) acc[i] = CHAIN_MFMA(f.w[i][c].w, f.a[c].w, acc[i]);
        }
    };
    // two steps in flight ahead of the one being contracted
    Frag f0, f1, f2;
    load(f0, 0);
    load(f1, 1);
    for (int st = 0; st < nk; st += 3) {
        load(f2, st + 2);
        mma(f0);
        if (st + 1 < nk) {
            load(f0, st + 3);
            mma(f1);
        }
        if (st + 2 < nk) {
            load(f1, st + 4);
            mma(f2);
        }
    }
    if (kq == 0) chain_epilogue<NGATE, TAIL, KS, 0, false>(p, acc, lds, m0, n0, 0, lane);
    else if (kq == 1) chain_epilogue<NGATE, TAIL, KS, 1, false>(p, acc, lds, m0, n0, 0, lane);
    else if (kq == 2) chain_epilogue<NGATE, TAIL, KS, 2, false>(p, acc, lds, m0, n0, 0, lane);
    else chain_epilogue<NGATE, TAIL, KS, 3, false>(p, acc, lds, m0, n0, 0, lane);
}

// The same once more for the FEWEST rows (batch 32, the reference's recipe): chain_small_kernel's 32 x 32 tiles give only
// Hp / 32 = 16 workgroups there, each with 5.1 us of MFMA issue (a quarter of K x all gates per wave at one SIMD's fp32
// rate) inside a 12-13 us launch.  v_mfma_f32_16x16x4_f32 tiles -- 16 rows x 16 hidden units x all gates per workgroup,
// 4 waves = 4 interleaved K slices -- make it (Hp / 16) x ceil(rows / 16) = 64 workgroups with a quarter of that work
// each.  Lane (i = lane % 16, kg = lane / 16) loads 4 consecutive k of row i of each operand at 16 s + 4 kg; MFMA j of a
// step contracts k = 16 s + 4 kg + j over kg.  The K order differs from chain_kernel's (results agree to rounding, not
// bit for bit).  The four slice sums meet in LDS in slice order; wave 0 runs the 64 (row, 4 hidden units) tails.
// KS = number of waves = K slices: 4 for the forward operations, 8 for the backward ones (K = 2Hp / 3Hp, one gate:
// twice the steps per wave and four strided dword loads of the K-major weight in front of each: 12.5 -> 9.5 and 8.3 -> 6.4 us).
// (body: chain_common.h, shared with the persistent small-batch chains of chain_persist.hip)
template <int NGATE, bool WKM, int TAIL, int KS>
__global__ __launch_bounds__(64 * KS) void chain_tiny_kernel(ChainParams p) {
    __shared__ float4 xch[KS * NGATE * 64];
    chain_tiny_body<NGATE, WKM, TAIL, KS, false>(p, xch, blockIdx.x);
}

// rows up to which chain_tiny_kernel serves a level (DVAE_CHAIN_TINY_M: 0 = never).  Measured on the train step (cfg 1
// model, same box, against chain_small_kernel / product + pointwise launches): B = 32 +17 %, 64 +20 %, 128 +17 %, 256 +15 %,
// 512 +-0 (2048 workgroups that re-read their operands from L2: the 64-row LDS-tiled kernels take over from there).
int chain_tiny_rows() {
    static const int knob = [] { const char* v = getenv("DVAE_CHAIN_TINY_M"); return v ? atoi(v) : 256; }();
    return knob;
}

// rows up to which chain_small_kernel serves a level (DVAE_CHAIN_SMALL_M: 0 = never).  Measured on the train step
// against GEMM + pointwise launches (cfg 1 / cfg 2 shapes, same box): B = 32 +10 %, 64 +9 %, 128 +1.5 %, 256 +-0.
int chain_small_rows() {
    static const int knob = [] { const char* v = getenv("DVAE_CHAIN_SMALL_M"); return v ? atoi(v) : 128; }();
    return knob;
}

// ---------------------------------------------------------------------------------- persistent encoder chain
// The whole forward chain of the encoder (models.py:675-685, both directions) as ONE launch.
//
// Graphs are independent, and level == vertex index keeps a graph in the same 64-row tile at every level; so a level
// needs no grid barrier: the only workgroups that exchange data are the tiles_x (16) column tiles of one row tile --
// the GRU product contracts over ALL hidden units of the previous level's sums `agg`, the gate/mapper product over all
// of `hs`, each written 32 columns at a time by those workgroups.  Per (direction, row tile) one monotonic counter:
// a workgroup adds 1 after each phase (GRU level, gate level, GRU level, ...), and waits for tiles_x * p before its
// phase p.  Hand-off form (cdna_hip_programming.md guideline 16, R1 / MI355X_MICROARCH.md visibility table, first row):
// payload stored write-through (sc1), every storing wave drains (s_waitcnt vmcnt(0)), workgroup barrier, ONE lane's
// agent-scope atomic add; consumer: one lane polls the counter with sc1 loads, workgroup barrier, then sc1 loads only.
// With two directions a workgroup alternates between them phase by phase, so the other direction's phase (10-15 us)
// lies between a signal and the wait that needs it: the hand-off latency is hidden, and no launch boundary, cold
// prologue burst or drain is paid per level.
// All tiles_x * tiles_y workgroups must be co-resident (512 threads, 87 KB of LDS: one per CU, so at most 256); every
// spin is bounded: on a timeout the workgroup raises status[0] and leaves, the others follow, and the host entry point's
// caller finds the flag (results are then invalid).  Never run two of these launches at once on one device.
// MEASURED (cfg 2, B = 1024, Hp = 512, 30 phases; profiles/r02_chain_ablation.txt): encoder forward 0.583 ms with this
// launch vs 0.578-0.600 ms with the 30 level launches on two streams; whole step 3.981 vs 3.965 ms.  No gain: a phase is
// 10.4 us of MFMA issue at 95 % of the fp32 matrix peak (the ablation below) + 3.9 us of tail, and the launch boundary
// this kernel removes was already hidden by the other direction's kernel on the second stream.  Starting the odd row
// tiles 1 / 2 / 3 us late (so that tails of one half fall into K loops of the other) changed nothing either.
// Kept behind DVAE_CHAIN_PERSIST=1 (default off), parity-tested in tests/test_gpu_chain_modes.py.


template <int NK>
__global__ __launch_bounds__(512) void enc_chain_fwd_kernel(EncChainArgs a) {
    using LG = ChainLds<3, false, 4>;
    __shared__ __attribute__((aligned(16))) float lds[LG::FLOATS];
    constexpr int PF = NK > 0 ? CHAIN_PF : 1;
    const int B = a.B, Hp = a.Hp, n = a.n;
    const int tiles_x = Hp / CBH, tiles_y = (B + CBM - 1) / CBM;
    int m0, n0;
    tile_origin<CBM, CBH>(blockIdx.x, tiles_x, tiles_x * tiles_y, m0, n0);
    const int rt = m0 / CBM;
    ChainBatch cb;
    ChainParams& p = cb.d[0];
    for (int step = 0; step < n; ++step) {
        // ---- GRU level of both directions
        for (int d = 0; d < a.ndirs; ++d) {
            const EncChainDir& e = a.d[d];
            unsigned* cnt = a.sync + ((size_t)d * tiles_y + rt) * 32;
            const int v = d == 0 ? step : n - 1 - step;
            const long r0 = (long)v * B;
            if (step == 0) {
                // no predecessors: h_prev = 0, no product (gru_fwd_first_kernel's arithmetic) for this tile
                const int row = m0 + (threadIdx.x >> 3), col = n0 + (threadIdx.x & 7) * 4;
                if (row < B) {
                    const float* gi = e.gi + (long)a.types[r0 + row] * 3 * Hp;
                    const float4 gir = cld4(gi + col), giz = cld4(gi + Hp + col), gin = cld4(gi + 2 * Hp + col);
                    const float4 bn = cld4(e.bhn + col);
                    float4 R, Z, N, Hn;
#define GRU0(f)                                          \
    {                                                    \
        float gn = 0.f + bn.f;                           \
        float rr = 1.f / (1.f + expf(-(gir.f + 0.f)));   \
        float zz = 1.f / (1.f + expf(-(giz.f + 0.f)));   \
        float nn = tanhf(gin.f + rr * gn);               \
        R.f = rr; Z.f = zz; N.f = nn;                    \
        Hn.f = (1.f - zz) * nn + zz * 0.f;               \
    }
                    GRU0(x) GRU0(y) GRU0(z) GRU0(w)
#undef GRU0
                    float* g = e.gates + (r0 + row) * 3 * Hp;
                    cst4(g + col, R);
                    cst4(g + Hp + col, Z);
                    cst4(g + 2 * Hp + col, N);
                    cst4(e.ghn + (r0 + row) * Hp + col, bn);
                    cst4(e.agg + (r0 + row) * Hp + col, make_float4(0.f, 0.f, 0.f, 0.f));
                    st4_sc1(rsrc_of(e.hs + r0 * Hp), (unsigned)((row * Hp + col) * 4), Hn);
                }
            } else {
                if (!chain_wait(cnt, (unsigned)(tiles_x * (2 * step)), a.status)) return;
                p = ChainParams{};
                p.A = e.agg + r0 * Hp; p.lda = Hp; p.W = e.whh; p.ldw = Hp; p.M = B; p.Hp = Hp; p.K = Hp;
                p.tiles_x = tiles_x; p.tiles_y = tiles_y;
                p.gi_tab = e.gi; p.types = a.types + r0; p.bhn = e.bhn; p.hprev = p.A; p.ld_hprev = Hp;
                p.gates = e.gates + r0 * 3 * Hp; p.ghn = e.ghn + r0 * Hp; p.hout = e.hs + r0 * Hp; p.ld_hout = Hp;
                chain_body<3, false, 0, 4, PF, NK, 1, true>(cb, lds, blockIdx.x);
            }
            chain_signal(cnt);
        }
        if (step == n - 1) break;  // the pass's last vertex sends no message
        // ---- gate / mapper level of both directions (+ the next level's predecessor sums)
        for (int d = 0; d < a.ndirs; ++d) {
            const EncChainDir& e = a.d[d];
            unsigned* cnt = a.sync + ((size_t)d * tiles_y + rt) * 32;
            const int v = d == 0 ? step : n - 1 - step;
            const long r0 = (long)v * B, r1 = (long)(d == 0 ? v + 1 : v - 1) * B;
            if (!chain_wait(cnt, (unsigned)(tiles_x * (2 * step + 1)), a.status)) return;
            p = ChainParams{};
            p.A = e.hs + r0 * Hp; p.lda = Hp; p.W = e.wgm; p.ldw = Hp; p.M = B; p.Hp = Hp; p.K = Hp;
            p.tiles_x = tiles_x; p.tiles_y = tiles_y;
            p.gv = e.gv + (size_t)v * 2 * Hp; p.gm = e.gm + r0 * 2 * Hp; p.msg = e.msg; p.ld_msg = Hp; p.cur_row0 = (int)r0;
            p.next_agg = e.agg + r1 * Hp; p.ld_agg = Hp; p.next_off = e.off + r1; p.idx = e.idx;
            chain_body<2, false, 1, 4, PF, NK, 1, true>(cb, lds, blockIdx.x);
            chain_signal(cnt);
        }
    }
}

int enc_chain_fwd(const EncChainArgs& a, hipStream_t s) {
    const int tiles = (a.Hp / CBH) * cdiv(a.B, CBM);
    DVAE_REQUIRE(a.ndirs >= 1 && a.ndirs <= 2 && a.Hp % 32 == 0 && tiles >= 1 && tiles <= 256,
                 "persistent chain: needs Hp %% 32 == 0 and at most 256 tiles (one workgroup per CU); got Hp %d, %d tiles", a.Hp, tiles);
    DVAE_REQUIRE((long)a.n * a.B * a.Hp * 4 < 4294967000L, "persistent chain: a state array exceeds 4 GiB");
    DVAE_REQUIRE(a.status == a.sync + ENC_CHAIN_SYNC_WORDS - 64, "persistent chain: the status word follows the counters");
    DVAE_HIP(hipMemsetAsync(a.sync, 0, sizeof(unsigned) * ENC_CHAIN_SYNC_WORDS, s));  // counters + status
    const dim3 grid(tiles), block(512);
    if (a.Hp == 512) hipLaunchKernelGGL((enc_chain_fwd_kernel<8>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((enc_chain_fwd_kernel<0>), grid, block, 0, s, a);
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}

// ---------------------------------------------------------------------------------- bf16 levels
// The same four level operations with bf16 matrix-unit inputs (dvae_model_desc.mfma_bf16) for LARGE levels (cfg 5:
// 2048 rows, Hp = 1024), where the unfused form -- a product that writes the [rows x 3Hp] pre-activations, then a
// pointwise kernel that reads them back -- is two launches of 25-55 us each per operation and the 64-level chains were a
// third of the step.  v_mfma_f32_32x32x16_bf16, fp32 accumulation, the fp32 tails above unchanged.
//   workgroup: 128 rows x 64 hidden units x all NGATE gates, 8 waves = (K half) x (row half) x (hidden half): a wave
//              contracts two of the four 16-k groups of every 64-k step for two 32-row blocks x NGATE gates (GRU: 6
//              accumulators, 5 fragment reads per 6 MFMAs), two waves per SIMD; the K halves meet through LDS in the
//              epilogue (chain_epilogue, KS = 2), which also halves the tails a wave runs one after the other;
//   operands:  A fp32 [rows][K] (states / gradients), rounded to bf16 on the way from registers to LDS, exactly as the
//              bf16 GEMMs round it; W the K-contiguous bf16 copy of the weight (pack.hip: straight for the forward
//              operations, transposed for the backward ones), so all four are products of two K-contiguous operands;
//   pipeline:  two LDS stages of [128 + 64 NGATE][64 + 8] bf16 and TWO register-staged tiles in flight (fixed register
//              sets, the loop unrolled by two: no rotation moves at the back-edge); scheduling barriers keep the
//              fp32 -> bf16 conversion of a staged tile where it is written, or hipcc hoists it -- and the wait for
//              the loads with it -- to the top of the step (measured: 1.3 us per K step, every load latency exposed);
//   tile order: gemm_common.h's XCD-aware order (an XCD's 32 tiles = 8 row tiles x 4 hidden tiles share its L2).
typedef __bf16 cbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 cbf16x4 __attribute__((ext_vector_type(4)));
//   RB (round 3): 32-row blocks per wave.  RB = 2 is the tile above (128 rows, 92-98 KB of LDS: ONE workgroup per CU, whose
//              tail -- 12-34 us of loads / stores with the matrix unit idle -- nothing overlaps).  RB = 1 is a 64-row tile:
//              74 KB of LDS and at most 128 registers, so TWO workgroups share a CU and one's tail runs under the other's
//              K loop (twice the workgroups, each re-reading the weight panel: more L2 -> LDS bytes per flop).
constexpr int HBH = 64, HBK = 64, HLD = HBK + 8;

template <int NGATE, int RB>
struct ChainBf16Lds {
    static constexpr int ROWS = 64 * RB;
    static constexpr int A_ELEMS = ROWS * HLD, STAGE = A_ELEMS + NGATE * HBH * HLD;  // bf16 elements
    static constexpr int XCH_BYTES = 2 * RB * 8 * NGATE * 64 * 16;  // 2 RB (hidden half, row block) regions of chain_epilogue<KS = 2>
    static constexpr int BYTES = 2 * STAGE * 2 > XCH_BYTES ? 2 * STAGE * 2 : XCH_BYTES;
};

template <int NGATE, int TAIL, int RB>
__global__ __launch_bounds__(512, RB == 1 ? 2 : 1) void chain_bf16_kernel(ChainParams p, const uint16_t* __restrict__ Wh, int ldwh) {
    using LD = ChainBf16Lds<NGATE, RB>;
    constexpr int A_ELEMS = LD::A_ELEMS, STAGE = LD::STAGE, HBM_ROWS = LD::ROWS;
    constexpr int AU = HBM_ROWS * HBK / 4 / 512;    // float4 units of A per thread (2 RB)
    constexpr int BU = NGATE * HBH * HBK / 8 / 512;  // 16-byte units of W per thread (one per gate)
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LD::BYTES];
    __bf16* lds = reinterpret_cast<__bf16*>(lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kq = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1, h = lane >> 5;
    int m0, n0;
    tile_origin<HBM_ROWS, HBH>(blockIdx.x, p.tiles_x, p.tiles_x * p.tiles_y, m0, n0);

    f32x16 acc[RB][NGATE];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int i = 0; i < NGATE; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rb][i][r] = 0.f;

    // staging: thread -> (row, 4 k) of A, (row, 8 k) of W; rows past M are clamped (computed, never stored)
    const float* arow[AU];
    const uint16_t* wrow[BU];
    const int ac4 = tid & 15, wc8 = tid & 7;
#pragma unroll
    for (int i = 0; i < AU; ++i) {
        int r = m0 + ((tid + 512 * i) >> 4);
        r = r < p.M ? r : p.M - 1;
        arow[i] = p.A + (long)r * p.lda + ac4 * 4;
    }
#pragma unroll
    for (int i = 0; i < BU; ++i) {
        const int r = (tid + 512 * i) >> 3;  // 0 .. 64 NGATE - 1: gate r / 64, hidden unit n0 + r % 64
        wrow[i] = Wh + (long)((r / HBH) * p.Hp + n0 + r % HBH) * ldwh + wc8 * 8;
    }
    float4 ra[2][AU];
    u32x4 rw[2][BU];  // (a native vector: hipcc keeps HIP's uint4 struct array in memory -- promoted to LDS -- and waits for every load at once)
    const int nk = (CHAIN_ABL & 4) ? 1 : p.K / HBK;
    auto gload = [&](auto set, int t) {
        constexpr int S = decltype(set)::value;
        const int k = (t < nk ? t : 0) * HBK;  // past the end: a valid tile, loaded and never used
#pragma unroll
        for (int i = 0; i < AU; ++i) ra[S][i] = cld4(arow[i] + k);
#pragma unroll
        for (int i = 0; i < BU; ++i) rw[S][i] = *reinterpret_cast<const u32x4*>(wrow[i] + k);
    };
    auto lstore = [&](auto set, __bf16* st) {
        constexpr int S = decltype(set)::value;
#pragma unroll
        for (int i = 0; i < AU; ++i) {
            const cbf16x4 b = {(__bf16)ra[S][i].x, (__bf16)ra[S][i].y, (__bf16)ra[S][i].z, (__bf16)ra[S][i].w};
            *reinterpret_cast<uint2*>(st + ((tid + 512 * i) >> 4) * HLD + ac4 * 4) = __builtin_bit_cast(uint2, b);
        }
#pragma unroll
        for (int i = 0; i < BU; ++i)
            *reinterpret_cast<u32x4*>(st + A_ELEMS + ((tid + 512 * i) >> 3) * HLD + wc8 * 8) = rw[S][i];
    };
    // step t contracts LDS stage t & 1; in its middle the staged tile t + 1 (register set (t + 1) & 1, requested two
    // steps ago) goes to the other stage and the set is refilled with tile t + 3
    auto step = [&](auto nset, int t) {
        constexpr int NS = decltype(nset)::value;  // (t + 1) & 1
        const __bf16* a_l = lds + (NS ^ 1) * STAGE;
        const __bf16* b_l = a_l + A_ELEMS;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int kc = (2 * (2 * kq + s2) + h) * 8;  // this wave's 16-k group 2 kq + s2 of the step, lane half h
            cbf16x8 af[RB], bf[NGATE];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
                af[rb] = *reinterpret_cast<const cbf16x8*>(a_l + (wm * 32 * RB + rb * 32 + (lane & 31)) * HLD + kc);
#pragma unroll
            for (int i = 0; i < NGATE; ++i)
                bf[i] = *reinterpret_cast<const cbf16x8*>(b_l + (i * HBH + wn * 32 + (lane & 31)) * HLD + kc);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int i = 0; i < NGATE; ++i)  // operands swapped: a lane holds 4 consecutive hidden units of one row per quad
                    if (!(CHAIN_ABL & 2)) acc[rb][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[i], af[rb], acc[rb][i], 0, 0, 0);
                    else acc[rb][i][0] += (float)bf[i][0] * (float)af[rb][0];  // (diagnostic: keeps the fragment reads alive)
            if (s2 == 0) {
                __builtin_amdgcn_sched_barrier(0);
                lstore(nset, lds + NS * STAGE);
                __builtin_amdgcn_sched_barrier(0);
                gload(nset, t + 3);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    };
    gload(IC<0>{}, 0);
    lstore(IC<0>{}, lds);
    gload(IC<1>{}, 1);
    gload(IC<0>{}, 2);
    __syncthreads();
    int t = 0;
    for (; t + 1 < nk; t += 2) {
        step(IC<1>{}, t);
        step(IC<0>{}, t + 1);
    }
    if (t < nk) step(IC<1>{}, t);
    // the K halves meet in chain_epilogue: one exchange region per (hidden half, row block)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        float* region = reinterpret_cast<float*>(lds_raw) + (wn * RB + rb) * (8 * NGATE * 64 * 4);
        const int mb = m0 + wm * 32 * (RB - 1) + rb * 32;  // chain_epilogue adds wm * 32: rows m0 + wm * 32 RB + rb * 32 ..
        if (kq == 0) chain_epilogue<NGATE, TAIL, 2, 0, false>(p, acc[rb], region, mb, n0 + wn * 32, wm, lane);
        else chain_epilogue<NGATE, TAIL, 2, 1, false>(p, acc[rb], region, mb, n0 + wn * 32, wm, lane);
    }
}

// Levels the bf16 kernels may serve inside the train step: whole 64-wide hidden tiles and 64-deep K steps, and enough
// 128 x 64 tiles to fill three quarters of the chip (cfg 5: 16 x 16 = 256 tiles).  OFF by default (DVAE_CHAIN_BF16=1
// turns them on, =2 also lifts the size rule); a descriptor that carries w_bf16 runs them regardless.
// MEASURED (cfg 5 level, 2048 rows, Hp 1024, tools/chain_bf16_bench.py; us alone, product + pointwise launches -> this
// kernel): GRU forward 45.2 -> 41.3, gate forward 34.9 -> 32.1, dgrad + GRU backward 38.7 -> 52.8, dgrad accumulate +
// gather tail 31.3 + ~35 -> 40.8.  Ablation (tools/chain_ablate.sh): the K loops are 19 / 15 / 19 / 28 us, the tails 22 /
// 17 / 34 / 12 us -- a level's outputs (75-100 MB: gates, W_hn h, states, gradients of all of them) move at 3-3.4 TB/s
// while the CU's matrix unit idles, one 512-thread workgroup (98 KB of LDS) per CU.  In the step the two encoder
// directions run on two streams, and the unfused pair overlaps one direction's pointwise kernel with the other's
// product; the fused kernels cannot share a CU: cfg 5 bf16 step 101.1 ms unfused, 109.1 ms with these kernels, 113.1 ms
// with both directions of a level on one stream.  Kept for the numbers and for callers with one chain; not the default.
bool chain_bf16_ok(int rows, int Hp) {
    static const int knob = [] { const char* v = getenv("DVAE_CHAIN_BF16"); return v ? atoi(v) : 0; }();
    if (!knob || Hp % HBH != 0 || rows < 1) return false;
    return knob == 2 || (long)cdiv(rows, 128) * (Hp / HBH) >= 192;
}

// The packed weights of the bf16-mode call in progress on this thread (set by the orchestration's entry points, orch.h:
// set_mode): dvae_chain_level_batch looks a level's fp32 weight up in it and runs the bf16 kernel on the copy.
static thread_local const float* tl_chain_pw = nullptr;
static thread_local PackedLayout tl_chain_layout;
void chain_bf16_context(const float* pw, const PackedLayout* L) {
    tl_chain_pw = pw;
    if (pw && L) tl_chain_layout = *L;
}
// -> the K-contiguous bf16 copy of weight `w` (forward: as stored, pitch Hp; backward: the transposed copy, pitch =
// the matrix' rows), or NULL when the context does not hold it
static const uint16_t* chain_bf16_lookup(const float* w, bool backward, int Hp, int K, int ngate, int* ldw) {
    const PackedLayout& L = tl_chain_layout;
    if (!tl_chain_pw || w < tl_chain_pw || w >= tl_chain_pw + L.total) return nullptr;
    const size_t off = (size_t)(w - tl_chain_pw);
    if (!backward) {
        if (off % 8) return nullptr;
        *ldw = Hp;
        return reinterpret_cast<const uint16_t*>(tl_chain_pw + L.half) + off;
    }
    for (int i = 0; i < L.nmats; ++i) {
        const PackedLayout::Mat& m = L.mats[i];
        if (off != m.off) continue;  // the backward operations contract a whole matrix [K][Hp]
        if (m.rows != K || m.cols != Hp || m.toff % 8) return nullptr;
        *ldw = m.rows;
        return reinterpret_cast<const uint16_t*>(tl_chain_pw + L.halfT) + m.toff;
    }
    return nullptr;
}

template <int NGATE, int TAIL>
static int chain_bf16_launch(ChainParams p, const uint16_t* wh, int ldwh, hipStream_t s) {
    DVAE_REQUIRE(p.Hp % HBH == 0 && p.K % HBK == 0 && p.K >= HBK, "bf16 chain level: Hp and K must be multiples of 64 (Hp=%d K=%d)", p.Hp, p.K);
    DVAE_REQUIRE(wh && ((uintptr_t)wh % 16) == 0 && ldwh % 8 == 0 && ldwh >= p.K, "bf16 chain level: the bf16 weight needs 16-byte aligned rows of at least K elements");
    // DVAE_CHAIN_BF16_ROWS = 64 (default: two workgroups per CU) or 128 (round 2's tile, one per CU)
    static const int knob_rows = [] { const char* v = getenv("DVAE_CHAIN_BF16_ROWS"); return v ? atoi(v) : 64; }();
    const int rows = knob_rows == 128 ? 128 : 64;
    p.tiles_x = p.Hp / HBH;
    p.tiles_y = cdiv(p.M, rows);
    static const int knob_log = [] { const char* v = getenv("DVAE_GEMM_LOG"); return v ? atoi(v) : 0; }();
    if (knob_log) fprintf(stderr, "CHAINSEL chain_bf16_kernel<%d, %d, %d rows> %d %d %d\n", NGATE, TAIL, rows, p.M, NGATE * p.Hp, p.K);
    const dim3 grid((unsigned)(p.tiles_x * p.tiles_y));
    if (rows == 128) hipLaunchKernelGGL((chain_bf16_kernel<NGATE, TAIL, 2>), grid, dim3(512), 0, s, p, wh, ldwh);
    else hipLaunchKernelGGL((chain_bf16_kernel<NGATE, TAIL, 1>), grid, dim3(512), 0, s, p, wh, ldwh);
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}

// ---------------------------------------------------------------------------------- host
template <int NGATE, bool WKM, int TAIL, int KS, int ND>
static int chain_launch_ks(ChainBatch& b, hipStream_t s) {
    const ChainParams& p = b.d[0];
    const dim3 grid((unsigned)(p.tiles_x * p.tiles_y)), block(128 * KS);
    constexpr int BK = 16 * KS;
#define CHAIN_CASE(NK_) \
    case NK_: hipLaunchKernelGGL((chain_kernel<NGATE, WKM, TAIL, KS, CHAIN_PF, NK_, ND>), grid, block, 0, s, b); break;
    // The unrolled K loop is built for the 8-wave shape at K = Hp, 2Hp, 3Hp of Hp = 512 (hs 501: 8, 16
    // and 24 steps); deeper unrolls make hipcc hoist loads until the registers spill, and the
    // 4-wave shape needs its registers for co-residency: those run the runtime loop.
    int nk = 0;
    if constexpr (KS == 4) nk = p.K % BK == 0 ? p.K / BK : 0;
    if constexpr (KS == 4 && !WKM) {
        switch (nk) {
            CHAIN_CASE(8)
        default: nk = 0; break;
        }
    } else if constexpr (KS == 4) {
        switch (nk) {
            CHAIN_CASE(8)
            CHAIN_CASE(16)
            CHAIN_CASE(24)
        default: nk = 0; break;
        }
    }
    if (nk == 0) hipLaunchKernelGGL((chain_kernel<NGATE, WKM, TAIL, KS, 1, 0, ND>), grid, block, 0, s, b);
#undef CHAIN_CASE
    static const int knob_log = [] { const char* v = getenv("DVAE_GEMM_LOG"); return v ? atoi(v) : 0; }();
    if (knob_log)  // one line per launch, as gemm.hip's (tools/roofline_by_kernel.py)
        fprintf(stderr, "CHAINSEL chain_kernel<%d, %s, %d, %d, %d, %d, %d> %d %d %d\n", NGATE, WKM ? "true" : "false", TAIL, KS,
                nk ? CHAIN_PF : 1, nk, ND, p.M * ND, NGATE * p.Hp, p.K);
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}

template <int NGATE, bool WKM, int TAIL>
static int chain_launch(ChainBatch& b, int count, hipStream_t s) {
    for (int i = 0; i < count; ++i) {
        b.d[i].tiles_x = b.d[i].Hp / CBH;
        b.d[i].tiles_y = cdiv(b.d[i].M, CBM);
    }
    if (count == 1 && b.d[0].M <= chain_tiny_rows()) {
        ChainParams& p = b.d[0];
        p.tiles_x = p.Hp / 16;
        p.tiles_y = cdiv(p.M, 16);
        constexpr int KS = WKM ? 8 : 4;
        hipLaunchKernelGGL((chain_tiny_kernel<NGATE, WKM, TAIL, KS>), dim3((unsigned)(p.tiles_x * p.tiles_y)), dim3(64 * KS), 0, s, p);
        DVAE_LAUNCH_CHECK();
        return DVAE_OK;
    }
    if (count == 1 && b.d[0].M <= chain_small_rows()) {
        ChainParams& p = b.d[0];
        p.tiles_y = cdiv(p.M, 32);
        hipLaunchKernelGGL((chain_small_kernel<NGATE, WKM, TAIL>), dim3((unsigned)(p.tiles_x * p.tiles_y)), dim3(256), 0, s, p);
        DVAE_LAUNCH_CHECK();
        return DVAE_OK;
    }
    // DVAE_CHAIN_KS = 2 / 4 forces the workgroup shape (A/B runs)
    static const int knob = [] { const char* v = getenv("DVAE_CHAIN_KS"); return v ? atoi(v) : 0; }();
    const int ks = knob == 2 || knob == 4 ? knob : 4;
    if (count == 2)
        return ks == 2 ? chain_launch_ks<NGATE, WKM, TAIL, 2, 2>(b, s) : chain_launch_ks<NGATE, WKM, TAIL, 4, 2>(b, s);
    return ks == 2 ? chain_launch_ks<NGATE, WKM, TAIL, 2, 1>(b, s) : chain_launch_ks<NGATE, WKM, TAIL, 4, 1>(b, s);
}

static inline bool al16(const void* q) { return ((uintptr_t)q % 16) == 0; }
static inline bool ldok(int ld, int w) { return ld >= w && ld % 4 == 0; }

// descriptor -> kernel parameters, with the checks of the operation
static int chain_fill(int op, const dvae_level_desc& v, int rows, int Hp, ChainParams& p) {
    p = ChainParams{};
    const int K = (op == DVAE_LEVEL_GRU_FWD || op == DVAE_LEVEL_GATED_FWD) ? Hp : v.K;
    DVAE_REQUIRE(K >= 32 && K % 32 == 0, "chain level: K must be a positive multiple of 32 (K=%d)", K);
    DVAE_REQUIRE(v.a && v.w && al16(v.a) && al16(v.w), "chain level: operands must be non-null and 16-byte aligned");
    DVAE_REQUIRE(ldok(v.lda, K), "chain level: bad leading dimension %d of the operand", v.lda);
    p.A = v.a; p.lda = v.lda; p.W = v.w; p.ldw = Hp; p.M = rows; p.Hp = Hp; p.K = K;
    switch (op) {
    case DVAE_LEVEL_GRU_FWD:
        DVAE_REQUIRE(v.gi_tab && v.types && v.bhn && v.gates && v.ghn && v.hout, "gru level fwd: null pointer");
        DVAE_REQUIRE(ldok(v.ld_hout, Hp), "gru level fwd: bad ld_hout %d", v.ld_hout);
        DVAE_REQUIRE(al16(v.gi_tab) && al16(v.bhn) && al16(v.gates) && al16(v.ghn) && al16(v.hout),
                     "gru level fwd: buffers must be 16-byte aligned");
        p.gi_tab = v.gi_tab; p.types = v.types; p.bhn = v.bhn; p.hprev = v.a; p.ld_hprev = v.lda;
        p.gates = v.gates; p.ghn = v.ghn; p.hout = v.hout; p.ld_hout = v.ld_hout;
        break;
    case DVAE_LEVEL_GATED_FWD:
        DVAE_REQUIRE(v.gv_row && v.gm && v.msg, "gated level fwd: null pointer");
        DVAE_REQUIRE(!v.next_agg || (v.next_off && v.idx), "gated level fwd: next_agg needs the CSR lists");
        DVAE_REQUIRE(ldok(v.ld_msg, Hp) && (!v.next_agg || ldok(v.ld_agg, Hp)), "gated level fwd: bad leading dimension");
        DVAE_REQUIRE(al16(v.gv_row) && al16(v.gm) && al16(v.msg) && al16(v.next_agg),
                     "gated level fwd: buffers must be 16-byte aligned");
        p.gv = v.gv_row; p.gm = v.gm; p.msg = v.msg; p.ld_msg = v.ld_msg; p.cur_row0 = v.cur_row0;
        p.next_agg = v.next_agg; p.ld_agg = v.ld_agg; p.next_off = v.next_off; p.idx = v.idx;
        break;
    case DVAE_LEVEL_GRU_BWD:
        DVAE_REQUIRE(v.gates && v.ghn && v.hprev && v.dgh && v.dgin && v.dhprev, "gru level bwd: null pointer");
        DVAE_REQUIRE(ldok(v.ld_hprev, Hp) && ldok(v.ld_dhprev, Hp), "gru level bwd: bad leading dimension");
        DVAE_REQUIRE(al16(v.gates) && al16(v.ghn) && al16(v.hprev) && al16(v.dgh) && al16(v.dgin) && al16(v.dhprev),
                     "gru level bwd: buffers must be 16-byte aligned");
        p.gates = v.gates; p.ghn = v.ghn; p.hprev = v.hprev; p.ld_hprev = v.ld_hprev;
        p.dgh = v.dgh; p.dgin = v.dgin; p.dhprev = v.dhprev; p.ld_dhprev = v.ld_dhprev;
        // fall through: dx is checked below
    case DVAE_LEVEL_DGRAD_ACC:
        DVAE_REQUIRE(v.dx && ldok(v.ld_dx, Hp) && al16(v.dx), "level dgrad: bad dx / ld_dx");
        p.C = v.dx; p.ldc = v.ld_dx;
        if (op == DVAE_LEVEL_DGRAD_ACC && v.dgm_next) {
            DVAE_REQUIRE(v.gather_src && v.next_off && v.idx && v.gm, "level dgrad: the fused gather needs gather_src, next_off, idx, gm");
            DVAE_REQUIRE(!v.off2 || (v.gather_src2 && v.idx2), "level dgrad: second gather list incomplete");
            DVAE_REQUIRE(al16(v.gather_src) && al16(v.gather_src2) && al16(v.gm) && al16(v.dgm_next),
                         "level dgrad: gather buffers must be 16-byte aligned");
            p.gsrc = v.gather_src; p.goff = v.next_off; p.gidx = v.idx; p.cur_row0 = v.cur_row0;
            p.gsrc2 = v.gather_src2; p.goff2 = v.off2; p.gidx2 = v.idx2;
            p.gm_in = v.gm; p.dgm_out = v.dgm_next;
        }
        break;
    default:
        return set_err(DVAE_ERR_ARG, "chain level: unknown operation %d", op);
    }
    return DVAE_OK;
}

}  // namespace dvae

using namespace dvae;

extern "C" int dvae_chain_level_batch(int32_t op, const dvae_level_desc* levels, int32_t count, int32_t rows,
                                      int32_t Hp, void* stream) {
    DVAE_REQUIRE(levels && count >= 1 && count <= 2, "chain level batch: 1 or 2 levels per launch (got %d)", count);
    DVAE_REQUIRE(rows >= 0 && Hp > 0 && Hp % 32 == 0, "chain level: Hp must be a positive multiple of 32 (Hp=%d)", Hp);
    if (rows == 0) return DVAE_OK;
    ChainBatch b;
    for (int i = 0; i < count; ++i) DVAE_TRY(chain_fill(op, levels[i], rows, Hp, b.d[i]));
    if (count == 1) b.d[1] = b.d[0];
    DVAE_REQUIRE(b.d[1].K == b.d[0].K, "chain level batch: the levels of a launch share K");
    hipStream_t s = (hipStream_t)stream;
    {   // bf16 matrix-unit inputs: an explicit bf16 weight in the descriptor, or the bf16-mode call in progress holds one
        const bool backward = op == DVAE_LEVEL_DGRAD_ACC || op == DVAE_LEVEL_GRU_BWD;
        const uint16_t* wh[2] = {nullptr, nullptr};
        int ldwh[2] = {0, 0};
        bool all = true, any_explicit = false;
        for (int i = 0; i < count; ++i) {
            if (levels[i].w_bf16) {
                wh[i] = levels[i].w_bf16; ldwh[i] = levels[i].ldw_bf16; any_explicit = true;
            } else if (chain_bf16_ok(rows, Hp) && b.d[i].K % HBK == 0) {
                wh[i] = chain_bf16_lookup(levels[i].w, backward, Hp, b.d[i].K, 0, &ldwh[i]);
            }
            all = all && wh[i] != nullptr;
        }
        DVAE_REQUIRE(all || !any_explicit, "chain level batch: w_bf16 must be given for every level of the launch or none");
        if (all) {
            for (int i = 0; i < count; ++i) {
                switch (op) {
                case DVAE_LEVEL_GRU_FWD: DVAE_TRY((chain_bf16_launch<3, 0>(b.d[i], wh[i], ldwh[i], s))); break;
                case DVAE_LEVEL_GATED_FWD: DVAE_TRY((chain_bf16_launch<2, 1>(b.d[i], wh[i], ldwh[i], s))); break;
                case DVAE_LEVEL_DGRAD_ACC: DVAE_TRY((chain_bf16_launch<1, 2>(b.d[i], wh[i], ldwh[i], s))); break;
                default: DVAE_TRY((chain_bf16_launch<1, 3>(b.d[i], wh[i], ldwh[i], s))); break;
                }
            }
            return DVAE_OK;
        }
    }
    switch (op) {
    case DVAE_LEVEL_GRU_FWD: return chain_launch<3, false, 0>(b, count, s);
    case DVAE_LEVEL_GATED_FWD: return chain_launch<2, false, 1>(b, count, s);
    case DVAE_LEVEL_DGRAD_ACC: return chain_launch<1, true, 2>(b, count, s);
    default: return chain_launch<1, true, 3>(b, count, s);
    }
}

extern "C" int dvae_gru_level_fwd(const float* hprev, int32_t ld_hprev, const float* whh, const float* gi_tab,
                                  const int32_t* types, const float* bhn, float* gates, float* ghn_out, float* hout,
                                  int32_t ld_hout, int32_t rows, int32_t Hp, void* stream) {
    dvae_level_desc v = {};
    v.a = hprev; v.lda = ld_hprev; v.w = whh; v.gi_tab = gi_tab; v.types = types; v.bhn = bhn;
    v.gates = gates; v.ghn = ghn_out; v.hout = hout; v.ld_hout = ld_hout;
    return dvae_chain_level_batch(DVAE_LEVEL_GRU_FWD, &v, 1, rows, Hp, stream);
}

extern "C" int dvae_gated_level_fwd(const float* h, int32_t ld_h, const float* wgm, const float* gv_row, float* gm,
                                    float* msg, int32_t ld_msg, int32_t cur_row0, float* next_agg, int32_t ld_agg,
                                    const int32_t* next_off, const int32_t* idx, int32_t rows, int32_t Hp,
                                    void* stream) {
    dvae_level_desc v = {};
    v.a = h; v.lda = ld_h; v.w = wgm; v.gv_row = gv_row; v.gm = gm; v.msg = msg; v.ld_msg = ld_msg;
    v.cur_row0 = cur_row0; v.next_agg = next_agg; v.ld_agg = ld_agg; v.next_off = next_off; v.idx = idx;
    return dvae_chain_level_batch(DVAE_LEVEL_GATED_FWD, &v, 1, rows, Hp, stream);
}

extern "C" int dvae_level_dgrad_acc(const float* dy, int32_t ld_dy, int32_t K, const float* w, float* dx, int32_t ld_dx,
                                    int32_t rows, int32_t Hp, void* stream) {
    dvae_level_desc v = {};
    v.a = dy; v.lda = ld_dy; v.K = K; v.w = w; v.dx = dx; v.ld_dx = ld_dx;
    return dvae_chain_level_batch(DVAE_LEVEL_DGRAD_ACC, &v, 1, rows, Hp, stream);
}

extern "C" int dvae_gru_level_bwd(const float* dy, int32_t ld_dy, int32_t K, const float* w, float* dh, int32_t ld_dh,
                                  const float* gates, const float* ghn, const float* hprev, int32_t ld_hprev,
                                  float* dgh, float* dgin, float* dhprev, int32_t ld_dhprev, int32_t rows, int32_t Hp,
                                  void* stream) {
    dvae_level_desc v = {};
    v.a = dy; v.lda = ld_dy; v.K = K; v.w = w; v.dx = dh; v.ld_dx = ld_dh;
    v.gates = const_cast<float*>(gates); v.ghn = const_cast<float*>(ghn); v.hprev = hprev; v.ld_hprev = ld_hprev;
    v.dgh = dgh; v.dgin = dgin; v.dhprev = dhprev; v.ld_dhprev = ld_dhprev;
    return dvae_chain_level_batch(DVAE_LEVEL_GRU_BWD, &v, 1, rows, Hp, stream);
}
