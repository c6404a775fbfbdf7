// bf16 MFMA products of two K-contiguous operands (the forward product of a Linear layer, C = A W^T, and a data
// gradient against a transposed weight copy), with the operands in HBM as bf16 OR fp32, each side independently.
//
// Why its own kernel: on gfx950 a 32x32x16 bf16 MFMA retires 16x the flops of the fp32 one, so the product is no longer
// bound by the matrix unit but by the bytes a workgroup pulls through L2 per flop.  gemm_bf16.hip's 128 x 128 tile fed
// from fp32 memory has 32 flop per byte and saturates at ~0.44 PFLOP/s (profiles/r01_gemm_shapes_bf16.txt: 418-445 TF
// at 4096^3 and 8192^3, the same on every large shape).  Two changes raise the ratio 4x:
//   * operands stored as bf16 (weights: a bf16 copy made once per optimizer step beside the packed fp32 master copy;
//     activations: written as bf16 by the kernel that produces them) -- half the bytes, no conversion in the loop;
//   * 256 x 256 x 64 tiles, 8 waves (2 x 4, 128 x 64 per wave: 8 accumulator blocks), 147 KB of LDS, one workgroup per CU.
// 128 x 128 and 64 x 64 / 4-wave instances serve products too small for 256 CUs' worth of 256 x 256 tiles.
//
// LDS image of either operand: [rows][64 + 8] bf16 (as gemm_bf16.hip): a lane's MFMA fragment is one ds_read_b128, the
// 144-byte row stride spreads a 16-lane group over all banks.  Register-staged double buffering: the next tile is
// written to the other stage in the middle of a step and the loads of the tile after it are issued right behind.
#include "gemm_common.h"

namespace dvae {

typedef __bf16 nbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 nbf16x4 __attribute__((ext_vector_type(4)));

constexpr int NBK = 64;          // K step
constexpr int NLD = NBK + 8;     // LDS row stride, bf16 elements

__device__ uint4 nn_zero16;  // zero-initialised: what a gathered load past the end of K reads

// One operand tile: ROWS rows x 64 k, loaded by NT threads, 16 bytes per thread and pass.
//   GATHER = false: buffer loads off a descriptor based at the tile's first row whose range ends with the matrix: one
//     32-bit offset register per operand, and rows past the last one or k past the end of a ragged last step (the
//     offset is then replaced by one out of range) read zeros -- no clamping, no select on the data, nothing conditional.
//   GATHER = true (a row index list): per-row 64-bit addresses; past the end of K the ADDRESS is redirected to 16 bytes of
//     zeros.  (A select on the loaded data makes hipcc branch around the LDS stores and drain the loads one by one.)
template <int ROWS, int NT, bool HALF, bool GATHER>
struct NnLoader {
    static constexpr int ES = HALF ? 2 : 4;         // bytes per element
    static constexpr int EPT = 16 / ES;             // elements per 16-byte unit
    static constexpr int TPR = NBK / EPT;           // threads per row: 8 / 16
    static constexpr int RPP = NT / TPR;            // rows per pass
    static constexpr int PASSES = ROWS / RPP;
    static constexpr int LDS_ELEMS = ROWS * NLD;
    static_assert(ROWS % RPP == 0, "tile rows must be a multiple of the rows per pass");
    uint4 r[PASSES];
    int ridx[GATHER ? PASSES : 1];  // GATHER: the (clamped) rows
    const char* base_;
    int ld_;
    __amdgpu_buffer_rsrc_t rsrc_;
    unsigned voff_, pass_;          // !GATHER: byte offset of this thread's unit in pass 0; bytes from pass to pass

    __device__ __forceinline__ void init(const void* base, const int* rows, int ld, int r0, int rdim, int kdim, int tid) {
        ld_ = ld;
        if (GATHER) {
            base_ = reinterpret_cast<const char*>(base);
#pragma unroll
            for (int i = 0; i < PASSES; ++i) {
                int rr = r0 + tid / TPR + RPP * i;
                rr = rr < rdim ? rr : rdim - 1;  // clamped rows are computed but never stored
                ridx[i] = rows[rr];
            }
        } else {
            const char* b0 = reinterpret_cast<const char*>(base) + (long)r0 * ld * ES;
            const long bytes = ((long)(rdim - r0 - 1) * ld + kdim) * ES;  // to the end of the last row's K elements
            rsrc_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(b0), 0, bytes < 0xFFFFFFF0L ? (unsigned)bytes : OOB, 0x00020000);
            voff_ = (unsigned)((tid / TPR) * ld * ES + (tid % TPR) * 16);
            pass_ = (unsigned)(RPP * ld * ES);
        }
    }
    __device__ __forceinline__ void load(int k0, int kend, int tid) {
        const int kk = k0 + (tid % TPR) * EPT;
        const bool ok = kk < kend;
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            if (GATHER) {
                const char* src = base_ + ((long)ridx[i] * ld_ + kk) * ES;
                r[i] = *reinterpret_cast<const uint4*>(ok ? src : reinterpret_cast<const char*>(&nn_zero16));
            } else {
                const unsigned off = ok ? voff_ + (unsigned)k0 * ES + i * pass_ : OOB;
                r[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_, off, 0, 0));
            }
        }
    }
    __device__ __forceinline__ void store(__bf16* lds, int tid) {
        const int c = tid % TPR;
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const int row = tid / TPR + RPP * i;
            if (HALF) {
                *reinterpret_cast<uint4*>(lds + row * NLD + c * 8) = r[i];
            } else {
                const float4 v = __builtin_bit_cast(float4, r[i]);
                nbf16x4 b = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
                *reinterpret_cast<uint2*>(lds + row * NLD + c * 4) = __builtin_bit_cast(uint2, b);
            }
        }
    }
};

// MFMA fragment of k-step s (16 k) for the 32-row block at `roff`: lane l holds row roff + (l & 31), k = 16 s + 8 (l >> 5) + 0..7
__device__ __forceinline__ nbf16x8 nn_frag(const __bf16* lds, int roff, int s, int lane) {
    return *reinterpret_cast<const nbf16x8*>(lds + (roff + (lane & 31)) * NLD + (2 * s + (lane >> 5)) * 8);
}

template <int BM, int BN, int WM, int WN, bool AH, bool BH, bool AG>
__global__ __launch_bounds__(WM * WN * 64) void gemm_bf16_nn_kernel(GemmParams p) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr bool SWAP = true;  // transposed accumulation, 16-byte epilogue (gemm_common.h); the launcher checks its conditions
    using LA = NnLoader<BM, NT, AH, AG>;
    using LB = NnLoader<BN, NT, BH, false>;
    constexpr int STAGE = LA::LDS_ELEMS + LB::LDS_ELEMS;
    __shared__ __attribute__((aligned(16))) __bf16 lds[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    int kbeg = 0, kend = p.K;
    if (p.split_k > 1) {
        kbeg = blockIdx.z * p.k_per_split;
        kend = min(p.K, kbeg + p.k_per_split);
        if (kbeg >= kend) return;
    }
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
        la.init(p.A, p.a_rows, p.lda, m0, p.M, p.K, tid);
        lb.init(p.B, nullptr, p.ldb, n0, p.N, p.K, tid);
        auto tile = [&](int t) { int k = kbeg + t * NBK; return k < kend ? k : kbeg; };  // past the end: loaded, never used
        la.load(tile(0), kend, tid);
        lb.load(tile(0), kend, tid);
        la.store(lds, tid);
        lb.store(lds + LA::LDS_ELEMS, tid);
        la.load(tile(1), kend, tid);
        lb.load(tile(1), kend, tid);
        __syncthreads();

        int cur = 0, t = 0;
        for (int k0 = kbeg; k0 < kend; k0 += NBK, ++t) {
            const __bf16* a_l = lds + cur * STAGE;
            const __bf16* b_l = a_l + LA::LDS_ELEMS;
#pragma unroll
            for (int s = 0; s < NBK / 16; ++s) {
                nbf16x8 af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = nn_frag(a_l, (wm * TM + i) * 32, s, lane);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = nn_frag(b_l, (wn * TN + j) * 32, s, lane);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j], af[i], acc[i][j], 0, 0, 0)
                                         : __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
                if (s == 1) {
                    // mid-step: staged registers (tile t+1) -> the other stage, then issue the loads of tile t+2
                    // (the same block right behind the step's barrier instead was measured: within +-4 % on every shape)
                    la.store(lds + (cur ^ 1) * STAGE, tid);
                    lb.store(lds + (cur ^ 1) * STAGE + LA::LDS_ELEMS, tid);
                    __builtin_amdgcn_sched_barrier(0);
                    la.load(tile(t + 2), kend, tid);
                    lb.load(tile(t + 2), kend, tid);
                    __builtin_amdgcn_sched_barrier(0);  // (the loads must not sink below the MFMAs of the second half)
                }
            }
            __syncthreads();
            cur ^= 1;
        }
        if (p.c_bf16) store_acc_bf16<TM, TN>(p, acc, m0, n0, wm, wn, lane, BM, BN);
        else store_acc<TM, TN, SWAP>(p, acc, m0, n0, wm, wn, lane, BM, BN);
    }
}

template <int BM, int BN, int WM, int WN>
static int launch_nn(GemmParams p, bool ah, bool bh, int cus, hipStream_t s) {
    p.tiles_x = cdiv(p.N, BN);
    p.tiles_y = cdiv(p.M, BM);
    const long ntiles = (long)p.tiles_x * p.tiles_y;
    const long cap = BM == 256 ? cus : BM == 128 ? 2L * cus : 4L * cus;  // 147 KB of LDS: one 256 x 256 workgroup per CU; 37 / 18 KB: several
    dim3 grid((unsigned)(ntiles < cap ? ntiles : cap), 1, p.split_k > 1 ? p.split_k : 1);
    dim3 block(WM * WN * 64);
    const bool ag = p.a_rows != nullptr;
#define NN_LAUNCH(AH_, BH_)                                                                                     \
    do {                                                                                                        \
        if (ag) hipLaunchKernelGGL((gemm_bf16_nn_kernel<BM, BN, WM, WN, AH_, BH_, true>), grid, block, 0, s, p); \
        else hipLaunchKernelGGL((gemm_bf16_nn_kernel<BM, BN, WM, WN, AH_, BH_, false>), grid, block, 0, s, p);  \
    } while (0)
    if (ah && bh) NN_LAUNCH(true, true);
    else if (bh) NN_LAUNCH(false, true);
    else if (ah) NN_LAUNCH(true, false);
    else NN_LAUNCH(false, false);
#undef NN_LAUNCH
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}

// Called by gemm_bf16() for K-contiguous x K-contiguous products (after its argument checks).  Returns 1 when the
// product is not one this file serves (the caller then runs its own kernel), 0 / negative as usual otherwise.
int gemm_bf16_nn(const dvae_gemm_desc& d, hipStream_t s) {
    if (d.a_kmajor || d.b_kmajor) return 1;
    static const int knob = getenv("DVAE_GEMM_NN") ? atoi(getenv("DVAE_GEMM_NN")) : 1;  // 0: never, 64 / 128 / 256: force the tile
    const bool typed = d.a_bf16 || d.b_bf16 || d.c_bf16;
    if (d.c_bf16)
        DVAE_REQUIRE(!d.accumulate && d.act == 0 && !d.relu_mask && d.ldc % 4 == 0 && (uintptr_t)d.C % 8 == 0 && d.N % 4 == 0 &&
                         ((uintptr_t)d.bias % 16) == 0,
                     "gemm_bf16: a bf16-stored C takes no activation / accumulate / mask and needs 8-byte aligned rows, N %% 4 == 0");
    if (!knob && !typed) return 1;
    const int ea = d.a_bf16 ? 8 : 4, eb = d.b_bf16 ? 8 : 4;
    DVAE_REQUIRE(d.K % ea == 0 && d.lda % ea == 0 && (uintptr_t)d.A % 16 == 0, "gemm_bf16: A needs 16-byte aligned rows and K %% %d == 0", ea);
    DVAE_REQUIRE(d.K % eb == 0 && d.ldb % eb == 0 && (uintptr_t)d.B % 16 == 0, "gemm_bf16: B needs 16-byte aligned rows and K %% %d == 0", eb);
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    const long t256 = (long)cdiv(d.M, 256) * cdiv(d.N, 256), t128 = (long)cdiv(d.M, 128) * cdiv(d.N, 128);
    bool big = t256 >= (3L * cus) / 4;  // at least three quarters of a round of the chip
    bool mid = t128 >= (3L * cus) / 2;  // as gemm_bf16.hip: 128 x 128 tiles from one and a half rounds, 64 x 64 below
    if (knob == 256) big = true;
    if (knob == 128) { big = false; mid = true; }
    if (knob == 64) big = mid = false;
    if (!big && !typed && knob == 1) return 1;  // small fp32-stored products: gemm_bf16.hip's tuned tile / split-K choice
    GemmParams p;
    p.A = d.A; p.B = d.B; p.C = d.C;
    p.a_rows = d.a_rows; p.b_rows = d.b_rows; p.c_rows = d.c_rows;
    p.bias = d.bias; p.relu_mask = d.relu_mask; p.lda = d.lda; p.ldb = d.ldb; p.ldc = d.ldc;
    p.M = d.M; p.N = d.N; p.K = d.K; p.act = d.act; p.accumulate = d.accumulate;
    p.split_k = 1; p.k_per_split = 0; p.c_bf16 = d.c_bf16;
    if (d.b_rows || (!d.c_bf16 && !gemm_vec_epilogue_ok(p))) {  // not built here: a gathered B, an output that cannot take 16-byte stores
        if (typed) return set_err(DVAE_ERR_ARG, "gemm_bf16: bf16-stored operands need N %% 4 == 0, ldc %% 4 == 0, 16-byte aligned C / bias, no b_rows");
        return 1;
    }
    // (split-K for accumulating products was built and measured on cfg 5's level data gradients, 2048 x 1024 x 3072 /
    // 2048, fp32 x bf16: 64 x 64 tiles unsplit 32.6 / 23.8 us; split 2 / 3 / 4: 37.7 / 43.3 / 47.9 us; 128 x 128 tiles
    // 49.9 unsplit, 45.8 at best -- the atomics on an 8 MB output cost more than the shorter K loop saves.  Not kept.)
    static const int knob_log = getenv("DVAE_GEMM_LOG") ? atoi(getenv("DVAE_GEMM_LOG")) : 0;
    if (knob_log)
        fprintf(stderr, "GEMMSELN %s %d %d %d abf=%d bbf=%d ar=%d acc=%d\n", big ? "256x256" : mid ? "128x128" : "64x64", d.M, d.N, d.K,
                d.a_bf16, d.b_bf16, d.a_rows != nullptr, d.accumulate);
    if (big) return launch_nn<256, 256, 2, 4>(p, d.a_bf16 != 0, d.b_bf16 != 0, cus, s);
    if (mid) return launch_nn<128, 128, 2, 2>(p, d.a_bf16 != 0, d.b_bf16 != 0, cus, s);
    return launch_nn<64, 64, 2, 2>(p, d.a_bf16 != 0, d.b_bf16 != 0, cus, s);
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, long n4) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(src)[i];
        nbf16x4 b = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
        reinterpret_cast<uint2*>(dst)[i] = __builtin_bit_cast(uint2, b);
    }
}

}  // namespace dvae

extern "C" int dvae_f32_to_bf16(const float* src, uint16_t* dst, int64_t n, void* stream) {
    using namespace dvae;
    if (n == 0) return DVAE_OK;
    DVAE_REQUIRE(src && dst && n > 0 && n % 4 == 0, "f32_to_bf16: null pointer or a length that is not a multiple of 4");
    DVAE_REQUIRE((uintptr_t)src % 16 == 0 && (uintptr_t)dst % 8 == 0, "f32_to_bf16: misaligned buffer");
    const long n4 = n / 4;
    long blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, n4);
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}
