// Pieces shared by the fp32-MFMA and bf16-MFMA GEMM kernels: launch parameters, the XCD-aware
// tile order and the accumulator epilogue (identical C/D register maps on gfx950).
#pragma once
#include "common.h"
#include "dvae_hip.h"

namespace dvae {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr unsigned OOB = 0xFFFFFFF0u;  // buffer offset past num_records: the load returns 0

struct GemmParams {
    const float* A; const float* B; float* C;
    const int* a_rows; const int* b_rows; const int* c_rows;
    const float* bias;
    const float* relu_mask;  // nullable: C = mask > 0 ? value : 0, mask laid out like C
    int lda, ldb, ldc;
    int M, N, K;
    int act;         // 0 none, 1 relu, 2 tanh
    int accumulate;  // C += result
    int split_k;     // >1: atomicAdd into C (C must hold the running sum / zeros)
    int k_per_split; // multiple of the kernel's K step
    int tiles_x, tiles_y;  // output tiles along N and M; workgroup w walks tiles w, w + gridDim.x, ...
    int c_bf16;            // gemm_bf16_nn only: C is stored as bf16 (ldc in bf16 elements)
};

// XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (each with its own
// L2), so workgroup ids that are equal mod 8 share an L2.  Give each such group a contiguous
// run of tiles, so the tiles that re-read one A row block meet in one L2 (4 MB per XCD).
template <int BM, int BN>
__device__ __forceinline__ void tile_origin(int orig, int nx, int nwg, int& m0, int& n0) {
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    // within the run: bands of 8 tile rows walked column by column, so the ~64 tiles an XCD works on
    // at once form an 8 x 8 block (8 A panels + 8 B panels = 4 MB at K = 512) instead of 4 x 16
    // (FETCH_SIZE of the 9251 x 2048 x 512 product 80.1 -> 58.7 MB counted, same run time)
    const int ny = nwg / nx;
    const int band = id / (8 * nx), rem = id % (8 * nx);
    const int rows_in_band = ny - band * 8 < 8 ? ny - band * 8 : 8;
    m0 = (band * 8 + rem % rows_in_band) * BM;
    n0 = (rem / rows_in_band) * BN;
}

// Epilogue.  The MFMA leaves D[i][j] with j = lane & 31 and i = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
//   SWAP == false: the MFMA ran as (A, B): i is the row of C, j the column.  One instruction of
//     a wave then covers 2 rows x 128 contiguous bytes: the form split-K uses, because the only
//     fp32 atomic is one dword per lane.
//   SWAP == true: the MFMA ran as (B, A), so D is the transposed block: j (the lane) is the ROW of
//     C and a register quad holds 4 CONSECUTIVE columns: every lane loads / stores 16 bytes
//     (4x fewer memory instructions; a row's 128-byte line is completed by 4 instructions).
//     Needs N % 4 == 0, ldc % 4 == 0 and 16-byte aligned C (and bias): the launcher checks.
// All mode tests are wave-uniform and hoisted; an interior tile runs with no bounds checks.
template <int TM, int TN, bool SWAP>
__device__ __forceinline__ void store_acc(const GemmParams& p, f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn,
                                          int lane, int BM, int BN) {
    const int h = lane >> 5, cl = lane & 31;
    const bool interior = m0 + BM <= p.M && n0 + BN <= p.N;
    if (!SWAP) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + (wn * TN + j) * 32 + cl;
            const bool colok = interior || col < p.N;
            const float bv = (p.bias && colok) ? p.bias[col] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int rbase = m0 + (wm * TM + i) * 32 + 4 * h;
                if (p.split_k > 1) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = rbase + (r & 3) + 8 * (r >> 2);
                        if (interior || (colok && row < p.M)) {
                            const long crow = p.c_rows ? (long)p.c_rows[row] : (long)row;
                            atomicAdd(p.C + crow * p.ldc + col, acc[i][j][r]);
                        }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = rbase + (r & 3) + 8 * (r >> 2);
                        if (interior || (colok && row < p.M)) {
                            const long crow = p.c_rows ? (long)p.c_rows[row] : (long)row;
                            float* dst = p.C + crow * p.ldc + col;
                            float v = acc[i][j][r] + bv;
                            if (p.act == 1) v = fmaxf(v, 0.f);
                            else if (p.act == 2) v = tanhf(v);
                            if (p.relu_mask) v = p.relu_mask[crow * p.ldc + col] > 0.f ? v : 0.f;
                            if (p.accumulate) v += *dst;
                            *dst = v;
                        }
                    }
                }
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = m0 + (wm * TM + i) * 32 + cl;
            if (!(interior || row < p.M)) continue;
            const long crow = p.c_rows ? (long)p.c_rows[row] : (long)row;
            float* drow = p.C + crow * p.ldc;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int col = n0 + (wn * TN + j) * 32 + 8 * q + 4 * h;
                    if (!(interior || col < p.N)) continue;  // N % 4 == 0: a quad is inside or outside as a whole
                    float4 v = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                    if (p.bias) {
                        const float4 b = *reinterpret_cast<const float4*>(p.bias + col);
                        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
                    }
                    if (p.act == 1) {
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    } else if (p.act == 2) {
                        v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w);
                    }
                    if (p.relu_mask) {
                        const float4 m = *reinterpret_cast<const float4*>(p.relu_mask + crow * p.ldc + col);
                        v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f;
                        v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
                    }
                    float4* dst = reinterpret_cast<float4*>(drow + col);
                    if (p.accumulate) {
                        const float4 o = *dst;
                        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                    }
                    *dst = v;
                }
            }
        }
    }
}

// The SWAP epilogue for a C stored as bf16 (gemm_bf16_nn.hip, dvae_gemm_desc.c_bf16): a register quad = 4 consecutive
// columns of one row = one 8-byte store of 4 bf16 (round to nearest even); bias and row scatter as above, nothing else.
template <int TM, int TN>
__device__ __forceinline__ void store_acc_bf16(const GemmParams& p, f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn,
                                               int lane, int BM, int BN) {
    typedef __bf16 sbf16x4 __attribute__((ext_vector_type(4)));
    const int h = lane >> 5, cl = lane & 31;
    const bool interior = m0 + BM <= p.M && n0 + BN <= p.N;
    uint16_t* Ch = reinterpret_cast<uint16_t*>(p.C);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = m0 + (wm * TM + i) * 32 + cl;
        if (!(interior || row < p.M)) continue;
        const long crow = p.c_rows ? (long)p.c_rows[row] : (long)row;
        uint16_t* drow = Ch + crow * p.ldc;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = n0 + (wn * TN + j) * 32 + 8 * q + 4 * h;
                if (!(interior || col < p.N)) continue;
                float4 v = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                if (p.bias) {
                    const float4 b = *reinterpret_cast<const float4*>(p.bias + col);
                    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
                }
                const sbf16x4 o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
                *reinterpret_cast<uint2*>(drow + col) = __builtin_bit_cast(uint2, o);
            }
        }
    }
}

// can this product use the 16-byte (SWAP) epilogue?
inline bool gemm_vec_epilogue_ok(const GemmParams& p) {
    return p.split_k <= 1 && p.N % 4 == 0 && p.ldc % 4 == 0 && ((uintptr_t)p.C % 16) == 0 &&
           (p.bias == nullptr || ((uintptr_t)p.bias % 16) == 0) && ((uintptr_t)p.relu_mask % 16) == 0;
}

int gemm_skinny(const dvae_gemm_desc& d, hipStream_t s, int force_nt);  // gemm_skinny.hip: 1 = not served there
int gemm_bf16(const dvae_gemm_desc& d, hipStream_t s);  // gemm_bf16.hip
int gemm_bf16_nn(const dvae_gemm_desc& d, hipStream_t s);  // gemm_bf16_nn.hip: 1 = not served there
int gemm_check(const dvae_gemm_desc& d, int kmult);     // gemm.hip: argument checks common to both
double kmajor_span_limit();                              // bytes a K-major operand may span (32-bit offsets)
int gemm_k_chunk(const dvae_gemm_desc& d, int bk);
bool gemm_next_chunk(const dvae_gemm_desc& d, int kc, int& k0, dvae_gemm_desc& sub);

}  // namespace dvae
