// Pieces shared by the fp32-MFMA and bf16-MFMA GEMM kernels: launch parameters, the XCD-aware
// tile order and the accumulator epilogue (identical C/D register maps on gfx950).
#pragma once
#include "common.h"
#include "dvae_hip.h"

namespace dvae {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0xFFFFFFF0u;  // buffer offset past num_records: the load returns 0

struct GemmParams {
    const float* A; const float* B; float* C;
    const int* a_rows; const int* b_rows; const int* c_rows;
    const float* bias;
    int lda, ldb, ldc;
    int M, N, K;
    int act;         // 0 none, 1 relu, 2 tanh
    int accumulate;  // C += result
    int split_k;     // >1: atomicAdd into C (C must hold the running sum / zeros)
    int k_per_split; // multiple of the kernel's K step
    int tiles_x, tiles_y;  // output tiles along N and M; workgroup w walks tiles w, w + gridDim.x, ...
};

// XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (each with its own
// L2), so workgroup ids that are equal mod 8 share an L2.  Give each such group a contiguous
// run of tiles (N fastest), so the tiles that re-read one A row block meet in one L2.
template <int BM, int BN>
__device__ __forceinline__ void tile_origin(int orig, int nx, int nwg, int& m0, int& n0) {
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    m0 = (id / nx) * BM;
    n0 = (id % nx) * BN;
}

// epilogue: D[row][col], col = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
template <int TM, int TN>
__device__ __forceinline__ void store_acc(const GemmParams& p, f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn,
                                          int lane) {
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

int gemm_bf16(const dvae_gemm_desc& d, hipStream_t s);  // gemm_bf16.hip
int gemm_check(const dvae_gemm_desc& d, int kmult);     // gemm.hip: argument checks common to both
int gemm_k_chunk(const dvae_gemm_desc& d, int bk);
bool gemm_next_chunk(const dvae_gemm_desc& d, int kc, int& k0, dvae_gemm_desc& sub);

}  // namespace dvae
