// HBM/L2-bound kernels of the chain passes: CSR gather-sum, row scatter, GRU and
// gate/mapper pointwise parts (forward and backward), segmented column sums.
// All rows are padded to a multiple of 32 floats, so every access is a 16-byte
// float4 per lane, coalesced along the feature dimension.
#include "common.h"
#include "dvae_hip.h"
#include "model.h"

namespace dvae {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
// bf16 SHADOW of an activation row (round 3, bf16 mode): the kernels that produce a state / predecessor sum that a
// bf16 product reads next also store it rounded to bf16 (8 bytes per lane beside the 16 of the fp32 row, which the
// backward pass and the gathers keep reading): the product then pulls half the A bytes through L2 and converts nothing.
typedef __bf16 pbf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st4h(uint16_t* p, float4 v) {
    const pbf16x4 b = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    *reinterpret_cast<uint2*>(p) = __builtin_bit_cast(uint2, b);
}

// HALF GATES (round 3, bf16 mode, train.hip: half_gates()): what a level's product hands to its tail (the pre-activations)
// and what the forward tails save for the backward ones (r, z, n, gh_n; g, m) is stored as bf16.  The kernels below take
// the same pointers either way; with HG they address 2-byte elements behind them (the caller offsets them in elements).
template <bool HG>
__device__ __forceinline__ float4 ldg4(const float* base, long e) {
    if constexpr (HG) {
        const uint2 r = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + e);
        return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                           __uint_as_float(r.y & 0xffff0000u));
    } else {
        return ld4(base + e);
    }
}
template <bool HG>
__device__ __forceinline__ void stg4(float* base, long e, float4 v) {
    if constexpr (HG) st4h(reinterpret_cast<uint16_t*>(base) + e, v);
    else st4(base + e, v);
}

// ---------------------------------------------------------------------------
__global__ void gather_sum_kernel(float* __restrict__ out, int ld_out, const float* __restrict__ src, int ld_src,
                                  const int* __restrict__ off, const int* __restrict__ idx, int rows, int w4,
                                  int accumulate, uint16_t* __restrict__ out_h) {
    long total = (long)rows * w4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int r = (int)(i / w4), c = (int)(i % w4) * 4;
        int k0 = off[r], k1 = off[r + 1];
        float4 acc = accumulate ? ld4(out + (long)r * ld_out + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        int k = k0;
        // two rows in flight per step
        for (; k + 1 < k1; k += 2) {
            float4 a = ld4(src + (long)idx[k] * ld_src + c);
            float4 b = ld4(src + (long)idx[k + 1] * ld_src + c);
            acc = add4(acc, a);
            acc = add4(acc, b);
        }
        if (k < k1) acc = add4(acc, ld4(src + (long)idx[k] * ld_src + c));
        st4(out + (long)r * ld_out + c, acc);
        if (out_h) st4h(out_h + (long)r * ld_out + c, acc);
    }
}

__global__ void scatter_rows_add_kernel(float* __restrict__ dst, int ld_dst, const float* __restrict__ src, int ld_src,
                                        const int* __restrict__ off, const int* __restrict__ idx, int rows, int w4) {
    long total = (long)rows * w4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int r = (int)(i / w4), c = (int)(i % w4) * 4;
        float4 v = ld4(src + (long)r * ld_src + c);
        for (int k = off[r]; k < off[r + 1]; ++k) {
            float* d = dst + (long)idx[k] * ld_dst + c;
            st4(d, add4(ld4(d), v));
        }
    }
}

// ---------------------------------------------------------------------------
// HG: the pre-activations come from `pre` (bf16, [rows][3 Hp]); gates and gh_n are saved as bf16
template <bool HG>
__global__ void gru_fwd_kernel(float* __restrict__ gh, const float* __restrict__ gi_tab, const int* __restrict__ types,
                               const float* __restrict__ bhn, const float* __restrict__ hprev, int ld_hprev,
                               float* __restrict__ hout, int ld_hout, float* __restrict__ ghn_out, int rows, int Hp,
                               uint16_t* __restrict__ hout_h, const float* __restrict__ pre) {
    int w4 = Hp / 4;
    long total = (long)rows * w4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int r = (int)(i / w4), c = (int)(i % w4) * 4;
        const long g0 = (long)r * 3 * Hp;
        const float* src = HG ? pre : gh;
        const float* gi = gi_tab + (long)types[r] * 3 * Hp;
        float4 ghr = ldg4<HG>(src, g0 + c), ghz = ldg4<HG>(src, g0 + Hp + c), ghn = ldg4<HG>(src, g0 + 2 * Hp + c);
        float4 gir = ld4(gi + c), giz = ld4(gi + Hp + c), gin = ld4(gi + 2 * Hp + c);
        float4 bn = ld4(bhn + c);
        float4 hp = hprev ? ld4(hprev + (long)r * ld_hprev + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 R, Z, N, Hn, GN;
#define GRU1(f)                                        \
    {                                                  \
        float gn = ghn.f + bn.f;                       \
        float rr = 1.f / (1.f + expf(-(gir.f + ghr.f))); \
        float zz = 1.f / (1.f + expf(-(giz.f + ghz.f))); \
        float nn = tanhf(gin.f + rr * gn);             \
        R.f = rr; Z.f = zz; N.f = nn; GN.f = gn;       \
        Hn.f = (1.f - zz) * nn + zz * hp.f;            \
    }
        GRU1(x) GRU1(y) GRU1(z) GRU1(w)
#undef GRU1
        stg4<HG>(gh, g0 + c, R);
        stg4<HG>(gh, g0 + Hp + c, Z);
        stg4<HG>(gh, g0 + 2 * Hp + c, N);
        stg4<HG>(ghn_out, (long)r * Hp + c, GN);
        st4(hout + (long)r * ld_hout + c, Hn);
        if (hout_h) st4h(hout_h + (long)r * ld_hout + c, Hn);
    }
}

// The first level of an encoder pass: no predecessors, so h_prev = 0 and the hidden product is 0 (models.py:682 with
// H = zeros).  Same arithmetic as gru_fwd_kernel on a zero gh / zero h_prev, without reading either; also leaves the
// zeros the backward pass expects in the level's agg rows (its h_prev) -- one launch instead of two memsets + one.
template <bool HG>
__global__ void gru_fwd_first_kernel(float* __restrict__ gates, const float* __restrict__ gi_tab,
                                     const int* __restrict__ types, const float* __restrict__ bhn,
                                     float* __restrict__ agg_zero, float* __restrict__ hout, float* __restrict__ ghn_out,
                                     int rows, int Hp, uint16_t* __restrict__ hout_h) {
    int w4 = Hp / 4;
    long total = (long)rows * w4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int r = (int)(i / w4), c = (int)(i % w4) * 4;
        const long g0 = (long)r * 3 * Hp;
        const float* gi = gi_tab + (long)types[r] * 3 * Hp;
        float4 gir = ld4(gi + c), giz = ld4(gi + Hp + c), gin = ld4(gi + 2 * Hp + c);
        float4 bn = ld4(bhn + c);
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
        stg4<HG>(gates, g0 + c, R);
        stg4<HG>(gates, g0 + Hp + c, Z);
        stg4<HG>(gates, g0 + 2 * Hp + c, N);
        stg4<HG>(ghn_out, (long)r * Hp + c, bn);
        st4(hout + (long)r * Hp + c, Hn);
        if (hout_h) st4h(hout_h + (long)r * Hp + c, Hn);
        st4(agg_zero + (long)r * Hp + c, make_float4(0.f, 0.f, 0.f, 0.f));
    }
}

template <bool HG>
__global__ void gru_bwd_kernel(const float* __restrict__ dh, int ld_dh, const float* __restrict__ gates,
                               const float* __restrict__ ghn, const float* __restrict__ hprev, int ld_hprev,
                               float* __restrict__ dgh, float* __restrict__ dgin, float* __restrict__ dhprev,
                               int ld_dhprev, int rows, int Hp, uint16_t* __restrict__ dgh_h) {
    int w4 = Hp / 4;
    long total = (long)rows * w4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int r = (int)(i / w4), c = (int)(i % w4) * 4;
        const long g0 = (long)r * 3 * Hp;
        float4 R = ldg4<HG>(gates, g0 + c), Z = ldg4<HG>(gates, g0 + Hp + c), N = ldg4<HG>(gates, g0 + 2 * Hp + c);
        float4 GN = ldg4<HG>(ghn, (long)r * Hp + c);
        float4 D = ld4(dh + (long)r * ld_dh + c);
        float4 hp = hprev ? ld4(hprev + (long)r * ld_hprev + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 dR, dZ, dGN, dPN, dHP;
#define GRUB(f)                                      \
    {                                                \
        float d = D.f;                               \
        float dn = d * (1.f - Z.f);                  \
        float dz = d * (hp.f - N.f);                 \
        float dpn = dn * (1.f - N.f * N.f);          \
        float dr = dpn * GN.f;                       \
        dGN.f = dpn * R.f;                           \
        dPN.f = dpn;                                 \
        dR.f = dr * R.f * (1.f - R.f);               \
        dZ.f = dz * Z.f * (1.f - Z.f);               \
        dHP.f = d * Z.f;                             \
    }
        GRUB(x) GRUB(y) GRUB(z) GRUB(w)
#undef GRUB
        if (dgh) {  // (null: the bf16 copy below is the only one, train.hip half_grads())
            float* o = dgh + (long)r * 3 * Hp;
            st4(o + c, dR);
            st4(o + Hp + c, dZ);
            st4(o + 2 * Hp + c, dGN);
        }
        if (dgh_h) {  // bf16 shadow of d gh: the A operand of the level's data-gradient product
            uint16_t* oh = dgh_h + (long)r * 3 * Hp;
            st4h(oh + c, dR);
            st4h(oh + Hp + c, dZ);
            st4h(oh + 2 * Hp + c, dGN);
        }
        // (half grads -- no fp32 d gh: d pre_n goes out as bf16 too; dgin then points at bf16 words)
        if (dgh) st4(dgin + (long)r * Hp + c, dPN);
        else stg4<true>(dgin, (long)r * Hp + c, dPN);
        if (dhprev) st4(dhprev + (long)r * ld_dhprev + c, dHP);
    }
}

// ---------------------------------------------------------------------------
__global__ void gated_fwd_kernel(float* __restrict__ gm, const float* __restrict__ gv, float* __restrict__ msg,
                                 int ld_msg, int rows, int Hp, int row_base, int rows_per_level) {
    int w4 = Hp / 4;
    long total = (long)rows * w4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int r = (int)(i / w4), c = (int)(i % w4) * 4;
        int lvl = (row_base + r) / rows_per_level;
        float* p = gm + (long)r * 2 * Hp;
        const float* v = gv + (long)lvl * 2 * Hp;
        float4 a = add4(ld4(p + c), ld4(v + c));
        float4 m = add4(ld4(p + Hp + c), ld4(v + Hp + c));
        float4 g = make_float4(1.f / (1.f + expf(-a.x)), 1.f / (1.f + expf(-a.y)), 1.f / (1.f + expf(-a.z)),
                               1.f / (1.f + expf(-a.w)));
        st4(p + c, g);
        st4(p + Hp + c, m);
        st4(msg + (long)r * ld_msg + c, make_float4(g.x * m.x, g.y * m.y, g.z * m.z, g.w * m.w));
    }
}

__global__ void gated_bwd_kernel(const float* __restrict__ dmsg, int ld_dmsg, const float* __restrict__ gm,
                                 float* __restrict__ dgm, int rows, int Hp) {
    int w4 = Hp / 4;
    long total = (long)rows * w4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int r = (int)(i / w4), c = (int)(i % w4) * 4;
        const float* p = gm + (long)r * 2 * Hp;
        float4 g = ld4(p + c), m = ld4(p + Hp + c), d = ld4(dmsg + (long)r * ld_dmsg + c);
        float4 dg = make_float4(d.x * m.x * g.x * (1.f - g.x), d.y * m.y * g.y * (1.f - g.y),
                                d.z * m.z * g.z * (1.f - g.z), d.w * m.w * g.w * (1.f - g.w));
        float4 dm = make_float4(d.x * g.x, d.y * g.y, d.z * g.z, d.w * g.w);
        float* o = dgm + (long)r * 2 * Hp;
        st4(o + c, dg);
        st4(o + Hp + c, dm);
    }
}

// gate/mapper pointwise part of the level at rows [cur_row0, cur_row0 + rows) fused with the
// CSR gather-sum that feeds the NEXT level: both are independent per feature column, and every
// source slot of graph b's next vertex lies in graph b, so the only not-yet-stored message a lane
// needs is the one it has just computed (slot cur_row0 + b), which it keeps in registers.
template <bool HG>
__global__ void gated_fwd_gather_kernel(float* __restrict__ gm, const float* __restrict__ gv, float* __restrict__ msg,
                                        int ld_msg, int cur_row0, float* __restrict__ next_agg, int ld_agg,
                                        const int* __restrict__ next_off, const int* __restrict__ idx, int rows,
                                        int Hp, uint16_t* __restrict__ next_agg_h, const float* __restrict__ pre) {
    int w4 = Hp / 4;
    long total = (long)rows * w4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int r = (int)(i / w4), c = (int)(i % w4) * 4;
        const long p0 = (long)r * 2 * Hp;
        const float* src = HG ? pre : gm;
        float4 a = add4(ldg4<HG>(src, p0 + c), ld4(gv + c));
        float4 m = add4(ldg4<HG>(src, p0 + Hp + c), ld4(gv + Hp + c));
        float4 g = make_float4(1.f / (1.f + expf(-a.x)), 1.f / (1.f + expf(-a.y)), 1.f / (1.f + expf(-a.z)),
                               1.f / (1.f + expf(-a.w)));
        float4 own = make_float4(g.x * m.x, g.y * m.y, g.z * m.z, g.w * m.w);
        stg4<HG>(gm, p0 + c, g);
        stg4<HG>(gm, p0 + Hp + c, m);
        st4(msg + (long)(cur_row0 + r) * ld_msg + c, own);
        if (next_agg) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int k = next_off[r]; k < next_off[r + 1]; ++k) {
                int s = idx[k];
                acc = add4(acc, s == cur_row0 + r ? own : ld4(msg + (long)s * ld_msg + c));
            }
            st4(next_agg + (long)r * ld_agg + c, acc);
            if (next_agg_h) st4h(next_agg_h + (long)r * ld_agg + c, acc);
        }
    }
}

// backward twin: d msg of a level = CSR gather-sum of the consumers' d agg rows (one or two lists),
// consumed in registers by the gate/mapper pointwise backward; d msg itself is never stored.
template <bool HG>
__global__ void gated_bwd_gather_kernel(const float* __restrict__ dagg, int ld, const int* __restrict__ off,
                                        const int* __restrict__ idx, const float* __restrict__ dagg2,
                                        const int* __restrict__ off2, const int* __restrict__ idx2,
                                        const float* __restrict__ gm, float* __restrict__ dgm, int rows, int Hp,
                                        uint16_t* __restrict__ dgm_h) {
    int w4 = Hp / 4;
    long total = (long)rows * w4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int r = (int)(i / w4), c = (int)(i % w4) * 4;
        float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = off[r]; k < off[r + 1]; ++k) d = add4(d, ld4(dagg + (long)idx[k] * ld + c));
        if (off2)
            for (int k = off2[r]; k < off2[r + 1]; ++k) d = add4(d, ld4(dagg2 + (long)idx2[k] * ld + c));
        const long p0 = (long)r * 2 * Hp;
        float4 g = ldg4<HG>(gm, p0 + c), m = ldg4<HG>(gm, p0 + Hp + c);
        float4 dg = make_float4(d.x * m.x * g.x * (1.f - g.x), d.y * m.y * g.y * (1.f - g.y),
                                d.z * m.z * g.z * (1.f - g.z), d.w * m.w * g.w * (1.f - g.w));
        float4 dm = make_float4(d.x * g.x, d.y * g.y, d.z * g.z, d.w * g.w);
        if (dgm) {  // (null: the bf16 copy below is the only one)
            float* o = dgm + (long)r * 2 * Hp;
            st4(o + c, dg);
            st4(o + Hp + c, dm);
        }
        if (dgm_h) {  // bf16 shadow of d gm: the A operand of the level's data-gradient product
            uint16_t* oh = dgm_h + (long)r * 2 * Hp;
            st4h(oh + c, dg);
            st4h(oh + Hp + c, dm);
        }
    }
}

// ---------------------------------------------------------------------------
// out[seg][c] += sum_{rows of seg} x[row][c].  grid: (width/4/64, chunks, segs)
// XH: x is stored as bf16 (a gradient table written so by its producer); ld_x counts elements either way.
template <bool XH>
__device__ __forceinline__ float4 ld4x(const void* x, long off) {
    if constexpr (XH) {
        const uint2 r = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(x) + off);
        return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                           __uint_as_float(r.y & 0xffff0000u));
    } else {
        return ld4(reinterpret_cast<const float*>(x) + off);
    }
}
template <bool XH>
__global__ __launch_bounds__(256) void segment_colsum_kernel(float* __restrict__ out, int ld_out,
                                                             const void* __restrict__ x, int ld_x, int rows,
                                                             int seg_rows, int w4, int rows_per_chunk) {
    __shared__ float4 red[4][64];
    int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    int c4 = blockIdx.x * 64 + tx;
    int seg = blockIdx.z;
    int r0 = seg * seg_rows + blockIdx.y * rows_per_chunk;
    int r1 = min(min(r0 + rows_per_chunk, (seg + 1) * seg_rows), rows);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < w4) {
        // four independent row streams per thread keep 4 loads in flight (the loop is latency bound)
        const long base = c4 * 4;
        float4 a1 = acc, a2 = acc, a3 = acc;
        int r = r0 + ty;
        for (; r + 12 < r1; r += 16) {
            float4 v0 = ld4x<XH>(x, base + (long)r * ld_x), v1 = ld4x<XH>(x, base + (long)(r + 4) * ld_x);
            float4 v2 = ld4x<XH>(x, base + (long)(r + 8) * ld_x), v3 = ld4x<XH>(x, base + (long)(r + 12) * ld_x);
            acc = add4(acc, v0); a1 = add4(a1, v1); a2 = add4(a2, v2); a3 = add4(a3, v3);
        }
        for (; r < r1; r += 4) acc = add4(acc, ld4x<XH>(x, base + (long)r * ld_x));
        acc = add4(add4(acc, a1), add4(a2, a3));
    }
    red[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && c4 < w4) {
        acc = add4(add4(red[0][tx], red[1][tx]), add4(red[2][tx], red[3][tx]));
        float* o = out + (long)seg * ld_out + c4 * 4;
        atomicAdd(o + 0, acc.x);
        atomicAdd(o + 1, acc.y);
        atomicAdd(o + 2, acc.z);
        atomicAdd(o + 3, acc.w);
    }
}

// Several column sums in ONE launch (the bias gradients of the loss backward: four reductions whose inputs are all final at
// the same point of the stream): a 1-D grid, each job owns the run of workgroups [first, first + xblocks * chunks) with its
// own chunking; out[c] += scale * sum_rows x[row][c].
struct ColsumJob {
    float* out;
    const void* x;
    const float* scale;  // nullable: one device float
    int ld_x, rows, w4, rows_per_chunk, chunks, x_bf16, first, xblocks;
};
struct ColsumJobs {
    ColsumJob j[4];
};
template <bool XH>
__device__ __forceinline__ float4 colsum_rows(const void* x, int ld_x, long base, int r0, int r1, int ty) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), a1 = acc, a2 = acc, a3 = acc;
    int r = r0 + ty;
    for (; r + 12 < r1; r += 16) {
        float4 v0 = ld4x<XH>(x, base + (long)r * ld_x), v1 = ld4x<XH>(x, base + (long)(r + 4) * ld_x);
        float4 v2 = ld4x<XH>(x, base + (long)(r + 8) * ld_x), v3 = ld4x<XH>(x, base + (long)(r + 12) * ld_x);
        acc = add4(acc, v0); a1 = add4(a1, v1); a2 = add4(a2, v2); a3 = add4(a3, v3);
    }
    for (; r < r1; r += 4) acc = add4(acc, ld4x<XH>(x, base + (long)r * ld_x));
    return add4(add4(acc, a1), add4(a2, a3));
}
__global__ __launch_bounds__(256) void colsum_jobs_kernel(ColsumJobs J) {
    __shared__ float4 red[4][64];
    int ji = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (J.j[i].chunks > 0 && (int)blockIdx.x >= J.j[i].first) ji = i;
    const ColsumJob& jb = J.j[ji];
    const int local = (int)blockIdx.x - jb.first, bx = local % jb.xblocks, by = local / jb.xblocks;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c4 = bx * 64 + tx;
    const int r0 = by * jb.rows_per_chunk, r1 = min(r0 + jb.rows_per_chunk, jb.rows);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < jb.w4)
        acc = jb.x_bf16 ? colsum_rows<true>(jb.x, jb.ld_x, c4 * 4L, r0, r1, ty) : colsum_rows<false>(jb.x, jb.ld_x, c4 * 4L, r0, r1, ty);
    red[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && c4 < jb.w4) {
        acc = add4(add4(red[0][tx], red[1][tx]), add4(red[2][tx], red[3][tx]));
        const float sc = jb.scale ? jb.scale[0] : 1.f;
        float* o = jb.out + c4 * 4;
        atomicAdd(o + 0, acc.x * sc);
        atomicAdd(o + 1, acc.y * sc);
        atomicAdd(o + 2, acc.z * sc);
        atomicAdd(o + 3, acc.w * sc);
    }
}

// ---------------------------------------------------------------------------
// Rows sorted by vertex type (counting sort, ONE workgroup): off[T + 1], idx[na + nb].  The rows
// are those of two type arrays laid end to end (slots, then the decoder's partial rows).
__global__ __launch_bounds__(1024) void type_csr_kernel(const int* __restrict__ ta, int na, const int* __restrict__ tb,
                                                         int nb, int T, int* __restrict__ off, int* __restrict__ idx) {
    __shared__ int hist[33], cur[32];
    const int tid = threadIdx.x, n = na + nb;
    if (tid < 33) hist[tid] = 0;
    __syncthreads();
    for (int r = tid; r < n; r += 1024) {
        const int t = r < na ? ta[r] : tb[r - na];
        if (t >= 0 && t < T) atomicAdd(&hist[t], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int s = 0;
        for (int t = 0; t < T; ++t) {
            const int c = hist[t];
            off[t] = s;
            cur[t] = s;
            s += c;
        }
        off[T] = s;
    }
    __syncthreads();
    for (int r = tid; r < n; r += 1024) {
        const int t = r < na ? ta[r] : tb[r - na];
        if (t >= 0 && t < T) idx[atomicAdd(&cur[t], 1)] = r;
    }
}

// Per-type column sums of the virtual row [dgh (3Hp) | dgin (Hp)] -- everything the GRU's input-side
// and bias gradients need from ONE read of the two arrays:
//   gi[t][0:2Hp]   += sum over rows of type t of dgh[:, 0:2Hp]      (d pre_r, d pre_z)
//   bhn[0:Hp]      += sum over all rows of dgh[:, 2Hp:3Hp]          (d (W_hn h + b_hn))
//   gi[t][2Hp:3Hp] += sum over rows of type t of dgin               (d pre_n)
// grid (ceil(Hp / 64) column blocks of 64 float4, row chunks, T); partial sums are added with atomics.
// GH: dgh and dgin point at bf16 words (the backward tails' bf16-only gradients: train.hip half_grads())
template <bool GH>
__global__ __launch_bounds__(256) void type_colsum_kernel(const float* __restrict__ dgh, const float* __restrict__ dgin,
                                                          const int* __restrict__ off, const int* __restrict__ idx,
                                                          float* __restrict__ gi, float* __restrict__ bhn, int Hp,
                                                          int rows_per_chunk) {
    __shared__ float4 red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int t = blockIdx.z;
    const int k0 = off[t] + blockIdx.y * rows_per_chunk;
    const int k1 = min(off[t + 1], k0 + rows_per_chunk);
    if (k0 >= k1) return;  // (uniform over the workgroup)
    const int c4 = blockIdx.x * 64 + tx;  // float4 column of the 4Hp-wide virtual row
    const bool on = c4 < Hp;
    const int col = c4 * 4;
    const bool in_gh = col < 3 * Hp;
    const long ld = in_gh ? 3 * Hp : Hp;
    const long c0 = in_gh ? col : col - 3 * Hp;
    auto row = [&](int r) -> float4 {
        if (GH) return ldg4<true>(in_gh ? dgh : dgin, c0 + r * ld);
        return ld4((in_gh ? dgh : dgin) + c0 + r * ld);
    };
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (on) {
        float4 a1 = acc, a2 = acc, a3 = acc;
        int k = k0 + ty;
        for (; k + 12 < k1; k += 16) {  // four independent row streams per thread
            float4 v0 = row(idx[k]), v1 = row(idx[k + 4]);
            float4 v2 = row(idx[k + 8]), v3 = row(idx[k + 12]);
            acc = add4(acc, v0); a1 = add4(a1, v1); a2 = add4(a2, v2); a3 = add4(a3, v3);
        }
        for (; k < k1; k += 4) acc = add4(acc, row(idx[k]));
        acc = add4(add4(acc, a1), add4(a2, a3));
    }
    red[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && on) {
        acc = add4(add4(red[0][tx], red[1][tx]), add4(red[2][tx], red[3][tx]));
        float* o = col < 2 * Hp ? gi + (long)t * 3 * Hp + col
                 : (col < 3 * Hp ? bhn + (col - 2 * Hp) : gi + (long)t * 3 * Hp + 2 * Hp + (col - 3 * Hp));
        atomicAdd(o + 0, acc.x);
        atomicAdd(o + 1, acc.y);
        atomicAdd(o + 2, acc.z);
        atomicAdd(o + 3, acc.w);
    }
}

static inline int ew_grid(long total) {
    long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace dvae

using namespace dvae;

namespace dvae {
// out_h (nullable): bf16 shadow of `out`, same row pitch in elements
int gather_sum_h(float* out, int ld_out, const float* src, int ld_src, const int* off, const int* idx, int rows, int width,
                 int accumulate, uint16_t* out_h, hipStream_t s) {
    if (rows == 0) return DVAE_OK;
    DVAE_REQUIRE(out && src && off, "gather_sum: null pointer");
    DVAE_REQUIRE(width > 0 && width % 4 == 0 && ld_out % 4 == 0 && ld_src % 4 == 0, "gather_sum: width/ld must be multiples of 4");
    long total = (long)rows * (width / 4);
    hipLaunchKernelGGL(gather_sum_kernel, dim3(ew_grid(total)), dim3(256), 0, s, out, ld_out, src, ld_src, off, idx, rows,
                       width / 4, accumulate, out_h);
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}
}  // namespace dvae
extern "C" int dvae_gather_sum_f32(float* out, int32_t ld_out, const float* src, int32_t ld_src, const int32_t* off,
                                   const int32_t* idx, int32_t rows, int32_t width, int32_t accumulate, void* stream) {
    return dvae::gather_sum_h(out, ld_out, src, ld_src, off, idx, rows, width, accumulate, nullptr, (hipStream_t)stream);
}

extern "C" int dvae_scatter_rows_add_f32(float* dst, int32_t ld_dst, const float* src, int32_t ld_src,
                                         const int32_t* off, const int32_t* idx, int32_t rows, int32_t width,
                                         void* stream) {
    if (rows == 0) return DVAE_OK;
    DVAE_REQUIRE(dst && src && off, "scatter_rows_add: null pointer");
    DVAE_REQUIRE(width > 0 && width % 4 == 0 && ld_dst % 4 == 0 && ld_src % 4 == 0, "scatter_rows_add: width/ld must be multiples of 4");
    long total = (long)rows * (width / 4);
    hipLaunchKernelGGL(scatter_rows_add_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, dst, ld_dst,
                       src, ld_src, off, idx, rows, width / 4);
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}

namespace dvae {
// hout_h (nullable): bf16 shadow of the new states, row pitch ld_hout elements
// pre (nullable): HALF-GATES form -- the pre-activations are read from `pre` (bf16 words, [rows][3 Hp]) and gh / ghn_out
// point at bf16 words too (see ldg4 / stg4)
int gru_pointwise_fwd_h(float* gh, const float* gi_tab, const int* types, const float* bhn, const float* hprev, int ld_hprev,
                        float* hout, int ld_hout, float* ghn_out, int rows, int Hp, uint16_t* hout_h, hipStream_t s,
                        const uint16_t* pre) {
    if (rows == 0) return DVAE_OK;
    DVAE_REQUIRE(gh && gi_tab && types && bhn && hout && ghn_out, "gru_pointwise_fwd: null pointer");
    DVAE_REQUIRE(Hp > 0 && Hp % 4 == 0 && ld_hout % 4 == 0 && (!hprev || ld_hprev % 4 == 0), "gru_pointwise_fwd: bad Hp/ld");
    long total = (long)rows * (Hp / 4);
    if (pre)
        hipLaunchKernelGGL(gru_fwd_kernel<true>, dim3(ew_grid(total)), dim3(256), 0, s, gh, gi_tab, types, bhn, hprev, ld_hprev,
                           hout, ld_hout, ghn_out, rows, Hp, hout_h, reinterpret_cast<const float*>(pre));
    else
        hipLaunchKernelGGL(gru_fwd_kernel<false>, dim3(ew_grid(total)), dim3(256), 0, s, gh, gi_tab, types, bhn, hprev, ld_hprev,
                           hout, ld_hout, ghn_out, rows, Hp, hout_h, (const float*)nullptr);
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}
}  // namespace dvae
extern "C" int dvae_gru_pointwise_fwd(float* gh, const float* gi_tab, const int32_t* types, const float* bhn,
                                      const float* hprev, int32_t ld_hprev, float* hout, int32_t ld_hout,
                                      float* ghn_out, int32_t rows, int32_t Hp, void* stream) {
    return dvae::gru_pointwise_fwd_h(gh, gi_tab, types, bhn, hprev, ld_hprev, hout, ld_hout, ghn_out, rows, Hp, nullptr,
                                     (hipStream_t)stream, nullptr);
}

namespace dvae {
int gru_first_level(float* gates, const float* gi_tab, const int32_t* types, const float* bhn, float* agg_zero, float* hout,
                    float* ghn_out, int rows, int Hp, hipStream_t s, uint16_t* hout_h, int hg) {
    if (rows == 0) return DVAE_OK;
    long total = (long)rows * (Hp / 4);
    if (hg)
        hipLaunchKernelGGL(gru_fwd_first_kernel<true>, dim3(ew_grid(total)), dim3(256), 0, s, gates, gi_tab, types, bhn, agg_zero,
                           hout, ghn_out, rows, Hp, hout_h);
    else
        hipLaunchKernelGGL(gru_fwd_first_kernel<false>, dim3(ew_grid(total)), dim3(256), 0, s, gates, gi_tab, types, bhn, agg_zero,
                           hout, ghn_out, rows, Hp, hout_h);
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}
}  // namespace dvae

namespace dvae {
// dgh_h (nullable): bf16 shadow of d gh, row pitch 3 Hp elements
// hg: gates / ghn point at bf16 words (HALF-GATES form)
int gru_pointwise_bwd_h(const float* dh, int ld_dh, const float* gates, const float* ghn, const float* hprev, int ld_hprev,
                        float* dgh, float* dgin, float* dhprev, int ld_dhprev, int rows, int Hp, uint16_t* dgh_h, hipStream_t s,
                        int hg) {
    if (rows == 0) return DVAE_OK;
    DVAE_REQUIRE(dh && gates && ghn && (dgh || dgh_h) && dgin, "gru_pointwise_bwd: null pointer");
    DVAE_REQUIRE(Hp > 0 && Hp % 4 == 0 && ld_dh % 4 == 0, "gru_pointwise_bwd: bad Hp/ld");
    long total = (long)rows * (Hp / 4);
    if (hg)
        hipLaunchKernelGGL(gru_bwd_kernel<true>, dim3(ew_grid(total)), dim3(256), 0, s, dh, ld_dh, gates, ghn, hprev, ld_hprev,
                           dgh, dgin, dhprev, ld_dhprev, rows, Hp, dgh_h);
    else
        hipLaunchKernelGGL(gru_bwd_kernel<false>, dim3(ew_grid(total)), dim3(256), 0, s, dh, ld_dh, gates, ghn, hprev, ld_hprev,
                           dgh, dgin, dhprev, ld_dhprev, rows, Hp, dgh_h);
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}
}  // namespace dvae
extern "C" int dvae_gru_pointwise_bwd(const float* dh, int32_t ld_dh, const float* gates, const float* ghn,
                                      const float* hprev, int32_t ld_hprev, float* dgh, float* dgin, float* dhprev,
                                      int32_t ld_dhprev, int32_t rows, int32_t Hp, void* stream) {
    return dvae::gru_pointwise_bwd_h(dh, ld_dh, gates, ghn, hprev, ld_hprev, dgh, dgin, dhprev, ld_dhprev, rows, Hp, nullptr,
                                     (hipStream_t)stream, 0);
}

extern "C" int dvae_gated_pointwise_fwd(float* gm, const float* gv, float* msg, int32_t ld_msg, int32_t rows,
                                        int32_t Hp, int32_t row_base, int32_t rows_per_level, void* stream) {
    if (rows == 0) return DVAE_OK;
    DVAE_REQUIRE(gm && gv && msg, "gated_pointwise_fwd: null pointer");
    DVAE_REQUIRE(Hp > 0 && Hp % 4 == 0 && ld_msg % 4 == 0 && rows_per_level > 0, "gated_pointwise_fwd: bad size");
    long total = (long)rows * (Hp / 4);
    hipLaunchKernelGGL(gated_fwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, gm, gv, msg, ld_msg,
                       rows, Hp, row_base, rows_per_level);
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}

extern "C" int dvae_gated_pointwise_bwd(const float* dmsg, int32_t ld_dmsg, const float* gm, float* dgm, int32_t rows,
                                        int32_t Hp, void* stream) {
    if (rows == 0) return DVAE_OK;
    DVAE_REQUIRE(dmsg && gm && dgm, "gated_pointwise_bwd: null pointer");
    DVAE_REQUIRE(Hp > 0 && Hp % 4 == 0 && ld_dmsg % 4 == 0, "gated_pointwise_bwd: bad size");
    long total = (long)rows * (Hp / 4);
    hipLaunchKernelGGL(gated_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, dmsg, ld_dmsg, gm,
                       dgm, rows, Hp);
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}

namespace dvae {
// next_agg_h (nullable): bf16 shadow of the next level's predecessor sums, row pitch ld_agg elements
// pre (nullable): HALF-GATES form -- pre-activations from `pre` (bf16 words, [rows][2 Hp]), gm points at bf16 words
int gated_fwd_gather_h(float* gm, const float* gv_row, float* msg, int ld_msg, int cur_row0, float* next_agg, int ld_agg,
                       const int* next_off, const int* idx, int rows, int Hp, uint16_t* next_agg_h, hipStream_t s,
                       const uint16_t* pre) {
    if (rows == 0) return DVAE_OK;
    DVAE_REQUIRE(gm && gv_row && msg, "gated_fwd_gather: null pointer");
    DVAE_REQUIRE(!next_agg || (next_off && idx), "gated_fwd_gather: the next level's CSR lists are missing");
    DVAE_REQUIRE(Hp > 0 && Hp % 4 == 0 && ld_msg % 4 == 0 && (!next_agg || ld_agg % 4 == 0) && cur_row0 >= 0,
                 "gated_fwd_gather: bad size");
    long total = (long)rows * (Hp / 4);
    if (pre)
        hipLaunchKernelGGL(gated_fwd_gather_kernel<true>, dim3(ew_grid(total)), dim3(256), 0, s, gm, gv_row, msg, ld_msg, cur_row0,
                           next_agg, ld_agg, next_off, idx, rows, Hp, next_agg ? next_agg_h : nullptr,
                           reinterpret_cast<const float*>(pre));
    else
        hipLaunchKernelGGL(gated_fwd_gather_kernel<false>, dim3(ew_grid(total)), dim3(256), 0, s, gm, gv_row, msg, ld_msg, cur_row0,
                           next_agg, ld_agg, next_off, idx, rows, Hp, next_agg ? next_agg_h : nullptr, (const float*)nullptr);
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}
}  // namespace dvae
extern "C" int dvae_gated_fwd_gather(float* gm, const float* gv_row, float* msg, int32_t ld_msg, int32_t cur_row0,
                                     float* next_agg, int32_t ld_agg, const int32_t* next_off, const int32_t* idx,
                                     int32_t rows, int32_t Hp, void* stream) {
    return dvae::gated_fwd_gather_h(gm, gv_row, msg, ld_msg, cur_row0, next_agg, ld_agg, next_off, idx, rows, Hp, nullptr,
                                    (hipStream_t)stream, nullptr);
}

namespace dvae {
// dgm_h (nullable): bf16 shadow of d gm, row pitch 2 Hp elements
// hg: gm points at bf16 words (HALF-GATES form)
int gated_bwd_gather_h(const float* dagg, int ld, const int* off, const int* idx, const float* dagg2, const int* off2,
                       const int* idx2, const float* gm, float* dgm, int rows, int Hp, uint16_t* dgm_h, hipStream_t s, int hg) {
    if (rows == 0) return DVAE_OK;
    DVAE_REQUIRE(dagg && off && idx && gm && (dgm || dgm_h), "gated_bwd_gather: null pointer");
    DVAE_REQUIRE(!off2 || (dagg2 && idx2), "gated_bwd_gather: second list incomplete");
    DVAE_REQUIRE(Hp > 0 && Hp % 4 == 0 && ld % 4 == 0, "gated_bwd_gather: bad size");
    long total = (long)rows * (Hp / 4);
    if (hg)
        hipLaunchKernelGGL(gated_bwd_gather_kernel<true>, dim3(ew_grid(total)), dim3(256), 0, s, dagg, ld, off, idx, dagg2, off2,
                           idx2, gm, dgm, rows, Hp, dgm_h);
    else
        hipLaunchKernelGGL(gated_bwd_gather_kernel<false>, dim3(ew_grid(total)), dim3(256), 0, s, dagg, ld, off, idx, dagg2, off2,
                           idx2, gm, dgm, rows, Hp, dgm_h);
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}
}  // namespace dvae
extern "C" int dvae_gated_bwd_gather(const float* dagg, int32_t ld, const int32_t* off, const int32_t* idx,
                                     const float* dagg2, const int32_t* off2, const int32_t* idx2, const float* gm,
                                     float* dgm, int32_t rows, int32_t Hp, void* stream) {
    return dvae::gated_bwd_gather_h(dagg, ld, off, idx, dagg2, off2, idx2, gm, dgm, rows, Hp, nullptr, (hipStream_t)stream, 0);
}

extern "C" int dvae_type_csr(const int32_t* types_a, int32_t na, const int32_t* types_b, int32_t nb, int32_t T,
                             int32_t* off, int32_t* idx, void* stream) {
    DVAE_REQUIRE(na >= 0 && nb >= 0 && T >= 1 && T <= 32, "type_csr: bad sizes (T=%d)", T);
    DVAE_REQUIRE(off && idx && (na == 0 || types_a) && (nb == 0 || types_b), "type_csr: null pointer");
    hipLaunchKernelGGL(type_csr_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, types_a, na, types_b, nb, T, off, idx);
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}

extern "C" int dvae_type_colsum(const float* dgh, const float* dgin, const int32_t* off, const int32_t* idx, int32_t T,
                                float* gi, float* bhn, int32_t rows, int32_t Hp, void* stream) {
    return dvae::type_colsum_h(dgh, dgin, off, idx, T, gi, bhn, rows, Hp, (hipStream_t)stream, 0);
}
// gh: dgh and dgin point at bf16 words
int dvae::type_colsum_h(const float* dgh, const float* dgin, const int32_t* off, const int32_t* idx, int32_t T, float* gi,
                        float* bhn, int32_t rows, int32_t Hp, hipStream_t stream, int gh) {
    if (rows == 0) return DVAE_OK;
    DVAE_REQUIRE(dgh && dgin && off && idx && gi && bhn, "type_colsum: null pointer");
    DVAE_REQUIRE(T >= 1 && T <= 32 && Hp > 0 && Hp % 4 == 0, "type_colsum: bad sizes (T=%d Hp=%d)", T, Hp);
    const int xblocks = cdiv(Hp, 64);
    // row chunks: the lists' lengths are not known on the host, so every type gets chunks for `rows`
    // rows and the surplus workgroups leave at once; bounded to ~16 k workgroups per launch
    long rpc = ((long)rows * T * xblocks + 16383) / 16384;
    if (rpc < 128) rpc = 128;  // (cfg 2, 8192 rows: 32 / 64 / 128 / 256 rows per chunk -> 37 / 22 / 18.5 / 28 us)
    const int chunks = cdiv(rows, rpc);
    if (gh)
        hipLaunchKernelGGL(type_colsum_kernel<true>, dim3(xblocks, chunks, T), dim3(256), 0, stream, dgh, dgin, off, idx, gi, bhn,
                           Hp, (int)rpc);
    else
        hipLaunchKernelGGL(type_colsum_kernel<false>, dim3(xblocks, chunks, T), dim3(256), 0, stream, dgh, dgin, off, idx, gi, bhn,
                           Hp, (int)rpc);
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}

static int segment_colsum_any(float* out, int32_t ld_out, const void* x, bool x_bf16, int32_t ld_x, int32_t rows,
                              int32_t seg_rows, int32_t width, void* stream);
namespace dvae {
// up to four column sums as one launch; jobs with no rows are skipped
int colsum_jobs(const ColsumJobDesc* jobs, int n, hipStream_t s) {
    DVAE_REQUIRE(jobs && n >= 0 && n <= 4, "colsum_jobs: 0 .. 4 jobs");
    ColsumJobs J = {};
    int nj = 0, total = 0;
    for (int i = 0; i < n; ++i) {
        const ColsumJobDesc& d = jobs[i];
        if (d.rows == 0) continue;
        DVAE_REQUIRE(d.out && d.x && d.rows > 0, "colsum_jobs: null pointer");
        DVAE_REQUIRE(d.width > 0 && d.width % 4 == 0 && d.ld_x % 4 == 0 && (uintptr_t)d.x % (d.x_bf16 ? 8 : 16) == 0,
                     "colsum_jobs: bad size or alignment");
        ColsumJob& j = J.j[nj++];
        j.out = d.out; j.x = d.x; j.scale = d.scale; j.ld_x = d.ld_x; j.rows = d.rows; j.w4 = d.width / 4; j.x_bf16 = d.x_bf16;
        const int xb = cdiv(j.w4, 64);
        int chunks = cdiv(d.rows, 32);  // as dvae_segment_colsum_f32: >= 2 workgroups per CU where the rows allow
        const int cap = cdiv(768, xb);
        if (chunks > cap) chunks = cap;
        j.rows_per_chunk = cdiv(d.rows, chunks);
        j.chunks = cdiv(d.rows, j.rows_per_chunk);
        j.xblocks = xb;
        j.first = total;
        total += xb * j.chunks;
    }
    if (nj == 0) return DVAE_OK;
    hipLaunchKernelGGL(colsum_jobs_kernel, dim3(total), dim3(256), 0, s, J);
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}
}  // namespace dvae
extern "C" int dvae_segment_colsum_f32(float* out, int32_t ld_out, const float* x, int32_t ld_x, int32_t rows,
                                       int32_t seg_rows, int32_t width, void* stream) {
    return segment_colsum_any(out, ld_out, x, false, ld_x, rows, seg_rows, width, stream);
}
namespace dvae {
int segment_colsum_h(float* out, int ld_out, const uint16_t* x, int ld_x, int rows, int seg_rows, int width, hipStream_t s) {
    return segment_colsum_any(out, ld_out, x, true, ld_x, rows, seg_rows, width, s);
}
}  // namespace dvae
static int segment_colsum_any(float* out, int32_t ld_out, const void* x, bool x_bf16, int32_t ld_x, int32_t rows,
                              int32_t seg_rows, int32_t width, void* stream) {
    if (rows == 0) return DVAE_OK;
    DVAE_REQUIRE(out && x, "segment_colsum: null pointer");
    DVAE_REQUIRE(width > 0 && width % 4 == 0 && ld_x % 4 == 0 && seg_rows > 0, "segment_colsum: bad size");
    int segs = cdiv(rows, seg_rows);
    int w4 = width / 4;
    // enough row chunks to put >= 2 workgroups on every CU (each adds its partial with atomics)
    int xblocks = cdiv(w4, 64);
    int chunks = cdiv(seg_rows, 32);
    int cap = cdiv(768, xblocks * segs);
    if (chunks > cap) chunks = cap;
    if (chunks < 1) chunks = 1;
    int rpc = cdiv(seg_rows, chunks);
    chunks = cdiv(seg_rows, rpc);
    if (x_bf16)
        hipLaunchKernelGGL(segment_colsum_kernel<true>, dim3(cdiv(w4, 64), chunks, segs), dim3(256), 0, (hipStream_t)stream,
                           out, ld_out, x, ld_x, rows, seg_rows, w4, rpc);
    else
        hipLaunchKernelGGL(segment_colsum_kernel<false>, dim3(cdiv(w4, 64), chunks, segs), dim3(256), 0, (hipStream_t)stream,
                           out, ld_out, x, ld_x, rows, seg_rows, w4, rpc);
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}
