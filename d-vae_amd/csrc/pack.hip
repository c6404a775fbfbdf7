// Weight packing (native state_dict tensors -> zero-padded, MFMA-friendly
// layout) and gradient unpacking (the exact transpose of the same maps).
// Padding is exact: padded GRU lanes stay 0 (n = tanh(0) = 0, h' = z*0), padded
// gate lanes multiply a zero mapper lane (models.py:585-586 keeps the mapper
// bias-free for the same reason).
#include "common.h"
#include "model.h"

namespace dvae {

// One launch moves everything: plain block copies AND the table builds / un-builds are entries of one op
// table (a train step packs once and unpacks twice: 3 launches where there were 7 + 4 + 5).
enum { OP_COPY = 0, OP_BUILD_GI, OP_BUILD_GV, OP_BUILD_BN, OP_UNBUILD_GI, OP_UNBUILD_GV, OP_UNBUILD_BN };
struct CopyOp {
    const float* src;
    float* dst;
    int rows, cols, src_ld, dst_ld;
    int kind;
    // table ops (see the device functions below for each kind's meaning)
    const float *a, *b;
    float *d1, *d2;
    int i0, i1, i2, i3, i4;
};
constexpr int MAX_OPS = 40;  // 40 x 88 bytes: inside the 4 KB kernel-argument segment
struct CopyOps {
    CopyOp op[MAX_OPS];
    int n;
    int accumulate;  // dst += src
};

__device__ __forceinline__ void put(float* d, float v, int acc) { *d = acc ? *d + v : v; }

// gi[t][q*Hp + j] = Wih[q*H + j][t] + bih[q*H + j] + (q < 2 ? bhh[q*H + j] : 0)
//   src = wih, a = bih, b = bhh, dst = gi; i0 = T, i1 = H, i2 = Hp
__device__ __forceinline__ void build_gi(const CopyOp& o, int i) {
    const int T = o.i0, H = o.i1, Hp = o.i2;
    if (i >= T * 3 * H) return;
    int t = i / (3 * H), r = i % (3 * H), q = r / H, j = r % H;
    o.dst[(long)t * 3 * Hp + q * Hp + j] = o.src[(long)r * T + t] + o.a[r] + (q < 2 ? o.b[r] : 0.f);
}
// the transpose: src = d gi, a = d bhn, dst = d wih, d1 = d bih, d2 = d bhh (each nullable)
__device__ __forceinline__ void unbuild_gi(const CopyOp& o, int r, int acc) {
    const int T = o.i0, H = o.i1, Hp = o.i2;
    if (r >= 3 * H) return;
    int q = r / H, j = r % H;
    float s = 0.f;
    for (int t = 0; t < T; ++t) {
        float v = o.src[(long)t * 3 * Hp + q * Hp + j];
        if (o.dst) put(o.dst + (long)r * T + t, v, acc);
        s += v;
    }
    if (o.d1) put(o.d1 + r, s, acc);
    if (o.d2) put(o.d2 + r, q < 2 ? s : o.a[j], acc);
}
// gv[v][j] = (vid ? Wg[j][H+v] : 0) + bg[j];  gv[v][Hp+j] = vid ? Wm[j][H+v] : 0
//   src = wg, a = bg, b = wm, dst = gv; i0 = n, i1 = H, i2 = Hp, i3 = V, i4 = vid
__device__ __forceinline__ void build_gv(const CopyOp& o, int i) {
    const int n = o.i0, H = o.i1, Hp = o.i2, V = o.i3, vid = o.i4;
    if (i >= n * H) return;
    int v = i / H, j = i % H;
    o.dst[(long)v * 2 * Hp + j] = (vid ? o.src[(long)j * V + H + v] : 0.f) + o.a[j];
    o.dst[(long)v * 2 * Hp + Hp + j] = vid ? o.b[(long)j * V + H + v] : 0.f;
}
// src = d gv, dst = d wg, d1 = d bg, d2 = d wm
__device__ __forceinline__ void unbuild_gv(const CopyOp& o, int j, int acc) {
    const int n = o.i0, H = o.i1, Hp = o.i2, V = o.i3, vid = o.i4;
    if (j >= H) return;
    float s = 0.f;
    for (int v = 0; v < n; ++v) {
        float a = o.src[(long)v * 2 * Hp + j];
        s += a;
        if (vid) {
            if (o.dst) put(o.dst + (long)j * V + H + v, a, acc);
            if (o.d2) put(o.d2 + (long)j * V + H + v, o.src[(long)v * 2 * Hp + Hp + j], acc);
        }
    }
    if (o.d1) put(o.d1 + j, s, acc);
}
// BN message table: gated(u) = sigmoid(Wg[:, t_u] + bg) * Wm[:, t_u]  (models.py:1044-1062)
//   src = wg, a = bg, b = wm, dst = gtab, d1 = gtg, d2 = gtm; i0 = T, i1 = H, i2 = Hp
__device__ __forceinline__ void build_bn(const CopyOp& o, int i) {
    const int T = o.i0, H = o.i1, Hp = o.i2;
    if (i >= T * H) return;
    int t = i / H, j = i % H;
    float g = 1.f / (1.f + expf(-(o.src[(long)j * T + t] + o.a[j])));
    float m = o.b[(long)j * T + t];
    o.d1[(long)t * Hp + j] = g;
    o.d2[(long)t * Hp + j] = m;
    o.dst[(long)t * Hp + j] = g * m;
}
// src = d gtab, a = gtg, b = gtm, dst = d wg, d1 = d bg, d2 = d wm
__device__ __forceinline__ void unbuild_bn(const CopyOp& o, int j, int acc) {
    const int T = o.i0, H = o.i1, Hp = o.i2;
    if (j >= H) return;
    float s = 0.f;
    for (int t = 0; t < T; ++t) {
        float d = o.src[(long)t * Hp + j], g = o.a[(long)t * Hp + j], m = o.b[(long)t * Hp + j];
        float da = d * m * g * (1.f - g);
        s += da;
        if (o.dst) put(o.dst + (long)j * T + t, da, acc);
        if (o.d2) put(o.d2 + (long)j * T + t, d * g, acc);
    }
    if (o.d1) put(o.d1 + j, s, acc);
}

// grid (COPY_WGS, ops): a workgroup walks whole rows of its op, its threads the columns of a row
// (coalesced on both sides, no division per element; H = 501 leaves the source rows unaligned, so
// the accesses stay one float wide); the table ops grid-stride over their (few) elements
constexpr int COPY_WGS = 256;
__global__ __launch_bounds__(256) void copy_ops_kernel(CopyOps ops) {
    const CopyOp& o = ops.op[blockIdx.y];
    const int acc = ops.accumulate;
    if (o.kind != OP_COPY) {  // (uniform over the workgroup)
        const int total = o.rows;  // number of work items of the table op
        for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += COPY_WGS * 256) {
            switch (o.kind) {
            case OP_BUILD_GI: build_gi(o, i); break;
            case OP_BUILD_GV: build_gv(o, i); break;
            case OP_BUILD_BN: build_bn(o, i); break;
            case OP_UNBUILD_GI: unbuild_gi(o, i, acc); break;
            case OP_UNBUILD_GV: unbuild_gv(o, i, acc); break;
            default: unbuild_bn(o, i, acc); break;
            }
        }
        return;
    }
    const int rows = o.rows, cols = o.cols;
    if (rows == 1) {  // a bias vector: spread its columns over the workgroups
        for (int c = blockIdx.x * 256 + threadIdx.x; c < cols; c += COPY_WGS * 256) {
            const float v = o.src[c];
            o.dst[c] = acc ? o.dst[c] + v : v;
        }
        return;
    }
    // four rows per trip: their (scalar, unaligned-row) loads are independent and go out together -- one row per trip
    // left every trip waiting for its own load (pack of hs = 501: 18 -> 11 us; the batch-32 step runs three of these)
    for (int r = blockIdx.x; r < rows; r += 4 * COPY_WGS) {
        for (int c = threadIdx.x; c < cols; c += 256) {
            float v[4], d[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int rr = r + k * COPY_WGS;
                v[k] = rr < rows ? o.src[(long)rr * o.src_ld + c] : 0.f;
                d[k] = (acc && rr < rows) ? o.dst[(long)rr * o.dst_ld + c] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int rr = r + k * COPY_WGS;
                if (rr < rows) o.dst[(long)rr * o.dst_ld + c] = d[k] + v[k];
            }
        }
    }
}

// native <-> packed block copies.  `nat[i]` may be null (absent parameter).
// to_packed: native -> packed ; else packed -> native.
static int block_ops(const Dims& d, const PackedLayout& L, float* const* nat, float* pk, bool to_packed, CopyOps& ops,
                     const bool* want /* nullable: which params to touch */) {
    ops.n = 0;
    ops.accumulate = 0;
    const int H = d.H, Hp = d.Hp;
    auto add = [&](int pid, size_t nat_off, int nat_ld, size_t pk_off, int pk_ld, int rows, int cols) {
        if (!nat[pid] || (want && !want[pid])) return;
        CopyOp o = {};
        if (to_packed) { o.src = nat[pid] + nat_off; o.dst = pk + pk_off; o.src_ld = nat_ld; o.dst_ld = pk_ld; }
        else { o.src = pk + pk_off; o.dst = nat[pid] + nat_off; o.src_ld = pk_ld; o.dst_ld = nat_ld; }
        o.rows = rows; o.cols = cols;
        ops.op[ops.n++] = o;
    };
    const int gru_w[3] = {P_EF_WHH, P_EB_WHH, P_D_WHH};
    for (int g = 0; g < 3; ++g)
        for (int q = 0; q < 3; ++q) add(gru_w[g], (size_t)q * H * H, H, L.whh[g] + (size_t)q * Hp * Hp, Hp, H, H);
    // b_hh of the n gate stays a separate vector (it sits inside r * (W_hn h + b_hn)); its
    // gradient is routed by unbuild_gi_kernel, so the copy exists in the packing direction only
    if (to_packed)
        for (int g = 0; g < 3; ++g) add(gru_w[g] + 2, (size_t)2 * H, H, L.bhn[g], H, 1, H);
    if (!d.bn) {
        const int gw[2] = {P_GF_W, P_GB_W}, mw[2] = {P_MF_W, P_MB_W};
        for (int k = 0; k < 2; ++k) {
            add(gw[k], 0, d.V, L.wgm[k], Hp, H, H);
            add(mw[k], 0, d.V, L.wgm[k] + (size_t)Hp * Hp, Hp, H, H);
        }
    }
    add(P_FC1_W, 0, H, L.fc12, Hp, d.nz, H);
    add(P_FC2_W, 0, H, L.fc12 + (size_t)d.NZp * Hp, Hp, d.nz, H);
    add(P_FC1_B, 0, d.nz, L.b12, d.nz, 1, d.nz);
    add(P_FC2_B, 0, d.nz, L.b12 + d.NZp, d.nz, 1, d.nz);
    add(P_FC3_W, 0, d.nz, L.fc3, d.NZp, H, d.nz);
    add(P_FC3_B, 0, H, L.b3, H, 1, H);
    add(P_AV0_W, 0, H, L.wv1, Hp, 2 * H, H);
    add(P_AV0_B, 0, 2 * H, L.bv1, 2 * H, 1, 2 * H);
    add(P_AV2_W, 0, 2 * H, L.wv2, d.HVp, d.T, 2 * H);
    add(P_AV2_B, 0, d.T, L.bv2, d.T, 1, d.T);
    if (d.bn) {
        add(P_AE0_W, 0, 3 * H, L.w1a, Hp, H, H);
        add(P_AE0_W, H, 3 * H, L.w1b, Hp, H, H);
        add(P_AE0_W, 2 * H, 3 * H, L.w1c, Hp, H, H);
        add(P_AE0_B, 0, H, L.be1, H, 1, H);
        add(P_AE2_W, 0, H, L.we2, H, 1, H);
    } else {
        add(P_AE0_W, 0, 2 * H, L.w1a, Hp, 4 * H, H);
        add(P_AE0_W, H, 2 * H, L.w1b, Hp, 4 * H, H);
        add(P_AE0_B, 0, 4 * H, L.be1, 4 * H, 1, 4 * H);
        add(P_AE2_W, 0, 4 * H, L.we2, 4 * H, 1, 4 * H);
    }
    add(P_AE2_B, 0, 1, L.be2, 1, 1, 1);
    if (d.bidir) {
        add(P_HG_W, 0, 2 * H, L.wu, 2 * Hp, H, H);
        add(P_HG_W, H, 2 * H, L.wu + Hp, 2 * Hp, H, H);
        add(P_HG_B, 0, H, L.bu, H, 1, H);
    }
    if (ops.n > MAX_OPS - 5) return set_err(DVAE_ERR_SIZE, "copy-op table overflow");  // (5 table ops may follow)
    return DVAE_OK;
}
static void table_op(CopyOps& ops, int kind, int items, const float* src, const float* a, const float* b, float* dst,
                     float* d1, float* d2, int i0, int i1, int i2, int i3 = 0, int i4 = 0) {
    CopyOp o = {};
    o.kind = kind; o.rows = items; o.src = src; o.a = a; o.b = b; o.dst = dst; o.d1 = d1; o.d2 = d2;
    o.i0 = i0; o.i1 = i1; o.i2 = i2; o.i3 = i3; o.i4 = i4;
    ops.op[ops.n++] = o;
}

static int run_ops(const CopyOps& ops, hipStream_t s) {
    if (ops.n == 0) return DVAE_OK;
    hipLaunchKernelGGL(copy_ops_kernel, dim3(COPY_WGS, ops.n), dim3(256), 0, s, ops);
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}

int check_params(const Dims& d, const float* const* p) {
    const int need_always[] = {P_EF_WIH, P_EF_WHH, P_EF_BIH, P_EF_BHH, P_FC1_W, P_FC1_B, P_FC2_W, P_FC2_B,
                               P_D_WIH, P_D_WHH, P_D_BIH, P_D_BHH, P_FC3_W, P_FC3_B, P_AV0_W, P_AV0_B, P_AV2_W,
                               P_AV2_B, P_AE0_W, P_AE0_B, P_AE2_W, P_AE2_B, P_GF_W, P_GF_B, P_MF_W};
    for (int id : need_always) DVAE_REQUIRE(p[id] != nullptr, "parameter %d is required but null", id);
    if (d.bidir) {
        const int need_bi[] = {P_EB_WIH, P_EB_WHH, P_EB_BIH, P_EB_BHH, P_GB_W, P_GB_B, P_MB_W, P_HG_W, P_HG_B};
        for (int id : need_bi) DVAE_REQUIRE(p[id] != nullptr, "parameter %d is required for bidirectional but null", id);
    }
    return DVAE_OK;
}

// dst [cols][rows] bf16 = src [rows][cols] fp32, transposed (32 x 32 tiles through LDS)
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst,
                                                             int rows, int cols) {
    __shared__ float t[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + threadIdx.x;
        t[j][threadIdx.x] = (r < rows && c < cols) ? src[(size_t)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + threadIdx.x;
        if (c < cols && r < rows) {
            const __bf16 b = (__bf16)t[threadIdx.x][j];
            dst[(size_t)c * rows + r] = __builtin_bit_cast(uint16_t, b);
        }
    }
}

}  // namespace dvae

using namespace dvae;

extern "C" int64_t dvae_packed_floats(const dvae_model_desc* m) {
    Dims d;
    if (make_dims(m, d) != 0) return -1;
    return (int64_t)make_layout(d).alloc;  // bf16 mode: the fp32 part + the bf16 operand copies behind it (model.h)
}

static int pack_impl(const dvae_model_desc* m, const float* const* params, float* packed, bool zero_first, void* stream) {
    Dims d;
    DVAE_TRY(make_dims(m, d));
    DVAE_REQUIRE(params && packed, "dvae_pack_weights: null pointer");
    DVAE_TRY(check_params(d, params));
    hipStream_t s = (hipStream_t)stream;
    PackedLayout L = make_layout(d);
    if (zero_first) DVAE_HIP(hipMemsetAsync(packed, 0, L.total * sizeof(float), s));
    CopyOps ops;
    DVAE_TRY(block_ops(d, L, const_cast<float* const*>(reinterpret_cast<const float* const*>(params)), packed, true, ops, nullptr));
    const int wih[3] = {P_EF_WIH, P_EB_WIH, P_D_WIH};
    for (int g = 0; g < 3; ++g) {
        if (!params[wih[g]]) continue;
        table_op(ops, OP_BUILD_GI, d.T * 3 * d.H, params[wih[g]], params[wih[g] + 2], params[wih[g] + 3], packed + L.gi[g],
                 nullptr, nullptr, d.T, d.H, d.Hp);
    }
    const int gw[2] = {P_GF_W, P_GB_W}, gb[2] = {P_GF_B, P_GB_B}, mw[2] = {P_MF_W, P_MB_W};
    for (int k = 0; k < 2; ++k) {
        if (!params[gw[k]] || !params[gb[k]] || !params[mw[k]]) continue;
        if (d.bn)
            table_op(ops, OP_BUILD_BN, d.T * d.H, params[gw[k]], params[gb[k]], params[mw[k]], packed + L.gtab[k],
                     packed + L.gtg[k], packed + L.gtm[k], d.T, d.H, d.Hp);
        else
            table_op(ops, OP_BUILD_GV, d.n * d.H, params[gw[k]], params[gb[k]], params[mw[k]], packed + L.gv[k], nullptr,
                     nullptr, d.n, d.H, d.Hp, d.V, d.vid);
    }
    DVAE_TRY(run_ops(ops, s));
    if (d.bf16) {  // the bf16 operand copies (model.h: PackedLayout::half / halfT), rebuilt with every (re)pack
        uint16_t* half = reinterpret_cast<uint16_t*>(packed + L.half);
        uint16_t* halfT = reinterpret_cast<uint16_t*>(packed + L.halfT);
        DVAE_TRY(dvae_f32_to_bf16(packed, half, (int64_t)L.total, s));
        for (int i = 0; i < L.nmats; ++i) {
            const PackedLayout::Mat& mt = L.mats[i];
            const dim3 grid(cdiv(mt.cols, 32), cdiv(mt.rows, 32)), block(32, 8);
            hipLaunchKernelGGL(transpose_bf16_kernel, grid, block, 0, s, packed + mt.off, halfT + mt.toff, mt.rows, mt.cols);
            DVAE_LAUNCH_CHECK();  // per launch: every launch is counted and checked (dvae_debug_launch_count)
        }
    }
    return DVAE_OK;
}

extern "C" int dvae_pack_weights(const dvae_model_desc* m, const float* const* params, float* packed, void* stream) {
    return pack_impl(m, params, packed, true, stream);
}
extern "C" int dvae_repack_weights(const dvae_model_desc* m, const float* const* params, float* packed, void* stream) {
    return pack_impl(m, params, packed, false, stream);
}

// packed gradient buffer -> native gradient tensors.  `want[i]` != 0 selects the
// parameters this call produces (others untouched); grads[i] may be null.
// accumulate != 0: grads[i] += (PyTorch's .grad accumulation, done in place).
extern "C" int dvae_unpack_grads(const dvae_model_desc* m, const float* packed_w, const float* packed_g,
                                 float* const* grads, const uint8_t* want, int32_t accumulate, void* stream) {
    Dims d;
    DVAE_TRY(make_dims(m, d));
    DVAE_REQUIRE(packed_g && grads && want, "dvae_unpack_grads: null pointer");
    hipStream_t s = (hipStream_t)stream;
    PackedLayout L = make_layout(d);
    bool w[P_COUNT];
    for (int i = 0; i < P_COUNT; ++i) w[i] = want[i] != 0 && grads[i] != nullptr;
    CopyOps ops;
    DVAE_TRY(block_ops(d, L, grads, const_cast<float*>(packed_g), false, ops, w));
    ops.accumulate = accumulate != 0;
    const int wih[3] = {P_EF_WIH, P_EB_WIH, P_D_WIH};
    for (int g = 0; g < 3; ++g) {
        int a = wih[g];
        if (!(w[a] || w[a + 2] || w[a + 3])) continue;
        table_op(ops, OP_UNBUILD_GI, 3 * d.H, packed_g + L.gi[g], packed_g + L.bhn[g], nullptr, w[a] ? grads[a] : nullptr,
                 w[a + 2] ? grads[a + 2] : nullptr, w[a + 3] ? grads[a + 3] : nullptr, d.T, d.H, d.Hp);
    }
    const int gw[2] = {P_GF_W, P_GB_W}, gb[2] = {P_GF_B, P_GB_B}, mw[2] = {P_MF_W, P_MB_W};
    for (int k = 0; k < 2; ++k) {
        if (!(w[gw[k]] || w[gb[k]] || w[mw[k]])) continue;
        float* dwg = w[gw[k]] ? grads[gw[k]] : nullptr;
        float* dbg = w[gb[k]] ? grads[gb[k]] : nullptr;
        float* dwm = w[mw[k]] ? grads[mw[k]] : nullptr;
        if (d.bn) {
            DVAE_REQUIRE(packed_w != nullptr, "dvae_unpack_grads: BN needs the packed weights (gate tables)");
            table_op(ops, OP_UNBUILD_BN, d.H, packed_g + L.gtab[k], packed_w + L.gtg[k], packed_w + L.gtm[k], dwg, dbg, dwm,
                     d.T, d.H, d.Hp);
        } else {
            table_op(ops, OP_UNBUILD_GV, d.H, packed_g + L.gv[k], nullptr, nullptr, dwg, dbg, dwm, d.n, d.H, d.Hp, d.V, d.vid);
        }
    }
    return run_ops(ops, s);
}
