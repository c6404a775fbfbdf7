// Pieces shared by the level kernels (chain.hip) and the persistent small-batch chains (chain_persist.hip):
// the kernel parameter block, the pointwise tails, and the write-through hand-off primitives.
#pragma once
#include "gemm_common.h"
#include "model.h"

namespace dvae {

struct ChainParams {
    const float* A; int lda;   // [M][K] K-contiguous operand (states, or dY of the backward products)
    const float* W; int ldw;   // forward: [NGATE * Hp][K] K-contiguous;  backward: [K][Hp]
    int M, Hp, K;
    int tiles_x, tiles_y;      // Hp / 32, ceil(M / 64)
    // GRU tail (gates / ghn are outputs of the forward tail, inputs of the backward one)
    const float* gi_tab; const int* types; const float* bhn; const float* hprev; int ld_hprev;
    float* gates; float* ghn; float* hout; int ld_hout;
    // gate / mapper tail
    const float* gv; float* gm; float* msg; int ld_msg; int cur_row0;
    float* next_agg; int ld_agg; const int* next_off; const int* idx;
    // backward: dX in/out and the outputs of the GRU backward tail
    float* C; int ldc;
    float* dgh; float* dgin; float* dhprev; int ld_dhprev;
    // backward, dX += tail only (optional: dgm_out != NULL): the gate/mapper pointwise backward of the
    // level processed NEXT, whose d msg rows are gather-sums of dX rows (this launch's and finished ones)
    const float* gsrc; const int* goff; const int* gidx;    // dX of all levels (row pitch ldc), CSR of consumer slots
    const float* gsrc2; const int* goff2; const int* gidx2; // optional second list
    const float* gm_in; float* dgm_out;
};

__device__ __forceinline__ float4 cld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void cst4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// Write-through (sc1) 16-byte accesses for data that ANOTHER workgroup of the same launch reads or wrote (the
// persistent chain kernel below): the store leaves L2 at once, the load bypasses this CU's L1, so the hand-off needs
// no release / acquire fence (cdna_hip_programming.md guideline 16, form R1).  aux 16 = sc1.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const float* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0xFFFFFFF0u, 0x00020000);
}
__device__ __forceinline__ float4 ld4_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void st4_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float4 f) {
    u32x4 v;
    v.x = __float_as_uint(f.x); v.y = __float_as_uint(f.y); v.z = __float_as_uint(f.z); v.w = __float_as_uint(f.w);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, byte_off, 0, 16);
}

// Diagnostic builds only (tools/chain_ablate.sh; results are wrong, only the time matters): CHAIN_ABL bit 0 drops the
// pointwise tail (the K-slice exchange stays), bit 1 the MFMAs, bit 2 runs one K step instead of all, bit 3 drops the
// GRU forward tail's gate / ghn stores (4 of its 5 output arrays).
#ifndef CHAIN_ABL
#define CHAIN_ABL 0
#endif
#ifndef CHAIN_PF
#define CHAIN_PF 2  // register-staged tiles in flight in the unrolled kernels (1, 2, 4 or 8; measured: no effect beyond 1)
#endif

// ---------------------------------------------------------------------------------- tails
// (Loading a tail's operands -- table rows, biases, h_prev, stored gates, dX -- before the K loop
// was measured: no change per kernel, and the extra live registers (GRU forward 146 -> 202) keep the
// side stream's kernels off the CUs the chain occupies: loss forward 0.86 -> 0.97 ms.  Not done.)
// GRUCell tail for 4 consecutive hidden units of one row (same arithmetic as gru_fwd_kernel)
template <bool SC1>
__device__ __forceinline__ void gru_tail(const ChainParams& p, int row, int col, float4 ghr, float4 ghz, float4 ghn) {
    const float* gi = p.gi_tab + (long)p.types[row] * 3 * p.Hp;
    const float4 gir = cld4(gi + col), giz = cld4(gi + p.Hp + col), gin = cld4(gi + 2 * p.Hp + col);
    const float4 bn = cld4(p.bhn + col);
    const float4 hp = SC1 ? ld4_sc1(rsrc_of(p.hprev), (unsigned)((row * p.ld_hprev + col) * 4))
                          : cld4(p.hprev + (long)row * p.ld_hprev + col);
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
    if (!(CHAIN_ABL & 8)) {  // (diagnostic: bit 3 keeps only the state store; the values stay alive through Hn's inputs)
        cst4(g + col, R);
        cst4(g + p.Hp + col, Z);
        cst4(g + 2 * p.Hp + col, N);
        cst4(p.ghn + (long)row * p.Hp + col, GN);
    }
    if (SC1) st4_sc1(rsrc_of(p.hout), (unsigned)((row * p.ld_hout + col) * 4), Hn);  // read by the row tile's other workgroups
    else cst4(p.hout + (long)row * p.ld_hout + col, Hn);
}

// gate/mapper tail + the next level's predecessor sum (same arithmetic and summation order as
// gated_fwd_gather_kernel)
template <bool SC1>
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
        if (SC1) st4_sc1(rsrc_of(p.next_agg), (unsigned)((row * p.ld_agg + col) * 4), acc);
        else cst4(p.next_agg + (long)row * p.ld_agg + col, acc);
    }
}

// GRUCell pointwise backward for 4 consecutive hidden units of one row (gru_bwd_kernel's arithmetic)
// SC1 (persistent chains): d gh is the next phase's A operand, read by the row tile's other workgroups
template <bool SC1 = false>
__device__ __forceinline__ void gru_bwd_tail(const ChainParams& p, int row, int col, float4 D) {
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
    if (SC1) {
        const __amdgpu_buffer_rsrc_t r = rsrc_of(p.dgh);
        const unsigned o = (unsigned)((row * 3 * p.Hp + col) * 4);
        st4_sc1(r, o, dR);
        st4_sc1(r, o + (unsigned)p.Hp * 4u, dZ);
        st4_sc1(r, o + (unsigned)p.Hp * 8u, dGN);
    } else {
        float* o = p.dgh + (long)row * 3 * p.Hp;
        cst4(o + col, dR);
        cst4(o + p.Hp + col, dZ);
        cst4(o + 2 * p.Hp + col, dGN);
    }
    cst4(p.dgin + (long)row * p.Hp + col, dPN);
    cst4(p.dhprev + (long)row * p.ld_dhprev + col, dHP);
}

// d msg of the level processed next = gather-sum of its consumers' d agg rows (gated_bwd_gather_kernel's
// arithmetic and summation order), consumed by the gate/mapper pointwise backward.  The consumers of
// row r are slots of the same graph: the one this lane has just finished (slot cur_row0 + r, taken
// from registers) and slots of levels finished by earlier launches.
// SC1 (persistent chains): d gm is the next phase's A operand, read by the row tile's other workgroups
template <bool SC1 = false>
__device__ __forceinline__ void gather_bwd_tail(const ChainParams& p, int row, int col, float4 own) {
    float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = p.goff[row]; k < p.goff[row + 1]; ++k) {
        const int s = p.gidx[k];
        const float4 v = s == p.cur_row0 + row ? own : cld4(p.gsrc + (long)s * p.ldc + col);
        d.x += v.x; d.y += v.y; d.z += v.z; d.w += v.w;
    }
    if (p.goff2)
        for (int k = p.goff2[row]; k < p.goff2[row + 1]; ++k) {
            const float4 v = cld4(p.gsrc2 + (long)p.gidx2[k] * p.ldc + col);
            d.x += v.x; d.y += v.y; d.z += v.z; d.w += v.w;
        }
    const float* q = p.gm_in + (long)row * 2 * p.Hp;
    const float4 g = cld4(q + col), m = cld4(q + p.Hp + col);
    const float4 dg = make_float4(d.x * m.x * g.x * (1.f - g.x), d.y * m.y * g.y * (1.f - g.y),
                                  d.z * m.z * g.z * (1.f - g.z), d.w * m.w * g.w * (1.f - g.w));
    const float4 dm = make_float4(d.x * g.x, d.y * g.y, d.z * g.z, d.w * g.w);
    if (SC1) {
        const __amdgpu_buffer_rsrc_t r = rsrc_of(p.dgm_out);
        const unsigned o = (unsigned)((row * 2 * p.Hp + col) * 4);
        st4_sc1(r, o, dg);
        st4_sc1(r, o + (unsigned)p.Hp * 4u, dm);
    } else {
        float* o = p.dgm_out + (long)row * 2 * p.Hp;
        cst4(o + col, dg);
        cst4(o + p.Hp + col, dm);
    }
}

// GRUCell tail of a pass's first level (no predecessors: h_prev = 0, the hidden product is 0; gru_fwd_first_kernel's
// arithmetic) for 4 consecutive hidden units of one row; also leaves the level's rows of agg at zero for the backward.
// `agg_zero` = the level's rows of agg.
template <bool SC1>
__device__ __forceinline__ void gru_first_tail(const ChainParams& p, float* agg_zero, int row, int col) {
    const float* gi = p.gi_tab + (long)p.types[row] * 3 * p.Hp;
    const float4 gir = cld4(gi + col), giz = cld4(gi + p.Hp + col), gin = cld4(gi + 2 * p.Hp + col);
    const float4 bn = cld4(p.bhn + col);
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
    float* g = p.gates + (long)row * 3 * p.Hp;
    cst4(g + col, R);
    cst4(g + p.Hp + col, Z);
    cst4(g + 2 * p.Hp + col, N);
    cst4(p.ghn + (long)row * p.Hp + col, bn);
    cst4(agg_zero + (long)row * p.Hp + col, make_float4(0.f, 0.f, 0.f, 0.f));
    if (SC1) st4_sc1(rsrc_of(p.hout), (unsigned)((row * p.ld_hout + col) * 4), Hn);
    else cst4(p.hout + (long)row * p.ld_hout + col, Hn);
}

// ---------------------------------------------------------------------------------- 16 x 16 tiles
// One level operation of one 16-row x 16-hidden-unit x all-gates tile: the body of chain_tiny_kernel (chain.hip), also
// called phase by phase by the persistent small-batch chains (chain_persist.hip; SC1: the A operand and the tail's
// hand-off arrays are accessed write-through).  KS waves = KS interleaved K slices; `xch` holds KS * NGATE * 64 float4.
// Every wave leaves through the end of the function (no early return): a persistent caller's barriers follow.
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int NGATE, bool WKM, int TAIL, int KS, bool SC1>
__device__ __forceinline__ void chain_tiny_body(const ChainParams& p, float4* xch, int bid) {
    const int tid = threadIdx.x, lane = tid & 63, kq = tid >> 6;
    const int n0 = (bid % p.tiles_x) * 16, m0 = (bid / p.tiles_x) * 16;
    const int li = lane & 15, kg = lane >> 4;
    int row = m0 + li;
    row = row < p.M ? row : p.M - 1;  // clamped rows are computed, never stored
    const float* pa = p.A + (long)row * p.lda;
    const __amdgpu_buffer_rsrc_t rsa = rsrc_of(p.A);
    const unsigned oa = (unsigned)(row * p.lda * 4);
    const int col = n0 + li;
    f32x4v acc[NGATE];
#pragma unroll
    for (int i = 0; i < NGATE; ++i) acc[i] = f32x4v{0.f, 0.f, 0.f, 0.f};
    struct Frag { float4 a; float4 w[NGATE]; };
    const int ns = p.K / 16;  // 16-k steps; wave kq takes steps kq, kq + KS, ...
    auto load = [&](Frag& f, int st) {
        const bool ok = st < ns;
        const int k = (ok ? 16 * st : 0) + 4 * kg;  // past the end: a valid address, the A part is zero
        if (SC1) {
            const float4 t = ld4_sc1(rsa, oa + (unsigned)k * 4u);
            f.a = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            f.a = ok ? cld4(pa + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < NGATE; ++i) {
            if (!WKM) {
                f.w[i] = cld4(p.W + (long)(i * p.Hp + col) * p.ldw + k);
            } else {
                const float* w = p.W + (long)k * p.ldw + col;
                f.w[i] = make_float4(w[0], w[p.ldw], w[2 * (long)p.ldw], w[3 * (long)p.ldw]);
            }
        }
    };
    auto mma = [&](const Frag& f) {
#pragma unroll
        for (int i = 0; i < NGATE; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.w[i].x, f.a.x, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.w[i].y, f.a.y, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.w[i].z, f.a.z, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.w[i].w, f.a.w, acc[i], 0, 0, 0);
        }
    };
    // two steps in flight ahead of the one being contracted (fixed register sets, as chain_small_kernel)
    Frag f0, f1, f2;
    load(f0, kq);
    load(f1, kq + KS);
    for (int st = kq; st < ns; st += 3 * KS) {
        load(f2, st + 2 * KS);
        mma(f0);
        if (st + KS < ns) {
            load(f0, st + 3 * KS);
            mma(f1);
        }
        if (st + 2 * KS < ns) {
            load(f1, st + 4 * KS);
            mma(f2);
        }
    }
#pragma unroll
    for (int i = 0; i < NGATE; ++i) xch[(kq * NGATE + i) * 64 + lane] = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
    __syncthreads();
    const int orow = m0 + li;
    if (kq == 0 && orow < p.M) {
        float4 v[NGATE];
#pragma unroll
        for (int i = 0; i < NGATE; ++i) {
            float4 sum = xch[i * 64 + lane];
#pragma unroll
            for (int k = 1; k < KS; ++k) {  // K slices in order
                const float4 t = xch[(k * NGATE + i) * 64 + lane];
                sum.x += t.x; sum.y += t.y; sum.z += t.z; sum.w += t.w;
            }
            v[i] = sum;
        }
        const int ocol = n0 + 4 * kg;  // D[i = 4 kg + r][j = li]: 4 consecutive hidden units of row li
        if (TAIL == 0) {
            gru_tail<SC1>(p, orow, ocol, v[0], v[NGATE > 1 ? 1 : 0], v[NGATE - 1]);
        } else if (TAIL == 1) {
            gate_tail<SC1>(p, orow, ocol, v[0], v[NGATE - 1]);
        } else {
            // (dX in/out is this tile's own: written before the launch, or by this very lane in an earlier phase)
            float* c = p.C + (long)orow * p.ldc + ocol;
            const float4 c0 = cld4(c);
            const float4 d = make_float4(c0.x + v[0].x, c0.y + v[0].y, c0.z + v[0].z, c0.w + v[0].w);
            cst4(c, d);
            if (TAIL == 3) gru_bwd_tail<SC1>(p, orow, ocol, d);
            if (TAIL == 2 && p.dgm_out) gather_bwd_tail<SC1>(p, orow, ocol, d);
        }
    }
}

template <int SLEEP = 4>
__device__ __forceinline__ bool chain_wait(unsigned* cnt, unsigned target, unsigned* status) {
    // ONE lane polls (relaxed, agent scope = sc1 load); bounded
    bool ok = true;
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(SLEEP);
            // ~1 us per poll: gives up after a few seconds, or within 64 polls of another workgroup giving up
            if (++spins > (1u << 22) ||
                ((spins & 63) == 0 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
        }
    }
    return __syncthreads_and(ok ? 1 : 0) != 0;  // the barrier between the poll and every load of the handed-off data
}
__device__ __forceinline__ void chain_signal(unsigned* cnt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its write-through stores
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace dvae
