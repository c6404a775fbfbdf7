// Persistent, WEIGHT-STATIONARY state chains for SMALL batches (batch 32 is the reference's own recipe, README.md:37 /
// train.py --batch-size 32): a whole topologically ordered pass -- the loops of models.py:675-685 (_propagate_from) and
// :816-846 (the teacher-forced decoder's _update_v chain), forward or backward -- as ONE launch.
//
// Why.  At 32 rows a level operation is a 16 x 16-tile kernel (chain_tiny_kernel, chain.hip) that runs 6-10 us for
// ~0.3 us of matrix-unit work: a chain of dependent memory latencies -- first weight tile, K loop two tiles deep, index
// -> table row -> tail operands, stores -- paid again by every one of the 58 level launches of a batch-32 step
// (profiles/r03_kstats_cfg1_*.txt).  A first persistent version that only removed the launch boundaries (same body,
// phase by phase) measured the same 7.3 us per phase.  Here the tiles stay resident AND keep what does not change:
//   * a workgroup (16 rows x 16 hidden units x all gates, 8 waves = 8 interleaved K slices) holds its tiles of BOTH
//     weights of its pass (W_hh and [Wg | Wm]; backward: their K-major twins) in REGISTERS, in MFMA-fragment layout, for
//     the whole pass: 80 VGPRs per lane at Hp = 512, loaded once instead of 160 KB per level;
//   * every tail operand that does not depend on the phase's product -- type -> input-table row, vertex-id row, stored
//     gates, the CSR gather of messages / gradients of finished levels -- is requested BEFORE the wait for the
//     previous phase, so its latency hides behind the hand-off;
//   * what is left on the critical path of a phase: wait -> one round of A-fragment loads -> 16..48 MFMAs per wave ->
//     the K slices meet in LDS -> tail arithmetic -> write-through stores drain -> signal.
//
// Hand-off.  Graphs are independent and level == vertex index keeps a graph in the same 16-row tile at every level, so
// only the Hp / 16 column-tile workgroups of one ROW TILE exchange data.  One monotonic counter per (pass, row tile) on
// a 128-byte line of its own: a workgroup adds 1 after each phase and waits for tiles_x * q before phase q.  Form:
// cdna_hip_programming.md guideline 16 R1 / MI355X_MICROARCH.md "Valid forms", first table row -- payload stored
// write-through (16-byte sc1 stores), every storing wave drains (s_waitcnt vmcnt(0)), workgroup barrier, ONE lane's
// agent-scope atomic add; consumer: one lane polls with sc1 loads, workgroup barrier, then sc1 loads only.  Data a lane
// wrote itself earlier in the launch is re-read with sc1 loads too (L1 bypassed, the XCD's L2 holds the stores).
// Placement-agnostic: correctness does not depend on which XCD or CU a workgroup lands on.
//
// Residency: all workgroups must be co-resident (tiny_chain_fits: TWICE the grid must fit the occupancy query's answer);
// every spin is bounded: on a timeout the workgroup raises the status word, adds NaN to *poison (a gradient / loss
// accumulator of the caller, so the step's results turn NaN instead of stale) and leaves; the others follow.
// The two directions of a bidirectional encoder are two sets of workgroups of one launch.
#include "chain_common.h"

namespace dvae {

__device__ __forceinline__ void tiny_fail(const TinyChainArgs& a) {
    if (threadIdx.x == 0 && a.poison) atomicAdd(a.poison, __builtin_nanf(""));
}
// Signal of a phase whose hand-off stores were all issued by wave 0 (the tail wave): that wave drains its write-through
// stores and its lane 0 adds to the counter -- no workgroup barrier (it signals for no other wave); the next phase's
// chain_wait is the barrier that keeps the other waves off the LDS exchange buffer until wave 0 has read it.
// Call with the hand-off stores issued and the phase's plain stores (read only by later kernels, or by this lane after
// the next drain) NOT yet issued: they then travel under the next phase's wait.
__device__ __forceinline__ void tail_signal(unsigned* cnt, int kq, int lane) {
    if (kq == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void f4add(float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
__device__ __forceinline__ float4 ld4_l2(const float* base, long elem) {  // L1-bypassing load of data written in this launch
    return ld4_sc1(rsrc_of(base), (unsigned)(elem * 4));
}

// K slices: wave kq owns the 16-k steps kq, kq + 8, ... (NSTEP of them; steps past the end contribute zeros)
template <int NSTEP>
__device__ __forceinline__ void load_a_frags(float4 (&fa)[NSTEP], const float* A, int lda, int row, int ns, int kq, int kg) {
    const __amdgpu_buffer_rsrc_t rs = rsrc_of(A);
#pragma unroll
    for (int j = 0; j < NSTEP; ++j) {
        const int s = kq + 8 * j;
        const bool ok = s < ns;
        const float4 t = ld4_sc1(rs, (unsigned)((row * lda + (ok ? 16 * s : 0) + 4 * kg) * 4));
        fa[j] = ok ? t : f4zero();
    }
}
// forward weights [NG * Hp][K] K-contiguous: lane (col li, kg) holds W[g * Hp + col][16 s + 4 kg .. + 3]
template <int NSTEP, int NG>
__device__ __forceinline__ void load_w_kcontig(float4 (&w)[NSTEP][NG], const float* W, int Hp, int ldw, int col, int ns, int kq, int kg) {
#pragma unroll
    for (int j = 0; j < NSTEP; ++j) {
        const int s = kq + 8 * j;
        const bool ok = s < ns;
#pragma unroll
        for (int g = 0; g < NG; ++g)
            w[j][g] = ok ? cld4(W + (long)(g * Hp + col) * ldw + 16 * s + 4 * kg) : f4zero();
    }
}
// backward weights [K][Hp] K-major: lane (col li, kg) holds W[16 s + 4 kg + {0..3}][col]
template <int NSTEP>
__device__ __forceinline__ void load_w_kmajor(float4 (&w)[NSTEP][1], const float* W, int ldw, int col, int ns, int kq, int kg) {
#pragma unroll
    for (int j = 0; j < NSTEP; ++j) {
        const int s = kq + 8 * j;
        if (s < ns) {
            const float* q = W + (long)(16 * s + 4 * kg) * ldw + col;
            w[j][0] = make_float4(q[0], q[ldw], q[2 * (long)ldw], q[3 * (long)ldw]);
        } else {
            w[j][0] = f4zero();
        }
    }
}
template <int NSTEP, int NG>
__device__ __forceinline__ void contract(f32x4v (&acc)[NG], const float4 (&w)[NSTEP][NG], const float4 (&fa)[NSTEP]) {
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NSTEP; ++j)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j][g].x, fa[j].x, acc[g], 0, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j][g].y, fa[j].y, acc[g], 0, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j][g].z, fa[j].z, acc[g], 0, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j][g].w, fa[j].w, acc[g], 0, 0, 0);
        }
}
// the 8 slice sums meet in LDS; returns (to wave 0) the sums in slice order
template <int NG>
__device__ __forceinline__ void slices_to_lds(float4* xch, const f32x4v (&acc)[NG], int kq, int lane) {
#pragma unroll
    for (int g = 0; g < NG; ++g) xch[(kq * NG + g) * 64 + lane] = make_float4(acc[g][0], acc[g][1], acc[g][2], acc[g][3]);
    __syncthreads();
}
template <int NG>
__device__ __forceinline__ void slices_sum(const float4* xch, float4 (&v)[NG], int lane) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        float4 s = xch[g * 64 + lane];
#pragma unroll
        for (int k = 1; k < 8; ++k) f4add(s, xch[(k * NG + g) * 64 + lane]);
        v[g] = s;
    }
}

// ---------------------------------------------------------------------------------- forward pass
// NS = 16-k steps per wave of an Hp-deep product: Hp <= 128 NS
template <int NS>
__global__ __launch_bounds__(512) void tiny_chain_fwd_kernel(TinyChainArgs a) {
    __shared__ float4 xch[8 * 3 * 64];
    const int B = a.B, Hp = a.Hp, n = a.n;
    const int tiles_x = Hp / 16, tiles_y = (B + 15) / 16, tiles = tiles_x * tiles_y;
    const int pass = blockIdx.x / tiles, bid = blockIdx.x % tiles;
    const int rt = bid / tiles_x, m0 = rt * 16, n0 = (bid % tiles_x) * 16;
    const int lane = threadIdx.x & 63, kq = threadIdx.x >> 6, li = lane & 15, kg = lane >> 4;
    const TinyChainDir& e = a.d[pass];
    const int orow = m0 + li, ocol = n0 + 4 * kg;  // this lane's (row, 4 hidden units) of the tile's tails (wave 0)
    const bool tail = kq == 0 && orow < B;
    const int row = orow < B ? orow : B - 1;        // clamped rows are computed, never stored
    const int ns = Hp / 16;
    unsigned* cnt = a.sync + ((size_t)pass * tiles_y + rt) * 32;

    if (a.zero_next && blockIdx.x == 0)  // the counters of the pass's next launch (visible to it across the kernel boundary)
        for (int i = threadIdx.x; i < TINY_CHAIN_SYNC_WORDS; i += 512) a.zero_next[i] = 0u;
    float4 wr[NS][3], wg[NS][2];
    load_w_kcontig<NS, 3>(wr, e.whh, Hp, Hp, n0 + li, ns, kq, kg);
    load_w_kcontig<NS, 2>(wg, e.wgm, Hp, Hp, n0 + li, ns, kq, kg);
    const float4 bn = cld4(e.bhn + ocol);

    for (int step = 0; step < n; ++step) {
        const int v = e.reverse ? n - 1 - step : step;
        const long r0 = (long)v * B;
        // ---- GRU level: h = GRUCell(x_v, agg)
        {
            float4 gir = f4zero(), giz = f4zero(), gin = f4zero();
            if (tail) {  // before the wait: the input-side table row of this vertex' type
                const float* gi = e.gi + (long)a.types[r0 + orow] * 3 * Hp;
                gir = cld4(gi + ocol); giz = cld4(gi + Hp + ocol); gin = cld4(gi + 2 * Hp + ocol);
            }
            float4 ghr = f4zero(), ghz = f4zero(), ghn = f4zero(), hp = f4zero();
            const bool product = step > 0 || e.level0_product;
            if (product) {
                if (step > 0 && !chain_wait<1>(cnt, (unsigned)(tiles_x * (2 * step)), a.status)) { tiny_fail(a); return; }
                float4 fa[NS];
                load_a_frags<NS>(fa, e.agg + r0 * Hp, Hp, row, ns, kq, kg);
                if (tail) hp = ld4_l2(e.agg + r0 * Hp, (long)orow * Hp + ocol);
                f32x4v acc[3];
                contract<NS, 3>(acc, wr, fa);
                slices_to_lds<3>(xch, acc, kq, lane);
                if (tail) {
                    float4 s3[3];
                    slices_sum<3>(xch, s3, lane);
                    ghr = s3[0]; ghz = s3[1]; ghn = s3[2];
                }
            }
            if (tail) {
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
                st4_sc1(rsrc_of(e.hs + r0 * Hp), (unsigned)((orow * Hp + ocol) * 4), Hn);
                tail_signal(cnt, kq, lane);
                float* g = e.gates + (r0 + orow) * 3 * Hp;
                cst4(g + ocol, R);
                cst4(g + Hp + ocol, Z);
                cst4(g + 2 * Hp + ocol, N);
                cst4(e.ghn + (r0 + orow) * Hp + ocol, GN);
                if (!product) cst4(e.agg + (r0 + orow) * Hp + ocol, f4zero());  // h_prev = 0: left for the backward pass
            } else {
                tail_signal(cnt, kq, lane);
            }
        }
        if (step == n - 1) break;  // the pass's last vertex sends no message
        // ---- gate / mapper level: msg = sigmoid(..) * (..), and the next level's predecessor sum
        {
            const long r1 = (long)(e.reverse ? v - 1 : v + 1) * B;
            float4 vg = f4zero(), vm = f4zero(), others = f4zero();
            bool mine = false;  // is this level a source of the next one (its slot is in the list)?
            if (tail) {  // before the wait: the vertex-id row, and the messages of the next level's OTHER sources
                const float* gv = e.gv + (size_t)v * 2 * Hp;
                vg = cld4(gv + ocol); vm = cld4(gv + Hp + ocol);
                const int own = (int)r0 + orow;
                for (int k = e.off[r1 + orow]; k < e.off[r1 + orow + 1]; ++k) {
                    const int s = e.idx[k];
                    if (s != own) f4add(others, ld4_l2(e.msg, (long)s * Hp + ocol));
                    else mine = true;
                }
            }
            if (!chain_wait<1>(cnt, (unsigned)(tiles_x * (2 * step + 1)), a.status)) { tiny_fail(a); return; }
            float4 fa[NS];
            load_a_frags<NS>(fa, e.hs + r0 * Hp, Hp, row, ns, kq, kg);
            f32x4v acc[2];
            contract<NS, 2>(acc, wg, fa);
            slices_to_lds<2>(xch, acc, kq, lane);
            if (tail) {
                float4 s2[2];
                slices_sum<2>(xch, s2, lane);
                const float4 pa = make_float4(s2[0].x + vg.x, s2[0].y + vg.y, s2[0].z + vg.z, s2[0].w + vg.w);
                const float4 m = make_float4(s2[1].x + vm.x, s2[1].y + vm.y, s2[1].z + vm.z, s2[1].w + vm.w);
                const float4 g = make_float4(1.f / (1.f + expf(-pa.x)), 1.f / (1.f + expf(-pa.y)), 1.f / (1.f + expf(-pa.z)),
                                             1.f / (1.f + expf(-pa.w)));
                const float4 own = make_float4(g.x * m.x, g.y * m.y, g.z * m.z, g.w * m.w);
                // every source of the next level other than this one is a finished level; this level's own message
                // (if it is a source) comes from registers
                float4 agg = others;
                if (mine) f4add(agg, own);
                st4_sc1(rsrc_of(e.agg + r1 * Hp), (unsigned)((orow * Hp + ocol) * 4), agg);
                tail_signal(cnt, kq, lane);
                float* o = e.gm + (r0 + orow) * 2 * Hp;
                cst4(o + ocol, g);
                cst4(o + Hp + ocol, m);
                cst4(e.msg + (r0 + orow) * Hp + ocol, own);  // (re-read by this lane behind a later phase's drain)
            } else {
                tail_signal(cnt, kq, lane);
            }
        }
    }
}

// ---------------------------------------------------------------------------------- backward pass
// Per level, in reverse pass order: (A) d h += d gm [Wg | Wm], then the GRUCell pointwise backward (the pass's last
// vertex sent no message: pointwise part only); (B) d agg += d gh W_hh, then the gate / mapper pointwise backward of the
// level processed next, whose d msg gathers d agg rows of this level (registers) and of finished ones.
template <int NS>
__global__ __launch_bounds__(512) void tiny_chain_bwd_kernel(TinyChainArgs a) {
    __shared__ float4 xch[8 * 64];
    const int B = a.B, Hp = a.Hp, n = a.n;
    const int tiles_x = Hp / 16, tiles_y = (B + 15) / 16, tiles = tiles_x * tiles_y;
    const int pass = blockIdx.x / tiles, bid = blockIdx.x % tiles;
    const int rt = bid / tiles_x, m0 = rt * 16, n0 = (bid % tiles_x) * 16;
    const int lane = threadIdx.x & 63, kq = threadIdx.x >> 6, li = lane & 15, kg = lane >> 4;
    const TinyChainDir& e = a.d[pass];
    const int orow = m0 + li, ocol = n0 + 4 * kg;
    const bool tail = kq == 0 && orow < B;
    const int row = orow < B ? orow : B - 1;
    unsigned* cnt = a.sync + ((size_t)pass * tiles_y + rt) * 32;

    float4 wgm[2 * NS][1], whh[3 * NS][1];
    load_w_kmajor<2 * NS>(wgm, e.wgm, Hp, n0 + li, 2 * Hp / 16, kq, kg);
    load_w_kmajor<3 * NS>(whh, e.whh, Hp, n0 + li, 3 * Hp / 16, kq, kg);

    for (int step = n - 1; step >= 0; --step) {
        const unsigned q = 2u * (unsigned)(n - 1 - step);
        const int v = e.reverse ? n - 1 - step : step;
        const long r0 = (long)v * B;
        const bool phase_b = step > 0 || e.level0_product;  // (h_prev of the pass's first level is zero: nothing to propagate)
        float4 dHP = f4zero();
        // ---- (A)
        {
            float4 R = f4zero(), Z = f4zero(), N = f4zero(), GN = f4zero(), hp = f4zero(), D = f4zero();
            if (tail) {  // before the wait: what the forward pass stored, and d h as accumulated before the launch
                const float* g = e.gates + (r0 + orow) * 3 * Hp;
                R = cld4(g + ocol); Z = cld4(g + Hp + ocol); N = cld4(g + 2 * Hp + ocol);
                GN = cld4(e.ghn + (r0 + orow) * Hp + ocol);
                hp = cld4(e.agg + (r0 + orow) * Hp + ocol);
                D = cld4(e.dhs + (r0 + orow) * Hp + ocol);
            }
            if (step < n - 1) {
                if (!chain_wait<1>(cnt, (unsigned)tiles_x * q, a.status)) { tiny_fail(a); return; }
                float4 fa[2 * NS];
                load_a_frags<2 * NS>(fa, e.dgm + r0 * 2 * Hp, 2 * Hp, row, 2 * Hp / 16, kq, kg);
                f32x4v acc[1];
                contract<2 * NS, 1>(acc, wgm, fa);
                slices_to_lds<1>(xch, acc, kq, lane);
                if (tail) {
                    float4 s1[1];
                    slices_sum<1>(xch, s1, lane);
                    f4add(D, s1[0]);
                }
            }
            if (tail) {
                float4 dR, dZ, dGN, dPN;
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
                const __amdgpu_buffer_rsrc_t rg = rsrc_of(e.dgh + r0 * 3 * Hp);
                const unsigned o = (unsigned)((orow * 3 * Hp + ocol) * 4);
                st4_sc1(rg, o, dR);
                st4_sc1(rg, o + (unsigned)Hp * 4u, dZ);
                st4_sc1(rg, o + (unsigned)Hp * 8u, dGN);
                tail_signal(cnt, kq, lane);
                cst4(e.dhs + (r0 + orow) * Hp + ocol, D);
                cst4(e.dgin + (r0 + orow) * Hp + ocol, dPN);
                if (!phase_b) cst4(e.dagg + (r0 + orow) * Hp + ocol, dHP);
            } else {
                tail_signal(cnt, kq, lane);
            }
        }
        if (!phase_b) break;
        // ---- (B)
        {
            const long r1 = (long)(e.reverse ? v + 1 : v - 1) * B;  // the level processed next (step - 1)
            float4 others = f4zero(), g = f4zero(), m = f4zero();
            bool mine = false;
            if (tail && step > 0) {  // before the wait: d agg of the next level's OTHER consumers, its stored gate and mapper
                const int own = (int)r0 + orow;
                for (int k = e.toff[r1 + orow]; k < e.toff[r1 + orow + 1]; ++k) {
                    const int s = e.tidx[k];
                    if (s != own) f4add(others, ld4_l2(e.dagg, (long)s * Hp + ocol));
                    else mine = true;
                }
                if (e.off2)
                    for (int k = e.off2[r1 + orow]; k < e.off2[r1 + orow + 1]; ++k)
                        f4add(others, cld4(e.gsrc2 + (long)e.idx2[k] * Hp + ocol));
                const float* gq = e.gm + (r1 + orow) * 2 * Hp;
                g = cld4(gq + ocol); m = cld4(gq + Hp + ocol);
            }
            if (!chain_wait<1>(cnt, (unsigned)tiles_x * (q + 1), a.status)) { tiny_fail(a); return; }
            float4 fa[3 * NS];
            load_a_frags<3 * NS>(fa, e.dgh + r0 * 3 * Hp, 3 * Hp, row, 3 * Hp / 16, kq, kg);
            f32x4v acc[1];
            contract<3 * NS, 1>(acc, whh, fa);
            slices_to_lds<1>(xch, acc, kq, lane);
            if (tail) {
                float4 s1[1];
                slices_sum<1>(xch, s1, lane);
                float4 d = dHP;
                f4add(d, s1[0]);
                if (step > 0) {
                    float4 dm_ = others;
                    if (mine) f4add(dm_, d);
                    const float4 dg = make_float4(dm_.x * m.x * g.x * (1.f - g.x), dm_.y * m.y * g.y * (1.f - g.y),
                                                  dm_.z * m.z * g.z * (1.f - g.z), dm_.w * m.w * g.w * (1.f - g.w));
                    const float4 dmap = make_float4(dm_.x * g.x, dm_.y * g.y, dm_.z * g.z, dm_.w * g.w);
                    const __amdgpu_buffer_rsrc_t ro = rsrc_of(e.dgm + r1 * 2 * Hp);
                    const unsigned o = (unsigned)((orow * 2 * Hp + ocol) * 4);
                    st4_sc1(ro, o, dg);
                    st4_sc1(ro, o + (unsigned)Hp * 4u, dmap);
                }
                tail_signal(cnt, kq, lane);
                cst4(e.dagg + (r0 + orow) * Hp + ocol, d);  // (re-read by this lane's later gathers, behind a later phase's drain)
            } else {
                tail_signal(cnt, kq, lane);
            }
        }
    }
}

// ---------------------------------------------------------------------------------- host
template <class K>
static int resident_blocks(K kernel) {
    int dev = 0, per_cu = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 512, 0) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    if (per_cu > 4) per_cu = 4;  // 512-thread workgroups: at most 4 per CU
    return per_cu * cus;
}
static int tiny_resident(int ns) {
    // workgroups the device holds at once, the smaller answer of the forward and the backward kernel (cached per device)
    static int cache[3][64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    const int slot = ns == 1 ? 0 : ns == 2 ? 1 : 2;
    int& c = cache[slot][dev];
    if (c == 0) {
        int f, b;
        if (ns == 1) { f = resident_blocks(tiny_chain_fwd_kernel<1>); b = resident_blocks(tiny_chain_bwd_kernel<1>); }
        else if (ns == 2) { f = resident_blocks(tiny_chain_fwd_kernel<2>); b = resident_blocks(tiny_chain_bwd_kernel<2>); }
        else { f = resident_blocks(tiny_chain_fwd_kernel<4>); b = resident_blocks(tiny_chain_bwd_kernel<4>); }
        c = (f < b ? f : b) > 0 ? (f < b ? f : b) : -1;
    }
    return c > 0 ? c : 0;
}
static int tiny_ns(int Hp) { return Hp <= 128 ? 1 : Hp <= 256 ? 2 : 4; }

// Does a pass of B rows per level (x `passes` directions in one launch) run as a persistent launch?
// DVAE_CHAIN_TINY_PERSIST: 0 = never, 1 = default.  Hp <= 512 (the weight tiles live in registers), at most
// chain_tiny_rows() rows, and only while TWICE the grid fits the device (margin: the occupancy query's known
// over-count, and a second process stepping on the same card, as the two-rank test on a one-GPU box does).
bool tiny_chain_fits(int B, int Hp, int passes) {
    static const int knob = [] { const char* v = getenv("DVAE_CHAIN_TINY_PERSIST"); return v ? atoi(v) : 1; }();
    if (!knob || B < 1 || B > chain_tiny_rows() || Hp % 16 != 0 || Hp > 512 || passes < 1 || passes > 2) return false;
    if ((long)passes * cdiv(B, 16) * 32 + 64 > TINY_CHAIN_SYNC_WORDS) return false;
    const long grid = (long)passes * (Hp / 16) * cdiv(B, 16);
    return 2 * grid <= tiny_resident(tiny_ns(Hp));
}

static int tiny_chain_check(const TinyChainArgs& a) {
    DVAE_REQUIRE(a.ndirs >= 1 && a.ndirs <= 2 && a.Hp % 16 == 0 && a.Hp <= 512 && a.B >= 1 && a.n >= 2, "persistent chain: bad shape");
    DVAE_REQUIRE((long)a.n * a.B * 3 * a.Hp * 4 < 4294967000L, "persistent chain: a state array exceeds 4 GiB");
    DVAE_REQUIRE(a.sync && a.status == a.sync + TINY_CHAIN_SYNC_WORDS - 64, "persistent chain: the status word follows the counters");
    DVAE_REQUIRE((long)a.ndirs * cdiv(a.B, 16) * 32 + 64 <= TINY_CHAIN_SYNC_WORDS, "persistent chain: too many row tiles");
    return DVAE_OK;
}

int tiny_chain_fwd(const TinyChainArgs& a, hipStream_t s) {
    DVAE_TRY(tiny_chain_check(a));
    DVAE_HIP(hipMemsetAsync(a.sync, 0, sizeof(unsigned) * TINY_CHAIN_SYNC_WORDS, s));  // counters + status
    const dim3 grid((unsigned)(a.ndirs * (a.Hp / 16) * cdiv(a.B, 16))), block(512);
    switch (tiny_ns(a.Hp)) {
    case 1: hipLaunchKernelGGL(tiny_chain_fwd_kernel<1>, grid, block, 0, s, a); break;
    case 2: hipLaunchKernelGGL(tiny_chain_fwd_kernel<2>, grid, block, 0, s, a); break;
    default: hipLaunchKernelGGL(tiny_chain_fwd_kernel<4>, grid, block, 0, s, a); break;
    }
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}

int tiny_chain_bwd(const TinyChainArgs& a, hipStream_t s) {
    DVAE_TRY(tiny_chain_check(a));
    if (!a.sync_is_zero) DVAE_HIP(hipMemsetAsync(a.sync, 0, sizeof(unsigned) * TINY_CHAIN_SYNC_WORDS, s));
    const dim3 grid((unsigned)(a.ndirs * (a.Hp / 16) * cdiv(a.B, 16))), block(512);
    switch (tiny_ns(a.Hp)) {
    case 1: hipLaunchKernelGGL(tiny_chain_bwd_kernel<1>, grid, block, 0, s, a); break;
    case 2: hipLaunchKernelGGL(tiny_chain_bwd_kernel<2>, grid, block, 0, s, a); break;
    default: hipLaunchKernelGGL(tiny_chain_bwd_kernel<4>, grid, block, 0, s, a); break;
    }
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}

}  // namespace dvae
