// Single-pass loss kernels of the teacher-forced decoder:
//   * edge predictor tail + BCE on probabilities (one wavefront per vertex pair,
//     wave-level reduction), forward and grouped backward
//   * vertex-type log-softmax + CE
//   * reparameterisation + KL divergence, forward and backward
// All of them are HBM/L2-bandwidth bound (the pair kernels stream two 4*Hp rows
// per pair); every row access is a coalesced float4 per lane.
#include "common.h"
#include "dvae_hip.h"

namespace dvae {

__device__ __forceinline__ float4 ld4c(const float* p) { return *reinterpret_cast<const float4*>(p); }

// The loss terms are summed into sums[0..2] = {pair BCE, vertex CE, KLD} by kernels that may run on two streams; each
// of their workgroups, after its own atomic add, draws a ticket from sums[4], and the one that draws the last of
// `total` tickets writes out3 = {res + beta kld, res, kld} (models.py:858-860): no separate finalize launch, and no
// order between the kernels.  Called by ONE thread per workgroup.
// Ordering without device-wide fences (two __threadfence() here were ~3.5 us each, MI355X_MICROARCH.md price list: 7 us
// of a 12 us single-workgroup CE launch at batch 32): every access to `sums` is a read-modify-write atomic, which
// executes at the memory side, for every XCD alike.  A workgroup's add RETURNS (s_waitcnt on its result) before it
// draws its ticket, so the add is performed before the ticket; tickets are performed in ticket order on one word; the
// last drawer's read-backs are issued after its ticket returned.  Nothing here is handed over through plain stores.
__device__ __forceinline__ void loss_sum_and_ticket(float* slot, float v, float* sums, unsigned total, float beta,
                                                    float* out3) {
    if (v != 0.f) {
        const float old = atomicAdd(slot, v);
        asm volatile("s_waitcnt vmcnt(0)" ::"v"(old) : "memory");  // performed: the returning form has come back
    }
    if (!sums) return;
    const unsigned ticket = atomicAdd(reinterpret_cast<unsigned*>(sums + 4), 1u);
    if (ticket == total - 1) {
        const float pair = atomicAdd(sums + 0, 0.f), ce = atomicAdd(sums + 1, 0.f), kld = atomicAdd(sums + 2, 0.f);
        const float res = pair + ce;
        out3[0] = res + beta * kld;
        out3[1] = res;
        out3[2] = kld;
    }
}

// ---------------------------------------------------------------------------
// Edge predictor tail + BCE.  A wave owns a pair; lane l holds the 4 hidden units
// c = 4 l + 256 i + {0..3} of iteration i (coalesced float4 rows).  NIT = number of
// iterations (HEp <= 256 NIT), compile-time so the per-pair values stay in registers.
//
// forward, per pair: s = b2 + sum_j w2[j] relu(pre[j]),  pre = EA[a] + EB[h] (+ EC[c]) + b1
//   -> loss term, dl = d loss / d s, the ReLU MASK of the pair (1 bit per hidden unit:
//   word (p, i / 8, lane), bit 4 (i % 8) + component) and, accumulated over the pairs of the
//   workgroup, the w2 gradient partial  sum_p dl[p] relu(pre_p[j])  (unscaled by the upstream
//   gradient).  With the mask stored, the backward passes never re-read EA / EB.
template <int NIT>
__global__ __launch_bounds__(256) void pair_fwd_kernel(const float* __restrict__ ea, const float* __restrict__ eb,
                                                       const float* __restrict__ ec, const int* __restrict__ pair_a,
                                                       const int* __restrict__ pair_h, const int* __restrict__ pair_c,
                                                       const uint8_t* __restrict__ pair_t, const float* __restrict__ b1,
                                                       const float* __restrict__ w2, const float* __restrict__ b2,
                                                       float* __restrict__ dl, float* __restrict__ loss_sum,
                                                       uint32_t* __restrict__ mask, float* __restrict__ dw2_partial,
                                                       int n_pairs, int HEp, float* __restrict__ sums,
                                                       unsigned tickets, float beta, float* __restrict__ out3,
                                                       int group) {
    constexpr int NW = (NIT + 7) / 8;
    extern __shared__ float sh[];  // [HEp] workgroup sum of the w2-gradient partials (+ 4 floats)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int waves_total = gridDim.x * 4;
    if (dw2_partial) {
        for (int c = threadIdx.x; c < HEp; c += 256) sh[c] = 0.f;
    }
    float local = 0.f;
    const float bias2 = b2[0];
    float aw[NIT * 4];
#pragma unroll
    for (int i = 0; i < NIT * 4; ++i) aw[i] = 0.f;
    // (a lane always works on the same hidden units, but keeping its slices of b1 / w2 in registers for all its pairs
    // was measured 0.5 % slower on the step: 159 instead of 124 VGPRs, one wave per SIMD fewer)
    // `group` > 0: the pairs come in runs of `group` that share table rows (the pairs of one graph, packer.cpp): a
    // workgroup then takes whole runs, its four waves the pairs of a run in turn, so the rows a graph's pairs share
    // are re-read through this CU's L1 / this XCD's L2 instead of from the Infinity Cache by workgroups all over the chip
    const int n_groups = group > 0 ? (n_pairs + group - 1) / group : 0;
    const int iters = group > 0 ? ((n_groups + gridDim.x - 1) / gridDim.x) * ((group + 3) / 4)
                                : (n_pairs + waves_total - 1) / waves_total;
    const int per_run = (group + 3) / 4;
    for (int it = 0; it < iters; ++it) {
        int p;
        if (group > 0) {
            const int run = blockIdx.x + (it / per_run) * gridDim.x;
            const int k = (it % per_run) * 4 + wave;
            p = (run < n_groups && k < group) ? run * group + k : n_pairs;
        } else {
            p = blockIdx.x * 4 + wave + it * waves_total;
        }
        if (p >= n_pairs) continue;
        const float* ra = ea + (long)pair_a[p] * HEp;
        const float* rb = eb + (long)pair_h[p] * HEp;
        const float* rc = ec ? ec + (long)pair_c[p] * HEp : nullptr;
        float rv[NIT * 4];
        uint32_t mw[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) mw[w] = 0u;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int c = lane * 4 + 256 * i;
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < HEp) {
                float4 a = ld4c(ra + c), b = ld4c(rb + c), bb = ld4c(b1 + c), w = ld4c(w2 + c);
                float4 pre = make_float4(a.x + b.x + bb.x, a.y + b.y + bb.y, a.z + b.z + bb.z, a.w + b.w + bb.w);
                if (rc) {
                    float4 cc = ld4c(rc + c);
                    pre.x += cc.x; pre.y += cc.y; pre.z += cc.z; pre.w += cc.w;
                }
                r = make_float4(fmaxf(pre.x, 0.f), fmaxf(pre.y, 0.f), fmaxf(pre.z, 0.f), fmaxf(pre.w, 0.f));
                uint32_t bits = (pre.x > 0.f ? 1u : 0u) | (pre.y > 0.f ? 2u : 0u) | (pre.z > 0.f ? 4u : 0u) |
                                (pre.w > 0.f ? 8u : 0u);
                mw[i / 8] |= bits << (4 * (i % 8));
                s += w.x * r.x + w.y * r.y + w.z * r.z + w.w * r.w;
            }
            rv[4 * i + 0] = r.x; rv[4 * i + 1] = r.y; rv[4 * i + 2] = r.z; rv[4 * i + 3] = r.w;
        }
        s = wave_sum(s) + bias2;
        // torch.sigmoid then F.binary_cross_entropy on the probability (models.py:745, :855)
        float prob = 1.f / (1.f + expf(-s));
        float t = (float)pair_t[p];
        float lp = fmaxf(logf(prob), -100.f);
        float lq = fmaxf(logf(1.f - prob), -100.f);
        // BCE backward: (p - t) / max(p (1-p), 1e-12), then sigmoid backward: * p (1-p)
        float pq = prob * (1.f - prob);
        float g = (prob - t) / fmaxf(pq, 1e-12f) * pq;
        if (lane == 0) {
            local += -(t * lp + (1.f - t) * lq);
            dl[p] = g;
        }
        if (mask) {
#pragma unroll
            for (int w = 0; w < NW; ++w) mask[((long)p * NW + w) * 64 + lane] = mw[w];
        }
        if (dw2_partial) {
#pragma unroll
            for (int i = 0; i < NIT * 4; ++i) aw[i] += g * rv[i];
        }
    }
    float* red = sh + HEp;
    if (lane == 0) red[wave] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        float v = red[0] + red[1] + red[2] + red[3];
        loss_sum_and_ticket(loss_sum, v, sums, tickets, beta, out3);
    }
    if (dw2_partial) {  // uniform
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int c = lane * 4 + 256 * i;
            if (c < HEp) {
                atomicAdd(&sh[c + 0], aw[4 * i + 0]);
                atomicAdd(&sh[c + 1], aw[4 * i + 1]);
                atomicAdd(&sh[c + 2], aw[4 * i + 2]);
                atomicAdd(&sh[c + 3], aw[4 * i + 3]);
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < HEp; c += 256) dw2_partial[(long)blockIdx.x * HEp + c] = sh[c];
    }
}

// The same forward for wide tables (HEp > 2048): the FOUR waves of a workgroup share a pair, wave w taking the hidden
// units of iterations 4 w .. 4 w + 3 (NITW = NIT / 4 iterations of 256 units each; same lane -> unit map and the same
// mask bits as above, a wave's 16 bits being one half of a mask word).  With a whole 4096-unit row per wave the per-pair
// values (relu outputs for the w2 gradient, mask words) take 230 VGPRs: two waves per SIMD, each waiting on its own 64
// loads -- the kernel ran at a fraction of the L2 rate (cfg 5: 22.6 ms for 4.1 M pairs).  A quarter row per wave is
// ~70 VGPRs; the four partial sums of a pair meet through LDS (one barrier per pair, double-buffered slots).
// TH (round 3, bf16 mode): the two tables are stored as bf16 (written so by the products that make them: half the
// bytes written, half the L2 bytes read here, and the two held rows take 16 registers instead of 32); `ea` / `eb` then
// point at bf16 words, rows HEp ELEMENTS apart, and the first layer's bias b1 is already IN the EB table (the product that
// makes it adds it in its epilogue; the table's no-input row holds b1): 16 registers and an add per unit less here
// (106 -> 90 registers: five workgroups per CU instead of four).  w2, the sums and the masks stay fp32.
__device__ __forceinline__ float4 bf16x4_to_f4(uint2 r) {
    return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                       __uint_as_float(r.y & 0xffff0000u));
}
typedef __bf16 lbf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 f4_to_bf16x4(float4 v) {  // round to nearest even, as the products' epilogues do
    const lbf16x4 b = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    return __builtin_bit_cast(uint2, b);
}
template <bool TH>
struct PairRow;
template <>
struct PairRow<false> {
    float4 v;
    __device__ __forceinline__ void load(const float* tab, long row, int HEp, int c) { v = ld4c(tab + row * HEp + c); }
    __device__ __forceinline__ void zero() { v = make_float4(0.f, 0.f, 0.f, 0.f); }
    __device__ __forceinline__ float4 get() const { return v; }
};
template <>
struct PairRow<true> {
    uint2 v;
    __device__ __forceinline__ void load(const float* tab, long row, int HEp, int c) {
        v = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(tab) + row * HEp + c);
    }
    __device__ __forceinline__ void zero() { v = make_uint2(0u, 0u); }
    __device__ __forceinline__ float4 get() const { return bf16x4_to_f4(v); }
};
// FULL: HEp == 1024 NITW (cfg 5: 4096), no unit of any lane lies past the row: the per-iteration range checks (an exec-mask
// branch each) are compiled out.
// The loop is bound by VECTOR INSTRUCTIONS, not by bytes or latency (round 3, counted in the ISA: ~330 per pair and wave
// = 1320 issue cycles per SIMD, and the measured 1.8 pairs per us and CU is exactly one pair per 1320 cycles; neither a
// 2x larger grid, nor five instead of four workgroups per CU, nor half the table bytes moved it): hence ReLU and mask
// bit from ONE compare (fmaxf costs a canonicalising max besides), the held EB row converted once per row instead of
// once per pair, no bias / third-table arithmetic where the template says there is none, the two logarithms of the loss
// in the one thread that uses them.
// Diagnostic builds only (tools/pair_ablate.sh; results are wrong, only the time matters): PAIR_ABL bit 0 loads every
// pair's rows from row 0 (always L1 / L2 hits), bit 1 skips the cross-wave exchange and its barrier, bit 2 skips the mask
// store, bit 3 skips the w2-gradient accumulation, bit 4 skips the table row loads altogether.
#ifndef PAIR_ABL
#define PAIR_ABL 0
#endif
template <int NITW, bool TH, bool FULL>
__global__ __launch_bounds__(256) void pair_fwd_split_kernel(const float* __restrict__ ea, const float* __restrict__ eb,
                                                             const float* __restrict__ ec, const int* __restrict__ pair_a,
                                                             const int* __restrict__ pair_h, const int* __restrict__ pair_c,
                                                             const uint8_t* __restrict__ pair_t, const float* __restrict__ b1,
                                                             const float* __restrict__ w2, const float* __restrict__ b2,
                                                             float* __restrict__ dl, float* __restrict__ loss_sum,
                                                             uint32_t* __restrict__ mask, float* __restrict__ dw2_partial,
                                                             int n_pairs, int HEp, float* __restrict__ sums,
                                                             unsigned tickets, float beta, float* __restrict__ out3,
                                                             int group) {
    static_assert(NITW == 4, "a wave's mask bits are one 16-bit half of a word: 4 iterations x 4 units");
    constexpr int NW = (4 * NITW + 7) / 8;  // mask words per pair and lane
    extern __shared__ float sh[];  // [HEp] workgroup sum of the w2-gradient partials, then 8 exchange slots
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (dw2_partial) {
        for (int c = threadIdx.x; c < HEp; c += 256) sh[c] = 0.f;
    }
    float* xch = sh + HEp;
    // the mask halves of a pair's four waves meet here so that waves 0 and 2 store whole words (see the store below)
    uint16_t* mb = reinterpret_cast<uint16_t*>(xch + 8);  // [2 parities][2 wave pairs][64 lanes][2 halves]
    float local = 0.f;
    const float bias2 = b2[0];
    // (pairs of floats: the w2 sum and the w2-gradient update then compile to packed fp32 multiply-adds, two units per
    // instruction -- this loop is bound by its vector instruction count)
    typedef float pf2 __attribute__((ext_vector_type(2)));
    pf2 aw[NITW * 2];
#pragma unroll
    for (int i = 0; i < NITW * 2; ++i) aw[i] = pf2{0.f, 0.f};
    // pairs of this workgroup.  `group` > 0 (runs of `group` pairs = the pairs of one graph, which share table rows: a
    // graph's pairs touch 1 MB of EA and ~3.6 MB of EB rows at cfg 5, each EB row ~9 times): the workgroups that share
    // an XCD (blockIdx % 8) work through the SAME graph together, pair k of the graph going to workgroup k mod J of the
    // XCD's J, so that the XCD's L2 (4 MB) holds one graph's rows while they are re-read, instead of every workgroup
    // streaming a graph of its own from HBM (4.1 M pairs x 32 KB = 132 GB per pass).  Otherwise: every gridDim.x-th pair.
    const int n_groups = group > 0 ? (n_pairs + group - 1) / group : 0;
    const int xcd = blockIdx.x & 7, J = gridDim.x >> 3, j = blockIdx.x >> 3;
    const bool coop = group > 0 && (gridDim.x & 7) == 0 && J > 0;
    // ... in runs of RUN consecutive pairs: the pairs (v, u = 0, 1, ...) of a vertex v read the SAME EB row until u
    // passes a predecessor of v (~9 pairs per row at cfg 5), so a workgroup that walks consecutive pairs keeps the row
    // in registers and reads one table row per pair instead of two (132 -> ~80 GB of L2 reads per pass).
    constexpr int RUN = 16;
    // (round 3) the XCD's graphs xcd, xcd + 8, ... form ONE sequence of runs dealt to its J workgroups round-robin, so
    // no workgroup idles when J exceeds a graph's runs (cfg 5: 126 runs; 160 workgroups per XCD at five per CU) and
    // the XCD works on at most two consecutive graphs at any moment
    const int runs = (group + RUN - 1) / RUN;
    const int graphs_here = coop && xcd < n_groups ? (n_groups - xcd + 7) / 8 : 0;
    const int iters = coop ? (int)(((long)graphs_here * runs + J - 1) / J) * RUN
                           : group > 0 ? ((n_groups + gridDim.x - 1) / gridDim.x) * group : (n_pairs + gridDim.x - 1) / gridDim.x;
    int executed = 0, held = -1;  // held: the EB row in `bq`
    float4 bq[NITW];              // this wave's quarter of the EB row (bf16 tables: converted when the row is loaded)
    float4 b1q[NITW], w2q[NITW];  // ... of b1 and of w2
#pragma unroll
    for (int ii = 0; ii < NITW; ++ii) {
        const int c = lane * 4 + 256 * (wave * NITW + ii);
        if (!TH) b1q[ii] = (FULL || c < HEp) ? ld4c(b1 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        w2q[ii] = (FULL || c < HEp) ? ld4c(w2 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // The pair of each iteration in turn (n_pairs when there is none; uniform over the workgroup), kept as running
    // counters: computed from `it` with divisions and remainders it was ~170 scalar instructions per pair and wave --
    // with 16-20 waves per CU on ONE scalar unit that, not the vector work, set the kernel's rate (cfg 5: 10.1 ms).
    int nx_it = 0, nx_w = 0, nx_t = j, nx_run = coop ? xcd : (int)blockIdx.x, nx_k = 0;
    if (coop)
        while (nx_t >= runs) { nx_t -= runs; nx_run += 8; }
    auto next_pair = [&]() -> int {
        int p = n_pairs;
        if (nx_it < iters) {
            if (coop) {
                const int k = nx_t * RUN + nx_w;
                if (nx_run < n_groups && k < group) p = nx_run * group + k;
                if (++nx_w == RUN) {
                    nx_w = 0;
                    nx_t += J;
                    while (nx_t >= runs) { nx_t -= runs; nx_run += 8; }
                }
            } else if (group > 0) {
                if (nx_run < n_groups) p = nx_run * group + nx_k;
                if (++nx_k == group) { nx_k = 0; nx_run += gridDim.x; }
            } else {
                p = nx_run;  // every gridDim.x-th pair
                nx_run += gridDim.x;
            }
            if (p > n_pairs) p = n_pairs;
        }
        ++nx_it;
        return p;
    };
    // The rows of the NEXT pair are requested as soon as the current pair's have been consumed, into the same
    // registers: their latency (and that of the index loads in front of them) then runs under the current pair's
    // reduction, barrier and loss arithmetic instead of in front of the next pair's (the kernel was bound by that
    // chain -- 3.4 us per pair and workgroup -- not by bytes: halving its L2 reads alone changed nothing).
    PairRow<TH> aq[NITW];
    auto fetch = [&](int p) {
        if (PAIR_ABL & 16) return;
        const long arow = (PAIR_ABL & 1) ? 0 : pair_a[p];
#pragma unroll
        for (int ii = 0; ii < NITW; ++ii) {
            const int c = lane * 4 + 256 * (wave * NITW + ii);
            if (FULL || c < HEp) aq[ii].load(ea, arow, HEp, c);
            else aq[ii].zero();
        }
        const int hrow = (PAIR_ABL & 1) ? 0 : pair_h[p];
        if (hrow != held) {  // uniform over the workgroup
#pragma unroll
            for (int ii = 0; ii < NITW; ++ii) {
                const int c = lane * 4 + 256 * (wave * NITW + ii);
                PairRow<TH> t;
                if (FULL || c < HEp) t.load(eb, hrow, HEp, c);
                else t.zero();
                bq[ii] = t.get();
            }
            held = hrow;
        }
    };
    int p = next_pair();
    if (p < n_pairs) fetch(p);
    for (int it = 0; it < iters; ++it) {
        const int pn = next_pair();
        if (p >= n_pairs) {  // uniform over the workgroup
            if (pn < n_pairs) fetch(pn);
            p = pn;
            continue;
        }
        const float* rc = ec ? ec + (long)pair_c[p] * HEp : nullptr;
        const float t = (float)pair_t[p];
        pf2 rv[NITW * 2];
        uint32_t bits16 = 0u;
        pf2 s2 = pf2{0.f, 0.f};
#pragma unroll
        for (int ii = 0; ii < NITW; ++ii) {
            const int c = lane * 4 + 256 * (wave * NITW + ii);
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (FULL || c < HEp) {
                const float4 a = aq[ii].get(), b = bq[ii], w = w2q[ii];
                float4 pre;
                if constexpr (TH) {  // (b1 is in the EB table)
                    pre = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
                } else {
                    const float4 bb = b1q[ii];
                    pre = make_float4(a.x + b.x + bb.x, a.y + b.y + bb.y, a.z + b.z + bb.z, a.w + b.w + bb.w);
                    if (rc) {
                        float4 cc = ld4c(rc + c);
                        pre.x += cc.x; pre.y += cc.y; pre.z += cc.z; pre.w += cc.w;
                    }
                }
                // ReLU and its mask bit from one compare (the same values as fmaxf(pre, 0): NaN and -0 give 0)
                const bool px = pre.x > 0.f, py = pre.y > 0.f, pz = pre.z > 0.f, pw_ = pre.w > 0.f;
                r = make_float4(px ? pre.x : 0.f, py ? pre.y : 0.f, pz ? pre.z : 0.f, pw_ ? pre.w : 0.f);
                const uint32_t b4 = (px ? 1u : 0u) | (py ? 2u : 0u) | (pz ? 4u : 0u) | (pw_ ? 8u : 0u);
                bits16 |= b4 << (4 * ii);
                s2 += pf2{w.x, w.y} * pf2{r.x, r.y};
                s2 += pf2{w.z, w.w} * pf2{r.z, r.w};
            }
            rv[2 * ii + 0] = pf2{r.x, r.y};
            rv[2 * ii + 1] = pf2{r.z, r.w};
        }
        float s = s2.x + s2.y;
        if (pn < n_pairs) fetch(pn);  // aq / bq are free again
        s = wave_sum(s);
        const int par = executed++ & 1;  // (parity of the pairs actually run: skipped iterations take no barrier)
        float* slot = xch + 4 * par;
        if (!(PAIR_ABL & 2)) {
            if (lane == 0) slot[wave] = s;
            if (mask) mb[((par * 2 + (wave >> 1)) * 64 + lane) * 2 + (wave & 1)] = (uint16_t)bits16;
            __syncthreads();  // (the slots of pair it - 1 are free again: every wave passed this barrier after reading them)
            s = slot[0] + slot[1] + slot[2] + slot[3];  // the same order in every wave
        }
        s += bias2;
        float prob, pq, g;
        if constexpr (TH) {
            // bf16 mode: hardware exp / reciprocal (v_exp_f32, v_rcp_f32: ~1 ulp each).  Every lane of the workgroup
            // computes this same scalar, i.e. it costs as many vector instructions as sixteen table units would: the
            // library routines (~60 instructions) were a fifth of the loop (ablation: profiles/r03_pair_fwd_ablation.txt)
            prob = __builtin_amdgcn_rcpf(1.f + __expf(-s));
            pq = prob * (1.f - prob);
            g = (prob - t) * __builtin_amdgcn_rcpf(fmaxf(pq, 1e-12f)) * pq;
        } else {
            prob = 1.f / (1.f + expf(-s));
            pq = prob * (1.f - prob);
            g = (prob - t) / fmaxf(pq, 1e-12f) * pq;
        }
        if (threadIdx.x == 0) {
            const float lp = fmaxf(logf(prob), -100.f);
            const float lq = fmaxf(logf(1.f - prob), -100.f);
            local += -(t * lp + (1.f - t) * lq);
            dl[p] = g;
        }
        // iterations 4 w .. 4 w + 3 = nibbles 4 (w % 2) .. + 3 of word w / 2: its low or high half.  Waves 0 and 2 store
        // the WORD (their half and, from LDS, their odd neighbour's): as 2-byte stores of alternate halves from two
        // waves they were byte-masked writes that L2 merges by read-modify-write (9.07 -> 8.4 ms at cfg 5).  (The whole
        // pair's 512 bytes as 32 x 16-byte write-through stores from wave 0 -- so that the mask does not displace
        // the graph's table rows in the XCD's L2 -- was measured too: 8.95 ms.)
        if (mask && !(PAIR_ABL & 4) && !(wave & 1)) {
            const uint32_t hi = (PAIR_ABL & 2) ? 0u : mb[((par * 2 + (wave >> 1)) * 64 + lane) * 2 + 1];
            mask[((long)p * NW + (wave >> 1)) * 64 + lane] = bits16 | (hi << 16);
        }
        if (dw2_partial && !(PAIR_ABL & 8)) {
#pragma unroll
            for (int i = 0; i < NITW * 2; ++i) aw[i] += pf2{g, g} * rv[i];
        }
        p = pn;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        loss_sum_and_ticket(loss_sum, local, sums, tickets, beta, out3);
    }
    if (dw2_partial) {  // uniform; the waves own disjoint hidden units
#pragma unroll
        for (int ii = 0; ii < NITW; ++ii) {
            const int c = lane * 4 + 256 * (wave * NITW + ii);
            if (c < HEp) {
                sh[c + 0] = aw[2 * ii].x; sh[c + 1] = aw[2 * ii].y; sh[c + 2] = aw[2 * ii + 1].x; sh[c + 3] = aw[2 * ii + 1].y;
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < HEp; c += 256) dw2_partial[(long)blockIdx.x * HEp + c] = sh[c];
    }
}

// ---------------------------------------------------------------------------
// grouped backward: one wave per chunk of pairs that share one row of an edge-predictor table.
//   chunk q covers grp_pairs[chunk_off[q] .. chunk_off[q+1]) (grp_pairs == NULL: the pair ids
//   are the positions themselves); chunk_row[q] = destination row, bit 31 set when several
//   chunks share the row (-> atomicAdd into the zeroed table).
//   d table[row][j] = scale * w2[j] * sum_p dl[p] mask_p[j]
__device__ __forceinline__ float sel(uint32_t w, int bit, float g) { return (w >> bit) & 1u ? g : 0.f; }

template <int NIT>
__global__ __launch_bounds__(256) void pair_bwd_kernel(const int* __restrict__ chunk_off,
                                                       const int* __restrict__ chunk_row,
                                                       const int* __restrict__ grp_pairs, const float* __restrict__ dl,
                                                       const uint32_t* __restrict__ mask, const float* __restrict__ w2,
                                                       const float* __restrict__ scale, float* __restrict__ dtab,
                                                       int n_chunks, int HEp) {
    constexpr int NW = (NIT + 7) / 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int waves_total = gridDim.x * 4;
    const float sc = scale ? scale[0] : 1.f;
    for (int q = blockIdx.x * 4 + wave; q < n_chunks; q += waves_total) {
        const int cr = chunk_row[q];
        const int row = cr & 0x7fffffff;
        float acc[NIT * 4];
#pragma unroll
        for (int i = 0; i < NIT * 4; ++i) acc[i] = 0.f;
        const int k1 = chunk_off[q + 1];
        for (int k = chunk_off[q]; k < k1; ++k) {
            const int p = grp_pairs ? grp_pairs[k] : k;
            const float g = dl[p] * sc;
            uint32_t mw[NW];
#pragma unroll
            for (int w = 0; w < NW; ++w) mw[w] = mask[((long)p * NW + w) * 64 + lane];
#pragma unroll
            for (int i = 0; i < NIT * 4; ++i) acc[i] += sel(mw[i / 32], i % 32, g);
        }
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int c = lane * 4 + 256 * i;
            if (c < HEp) {
                float4 w = ld4c(w2 + c);
                float4 v = make_float4(acc[4 * i] * w.x, acc[4 * i + 1] * w.y, acc[4 * i + 2] * w.z, acc[4 * i + 3] * w.w);
                float* d = dtab + (long)row * HEp + c;
                if (cr < 0) {
                    atomicAdd(d + 0, v.x); atomicAdd(d + 1, v.y); atomicAdd(d + 2, v.z); atomicAdd(d + 3, v.w);
                } else {
                    *reinterpret_cast<float4*>(d) = v;
                }
            }
        }
    }
}

// Wide tables (HEp > 2048): a wave takes a QUARTER of a chunk's row (iterations 4 w .. 4 w + 3: one 16-bit half of a
// mask word per pair), so that 16 accumulators per lane replace 64 (130 VGPRs, three waves per SIMD, before).
// TH (bf16 mode, train.hip's edge_tables_bf16): rows written whole by their one chunk go to the bf16 table dtab_h, which
// is what the data- and weight-gradient products read; rows shared by several chunks are still summed with fp32 atomics
// in dtab and converted afterwards (pair_bwd_fix_kernel).
template <bool TH>
__global__ __launch_bounds__(256) void pair_bwd_split_kernel(const int* __restrict__ chunk_off,
                                                             const int* __restrict__ chunk_row,
                                                             const int* __restrict__ grp_pairs, const float* __restrict__ dl,
                                                             const uint32_t* __restrict__ mask, const float* __restrict__ w2,
                                                             const float* __restrict__ scale, float* __restrict__ dtab,
                                                             uint16_t* __restrict__ dtab_h, int n_chunks, int HEp) {
    constexpr int NW = 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long waves_total = (long)gridDim.x * 4;
    const float sc = scale ? scale[0] : 1.f;
    const uint16_t* mask16 = reinterpret_cast<const uint16_t*>(mask);
    for (long item = (long)blockIdx.x * 4 + wave; item < (long)n_chunks * 4; item += waves_total) {
        const int q = (int)(item >> 2), w = (int)(item & 3);  // the four quarters of a chunk run in one workgroup
        const int cr = chunk_row[q];
        const int row = cr & 0x7fffffff;
        float acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const int k1 = chunk_off[q + 1];
        int k = chunk_off[q];
        // four pairs at a time: their index, gradient and mask loads are independent and go out together (one pair per
        // trip left every trip waiting for its own three dependent loads: 3.63 -> 2.67 ms per pass at cfg 5; eight: 2.63)
        for (; k + 3 < k1; k += 4) {
            int pp[4];
            float gg[4];
            uint32_t mm[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) pp[u] = grp_pairs ? grp_pairs[k + u] : k + u;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                gg[u] = dl[pp[u]] * sc;
                mm[u] = mask16[(((long)pp[u] * NW + (w >> 1)) * 64 + lane) * 2 + (w & 1)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] += sel(mm[u], i, gg[u]);
        }
        for (; k < k1; ++k) {
            const int p = grp_pairs ? grp_pairs[k] : k;
            const float g = dl[p] * sc;
            const uint32_t mw = mask16[(((long)p * NW + (w >> 1)) * 64 + lane) * 2 + (w & 1)];
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] += sel(mw, i, g);
        }
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const int c = lane * 4 + 256 * (4 * w + ii);
            if (c < HEp) {
                float4 wv = ld4c(w2 + c);
                float4 v = make_float4(acc[4 * ii] * wv.x, acc[4 * ii + 1] * wv.y, acc[4 * ii + 2] * wv.z, acc[4 * ii + 3] * wv.w);
                float* d = dtab + (long)row * HEp + c;
                if (cr < 0) {
                    atomicAdd(d + 0, v.x); atomicAdd(d + 1, v.y); atomicAdd(d + 2, v.z); atomicAdd(d + 3, v.w);
                } else if (TH) {
                    *reinterpret_cast<uint2*>(dtab_h + (long)row * HEp + c) = f4_to_bf16x4(v);
                } else {
                    *reinterpret_cast<float4*>(d) = v;
                }
            }
        }
    }
}

// bf16 tables: the rows the chunk passes summed with atomics (bit 31 of chunk_row), fp32 -> bf16, both tables in one
// launch.  One wave per chunk; the first chunk of a row converts it.  Wave `nch` stands for the table's last row (the
// no-input row, index NA / NB): zeros unless a chunk names it (chunks are sorted by row).
__global__ __launch_bounds__(256) void pair_bwd_fix_kernel(const int* __restrict__ ca_row, int CA, const float* __restrict__ dea,
                                                           uint16_t* __restrict__ dea_h, int NA, int blocks_a,
                                                           const int* __restrict__ ch_row, int CH, const float* __restrict__ deb,
                                                           uint16_t* __restrict__ deb_h, int NB, int HEp) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int blk = blockIdx.x;
    const int* rowv = ch_row;
    const float* tab = deb;
    uint16_t* tab_h = deb_h;
    int nch = CH, last = NB;
    if (blk < blocks_a) { rowv = ca_row; tab = dea; tab_h = dea_h; nch = CA; last = NA; }
    else blk -= blocks_a;
    const int q = blk * 4 + wave;
    if (q > nch) return;
    if (q == nch) {
        if (nch > 0 && (rowv[nch - 1] & 0x7fffffff) == last) return;  // written by its chunk, or converted below
        uint16_t* h = tab_h + (long)last * HEp;
        for (int c = lane * 4; c < HEp; c += 256) *reinterpret_cast<uint2*>(h + c) = make_uint2(0u, 0u);
        return;
    }
    const int cr = rowv[q];
    if (cr >= 0) return;
    const int row = cr & 0x7fffffff;
    if (q > 0 && (rowv[q - 1] & 0x7fffffff) == row) return;
    const float* d = tab + (long)row * HEp;
    uint16_t* h = tab_h + (long)row * HEp;
    for (int c = lane * 4; c < HEp; c += 256) *reinterpret_cast<uint2*>(h + c) = f4_to_bf16x4(ld4c(d + c));
}

__global__ void scale_inplace_kernel(float* x, const float* s, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] *= s[0];
}

// ---------------------------------------------------------------------------
// vertex-type CE: one thread per row (T <= 32 columns)
__global__ void vertex_ce_kernel(const float* __restrict__ logits, int ld, const int* __restrict__ ce_off,
                                 const int* __restrict__ ce_tgt, float* __restrict__ dlogits,
                                 float* __restrict__ loss_sum, int R, int T, int Tp) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    float local = 0.f;
    if (r < R) {
        const float* x = logits + (long)r * ld;
        float mx = -INFINITY;
        for (int t = 0; t < T; ++t) mx = fmaxf(mx, x[t]);
        float se = 0.f;
        for (int t = 0; t < T; ++t) se += expf(x[t] - mx);
        float lse = mx + logf(se);
        int k0 = ce_off[r], k1 = ce_off[r + 1];
        float cnt = (float)(k1 - k0);
        float* d = dlogits + (long)r * Tp;
        for (int t = 0; t < Tp; ++t) d[t] = t < T ? cnt * expf(x[t] - lse) : 0.f;
        for (int k = k0; k < k1; ++k) {
            int tg = ce_tgt[k];
            local += lse - x[tg];
            d[tg] -= 1.f;
        }
    }
    // block reduction
    local = wave_sum(local);
    __shared__ float red[4];
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        float v = red[0] + red[1] + red[2] + red[3];
        if (v != 0.f) atomicAdd(loss_sum, v);
    }
}

// ---------------------------------------------------------------------------
// z = mu + eps * eps_scale * exp(logvar / 2)  (models.py:733-740), written into a
// zero-padded [B][NZp] buffer; kld = -0.5 sum(1 + logvar - mu^2 - exp(logvar)) (:859)
__global__ void reparam_kld_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ logvar,
                                       const float* __restrict__ eps, float eps_scale, int training,
                                       float* __restrict__ z, float* __restrict__ kld_sum, int B, int nz, int NZp,
                                       float* __restrict__ zero_a, float* __restrict__ zero_b, int HEp) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (zero_a)  // the all-zero rows of the two edge-predictor tables (two memsets otherwise)
        for (int c = i; c < HEp; c += gridDim.x * blockDim.x) {
            zero_a[c] = 0.f;
            zero_b[c] = 0.f;
        }
    float local = 0.f;
    if (i < B * NZp) {
        int b = i / NZp, j = i % NZp;
        float zv = 0.f;
        if (j < nz) {
            float m = mu[b * nz + j], lv = logvar[b * nz + j];
            zv = m;
            if (training) zv = eps[b * nz + j] * eps_scale * expf(0.5f * lv) + m;
            local = -0.5f * (1.f + lv - m * m - expf(lv));
        }
        z[i] = zv;
    }
    local = wave_sum(local);
    __shared__ float red[4];
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        float v = red[0] + red[1] + red[2] + red[3];
        atomicAdd(kld_sum, v);
    }
}

// dmu = g_res-side dz + g_kld * mu ; dlogvar = dz * eps*scale*0.5*std + g_kld * (-0.5)(1 - exp(logvar))
__global__ void reparam_kld_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ logvar,
                                       const float* __restrict__ eps, float eps_scale, int training,
                                       const float* __restrict__ dz, const float* __restrict__ gk,
                                       float* __restrict__ dmu, float* __restrict__ dlogvar, int B, int nz, int NZp) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * nz) return;
    int b = i / nz, j = i % nz;
    float m = mu[i], lv = logvar[i];
    float d = dz[b * NZp + j];
    float g = gk[0];
    float dm = d + g * m;
    float dv = g * (-0.5f) * (1.f - expf(lv));
    if (training) dv += d * eps[i] * eps_scale * 0.5f * expf(0.5f * lv);
    dmu[i] = dm;
    dlogvar[i] = dv;
}

// ---------------------------------------------------------------------------
// Small batches / latent sizes: the head of loss() as ONE single-workgroup launch -- reparameterisation + KLD (no
// atomics: the workgroup owns the sum), the zeroing of the loss accumulators `sums[0..7]` and of the two all-zero
// rows of the edge-predictor tables (three memsets otherwise).
__global__ __launch_bounds__(1024) void loss_fwd_init_kernel(const float* __restrict__ mu, const float* __restrict__ logvar,
                                                            const float* __restrict__ eps, float eps_scale, int training,
                                                            float* __restrict__ z, float* __restrict__ sums,
                                                            float* __restrict__ zero_a, float* __restrict__ zero_b,
                                                            int HEp, int B, int nz, int NZp) {
    float local = 0.f;
    for (int i = threadIdx.x; i < B * NZp; i += 1024) {
        int b = i / NZp, j = i % NZp;
        float zv = 0.f;
        if (j < nz) {
            float m = mu[b * nz + j], lv = logvar[b * nz + j];
            zv = m;
            if (training) zv = eps[b * nz + j] * eps_scale * expf(0.5f * lv) + m;
            local += -0.5f * (1.f + lv - m * m - expf(lv));
        }
        z[i] = zv;
    }
    for (int c = threadIdx.x; c < HEp; c += 1024) {
        zero_a[c] = 0.f;
        zero_b[c] = 0.f;
    }
    local = wave_sum(local);
    __shared__ float red[16];
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = local;
    __syncthreads();
    if (threadIdx.x < 8) {
        float v = 0.f;
        if (threadIdx.x == 2)
            for (int w = 0; w < 16; ++w) v += red[w];
        sums[threadIdx.x] = v;  // {pair loss, CE loss, kld, -, arrival counter of vertex_ce_final, -, -, -}
    }
}

// vertex-type CE whose LAST workgroup to finish also writes the three outputs of loss()
// (out3 = {res + beta kld, res, kld}): the separate finalize launch is gone.  sums[4] is the arrival counter
// (zeroed with the other accumulators); sums are read back with atomics, which execute at the memory side.
__global__ void vertex_ce_final_kernel(const float* __restrict__ logits, int ld, const int* __restrict__ ce_off,
                                       const int* __restrict__ ce_tgt, float* __restrict__ dlogits,
                                       float* __restrict__ sums, unsigned tickets, float beta,
                                       float* __restrict__ out3, int R, int T, int Tp) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    float local = 0.f;
    if (r < R) {
        // Tp == ld == 32 (the launcher checks): the row is 8 float4 requested together, everything after that is
        // register arithmetic (the scalar form was a chain of dependent loads: 12 us for 224 rows at batch 32)
        float x[32];
        const float4* xr = reinterpret_cast<const float4*>(logits + (long)r * ld);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 v = xr[q];
            x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
        }
        const int k0 = ce_off[r], k1 = ce_off[r + 1];
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 32; ++t) mx = t < T ? fmaxf(mx, x[t]) : mx;
        float se = 0.f;
#pragma unroll
        for (int t = 0; t < 32; ++t) se += t < T ? expf(x[t] - mx) : 0.f;  // (in column order, as the scalar form)
        const float lse = mx + logf(se);
        const float cnt = (float)(k1 - k0);
        float d[32];
#pragma unroll
        for (int t = 0; t < 32; ++t) d[t] = t < T ? cnt * expf(x[t] - lse) : 0.f;
        for (int k = k0; k < k1; ++k) {
            const int tg = ce_tgt[k];
            float xt = 0.f;
#pragma unroll
            for (int t = 0; t < 32; ++t) {
                xt = t == tg ? x[t] : xt;
                d[t] -= t == tg ? 1.f : 0.f;
            }
            local += lse - xt;
        }
        float4* dr = reinterpret_cast<float4*>(dlogits + (long)r * Tp);
#pragma unroll
        for (int q = 0; q < 8; ++q) dr[q] = make_float4(d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]);
    }
    local = wave_sum(local);
    __shared__ float red[4];
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        float v = red[0] + red[1] + red[2] + red[3];
        loss_sum_and_ticket(sums + 1, v, sums, tickets, beta, out3);
    }
}

// Head of loss backward as ONE launch (was: grad_scales, scaled_copy and two whole-table memsets):
//   gs = {d/d res, d/d kld} from the upstream gradients g3 of (loss, res, kld)
//   dlogits_s = dlogits * gs[0]
//   rows of the two edge-predictor gradient tables that pair_bwd_kernel ADDS into (rows with several chunks: bit 31
//   of chunk_row) and their last, all-zero-input row are zeroed; every other row is written whole by its one chunk
__global__ __launch_bounds__(256) void loss_bwd_init_kernel(const float* __restrict__ g3, float beta, float* __restrict__ gs,
                                                            const float* __restrict__ dlogits, float* __restrict__ dlogits_s,
                                                            long n_logits, int blocks_copy, const int* __restrict__ ca_row,
                                                            int CA, float* __restrict__ dea, int NA, int blocks_a,
                                                            const int* __restrict__ ch_row, int CH, float* __restrict__ deb,
                                                            int NB, int HEp) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int blk = blockIdx.x;
    if (blk < blocks_copy) {
        const float g0 = g3[0] + g3[1];
        if (blk == 0 && threadIdx.x == 0) {
            gs[0] = g0;
            gs[1] = beta * g3[0] + g3[2];
        }
        for (long i = blk * 256L + threadIdx.x; i < n_logits; i += blocks_copy * 256L) dlogits_s[i] = dlogits[i] * g0;
        return;
    }
    blk -= blocks_copy;
    const int* rowv = ch_row;
    float* tab = deb;
    int nch = CH, last = NB;
    if (blk < blocks_a) { rowv = ca_row; tab = dea; nch = CA; last = NA; }
    else blk -= blocks_a;
    // one wave per chunk; chunk nch stands for the table's last row
    const int q = blk * 4 + wave;
    if (q > nch) return;
    int row = last;
    if (q < nch) {
        const int cr = rowv[q];
        if (cr >= 0) return;  // single-chunk row: written whole by its chunk
        row = cr & 0x7fffffff;
        if (q > 0 && (rowv[q - 1] & 0x7fffffff) == row) return;  // not the row's first chunk
    }
    float* d = tab + (long)row * HEp;
    for (int c = lane * 4; c < HEp; c += 256) *reinterpret_cast<float4*>(d + c) = make_float4(0.f, 0.f, 0.f, 0.f);
}

}  // namespace dvae

using namespace dvae;

static int pair_nit(int HEp) {
    int it = cdiv(HEp, 256);
    return it <= 1 ? 1 : it <= 2 ? 2 : it <= 4 ? 4 : it <= 8 ? 8 : it <= 16 ? 16 : -1;
}

extern "C" int64_t dvae_pair_mask_words(int32_t n_pairs, int32_t HEp) {
    int nit = pair_nit(HEp);
    if (nit < 0 || n_pairs < 0) return -1;
    return (int64_t)n_pairs * ((nit + 7) / 8) * 64;
}

namespace dvae {
int pair_bce_fwd(const float* ea, const float* eb, const float* ec, const int32_t* pair_a, const int32_t* pair_h,
                 const int32_t* pair_c, const uint8_t* pair_t, const float* b1, const float* w2, const float* b2, float* dl,
                 float* loss_sum, uint32_t* mask, float* dw2_partial, int32_t n_blocks, int32_t n_pairs, int32_t HEp,
                 float* sums, unsigned tickets, float beta, float* out3, void* stream, int group, int tables_bf16) {
    DVAE_REQUIRE(ea && eb && pair_a && pair_h && pair_t && b1 && w2 && b2 && dl && loss_sum, "pair_bce_fwd: null pointer");
    DVAE_REQUIRE((ec == nullptr) == (pair_c == nullptr), "pair_bce_fwd: ec and pair_c go together");
    DVAE_REQUIRE(HEp > 0 && HEp % 4 == 0 && n_blocks > 0, "pair_bce_fwd: HEp must be a multiple of 4, n_blocks > 0");
    const int nit = pair_nit(HEp);
    DVAE_REQUIRE(nit > 0, "pair_bce_fwd: HEp %d exceeds 4096", HEp);
    if (n_pairs == 0 && !dw2_partial && !sums) return DVAE_OK;
    const size_t shm = ((size_t)HEp + 8) * sizeof(float) + 2 * 4 * 64 * sizeof(uint16_t);  // (+ the split kernel's mask halves)
    dim3 grid(n_blocks), block(256);
    hipStream_t s = (hipStream_t)stream;
    static const int knob_split = getenv("DVAE_PAIR_SPLIT") ? atoi(getenv("DVAE_PAIR_SPLIT")) : 1;  // 0: one wave per pair at every width
    if (nit == 16 && knob_split) {  // wide tables: four waves per pair
#define DVAE_PAIR_SPLIT(TH_, FULL_)                                                                                    \
    hipLaunchKernelGGL((pair_fwd_split_kernel<4, TH_, FULL_>), grid, block, shm, s, ea, eb, ec, pair_a, pair_h, pair_c, pair_t, \
                       b1, w2, b2, dl, loss_sum, mask, dw2_partial, n_pairs, HEp, sums, tickets, beta, out3, group)
        const bool full = HEp == 4096;
        if (tables_bf16) {
            DVAE_REQUIRE(!ec, "pair_bce_fwd: bf16-stored tables are not built for the three-table (DVAE_BN) form");
            if (full) DVAE_PAIR_SPLIT(true, true);
            else DVAE_PAIR_SPLIT(true, false);
        } else {
            if (full) DVAE_PAIR_SPLIT(false, true);
            else DVAE_PAIR_SPLIT(false, false);
        }
#undef DVAE_PAIR_SPLIT
        DVAE_LAUNCH_CHECK();
        return DVAE_OK;
    }
    DVAE_REQUIRE(!tables_bf16, "pair_bce_fwd: bf16-stored tables need the wide-table kernel (2048 < HEp <= 4096)");
#define DVAE_PAIR_FWD(N)                                                                                              \
    hipLaunchKernelGGL((pair_fwd_kernel<N>), grid, block, shm, s, ea, eb, ec, pair_a, pair_h, pair_c, pair_t, b1, w2, \
                       b2, dl, loss_sum, mask, dw2_partial, n_pairs, HEp, sums, tickets, beta, out3, group)
    switch (nit) {
        case 1: DVAE_PAIR_FWD(1); break;
        case 2: DVAE_PAIR_FWD(2); break;
        case 4: DVAE_PAIR_FWD(4); break;
        case 8: DVAE_PAIR_FWD(8); break;
        default: DVAE_PAIR_FWD(16); break;
    }
#undef DVAE_PAIR_FWD
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}
}  // namespace dvae

extern "C" int dvae_pair_bce_fwd(const float* ea, const float* eb, const float* ec, const int32_t* pair_a,
                                 const int32_t* pair_h, const int32_t* pair_c, const uint8_t* pair_t, const float* b1,
                                 const float* w2, const float* b2, float* dl, float* loss_sum, uint32_t* mask,
                                 float* dw2_partial, int32_t n_blocks, int32_t n_pairs, int32_t HEp, void* stream) {
    return dvae::pair_bce_fwd(ea, eb, ec, pair_a, pair_h, pair_c, pair_t, b1, w2, b2, dl, loss_sum, mask, dw2_partial, n_blocks,
                              n_pairs, HEp, nullptr, 0u, 0.f, nullptr, stream, 0, 0);
}

namespace dvae {
int pair_bce_bwd_grouped(const int32_t* chunk_off, const int32_t* chunk_row, const int32_t* grp_pairs, const float* dl,
                         const uint32_t* mask, const float* w2, const float* scale, float* dtab, int32_t n_chunks,
                         int32_t HEp, hipStream_t stream, uint16_t* dtab_h);
int pair_bwd_fix(const int* ca_row, int CA, const float* dea, uint16_t* dea_h, int NA, const int* ch_row, int CH,
                 const float* deb, uint16_t* deb_h, int NB, int HEp, hipStream_t s) {
    const int blocks_a = cdiv(CA + 1, 4), blocks_h = cdiv(CH + 1, 4);
    DVAE_REQUIRE(HEp % 4 == 0 && (uintptr_t)dea_h % 8 == 0 && (uintptr_t)deb_h % 8 == 0, "pair_bwd_fix: misaligned table");
    hipLaunchKernelGGL(pair_bwd_fix_kernel, dim3(blocks_a + blocks_h), dim3(256), 0, s, ca_row, CA, dea, dea_h, NA, blocks_a,
                       ch_row, CH, deb, deb_h, NB, HEp);
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}
}  // namespace dvae

extern "C" int dvae_pair_bce_bwd_grouped(const int32_t* chunk_off, const int32_t* chunk_row, const int32_t* grp_pairs,
                                         const float* dl, const uint32_t* mask, const float* w2, const float* scale,
                                         float* dtab, int32_t n_chunks, int32_t HEp, void* stream) {
    return dvae::pair_bce_bwd_grouped(chunk_off, chunk_row, grp_pairs, dl, mask, w2, scale, dtab, n_chunks, HEp,
                                      (hipStream_t)stream, nullptr);
}

// dtab_h (nullable; wide tables only): rows with one chunk are written there as bf16 instead (pair_bwd_fix converts the
// others afterwards)
int dvae::pair_bce_bwd_grouped(const int32_t* chunk_off, const int32_t* chunk_row, const int32_t* grp_pairs,
                               const float* dl, const uint32_t* mask, const float* w2, const float* scale, float* dtab,
                               int32_t n_chunks, int32_t HEp, hipStream_t stream, uint16_t* dtab_h) {
    if (n_chunks == 0) return DVAE_OK;
    DVAE_REQUIRE(chunk_off && chunk_row && dl && mask && w2 && dtab, "pair_bce_bwd: null pointer");
    DVAE_REQUIRE(HEp > 0 && HEp % 4 == 0, "pair_bce_bwd: bad size");
    const int nit = pair_nit(HEp);
    DVAE_REQUIRE(nit > 0, "pair_bce_bwd: HEp %d exceeds 4096", HEp);
    int blocks = cdiv(n_chunks, 4);
    if (blocks > 8192) blocks = 8192;
    dim3 grid(blocks), block(256);
    hipStream_t s = (hipStream_t)stream;
    static const int knob_split = getenv("DVAE_PAIR_SPLIT") ? atoi(getenv("DVAE_PAIR_SPLIT")) : 1;
    if (nit == 16 && knob_split) {  // wide tables: a quarter row per wave
        grid.x = n_chunks < 16384 ? n_chunks : 16384;
        if (dtab_h) {
            DVAE_REQUIRE((uintptr_t)dtab_h % 8 == 0, "pair_bce_bwd: misaligned bf16 table");
            hipLaunchKernelGGL(pair_bwd_split_kernel<true>, grid, block, 0, s, chunk_off, chunk_row, grp_pairs, dl, mask, w2,
                               scale, dtab, dtab_h, n_chunks, HEp);
        } else {
            hipLaunchKernelGGL(pair_bwd_split_kernel<false>, grid, block, 0, s, chunk_off, chunk_row, grp_pairs, dl, mask, w2,
                               scale, dtab, dtab_h, n_chunks, HEp);
        }
        DVAE_LAUNCH_CHECK();
        return DVAE_OK;
    }
    DVAE_REQUIRE(!dtab_h, "pair_bce_bwd: bf16 tables are built for the wide-table kernel only");
#define DVAE_PAIR_BWD(N)                                                                                            \
    hipLaunchKernelGGL((pair_bwd_kernel<N>), grid, block, 0, s, chunk_off, chunk_row, grp_pairs, dl, mask, w2, scale, \
                       dtab, n_chunks, HEp)
    switch (nit) {
        case 1: DVAE_PAIR_BWD(1); break;
        case 2: DVAE_PAIR_BWD(2); break;
        case 4: DVAE_PAIR_BWD(4); break;
        case 8: DVAE_PAIR_BWD(8); break;
        default: DVAE_PAIR_BWD(16); break;
    }
#undef DVAE_PAIR_BWD
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}

extern "C" int dvae_vertex_ce(const float* logits, int32_t ld, const int32_t* ce_off, const int32_t* ce_tgt,
                              float* dlogits, float* loss_sum, int32_t R, int32_t T, int32_t Tp, void* stream) {
    if (R == 0) return DVAE_OK;
    DVAE_REQUIRE(logits && ce_off && ce_tgt && dlogits && loss_sum, "vertex_ce: null pointer");
    DVAE_REQUIRE(T > 0 && T <= Tp && ld >= T, "vertex_ce: bad T/Tp/ld");
    hipLaunchKernelGGL(vertex_ce_kernel, dim3(cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream, logits, ld, ce_off,
                       ce_tgt, dlogits, loss_sum, R, T, Tp);
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}

namespace dvae {
int loss_fwd_init(const float* mu, const float* logvar, const float* eps, float eps_scale, int training, float* z,
                  float* sums, float* zero_row_a, float* zero_row_b, int HEp, int B, int nz, int NZp, hipStream_t s) {
    hipLaunchKernelGGL(loss_fwd_init_kernel, dim3(1), dim3(1024), 0, s, mu, logvar, eps, eps_scale, training, z, sums,
                       zero_row_a, zero_row_b, HEp, B, nz, NZp);
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}
int vertex_ce_blocks(int R) { return cdiv(R, 256); }
int vertex_ce_final(const float* logits, int ld, const int* ce_off, const int* ce_tgt, float* dlogits, float* sums,
                    unsigned tickets, float beta, float* out3, int R, int T, int Tp, hipStream_t s) {
    DVAE_REQUIRE(R > 0 && T > 0 && T <= Tp && Tp == 32 && ld == 32, "vertex_ce_final: needs 0 < T <= Tp == ld == 32 (R=%d T=%d Tp=%d ld=%d)", R, T, Tp, ld);
    DVAE_REQUIRE(((uintptr_t)logits % 16) == 0 && ((uintptr_t)dlogits % 16) == 0, "vertex_ce_final: logits / dlogits must be 16-byte aligned");
    hipLaunchKernelGGL(vertex_ce_final_kernel, dim3(vertex_ce_blocks(R)), dim3(256), 0, s, logits, ld, ce_off, ce_tgt, dlogits,
                       sums, tickets, beta, out3, R, T, Tp);
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}
int loss_bwd_init(const float* g3, float beta, float* gs, const float* dlogits, float* dlogits_s, long n_logits,
                  const int* ca_row, int CA, float* dea, int NA, const int* ch_row, int CH, float* deb, int NB, int HEp,
                  hipStream_t s) {
    int blocks_copy = cdiv(n_logits, 256 * 4);
    if (blocks_copy < 1) blocks_copy = 1;
    if (blocks_copy > 512) blocks_copy = 512;
    const int blocks_a = cdiv(CA + 1, 4), blocks_b = cdiv(CH + 1, 4);
    hipLaunchKernelGGL(loss_bwd_init_kernel, dim3(blocks_copy + blocks_a + blocks_b), dim3(256), 0, s, g3, beta, gs, dlogits,
                       dlogits_s, n_logits, blocks_copy, ca_row, CA, dea, NA, blocks_a, ch_row, CH, deb, NB, HEp);
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}
int scale_inplace(float* x, const float* s, int n, hipStream_t st) {
    if (n <= 0) return DVAE_OK;
    hipLaunchKernelGGL(scale_inplace_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, x, s, n);
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}
int reparam_kld_fwd(const float* mu, const float* logvar, const float* eps, float eps_scale, int training, float* z,
                    float* kld_sum, int B, int nz, int NZp, hipStream_t s, float* zero_a, float* zero_b, int HEp) {
    hipLaunchKernelGGL(reparam_kld_fwd_kernel, dim3(cdiv((long)B * NZp, 256)), dim3(256), 0, s, mu, logvar, eps,
                       eps_scale, training, z, kld_sum, B, nz, NZp, zero_a, zero_b, HEp);
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}
int reparam_kld_bwd(const float* mu, const float* logvar, const float* eps, float eps_scale, int training,
                    const float* dz, const float* gk, float* dmu, float* dlogvar, int B, int nz, int NZp, hipStream_t s) {
    hipLaunchKernelGGL(reparam_kld_bwd_kernel, dim3(cdiv((long)B * nz, 256)), dim3(256), 0, s, mu, logvar, eps,
                       eps_scale, training, dz, gk, dmu, dlogvar, B, nz, NZp);
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}
}  // namespace dvae
