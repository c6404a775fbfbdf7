// Orchestration of the D-VAE train-step hot path on one GPU:
//   dvae_encode_forward / _backward   models.py:720-731 (BN :1068-1080) and its autograd
//   dvae_loss_forward  / _backward    models.py:806-860 and its autograd
// Every function only enqueues kernels on the caller's stream, works out of a
// caller-provided workspace (sizes from dvae_*_ws_floats) and never allocates.
//
// Level-batched evaluation: level == vertex index,
// slot(v, b) = v*B + b, all per-vertex tensors are [n*B][..] level-major so a level
// is a contiguous row range.  Dependent kernel launches on one stream are the
// level barrier (cheaper on MI355X than an in-kernel grid barrier: ~1.5 us vs
// ~4-7 us, MI355X_MICROARCH.md price list rows `boundary` / `barrier-xcd`).
#include "common.h"
#include "model.h"
#include "batch.h"
#include "orch.h"
#include <mutex>

namespace dvae {

// ------------------------------------------------------------------ small kernels
__global__ void count_kernel(float* cnt, const int* off, const int* idx, int rows, int Tp) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * Tp) return;
    int r = i / Tp, t = i % Tp;
    int c = 0;
    for (int k = off[r]; k < off[r + 1]; ++k) c += idx[k] == t;
    cnt[i] = (float)c;
}
// out12 [B][2NZp] -> mu, logvar [B][nz]
__global__ void split_mu_logvar_kernel(const float* out12, float* mu, float* logvar, int B, int nz, int NZp,
                                       const unsigned* failed) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * nz) return;
    int b = i / nz, j = i % nz;
    const float poison = (failed && *failed) ? __builtin_nanf("") : 0.f;  // see dvae_encode_forward
    mu[i] = out12[(long)b * 2 * NZp + j] + poison;
    logvar[i] = out12[(long)b * 2 * NZp + NZp + j] + poison;
}
__global__ void join_mu_logvar_kernel(float* d12, const float* dmu, const float* dlogvar, int B, int nz, int NZp) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 2 * NZp) return;
    int b = i / (2 * NZp), c = i % (2 * NZp);
    float v = 0.f;
    if (c < nz) v = dmu[b * nz + c];
    else if (c >= NZp && c < NZp + nz) v = dlogvar[b * nz + (c - NZp)];
    d12[i] = v;
}
// d *= (1 - y^2)
__global__ void tanh_bwd_kernel(float* d, const float* y, long n) {
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i < n) d[i] *= (1.f - y[i] * y[i]);
}
__global__ void add_inplace_kernel(float* d, const float* a, long n) {
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i < n) d[i] += a[i];
}
__global__ void scaled_copy_kernel(float* d, const float* x, const float* s, long n) {
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i < n) d[i] = x[i] * s[0];
}
// sums = {pair loss, ce loss, kld, -} -> out3 = {res + beta kld, res, kld}
__global__ void finalize_loss_kernel(const float* sums, float beta, float* out3) {
    float res = sums[0] + sums[1];
    out3[0] = res + beta * sums[2];
    out3[1] = res;
    out3[2] = sums[2];
}
// g3 = upstream grads of (loss, res, kld) -> gs = {d/d res, d/d kld}
__global__ void grad_scales_kernel(const float* g3, float beta, float* gs) {
    gs[0] = g3[0] + g3[1];
    gs[1] = beta * g3[0] + g3[2];
}
// *out = scale * sum(x[0:n])
__global__ void sum_scaled_kernel(const float* x, int n, const float* scale, float* out) {
    float v = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) v += x[i];
    v = wave_sum(v);
    __shared__ float red[4];
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (red[0] + red[1] + red[2] + red[3]) * scale[0]);
}

#define EW(kern, total, ...)                                                                             \
    do {                                                                                                 \
        long t__ = (total);                                                                              \
        if (t__ > 0) {                                                                                   \
            hipLaunchKernelGGL(kern, dim3(cdiv(t__, 256)), dim3(256), 0, s, __VA_ARGS__);                \
            DVAE_LAUNCH_CHECK();                                                                         \
        }                                                                                                \
    } while (0)

// ------------------------------------------------------------------ side stream
// The two encoder directions are independent chains of small dependent launches
// (a level's GEMM fills at most half the chip).  They run concurrently: direction 0 on
// the caller's stream, direction 1 on a non-blocking side stream, forked and joined with
// events, so the function's contract (all work ordered on the caller's stream when it
// returns) is unchanged.
// One side stream and event pair exists per (device, caller stream): calls on different caller
// streams, from one host thread or several, never share events, so one call's fork cannot be
// overwritten between its record and its wait by another call's (which would order that call's
// side work behind the wrong point).  Calls on the SAME caller stream share the entry; their
// records land on the one stream in issue order, so a wait can only pick up a LATER point of the
// same stream: over-ordered, never under-ordered.  Entries are created on first use and recycled
// least-recently-used beyond SIDE_CAP (hipStreamDestroy lets enqueued work finish).
struct Side {
    int dev = -1;
    hipStream_t caller = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    // third stream: the weight-gradient chunks of the encoder's backward chains (both other streams walk a chain there)
    hipStream_t aux = nullptr;
    hipEvent_t aux_ev = nullptr, aux_join = nullptr;
    unsigned long last_use = 0;
    int users = 0;  // calls in progress that hold this entry (under g_side_mu): an entry in use is never recycled
};
constexpr int SIDE_CAP = 64;
static std::mutex g_side_mu;
static Side* g_sides[SIDE_CAP];
static int g_nsides = 0;
static unsigned long g_side_clock = 0;

// One call's hold on a Side entry.  The per-call state (is a fork open? is work pending on the aux stream?) lives HERE,
// on the caller's stack, not in the shared entry: two host threads that share a caller stream (e.g. the null stream)
// share the entry's streams and events -- their records land on the one stream in issue order, so a wait can only pick
// up a later point of the same stream: over-ordered, never under-ordered -- but neither can clear or join the other's
// state.  The destructor joins what an error path left open (whatever was enqueued on the side streams stays ordered
// before the caller's later work) and releases the entry.
struct SideGuard {
    Side* sd = nullptr;
    hipStream_t main = nullptr;
    bool forked = false;    // set by fork_side, cleared by join_side
    bool aux_open = false;  // work enqueued on aux since the last join_aux
    bool inline_mode = false;  // DVAE_SIDE_STREAMS=0: the "side" streams ARE the caller's stream
    explicit SideGuard(hipStream_t m) : main(m) {}
    SideGuard(const SideGuard&) = delete;
    SideGuard& operator=(const SideGuard&) = delete;
    ~SideGuard();
};

// DVAE_SIDE_STREAMS=0: no side streams at all -- every "side" launch goes to the caller's stream in issue order (a
// linear launch sequence; A/B runs, and captured graphs without forks)
static bool side_streams_on() {
    static const int knob = [] { const char* v = getenv("DVAE_SIDE_STREAMS"); return v ? atoi(v) : 1; }();
    return knob != 0;
}
static int side_acquire(SideGuard& g) {
    if (g.sd) return DVAE_OK;
    hipStream_t caller = g.main;
    if (!side_streams_on()) {
        thread_local Side inline_side;  // stream == aux == the caller's stream; no events are ever used
        inline_side.caller = inline_side.stream = inline_side.aux = caller;
        inline_side.users = 1 << 30;
        g.sd = &inline_side;
        g.inline_mode = true;
        return DVAE_OK;
    }
    int dev = 0;
    DVAE_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_side_mu);
    Side* sd = nullptr;
    for (int i = 0; i < g_nsides; ++i)
        if (g_sides[i]->dev == dev && g_sides[i]->caller == caller) sd = g_sides[i];
    if (!sd) {
        if (g_nsides < SIDE_CAP) {
            sd = g_sides[g_nsides++] = new Side();
        } else {  // recycle the idle entry that has been idle longest (its stream keeps running what it holds)
            for (int i = 0; i < g_nsides; ++i)
                if (g_sides[i]->users == 0 && (!sd || g_sides[i]->last_use < sd->last_use)) sd = g_sides[i];
            DVAE_REQUIRE(sd != nullptr, "more than %d caller streams with calls in progress at once", SIDE_CAP);
            int cur = dev;
            if (sd->dev != cur) DVAE_HIP(hipSetDevice(sd->dev));
            (void)hipStreamDestroy(sd->stream);
            (void)hipEventDestroy(sd->fork);
            (void)hipEventDestroy(sd->join);
            (void)hipStreamDestroy(sd->aux);
            (void)hipEventDestroy(sd->aux_ev);
            (void)hipEventDestroy(sd->aux_join);
            if (sd->dev != cur) DVAE_HIP(hipSetDevice(cur));
            *sd = Side();
        }
        sd->dev = dev;
        sd->caller = caller;
        static const int prio_knob = [] { const char* v = getenv("DVAE_SIDE_PRIO"); return v ? atoi(v) : 0; }();
        int plo = 0, phi = 0;  // numerically greatest = lowest priority
        DVAE_HIP(hipDeviceGetStreamPriorityRange(&plo, &phi));
        // Fork / join events order kernels of two streams of ONE device.  A default event makes its record a
        // system-scope fence (cache write-back and invalidation, hip_runtime_api.h: hipEventDisableSystemFence) that
        // nothing here needs -- no host and no other device reads between a fork and its join -- and that the next
        // kernels pay for with a cold L2.  DVAE_EVENT_FENCE=1 restores the default (A/B runs).
        static const int fence_knob = [] { const char* v = getenv("DVAE_EVENT_FENCE"); return v ? atoi(v) : 0; }();
        const unsigned evf = hipEventDisableTiming | (fence_knob ? 0u : (unsigned)hipEventDisableSystemFence);
        if (prio_knob & 1) DVAE_HIP(hipStreamCreateWithPriority(&sd->stream, hipStreamNonBlocking, plo));
        else DVAE_HIP(hipStreamCreateWithFlags(&sd->stream, hipStreamNonBlocking));
        DVAE_HIP(hipEventCreateWithFlags(&sd->fork, evf));
        DVAE_HIP(hipEventCreateWithFlags(&sd->join, evf));
        if (prio_knob & 2) DVAE_HIP(hipStreamCreateWithPriority(&sd->aux, hipStreamNonBlocking, plo));
        else DVAE_HIP(hipStreamCreateWithFlags(&sd->aux, hipStreamNonBlocking));
        DVAE_HIP(hipEventCreateWithFlags(&sd->aux_ev, evf));
        DVAE_HIP(hipEventCreateWithFlags(&sd->aux_join, evf));
    }
    sd->last_use = ++g_side_clock;
    sd->users++;
    g.sd = sd;
    return DVAE_OK;
}
static int fork_side(SideGuard& g) {
    if (g.inline_mode) return DVAE_OK;
    DVAE_HIP(hipEventRecord(g.sd->fork, g.main));
    DVAE_HIP(hipStreamWaitEvent(g.sd->stream, g.sd->fork, 0));
    g.forked = true;
    return DVAE_OK;
}
static int join_side(SideGuard& g) {
    g.forked = false;
    if (g.inline_mode) return DVAE_OK;
    DVAE_HIP(hipEventRecord(g.sd->join, g.sd->stream));
    DVAE_HIP(hipStreamWaitEvent(g.main, g.sd->join, 0));
    return DVAE_OK;
}
// aux stream: `after(producer)` orders everything enqueued on aux from now on behind what `producer` holds now
// (a stream wait captures the event's latest record at the time of the call, so one event serves every hand-off)
static int aux_after(SideGuard& g, hipStream_t producer) {
    if (g.inline_mode) return DVAE_OK;
    DVAE_HIP(hipEventRecord(g.sd->aux_ev, producer));
    DVAE_HIP(hipStreamWaitEvent(g.sd->aux, g.sd->aux_ev, 0));
    g.aux_open = true;
    return DVAE_OK;
}
static int join_aux(SideGuard& g) {
    if (!g.aux_open || g.inline_mode) return DVAE_OK;
    g.aux_open = false;
    DVAE_HIP(hipEventRecord(g.sd->aux_join, g.sd->aux));
    DVAE_HIP(hipStreamWaitEvent(g.main, g.sd->aux_join, 0));
    return DVAE_OK;
}
SideGuard::~SideGuard() {
    if (!sd || inline_mode) return;
    if (forked) join_side(*this);
    if (aux_open) join_aux(*this);
    std::lock_guard<std::mutex> lock(g_side_mu);
    sd->users--;
}

struct EncWs {
    float *agg[2], *gates[2], *ghn[2], *hs[2], *gm[2], *msg[2];
    float *hg, *hgu, *out12;
    float* xoh;  // [nB][Tp] floats, used as int storage: rows sorted by vertex type (off[T+1] at 0, idx[nB] at 40)
    // backward
    float *dhs[2], *dgm[2], *dgh[2], *dgin[2], *dagg[2], *cnt[2];
    float *d12, *dhgu, *dhg;
    float* sync;  // counters of the persistent chain kernel (chain.hip: enc_chain_fwd), used as unsigned storage
    uint16_t *hs_h[2], *agg_h[2];  // bf16 shadows of hs / agg (act_shadows(): what the forward products read), or null
    uint16_t *dgh_h[2], *dgm_h[2]; // ... of d gh / d gm (what the backward chains' data-gradient products read)
    size_t total;
};
// bf16 mode, levels large enough for the bf16 product kernels (above the exact-fp32 small-level kernels), DVAE: the
// forward chains' states and predecessor sums get bf16 shadows (pointwise.hip: st4h) that the products read
// (DVAE_ACT_BF16=0: fp32 activations converted on their way into LDS, as in round 2; =2: at every batch size -- tests)
static bool act_shadows(const Dims& d, int B) {
    static const int knob = [] { const char* v = getenv("DVAE_ACT_BF16"); return v ? atoi(v) : 1; }();
    // (the products read a shadow only beside the bf16 weight copy: DVAE_BF16_WEIGHTS=0 turns the shadows off too)
    return knob && bf16_weight_copies_on() && d.bf16 && !d.bn && (B > 256 || knob == 2) && d.Hp % 8 == 0;
}
static EncWs carve_enc(const Dims& d, const dvae_batch_counts& c, float* base) {
    EncWs w;
    Bump b{base, 0};
    const size_t nB = c.nB, Hp = d.Hp;
    const int dirs = d.bidir ? 2 : 1;
    for (int k = 0; k < 2; ++k) {
        bool on = k < dirs;
        w.agg[k] = on ? b.take(nB * Hp) : nullptr;
        w.gates[k] = on ? b.take(nB * 3 * Hp) : nullptr;
        w.ghn[k] = on ? b.take(nB * Hp) : nullptr;
        w.hs[k] = on ? b.take(nB * Hp) : nullptr;
        w.gm[k] = (on && !d.bn) ? b.take(nB * 2 * Hp) : nullptr;
        w.msg[k] = (on && !d.bn) ? b.take(nB * Hp) : nullptr;
    }
    w.hg = b.take((size_t)c.B * 2 * Hp);
    w.hgu = b.take((size_t)c.B * Hp);
    w.out12 = b.take((size_t)c.B * 2 * d.NZp);
    w.xoh = b.take(nB * d.Tp);
    for (int k = 0; k < 2; ++k) {
        bool on = k < dirs;
        w.dhs[k] = on ? b.take(nB * Hp) : nullptr;
        w.dgm[k] = (on && !d.bn) ? b.take(nB * 2 * Hp) : nullptr;
        w.dgh[k] = on ? b.take(nB * 3 * Hp) : nullptr;
        w.dgin[k] = on ? b.take(nB * Hp) : nullptr;
        w.dagg[k] = on ? b.take(nB * Hp) : nullptr;
        w.cnt[k] = (on && d.bn) ? b.take(nB * d.Tp) : nullptr;
    }
    w.d12 = b.take((size_t)c.B * 2 * d.NZp);
    w.dhgu = b.take((size_t)c.B * Hp);
    w.dhg = b.take((size_t)c.B * 2 * Hp);
    w.sync = b.take(ENC_CHAIN_SYNC_WORDS);  // (>= 2 TINY_CHAIN_SYNC_WORDS: slot 0 forward, slot 1 backward launch)
    for (int k = 0; k < 2; ++k) {
        const bool on = k < dirs && act_shadows(d, c.B);
        w.hs_h[k] = on ? reinterpret_cast<uint16_t*>(b.take(nB * Hp / 2)) : nullptr;
        w.agg_h[k] = on ? reinterpret_cast<uint16_t*>(b.take(nB * Hp / 2)) : nullptr;
        w.dgh_h[k] = on ? reinterpret_cast<uint16_t*>(b.take(nB * 3 * Hp / 2)) : nullptr;
        w.dgm_h[k] = on ? reinterpret_cast<uint16_t*>(b.take(nB * 2 * Hp / 2)) : nullptr;
    }
    w.total = b.off;
    return w;
}

struct DecWs {
    float *z, *agg, *gates, *ghn, *st, *gm, *msg;
    float *ea, *eb, *ec, *dl, *vs, *v1, *logits, *dlogits, *sums, *xoh, *cnt;
    uint32_t* mask;  // ReLU masks of the edge predictor, one bit per (pair, hidden unit)
    // backward
    float *dlogits_s, *dea, *deb, *dec, *dw2p, *dst, *dv1, *dvs, *dgh, *dgin, *dagg, *dgm, *dh0c, *dz, *gs;
    float* sync;  // counters of the persistent small-batch chain (chain_persist.hip), used as unsigned storage
    uint16_t *st_h, *agg_h;  // bf16 shadows of st / agg (act_shadows()), or null
    uint16_t *dgh_h, *dgm_h; // ... of d gh / d gm
    int dw2_blocks;
    size_t total;
};
// bf16 mode, wide edge-predictor tables (2048 < HEp <= 4096: the four-waves-per-pair kernels), DVAE: EA / EB stored as bf16
// (DVAE_TABLES_BF16=0: fp32 as in round 2)
static bool edge_tables_bf16(const Dims& d) {
    static const int knob = [] { const char* v = getenv("DVAE_TABLES_BF16"); return v ? atoi(v) : 1; }();
    static const int knob_split = [] { const char* v = getenv("DVAE_PAIR_SPLIT"); return v ? atoi(v) : 1; }();
    return knob && knob_split && d.bf16 && !d.bn && d.HEp > 2048 && d.HEp <= 4096 && d.HEp % 8 == 0;
}
// ... and then d EA / d EB too (DVAE_DTABLES_BF16=0: fp32): the pair backward writes bf16 rows, which the data- and
// weight-gradient products and the bias column sum read.  They live in the upper halves of ea / eb's fp32-sized
// allocations, which the bf16 tables leave unused.  Rows summed over several chunks still go through the fp32 tables.
static bool edge_dtables_bf16(const Dims& d) {
    static const int knob = [] { const char* v = getenv("DVAE_DTABLES_BF16"); return v ? atoi(v) : 1; }();
    return knob && bf16_weight_copies_on() && edge_tables_bf16(d);
}
static DecWs carve_dec(const Dims& d, const dvae_batch_counts& c, float* base) {
    DecWs w;
    Bump b{base, 0};
    const size_t nB = c.nB, Hp = d.Hp, NST = (size_t)c.nB + c.PT, HEp = d.HEp;
    w.z = b.take((size_t)c.B * d.NZp);
    w.agg = b.take(NST * Hp);
    w.gates = b.take(NST * 3 * Hp);
    w.ghn = b.take(NST * Hp);
    w.st = b.take(NST * Hp);
    w.gm = d.bn ? nullptr : b.take(nB * 2 * Hp);
    w.msg = d.bn ? nullptr : b.take(nB * Hp);
    w.ea = b.take((size_t)(c.NA + 1) * HEp);
    w.eb = b.take((size_t)(c.NB + 1) * HEp);
    w.ec = d.bn ? b.take((size_t)c.B * HEp) : nullptr;
    w.dl = b.take(c.NP);
    w.mask = reinterpret_cast<uint32_t*>(b.take((size_t)dvae_pair_mask_words(c.NP, d.HEp)));
    w.vs = b.take((size_t)c.R * Hp);
    w.v1 = b.take((size_t)c.R * d.HVp);
    w.logits = b.take((size_t)c.R * d.Tp);
    w.dlogits = b.take((size_t)c.R * d.Tp);
    w.sums = b.take(8);
    w.xoh = b.take(NST * d.Tp);
    w.cnt = d.bn ? b.take(NST * d.Tp) : nullptr;
    w.dlogits_s = b.take((size_t)c.R * d.Tp);
    w.dea = b.take((size_t)(c.NA + 1) * HEp);
    w.deb = b.take((size_t)(c.NB + 1) * HEp);
    w.dec = d.bn ? b.take((size_t)c.B * HEp) : nullptr;
    // (the pair forward's workgroups: four per CU; five where its bf16-table form, 90 registers, runs)
    w.dw2_blocks = edge_tables_bf16(d) ? 1280 : 1024;
    w.dw2p = b.take((size_t)w.dw2_blocks * HEp);
    w.dst = b.take(NST * Hp);
    w.dv1 = b.take((size_t)c.R * d.HVp);
    w.dvs = b.take((size_t)c.R * Hp);
    w.dgh = b.take(NST * 3 * Hp);
    w.dgin = b.take(NST * Hp);
    w.dagg = b.take(NST * Hp);
    w.dgm = d.bn ? nullptr : b.take(nB * 2 * Hp);
    w.dh0c = d.bn ? b.take((size_t)c.B * Hp) : nullptr;
    w.dz = b.take((size_t)c.B * d.NZp);
    w.gs = b.take(8);
    w.sync = b.take(2 * TINY_CHAIN_SYNC_WORDS);  // slot 0: the forward launch, slot 1: the backward launch
    const bool sh = act_shadows(d, c.B);
    w.st_h = sh ? reinterpret_cast<uint16_t*>(b.take(NST * Hp / 2)) : nullptr;
    w.agg_h = sh ? reinterpret_cast<uint16_t*>(b.take(NST * Hp / 2)) : nullptr;
    w.dgh_h = sh ? reinterpret_cast<uint16_t*>(b.take(NST * 3 * Hp / 2)) : nullptr;
    w.dgm_h = (sh && !d.bn) ? reinterpret_cast<uint16_t*>(b.take(nB * 2 * Hp / 2)) : nullptr;
    w.total = b.off;
    return w;
}

static int check_batch(const Dims& d, const dvae_batch_counts* c) {
    DVAE_REQUIRE(c != nullptr, "null batch counts");
    DVAE_REQUIRE(c->n == d.n && c->T == d.T && (c->is_bn != 0) == (d.bn != 0),
                 "batch was packed for a different model (max_n %d vs %d, nvt %d vs %d, bn %d vs %d)", c->n, d.n, c->T,
                 d.T, c->is_bn, d.bn);
    DVAE_REQUIRE(c->B > 0 && c->nB == c->n * c->B, "corrupt batch counts");
    return DVAE_OK;
}

// weight gradients shared by every GRU pass: dWhh over the rows [lo, hi) (the passes call it level by level, as soon
// as a level's d gh is final, on a stream beside the chain), and d gi_tab, d bhn over all rows at the end
// dgh_h / agg_h (half_grads(): d gh exists as bf16 only; nullable together): the bf16-stored operands
static int gru_whh_wgrad(const Dims& d, const PackedLayout& L, int g, float* pg, const float* dgh, const float* agg,
                         long lo, long hi, hipStream_t s, int max_split = 0, const uint16_t* dgh_h = nullptr,
                         const uint16_t* agg_h = nullptr) {
    const int Hp = d.Hp;
    if (hi <= lo) return DVAE_OK;
    return lin_wgrad(dgh + (size_t)lo * 3 * Hp, 3 * Hp, agg + (size_t)lo * Hp, Hp, nullptr, pg + L.whh[g], Hp, (int)(hi - lo),
                     3 * Hp, Hp, s, 0, max_split, dgh_h ? dgh_h + (size_t)lo * 3 * Hp : nullptr,
                     (dgh_h && agg_h) ? agg_h + (size_t)lo * Hp : nullptr);
}
static int gru_type_wgrads(const Dims& d, const PackedLayout& L, int g, float* pg, const float* dgh, const float* dgin,
                           const float* xoh, int rows, hipStream_t s, const uint16_t* dgh_h = nullptr) {
    // d gi_tab[t] = sum_{rows of type t}: (d pre_r, d pre_z) from dgh, d pre_n from dgin; d bhn = column sum of
    // dgh's n block: one sweep over the rows sorted by type (instead of two one-hot products and a column sum)
    const int* tcsr = reinterpret_cast<const int*>(xoh);
    if (dgh_h)
        return type_colsum_h(reinterpret_cast<const float*>(dgh_h), dgin, tcsr, tcsr + 40, d.T, pg + L.gi[g], pg + L.bhn[g], rows,
                             d.Hp, s, 1);
    return dvae_type_colsum(dgh, dgin, tcsr, tcsr + 40, d.T, pg + L.gi[g], pg + L.bhn[g], rows, d.Hp, s);
}
// (rows [zero_lo, zero_hi) have h_prev = 0 -- the first level of an encoder pass -- and are left out of dWhh)
static int gru_wgrads(const Dims& d, const PackedLayout& L, int g, float* pg, const float* dgh, const float* dgin,
                      const float* agg, const float* xoh, int rows, hipStream_t s, int zero_lo = 0, int zero_hi = 0,
                      const uint16_t* dgh_h = nullptr, const uint16_t* agg_h = nullptr) {
    const int lo = zero_lo == 0 ? zero_hi : 0, hi = zero_hi == rows && zero_lo > 0 ? zero_lo : rows;  // a prefix or a suffix
    DVAE_TRY(gru_whh_wgrad(d, L, g, pg, dgh, agg, lo, hi, s, 0, dgh_h, agg_h));
    return gru_type_wgrads(d, L, g, pg, dgh, dgin, xoh, rows, s, dgh_h);
}

}  // namespace dvae

using namespace dvae;

extern "C" int dvae_debug_side_stream(void* caller_stream, void** side_stream) {
    DVAE_REQUIRE(side_stream != nullptr, "dvae_debug_side_stream: null output");
    SideGuard g((hipStream_t)caller_stream);
    DVAE_TRY(side_acquire(g));
    *side_stream = (void*)g.sd->stream;
    return DVAE_OK;
}

extern "C" int64_t dvae_encode_ws_floats(const dvae_model_desc* m, const dvae_batch_counts* c) {
    Dims d;
    if (make_dims(m, d) != 0 || check_batch(d, c) != 0) return -1;
    return (int64_t)carve_enc(d, *c, nullptr).total;
}
extern "C" int64_t dvae_loss_ws_floats(const dvae_model_desc* m, const dvae_batch_counts* c) {
    Dims d;
    if (make_dims(m, d) != 0 || check_batch(d, c) != 0) return -1;
    return (int64_t)carve_dec(d, *c, nullptr).total;
}

// The chain levels of the fp32 mode run the fused level kernels of chain.hip (hidden product +
// pointwise tail in one launch); DVAE_CHAIN_FUSED=0 falls back to GEMM + pointwise launches (A/B runs),
// =2 forces them for every level size.
// They pay for levels of one to two rounds of the chip (64 x 32 tiles; cfg 2 / cfg 4: 256 workgroups,
// +5 % / +4 % on the step).  Smaller levels leave CUs idle where the generic kernel's split-K and
// small tiles spread the work (same-box sweep of cfg 2 at B = 64 / 128 / 256 / 512: -7 / -8 / -5 / 0 %),
// and a larger level (cfg 5: 2048 rows x Hp 1024 = 4 rounds) is a well-shaped product for the
// 128 x 128 tiles of gemm.hip, whose operand reuse then wins (6.08 k vs 5.93 k DAGs/s).
// (round 2, cfg 2 model at B = 384 / 512 / 768, fused vs product + pointwise: 153.1 k vs 156.0 k, 185.7 k vs 183.2 k,
// 226.3 k vs 203.5 k DAGs/s: from half a round of the chip on; up to 256 rows chain_tiny_kernel serves the level.)
static bool chain_level_small(const Dims& d, int B) {
    const long tiles = (long)cdiv(B, 64) * (d.Hp / 32);
    return tiles >= 128 && tiles <= 512;
}
static bool chain_fused(const Dims& d, int B) {
    static const int knob = [] { const char* v = getenv("DVAE_CHAIN_FUSED"); return v ? atoi(v) : 1; }();
    // bf16 mode: the bf16 level kernels (chain.hip) for large levels when enabled; the smallest levels run the exact-fp32
    // chain_tiny_kernel in this mode too (bf16 inputs are a permission, not an obligation: at <= 256 rows the level is
    // bound by launch and load latencies, not by the matrix unit -- cfg 1 bf16 25.4 k -> fp32-mode speed)
    if (d.bf16) return knob != 0 && !d.bn && (chain_bf16_ok(B, d.Hp) || B <= chain_tiny_rows());
    return (knob == 2 || (knob == 1 && (chain_level_small(d, B) || B <= chain_small_rows() || B <= chain_tiny_rows()))) && !d.bn;
}
// both encoder directions of a level as one grid (dvae_chain_level_batch) instead of two streams:
// measured no faster forward (0.64 ms either way) and slower backward (1.08 -> 1.27 ms), because
// workgroups launched together run in lockstep and nothing overlaps; kept behind DVAE_CHAIN_MERGE=1
// the whole forward chain of the encoder as ONE persistent launch (chain.hip: enc_chain_fwd): DVAE_CHAIN_PERSIST=1
static int device_cu_count() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return cus;
}
static bool chain_persist(const Dims& d, int B) {
    static const int knob = [] { const char* v = getenv("DVAE_CHAIN_PERSIST"); return v ? atoi(v) : 0; }();
    if (!knob || d.bf16 || d.bn) return false;
    const long tiles = (long)cdiv(B, 64) * (d.Hp / 32);
    return tiles <= 256 && tiles <= device_cu_count();
}
// Small batches (<= 256 rows per level): a whole pass as ONE persistent launch of 16 x 16 tiles (chain_persist.hip)
// instead of two launches per level; DVAE_CHAIN_TINY_PERSIST=0 restores the per-level launches.
static bool chain_tiny_persist(const Dims& d, int B, int passes = 1) {
    return !d.bn && B <= chain_tiny_rows() && tiny_chain_fits(B, d.Hp, passes);
}
static void tiny_dir_fwd(TinyChainDir& e, float* agg, float* gates, float* ghn, float* hs, float* gm, float* msg,
                         const float* whh, const float* wgm, const float* gi, const float* bhn, const float* gv,
                         const int* off, const int* idx, int reverse, int level0_product) {
    e.agg = agg; e.gates = gates; e.ghn = ghn; e.hs = hs; e.gm = gm; e.msg = msg;
    e.whh = whh; e.wgm = wgm; e.gi = gi; e.bhn = bhn; e.gv = gv; e.off = off; e.idx = idx;
    e.reverse = reverse; e.level0_product = level0_product;
}
static bool chain_merge_dirs() {
    static const int knob = [] { const char* v = getenv("DVAE_CHAIN_MERGE"); return v ? atoi(v) : 0; }();
    return knob != 0;
}
// the backward twins (dgrad + GRU backward tail, dgrad accumulate): DVAE_CHAIN_FUSED_BWD=0 falls back
static bool chain_fused_bwd(const Dims& d, int B) {
    static const int knob = [] { const char* v = getenv("DVAE_CHAIN_FUSED_BWD"); return v ? atoi(v) : 1; }();
    static const int small_bwd = [] { const char* v = getenv("DVAE_CHAIN_SMALL_M_BWD"); return v ? atoi(v) : 128; }();
    if (d.bf16) return knob != 0 && !d.bn && (chain_bf16_ok(B, d.Hp) || B <= chain_tiny_rows());
    return (knob == 2 || (knob == 1 && (chain_level_small(d, B) || B <= (small_bwd < chain_small_rows() ? small_bwd : chain_small_rows()) ||
                                        B <= chain_tiny_rows()))) && !d.bn;
}

// HALF GATES (round 3; bf16 mode, the unfused path of large levels -- wherever act_shadows() is on and neither chain runs
// fused level kernels; DVAE_GATES_BF16=0: off).  The level products write their pre-activations as bf16 (c_bf16) into the
// UPPER half of the fp32-sized gates / gm allocation, the forward tails read them from there and save r, z, n, gh_n /
// g, m as bf16 in the LOWER half (ghn: the lower half of its allocation), the backward tails read those.  Rows keep their
// indices; only the element size changes (hoff / pre_of below).  ~16 of the ~38 KB a GRU row moves forward, ~8 of ~50 backward.
static bool half_gates(const Dims& d, int B) {
    static const int knob = [] { const char* v = getenv("DVAE_GATES_BF16"); return v ? atoi(v) : 1; }();
    return knob && act_shadows(d, B) && bf16_weight_copies_on() && !chain_fused(d, B) && !chain_fused_bwd(d, B);
}
// HALF GRADS (on top of half gates; DVAE_GRADS_BF16=0: off): d gh / d gm (and d pre_n, `dgin`) exist ONLY as bf16 -- the arrays that were their
// shadows -- the weight-gradient products rounded them to bf16 on the way to LDS anyway (now: gemm_bf16.hip's bf16-stored
// K-major loaders, with the state / predecessor-sum shadows on the other side), the data gradients already read the
// shadows, and the per-type / per-level column sums read bf16 (type_colsum_kernel<true>, segment_colsum_kernel<true>).
static bool half_grads(const Dims& d, int B) {
    static const int knob = [] { const char* v = getenv("DVAE_GRADS_BF16"); return v ? atoi(v) : 1; }();
    return knob && half_gates(d, B);
}
static inline float* hoff(float* base, size_t elems, bool hg) {
    return hg ? reinterpret_cast<float*>(reinterpret_cast<uint16_t*>(base) + elems) : base + elems;
}
// row `elems / width` of the pre-activation scratch behind an allocation of `total` fp32 elements
static inline uint16_t* pre_of(float* base, size_t total, size_t elems, bool hg) {
    return hg ? reinterpret_cast<uint16_t*>(base) + total + elems : nullptr;
}

// The weight gradients of a backward chain (dWhh, d[Wg|Wm]) are sums over levels: each level's term is enqueued on a
// stream beside the chain as soon as that level's d gh / d gm is final, where it fills the matrix-unit time the
// latency-bound level kernels leave idle, instead of as whole-pass products after the chain with nothing beside them
// (decoder: 0.45 ms of idle side stream under the chain, then 0.2 ms of products alone).
// Measured (cfg 2, same box, interleaved runs): decoder pass 4.086 -> 4.058 ms per step; the encoder's two passes, whose
// chains already occupy two streams, LOSE with a third stream of per-level terms (4.06 -> 4.33 ms: the small products
// delay the level kernels they share CUs with), so the default is the decoder only.  Capping the terms' split-K at 1
// (fewer workgroups, no atomics) and a low-priority side stream were measured too: 4.16 ms and no change.
// DVAE_WGRAD_BY_LEVEL: bit 0 decoder, bit 1 encoder (default 1; 0 = whole-pass products after the chains).
// workgroups of the pair forward (= rows of its w2-gradient partials, summed by the backward): one pair per wave for
// small batches instead of the full 1024 (batch 32: 896 pairs; every workgroup clears its LDS, writes HEp partial sums
// and draws a ticket whether it has a pair or not), whole multiples of 8 for the per-XCD graph mapping
static int pair_blocks(int n_pairs, int cap) {
    int nb = ((n_pairs + 3) / 4 + 7) & ~7;
    if (nb < 64) nb = 64;
    return nb < cap ? nb : cap;
}
static int pair_group_knob() {
    static const int knob = [] { const char* v = getenv("DVAE_PAIR_GROUP"); return v ? atoi(v) : 1; }();
    return knob;
}
static int wbl_knob() {
    static const int knob = [] { const char* v = getenv("DVAE_WGRAD_BY_LEVEL"); return v ? atoi(v) : 1; }();
    return knob;
}
static bool wgrad_by_level(int which = 1) { return (wbl_knob() & which) != 0; }
// (bf16 mode, large levels -- cfg 5: 2048 rows x Hp 1024 -- the whole-pass products measured 1 % better; bit 2 of the
// knob keeps the per-level terms there too)
// (and not below 65 graphs: the per-level terms are 15 more launches and 7 forks per cfg 1 step, whose host time bounds
// the batch-32 step -- 1.04 -> 0.99 ms, host enqueue 1.05 -> 0.85 ms; at 128 / 256 graphs they win by 0.5 %; bit 3 of the
// knob keeps them at every batch size)
static bool wbl_size_ok(const Dims& d, int B) {
    if (B <= 64 && (wbl_knob() & 8) == 0) return false;
    return !d.bf16 || (long)B * d.Hp <= 1024L * 512 || (wbl_knob() & 4) != 0;
}
static int wbl_split() {
    static const int knob = [] { const char* v = getenv("DVAE_WBL_SPLIT"); return v ? atoi(v) : 0; }();
    return knob;
}
// DVAE_WBL_GROUP=g: the decoder's per-level weight-gradient terms are issued for g finished levels at a time (their
// rows are contiguous): g times fewer launches, g times longer K per workgroup, the side stream g - 1 levels later
static int wbl_group() {
    static const int knob = [] { const char* v = getenv("DVAE_WBL_GROUP"); return v ? atoi(v) : 1; }();
    return knob < 1 ? 1 : knob;
}

// level-major batches (every graph full size: all of the reference's data sets) run the loss forward's per-level
// predictor products beside the chain
// (small batches: the per-level side-stream launches cost more host time -- a fork and three launches per level --
// than their overlap gains, and batch 32 is bound by the host's enqueue rate; measured on the step, by level vs whole
// batch: B = 32 26.2 k vs 29.9 k DAGs/s, 128 93.3 k vs 98.2 k, 256 149.6 k vs 152.2 k, 512 200.4 k vs 197.3 k;
// DVAE_BY_LEVEL=2 forces the per-level form at every size)
static bool loss_by_level(const Dims& d, const dvae_batch_counts& c) {
    static const int by_level_knob = [] { const char* v = getenv("DVAE_BY_LEVEL"); return v ? atoi(v) : 1; }();
    return !d.bn && c.level_major != 0 && (by_level_knob == 2 || (by_level_knob == 1 && c.B > 256));
}
// does dvae_loss_forward run the decoder's chain as ONE persistent launch?  (dvae_loss_backward asks too: that launch
// also zeroes the counters of the backward pass's launch)
static bool loss_fwd_tiny(const Dims& d, const dvae_batch_counts& c) {
    return chain_fused(d, c.B) && !loss_by_level(d, c) && chain_tiny_persist(d, c.B);
}
static bool enc_fwd_tiny(const Dims& d, const dvae_batch_counts& c) {
    return chain_fused(d, c.B) && chain_tiny_persist(d, c.B, d.bidir ? 2 : 1);
}

// =====================================================================================
// encode
// =====================================================================================
extern "C" int dvae_encode_forward(const dvae_model_desc* m, const dvae_batch_counts* c, const int32_t* blob,
                                   const float* pw, float* ws, float* mu, float* logvar, void* stream) {
    Dims d;
    DVAE_TRY(make_dims(m, d));
    DVAE_TRY(check_batch(d, c));
    tl_bf16 = d.bf16;
    DVAE_REQUIRE(blob && pw && ws && mu && logvar, "dvae_encode_forward: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const PackedLayout L = make_layout(d);
    ModeGuard mode_guard(d, L, pw);
    const BatchView bv = bind_batch(*c, blob);
    EncWs w = carve_enc(d, *c, ws);
    const int B = c->B, n = d.n, Hp = d.Hp, nB = c->nB;
    const int dirs = d.bidir ? 2 : 1;
    const bool fused = chain_fused(d, c->B);

    // DVAE_BN has no recurrence: each direction is one batched pass over all levels.
    auto run_dir_bn = [&](int k, hipStream_t s) -> int {
        const int* off = k == 0 ? bv.pred_off : bv.succ_off;
        const int* idx = k == 0 ? bv.pred_idx : bv.succ_idx;
        DVAE_TRY(dvae_gather_sum_f32(w.agg[k], Hp, pw + L.gtab[k], Hp, off, idx, nB, Hp, 0, s));
        DVAE_TRY(lin_fwd(w.agg[k], Hp, nullptr, pw + L.whh[k], Hp, nullptr, w.gates[k], 3 * Hp, nB, 3 * Hp, Hp, 0, 0, s));
        DVAE_TRY(dvae_gru_pointwise_fwd(w.gates[k], pw + L.gi[k], bv.types, pw + L.bhn[k], w.agg[k], Hp, w.hs[k],
                                        Hp, w.ghn[k], nB, Hp, s));
        return DVAE_OK;
    };
    const bool hg = half_gates(d, B);
    // One level of one direction of the DVAE chain: GRU (hidden product + pointwise), then the
    // gate/mapper product whose pointwise tail also gathers the next level's predecessor sum.
    auto level = [&](int k, int step, hipStream_t s) -> int {
        const int* off = k == 0 ? bv.pred_off : bv.succ_off;
        const int* idx = k == 0 ? bv.pred_idx : bv.succ_idx;
        const int v = k == 0 ? step : n - 1 - step;
        const long r0 = (long)v * B;
        float* agg = w.agg[k] + r0 * Hp;
        float* gates = hoff(w.gates[k], r0 * 3 * Hp, hg);
        float* ghn = hoff(w.ghn[k], r0 * Hp, hg);
        uint16_t* pre_gh = pre_of(w.gates[k], (size_t)nB * 3 * Hp, r0 * 3 * Hp, hg);
        // bf16 shadows (act_shadows; unfused path only -- the fused level kernels read fp32): this level's rows
        uint16_t* hs_h = (w.hs_h[k] && !fused) ? w.hs_h[k] + r0 * Hp : nullptr;
        uint16_t* agg_h = (w.agg_h[k] && !fused) ? w.agg_h[k] + r0 * Hp : nullptr;
        if (step == 0) {  // no predecessors yet: h_prev = 0, hidden product = 0 (one launch; leaves agg = 0 for the backward)
            DVAE_TRY(gru_first_level(gates, pw + L.gi[k], bv.types + r0, pw + L.bhn[k], agg, w.hs[k] + r0 * Hp, ghn, B, Hp, s,
                                     hs_h, hg));
        } else if (!fused) {
            DVAE_TRY(lin_fwd(agg, Hp, nullptr, pw + L.whh[k], Hp, nullptr, hg ? reinterpret_cast<float*>(pre_gh) : gates, 3 * Hp, B,
                             3 * Hp, Hp, 0, 0, s, hg, agg_h));
        }
        if (fused && step > 0)  // hidden product + GRU tail in one launch (chain.hip)
            DVAE_TRY(dvae_gru_level_fwd(agg, Hp, pw + L.whh[k], pw + L.gi[k], bv.types + r0, pw + L.bhn[k], gates,
                                        w.ghn[k] + r0 * Hp, w.hs[k] + r0 * Hp, Hp, B, Hp, s));
        else if (step > 0)
            DVAE_TRY(gru_pointwise_fwd_h(gates, pw + L.gi[k], bv.types + r0, pw + L.bhn[k], agg, Hp, w.hs[k] + r0 * Hp, Hp, ghn,
                                         B, Hp, hs_h, s, pre_gh));
        if (step < n - 1) {
            const long r1 = (long)(k == 0 ? v + 1 : v - 1) * B;
            float* gm = hoff(w.gm[k], r0 * 2 * Hp, hg);
            uint16_t* pre_gm = pre_of(w.gm[k], (size_t)nB * 2 * Hp, r0 * 2 * Hp, hg);
            if (fused) {  // gate/mapper product + its tail + the next level's predecessor sum
                DVAE_TRY(dvae_gated_level_fwd(w.hs[k] + r0 * Hp, Hp, pw + L.wgm[k], pw + L.gv[k] + (size_t)v * 2 * Hp, gm,
                                              w.msg[k], Hp, (int)r0, w.agg[k] + r1 * Hp, Hp, off + r1, idx, B, Hp, s));
            } else {
                DVAE_TRY(lin_fwd(w.hs[k] + r0 * Hp, Hp, nullptr, pw + L.wgm[k], Hp, nullptr, hg ? reinterpret_cast<float*>(pre_gm) : gm,
                                 2 * Hp, B, 2 * Hp, Hp, 0, 0, s, hg, hs_h));
                DVAE_TRY(gated_fwd_gather_h(gm, pw + L.gv[k] + (size_t)v * 2 * Hp, w.msg[k], Hp, (int)r0, w.agg[k] + r1 * Hp, Hp,
                                            off + r1, idx, B, Hp, w.agg_h[k] ? w.agg_h[k] + r1 * Hp : nullptr, s, pre_gm));
            }
        }
        return DVAE_OK;
    };
    // The two directions are independent chains of small dependent launches; they run on two
    // streams and are ENQUEUED level by level in turn, so that neither stream starves while the
    // host is still issuing the other one's launches.
    // (optional, DVAE_CHAIN_MERGE=1: both directions of a level as ONE grid on the caller's stream,
    // dvae_chain_level_batch, instead of two streams)
    const bool tiny = enc_fwd_tiny(d, *c);
    const bool persist = tiny || chain_persist(d, c->B);
    const bool merged = !persist && fused && dirs == 2 && !d.bn && chain_merge_dirs();
    SideGuard side_guard(s);
    Side*& side = side_guard.sd;
    if (dirs == 2 && !merged && !persist) {
        DVAE_TRY(side_acquire(side_guard));
        DVAE_TRY(fork_side(side_guard));
    }
    if (tiny) {
        TinyChainArgs a = {};
        for (int k = 0; k < dirs; ++k)
            tiny_dir_fwd(a.d[k], w.agg[k], w.gates[k], w.ghn[k], w.hs[k], w.gm[k], w.msg[k], pw + L.whh[k], pw + L.wgm[k],
                         pw + L.gi[k], pw + L.bhn[k], pw + L.gv[k], k == 0 ? bv.pred_off : bv.succ_off,
                         k == 0 ? bv.pred_idx : bv.succ_idx, k, 0);
        a.types = bv.types; a.B = B; a.Hp = Hp; a.n = n; a.ndirs = dirs;
        a.sync = reinterpret_cast<unsigned*>(w.sync);
        a.status = a.sync + TINY_CHAIN_SYNC_WORDS - 64;
        a.zero_next = a.sync + TINY_CHAIN_SYNC_WORDS;
        DVAE_TRY(tiny_chain_fwd(a, s));
    } else if (persist) {
        EncChainArgs a = {};
        for (int k = 0; k < dirs; ++k) {
            EncChainDir& e = a.d[k];
            e.agg = w.agg[k]; e.gates = w.gates[k]; e.ghn = w.ghn[k]; e.hs = w.hs[k]; e.gm = w.gm[k]; e.msg = w.msg[k];
            e.whh = pw + L.whh[k]; e.wgm = pw + L.wgm[k]; e.gi = pw + L.gi[k]; e.bhn = pw + L.bhn[k]; e.gv = pw + L.gv[k];
            e.off = k == 0 ? bv.pred_off : bv.succ_off;
            e.idx = k == 0 ? bv.pred_idx : bv.succ_idx;
        }
        a.types = bv.types; a.B = B; a.Hp = Hp; a.n = n; a.ndirs = dirs;
        a.sync = reinterpret_cast<unsigned*>(w.sync);
        a.status = a.sync + ENC_CHAIN_SYNC_WORDS - 64;
        DVAE_TRY(enc_chain_fwd(a, s));
    } else if (d.bn) {
        if (dirs == 2) DVAE_TRY(run_dir_bn(1, side->stream));
        DVAE_TRY(run_dir_bn(0, s));
    } else if (merged) {
        // first level of each direction: no predecessors, h_prev = 0 (the product over zeros is exact)
        DVAE_HIP(hipMemsetAsync(w.agg[0], 0, sizeof(float) * (size_t)B * Hp, s));
        DVAE_HIP(hipMemsetAsync(w.agg[1] + (size_t)(n - 1) * B * Hp, 0, sizeof(float) * (size_t)B * Hp, s));
        for (int step = 0; step < n; ++step) {
            dvae_level_desc gru[2] = {}, gat[2] = {};
            for (int k = 0; k < 2; ++k) {
                const int v = k == 0 ? step : n - 1 - step;
                const long r0 = (long)v * B, r1 = (long)(k == 0 ? v + 1 : v - 1) * B;
                gru[k].a = w.agg[k] + r0 * Hp; gru[k].lda = Hp; gru[k].w = pw + L.whh[k];
                gru[k].gi_tab = pw + L.gi[k]; gru[k].types = bv.types + r0; gru[k].bhn = pw + L.bhn[k];
                gru[k].gates = w.gates[k] + r0 * 3 * Hp; gru[k].ghn = w.ghn[k] + r0 * Hp;
                gru[k].hout = w.hs[k] + r0 * Hp; gru[k].ld_hout = Hp;
                if (step < n - 1) {
                    gat[k].a = w.hs[k] + r0 * Hp; gat[k].lda = Hp; gat[k].w = pw + L.wgm[k];
                    gat[k].gv_row = pw + L.gv[k] + (size_t)v * 2 * Hp; gat[k].gm = w.gm[k] + r0 * 2 * Hp;
                    gat[k].msg = w.msg[k]; gat[k].ld_msg = Hp; gat[k].cur_row0 = (int)r0;
                    gat[k].next_agg = w.agg[k] + r1 * Hp; gat[k].ld_agg = Hp;
                    gat[k].next_off = (k == 0 ? bv.pred_off : bv.succ_off) + r1;
                    gat[k].idx = k == 0 ? bv.pred_idx : bv.succ_idx;
                }
            }
            DVAE_TRY(dvae_chain_level_batch(DVAE_LEVEL_GRU_FWD, gru, 2, B, Hp, s));
            if (step < n - 1) DVAE_TRY(dvae_chain_level_batch(DVAE_LEVEL_GATED_FWD, gat, 2, B, Hp, s));
        }
    } else {
        for (int step = 0; step < n; ++step) {
            if (dirs == 2) DVAE_TRY(level(1, step, side->stream));
            DVAE_TRY(level(0, step, s));
        }
    }
    if (dirs == 2 && !merged && !persist) DVAE_TRY(join_side(side_guard));
    // graph state (models.py:704-718 / 1006-1027)
    DVAE_TRY(dvae_gather_sum_f32(w.hg, 2 * Hp, w.hs[0], Hp, bv.hgf_off, bv.hgf_idx, B, Hp, 0, s));
    const float* hgu = w.hg;
    int ld_hgu = 2 * Hp;
    if (d.bidir) {
        DVAE_TRY(dvae_gather_sum_f32(w.hg + Hp, 2 * Hp, w.hs[1], Hp, bv.hgb_off, bv.hgb_idx, B, Hp, 0, s));
        DVAE_TRY(lin_fwd(w.hg, 2 * Hp, nullptr, pw + L.wu, 2 * Hp, pw + L.bu, w.hgu, Hp, B, Hp, 2 * Hp, 0, 0, s));
        hgu = w.hgu;
        ld_hgu = Hp;
    }
    DVAE_TRY(lin_fwd(hgu, ld_hgu, nullptr, pw + L.fc12, Hp, pw + L.b12, w.out12, 2 * d.NZp, B, 2 * d.NZp, Hp, 0, 0, s));
    // (a persistent chain whose bounded wait ran out leaves its flag set: mu and logvar become NaN, never stale numbers)
    EW(split_mu_logvar_kernel, (long)B * d.nz, w.out12, mu, logvar, B, d.nz, d.NZp,
       tiny ? reinterpret_cast<const unsigned*>(w.sync) + TINY_CHAIN_SYNC_WORDS - 64
            : persist ? reinterpret_cast<const unsigned*>(w.sync) + ENC_CHAIN_SYNC_WORDS - 64 : nullptr);
    return DVAE_OK;
}

extern "C" int dvae_encode_backward(const dvae_model_desc* m, const dvae_batch_counts* c, const int32_t* blob,
                                    const float* pw, float* ws, const float* dmu, const float* dlogvar, float* pg,
                                    void* stream) {
    Dims d;
    DVAE_TRY(make_dims(m, d));
    DVAE_TRY(check_batch(d, c));
    tl_bf16 = d.bf16;
    DVAE_REQUIRE(blob && pw && ws && dmu && dlogvar && pg, "dvae_encode_backward: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const PackedLayout L = make_layout(d);
    ModeGuard mode_guard(d, L, pw);
    const BatchView bv = bind_batch(*c, blob);
    EncWs w = carve_enc(d, *c, ws);
    const int B = c->B, n = d.n, Hp = d.Hp, nB = c->nB, NZ2 = 2 * d.NZp;
    const int dirs = d.bidir ? 2 : 1;
    const bool fused = chain_fused_bwd(d, c->B);

    DVAE_HIP(hipMemsetAsync(pg, 0, L.total * sizeof(float), s));
    // csr_on_side: the type sort is a one-workgroup, latency-bound kernel that depends on the batch only; with
    // two directions it runs on the (still idle) side stream under the head of this function, and the main
    // stream picks it up with the join below, long after it has finished
    SideGuard side_guard(s);
    Side*& side = side_guard.sd;
    const bool tiny = fused && chain_tiny_persist(d, c->B, dirs);  // both directions' reverse chains as one persistent launch
    const bool merged = !tiny && fused && dirs == 2 && !d.bn && chain_merge_dirs();
    // per-level weight-gradient terms beside the chains: on the (idle) side stream with one direction, on a third
    // stream with two (both other streams walk a chain)
    const bool wbl = !d.bn && !merged && !tiny && wgrad_by_level(2);
    if (dirs == 2 || wbl) DVAE_TRY(side_acquire(side_guard));
    hipStream_t wg_stream = wbl ? (dirs == 2 ? side->aux : side->stream) : nullptr;
    if (dirs == 2) {
        DVAE_TRY(fork_side(side_guard));
        DVAE_TRY(dvae_type_csr(bv.types, nB, nullptr, 0, d.T, reinterpret_cast<int*>(w.xoh), reinterpret_cast<int*>(w.xoh) + 40,
                               side->stream));
    }
    EW(join_mu_logvar_kernel, (long)B * NZ2, w.d12, dmu, dlogvar, B, d.nz, d.NZp);
    const float* hgu = d.bidir ? w.hgu : w.hg;
    const int ld_hgu = d.bidir ? Hp : 2 * Hp;
    DVAE_TRY(dvae_segment_colsum_f32(pg + L.b12, NZ2, w.d12, NZ2, B, B, NZ2, s));
    DVAE_TRY(lin_wgrad(w.d12, NZ2, hgu, ld_hgu, nullptr, pg + L.fc12, Hp, B, NZ2, Hp, s));
    DVAE_TRY(lin_dgrad(w.d12, NZ2, pw + L.fc12, Hp, w.dhgu, Hp, nullptr, B, NZ2, Hp, 0, s));
    const float* dhg = w.dhgu;
    int ld_dhg = Hp;
    if (d.bidir) {
        DVAE_TRY(dvae_segment_colsum_f32(pg + L.bu, Hp, w.dhgu, Hp, B, B, Hp, s));
        DVAE_TRY(lin_wgrad(w.dhgu, Hp, w.hg, 2 * Hp, nullptr, pg + L.wu, 2 * Hp, B, Hp, 2 * Hp, s));
        DVAE_TRY(lin_dgrad(w.dhgu, Hp, pw + L.wu, 2 * Hp, w.dhg, 2 * Hp, nullptr, B, Hp, 2 * Hp, 0, s));
        dhg = w.dhg;
        ld_dhg = 2 * Hp;
    }
    // rows sorted by vertex type, for the GRUs' input-side gradients (off at xoh[0..T], idx from xoh[40]): with two
    // directions it was enqueued on the side stream at the top of this function (csr_on_side below)
    if (dirs == 1)
        DVAE_TRY(dvae_type_csr(bv.types, nB, nullptr, 0, d.T, reinterpret_cast<int*>(w.xoh), reinterpret_cast<int*>(w.xoh) + 40, s));

    auto begin_dir = [&](int k, hipStream_t s) -> int {
        DVAE_HIP(hipMemsetAsync(w.dhs[k], 0, sizeof(float) * (size_t)nB * Hp, s));
        DVAE_TRY(dvae_scatter_rows_add_f32(w.dhs[k], Hp, dhg + (k ? Hp : 0), ld_dhg, k ? bv.hgb_off : bv.hgf_off,
                                           k ? bv.hgb_idx : bv.hgf_idx, B, Hp, s));
        // d gm: every level's rows are written whole by the chain below except the level that sends no
        // message (the last vertex of the pass): only those rows need the zeros
        if (!d.bn)
            DVAE_HIP(hipMemsetAsync(w.dgm[k] + (size_t)(k == 0 ? n - 1 : 0) * B * 2 * Hp, 0, sizeof(float) * (size_t)B * 2 * Hp, s));
        if (!d.bn && half_grads(d, B))
            DVAE_HIP(hipMemsetAsync(w.dgm_h[k] + (size_t)(k == 0 ? n - 1 : 0) * B * 2 * Hp, 0, sizeof(uint16_t) * (size_t)B * 2 * Hp, s));
        return DVAE_OK;
    };
    auto run_dir_bn = [&](int k, hipStream_t s) -> int {
        DVAE_TRY(dvae_gru_pointwise_bwd(w.dhs[k], Hp, w.gates[k], w.ghn[k], w.agg[k], Hp, w.dgh[k], w.dgin[k],
                                        w.dagg[k], Hp, nB, Hp, s));
        DVAE_TRY(lin_dgrad(w.dgh[k], 3 * Hp, pw + L.whh[k], Hp, w.dagg[k], Hp, nullptr, nB, 3 * Hp, Hp, 1, s));
        // d gtab[t] = sum over slots of (#neighbours of type t) * dagg[slot]
        const int* off = k == 0 ? bv.pred_off : bv.succ_off;
        const int* idx = k == 0 ? bv.pred_idx : bv.succ_idx;
        EW(count_kernel, (long)nB * d.Tp, w.cnt[k], off, idx, nB, d.Tp);
        DVAE_TRY(lin_wgrad(w.cnt[k], d.Tp, w.dagg[k], Hp, nullptr, pg + L.gtab[k], Hp, nB, d.Tp, Hp, s));
        return DVAE_OK;
    };
    // one level of the reverse chain.  Messages of v went to its successors (forward pass) /
    // predecessors (backward pass): the transposed lists gather their d agg rows.
    auto level = [&](int k, int step, hipStream_t s) -> int {
        const int* toff = k == 0 ? bv.succ_off : bv.pred_off;
        const int* tidx = k == 0 ? bv.succ_idx : bv.pred_idx;
        const int v = k == 0 ? step : n - 1 - step;
        const long r0 = (long)v * B;
        float* dhs = w.dhs[k] + r0 * Hp;
        const bool hg = half_gates(d, B);  // (false whenever `fused`)
        const bool gh = half_grads(d, B);
        // bf16 shadows of this level's d gm / d gh (act_shadows; unfused path)
        uint16_t* dgm_h = (w.dgm_h[k] && !fused) ? w.dgm_h[k] + r0 * 2 * Hp : nullptr;
        uint16_t* dgh_h = (w.dgh_h[k] && !fused) ? w.dgh_h[k] + r0 * 3 * Hp : nullptr;
        if (step < n - 1 && !fused) {
            DVAE_TRY(gated_bwd_gather_h(w.dagg[k], Hp, toff + r0, tidx, nullptr, nullptr, nullptr, hoff(w.gm[k], r0 * 2 * Hp, hg),
                                        gh ? nullptr : w.dgm[k] + r0 * 2 * Hp, B, Hp, dgm_h, s, hg));
            DVAE_TRY(lin_dgrad(w.dgm[k] + r0 * 2 * Hp, 2 * Hp, pw + L.wgm[k], Hp, dhs, Hp, nullptr, B, 2 * Hp, Hp, 1, s, nullptr,
                               dgm_h, gh));
        }
        // (fused: d gm of this level was written by the tail of the previous level's dgrad kernel, below)
        if (fused && step < n - 1)  // d h += d gm [Wg | Wm] and the GRU pointwise backward in one launch (chain.hip)
            DVAE_TRY(dvae_gru_level_bwd(w.dgm[k] + r0 * 2 * Hp, 2 * Hp, 2 * Hp, pw + L.wgm[k], dhs, Hp,
                                        w.gates[k] + r0 * 3 * Hp, w.ghn[k] + r0 * Hp, w.agg[k] + r0 * Hp, Hp,
                                        w.dgh[k] + r0 * 3 * Hp, w.dgin[k] + r0 * Hp, w.dagg[k] + r0 * Hp, Hp, B, Hp, s));
        else
            DVAE_TRY(gru_pointwise_bwd_h(dhs, Hp, hoff(w.gates[k], r0 * 3 * Hp, hg), hoff(w.ghn[k], r0 * Hp, hg), w.agg[k] + r0 * Hp,
                                         Hp, gh ? nullptr : w.dgh[k] + r0 * 3 * Hp, hoff(w.dgin[k], r0 * Hp, gh), w.dagg[k] + r0 * Hp, Hp, B, Hp,
                                         (step > 0 || gh) ? dgh_h : nullptr, s, hg));
        if (step > 0 && fused) {
            // d agg += d gh W_hh, then -- same launch -- the gate/mapper pointwise backward of the level
            // processed next, whose d msg gathers d agg rows of this level (registers) and finished ones
            const long r1 = (long)(k == 0 ? v - 1 : v + 1) * B;
            dvae_level_desc da = {};
            da.a = w.dgh[k] + r0 * 3 * Hp; da.lda = 3 * Hp; da.K = 3 * Hp; da.w = pw + L.whh[k];
            da.dx = w.dagg[k] + r0 * Hp; da.ld_dx = Hp;
            da.gather_src = w.dagg[k]; da.next_off = toff + r1; da.idx = tidx; da.cur_row0 = (int)r0;
            da.gm = w.gm[k] + r1 * 2 * Hp; da.dgm_next = w.dgm[k] + r1 * 2 * Hp;
            DVAE_TRY(dvae_chain_level_batch(DVAE_LEVEL_DGRAD_ACC, &da, 1, B, Hp, s));
        } else if (step > 0)
            DVAE_TRY(lin_dgrad(w.dgh[k] + r0 * 3 * Hp, 3 * Hp, pw + L.whh[k], Hp, w.dagg[k] + r0 * Hp, Hp, nullptr, B,
                               3 * Hp, Hp, 1, s, nullptr, dgh_h, gh));
        if (wbl) {
            // final now: d gh of this level; d gm of the level processed next (fused: the tail above) or of this one
            if (dirs == 2) DVAE_TRY(aux_after(side_guard, s));
            else DVAE_TRY(fork_side(side_guard));
            if (step > 0)  // (the pass's first level has h_prev = 0: no term)
                DVAE_TRY(gru_whh_wgrad(d, L, k, pg, w.dgh[k], w.agg[k], r0, r0 + B, wg_stream, wbl_split(), gh ? w.dgh_h[k] : nullptr,
                                       w.agg_h[k]));
            const int sg = fused ? step - 1 : step;  // forward-pass step of the level whose d gm is final
            if (sg >= 0 && sg < n - 1) {
                const long rg = (long)(k == 0 ? sg : n - 1 - sg) * B;
                DVAE_TRY(lin_wgrad(w.dgm[k] + rg * 2 * Hp, 2 * Hp, w.hs[k] + rg * Hp, Hp, nullptr, pg + L.wgm[k], Hp, B,
                                   2 * Hp, Hp, wg_stream, 0, wbl_split(), gh ? w.dgm_h[k] + rg * 2 * Hp : nullptr,
                                   gh ? w.hs_h[k] + rg * Hp : nullptr));
            }
        }
        return DVAE_OK;
    };
    auto end_dir = [&](int k, hipStream_t s) -> int {
        if (!d.bn) {
            // (d gm of the pass's last vertex is zero -- no message leaves it: those B rows are left out)
            const size_t o = k == 0 ? 0 : (size_t)B;
            const bool gh = half_grads(d, B);
            if (!wbl)
                DVAE_TRY(lin_wgrad(w.dgm[k] + o * 2 * Hp, 2 * Hp, w.hs[k] + o * Hp, Hp, nullptr, pg + L.wgm[k], Hp, nB - B,
                                   2 * Hp, Hp, s, 0, 0, gh ? w.dgm_h[k] + o * 2 * Hp : nullptr, gh ? w.hs_h[k] + o * Hp : nullptr));
            if (gh) DVAE_TRY(segment_colsum_h(pg + L.gv[k], 2 * Hp, w.dgm_h[k], 2 * Hp, nB, B, 2 * Hp, s));
            else DVAE_TRY(dvae_segment_colsum_f32(pg + L.gv[k], 2 * Hp, w.dgm[k], 2 * Hp, nB, B, 2 * Hp, s));
        }
        const uint16_t* dgh_only = half_grads(d, B) ? w.dgh_h[k] : nullptr;
        if (wbl) return gru_type_wgrads(d, L, k, pg, w.dgh[k], w.dgin[k], w.xoh, nB, s, dgh_only);
        // DVAE: the pass's first level (vertex 0 forward, vertex n-1 backward) has no predecessors, h_prev = 0
        const int z_lo = d.bn ? 0 : (k == 0 ? 0 : nB - B), z_hi = d.bn ? 0 : (k == 0 ? B : nB);
        DVAE_TRY(gru_wgrads(d, L, k, pg, w.dgh[k], w.dgin[k], w.agg[k], w.xoh, nB, s, z_lo, z_hi, dgh_only, w.agg_h[k]));
        return DVAE_OK;
    };
    // as in the forward pass: two streams, enqueued level by level in turn
    // (as in the forward pass: optionally both directions of a level as one grid)
    if (dirs == 2) DVAE_TRY(join_side(side_guard));  // the type sort (csr_on_side)
    if (dirs == 2 && !merged && !tiny) {
        DVAE_TRY(fork_side(side_guard));
        DVAE_TRY(begin_dir(1, side->stream));
    }
    DVAE_TRY(begin_dir(0, s));
    if (tiny) {
        if (dirs == 2) DVAE_TRY(begin_dir(1, s));
        TinyChainArgs a = {};
        for (int k = 0; k < dirs; ++k) {
            TinyChainDir& e = a.d[k];
            tiny_dir_fwd(e, w.agg[k], w.gates[k], w.ghn[k], w.hs[k], w.gm[k], w.msg[k], pw + L.whh[k], pw + L.wgm[k],
                         pw + L.gi[k], pw + L.bhn[k], pw + L.gv[k], nullptr, nullptr, k, 0);
            e.dhs = w.dhs[k]; e.dgm = w.dgm[k]; e.dgh = w.dgh[k]; e.dgin = w.dgin[k]; e.dagg = w.dagg[k];
            e.toff = k == 0 ? bv.succ_off : bv.pred_off;
            e.tidx = k == 0 ? bv.succ_idx : bv.pred_idx;
        }
        a.types = bv.types; a.B = B; a.Hp = Hp; a.n = n; a.ndirs = dirs;
        a.sync = reinterpret_cast<unsigned*>(w.sync) + TINY_CHAIN_SYNC_WORDS;  // slot 1
        a.status = a.sync + TINY_CHAIN_SYNC_WORDS - 64;
        a.sync_is_zero = enc_fwd_tiny(d, *c) ? 1 : 0;  // dvae_encode_forward's persistent launch zeroed it
        a.poison = pg + L.whh[0];  // a timed-out wait turns the gradients NaN (the weight-gradient products below add to it)
        DVAE_TRY(tiny_chain_bwd(a, s));
        if (dirs == 2) DVAE_TRY(fork_side(side_guard));  // the two directions' weight gradients on two streams
    } else if (d.bn) {
        if (dirs == 2) DVAE_TRY(run_dir_bn(1, side->stream));
        DVAE_TRY(run_dir_bn(0, s));
    } else if (merged) {
        DVAE_TRY(begin_dir(1, s));
        for (int step = n - 1; step >= 0; --step) {
            dvae_level_desc gb[2] = {}, da[2] = {};
            for (int k = 0; k < 2; ++k) {
                const int v = k == 0 ? step : n - 1 - step;
                const long r0 = (long)v * B;
                if (step < n - 1) {
                    gb[k].a = w.dgm[k] + r0 * 2 * Hp; gb[k].lda = 2 * Hp; gb[k].K = 2 * Hp; gb[k].w = pw + L.wgm[k];
                    gb[k].dx = w.dhs[k] + r0 * Hp; gb[k].ld_dx = Hp;
                    gb[k].gates = w.gates[k] + r0 * 3 * Hp; gb[k].ghn = w.ghn[k] + r0 * Hp;
                    gb[k].hprev = w.agg[k] + r0 * Hp; gb[k].ld_hprev = Hp;
                    gb[k].dgh = w.dgh[k] + r0 * 3 * Hp; gb[k].dgin = w.dgin[k] + r0 * Hp;
                    gb[k].dhprev = w.dagg[k] + r0 * Hp; gb[k].ld_dhprev = Hp;
                } else {  // last vertex of the pass: no message left it
                    DVAE_TRY(dvae_gru_pointwise_bwd(w.dhs[k] + r0 * Hp, Hp, w.gates[k] + r0 * 3 * Hp, w.ghn[k] + r0 * Hp,
                                                    w.agg[k] + r0 * Hp, Hp, w.dgh[k] + r0 * 3 * Hp, w.dgin[k] + r0 * Hp,
                                                    w.dagg[k] + r0 * Hp, Hp, B, Hp, s));
                }
                da[k].a = w.dgh[k] + r0 * 3 * Hp; da[k].lda = 3 * Hp; da[k].K = 3 * Hp; da[k].w = pw + L.whh[k];
                da[k].dx = w.dagg[k] + r0 * Hp; da[k].ld_dx = Hp;
                if (step > 0) {  // fused tail: d gm of the level processed next
                    const long r1 = (long)(k == 0 ? v - 1 : v + 1) * B;
                    da[k].gather_src = w.dagg[k]; da[k].next_off = (k == 0 ? bv.succ_off : bv.pred_off) + r1;
                    da[k].idx = k == 0 ? bv.succ_idx : bv.pred_idx; da[k].cur_row0 = (int)r0;
                    da[k].gm = w.gm[k] + r1 * 2 * Hp; da[k].dgm_next = w.dgm[k] + r1 * 2 * Hp;
                }
            }
            if (step < n - 1) DVAE_TRY(dvae_chain_level_batch(DVAE_LEVEL_GRU_BWD, gb, 2, B, Hp, s));
            if (step > 0) DVAE_TRY(dvae_chain_level_batch(DVAE_LEVEL_DGRAD_ACC, da, 2, B, Hp, s));
        }
        DVAE_TRY(fork_side(side_guard));
    } else {
        for (int step = n - 1; step >= 0; --step) {
            if (dirs == 2) DVAE_TRY(level(1, step, side->stream));
            DVAE_TRY(level(0, step, s));
        }
    }
    if (dirs == 2) DVAE_TRY(end_dir(1, side->stream));
    DVAE_TRY(end_dir(0, s));
    if (dirs == 2 || side_guard.forked) DVAE_TRY(join_side(side_guard));
    if (side) DVAE_TRY(join_aux(side_guard));
    return DVAE_OK;
}

// =====================================================================================
// loss (teacher-forced decoder)
// =====================================================================================
extern "C" int dvae_loss_forward(const dvae_model_desc* m, const dvae_batch_counts* c, const int32_t* blob,
                                 const float* pw, float* ws, const float* mu, const float* logvar, const float* eps,
                                 float eps_scale, int32_t training, float beta, float* out3, void* stream) {
    Dims d;
    DVAE_TRY(make_dims(m, d));
    DVAE_TRY(check_batch(d, c));
    tl_bf16 = d.bf16;
    DVAE_REQUIRE(blob && pw && ws && mu && logvar && out3, "dvae_loss_forward: null pointer");
    DVAE_REQUIRE(!training || eps, "dvae_loss_forward: training mode needs the normal draw eps");
    hipStream_t s = (hipStream_t)stream;
    const PackedLayout L = make_layout(d);
    ModeGuard mode_guard(d, L, pw);
    const BatchView bv = bind_batch(*c, blob);
    DecWs w = carve_dec(d, *c, ws);
    const int B = c->B, n = d.n, Hp = d.Hp, nB = c->nB, PT = c->PT, HEp = d.HEp;
    const float* whh = pw + L.whh[2];
    const float* gi = pw + L.gi[2];
    const float* bhn = pw + L.bhn[2];

    const bool by_level = loss_by_level(d, *c);
    // bf16 mode, wide tables (cfg 5: HEp = 4096): EA / EB are STORED as bf16 by the products that make them (the pair
    // kernel is their only reader; the backward passes use its ReLU masks).  Same allocation, rows HEp bf16 words apart.
    const bool th = edge_tables_bf16(d);
    uint16_t* ea_h = reinterpret_cast<uint16_t*>(w.ea);
    uint16_t* eb_h = reinterpret_cast<uint16_t*>(w.eb);
    float* ea_zero = th ? reinterpret_cast<float*>(ea_h + (size_t)c->NA * HEp) : w.ea + (size_t)c->NA * HEp;
    float* eb_zero = th ? reinterpret_cast<float*>(eb_h + (size_t)c->NB * HEp) : w.eb + (size_t)c->NB * HEp;
    const int zero_floats = th ? HEp / 2 : HEp;  // the all-zero row of each table
    SideGuard side_guard(s);
    Side*& side = side_guard.sd;
    if (by_level) DVAE_TRY(side_acquire(side_guard));
    // small batches (launch-bound): one single-workgroup launch does the accumulators' and the tables' zero rows'
    // zeroing too; larger ones keep the multi-workgroup kernel (a lone workgroup walking 64 k elements was measured
    // 0.4 % slower on the cfg-2 step than the launches it saved), which still zeroes the two rows
    const bool one_init = (long)B * d.NZp <= 8192;
    if (one_init) {
        DVAE_TRY(loss_fwd_init(mu, logvar, eps, eps_scale, training, w.z, w.sums, ea_zero, eb_zero, zero_floats, B, d.nz,
                               d.NZp, s));
    } else {
        DVAE_HIP(hipMemsetAsync(w.sums, 0, 8 * sizeof(float), s));
        DVAE_TRY(reparam_kld_fwd(mu, logvar, eps, eps_scale, training, w.z, w.sums + 2, B, d.nz, d.NZp, s, ea_zero, eb_zero,
                                 zero_floats));
    }
    // H0 = tanh(fc3 z) lives in the level-0 rows of agg: it is h_prev of vertex 0 (models.py:810-814)
    DVAE_TRY(lin_fwd(w.z, d.NZp, nullptr, pw + L.fc3, d.NZp, pw + L.b3, w.agg, Hp, B, Hp, d.NZp, 2, 0, s));
    // (half grads: the backward's d W_hh product reads the predecessor-sum shadow of every row, level 0 = H0 included)
    if (half_grads(d, c->B)) DVAE_TRY(dvae_f32_to_bf16(w.agg, w.agg_h, (int64_t)B * Hp, s));
    const bool fused = chain_fused(d, c->B);
    const bool tiny = loss_fwd_tiny(d, *c);  // the decoder's chain as one persistent launch
    const bool hg = half_gates(d, c->B);     // (false whenever fused / tiny)
    const size_t NG = (size_t)(nB + PT) * 3 * Hp, NM = (size_t)nB * 2 * Hp;  // elements of the gates / gm allocations
    if (tiny) {
        TinyChainArgs a = {};
        tiny_dir_fwd(a.d[0], w.agg, w.gates, w.ghn, w.st, w.gm, w.msg, whh, pw + L.wgm[0], gi, bhn, pw + L.gv[0],
                     bv.pred_off, bv.pred_idx, 0, 1);
        a.types = bv.types; a.B = B; a.Hp = Hp; a.n = n; a.ndirs = 1;
        a.sync = reinterpret_cast<unsigned*>(w.sync);
        a.status = a.sync + TINY_CHAIN_SYNC_WORDS - 64;
        a.poison = w.sums;  // a timed-out wait turns the loss NaN (the pair kernel adds its sum to it)
        a.zero_next = a.sync + TINY_CHAIN_SYNC_WORDS;
        DVAE_TRY(tiny_chain_fwd(a, s));
    } else if (fused) {
        DVAE_TRY(dvae_gru_level_fwd(w.agg, Hp, whh, gi, bv.types, bhn, w.gates, w.ghn, w.st, Hp, B, Hp, s));
    } else {
        DVAE_TRY(lin_fwd(w.agg, Hp, nullptr, whh, Hp, nullptr, hg ? reinterpret_cast<float*>(pre_of(w.gates, NG, 0, hg)) : w.gates,
                         3 * Hp, B, 3 * Hp, Hp, 0, 0, s, hg));
        DVAE_TRY(gru_pointwise_fwd_h(w.gates, gi, bv.types, bhn, w.agg, Hp, w.st, Hp, w.ghn, B, Hp, w.st_h, s, pre_of(w.gates, NG, 0, hg)));
    }
    // bf16 shadows of the states and predecessor sums (act_shadows; the unfused path writes and reads them)
    uint16_t* const st_h = (!tiny && !fused) ? w.st_h : nullptr;
    uint16_t* const agg_h = (!tiny && !fused) ? w.agg_h : nullptr;
    if (d.bn) {
        // no recurrence: messages depend on predecessor types only (models.py:1044-1062)
        const long r0 = B;
        DVAE_TRY(dvae_gather_sum_f32(w.agg + r0 * Hp, Hp, pw + L.gtab[0], Hp, bv.pred_off + r0, bv.pred_idx, nB - B, Hp, 0, s));
        DVAE_TRY(dvae_gather_sum_f32(w.agg + (long)nB * Hp, Hp, pw + L.gtab[0], Hp, bv.p_off, bv.p_idx, PT, Hp, 0, s));
        const int rows = nB - B + PT;
        DVAE_TRY(lin_fwd(w.agg + r0 * Hp, Hp, nullptr, whh, Hp, nullptr, w.gates + r0 * 3 * Hp, 3 * Hp, rows, 3 * Hp, Hp, 0, 0, s));
        DVAE_TRY(dvae_gru_pointwise_fwd(w.gates + r0 * 3 * Hp, gi, bv.types + r0, bhn, w.agg + r0 * Hp, Hp,
                                        w.st + r0 * Hp, Hp, w.ghn + r0 * Hp, nB - B, Hp, s));
        DVAE_TRY(dvae_gru_pointwise_fwd(w.gates + (long)nB * 3 * Hp, gi, bv.p_type, bhn, w.agg + (long)nB * Hp, Hp,
                                        w.st + (long)nB * Hp, Hp, w.ghn + (long)nB * Hp, PT, Hp, s));
    } else {
        for (int v = 0; v < n && !tiny; ++v) {
            const long r0 = (long)v * B;
            if (v > 0 && fused) {
                DVAE_TRY(dvae_gru_level_fwd(w.agg + r0 * Hp, Hp, whh, gi, bv.types + r0, bhn, w.gates + r0 * 3 * Hp,
                                            w.ghn + r0 * Hp, w.st + r0 * Hp, Hp, B, Hp, s));
            } else if (v > 0) {
                uint16_t* pre_gh = pre_of(w.gates, NG, r0 * 3 * Hp, hg);
                DVAE_TRY(lin_fwd(w.agg + r0 * Hp, Hp, nullptr, whh, Hp, nullptr,
                                 hg ? reinterpret_cast<float*>(pre_gh) : w.gates + r0 * 3 * Hp, 3 * Hp, B, 3 * Hp, Hp, 0, 0, s, hg,
                                 agg_h ? agg_h + r0 * Hp : nullptr));
                DVAE_TRY(gru_pointwise_fwd_h(hoff(w.gates, r0 * 3 * Hp, hg), gi, bv.types + r0, bhn, w.agg + r0 * Hp, Hp,
                                             w.st + r0 * Hp, Hp, hoff(w.ghn, r0 * Hp, hg), B, Hp, st_h ? st_h + r0 * Hp : nullptr, s,
                                             pre_gh));
            }
            if (by_level && v < n - 1) {
                // The state of level v is final: its rows of the EA table and its vertex-type rows
                // (graph state after v + 1 vertices) run now on the side stream, small launches that
                // fill the CUs the latency-bound chain leaves idle (as the second encoder direction does).
                hipStream_t s2 = side->stream;
                DVAE_TRY(fork_side(side_guard));
                DVAE_TRY(lin_fwd(w.st + r0 * Hp, Hp, nullptr, pw + L.w1a, Hp, nullptr,
                                 th ? reinterpret_cast<float*>(ea_h + r0 * HEp) : w.ea + r0 * HEp, HEp, B, HEp, Hp, 0, 0, s2, th,
                                 st_h ? st_h + r0 * Hp : nullptr));
                DVAE_TRY(dvae_gather_sum_f32(w.vs + r0 * Hp, Hp, w.st, Hp, bv.vs_off + r0, bv.vs_idx, B, Hp, 0, s2));
                DVAE_TRY(lin_fwd(w.vs + r0 * Hp, Hp, nullptr, pw + L.wv1, Hp, pw + L.bv1, w.v1 + r0 * d.HVp, d.HVp, B, d.HVp,
                                 Hp, 1, 0, s2));
            }
            if (v < n - 1) {
                float* gm = hoff(w.gm, r0 * 2 * Hp, hg);
                uint16_t* pre_gm = pre_of(w.gm, NM, r0 * 2 * Hp, hg);
                if (fused) {
                    DVAE_TRY(dvae_gated_level_fwd(w.st + r0 * Hp, Hp, pw + L.wgm[0], pw + L.gv[0] + (size_t)v * 2 * Hp, gm,
                                                  w.msg, Hp, (int)r0, w.agg + (r0 + B) * Hp, Hp, bv.pred_off + r0 + B,
                                                  bv.pred_idx, B, Hp, s));
                } else {
                    DVAE_TRY(lin_fwd(w.st + r0 * Hp, Hp, nullptr, pw + L.wgm[0], Hp, nullptr, hg ? reinterpret_cast<float*>(pre_gm) : gm,
                                     2 * Hp, B, 2 * Hp, Hp, 0, 0, s, hg, st_h ? st_h + r0 * Hp : nullptr));
                    // pointwise tail + predecessor sum of vertex v + 1
                    DVAE_TRY(gated_fwd_gather_h(gm, pw + L.gv[0] + (size_t)v * 2 * Hp, w.msg, Hp, (int)r0, w.agg + (r0 + B) * Hp,
                                                Hp, bv.pred_off + r0 + B, bv.pred_idx, B, Hp,
                                                agg_h ? agg_h + (r0 + B) * Hp : nullptr, s, pre_gm));
                }
            }
        }
        // partial states: non-recurrent, one batched pass
        const long p0 = nB;
        DVAE_TRY(gather_sum_h(w.agg + p0 * Hp, Hp, w.msg, Hp, bv.p_off, bv.p_idx, PT, Hp, 0, agg_h ? agg_h + p0 * Hp : nullptr, s));
        uint16_t* pre_p = pre_of(w.gates, NG, p0 * 3 * Hp, hg);
        DVAE_TRY(lin_fwd(w.agg + p0 * Hp, Hp, nullptr, whh, Hp, nullptr, hg ? reinterpret_cast<float*>(pre_p) : w.gates + p0 * 3 * Hp,
                         3 * Hp, PT, 3 * Hp, Hp, 0, 0, s, hg, agg_h ? agg_h + p0 * Hp : nullptr));
        DVAE_TRY(gru_pointwise_fwd_h(hoff(w.gates, p0 * 3 * Hp, hg), gi, bv.p_type, bhn, w.agg + p0 * Hp, Hp, w.st + p0 * Hp, Hp,
                                     hoff(w.ghn, p0 * Hp, hg), PT, Hp, st_h ? st_h + p0 * Hp : nullptr, s, pre_p));
    }
    // The loss terms meet through tickets (loss_kernels.hip: loss_ticket): the pair kernel's and the CE kernel's
    // workgroups each draw one, the last writes out3 -- so the vertex-type predictor's tail can run on the side stream.
    const int nblk = pair_blocks(c->NP, w.dw2_blocks);
    const unsigned tickets = (unsigned)(nblk + vertex_ce_blocks(c->R));
    auto vertex_tail = [&](hipStream_t st) -> int {
        // (the Tp-wide logits product stays one launch: a level's slice of it is 16 workgroups)
        DVAE_TRY(lin_fwd(w.v1, d.HVp, nullptr, pw + L.wv2, d.HVp, pw + L.bv2, w.logits, d.Tp, c->R, d.Tp, d.HVp, 0, 0, st));
        return vertex_ce_final(w.logits, d.Tp, bv.ce_off, bv.ce_tgt, w.dlogits, w.sums, tickets, beta, out3, c->R, d.T, d.Tp, st);
    };
    // (level-major batches: running this tail on the side stream under the products below was measured: 0.4 % slower
    // step -- the Tp-wide product and the CE kernel then share the CUs with the edge predictor's large product)
    // edge predictor: first layer split over its inputs
    if (!by_level) DVAE_TRY(lin_fwd(w.st, Hp, bv.rowsA, pw + L.w1a, Hp, nullptr, w.ea, HEp, c->NA, HEp, Hp, 0, 0, s, th, st_h));
    // (bf16 tables: b1 goes into EB here -- and into its no-input row -- instead of into every pair's sum)
    // (the row-GATHERED product keeps reading the fp32 states: on the bf16 shadow the gathering loader of
    // gemm_bf16_nn_kernel -- per-row 64-bit addresses, 80 bytes of scratch -- ran 6.7 ms against 5.1)
    DVAE_TRY(lin_fwd(w.st, Hp, bv.rowsB, pw + L.w1b, Hp, th ? pw + L.be1 : nullptr, w.eb, HEp, c->NB, HEp, Hp, 0, 0, s, th));
    if (th) DVAE_TRY(dvae_f32_to_bf16(pw + L.be1, eb_h + (size_t)c->NB * HEp, HEp, s));
    if (d.bn) DVAE_TRY(lin_fwd(w.agg, Hp, nullptr, pw + L.w1c, Hp, nullptr, w.ec, HEp, B, HEp, Hp, 0, 0, s));
    if (by_level) DVAE_TRY(join_side(side_guard));
    DVAE_TRY(pair_bce_fwd(w.ea, w.eb, w.ec, bv.pair_a, bv.pair_h, d.bn ? bv.pair_c : nullptr, bv.pair_t, pw + L.be1,
                          pw + L.we2, pw + L.be2, w.dl, w.sums + 0, w.mask, w.dw2p, nblk, c->NP, HEp, w.sums, tickets,
                          beta, out3, s, (pair_group_knob() && c->B >= w.dw2_blocks / 2) ? c->NP / c->B : 0, th));
    // (whole graphs per workgroup only when there are graphs enough to fill the grid: at batch 32 that mapping left 32
    // workgroups walking 28 pairs each, 63 us instead of ~12)
    // vertex-type predictor
    if (!by_level) {
        DVAE_TRY(dvae_gather_sum_f32(w.vs, Hp, w.st, Hp, bv.vs_off, bv.vs_idx, c->R, Hp, 0, s));
        DVAE_TRY(lin_fwd(w.vs, Hp, nullptr, pw + L.wv1, Hp, pw + L.bv1, w.v1, d.HVp, c->R, d.HVp, Hp, 1, 0, s));
    }
    DVAE_TRY(vertex_tail(s));
    return DVAE_OK;
}

extern "C" int dvae_loss_backward(const dvae_model_desc* m, const dvae_batch_counts* c, const int32_t* blob,
                                  const float* pw, float* ws, const float* mu, const float* logvar, const float* eps,
                                  float eps_scale, int32_t training, float beta, const float* g3, float* dmu,
                                  float* dlogvar, float* pg, void* stream) {
    Dims d;
    DVAE_TRY(make_dims(m, d));
    DVAE_TRY(check_batch(d, c));
    tl_bf16 = d.bf16;
    DVAE_REQUIRE(blob && pw && ws && mu && logvar && g3 && dmu && dlogvar && pg, "dvae_loss_backward: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const PackedLayout L = make_layout(d);
    ModeGuard mode_guard(d, L, pw);
    const BatchView bv = bind_batch(*c, blob);
    DecWs w = carve_dec(d, *c, ws);
    const int B = c->B, n = d.n, Hp = d.Hp, nB = c->nB, PT = c->PT, HEp = d.HEp, NST = nB + PT;
    const float* whh = pw + L.whh[2];
    const bool fused = chain_fused_bwd(d, c->B);

    DVAE_HIP(hipMemsetAsync(pg, 0, L.total * sizeof(float), s));
    const float* g_res = w.gs;
    const float* g_kld = w.gs + 1;
    // one launch: gs from g3, dlogits_s = dlogits * gs[0], and the rows of the two edge-predictor gradient tables that
    // the grouped passes below ADD into are zeroed (all other rows are written whole by their one chunk)
    DVAE_TRY(loss_bwd_init(g3, beta, w.gs, w.dlogits, w.dlogits_s, (long)c->R * d.Tp, bv.ca_row, c->CA, w.dea, c->NA,
                           bv.ch_row, c->CH, w.deb, c->NB, HEp, s));

    // ---- edge predictor ------------------------------------------------------------
    // the forward left the ReLU masks: each pass is one sweep over (dl, mask) and writes its table
    const bool dth = edge_dtables_bf16(d);
    // the forward's bf16 shadow of the states (written on the unfused path only): the weight gradients' other operand
    const uint16_t* const st_h = (!chain_fused(d, c->B) && !loss_fwd_tiny(d, *c)) ? w.st_h : nullptr;
    uint16_t* const dea_h = dth ? reinterpret_cast<uint16_t*>(w.ea) + (size_t)(c->NA + 1) * HEp : nullptr;
    uint16_t* const deb_h = dth ? reinterpret_cast<uint16_t*>(w.eb) + (size_t)(c->NB + 1) * HEp : nullptr;
    DVAE_TRY(pair_bce_bwd_grouped(bv.ca_off, bv.ca_row, bv.ca_pairs, w.dl, w.mask, pw + L.we2, g_res, w.dea, c->CA, HEp,
                                  s, dea_h));
    DVAE_TRY(pair_bce_bwd_grouped(bv.ch_off, bv.ch_row, bv.ch_pairs, w.dl, w.mask, pw + L.we2, g_res, w.deb, c->CH, HEp,
                                  s, deb_h));
    if (dth) DVAE_TRY(pair_bwd_fix(bv.ca_row, c->CA, w.dea, dea_h, c->NA, bv.ch_row, c->CH, w.deb, deb_h, c->NB, HEp, s));
    if (d.bn) {
        DVAE_HIP(hipMemsetAsync(w.dec, 0, sizeof(float) * (size_t)B * HEp, s));
        DVAE_TRY(dvae_pair_bce_bwd_grouped(bv.cc_off, bv.cc_row, nullptr, w.dl, w.mask, pw + L.we2, g_res, w.dec,
                                           c->CC, HEp, s));
    }
    // The predictors' weight gradients depend only on what is already computed; they run on the
    // side stream while the main stream continues into the (latency-bound) state chain.
    SideGuard side_guard(s);
    Side*& side = side_guard.sd;
    DVAE_TRY(side_acquire(side_guard));
    DVAE_HIP(hipMemsetAsync(w.dst, 0, sizeof(float) * (size_t)NST * Hp, s));
    // vertex-type predictor, data path (main stream)
    // (the ReLU backward of add_vertex's hidden layer is the product's epilogue: masked by v1 > 0)
    DVAE_TRY(lin_dgrad(w.dlogits_s, d.Tp, pw + L.wv2, d.HVp, w.dv1, d.HVp, nullptr, c->R, d.Tp, d.HVp, 0, s, w.v1));
    DVAE_TRY(fork_side(side_guard));
    {
        hipStream_t s2 = side->stream;
        // rows sorted by vertex type (off at xoh[0..T], idx from xoh[40]) for the GRU's input-side gradients, which
        // run on this stream at the end: a one-workgroup kernel bound by latency, kept off the main stream
        DVAE_TRY(dvae_type_csr(bv.types, nB, bv.p_type, PT, d.T, reinterpret_cast<int*>(w.xoh), reinterpret_cast<int*>(w.xoh) + 40, s2));
        const int nblk = pair_blocks(c->NP, w.dw2_blocks);  // as the forward launched them
        // the four bias-like gradients whose inputs are final here, as one launch: d we2 (the forward's partials are
        // unscaled), d be1, d bv2, d bv1
        const ColsumJobDesc bias_jobs[4] = {
            {pg + L.we2, w.dw2p, g_res, HEp, nblk, HEp, 0},
            {pg + L.be1, dth ? (const void*)dea_h : (const void*)w.dea, nullptr, HEp, c->NA + 1, HEp, dth ? 1 : 0},
            {pg + L.bv2, w.dlogits_s, nullptr, d.Tp, c->R, d.Tp, 0},
            {pg + L.bv1, w.dv1, nullptr, d.HVp, c->R, d.HVp, 0}};
        DVAE_TRY(colsum_jobs(bias_jobs, 4, s2));
        hipLaunchKernelGGL(sum_scaled_kernel, dim3(64), dim3(256), 0, s2, w.dl, c->NP, g_res, pg + L.be2);
        DVAE_LAUNCH_CHECK();
        DVAE_TRY(lin_wgrad(w.dea, HEp, w.st, Hp, bv.rowsA, pg + L.w1a, Hp, c->NA, HEp, Hp, s2, NST, 0, dea_h, dth ? st_h : nullptr));
        DVAE_TRY(lin_wgrad(w.deb, HEp, w.st, Hp, bv.rowsB, pg + L.w1b, Hp, c->NB, HEp, Hp, s2, NST, 0, deb_h, dth ? st_h : nullptr));
        if (d.bn) DVAE_TRY(lin_wgrad(w.dec, HEp, w.agg, Hp, nullptr, pg + L.w1c, Hp, B, HEp, Hp, s2));
        DVAE_TRY(lin_wgrad(w.dlogits_s, d.Tp, w.v1, d.HVp, nullptr, pg + L.wv2, d.HVp, c->R, d.Tp, d.HVp, s2));
        DVAE_TRY(lin_wgrad(w.dv1, d.HVp, w.vs, Hp, nullptr, pg + L.wv1, Hp, c->R, d.HVp, Hp, s2));
    }
    // data path continues on the main stream
    DVAE_TRY(lin_dgrad(w.dea, HEp, pw + L.w1a, Hp, w.dst, Hp, bv.rowsA, c->NA, HEp, Hp, 1, s, nullptr, dea_h, dth));
    DVAE_TRY(lin_dgrad(w.deb, HEp, pw + L.w1b, Hp, w.dst, Hp, bv.rowsB, c->NB, HEp, Hp, 1, s, nullptr, deb_h, dth));
    if (d.bn) DVAE_TRY(lin_dgrad(w.dec, HEp, pw + L.w1c, Hp, w.dh0c, Hp, nullptr, B, HEp, Hp, 0, s));
    DVAE_TRY(lin_dgrad(w.dv1, d.HVp, pw + L.wv1, Hp, w.dvs, Hp, nullptr, c->R, d.HVp, Hp, 0, s));
    DVAE_TRY(dvae_gather_sum_f32(w.dst, Hp, w.dvs, Hp, bv.vst_off, bv.vst_idx, nB, Hp, 1, s));

    // ---- states --------------------------------------------------------------------
    if (d.bn) {
        DVAE_TRY(dvae_gru_pointwise_bwd(w.dst, Hp, w.gates, w.ghn, w.agg, Hp, w.dgh, w.dgin, w.dagg, Hp, NST, Hp, s));
        DVAE_TRY(lin_dgrad(w.dgh, 3 * Hp, whh, Hp, w.dagg, Hp, nullptr, NST, 3 * Hp, Hp, 1, s));
        // message-table gradient from every row whose h_prev was a table sum (all but level 0)
        EW(count_kernel, (long)(nB - B) * d.Tp, w.cnt + (size_t)B * d.Tp, bv.pred_off + B, bv.pred_idx, nB - B, d.Tp);
        EW(count_kernel, (long)PT * d.Tp, w.cnt + (size_t)nB * d.Tp, bv.p_off, bv.p_idx, PT, d.Tp);
        DVAE_TRY(lin_wgrad(w.cnt + (size_t)B * d.Tp, d.Tp, w.dagg + (size_t)B * Hp, Hp, nullptr, pg + L.gtab[0], Hp,
                           NST - B, d.Tp, Hp, s));
        EW(add_inplace_kernel, (long)B * Hp, w.dagg, w.dh0c, (long)B * Hp);
    } else {
        const long p0 = nB;
        // bf16 shadows of d gh / d gm (act_shadows; the unfused path writes and reads them)
        uint16_t* const dgh_h = !fused ? w.dgh_h : nullptr;
        uint16_t* const dgm_h = !fused ? w.dgm_h : nullptr;
        const bool hg = half_gates(d, B);  // as dvae_loss_forward saved them (false whenever fused)
        const bool gh = half_grads(d, B);  // d gh / d gm as bf16 only
        DVAE_TRY(gru_pointwise_bwd_h(w.dst + p0 * Hp, Hp, hoff(w.gates, p0 * 3 * Hp, hg), hoff(w.ghn, p0 * Hp, hg), w.agg + p0 * Hp,
                                     Hp, gh ? nullptr : w.dgh + p0 * 3 * Hp, hoff(w.dgin, p0 * Hp, gh), w.dagg + p0 * Hp, Hp, PT, Hp,
                                     dgh_h ? dgh_h + p0 * 3 * Hp : nullptr, s, hg));
        DVAE_TRY(lin_dgrad(w.dgh + p0 * 3 * Hp, 3 * Hp, whh, Hp, w.dagg + p0 * Hp, Hp, nullptr, PT, 3 * Hp, Hp, 1, s, nullptr,
                           dgh_h ? dgh_h + p0 * 3 * Hp : nullptr, gh));
        // (only the last vertex's rows of d gm are not written by the chain below)
        DVAE_HIP(hipMemsetAsync(w.dgm + (size_t)(n - 1) * B * 2 * Hp, 0, sizeof(float) * (size_t)B * 2 * Hp, s));
        if (gh) DVAE_HIP(hipMemsetAsync(w.dgm_h + (size_t)(n - 1) * B * 2 * Hp, 0, sizeof(uint16_t) * (size_t)B * 2 * Hp, s));
        const uint16_t* const dgh_only = gh ? w.dgh_h : nullptr;
        const uint16_t* const dgm_only = gh ? w.dgm_h : nullptr;
        const uint16_t* const agg_sh = gh ? w.agg_h : nullptr;  // (level 0: converted by dvae_loss_forward)
        const uint16_t* const st_sh = gh ? w.st_h : nullptr;
        const bool tiny = fused && chain_tiny_persist(d, B);  // the reverse chain as one persistent launch
        const bool wbl = !tiny && wgrad_by_level() && wbl_size_ok(d, B);
        hipStream_t s2 = side->stream;
        if (tiny) {
            TinyChainArgs a = {};
            TinyChainDir& e = a.d[0];
            tiny_dir_fwd(e, w.agg, w.gates, w.ghn, w.st, w.gm, w.msg, whh, pw + L.wgm[0], pw + L.gi[2], pw + L.bhn[2],
                         pw + L.gv[0], nullptr, nullptr, 0, 1);
            e.dhs = w.dst; e.dgm = w.dgm; e.dgh = w.dgh; e.dgin = w.dgin; e.dagg = w.dagg;
            e.toff = bv.succ_off; e.tidx = bv.succ_idx;
            e.gsrc2 = w.dagg + p0 * Hp; e.off2 = bv.pt_off; e.idx2 = bv.pt_idx;
            a.types = bv.types; a.B = B; a.Hp = Hp; a.n = n; a.ndirs = 1;
            a.sync = reinterpret_cast<unsigned*>(w.sync) + TINY_CHAIN_SYNC_WORDS;  // slot 1
            a.status = a.sync + TINY_CHAIN_SYNC_WORDS - 64;
            a.sync_is_zero = loss_fwd_tiny(d, *c) ? 1 : 0;  // dvae_loss_forward's persistent launch zeroed it
            a.poison = pg + L.whh[2];  // a timed-out wait turns the gradients NaN (the weight-gradient products add to it)
            DVAE_TRY(tiny_chain_bwd(a, s));
        }
        if (wbl) {  // the partial rows' d gh is final: their term of dWhh starts the side stream's run beside the chain
            DVAE_TRY(fork_side(side_guard));
            DVAE_TRY(gru_whh_wgrad(d, L, 2, pg, w.dgh, w.agg, p0, NST, s2, 0, dgh_only, agg_sh));
        }
        int pend_h = 0, pend_g = 0;  // finished levels whose weight-gradient terms are not issued yet (wbl_group)
        long lo_g = 0;
        for (int v = n - 1; v >= 0 && !tiny; --v) {
            const long r0 = (long)v * B;
            float* dst = w.dst + r0 * Hp;
            if (v < n - 1 && !fused) {
                // d msg = consumers' d agg (successor slots + partial rows), used in registers
                DVAE_TRY(gated_bwd_gather_h(w.dagg, Hp, bv.succ_off + r0, bv.succ_idx, w.dagg + p0 * Hp, bv.pt_off + r0, bv.pt_idx,
                                            hoff(w.gm, r0 * 2 * Hp, hg), gh ? nullptr : w.dgm + r0 * 2 * Hp, B, Hp,
                                            dgm_h ? dgm_h + r0 * 2 * Hp : nullptr, s, hg));
                DVAE_TRY(lin_dgrad(w.dgm + r0 * 2 * Hp, 2 * Hp, pw + L.wgm[0], Hp, dst, Hp, nullptr, B, 2 * Hp, Hp, 1, s, nullptr,
                                   dgm_h ? dgm_h + r0 * 2 * Hp : nullptr, gh));
            }
            // (fused: d gm of vertex v was written by the tail of vertex v + 1's dgrad kernel, below)
            if (fused && v < n - 1)
                DVAE_TRY(dvae_gru_level_bwd(w.dgm + r0 * 2 * Hp, 2 * Hp, 2 * Hp, pw + L.wgm[0], dst, Hp, w.gates + r0 * 3 * Hp,
                                            w.ghn + r0 * Hp, w.agg + r0 * Hp, Hp, w.dgh + r0 * 3 * Hp, w.dgin + r0 * Hp,
                                            w.dagg + r0 * Hp, Hp, B, Hp, s));
            else
                DVAE_TRY(gru_pointwise_bwd_h(dst, Hp, hoff(w.gates, r0 * 3 * Hp, hg), hoff(w.ghn, r0 * Hp, hg), w.agg + r0 * Hp, Hp,
                                             gh ? nullptr : w.dgh + r0 * 3 * Hp, hoff(w.dgin, r0 * Hp, gh), w.dagg + r0 * Hp, Hp, B, Hp,
                                             dgh_h ? dgh_h + r0 * 3 * Hp : nullptr, s, hg));
            if (fused) {
                dvae_level_desc da = {};
                da.a = w.dgh + r0 * 3 * Hp; da.lda = 3 * Hp; da.K = 3 * Hp; da.w = whh;
                da.dx = w.dagg + r0 * Hp; da.ld_dx = Hp;
                if (v > 0) {  // same launch: d gm of vertex v - 1 from its consumers' d agg (successor slots + partial rows)
                    const long r1 = r0 - B;
                    da.gather_src = w.dagg; da.next_off = bv.succ_off + r1; da.idx = bv.succ_idx; da.cur_row0 = (int)r0;
                    da.gather_src2 = w.dagg + p0 * Hp; da.off2 = bv.pt_off + r1; da.idx2 = bv.pt_idx;
                    da.gm = w.gm + r1 * 2 * Hp; da.dgm_next = w.dgm + r1 * 2 * Hp;
                }
                DVAE_TRY(dvae_chain_level_batch(DVAE_LEVEL_DGRAD_ACC, &da, 1, B, Hp, s));
            } else
                DVAE_TRY(lin_dgrad(w.dgh + r0 * 3 * Hp, 3 * Hp, whh, Hp, w.dagg + r0 * Hp, Hp, nullptr, B, 3 * Hp, Hp, 1, s, nullptr,
                                   dgh_h ? dgh_h + r0 * 3 * Hp : nullptr, gh));
            if (wbl) {
                // final now: d gh of vertex v, and d gm of vertex v - 1 (fused: the tail above) or v (unfused: this iteration)
                const int vg = fused ? v - 1 : v;
                ++pend_h;  // levels v .. v + pend_h - 1
                if (vg >= 0 && vg < n - 1) {  // (d gm of the last vertex is zero: no message leaves it)
                    ++pend_g;  // levels vg .. vg + pend_g - 1
                    lo_g = vg;
                }
                if (pend_h >= wbl_group() || v == 0) {
                    DVAE_TRY(fork_side(side_guard));
                    DVAE_TRY(gru_whh_wgrad(d, L, 2, pg, w.dgh, w.agg, r0, r0 + (long)pend_h * B, s2, wbl_split(), dgh_only, agg_sh));
                    if (pend_g > 0)
                        DVAE_TRY(lin_wgrad(w.dgm + (size_t)lo_g * B * 2 * Hp, 2 * Hp, w.st + (size_t)lo_g * B * Hp, Hp, nullptr,
                                           pg + L.wgm[0], Hp, pend_g * B, 2 * Hp, Hp, s2, 0, wbl_split(),
                                           dgm_only ? dgm_only + (size_t)lo_g * B * 2 * Hp : nullptr,
                                           dgm_only ? st_sh + (size_t)lo_g * B * Hp : nullptr));
                    pend_h = pend_g = 0;
                }
            }
        }
    }
    // the GRU's remaining weight gradients go to the side stream, behind the predictors' and the per-level terms
    DVAE_TRY(fork_side(side_guard));
    const bool wbl_done = !d.bn && wgrad_by_level() && wbl_size_ok(d, B) && !(fused && chain_tiny_persist(d, B));
    const bool gh_all = !d.bn && half_grads(d, B);
    if (wbl_done) DVAE_TRY(gru_type_wgrads(d, L, 2, pg, w.dgh, w.dgin, w.xoh, NST, side->stream, gh_all ? w.dgh_h : nullptr));
    else DVAE_TRY(gru_wgrads(d, L, 2, pg, w.dgh, w.dgin, w.agg, w.xoh, NST, side->stream, 0, 0, gh_all ? w.dgh_h : nullptr, w.agg_h));
    // ---- H0 = tanh(fc3 z), z = reparameterize(mu, logvar) ------------------------------
    EW(tanh_bwd_kernel, (long)B * Hp, w.dagg, w.agg, (long)B * Hp);
    DVAE_TRY(dvae_segment_colsum_f32(pg + L.b3, Hp, w.dagg, Hp, B, B, Hp, s));
    DVAE_TRY(lin_wgrad(w.dagg, Hp, w.z, d.NZp, nullptr, pg + L.fc3, d.NZp, B, Hp, d.NZp, s));
    DVAE_TRY(lin_dgrad(w.dagg, Hp, pw + L.fc3, d.NZp, w.dz, d.NZp, nullptr, B, Hp, d.NZp, 0, s));
    DVAE_TRY(reparam_kld_bwd(mu, logvar, eps, eps_scale, training, w.dz, g_kld, dmu, dlogvar, B, d.nz, d.NZp, s));
    if (!d.bn) {
        // (d gm of the last vertex is zero: left out)
        if (!wbl_done)
            DVAE_TRY(lin_wgrad(w.dgm, 2 * Hp, w.st, Hp, nullptr, pg + L.wgm[0], Hp, nB - B, 2 * Hp, Hp, s, 0, 0, gh_all ? w.dgm_h : nullptr,
                               gh_all ? w.st_h : nullptr));
        if (gh_all) DVAE_TRY(segment_colsum_h(pg + L.gv[0], 2 * Hp, w.dgm_h, 2 * Hp, nB, B, 2 * Hp, s));
        else DVAE_TRY(dvae_segment_colsum_f32(pg + L.gv[0], 2 * Hp, w.dgm, 2 * Hp, nB, B, 2 * Hp, s));
    }
    DVAE_TRY(join_side(side_guard));
    return DVAE_OK;
}
