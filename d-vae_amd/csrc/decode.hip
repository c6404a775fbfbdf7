// Generation on the GPU: DVAE.decode (models.py:747-804; DVAE_BN inherits it with its own
// message table, graph state and edge scorer, models.py:1006-1085).
//
// The reference grows B igraph objects vertex by vertex in Python, with one GRUCell call per
// (new vertex, candidate predecessor) and host-side sampling.  Here the B graphs live in device
// arrays (types, predecessor bitmasks, sizes, finished flags) and the whole autoregressive
// loop is enqueued without a host round trip:
//   per new vertex idx:  type logits (2 GEMMs) -> pick_type_kernel (softmax + inverse-CDF
//                        sample / argmax / forced END)
//   per candidate vi:    EB = Hcur W1b^T (GEMM) -> edge_step_kernel (score, decision, graph and
//                        predecessor-sum update, one wavefront per graph) -> GRU (GEMM + pointwise)
//   once per vertex:     gate/mapper message and the EA = Hd W1a^T row block (the first edge-
//                        predictor layer is split over its inputs as in the train step).
// Random draws are INJECTED as uniforms (u_type[b][idx], u_edge[b][idx][vi]) so that a run is
// replayable and checkable against the oracle; the host fills them from torch's device RNG.
#include "common.h"
#include "model.h"
#include "orch.h"
#include <mutex>

namespace dvae {

struct DecState {
    int32_t* types;      // [B][n]
    uint64_t* pmask;     // [B][n]  bit u of pmask[b][v] <=> edge u -> v
    int32_t* ng;         // [B]
    int32_t* fin;        // [B] graph closed by an END vertex
    int32_t* active;     // [B] the vertex being added exists in this graph
    int32_t* cur_type;   // [B] its type
    int32_t* lastrow;    // [B] state row of the graph's last vertex (slot = v * B + b)
    uint64_t* hasout;    // [B] bit u <=> vertex u has an out-edge
};

__global__ void decode_init_kernel(DecState st, const float* __restrict__ z, float* __restrict__ zp, int B, int n,
                                   int nz, int NZp, int start_type) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B * NZp) {
        int b = i / NZp, j = i % NZp;
        zp[i] = j < nz ? z[b * nz + j] : 0.f;
    }
    if (i < B * n) {
        st.types[i] = (i % n == 0) ? start_type : 0;
        st.pmask[i] = 0;
    }
    if (i < B) {
        st.ng[i] = 1;
        st.fin[i] = 0;
        st.active[i] = 1;
        st.cur_type[i] = start_type;
        st.lastrow[i] = i;
        st.hasout[i] = 0;
    }
}

// Type of vertex idx for every open graph.  stochastic: inverse-CDF sample of softmax(logits) with
// the uniform u (numpy's legacy choice: float64 CDF normalised by its last entry, searchsorted
// side='right'); else argmax (first maximum, as torch.argmax); forced != 0: END for all.
__global__ void pick_type_kernel(DecState st, const float* __restrict__ logits, int ld, const float* __restrict__ u_type,
                                 int B, int n, int T, int idx, int stochastic, int forced, int end_type) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (st.fin[b]) {
        st.active[b] = 0;
        return;
    }
    int t = end_type;
    if (!forced) {
        const float* x = logits + (long)b * ld;
        float mx = x[0];
        int am = 0;
        for (int k = 1; k < T; ++k)
            if (x[k] > mx) { mx = x[k]; am = k; }
        if (!stochastic) {
            t = am;
        } else {
            float se = 0.f;
            for (int k = 0; k < T; ++k) se += expf(x[k] - mx);
            double tot = 0.0;
            for (int k = 0; k < T; ++k) tot += (double)(expf(x[k] - mx) / se);
            const double u = (double)u_type[(long)b * n + idx];
            double c = 0.0;
            t = T - 1;
            for (int k = 0; k < T; ++k) {
                c += (double)(expf(x[k] - mx) / se);
                if (u < c / tot) { t = k; break; }
            }
        }
    }
    st.types[(long)b * n + idx] = t;
    st.ng[b] = idx + 1;
    st.cur_type[b] = t;
    st.active[b] = 1;
    st.lastrow[b] = idx * B + b;
}

// One candidate predecessor vi of the vertex idx being added, one wavefront per graph:
//   p = sigmoid(b2 + w2 . relu(EA[vi][b] + EB[b] (+ EC[b]) + b1))        models.py:742-745 / 1082-1085
//   edge vi -> idx  iff  u < p (stochastic) / p > 0.5                     models.py:782-786
//   an END vertex instead takes every loose end (out-degree 0) as predecessor at its first
//   candidate and closes the graph                                        models.py:787-794
// and the predecessor sum S[b] (the GRU's h_prev of models.py:661-670) follows the new edges:
// DVAE adds the stored message row of vi, DVAE_BN the table row of vi's type.
__global__ __launch_bounds__(256) void edge_step_kernel(DecState st, const float* __restrict__ ea,
                                                        const float* __restrict__ eb, const float* __restrict__ ec,
                                                        const float* __restrict__ b1, const float* __restrict__ w2,
                                                        const float* __restrict__ b2, const float* __restrict__ u_edge,
                                                        const float* __restrict__ msg, const float* __restrict__ gtab,
                                                        float* __restrict__ S, int B, int n, int Hp, int HEp, int idx,
                                                        int vi, int stochastic, int end_type) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    if (!st.active[b]) return;
    auto msg_row = [&](int u) -> const float* {
        return gtab ? gtab + (long)st.types[(long)b * n + u] * Hp : msg + ((long)u * B + b) * Hp;
    };
    float* Sb = S + (long)b * Hp;
    if (st.cur_type[b] == end_type) {
        if (vi != idx - 1) return;
        const uint64_t loose = ~st.hasout[b] & ((idx >= 64) ? ~0ull : ((1ull << idx) - 1ull));
        for (int c = lane * 4; c < Hp; c += 256) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int u = 0; u < idx; ++u) {
                if ((loose >> u) & 1ull) {
                    float4 m = *reinterpret_cast<const float4*>(msg_row(u) + c);
                    acc.x += m.x; acc.y += m.y; acc.z += m.z; acc.w += m.w;
                }
            }
            *reinterpret_cast<float4*>(Sb + c) = acc;
        }
        if (lane == 0) {
            st.pmask[(long)b * n + idx] = loose;
            st.hasout[b] |= loose;
            st.fin[b] = 1;
        }
        return;
    }
    const float* ra = ea + ((long)vi * B + b) * HEp;
    const float* rb = eb + (long)b * HEp;
    const float* rc = ec ? ec + (long)b * HEp : nullptr;
    float s = 0.f;
    for (int c = lane * 4; c < HEp; c += 256) {
        float4 a = *reinterpret_cast<const float4*>(ra + c), h = *reinterpret_cast<const float4*>(rb + c);
        float4 bb = *reinterpret_cast<const float4*>(b1 + c), w = *reinterpret_cast<const float4*>(w2 + c);
        float4 pre = make_float4(a.x + h.x + bb.x, a.y + h.y + bb.y, a.z + h.z + bb.z, a.w + h.w + bb.w);
        if (rc) {
            float4 cc = *reinterpret_cast<const float4*>(rc + c);
            pre.x += cc.x; pre.y += cc.y; pre.z += cc.z; pre.w += cc.w;
        }
        s += w.x * fmaxf(pre.x, 0.f) + w.y * fmaxf(pre.y, 0.f) + w.z * fmaxf(pre.z, 0.f) + w.w * fmaxf(pre.w, 0.f);
    }
    s = wave_sum(s) + b2[0];
    const float prob = 1.f / (1.f + expf(-s));
    const bool take = stochastic ? (u_edge[((long)b * n + idx) * n + vi] < prob) : (prob > 0.5f);
    if (!take) return;  // uniform across the wave
    const float* mr = msg_row(vi);
    for (int c = lane * 4; c < Hp; c += 256) {
        float4 m = *reinterpret_cast<const float4*>(mr + c);
        float4 v = *reinterpret_cast<float4*>(Sb + c);
        v.x += m.x; v.y += m.y; v.z += m.z; v.w += m.w;
        *reinterpret_cast<float4*>(Sb + c) = v;
    }
    if (lane == 0) {
        st.pmask[(long)b * n + idx] |= 1ull << vi;
        st.hasout[b] |= 1ull << vi;
    }
}

// DVAE_BN graph state (models.py:1006-1027 with decode=True): running sum of the vertex states
__global__ void add_active_rows_kernel(float* __restrict__ acc, const float* __restrict__ rows,
                                       const int32_t* __restrict__ active, int B, int Hp) {
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)B * Hp) return;
    if (active[i / Hp]) acc[i] += rows[i];
}

struct GenWs {
    float *zp, *h0, *hd, *msg, *gm, *ea, *ec, *eb, *S, *gates, *ghn, *vsum, *v1, *logits;
    int32_t* ints;
    size_t total;
};
static GenWs carve_gen(const Dims& d, int B, float* base) {
    GenWs w;
    Bump b{base, 0};
    const size_t nB = (size_t)d.n * B, Hp = d.Hp, HEp = d.HEp;
    w.zp = b.take((size_t)B * d.NZp);
    w.h0 = b.take((size_t)B * Hp);
    w.hd = b.take(nB * Hp);
    w.msg = d.bn ? nullptr : b.take(nB * Hp);
    w.gm = d.bn ? nullptr : b.take((size_t)B * 2 * Hp);
    w.ea = b.take(nB * HEp);
    w.ec = d.bn ? b.take((size_t)B * HEp) : nullptr;
    w.eb = b.take((size_t)B * HEp);
    w.S = b.take((size_t)B * Hp);
    w.gates = b.take((size_t)B * 3 * Hp);
    w.ghn = b.take((size_t)B * Hp);
    w.vsum = d.bn ? b.take((size_t)B * Hp) : nullptr;
    w.v1 = b.take((size_t)B * d.HVp);
    w.logits = b.take((size_t)B * d.Tp);
    w.ints = reinterpret_cast<int32_t*>(b.take((size_t)B * 8));  // fin, active, cur_type, lastrow, hasout (64-bit)
    w.total = b.off;
    return w;
}

}  // namespace dvae

using namespace dvae;

extern "C" int64_t dvae_decode_ws_floats(const dvae_model_desc* m, int32_t B) {
    Dims d;
    if (make_dims(m, d) != 0 || B <= 0) return -1;
    return (int64_t)carve_gen(d, B, nullptr).total;
}

static int decode_impl(const dvae_model_desc* m, int32_t B, const float* pw, float* ws, const float* z,
                       int32_t stochastic, const float* u_type, const float* u_edge, int32_t* types,
                       uint64_t* pred_mask, int32_t* n_vertices, void* stream) {
    Dims d;
    DVAE_TRY(make_dims(m, d));
    DVAE_REQUIRE(B > 0, "dvae_decode: empty batch");
    DVAE_REQUIRE(pw && ws && z && types && pred_mask && n_vertices, "dvae_decode: null pointer");
    DVAE_REQUIRE(!stochastic || (u_type && u_edge), "dvae_decode: stochastic decoding needs the uniform draws");
    DVAE_REQUIRE(m->end_type >= 0 && m->end_type < d.T && d.start_type >= 0 && d.start_type < d.T,
                 "dvae_decode: START/END type out of range");
    tl_bf16 = d.bf16;
    hipStream_t s = (hipStream_t)stream;
    const PackedLayout L = make_layout(d);
    ModeGuard mode_guard(d, L, pw);
    GenWs w = carve_gen(d, B, ws);
    const int n = d.n, Hp = d.Hp, HEp = d.HEp, end_type = m->end_type;
    DecState st;
    st.types = types;
    st.pmask = pred_mask;
    st.ng = n_vertices;
    st.fin = w.ints;
    st.active = w.ints + B;
    st.cur_type = w.ints + 2 * (size_t)B;
    st.lastrow = w.ints + 3 * (size_t)B;
    st.hasout = reinterpret_cast<uint64_t*>(w.ints + 4 * (size_t)B);  // 256-byte aligned carve, 4 B ints -> 8-byte aligned
    const float* whh = pw + L.whh[2];
    const float* gi = pw + L.gi[2];
    const float* bhn = pw + L.bhn[2];
    const float* gtab = d.bn ? pw + L.gtab[0] : nullptr;

    {
        long tot = (long)B * (d.NZp > n ? d.NZp : n);
        hipLaunchKernelGGL(decode_init_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, s, st, z, w.zp, B, n, d.nz, d.NZp,
                           d.start_type);
        DVAE_LAUNCH_CHECK();
    }
    // per finished vertex: its message (DVAE) and its row block of the split edge-predictor layer
    auto finish_vertex = [&](int v) -> int {
        const float* hv = w.hd + (size_t)v * B * Hp;
        if (!d.bn) {
            DVAE_TRY(lin_fwd(hv, Hp, nullptr, pw + L.wgm[0], Hp, nullptr, w.gm, 2 * Hp, B, 2 * Hp, Hp, 0, 0, s));
            DVAE_TRY(dvae_gated_pointwise_fwd(w.gm, pw + L.gv[0] + (size_t)v * 2 * Hp, w.msg + (size_t)v * B * Hp, Hp, B,
                                              Hp, 0, B, s));
        } else {
            hipLaunchKernelGGL(add_active_rows_kernel, dim3(cdiv((long)B * Hp, 256)), dim3(256), 0, s, w.vsum, hv,
                               st.active, B, Hp);
            DVAE_LAUNCH_CHECK();
        }
        DVAE_TRY(lin_fwd(hv, Hp, nullptr, pw + L.w1a, Hp, nullptr, w.ea + (size_t)v * B * HEp, HEp, B, HEp, Hp, 0, 0, s));
        return DVAE_OK;
    };
    // vertex 0: H0 = tanh(fc3 z), Hd[0] = grud(onehot(START), H0)        models.py:750-754
    DVAE_TRY(lin_fwd(w.zp, d.NZp, nullptr, pw + L.fc3, d.NZp, pw + L.b3, w.h0, Hp, B, Hp, d.NZp, 2, 0, s));
    DVAE_TRY(lin_fwd(w.h0, Hp, nullptr, whh, Hp, nullptr, w.gates, 3 * Hp, B, 3 * Hp, Hp, 0, 0, s));
    DVAE_TRY(dvae_gru_pointwise_fwd(w.gates, gi, st.cur_type, bhn, w.h0, Hp, w.hd, Hp, w.ghn, B, Hp, s));
    if (d.bn) {
        DVAE_TRY(lin_fwd(w.h0, Hp, nullptr, pw + L.w1c, Hp, nullptr, w.ec, HEp, B, HEp, Hp, 0, 0, s));
        DVAE_HIP(hipMemsetAsync(w.vsum, 0, sizeof(float) * (size_t)B * Hp, s));
    }
    DVAE_TRY(finish_vertex(0));

    for (int idx = 1; idx < n; ++idx) {
        const int forced = idx == n - 1;
        if (!forced) {  // type logits from the graph state: last vertex (DVAE) / sum of states (BN)
            if (d.bn) DVAE_TRY(lin_fwd(w.vsum, Hp, nullptr, pw + L.wv1, Hp, pw + L.bv1, w.v1, d.HVp, B, d.HVp, Hp, 1, 0, s));
            else DVAE_TRY(lin_fwd(w.hd, Hp, st.lastrow, pw + L.wv1, Hp, pw + L.bv1, w.v1, d.HVp, B, d.HVp, Hp, 1, 0, s));
            DVAE_TRY(lin_fwd(w.v1, d.HVp, nullptr, pw + L.wv2, d.HVp, pw + L.bv2, w.logits, d.Tp, B, d.Tp, d.HVp, 0, 0, s));
        }
        hipLaunchKernelGGL(pick_type_kernel, dim3(cdiv(B, 256)), dim3(256), 0, s, st, w.logits, d.Tp, u_type, B, n, d.T,
                           idx, stochastic, forced, end_type);
        DVAE_LAUNCH_CHECK();
        // new vertex with no predecessor yet: Hcur = grud(x, 0)            models.py:773
        float* hcur = w.hd + (size_t)idx * B * Hp;
        DVAE_HIP(hipMemsetAsync(w.S, 0, sizeof(float) * (size_t)B * Hp, s));
        DVAE_HIP(hipMemsetAsync(w.gates, 0, sizeof(float) * (size_t)B * 3 * Hp, s));
        DVAE_TRY(dvae_gru_pointwise_fwd(w.gates, gi, st.cur_type, bhn, w.S, Hp, hcur, Hp, w.ghn, B, Hp, s));
        for (int vi = idx - 1; vi >= 0; --vi) {
            DVAE_TRY(lin_fwd(hcur, Hp, nullptr, pw + L.w1b, Hp, nullptr, w.eb, HEp, B, HEp, Hp, 0, 0, s));
            hipLaunchKernelGGL(edge_step_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, st, w.ea, w.eb, w.ec, pw + L.be1,
                               pw + L.we2, pw + L.be2, u_edge, w.msg, gtab, w.S, B, n, Hp, HEp, idx, vi, stochastic,
                               end_type);
            DVAE_LAUNCH_CHECK();
            DVAE_TRY(lin_fwd(w.S, Hp, nullptr, whh, Hp, nullptr, w.gates, 3 * Hp, B, 3 * Hp, Hp, 0, 0, s));
            DVAE_TRY(dvae_gru_pointwise_fwd(w.gates, gi, st.cur_type, bhn, w.S, Hp, hcur, Hp, w.ghn, B, Hp, s));
        }
        if (!forced) DVAE_TRY(finish_vertex(idx));
    }
    return DVAE_OK;
}

extern "C" int dvae_decode(const dvae_model_desc* m, int32_t B, const float* pw, float* ws, const float* z,
                           int32_t stochastic, const float* u_type, const float* u_edge, int32_t* types,
                           uint64_t* pred_mask, int32_t* n_vertices, void* stream) {
    return decode_impl(m, B, pw, ws, z, stochastic, u_type, u_edge, types, pred_mask, n_vertices, stream);
}

// ---------------------------------------------------------------------------------------------
// dvae_decode_static: the same generation with the workspace taken from a fixed device array that
// lives in the library's code object, for callers that cannot allocate (see include/dvae_hip.h).
// `scratch` is a host-side stand-in for the symbol: its device address is looked up once per device
// with hipGetSymbolAddress (every GPU holds its own copy at its own address) and only that address
// is ever handed to kernels or memsets.
// ---------------------------------------------------------------------------------------------
static __device__ __align__(256) float scratch[DVAE_DECODE_STATIC_FLOATS];

// Per device: the array's address, and who used it last.  Calls that overlap on one GPU share the array and would
// race; calls on ONE stream are ordered by the stream and are fine.  So each call leaves an event behind its work, and
// a call that arrives on ANOTHER stream while that event has not completed is refused (an error, not a race).
// Set-up (first call on a device): hipGetSymbolAddress + one hipEventCreate; after that a call allocates nothing.
// Under stream capture no event is left or queried (a captured record cannot be queried): there the ordering of
// captured calls is the graph's, i.e. the capturing caller's.
struct StaticScratch {
    float* ptr = nullptr;
    hipEvent_t done = nullptr;
    hipStream_t last_stream = nullptr;
    bool pending = false;
};
static StaticScratch g_static[64];
static std::mutex g_static_mu;

static int static_scratch_for_current_device(StaticScratch** out) {
    int dev = 0;
    DVAE_HIP(hipGetDevice(&dev));
    DVAE_REQUIRE(dev >= 0 && dev < 64, "device index %d out of range", dev);
    StaticScratch& st = g_static[dev];
    if (!st.ptr) {
        float* p = nullptr;
        DVAE_HIP(hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch)));
        DVAE_REQUIRE(p != nullptr && ((uintptr_t)p % 256) == 0, "static scratch: bad symbol address");
        DVAE_HIP(hipEventCreateWithFlags(&st.done, hipEventDisableTiming));
        st.ptr = p;
    }
    *out = &st;
    return DVAE_OK;
}

extern "C" int64_t dvae_decode_static_max_floats(void) { return (int64_t)DVAE_DECODE_STATIC_FLOATS; }

extern "C" int dvae_decode_static(const dvae_model_desc* m, int32_t B, const float* pw, const float* z,
                                  int32_t stochastic, const float* u_type, const float* u_edge, int32_t* types,
                                  uint64_t* pred_mask, int32_t* n_vertices, void* stream) {
    const int64_t need = dvae_decode_ws_floats(m, B);
    DVAE_REQUIRE(need > 0, "dvae_decode_static: bad model descriptor or batch size %d", B);
    DVAE_REQUIRE(need <= (int64_t)DVAE_DECODE_STATIC_FLOATS,
                 "dvae_decode_static: %d graphs of this model need a workspace of %lld floats, the static scratch holds "
                 "%lld; decode fewer graphs per call or use dvae_decode with a caller-provided workspace",
                 B, (long long)need, (long long)DVAE_DECODE_STATIC_FLOATS);
    std::lock_guard<std::mutex> lock(g_static_mu);  // check, enqueue and record as one step against other host threads
    StaticScratch* st = nullptr;
    DVAE_TRY(static_scratch_for_current_device(&st));
    hipStream_t s = (hipStream_t)stream;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    const bool capturing = cap != hipStreamCaptureStatusNone;
    if (st->pending && !capturing && st->last_stream != s) {
        const hipError_t q = hipEventQuery(st->done);
        if (q == hipErrorNotReady)
            return set_err(DVAE_ERR_ARG, "dvae_decode_static: an earlier call on another stream of this GPU is still running; "
                                         "overlapping calls share the static scratch and would race -- order them on one "
                                         "stream, wait for the earlier call, or use dvae_decode with workspaces of your own");
        if (q != hipSuccess) (void)hipGetLastError();
        st->pending = false;
    }
    // (nothing in the workspace has to start at zero: decode_impl writes every word it later reads --
    // running sums and gate buffers are cleared per vertex on the stream -- so the previous call's
    // contents are harmless)
    const int rc = decode_impl(m, B, pw, st->ptr, z, stochastic, u_type, u_edge, types, pred_mask, n_vertices, stream);
    if (!capturing) {  // also after a failed call: whatever it enqueued before failing still uses the array
        DVAE_HIP(hipEventRecord(st->done, s));
        st->pending = true;
        st->last_stream = s;
    }
    return rc;
}
