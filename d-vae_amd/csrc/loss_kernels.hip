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

// ---------------------------------------------------------------------------
// forward: one wave per pair (grid-stride)
__global__ __launch_bounds__(256) void pair_fwd_kernel(const float* __restrict__ ea, const float* __restrict__ eb,
                                                       const float* __restrict__ ec, const int* __restrict__ pair_a,
                                                       const int* __restrict__ pair_h, const int* __restrict__ pair_c,
                                                       const uint8_t* __restrict__ pair_t, const float* __restrict__ b1,
                                                       const float* __restrict__ w2, const float* __restrict__ b2,
                                                       float* __restrict__ dl, float* __restrict__ loss_sum,
                                                       int n_pairs, int HEp) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int waves_total = gridDim.x * 4;
    float local = 0.f;
    const float bias2 = b2[0];
    for (int p = blockIdx.x * 4 + wave; p < n_pairs; p += waves_total) {
        const float* ra = ea + (long)pair_a[p] * HEp;
        const float* rb = eb + (long)pair_h[p] * HEp;
        const float* rc = ec ? ec + (long)pair_c[p] * HEp : nullptr;
        float s = 0.f;
        for (int c = lane * 4; c < HEp; c += 256) {
            float4 a = ld4c(ra + c), b = ld4c(rb + c), bb = ld4c(b1 + c), w = ld4c(w2 + c);
            float4 pre = make_float4(a.x + b.x + bb.x, a.y + b.y + bb.y, a.z + b.z + bb.z, a.w + b.w + bb.w);
            if (rc) {
                float4 cc = ld4c(rc + c);
                pre.x += cc.x; pre.y += cc.y; pre.z += cc.z; pre.w += cc.w;
            }
            s += w.x * fmaxf(pre.x, 0.f) + w.y * fmaxf(pre.y, 0.f) + w.z * fmaxf(pre.z, 0.f) + w.w * fmaxf(pre.w, 0.f);
        }
        s = wave_sum(s) + bias2;
        if (lane == 0) {
            // torch.sigmoid then F.binary_cross_entropy on the probability (models.py:745, :855)
            float prob = 1.f / (1.f + expf(-s));
            float t = (float)pair_t[p];
            float lp = fmaxf(logf(prob), -100.f);
            float lq = fmaxf(logf(1.f - prob), -100.f);
            local += -(t * lp + (1.f - t) * lq);
            // BCE backward: (p - t) / max(p (1-p), 1e-12), then sigmoid backward: * p (1-p)
            float pq = prob * (1.f - prob);
            dl[p] = (prob - t) / fmaxf(pq, 1e-12f) * pq;
        }
    }
    if (lane == 0) red[wave] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        float v = red[0] + red[1] + red[2] + red[3];
        if (v != 0.f) atomicAdd(loss_sum, v);
    }
}

// ---------------------------------------------------------------------------
// grouped backward: one wave per chunk of pairs that share the same `mine` row.
//   chunk q covers grp_pairs[chunk_off[q] .. chunk_off[q+1]) (grp_pairs == NULL:
//   the pair ids are the positions themselves); chunk_row[q] = destination row,
//   bit 31 set when several chunks share the row (-> atomicAdd into zeroed dmine).
__global__ __launch_bounds__(256) void pair_bwd_kernel(const float* __restrict__ mine,
                                                       const int* __restrict__ chunk_off,
                                                       const int* __restrict__ chunk_row,
                                                       const int* __restrict__ grp_pairs,
                                                       const float* __restrict__ oth1, const int* __restrict__ oth1_idx,
                                                       const float* __restrict__ oth2, const int* __restrict__ oth2_idx,
                                                       const float* __restrict__ b1, const float* __restrict__ w2,
                                                       const float* __restrict__ dl, const float* __restrict__ scale,
                                                       float* __restrict__ dmine, float* __restrict__ dw2_partial,
                                                       int n_chunks, int HEp) {
    __shared__ float4 red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int waves_total = gridDim.x * 4;
    const float sc = scale ? scale[0] : 1.f;
    for (int c0 = 0; c0 < HEp; c0 += 256) {  // uniform trip count: the loop body holds barriers
        const bool act = (c0 + lane * 4) < HEp;
        const int c = act ? c0 + lane * 4 : 0;
        float4 bb = ld4c(b1 + c), w = ld4c(w2 + c);
        float4 aw = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int q = blockIdx.x * 4 + wave; q < n_chunks; q += waves_total) {
            int cr = chunk_row[q];
            int row = cr & 0x7fffffff;
            float4 m = ld4c(mine + (long)row * HEp + c);
            m.x += bb.x; m.y += bb.y; m.z += bb.z; m.w += bb.w;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            int k1 = chunk_off[q + 1];
            for (int k = chunk_off[q]; k < k1; ++k) {
                int p = grp_pairs ? grp_pairs[k] : k;
                float g = dl[p] * sc;
                float4 o = ld4c(oth1 + (long)oth1_idx[p] * HEp + c);
                float4 pre = make_float4(m.x + o.x, m.y + o.y, m.z + o.z, m.w + o.w);
                if (oth2) {
                    float4 o2 = ld4c(oth2 + (long)oth2_idx[p] * HEp + c);
                    pre.x += o2.x; pre.y += o2.y; pre.z += o2.z; pre.w += o2.w;
                }
                acc.x += pre.x > 0.f ? g * w.x : 0.f;
                acc.y += pre.y > 0.f ? g * w.y : 0.f;
                acc.z += pre.z > 0.f ? g * w.z : 0.f;
                acc.w += pre.w > 0.f ? g * w.w : 0.f;
                aw.x += g * fmaxf(pre.x, 0.f);
                aw.y += g * fmaxf(pre.y, 0.f);
                aw.z += g * fmaxf(pre.z, 0.f);
                aw.w += g * fmaxf(pre.w, 0.f);
            }
            if (act) {
                float* d = dmine + (long)row * HEp + c;
                if (cr < 0) {
                    atomicAdd(d + 0, acc.x); atomicAdd(d + 1, acc.y); atomicAdd(d + 2, acc.z); atomicAdd(d + 3, acc.w);
                } else {
                    *reinterpret_cast<float4*>(d) = acc;
                }
            }
        }
        if (dw2_partial) {  // uniform branch
            red[wave][lane] = aw;
            __syncthreads();
            if (wave == 0 && act) {
                float4 a = red[0][lane], b = red[1][lane], cc = red[2][lane], dd = red[3][lane];
                float4 s = make_float4(a.x + b.x + cc.x + dd.x, a.y + b.y + cc.y + dd.y, a.z + b.z + cc.z + dd.z,
                                       a.w + b.w + cc.w + dd.w);
                *reinterpret_cast<float4*>(dw2_partial + (long)blockIdx.x * HEp + c) = s;
            }
            __syncthreads();
        }
    }
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
                                       float* __restrict__ z, float* __restrict__ kld_sum, int B, int nz, int NZp) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
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

}  // namespace dvae

using namespace dvae;

extern "C" int dvae_pair_bce_fwd(const float* ea, const float* eb, const float* ec, const int32_t* pair_a,
                                 const int32_t* pair_h, const int32_t* pair_c, const uint8_t* pair_t, const float* b1,
                                 const float* w2, const float* b2, float* dl, float* loss_sum, int32_t n_pairs,
                                 int32_t HEp, void* stream) {
    if (n_pairs == 0) return DVAE_OK;
    DVAE_REQUIRE(ea && eb && pair_a && pair_h && pair_t && b1 && w2 && b2 && dl && loss_sum, "pair_bce_fwd: null pointer");
    DVAE_REQUIRE((ec == nullptr) == (pair_c == nullptr), "pair_bce_fwd: ec and pair_c go together");
    DVAE_REQUIRE(HEp > 0 && HEp % 4 == 0, "pair_bce_fwd: HEp must be a multiple of 4");
    int blocks = cdiv(n_pairs, 4);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(pair_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ea, eb, ec, pair_a, pair_h,
                       pair_c, pair_t, b1, w2, b2, dl, loss_sum, n_pairs, HEp);
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}

extern "C" int dvae_pair_bce_bwd_grouped(const float* mine, const int32_t* chunk_off, const int32_t* chunk_row,
                                         const int32_t* grp_pairs, const float* oth1, const int32_t* oth1_idx,
                                         const float* oth2, const int32_t* oth2_idx, const float* b1, const float* w2,
                                         const float* dl, const float* scale, float* dmine, float* dw2_partial,
                                         int32_t n_chunks, int32_t HEp, int32_t n_blocks, void* stream) {
    if (n_chunks == 0) return DVAE_OK;
    DVAE_REQUIRE(mine && chunk_off && chunk_row && oth1 && oth1_idx && b1 && w2 && dl && dmine, "pair_bce_bwd: null pointer");
    DVAE_REQUIRE((oth2 == nullptr) == (oth2_idx == nullptr), "pair_bce_bwd: oth2 and oth2_idx go together");
    DVAE_REQUIRE(HEp > 0 && HEp % 4 == 0 && n_blocks > 0, "pair_bce_bwd: bad size");
    hipLaunchKernelGGL(pair_bwd_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, mine, chunk_off, chunk_row,
                       grp_pairs, oth1, oth1_idx, oth2, oth2_idx, b1, w2, dl, scale, dmine, dw2_partial, n_chunks, HEp);
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
int reparam_kld_fwd(const float* mu, const float* logvar, const float* eps, float eps_scale, int training, float* z,
                    float* kld_sum, int B, int nz, int NZp, hipStream_t s) {
    hipLaunchKernelGGL(reparam_kld_fwd_kernel, dim3(cdiv((long)B * NZp, 256)), dim3(256), 0, s, mu, logvar, eps,
                       eps_scale, training, z, kld_sum, B, nz, NZp);
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
