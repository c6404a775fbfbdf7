// Model dimensions, parameter indices and the packed (zero-padded) weight layout.
#pragma once
#include <stddef.h>
#include "common.h"
#include "dvae_hip.h"

namespace dvae {

// state_dict order of models.DVAE / DVAE_BN
enum ParamId {
    P_EF_WIH = 0, P_EF_WHH, P_EF_BIH, P_EF_BHH,
    P_EB_WIH, P_EB_WHH, P_EB_BIH, P_EB_BHH,
    P_FC1_W, P_FC1_B, P_FC2_W, P_FC2_B,
    P_D_WIH, P_D_WHH, P_D_BIH, P_D_BHH,
    P_FC3_W, P_FC3_B,
    P_AV0_W, P_AV0_B, P_AV2_W, P_AV2_B,
    P_AE0_W, P_AE0_B, P_AE2_W, P_AE2_B,
    P_GF_W, P_GF_B, P_GB_W, P_GB_B,
    P_MF_W, P_MB_W,
    P_HV_W, P_HV_B, P_HG_W, P_HG_B,
    P_COUNT
};
static_assert(P_COUNT == DVAE_NUM_PARAMS, "parameter table size");

struct Dims {
    int n, T, H, nz, bidir, vid, bn, start_type, bf16;
    int Hp, NZp, Tp, HVp, HEp, V;
};

static inline int make_dims(const dvae_model_desc* m, Dims& d) {
    DVAE_REQUIRE(m != nullptr, "null model descriptor");
    DVAE_REQUIRE(m->max_n >= 2 && m->max_n <= 64, "max_n must be in 2..64 (got %d)", m->max_n);
    DVAE_REQUIRE(m->nvt >= 2 && m->nvt <= 32, "nvt must be in 2..32 (got %d)", m->nvt);
    DVAE_REQUIRE(m->hs >= 1 && m->nz >= 1, "hs and nz must be positive");
    d.n = m->max_n; d.T = m->nvt; d.H = m->hs; d.nz = m->nz;
    d.bidir = m->bidir != 0; d.bn = m->is_bn != 0; d.vid = (m->vid != 0) && !d.bn;
    d.start_type = m->start_type;
    d.bf16 = m->mfma_bf16 != 0;
    d.Hp = round_up(d.H, 32);
    d.NZp = round_up(d.nz, 32);
    d.Tp = 32;
    d.HVp = 2 * d.Hp;
    d.HEp = d.bn ? d.Hp : 4 * d.Hp;
    d.V = d.bn ? d.T : (d.vid ? d.H + d.n : d.H);
    return DVAE_OK;
}

// offsets (in floats) into the packed weight buffer; the packed GRADIENT buffer
// uses the same layout (gi -> d gi_tab, gv -> d gv, gtab -> d gtab, ...).
struct PackedLayout {
    size_t whh[3], gi[3], bhn[3];        // GRUs: 0 enc fwd, 1 enc bwd, 2 dec.  [3Hp][Hp], [Tp][3Hp], [Hp]
    size_t wgm[2], gv[2];                // gate|mapper h-part [2Hp][Hp], vertex-id table [n][2Hp]   (DVAE)
    size_t gtab[2], gtg[2], gtm[2];      // BN message tables [Tp][Hp]: g*m, g, m
    size_t fc12, b12, fc3, b3;           // [2NZp][Hp], [2NZp], [Hp][NZp], [Hp]
    size_t wv1, bv1, wv2, bv2;           // [2Hp][Hp], [2Hp], [Tp][2Hp], [Tp]
    size_t w1a, w1b, w1c, be1, we2, be2; // [HEp][Hp] x3, [HEp], [HEp], [4]
    size_t wu, bu;                       // [Hp][2Hp], [Hp]
    size_t total;                        // floats of the fp32 part (also the size of the packed GRADIENT buffer's live part)
    // bf16 mode only (dvae_model_desc.mfma_bf16): bf16 operand copies behind the fp32 part, rebuilt by every (re)pack
    //   half  (float offset): bf16 copy of [0, total), element for element
    //   halfT (float offset): the matrices below once more, TRANSPOSED ([cols][rows] bf16), so that a data gradient
    //                         dX = dY W is a product of two K-contiguous operands too (gemm_bf16_nn.hip)
    struct Mat { size_t off; int rows, cols; size_t toff; };  // off: floats into the fp32 part; toff: bf16 elements into halfT
    Mat mats[20];
    int nmats;
    size_t half, halfT;
    size_t alloc;                        // floats of the whole buffer (== total in fp32 mode)
};

static inline PackedLayout make_layout(const Dims& d) {
    PackedLayout L;
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o += (n + 63) & ~(size_t)63; return r; };
    const size_t Hp = d.Hp;
    for (int g = 0; g < 3; ++g) { L.whh[g] = take(3 * Hp * Hp); L.gi[g] = take((size_t)d.Tp * 3 * Hp); L.bhn[g] = take(Hp); }
    for (int k = 0; k < 2; ++k) {
        L.wgm[k] = take(2 * Hp * Hp);
        L.gv[k] = take((size_t)d.n * 2 * Hp);
        L.gtab[k] = take((size_t)d.Tp * Hp);
        L.gtg[k] = take((size_t)d.Tp * Hp);
        L.gtm[k] = take((size_t)d.Tp * Hp);
    }
    L.fc12 = take((size_t)2 * d.NZp * Hp); L.b12 = take(2 * d.NZp);
    L.fc3 = take(Hp * d.NZp); L.b3 = take(Hp);
    L.wv1 = take((size_t)d.HVp * Hp); L.bv1 = take(d.HVp);
    L.wv2 = take((size_t)d.Tp * d.HVp); L.bv2 = take(d.Tp);
    L.w1a = take((size_t)d.HEp * Hp); L.w1b = take((size_t)d.HEp * Hp); L.w1c = take((size_t)d.HEp * Hp);
    L.be1 = take(d.HEp); L.we2 = take(d.HEp); L.be2 = take(4);
    L.wu = take(Hp * 2 * Hp); L.bu = take(Hp);
    L.total = o;
    L.nmats = 0; L.half = L.halfT = 0; L.alloc = o;
    if (d.bf16) {
        size_t t = 0;
        auto mat = [&](size_t off, size_t rows, size_t cols) {
            L.mats[L.nmats++] = PackedLayout::Mat{off, (int)rows, (int)cols, t};
            t += (rows * cols + 63) & ~(size_t)63;
        };
        for (int g = 0; g < 3; ++g) mat(L.whh[g], 3 * Hp, Hp);
        for (int k = 0; k < 2; ++k) mat(L.wgm[k], 2 * Hp, Hp);
        mat(L.fc12, 2 * d.NZp, Hp); mat(L.fc3, Hp, d.NZp);
        mat(L.wv1, d.HVp, Hp); mat(L.wv2, d.Tp, d.HVp);
        mat(L.w1a, d.HEp, Hp); mat(L.w1b, d.HEp, Hp); mat(L.w1c, d.HEp, Hp);
        mat(L.wu, Hp, 2 * Hp);
        L.half = take(L.total / 2);
        L.halfT = take((t + 1) / 2);
        L.alloc = o;
    }
    return L;
}

// internal kernels shared between translation units
int gemm(const dvae_gemm_desc& d, hipStream_t s, int max_split = 0);  // max_split > 0 caps the split-K factor
int gemm_bf16(const dvae_gemm_desc& d, hipStream_t s);
int gemm_skinny(const dvae_gemm_desc& d, hipStream_t s, int force_nt);  // gemm_skinny.hip: products of few rows; 1 = not served there
int scale_inplace(float* x, const float* s, int n, hipStream_t st);
int gru_first_level(float* gates, const float* gi_tab, const int32_t* types, const float* bhn, float* agg_zero, float* hout,
                    float* ghn_out, int rows, int Hp, hipStream_t s, uint16_t* hout_h = nullptr, int hg = 0);
// the pointwise kernels with a bf16 SHADOW of their output beside it (pointwise.hip; nullable); pre / hg: the HALF-GATES
// form (bf16 pre-activations in, bf16 saved gates: pointwise.hip ldg4 / stg4, train.hip half_gates())
int gather_sum_h(float* out, int ld_out, const float* src, int ld_src, const int* off, const int* idx, int rows, int width,
                 int accumulate, uint16_t* out_h, hipStream_t s);
int gru_pointwise_fwd_h(float* gh, const float* gi_tab, const int* types, const float* bhn, const float* hprev, int ld_hprev,
                        float* hout, int ld_hout, float* ghn_out, int rows, int Hp, uint16_t* hout_h, hipStream_t s,
                        const uint16_t* pre = nullptr);
int gru_pointwise_bwd_h(const float* dh, int ld_dh, const float* gates, const float* ghn, const float* hprev, int ld_hprev,
                        float* dgh, float* dgin, float* dhprev, int ld_dhprev, int rows, int Hp, uint16_t* dgh_h, hipStream_t s,
                        int hg = 0);
int gated_bwd_gather_h(const float* dagg, int ld, const int* off, const int* idx, const float* dagg2, const int* off2,
                       const int* idx2, const float* gm, float* dgm, int rows, int Hp, uint16_t* dgm_h, hipStream_t s, int hg = 0);
int gated_fwd_gather_h(float* gm, const float* gv_row, float* msg, int ld_msg, int cur_row0, float* next_agg, int ld_agg,
                       const int* next_off, const int* idx, int rows, int Hp, uint16_t* next_agg_h, hipStream_t s,
                       const uint16_t* pre = nullptr);
int loss_fwd_init(const float* mu, const float* logvar, const float* eps, float eps_scale, int training, float* z,
                  float* sums, float* zero_row_a, float* zero_row_b, int HEp, int B, int nz, int NZp, hipStream_t s);
int vertex_ce_blocks(int R);
int vertex_ce_final(const float* logits, int ld, const int* ce_off, const int* ce_tgt, float* dlogits, float* sums,
                    unsigned tickets, float beta, float* out3, int R, int T, int Tp, hipStream_t s);
int pair_bce_fwd(const float* ea, const float* eb, const float* ec, const int32_t* pair_a, const int32_t* pair_h,
                 const int32_t* pair_c, const uint8_t* pair_t, const float* b1, const float* w2, const float* b2, float* dl,
                 float* loss_sum, uint32_t* mask, float* dw2_partial, int32_t n_blocks, int32_t n_pairs, int32_t HEp,
                 float* sums, unsigned tickets, float beta, float* out3, void* stream, int group = 0, int tables_bf16 = 0);
int loss_bwd_init(const float* g3, float beta, float* gs, const float* dlogits, float* dlogits_s, long n_logits,
                  const int* ca_row, int CA, float* dea, int NA, const int* ch_row, int CH, float* deb, int NB, int HEp,
                  hipStream_t s);
int pair_bce_bwd_grouped(const int32_t* chunk_off, const int32_t* chunk_row, const int32_t* grp_pairs, const float* dl,
                         const uint32_t* mask, const float* w2, const float* scale, float* dtab, int32_t n_chunks,
                         int32_t HEp, hipStream_t stream, uint16_t* dtab_h);
int pair_bwd_fix(const int* ca_row, int CA, const float* dea, uint16_t* dea_h, int NA, const int* ch_row, int CH,
                 const float* deb, uint16_t* deb_h, int NB, int HEp, hipStream_t s);
struct ColsumJobDesc {  // out[c] += scale * sum over rows of x[row][c]   (pointwise.hip: colsum_jobs)
    float* out;
    const void* x;       // fp32, or bf16 words when x_bf16
    const float* scale;  // nullable: one device float
    int ld_x, rows, width, x_bf16;
};
int colsum_jobs(const ColsumJobDesc* jobs, int n, hipStream_t s);
int type_colsum_h(const float* dgh, const float* dgin, const int32_t* off, const int32_t* idx, int32_t T, float* gi,
                  float* bhn, int32_t rows, int32_t Hp, hipStream_t stream, int gh);
int segment_colsum_h(float* out, int ld_out, const uint16_t* x, int ld_x, int rows, int seg_rows, int width, hipStream_t s);
int reparam_kld_fwd(const float* mu, const float* logvar, const float* eps, float eps_scale, int training, float* z,
                    float* kld_sum, int B, int nz, int NZp, hipStream_t s, float* zero_a = nullptr, float* zero_b = nullptr,
                    int HEp = 0);
int reparam_kld_bwd(const float* mu, const float* logvar, const float* eps, float eps_scale, int training,
                    const float* dz, const float* gk, float* dmu, float* dlogvar, int B, int nz, int NZp, hipStream_t s);

int chain_small_rows();  // chain.hip: levels of at most this many rows run chain_small_kernel
int chain_tiny_rows();   // chain.hip: levels of at most this many rows run chain_tiny_kernel (16 x 16 tiles)
bool chain_bf16_ok(int rows, int Hp);  // chain.hip: does the bf16 level kernel serve levels of this shape?
void chain_bf16_context(const float* pw, const PackedLayout* L);  // chain.hip: the bf16-mode call in progress (NULL: none)
// chain.hip: the encoder's forward chain as one persistent launch (DVAE_CHAIN_PERSIST=1)
constexpr int ENC_CHAIN_SYNC_WORDS = 2 * 256 * 32 + 64;  // [2 directions][<= 256 row tiles][32] counters, then the status word
struct EncChainDir {
    float *agg, *gates, *ghn, *hs, *gm, *msg;
    const float *whh, *wgm, *gi, *bhn, *gv;
    const int *off, *idx;  // CSR of the level's sources: predecessors (forward pass) / successors (backward pass)
};
struct EncChainArgs {
    EncChainDir d[2];
    const int* types;
    int B, Hp, n, ndirs;
    unsigned* sync;   // [ndirs][tiles_y][32] counters (128 bytes apart), zeroed before the launch
    unsigned* status; // [1], zeroed before the launch; != 0: a wait timed out
};
int enc_chain_fwd(const EncChainArgs& a, hipStream_t s);

// chain_persist.hip: a whole pass (encoder direction(s) or the decoder chain, forward or backward) of a SMALL batch
// as one persistent launch of 16 x 16 tiles
constexpr int TINY_CHAIN_SYNC_WORDS = 2 * 16 * 32 + 64;  // [2 passes][<= 16 row tiles][32] counters, then the status word
struct TinyChainDir {
    // arrays of the forward pass, [n * B rows], level-major
    float *agg, *gates, *ghn, *hs, *gm, *msg;
    const float *whh, *wgm, *gi, *bhn, *gv;
    const int *off, *idx;    // CSR over slots of a level's source slots (predecessors; the encoder's backward direction: successors)
    // arrays of the backward pass
    float *dhs, *dgm, *dgh, *dgin, *dagg;
    const int *toff, *tidx;  // the transposed lists: the consumers of a slot's message
    const float* gsrc2; const int *off2, *idx2;  // decoder: d agg of the partial rows and slot -> partial rows (else NULL)
    int reverse;             // 1: the pass walks the vertices n-1 .. 0 (the encoder's backward direction)
    int level0_product;      // 1: the pass's first level has a non-zero h_prev (the decoder's H0)
};
struct TinyChainArgs {
    TinyChainDir d[2];
    const int* types;
    int B, Hp, n, ndirs;
    unsigned* sync;    // TINY_CHAIN_SYNC_WORDS words; zeroed by the launcher unless `sync_is_zero`
    unsigned* status;  // == sync + TINY_CHAIN_SYNC_WORDS - 64; != 0 after the launch: a wait timed out
    float* poison;     // nullable: receives + NaN when a wait times out
    unsigned* zero_next;  // nullable: TINY_CHAIN_SYNC_WORDS words this launch zeroes for the pass's NEXT launch (the forward
                          // launch prepares the backward launch's counters: one memset per pass and step instead of two)
    int sync_is_zero;     // 1: `sync` was zeroed by the previous launch's zero_next
};
bool tiny_chain_fits(int B, int Hp, int passes);
int tiny_chain_fwd(const TinyChainArgs& a, hipStream_t s);
int tiny_chain_bwd(const TinyChainArgs& a, hipStream_t s);

}  // namespace dvae
