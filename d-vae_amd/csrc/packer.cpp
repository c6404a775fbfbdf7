// Host-side DAG batcher: turns a batch of DAGs (types + predecessor bitmasks)
// into the index tables the HIP kernels consume ("padded CSR of predecessors by
// topological level").  Replaces the reference's batching, which is a Python
// list of igraph copies (models.py:631-632), per-vertex filtering / gathering /
// zero-padding of predecessor states (models.py:637-668) and the per-pair loop
// structure of the teacher-forced decoder (models.py:816-856).
//
// Level == vertex index, as in the reference (models.py:677: vertex ids are a
// topological order).  slot(v, b) = v * B + b.
//
// Decoder tables implement the de-duplication of SURVEY.md section 3.3: the
// "current vertex" state seen by pair (v, vi) is grud(x_v, sum of messages of the
// predecessors u > vi), so vertex v has only indeg(v) + 1 distinct such states:
//   j = 0 predecessors included  -> depends on type(v) only: one row per TYPE
//   0 < j < indeg                -> a "partial" row of its own
//   j = indeg                    -> the final state of v (a slot row)
#include <vector>
#include <new>
#include <string.h>
#include "common.h"
#include "batch.h"

namespace dvae {

struct PackedBatch {
    dvae_batch_counts c;
    std::vector<int32_t> arr[34];
    std::vector<uint8_t> pair_t;
};

static inline int popc(uint64_t x) { return __builtin_popcountll(x); }

static void chunkify(const std::vector<int32_t>& key, int n_rows, int chunk, std::vector<int32_t>& c_off,
                     std::vector<int32_t>& c_row, std::vector<int32_t>& pairs) {
    // counting sort of pair ids by key (row), then split every row's list into chunks
    int np = (int)key.size();
    std::vector<int32_t> cnt(n_rows + 1, 0);
    for (int p = 0; p < np; ++p) cnt[key[p] + 1]++;
    for (int r = 0; r < n_rows; ++r) cnt[r + 1] += cnt[r];
    pairs.assign(np, 0);
    std::vector<int32_t> pos(cnt.begin(), cnt.end() - 1);
    for (int p = 0; p < np; ++p) pairs[pos[key[p]]++] = p;
    c_off.clear(); c_row.clear();
    for (int r = 0; r < n_rows; ++r) {
        int b = cnt[r], e = cnt[r + 1];
        if (b == e) continue;
        bool multi = (e - b) > chunk;
        for (int s = b; s < e; s += chunk) {
            c_off.push_back(s);
            c_row.push_back(multi ? (int32_t)(r | 0x80000000u) : r);
        }
    }
    c_off.push_back(np);
}

static int build(PackedBatch& pb, int B, int n, int T, int start_type, int is_bn, const int32_t* types,
                 const uint64_t* pmask, const int32_t* n_g, int chunk) {
    DVAE_REQUIRE(B > 0 && n >= 2 && n <= 64, "pack_batch: need B > 0 and 2 <= max_n <= 64 (B=%d max_n=%d)", B, n);
    DVAE_REQUIRE(T > 0 && T <= 32, "pack_batch: number of vertex types must be in 1..32 (got %d)", T);
    DVAE_REQUIRE(start_type >= 0 && start_type < T, "pack_batch: START_TYPE out of range");
    DVAE_REQUIRE((long)B * n < (1L << 30), "pack_batch: batch too large");
    const int nB = n * B;
    for (int b = 0; b < B; ++b) {
        int ng = n_g[b];
        DVAE_REQUIRE(ng >= 1 && ng <= n, "pack_batch: graph %d has %d vertices, max_n is %d", b, ng, n);
        for (int v = 0; v < ng; ++v) {
            int t = types[(long)b * n + v];
            DVAE_REQUIRE(t >= 0 && t < T, "pack_batch: graph %d vertex %d has type %d outside [0, %d)", b, v, t, T);
            uint64_t m = pmask[(long)b * n + v];
            uint64_t allowed = v == 0 ? 0 : ((v >= 64 ? ~0ull : ((1ull << v) - 1)));
            DVAE_REQUIRE((m & ~allowed) == 0, "pack_batch: graph %d vertex %d has a predecessor with id >= %d (vertex ids must be a topological order)", b, v, v);
        }
        for (int v = ng; v < n; ++v)
            DVAE_REQUIRE(pmask[(long)b * n + v] == 0, "pack_batch: graph %d has edges into absent vertex %d", b, v);
    }
    auto TY = [&](int b, int v) { return types[(long)b * n + v]; };
    auto PM = [&](int b, int v) { return pmask[(long)b * n + v]; };
    auto slot = [&](int v, int b) { return v * B + b; };

    std::vector<uint64_t> smask((size_t)B * n, 0);
    for (int b = 0; b < B; ++b)
        for (int v = 0; v < n_g[b]; ++v) {
            uint64_t m = PM(b, v);
            while (m) {
                int u = __builtin_ctzll(m);
                m &= m - 1;
                smask[(long)b * n + u] |= 1ull << v;
            }
        }

    dvae_batch_counts& c = pb.c;
    memset(&c, 0, sizeof(c));
    c.B = B; c.n = n; c.T = T; c.is_bn = is_bn; c.nB = nB;

    // ---- 0: types, 1-4: pred / succ CSR --------------------------------------
    auto& a_types = pb.arr[0];
    a_types.assign(nB, 0);
    auto &pred_off = pb.arr[1], &pred_idx = pb.arr[2], &succ_off = pb.arr[3], &succ_idx = pb.arr[4];
    pred_off.assign(nB + 1, 0);
    succ_off.assign(nB + 1, 0);
    for (int v = 0; v < n; ++v)
        for (int b = 0; b < B; ++b) {
            int s = slot(v, b);
            bool ex = v < n_g[b];
            a_types[s] = ex ? TY(b, v) : 0;
            pred_off[s + 1] = pred_off[s] + (ex ? popc(PM(b, v)) : 0);
            succ_off[s + 1] = succ_off[s] + (ex ? popc(smask[(long)b * n + v]) : 0);
        }
    c.E = pred_off[nB];
    pred_idx.resize(c.E);
    succ_idx.resize(c.E);
    for (int v = 0; v < n; ++v)
        for (int b = 0; b < B; ++b) {
            if (v >= n_g[b]) continue;
            int s = slot(v, b);
            int k = pred_off[s];
            uint64_t m = PM(b, v);
            while (m) {
                int u = __builtin_ctzll(m);
                m &= m - 1;
                pred_idx[k++] = is_bn ? TY(b, u) : slot(u, b);
            }
            k = succ_off[s];
            m = smask[(long)b * n + v];
            while (m) {
                int w = __builtin_ctzll(m);
                m &= m - 1;
                succ_idx[k++] = is_bn ? TY(b, w) : slot(w, b);
            }
        }

    // ---- 5-8: graph-state source lists (models.py:704-718, BN 1006-1027 start=1,end_offset=1)
    auto &hgf_off = pb.arr[5], &hgf_idx = pb.arr[6], &hgb_off = pb.arr[7], &hgb_idx = pb.arr[8];
    hgf_off.assign(1, 0);
    hgb_off.assign(1, 0);
    for (int b = 0; b < B; ++b) {
        int ng = n_g[b];
        if (is_bn) {
            for (int v = 1; v < ng - 1; ++v) { hgf_idx.push_back(slot(v, b)); hgb_idx.push_back(slot(v, b)); }
        } else {
            hgf_idx.push_back(slot(ng - 1, b));
            hgb_idx.push_back(slot(0, b));
        }
        hgf_off.push_back((int32_t)hgf_idx.size());
        hgb_off.push_back((int32_t)hgb_idx.size());
    }
    c.HGF = (int32_t)hgf_idx.size();
    c.HGB = (int32_t)hgb_idx.size();

    // ---- 9-13: partial-state rows ---------------------------------------------
    auto &p_off = pb.arr[9], &p_idx = pb.arr[10], &p_type = pb.arr[11], &pt_off = pb.arr[12], &pt_idx = pb.arr[13];
    p_off.assign(1, 0);
    for (int t = 0; t < T; ++t) { p_type.push_back(t); p_off.push_back(0); }
    // first_partial[(b, v)] = row id of the j = 1 partial row of vertex v (rows j = 1..k-1 are consecutive)
    std::vector<int32_t> first_partial((size_t)B * n, -1);
    for (int v = 1; v < n; ++v)
        for (int b = 0; b < B; ++b) {
            if (v >= n_g[b]) continue;
            uint64_t m = PM(b, v);
            int k = popc(m);
            if (k < 2) continue;
            first_partial[(long)b * n + v] = (int32_t)p_type.size();
            int preds[64], np_ = 0;
            uint64_t mm = m;
            while (mm) { int u = 63 - __builtin_clzll(mm); mm &= ~(1ull << u); preds[np_++] = u; }  // descending
            for (int j = 1; j < k; ++j) {
                for (int i = 0; i < j; ++i) p_idx.push_back(is_bn ? TY(b, preds[i]) : slot(preds[i], b));
                p_off.push_back((int32_t)p_idx.size());
                p_type.push_back(TY(b, v));
            }
        }
    c.PT = (int32_t)p_type.size();
    c.PL = (int32_t)p_idx.size();
    pt_off.assign(nB + 1, 0);
    pt_idx.assign(c.PL, 0);
    if (!is_bn) {
        for (int k = 0; k < c.PL; ++k) pt_off[p_idx[k] + 1]++;
        for (int s = 0; s < nB; ++s) pt_off[s + 1] += pt_off[s];
        std::vector<int32_t> pos(pt_off.begin(), pt_off.end() - 1);
        for (int r = 0; r < c.PT; ++r)
            for (int k = p_off[r]; k < p_off[r + 1]; ++k) pt_idx[pos[p_idx[k]]++] = r;
    }

    // ---- pairs ----------------------------------------------------------------
    const int PPG = n * (n - 1) / 2;
    c.NP = B * PPG;
    const int ST_ZERO = -1;
    std::vector<int32_t> a_st(c.NP), h_st(c.NP);
    pb.pair_t.assign(c.NP, 0);
    auto& pair_c = pb.arr[18];
    pair_c.resize(c.NP);
    for (int b = 0; b < B; ++b) {
        int ng = n_g[b];
        int p = b * PPG;
        for (int v = 1; v < n; ++v) {
            uint64_t m = v < ng ? PM(b, v) : 0;
            int k = popc(m);
            for (int vi = 0; vi < v; ++vi, ++p) {
                pair_c[p] = b;
                if (v < ng) {
                    uint64_t above = vi >= 63 ? 0 : (m & ~((2ull << vi) - 1));
                    int j = popc(above);
                    int hs;
                    if (j == 0) hs = nB + TY(b, v);
                    else if (j < k) hs = nB + first_partial[(long)b * n + v] + (j - 1);
                    else hs = slot(v, b);
                    h_st[p] = hs;
                    a_st[p] = slot(vi, b);
                    pb.pair_t[p] = (uint8_t)((m >> vi) & 1);
                } else {
                    h_st[p] = ST_ZERO;
                    a_st[p] = vi < ng ? slot(vi, b) : ST_ZERO;
                }
            }
        }
    }
    // EA / EB row numbering in increasing ST-row order
    const int NST = nB + c.PT;
    std::vector<int32_t> mapA(nB, -1), mapB(NST, -1);
    for (int p = 0; p < c.NP; ++p) {
        if (a_st[p] >= 0) mapA[a_st[p]] = 0;
        if (h_st[p] >= 0) mapB[h_st[p]] = 0;
    }
    auto &rowsA = pb.arr[14], &rowsB = pb.arr[15];
    for (int s = 0; s < nB; ++s) if (mapA[s] == 0) { mapA[s] = (int32_t)rowsA.size(); rowsA.push_back(s); }
    for (int s = 0; s < NST; ++s) if (mapB[s] == 0) { mapB[s] = (int32_t)rowsB.size(); rowsB.push_back(s); }
    c.NA = (int32_t)rowsA.size();
    c.NB = (int32_t)rowsB.size();
    auto &pair_a = pb.arr[16], &pair_h = pb.arr[17];
    pair_a.resize(c.NP);
    pair_h.resize(c.NP);
    for (int p = 0; p < c.NP; ++p) {
        pair_a[p] = a_st[p] >= 0 ? mapA[a_st[p]] : c.NA;  // row NA / NB = the all-zero row
        pair_h[p] = h_st[p] >= 0 ? mapB[h_st[p]] : c.NB;
    }
    chunkify(pair_a, c.NA + 1, chunk, pb.arr[20], pb.arr[21], pb.arr[22]);
    chunkify(pair_h, c.NB + 1, chunk, pb.arr[23], pb.arr[24], pb.arr[25]);
    c.CA = (int32_t)pb.arr[21].size();
    c.CH = (int32_t)pb.arr[24].size();
    // by graph: pairs of graph b are contiguous [b*PPG, (b+1)*PPG)
    {
        auto &cc_off = pb.arr[26], &cc_row = pb.arr[27];
        int cchunk = chunk * 4;
        for (int b = 0; b < B; ++b) {
            bool multi = PPG > cchunk;
            for (int s = 0; s < PPG; s += cchunk) {
                cc_off.push_back(b * PPG + s);
                cc_row.push_back(multi ? (int32_t)(b | 0x80000000u) : b);
            }
        }
        cc_off.push_back(c.NP);
        c.CC = (int32_t)cc_row.size();
    }

    // ---- vertex-type rows (models.py:820-826; BN graph state :1006-1027 start=0) ----
    auto &vs_off = pb.arr[28], &vs_idx = pb.arr[29], &vst_off = pb.arr[30], &vst_idx = pb.arr[31];
    auto &ce_off = pb.arr[32], &ce_tgt = pb.arr[33];
    vs_off.assign(1, 0);
    ce_off.assign(1, 0);
    auto push_row = [&](int b, int upto /* states 0..upto-1 exist */) {
        if (is_bn) for (int u = 0; u < upto; ++u) vs_idx.push_back(slot(u, b));
        else vs_idx.push_back(slot(upto - 1, b));
        vs_off.push_back((int32_t)vs_idx.size());
    };
    for (int v = 1; v < n; ++v)
        for (int b = 0; b < B; ++b) {
            if (v >= n_g[b]) continue;
            push_row(b, v);
            ce_tgt.push_back(TY(b, v));
            ce_off.push_back((int32_t)ce_tgt.size());
        }
    for (int b = 0; b < B; ++b) {
        int ng = n_g[b];
        if (ng >= n) continue;
        push_row(b, ng);  // all padded steps see the same graph state, target START_TYPE
        for (int v = ng; v < n; ++v) ce_tgt.push_back(start_type);
        ce_off.push_back((int32_t)ce_tgt.size());
    }
    c.R = (int32_t)vs_off.size() - 1;
    c.RL = (int32_t)vs_idx.size();
    c.NCE = (int32_t)ce_tgt.size();
    vst_off.assign(nB + 1, 0);
    vst_idx.assign(c.RL, 0);
    for (int k = 0; k < c.RL; ++k) vst_off[vs_idx[k] + 1]++;
    for (int s = 0; s < nB; ++s) vst_off[s + 1] += vst_off[s];
    {
        std::vector<int32_t> pos(vst_off.begin(), vst_off.end() - 1);
        for (int r = 0; r < c.R; ++r)
            for (int k = vs_off[r]; k < vs_off[r + 1]; ++k) vst_idx[pos[vs_idx[k]]++] = r;
    }

    // level-major batch: all graphs full size, EA rows = the slots of levels 0..n-2 in order, one
    // vertex-type row per (level, graph) in level order
    {
        bool reg = !is_bn && n >= 2;
        for (int b = 0; b < B && reg; ++b) reg = n_g[b] == n;
        reg = reg && c.NA == (n - 1) * B && c.R == (n - 1) * B && c.RL == c.R;
        for (int i = 0; i < c.NA && reg; ++i) reg = rowsA[i] == i;
        for (int i = 0; i < c.R && reg; ++i) reg = vs_idx[i] == i;
        c.level_major = reg ? 1 : 0;
    }
    c.total_ints = walk_batch(c, [](int, int64_t, int64_t) {});
    // sanity: every array has exactly the size the layout expects
    int bad = -1;
    walk_batch(c, [&](int id, int64_t, int64_t nn) {
        if (id == 19) return;
        if ((int64_t)pb.arr[id].size() != nn) bad = id;
    });
    if (bad >= 0) return set_err(DVAE_ERR_SIZE, "pack_batch: internal size mismatch in array %d", bad);
    return DVAE_OK;
}

}  // namespace dvae

using namespace dvae;

extern "C" int dvae_batch_create(int32_t B, int32_t max_n, int32_t nvt, int32_t start_type, int32_t is_bn,
                                 const int32_t* types, const uint64_t* pred_mask, const int32_t* n_vertices,
                                 void** handle, dvae_batch_counts* counts) {
    DVAE_REQUIRE(types && pred_mask && n_vertices && handle && counts, "dvae_batch_create: null pointer");
    PackedBatch* pb = new (std::nothrow) PackedBatch();
    DVAE_REQUIRE(pb != nullptr, "dvae_batch_create: out of memory");
    // pairs per chunk of the grouped backward (loss_kernels.hip): rows with more pairs are split into chunks that combine
    // with atomics.  Large graphs (a vertex has up to max_n - 1 candidate predecessors, so a table row up to that many
    // pairs) take whole rows: cfg 5's dEA pass otherwise issues 1.3 G float atomics per step.
    const int chunk = max_n > 16 ? 64 : 16;
    int rc = build(*pb, B, max_n, nvt, start_type, is_bn, types, pred_mask, n_vertices, chunk);
    if (rc != 0) { delete pb; return rc; }
    *handle = pb;
    *counts = pb->c;
    return DVAE_OK;
}

extern "C" int dvae_batch_export(void* handle, int32_t* out, int64_t out_ints) {
    DVAE_REQUIRE(handle && out, "dvae_batch_export: null pointer");
    PackedBatch* pb = (PackedBatch*)handle;
    DVAE_REQUIRE(out_ints >= pb->c.total_ints, "dvae_batch_export: buffer too small");
    memset(out, 0, sizeof(int32_t) * (size_t)pb->c.total_ints);
    walk_batch(pb->c, [&](int id, int64_t off, int64_t nn) {
        if (id == 19) memcpy(out + off, pb->pair_t.data(), pb->pair_t.size());
        else if (nn) memcpy(out + off, pb->arr[id].data(), sizeof(int32_t) * (size_t)nn);
    });
    return DVAE_OK;
}

// Shape-stable export (include/dvae_hip.h: dvae_batch_export_padded).  The counts that depend on the batch's EDGES --
// E, PT, PL, NB, CH -- are raised to the caller's caps, so that every batch of the same (B, max_n, nvt, kind, caps) has
// the SAME counts and therefore the same blob layout, workspace carve-up, launch grids and kernel arguments: a train
// step on such batches is one fixed launch sequence (capturable once, replayable for ever).  The padding is inert:
//   pred / succ / partial-list index arrays   longer, the tail is never addressed (the offset arrays bound every read);
//   partial rows PT .. PT_cap-1               empty predecessor lists, type 0: computed, referenced by no pair, and their
//                                             gradient rows stay exactly zero;
//   EB table rows NB .. NB_cap-1              gathered from the first type row: computed, referenced by no pair; the
//                                             all-zero row moves from NB to NB_cap (pair_h entries are renumbered);
//   backward chunks CH .. CH_cap-1            EMPTY chunks, one per padding row of the EB table (its gradient row is
//                                             written whole, with zeros) and the rest on the all-zero row.
// Only batches whose graphs all have max_n vertices qualify (everything else in the counts is then a function of
// (B, max_n)); anything above a cap is an error the caller answers with the plain export.
extern "C" int dvae_batch_export_padded(void* handle, const dvae_batch_caps* caps, int32_t* out, int64_t out_ints,
                                        dvae_batch_counts* counts_out) {
    DVAE_REQUIRE(handle && caps && out && counts_out, "dvae_batch_export_padded: null pointer");
    PackedBatch* pb = (PackedBatch*)handle;
    const dvae_batch_counts& c = pb->c;
    DVAE_REQUIRE(c.R == (c.n - 1) * c.B && c.NA == (c.n - 1) * c.B && c.HGF == (c.is_bn ? (c.n - 2) * c.B : c.B),
                 "dvae_batch_export_padded: every graph of the batch must have max_n vertices");
    DVAE_REQUIRE(c.E <= caps->E && c.PT <= caps->PT && c.PL <= caps->PL && c.NB <= caps->NB,
                 "dvae_batch_export_padded: the batch exceeds the caps (E %d/%d, PT %d/%d, PL %d/%d, NB %d/%d)", c.E, caps->E,
                 c.PT, caps->PT, c.PL, caps->PL, c.NB, caps->NB);
    const int pad_rows = caps->NB - c.NB;
    DVAE_REQUIRE(c.CH + pad_rows <= caps->CH, "dvae_batch_export_padded: the batch exceeds the caps (CH %d + %d padding rows / %d)",
                 c.CH, pad_rows, caps->CH);
    dvae_batch_counts q = c;
    q.E = caps->E; q.PT = caps->PT; q.PL = caps->PL; q.NB = caps->NB; q.CH = caps->CH;
    q.total_ints = walk_batch(q, [](int, int64_t, int64_t) {});
    DVAE_REQUIRE(out_ints >= q.total_ints, "dvae_batch_export_padded: buffer too small (%lld < %lld)", (long long)out_ints,
                 (long long)q.total_ints);
    memset(out, 0, sizeof(int32_t) * (size_t)q.total_ints);
    walk_batch(q, [&](int id, int64_t off, int64_t) {
        int32_t* dst = out + off;
        if (id == 19) { memcpy(dst, pb->pair_t.data(), pb->pair_t.size()); return; }
        const std::vector<int32_t>& a = pb->arr[id];
        if (!a.empty()) memcpy(dst, a.data(), sizeof(int32_t) * a.size());
        switch (id) {
        case 9:  // p_off [PT_cap + 1]: the padding rows' lists are empty
            for (int r = c.PT + 1; r <= q.PT; ++r) dst[r] = c.PL;
            break;
        case 15:  // rowsB [NB_cap]: the padding rows read the first type row of the state table
            for (int r = c.NB; r < q.NB; ++r) dst[r] = c.nB;
            break;
        case 17:  // pair_h: the all-zero row is row NB_cap now
            for (int k = 0; k < c.NP; ++k)
                if (dst[k] == c.NB) dst[k] = q.NB;
            break;
        case 23:  // ch_off [CH_cap + 1]: empty chunks behind the real ones
            for (int k = c.CH + 1; k <= q.CH; ++k) dst[k] = c.NP;
            break;
        case 24:  // ch_row [CH_cap]: one empty chunk per padding row (written whole: zeros), the rest on the all-zero row
            for (int k = 0; k < c.CH; ++k)
                if ((dst[k] & 0x7fffffff) == c.NB) dst[k] = (int32_t)((uint32_t)q.NB | ((uint32_t)dst[k] & 0x80000000u));
            for (int k = c.CH; k < q.CH; ++k) dst[k] = k - c.CH < pad_rows ? c.NB + (k - c.CH) : q.NB;
            break;
        default: break;
        }
    });
    *counts_out = q;
    return DVAE_OK;
}

extern "C" void dvae_batch_destroy(void* handle) { delete (PackedBatch*)handle; }
