// Internal view of a packed DAG batch: device pointers carved out of the int32
// blob written by dvae_batch_export() (layout defined once, in bind_batch()).
#pragma once
#include <stdint.h>
#include "dvae_hip.h"

namespace dvae {

struct BatchView {
    const int32_t* types;                         // [nB] slot-major (slot = v*B + b)
    const int32_t *pred_off, *pred_idx;           // CSR over slots; idx = slot (DVAE) or type (BN)
    const int32_t *succ_off, *succ_idx;
    const int32_t *hgf_off, *hgf_idx, *hgb_off, *hgb_idx;  // graph-state source slots per graph
    const int32_t *p_off, *p_idx, *p_type;        // partial-state rows [PT]
    const int32_t *pt_off, *pt_idx;               // transposed: slot -> partial rows (DVAE)
    const int32_t *rowsA, *rowsB;                 // ST rows feeding the EA / EB tables
    const int32_t *pair_a, *pair_h, *pair_c;      // per pair: EA row, EB row, graph
    const uint8_t* pair_t;                        // per pair: BCE target
    const int32_t *ca_off, *ca_row, *ca_pairs;    // backward chunks grouped by EA row
    const int32_t *ch_off, *ch_row, *ch_pairs;    // ... by EB row
    const int32_t *cc_off, *cc_row;               // ... by graph (pairs contiguous)
    const int32_t *vs_off, *vs_idx;               // vertex-type rows: source slots
    const int32_t *vst_off, *vst_idx;             // transposed: slot -> vertex-type rows
    const int32_t *ce_off, *ce_tgt;               // CE targets per vertex-type row
};

static inline int64_t al4(int64_t x) { return (x + 3) & ~(int64_t)3; }

// Walks the canonical array order; returns total size in int32 units.
template <class F>
static inline int64_t walk_batch(const dvae_batch_counts& c, F&& f) {
    int64_t o = 0;
    auto take = [&](int id, int64_t n) { f(id, o, n); o += al4(n); };
    take(0, c.nB);
    take(1, c.nB + 1); take(2, c.E);
    take(3, c.nB + 1); take(4, c.E);
    take(5, c.B + 1); take(6, c.HGF);
    take(7, c.B + 1); take(8, c.HGB);
    take(9, c.PT + 1); take(10, c.PL); take(11, c.PT);
    take(12, c.nB + 1); take(13, c.PL);
    take(14, c.NA); take(15, c.NB);
    take(16, c.NP); take(17, c.NP); take(18, c.NP);
    take(19, (c.NP + 3) / 4);
    take(20, c.CA + 1); take(21, c.CA); take(22, c.NP);
    take(23, c.CH + 1); take(24, c.CH); take(25, c.NP);
    take(26, c.CC + 1); take(27, c.CC);
    take(28, c.R + 1); take(29, c.RL);
    take(30, c.nB + 1); take(31, c.RL);
    take(32, c.R + 1); take(33, c.NCE);
    return o;
}

static inline BatchView bind_batch(const dvae_batch_counts& c, const int32_t* base) {
    const int32_t* p[34];
    walk_batch(c, [&](int id, int64_t off, int64_t) { p[id] = base + off; });
    BatchView v;
    v.types = p[0];
    v.pred_off = p[1]; v.pred_idx = p[2];
    v.succ_off = p[3]; v.succ_idx = p[4];
    v.hgf_off = p[5]; v.hgf_idx = p[6]; v.hgb_off = p[7]; v.hgb_idx = p[8];
    v.p_off = p[9]; v.p_idx = p[10]; v.p_type = p[11];
    v.pt_off = p[12]; v.pt_idx = p[13];
    v.rowsA = p[14]; v.rowsB = p[15];
    v.pair_a = p[16]; v.pair_h = p[17]; v.pair_c = p[18];
    v.pair_t = reinterpret_cast<const uint8_t*>(p[19]);
    v.ca_off = p[20]; v.ca_row = p[21]; v.ca_pairs = p[22];
    v.ch_off = p[23]; v.ch_row = p[24]; v.ch_pairs = p[25];
    v.cc_off = p[26]; v.cc_row = p[27];
    v.vs_off = p[28]; v.vs_idx = p[29];
    v.vst_off = p[30]; v.vst_idx = p[31];
    v.ce_off = p[32]; v.ce_tgt = p[33];
    return v;
}

}  // namespace dvae
