// Host side of the block path: the tables of modest_pp_score_block[_mixed] for a batch of scans of a frame store.
//
// The reference stacks the history frames of every scan anew (pre_compute_pp_score.py:132-150) from the scan's index list
// (valid_idx_info.pkl, data_preprocessing/lyft/split_traintest.py:79-101: a traversal's frames within a distance window; a
// frame may be listed twice).  A block call needs the UNION of the scans' lists as (frame, occurrence) entries, in an order in
// which every scan's entries lie close together, every scan's members as indices into it, and the checks under which the
// world lattice is a valid filter for all of them.  This was ~45 numpy calls per block in FrameStore.block_tables (0.9 ms
// for 7 scans, 3.7 ms for 32 x 360 members: a third of the host time of a PP call in the driver's 20-step window); here it
// is one pass over the members.  Pure host code: no context, no stream, no device.
#include "../../include/modest_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

struct Entry {
    uint64_t key;   // slot * K + occurrence
    int32_t member, scan;
};

// rows x, y of A = W_live * inv(R), R = the live scan's float32 relative pose (rows 0..2; last row 0 0 0 1)
bool scan_to_lattice(const double *W16, const float *rel12, double A2[8]) {
    double R[3][4];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) R[r][c] = (double)rel12[r * 4 + c];
    const double c00 = R[1][1] * R[2][2] - R[1][2] * R[2][1], c01 = R[1][2] * R[2][0] - R[1][0] * R[2][2],
                 c02 = R[1][0] * R[2][1] - R[1][1] * R[2][0];
    const double det = R[0][0] * c00 + R[0][1] * c01 + R[0][2] * c02;
    if (!(std::fabs(det) > 0.0) || !std::isfinite(det)) return false;
    double I[3][4];   // inv(R) rows 0..2
    I[0][0] = c00 / det, I[0][1] = (R[0][2] * R[2][1] - R[0][1] * R[2][2]) / det, I[0][2] = (R[0][1] * R[1][2] - R[0][2] * R[1][1]) / det;
    I[1][0] = c01 / det, I[1][1] = (R[0][0] * R[2][2] - R[0][2] * R[2][0]) / det, I[1][2] = (R[0][2] * R[1][0] - R[0][0] * R[1][2]) / det;
    I[2][0] = c02 / det, I[2][1] = (R[0][1] * R[2][0] - R[0][0] * R[2][1]) / det, I[2][2] = (R[0][0] * R[1][1] - R[0][1] * R[1][0]) / det;
    for (int r = 0; r < 3; ++r) I[r][3] = -(I[r][0] * R[0][3] + I[r][1] * R[1][3] + I[r][2] * R[2][3]);
    for (int r = 0; r < 2; ++r) {
        const double *w = W16 + 4 * r;
        for (int c = 0; c < 4; ++c) A2[r * 4 + c] = w[0] * I[0][c] + w[1] * I[1][c] + w[2] * I[2][c] + (c == 3 ? w[3] : 0.0);
    }
    return true;
}

}   // namespace

extern "C" int modest_pp_block_tables(const modest_pp_store_view *st, int n_scans, const modest_pp_frame *const *live_host,
                                      const modest_pp_frame *const *members_host, const int64_t *const *slots_host,
                                      const int32_t *n_members, const int32_t *n_trav_scan, int apply_rule,
                                      modest_pp_block_frame *frames_out, int32_t frames_cap, int32_t *n_frames_out,
                                      modest_pp_block_scan *scans_out, int32_t *member_slot_out, int32_t *member_trav_out,
                                      float *member_rel_out) {
    if (st == nullptr || n_scans < 1 || live_host == nullptr || members_host == nullptr || slots_host == nullptr || n_members == nullptr ||
        n_trav_scan == nullptr || frames_out == nullptr || n_frames_out == nullptr || scans_out == nullptr || member_slot_out == nullptr ||
        member_trav_out == nullptr || member_rel_out == nullptr)
        return -1;
    const int B = n_scans;
    int64_t members = 0, sumT = 0;
    for (int s = 0; s < B; ++s) {
        if (n_members[s] < 0) return -1;
        members += n_members[s];
        sumT += n_trav_scan[s];
    }
    *n_frames_out = 0;
    if (members == 0) return MODEST_BLOCK_TABLES_CHAIN;
    // (frame, occurrence): the k-th listing of a frame inside a scan is union entry (frame, k)
    std::vector<Entry> ent((size_t)members);
    std::vector<int32_t> seen_scan((size_t)st->n_slots, -1), seen_cnt((size_t)st->n_slots, 0);
    std::vector<int32_t> occ((size_t)members);
    int32_t K = 1;
    {
        int64_t m = 0;
        for (int s = 0; s < B; ++s)
            for (int k = 0; k < n_members[s]; ++k, ++m) {
                const int64_t slot = slots_host[s][k];
                if (slot < 0 || slot >= st->n_slots) return -1;
                if (seen_scan[(size_t)slot] != s) seen_scan[(size_t)slot] = s, seen_cnt[(size_t)slot] = 0;
                occ[(size_t)m] = seen_cnt[(size_t)slot]++;
                K = std::max(K, occ[(size_t)m] + 1);
            }
        m = 0;
        for (int s = 0; s < B; ++s)
            for (int k = 0; k < n_members[s]; ++k, ++m)
                ent[(size_t)m] = Entry{(uint64_t)slots_host[s][k] * (uint64_t)K + (uint64_t)occ[(size_t)m], (int32_t)m, s};
    }
    struct Uni {
        uint64_t key;
        int32_t first, last, run0, run1;   // users: scans first..last; run0..run1: its members in `ent` (sorted path only)
    };
    std::vector<Uni> un;
    std::vector<int32_t> uni_of;   // key -> index into `un` (direct path)
    const uint64_t n_keys = (uint64_t)st->n_slots * (uint64_t)K;
    // (a key-indexed table costs its own clearing: stores of more than 16 M (slot, occurrence) keys sort the members instead;
    // MODEST_BLOCK_TABLES_DIRECT_MAX moves the limit -- the tests run both paths)
    uint64_t direct_max = 1ull << 24;
    if (const char *e = getenv("MODEST_BLOCK_TABLES_DIRECT_MAX")) direct_max = strtoull(e, nullptr, 10);
    const bool direct = n_keys <= direct_max;
    if (direct) {   // one pass, no sort of the members: the store has a few thousand slots
        uni_of.assign((size_t)n_keys, -1);
        un.reserve(4096);
        for (const Entry &e : ent) {
            int32_t &u = uni_of[(size_t)e.key];
            if (u < 0) {
                u = (int32_t)un.size();
                un.push_back(Uni{e.key, e.scan, e.scan, 0, 0});
            } else {
                un[(size_t)u].last = e.scan;   // (members come scan after scan)
            }
        }
        std::sort(un.begin(), un.end(), [](const Uni &a, const Uni &b) { return a.key < b.key; });
        for (size_t i = 0; i < un.size(); ++i) uni_of[(size_t)un[i].key] = (int32_t)i;
    } else {
        std::stable_sort(ent.begin(), ent.end(), [](const Entry &a, const Entry &b) { return a.key < b.key; });   // (scans ascend inside a key)
        un.reserve((size_t)members);
        for (int64_t i = 0; i < members;) {
            int64_t j = i;
            while (j < members && ent[(size_t)j].key == ent[(size_t)i].key) ++j;
            un.push_back(Uni{ent[(size_t)i].key, ent[(size_t)i].scan, ent[(size_t)j - 1].scan, (int32_t)i, (int32_t)j});
            i = j;
        }
    }
    const int64_t U = (int64_t)un.size();
    if (apply_rule) {   // (the measurements behind these numbers: FrameStore.block_tables)
        const double per_scan = (double)members / (double)B;
        if (B < 4 || members < 12 * sumT) return MODEST_BLOCK_TABLES_CHAIN;
        if ((double)U > 4.0 * per_scan) return B >= 8 ? MODEST_BLOCK_TABLES_SPLIT : MODEST_BLOCK_TABLES_CHAIN;
        if (U > 2048 && B >= 8) return MODEST_BLOCK_TABLES_SPLIT;
    }
    if (U >= (1 << 16) || U > frames_cap) return MODEST_BLOCK_TABLES_CHAIN;
    for (int s = 0; s < B; ++s) {
        const int64_t ls = slots_host[s][n_members[s]];
        if (ls < 0 || ls >= st->n_slots) return -1;
        if (!st->clean[ls]) return MODEST_BLOCK_TABLES_CHAIN;
    }
    for (const Uni &u : un)
        if (!st->clean[u.key / (uint64_t)K]) return MODEST_BLOCK_TABLES_CHAIN;
    // the lattice is a conservative filter only while two points within r of each other (pose error < 1e-4 m checked below,
    // float32 evaluation < 4e-5 m, per point) stay within one cell
    if (st->cell - st->radius < 2.0 * (1e-4 + 4e-5)) return MODEST_BLOCK_TABLES_CHAIN;
    {
        int32_t x0 = INT32_MAX, x1 = INT32_MIN, y0 = INT32_MAX, y1 = INT32_MIN;
        for (int s = 0; s < B; ++s) {
            const modest_pp_frame &r = st->records[slots_host[s][n_members[s]]];
            x0 = std::min(x0, r.TX0), x1 = std::max(x1, r.TX0), y0 = std::min(y0, r.TY0), y1 = std::max(y1, r.TY0);
        }
        if ((int64_t)x1 - x0 > st->window_span || (int64_t)y1 - y0 > st->window_span)
            return (apply_rule && B >= 8) ? MODEST_BLOCK_TABLES_SPLIT : MODEST_BLOCK_TABLES_CHAIN;
    }
    int32_t flags = 0;
    {
        bool have = false;
        for (int s = 0; s < B; ++s)
            for (int k = 0; k < n_members[s]; ++k) {
                const int32_t f = members_host[s][k].flags;
                if (!have) flags = f, have = true;
                else if (f != flags) return MODEST_BLOCK_TABLES_CHAIN;
            }
    }
    // every pose against the lattice: rows x, y of A_scan * rel_f - W_f, A_scan = W_live * inv(rel_live)
    {
        int64_t m = 0;
        for (int s = 0; s < B; ++s) {
            if (n_members[s] == 0) continue;
            double A2[8];
            if (!scan_to_lattice(st->W + 16 * slots_host[s][n_members[s]], live_host[s]->rel, A2)) return MODEST_BLOCK_TABLES_CHAIN;
            for (int k = 0; k < n_members[s]; ++k, ++m) {
                const float *rel = members_host[s][k].rel;
                const double *Wf = st->W + 16 * slots_host[s][k];
                for (int r = 0; r < 2; ++r) {
                    double lin = 0.0, off = 0.0;
                    for (int c = 0; c < 4; ++c) {
                        double d = A2[r * 4 + 0] * (double)rel[c] + A2[r * 4 + 1] * (double)rel[4 + c] + A2[r * 4 + 2] * (double)rel[8 + c];
                        if (c == 3) d += A2[r * 4 + 3];
                        d -= Wf[r * 4 + c];
                        if (c < 3) lin += std::fabs(d);
                        else off = std::fabs(d);
                    }
                    const double dev = lin * 160.0 + off;
                    if (!(std::isfinite(dev) && dev < 1e-4)) return MODEST_BLOCK_TABLES_CHAIN;
                }
            }
        }
    }
    // the union in the order (first + last, first) of the scans that use an entry -- the middle of the interval of its users --, ties in
    // (slot, occurrence) order: every scan's entries are then (close to) one range of the table
    std::vector<int32_t> order((size_t)U);
    for (int64_t i = 0; i < U; ++i) order[(size_t)i] = (int32_t)i;
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
        const int32_t sa = un[(size_t)a].first + un[(size_t)a].last, sb = un[(size_t)b].first + un[(size_t)b].last;
        return sa != sb ? sa < sb : un[(size_t)a].first < un[(size_t)b].first;
    });
    for (int64_t p = 0; p < U; ++p) {
        const Uni &u = un[(size_t)order[(size_t)p]];
        const int64_t slot = (int64_t)(u.key / (uint64_t)K);
        const modest_pp_frame &r = st->records[slot];
        modest_pp_block_frame &f = frames_out[p];
        f.xyz_dev = r.xyz_dev, f.tab_dev = r.tab_dev, f.n = r.n, f.TX0 = r.TX0, f.TY0 = r.TY0, f.flags = flags;
        std::memcpy(f.lat, st->lat + 8 * slot, 8 * sizeof(double));
        if (direct) uni_of[(size_t)u.key] = (int32_t)p;   // (key -> position in the table from here on)
        else
            for (int32_t i = u.run0; i < u.run1; ++i) member_slot_out[ent[(size_t)i].member] = (int32_t)p;
    }
    if (direct)
        for (const Entry &e : ent) member_slot_out[e.member] = uni_of[(size_t)e.key];
    int64_t m = 0;
    for (int s = 0; s < B; ++s) {
        const int64_t ls = slots_host[s][n_members[s]];
        const modest_pp_frame &r = st->records[ls];
        modest_pp_block_scan &sc = scans_out[s];
        std::memset(&sc, 0, sizeof(sc));
        sc.xyz_dev = r.xyz_dev, sc.tab_dev = r.tab_dev, sc.n = r.n, sc.TX0 = r.TX0, sc.TY0 = r.TY0;
        sc.perm_dev = reinterpret_cast<const uint32_t *>((uintptr_t)st->perm_dev[ls]);
        sc.n_members = n_members[s];
        std::memcpy(sc.lat, st->lat + 8 * ls, 8 * sizeof(double));
        std::memcpy(sc.rel, live_host[s]->rel, 12 * sizeof(float));
        sc.member_slot = member_slot_out + m, sc.member_trav = member_trav_out + m, sc.member_rel = member_rel_out + 12 * m;
        for (int k = 0; k < n_members[s]; ++k, ++m) {
            member_trav_out[m] = members_host[s][k].trav;
            std::memcpy(member_rel_out + 12 * m, members_host[s][k].rel, 12 * sizeof(float));
        }
    }
    *n_frames_out = (int32_t)U;
    return MODEST_BLOCK_TABLES_BLOCK;
}
