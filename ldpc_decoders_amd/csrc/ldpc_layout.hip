// LDS layout planner of the fused backend -- host code only (see ldpc_layout.hpp for the problem statement).
#include "ldpc_layout.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <numeric>
#include <cstdlib>
#include <cstdio>

namespace ldpc {

namespace {

struct Rng {  // xorshift64*
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull) {}
    uint64_t next() {
        s ^= s >> 12;
        s ^= s << 25;
        s ^= s >> 27;
        return s * 0x2545F4914F6CDD1Dull;
    }
    int below(int n) { return (int)((next() >> 33) % (uint64_t)n); }
    double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};

inline int slot_of(int group, int bank) { return (group >> 1) * 64 + (group & 1) * 32 + bank; }
inline int group_of_slot(int s) { return (s >> 6) * 2 + ((s >> 5) & 1); }
inline int bank_of_slot(int s) { return s & 31; }

}  // namespace

void identity_layout(const Code& c, int DC, const VarRounds& vr, FusedLayout* L) {
    L->chk_slot.resize(c.m);
    L->var_slot.resize(c.n);
    std::iota(L->chk_slot.begin(), L->chk_slot.end(), 0);
    // variables: those that need a wide round first into the wide slots, the others into the narrow slots in index order
    // (spilling into left-over wide slots when the narrow ones are full); reserved rounds stay empty.  With pair rounds: three classes --
    // more than DV edges, three..DV edges, at most two -- each into its own rounds first, left-overs into the wider ones
    std::vector<int> free_of[3];  // usable slots of the wide / DV / pair rounds
    for (int q = 0; q < vr.VR; ++q) {
        for (int l = 0; l < 64; ++l)
            if (vr.usable_slot(q * 64 + l)) free_of[vr.width(q) > vr.DV ? 0 : (vr.width(q) == vr.DV ? 1 : 2)].push_back(q * 64 + l);
    }
    size_t next[3] = {0, 0, 0};
    auto take = [&](int cls) {  // a slot of class `cls`, or of the next wider class that still has one
        for (int k = cls; k >= 0; --k)
            if (next[k] < free_of[k].size()) return free_of[k][next[k]++];
        return -1;
    };
    auto class_of = [&](int v) { const int d = c.col_ptr[v + 1] - c.col_ptr[v]; return d > vr.DV ? 0 : ((vr.vr2 == 0 || d > 2) ? 1 : 2); };
    for (int cls = 0; cls < 3; ++cls)
        for (int v = 0; v < c.n; ++v)
            if (class_of(v) == cls) L->var_slot[v] = take(cls);
    L->edge_pos.assign(c.E, 0);
    L->var_pos.assign(c.E, 0);
    for (int cc = 0; cc < c.m; ++cc)
        for (int k = c.row_ptr[cc]; k < c.row_ptr[cc + 1]; ++k) L->edge_pos[k] = k - c.row_ptr[cc];
    for (int v = 0; v < c.n; ++v)
        for (int p = c.col_ptr[v]; p < c.col_ptr[v + 1]; ++p) L->var_pos[c.col_edge[p]] = p - c.col_ptr[v];
}

double layout_extra_cycles(const Code& c, int DC, int CR, const VarRounds& vr, const FusedLayout& L) {
    // address seen by lane `lane` of gather instruction (round, position); -1 == padded lane (free to broadcast)
    double extra = 0;
    auto group_cost = [](const int* addr32) {
        int mx = 1;
        for (int b = 0; b < 32; ++b) {
            int distinct = 0, seen[32];
            for (int l = 0; l < 32; ++l) {
                if (addr32[l] < 0 || (addr32[l] & 31) != b) continue;
                bool dup = false;
                for (int i = 0; i < distinct; ++i) dup |= seen[i] == addr32[l];
                if (!dup) seen[distinct++] = addr32[l];
            }
            mx = std::max(mx, distinct);
        }
        return mx - 1;
    };
    std::vector<int> cn((size_t)CR * DC * 64, -1), vn((size_t)vr.total_gathers() * 64, -1);
    for (int cc = 0; cc < c.m; ++cc)
        for (int k = c.row_ptr[cc]; k < c.row_ptr[cc + 1]; ++k) {
            const int s = L.chk_slot[cc];
            cn[(size_t)((s / 64) * DC + L.edge_pos[k]) * 64 + s % 64] = L.var_slot[c.edge_var[k]];
        }
    for (int k = 0; k < c.E; ++k) {
        const int vs = L.var_slot[c.edge_var[k]], cs = L.chk_slot[c.edge_chk[k]];
        vn[(size_t)(vr.first_gather(vs / 64) + L.var_pos[k]) * 64 + vs % 64] = ((cs / 64) * DC + L.edge_pos[k]) * 64 + cs % 64;
    }
    for (size_t i = 0; i < cn.size(); i += 32) extra += group_cost(&cn[i]);
    for (size_t i = 0; i < vn.size(); i += 32) extra += group_cost(&vn[i]);
    return extra;
}

namespace {

// Annealing state of one replica: slot assignments + the incremental surrogate cost.  Copyable (replicas adopt the best state
// of a generation), every move is an involution.
struct Anneal {
    const Code& c;
    const VarRounds& vr;
    int DC, DVM, NGC, NGV, m, n;
    std::vector<int> cgrp, cbank, vgrp, vbank, vflip, chk_at, var_at, cnA, cnB;
    const std::vector<int>* edge_vj;  // canonical index of edge k in its variable's list (shared, read-only)
    long cost = 0;
    // check-phase cells: (check group, variable bank) with capacity DC when the edge positions are free (edge colouring places them
    // afterwards), (check group, position, variable bank) with capacity 1 when every check keeps its canonical edge order
    bool fixed = false;
    int capA = 0;

    Anneal(const Code& c_, int DC_, int CR, const VarRounds& vr_, const FusedLayout& L, const std::vector<int>* evj)
        : c(c_), vr(vr_), DC(DC_), DVM(std::max(vr_.DV, vr_.dvx)), NGC(2 * CR), NGV(2 * vr_.VR), m(c_.m), n(c_.n), edge_vj(evj) {
        fixed = vr_.fixed_edge_order;
        capA = fixed ? 1 : DC;
        cgrp.resize(m); cbank.resize(m); vgrp.resize(n); vbank.resize(n); vflip.assign(n, 0);
        chk_at.assign((size_t)NGC * 32, -1); var_at.assign((size_t)NGV * 32, -1);
        for (int cc = 0; cc < m; ++cc) {
            cgrp[cc] = group_of_slot(L.chk_slot[cc]);
            cbank[cc] = bank_of_slot(L.chk_slot[cc]);
            chk_at[(size_t)cgrp[cc] * 32 + cbank[cc]] = cc;
        }
        for (int v = 0; v < n; ++v) {
            vgrp[v] = group_of_slot(L.var_slot[v]);
            vbank[v] = bank_of_slot(L.var_slot[v]);
            var_at[(size_t)vgrp[v] * 32 + vbank[v]] = v;
        }
        cnA.assign((size_t)NGC * (fixed ? DC : 1) * 32, 0);
        cnB.assign((size_t)NGV * DVM * 32, 0);
        for (int v = 0; v < n; ++v) touch_var(v, +1);
    }
    Anneal& operator=(const Anneal& o) {  // same problem instance: copy the mutable state only
        cgrp = o.cgrp; cbank = o.cbank; vgrp = o.vgrp; vbank = o.vbank; vflip = o.vflip; chk_at = o.chk_at; var_at = o.var_at;
        cnA = o.cnA; cnB = o.cnB; cost = o.cost;
        return *this;
    }
    // placement constraint: a variable with more than DV edges only fits a slot of a wide round (group g = 2*round + half)
    bool fits(int v, int s) const { return v < 0 || (vr.usable_slot(slot_of(s / 32, s % 32)) && c.col_ptr[v + 1] - c.col_ptr[v] <= vr.width((s / 32) / 2)); }
    int vpos(int v, int j) const { return (j < 2 && vflip[v] && (c.col_ptr[v + 1] - c.col_ptr[v]) >= 2) ? 1 - j : j; }
    size_t cellA(int k, int bv) const {  // cell of edge k (row-major edge index) when its variable sits on bank bv
        const int cc = c.edge_chk[k];
        return fixed ? ((size_t)cgrp[cc] * DC + (k - c.row_ptr[cc])) * 32 + bv : (size_t)cgrp[cc] * 32 + bv;
    }
    void addA(int k, int bv, int d) {
        int& x = cnA[cellA(k, bv)];
        cost -= std::max(0, x - capA);
        x += d;
        cost += std::max(0, x - capA);
    }
    void addB(int gv, int pos, int bc, int d) {
        int& x = cnB[((size_t)gv * DVM + pos) * 32 + bc];
        cost -= std::max(0, x - 1);
        x += d;
        cost += std::max(0, x - 1);
    }
    void touch_var(int v, int d) {
        for (int p = c.col_ptr[v]; p < c.col_ptr[v + 1]; ++p) {
            const int k = c.col_edge[p], cc = c.edge_chk[k];
            addA(k, vbank[v], d);
            addB(vgrp[v], vpos(v, p - c.col_ptr[v]), cbank[cc], d);
        }
    }
    void touch_chk(int cc, int d) {
        for (int k = c.row_ptr[cc]; k < c.row_ptr[cc + 1]; ++k) {
            const int v = c.edge_var[k];
            addA(k, vbank[v], d);
            addB(vgrp[v], vpos(v, (*edge_vj)[k]), cbank[cc], d);
        }
    }
    // does this variable / check sit on an overloaded cell?  (directed move selection)
    bool hot_var(int v) const {
        for (int p = c.col_ptr[v]; p < c.col_ptr[v + 1]; ++p) {
            const int k = c.col_edge[p], cc = c.edge_chk[k];
            if (cnA[cellA(k, vbank[v])] > capA) return true;
            if (cnB[((size_t)vgrp[v] * DVM + vpos(v, p - c.col_ptr[v])) * 32 + cbank[cc]] > 1) return true;
        }
        return false;
    }
    bool hot_chk(int cc) const {
        for (int k = c.row_ptr[cc]; k < c.row_ptr[cc + 1]; ++k) {
            const int v = c.edge_var[k];
            if (cnA[cellA(k, vbank[v])] > capA) return true;
            if (cnB[((size_t)vgrp[v] * DVM + vpos(v, (*edge_vj)[k])) * 32 + cbank[cc]] > 1) return true;
        }
        return false;
    }
    void swap_vars(int s1, int s2) {
        const int a = var_at[s1], b = var_at[s2];
        if (a >= 0) touch_var(a, -1);
        if (b >= 0) touch_var(b, -1);
        if (a >= 0) { vgrp[a] = s2 / 32; vbank[a] = s2 % 32; }
        if (b >= 0) { vgrp[b] = s1 / 32; vbank[b] = s1 % 32; }
        std::swap(var_at[s1], var_at[s2]);
        if (a >= 0) touch_var(a, +1);
        if (b >= 0) touch_var(b, +1);
    }
    void swap_chks(int s1, int s2) {
        const int a = chk_at[s1], b = chk_at[s2];
        if (a >= 0) touch_chk(a, -1);
        if (b >= 0) touch_chk(b, -1);
        if (a >= 0) { cgrp[a] = s2 / 32; cbank[a] = s2 % 32; }
        if (b >= 0) { cgrp[b] = s1 / 32; cbank[b] = s1 % 32; }
        std::swap(chk_at[s1], chk_at[s2]);
        if (a >= 0) touch_chk(a, +1);
        if (b >= 0) touch_chk(b, +1);
    }
    void flip_var(int v) {
        touch_var(v, -1);
        vflip[v] ^= 1;
        touch_var(v, +1);
    }

    // moves [it0, it1) of a schedule of `total` moves (the temperature is a function of the move index only); keeps the state
    // with the lowest cost seen in `best` (which starts as a copy of the entry state)
    void run(Rng& rng, long it0, long it1, long total, Anneal* best) {
        const int nvs = NGV * 32, ncs = NGC * 32;
        const double T0 = 0.8, T_end = 0.08;
        double T = T0 * std::pow(T_end / T0, (double)it0 / (double)total);
        for (long it = it0; it < it1 && best->cost > 0; ++it) {
            if ((it & 4095) == 0) T = T0 * std::pow(T_end / T0, (double)it / (double)total);
            const long before = cost;
            const int kind = rng.below(100);
            int a = 0, b = 0;
            // directed selection: three moves in four start from an item that sits on an overloaded cell (a few tries to find one)
            const bool directed = (rng.next() >> 62) != 0;
            if (kind < 45) {
                a = rng.below(nvs);
                if (directed)
                    for (int t = 0; t < 6 && (var_at[a] < 0 || !hot_var(var_at[a])); ++t) a = rng.below(nvs);
                // partner: any slot, or -- the check-phase cost depends on a variable's BANK only, the variable-phase cost on its
                // GROUP only -- a slot of the same group (another bank) or of the same bank (another group): such a move repairs
                // one phase without disturbing the other
                const int shape = rng.below(8);
                b = shape < 3 ? (a / 32) * 32 + rng.below(32) : (shape < 6 ? rng.below(NGV) * 32 + a % 32 : rng.below(nvs));
                if (a == b || (var_at[a] < 0 && var_at[b] < 0) || !fits(var_at[a], b) || !fits(var_at[b], a)) continue;
                swap_vars(a, b);
            } else if (kind < 85) {
                a = rng.below(ncs);
                if (directed)
                    for (int t = 0; t < 6 && (chk_at[a] < 0 || !hot_chk(chk_at[a])); ++t) a = rng.below(ncs);
                const int shape = rng.below(8);  // likewise: a check's GROUP enters the check-phase cost only, its BANK the variable-phase cost only
                b = shape < 3 ? (a / 32) * 32 + rng.below(32) : (shape < 6 ? rng.below(NGC) * 32 + a % 32 : rng.below(ncs));
                if (a == b || (chk_at[a] < 0 && chk_at[b] < 0)) continue;
                swap_chks(a, b);
            } else {
                a = rng.below(n);
                if (directed)
                    for (int t = 0; t < 6 && !hot_var(a); ++t) a = rng.below(n);
                flip_var(a);
            }
            const long delta = cost - before;
            if (delta > 0 && rng.unit() >= std::exp(-(double)delta / T)) {  // reject: every move is an involution
                if (kind < 45) swap_vars(a, b);
                else if (kind < 85) swap_chks(a, b);
                else flip_var(a);
            } else if (cost < best->cost) {
                *best = *this;
            }
        }
    }
};

}  // namespace

void plan_fused_layout(const Code& c, int DC, int CR, const VarRounds& vr, uint64_t seed, long moves, FusedLayout* L) {
    identity_layout(c, DC, vr, L);
    L->base_cycles = 2.0 * (CR * DC + vr.total_gathers());
    L->extra_cycles_identity = layout_extra_cycles(c, DC, CR, vr, *L);
    const int m = c.m, n = c.n;
    const int NGC = 2 * CR;
    const int64_t E = c.E;
    std::vector<int> edge_vj(E);  // canonical index of edge k in its variable's list
    for (int v = 0; v < n; ++v)
        for (int p = c.col_ptr[v]; p < c.col_ptr[v + 1]; ++p) edge_vj[c.col_edge[p]] = p - c.col_ptr[v];

    // ---- annealing: one chain; the schedule is a function of the move count only, so a plan depends on (code, shape, seed, moves)
    // and not on the machine it was computed on.  (Independent or best-state-sharing replicas on several host threads were
    // measured: 8 x 2.5 M moves end where ONE chain of 2.5 M moves ends -- the result is set by the length of the chain.)
    const long max_moves = moves < 1 ? 1 : moves;
    Anneal chain(c, DC, CR, vr, *L, &edge_vj);
    Anneal champion = chain;
    Rng rng(seed);
    chain.run(rng, 0, max_moves, max_moves, &champion);
    const std::vector<int>&cgrp = champion.cgrp, &cbank = champion.cbank, &vgrp = champion.vgrp, &vbank = champion.vbank;
    auto vpos = [&](int v, int j) { return champion.vpos(v, j); };

    for (int cc = 0; cc < m; ++cc) L->chk_slot[cc] = slot_of(cgrp[cc], cbank[cc]);
    for (int v = 0; v < n; ++v) L->var_slot[v] = slot_of(vgrp[v], vbank[v]);
    for (int k = 0; k < E; ++k) L->var_pos[k] = vpos(c.edge_var[k], edge_vj[k]);

    // ---- positions inside each check: edge-colour every check group against the variable banks (unless the order is prescribed)
    std::vector<std::vector<int>> group_checks(NGC);
    if (vr.fixed_edge_order) group_checks.clear();
    if (!vr.fixed_edge_order)
        for (int cc = 0; cc < m; ++cc) group_checks[cgrp[cc]].push_back(cc);
    for (int g = 0; g < (int)group_checks.size(); ++g) {
        std::vector<int> edges;  // edge ids of this group
        for (int cc : group_checks[g])
            for (int k = c.row_ptr[cc]; k < c.row_ptr[cc + 1]; ++k) edges.push_back(k);
        // "proper" edges: at most DC per bank; the overflow is placed afterwards
        std::vector<int> bank_deg(32, 0);
        std::vector<int> proper, overflow;
        for (int k : edges) {
            const int b = vbank[c.edge_var[k]];
            if (bank_deg[b] < DC) { ++bank_deg[b]; proper.push_back(k); } else overflow.push_back(k);
        }
        std::vector<int> colour_of(edges.size(), -1);
        auto eidx = [&](int k) { return (int)(std::lower_bound(edges.begin(), edges.end(), k) - edges.begin()); };
        std::sort(edges.begin(), edges.end());
        std::vector<std::vector<int>> cAt(m > 0 ? group_checks[g].size() : 0, std::vector<int>(DC, -1));
        std::vector<std::vector<int>> bAt(32, std::vector<int>(DC, -1));
        auto cloc = [&](int cc) { return (int)(std::find(group_checks[g].begin(), group_checks[g].end(), cc) - group_checks[g].begin()); };
        auto set_col = [&](int k, int colr) {
            colour_of[eidx(k)] = colr;
            cAt[cloc(c.edge_chk[k])][colr] = k;
            bAt[vbank[c.edge_var[k]]][colr] = k;
        };
        for (int k : proper) {
            const int cl = cloc(c.edge_chk[k]), b = vbank[c.edge_var[k]];
            int a = -1, bf = -1;
            for (int x = 0; x < DC && a < 0; ++x) if (cAt[cl][x] < 0) a = x;
            for (int x = 0; x < DC && bf < 0; ++x) if (bAt[b][x] < 0) bf = x;
            if (a < 0 || bf < 0) { if (a >= 0) { colour_of[eidx(k)] = a; cAt[cl][a] = k; } continue; }
            if (bAt[b][a] < 0) { set_col(k, a); continue; }
            if (cAt[cl][bf] < 0) { set_col(k, bf); continue; }
            // alternating a/bf path from bank b; afterwards colour a is free at b (and still free at the check)
            std::vector<int> path;
            int cur = b;
            for (int guard = 0; guard < 4096; ++guard) {
                const int e1 = bAt[cur][a];
                if (e1 < 0) break;
                path.push_back(e1);
                const int e2 = cAt[cloc(c.edge_chk[e1])][bf];
                if (e2 < 0) break;
                path.push_back(e2);
                cur = vbank[c.edge_var[e2]];
            }
            for (int e : path) {
                const int old = colour_of[eidx(e)];
                cAt[cloc(c.edge_chk[e])][old] = -1;
                bAt[vbank[c.edge_var[e]]][old] = -1;
            }
            for (int e : path) set_col(e, colour_of[eidx(e)] == a ? bf : a);
            set_col(k, a);
        }
        for (int k : overflow) {  // leftover colours of the check; pick the one whose bank cell is least loaded
            const int cl = cloc(c.edge_chk[k]);
            int a = -1;
            for (int x = 0; x < DC && a < 0; ++x) if (cAt[cl][x] < 0) a = x;
            if (a >= 0) { colour_of[eidx(k)] = a; cAt[cl][a] = k; }
        }
        // any edge still uncoloured (should not happen): give it the first free colour of its check
        for (int cc : group_checks[g]) {
            std::vector<char> used(DC, 0);
            for (int k = c.row_ptr[cc]; k < c.row_ptr[cc + 1]; ++k)
                if (colour_of[eidx(k)] >= 0) used[colour_of[eidx(k)]] = 1;
            for (int k = c.row_ptr[cc]; k < c.row_ptr[cc + 1]; ++k)
                if (colour_of[eidx(k)] < 0)
                    for (int x = 0; x < DC; ++x)
                        if (!used[x]) { colour_of[eidx(k)] = x; used[x] = 1; break; }
        }
        for (int k : edges) L->edge_pos[k] = colour_of[eidx(k)];
    }
    L->extra_cycles_planned = layout_extra_cycles(c, DC, CR, vr, *L);
    if (L->extra_cycles_planned >= L->extra_cycles_identity) {  // never ship a layout worse than the trivial one
        const double id_cost = L->extra_cycles_identity;
        identity_layout(c, DC, vr, L);
        L->extra_cycles_planned = id_cost;
    }
}

// ---- plan store ---------------------------------------------------------------------------------------------------
namespace {
constexpr uint64_t kPlanMagic = 0x314e4c5043504c44ull;  // "DLPCPLN1"
constexpr uint32_t kPlannerVersion = 1;                 // bump when slot numbering / position semantics change
inline uint64_t fnv(uint64_t h, const void* p, size_t nbytes) {
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < nbytes; ++i) h = (h ^ b[i]) * 0x100000001b3ull;
    return h;
}
}  // namespace

uint64_t layout_key(const Code& c, int DC, int CR, const VarRounds& vr, int NW) {
    uint64_t h = 0xcbf29ce484222325ull;
    const int32_t hdr[10] = {(int32_t)kPlannerVersion, c.m, c.n, DC, CR, vr.VR, vr.DV, vr.vrx, vr.dvx, NW};
    h = fnv(h, hdr, sizeof(hdr));
    if (vr.reserved) h = fnv(h, &vr.reserved, sizeof(vr.reserved));
    if (vr.reserved && vr.reserved_half) h = fnv(h, "half-row", 8);
    if (vr.fixed_edge_order) h = fnv(h, "fixed-edge-order", 16);
    if (vr.vr2) h = fnv(h, &vr.vr2, sizeof(vr.vr2));
    h = fnv(h, c.edge_chk.data(), c.edge_chk.size() * sizeof(int32_t));
    h = fnv(h, c.edge_var.data(), c.edge_var.size() * sizeof(int32_t));
    return h;
}

bool layout_valid(const Code& c, int DC, int CR, const VarRounds& vr, const FusedLayout& L) {
    const size_t E = (size_t)c.E;
    if (L.chk_slot.size() != (size_t)c.m || L.var_slot.size() != (size_t)c.n || L.edge_pos.size() != E || L.var_pos.size() != E) return false;
    std::vector<char> seen_c((size_t)CR * 64, 0), seen_v((size_t)vr.VR * 64, 0);
    for (int s : L.chk_slot) {
        if (s < 0 || s >= CR * 64 || seen_c[s]++) return false;
    }
    for (int v = 0; v < c.n; ++v) {
        const int s = L.var_slot[v];
        if (s < 0 || s >= vr.VR * 64 || seen_v[s]++) return false;
        if (!vr.usable_slot(s) || c.col_ptr[v + 1] - c.col_ptr[v] > vr.width(s / 64)) return false;
    }
    for (int cc = 0; cc < c.m; ++cc) {  // positions inside a check: distinct, below DC
        unsigned mask = 0;
        for (int k = c.row_ptr[cc]; k < c.row_ptr[cc + 1]; ++k) {
            const int p = L.edge_pos[k];
            if (p < 0 || p >= DC || ((mask >> p) & 1u)) return false;
            if (vr.fixed_edge_order && p != k - c.row_ptr[cc]) return false;
            mask |= 1u << p;
        }
    }
    for (int v = 0; v < c.n; ++v) {  // positions inside a variable: a permutation of 0..deg-1 that only swaps the first two
        const int deg = c.col_ptr[v + 1] - c.col_ptr[v];
        for (int j = 0; j < deg; ++j) {
            const int p = L.var_pos[c.col_edge[c.col_ptr[v] + j]];
            const bool ok = (j >= 2) ? (p == j) : (deg >= 2 ? (p == j || p == 1 - j) : p == j);
            if (!ok) return false;
        }
        if (deg >= 2 && L.var_pos[c.col_edge[c.col_ptr[v]]] == L.var_pos[c.col_edge[c.col_ptr[v] + 1]]) return false;
    }
    return true;
}

bool layout_save(const std::string& path, uint64_t key, const Code& c, const FusedLayout& L) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return false;
    const uint64_t hdr[3] = {kPlanMagic, key, (uint64_t)c.E};
    const int32_t dims[2] = {c.m, c.n};
    const double info[3] = {L.base_cycles, L.extra_cycles_identity, L.extra_cycles_planned};
    bool ok = fwrite(hdr, sizeof(hdr), 1, f) == 1 && fwrite(dims, sizeof(dims), 1, f) == 1 && fwrite(info, sizeof(info), 1, f) == 1;
    auto put = [&](const std::vector<int>& v) { ok = ok && (v.empty() || fwrite(v.data(), sizeof(int), v.size(), f) == v.size()); };
    put(L.chk_slot); put(L.var_slot); put(L.edge_pos); put(L.var_pos);
    ok = (fclose(f) == 0) && ok;
    return ok;
}

bool layout_load(const std::string& path, uint64_t key, const Code& c, int DC, int CR, const VarRounds& vr, FusedLayout* L) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    uint64_t hdr[3];
    int32_t dims[2];
    double info[3];
    FusedLayout T;
    bool ok = fread(hdr, sizeof(hdr), 1, f) == 1 && fread(dims, sizeof(dims), 1, f) == 1 && fread(info, sizeof(info), 1, f) == 1;
    ok = ok && hdr[0] == kPlanMagic && hdr[1] == key && hdr[2] == (uint64_t)c.E && dims[0] == c.m && dims[1] == c.n;
    auto get = [&](std::vector<int>& v, size_t cnt) {
        if (!ok) return;
        v.resize(cnt);
        ok = cnt == 0 || fread(v.data(), sizeof(int), cnt, f) == cnt;
    };
    get(T.chk_slot, (size_t)c.m); get(T.var_slot, (size_t)c.n); get(T.edge_pos, (size_t)c.E); get(T.var_pos, (size_t)c.E);
    fclose(f);
    if (!ok || !layout_valid(c, DC, CR, vr, T)) return false;
    T.base_cycles = 2.0 * (CR * DC + vr.total_gathers());
    T.extra_cycles_identity = info[1];
    T.extra_cycles_planned = layout_extra_cycles(c, DC, CR, vr, T);  // recomputed, never trusted
    *L = T;
    return true;
}

}  // namespace ldpc
