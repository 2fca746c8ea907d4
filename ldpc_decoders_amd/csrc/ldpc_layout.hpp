// LDS layout planner for the fused backend (host side).
//
// The fused kernel gathers marginals (check phase) and check messages (variable phase) from the LDS with
// ds_read_b32, which services a wave64 access as two 32-lane groups over 32 dword banks: k distinct addresses
// on one bank inside a group cost k cycles.  With an arbitrary placement the 32 random addresses of a group
// collide ~3.5-way on average, and the gathers dominate the kernel.  H is fixed, so the placement can be chosen:
//
//   * every variable gets an LDS slot (group gv, bank bv), every check a slot (group gc, bank bc)
//     (slot = 64*(group/2) + 32*(group%2) + bank; group == the 32-lane half-wave that owns it);
//   * every check orders its dc edges freely (min-sum is order independent); a variable keeps the reference's
//     summation order ((c_a + c_b) + c_c), only its first two (commutative) positions may be swapped;
//   * check phase  : the 32 edges read by one half-wave instruction must hit 32 different variable banks.  For a
//     check group this is possible for all dc positions iff no variable bank receives more than dc of the
//     group's edges (Koenig: a bipartite multigraph of maximum degree dc splits into dc matchings);
//   * variable phase: for a variable group and position j the 32 checks read must sit on 32 different check banks.
//
// plan_fused_layout() minimises the excess over those capacities by simulated annealing (swap two variable
// slots / two check slots / flip a variable's first two positions), then edge-colours every check group.
// Two things make the annealer about ten times more effective per move than uniform random swaps: (1) the check-phase cost
// depends on (check GROUP, variable BANK) only and the variable-phase cost on (variable GROUP, position, check BANK) only, so
// most proposals keep the group and change the bank, or keep the bank and change the group -- they repair one phase without
// disturbing the other; (2) three proposals in four start from an item that currently sits on an overloaded cell.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "ldpc_common.hpp"

namespace ldpc {

struct FusedLayout {
    std::vector<int> chk_slot;  // [m]  slot index r*64 + lane
    std::vector<int> var_slot;  // [n]  slot index q*64 + lane
    std::vector<int> edge_pos;  // [E]  position of edge k inside its check's gather order (0..dc-1)
    std::vector<int> var_pos;   // [E]  position of edge k inside its variable's gather order (0..dv-1)
    double extra_cycles_identity = 0;  // LDS conflict cycles per sweep beyond the conflict-free minimum
    double extra_cycles_planned = 0;
    double base_cycles = 0;  // conflict-free LDS cycles of the gathers per sweep (one per half-wave instruction)
};

// Variable rounds come in up to three widths: each wave owns VR/nw consecutive rounds, the first `vrx` of them gather `dvx` messages
// per variable ("wide" rounds, for the high-degree variables of irregular codes), the last `vr2` gather two ("pair" rounds: most
// variables of the reference's irregular ensembles have two edges), the others gather DV.  vrx = vr2 = 0 for regular codes.  `reserved`
// trailing rounds hold no variables (the system row of the 16-wave shape, ldpc_fused.hip); with `reserved_half`
// only the upper 32 slots of that last round are reserved (the fp64 shapes: their system words fit 32 eight-byte slots).
struct VarRounds {
    int VR = 0, DV = 0, vrx = 0, dvx = 0;
    int nw = 1, reserved = 0;
    bool reserved_half = false;
    bool fixed_edge_order = false;  // every check keeps its edges in ascending-variable order (fp64 sum-product: the row sum of logs is order dependent)
    int vr2 = 0;
    int vrw() const { return VR / nw; }
    int per_wave() const { return vrx * dvx + (vrw() - vrx - vr2) * DV + vr2 * 2; }
    int width(int q) const { const int l = q % vrw(); return l < vrx ? dvx : (l >= vrw() - vr2 ? 2 : DV); }
    int first_gather(int q) const {  // index of (q, position 0) in the frame's list of variable-phase gathers
        const int w = q / vrw(), l = q % vrw(), mid = vrw() - vrx - vr2;
        return w * per_wave() + (l < vrx ? l * dvx : (l < vrx + mid ? vrx * dvx + (l - vrx) * DV : vrx * dvx + mid * DV + (l - vrx - mid) * 2));
    }
    int total_gathers() const { return nw * per_wave(); }
    int usable_slots() const { return VR * 64 - (reserved ? (reserved_half ? 32 : reserved * 64) : 0); }
    bool usable_slot(int s) const { return s < usable_slots(); }
};

// exact conflict model: sum over gather instructions and half-waves of (max distinct-address multiplicity - 1)
double layout_extra_cycles(const Code& c, int DC, int CR, const VarRounds& vr, const FusedLayout& L);

// trivial placement: checks in index order; variables in index order, except that a variable only fits a round that gathers at least
// as many messages as it has edges (the only placement constraint)
void identity_layout(const Code& c, int DC, const VarRounds& vr, FusedLayout* L);
// `moves` = annealing steps (about 1.7 M per second on one host core); the result is a deterministic function of the arguments
constexpr long kDefaultPlanMoves = 4000000;
void plan_fused_layout(const Code& c, int DC, int CR, const VarRounds& vr, uint64_t seed, long moves, FusedLayout* L);


// ---- plan store ---------------------------------------------------------------------------------------------------
// A plan is a pure function of (H, shape, planner version), so long annealing runs can be done once and kept: files
// <key>.plan, looked up in $LDPC_FUSED_PLAN_DIR (colon-separated) and then in the `plans` directory next to the package's
// csrc/ (tools/plan_codes.py writes them).  A file that does not validate against the code is ignored.
uint64_t layout_key(const Code& c, int DC, int CR, const VarRounds& vr, int NW);
bool layout_valid(const Code& c, int DC, int CR, const VarRounds& vr, const FusedLayout& L);
bool layout_load(const std::string& path, uint64_t key, const Code& c, int DC, int CR, const VarRounds& vr, FusedLayout* L);
bool layout_save(const std::string& path, uint64_t key, const Code& c, const FusedLayout& L);

}  // namespace ldpc
