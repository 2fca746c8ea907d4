// Fused on-chip backend, host side: chooses a kernel shape for a (code, algorithm, arithmetic), plans the LDS layout, builds the
// gather tables and launches.  The kernels are in ldpc_fused_kernels.hpp, instantiated by the ldpc_fused_shapes_*.hip units.
#include <dlfcn.h>
#include <fcntl.h>
#include <signal.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <ctime>

#include <algorithm>
#include <cstdlib>
#include <string>
#include <numeric>

#include "ldpc_bec_planes.hpp"
#include "ldpc_fused_kernels.hpp"
#include "ldpc_layout.hpp"

namespace ldpc {

struct FusedPlan {
    bool ok = false;
    int shape = -1;                    // index into all_shapes()
    int DC = 0, DV = 0, CR = 0, VR = 0, NW = 1;
    uint32_t* d_cn_tab = nullptr;      // [NW][(CRW*DC+1)/2][64] two 16-bit marg byte offsets per word
    uint32_t* d_vn_tab = nullptr;      // [NW][(VRW*DV+1)/2][64] two 16-bit c2v byte offsets per word
    int32_t* d_var_of_slot = nullptr;  // [VR*64] variable index of a slot, -1 for padding
    int32_t* d_slot_of_var = nullptr;  // [n rounded up to 4] slot (LDS dword index) of a variable
    unsigned long long* d_cn_active = nullptr;  // [CR] lanes holding a real check in round R
    // Frame dispenser: TWO sets of 8 counters (64 B apart); a launch uses one while the other is zeroed behind it on the same stream,
    // so no memset sits in front of the next launch.  The low-latency host path (ldpc_decode_host, a private non-blocking stream) has
    // a dispenser of its own: sharing one, its launch could zero the set a kernel of an asynchronous decode / simulate on another
    // stream is still drawing frames from (frames handed out twice), and that kernel's trailing memset could hit the set in use here.
    struct Dispenser {
        unsigned long long* d = nullptr;  // 2 x 8 counters
        int sel = 0;                      // set the next launch uses
        bool clean = false;               // set `sel` was zeroed by a memset enqueued on `stream` behind the last launch
        hipStream_t stream = nullptr;
    };
    Dispenser disp[2];                          // [0] decode / simulate on the caller's stream, [1] the low-latency host path
    int sync_off[4] = {0, 0, 0, 0};    // NW > 1: byte offset of a padded c2v slot owned by wave w (verdict / frame hand-off)
    int msync_off[4] = {0, 0, 0, 0};   // NW > 1: byte offset of a padded marginal slot owned by wave w (end-of-sweep hand-off)
    int zero_row = 0;                  // 1: the c2v area ends with an always-zero row
    int sys_off = 0;                   // NW = 16: byte offset of the system row
    uint32_t certain_entry = 0xffffffffu;  // gather-table entry of one of the certain slots (short check rows), none otherwise
    size_t lds_bytes = 0;
    int groups_per_cu = 0, num_cu = 0;
    double extra_identity = 0, extra_planned = 0, extra_built = -1, base_cycles = 0;
    bool plan_from_store = false;      // layout read from a stored plan file instead of annealed now
};

namespace {

template <typename T>
int upload_vec(const std::vector<T>& h, T** d) {
    LDPC_HIP_TRY(hipMalloc((void**)d, (h.size() + 1) * sizeof(T)));
    LDPC_HIP_TRY(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return LDPC_OK;
}


// every instantiated shape, in preference order (first match wins)
const std::vector<ShapeEntry>& all_shapes() {
    static const std::vector<ShapeEntry> v = [] {
        std::vector<ShapeEntry> out;
        for (auto fn : {fused_shapes_f32_dc6, fused_shapes_f32_dcx, fused_shapes_f64_dc6, fused_shapes_f64_dcx, fused_shapes_bec}) {
            int cnt = 0;
            const ShapeEntry* p = fn(&cnt);
            out.insert(out.end(), p, p + cnt);
        }
        return out;
    }();
    return v;
}

}  // namespace

// per-user plan cache: $XDG_CACHE_HOME/ldpc_decoders_amd/plans or ~/.cache/ldpc_decoders_amd/plans ("" if neither variable is set)
static std::string user_plan_dir() {
    if (const char* x = std::getenv("XDG_CACHE_HOME"))
        if (*x) return std::string(x) + "/ldpc_decoders_amd/plans";
    if (const char* h = std::getenv("HOME"))
        if (*h) return std::string(h) + "/.cache/ldpc_decoders_amd/plans";
    return std::string();
}

// where a freshly annealed plan is kept: $LDPC_FUSED_PLAN_SAVE ("none" = nowhere), by default the per-user cache -- so the
// second construction of a decoder for the same (code, shape) finds its plan instead of annealing again
static std::string plan_save_dir() {
    if (const char* e = std::getenv("LDPC_FUSED_PLAN_SAVE")) return std::string(e) == "none" ? std::string() : std::string(e);
    return user_plan_dir();
}

static void make_dirs(const std::string& path) {
    // directories this library creates are private to the user (plans are later LOADED from here); existing ones are left as they are
    for (size_t i = 1; i <= path.size(); ++i)
        if (i == path.size() || path[i] == '/') (void)mkdir(path.substr(0, i).c_str(), 0700);
}

// directories searched for stored layout plans, in this order: $LDPC_FUSED_PLAN_DIR (colon-separated), <package>/plans next to csrc/
// (the shipped long annealing runs), then the per-user cache (short runs made at decoder construction)
static std::vector<std::string> plan_dirs() {
    std::vector<std::string> dirs;
    if (const char* e = std::getenv("LDPC_FUSED_PLAN_DIR")) {
        std::string s(e);
        size_t a = 0;
        while (a <= s.size()) {
            const size_t b = s.find(':', a);
            const std::string d = s.substr(a, b == std::string::npos ? std::string::npos : b - a);
            if (!d.empty()) dirs.push_back(d);
            if (b == std::string::npos) break;
            a = b + 1;
        }
    }
    Dl_info info;
    if (dladdr((const void*)&plan_dirs, &info) && info.dli_fname) {
        std::string lib(info.dli_fname);
        const size_t slash = lib.rfind('/');
        const std::string here = slash == std::string::npos ? std::string(".") : lib.substr(0, slash);
        dirs.push_back(here + "/../plans");  // shipped plans (long annealing runs) before the user's own short ones
    }
    const std::string user = user_plan_dir();
    if (!user.empty()) dirs.push_back(user);
    return dirs;
}

bool fused_supported(const Decoder* d) { return d->fused && d->fused->ok; }

int fused_info(const Decoder* d, double* out8) {
    const FusedPlan* p = d->fused;
    for (int i = 0; i < 8; ++i) out8[i] = 0;
    if (!p || !p->ok) return LDPC_OK;
    out8[0] = p->NW;              // wavefronts per frame (0 = fused backend not available)
    out8[1] = p->base_cycles;     // conflict-free LDS cycles of the gathers per sweep
    out8[2] = p->extra_identity;  // extra bank-conflict cycles per sweep, trivial placement
    out8[3] = p->extra_built >= 0 ? p->extra_built : p->extra_planned;  // ... with the planned placement: of the gather tables as built (every lane's final
                                                                        // address, padding included) where the plan builder evaluates them, else the planner's figure (real edges)
    out8[4] = p->groups_per_cu * p->NW;  // resident waves per CU
    out8[5] = (double)p->lds_bytes;      // LDS bytes per frame
    out8[6] = p->CR;
    out8[7] = p->VR;
    return LDPC_OK;
}

// "k_fused_bp<0, 6, 3, 5, 10, 2, true, 0, 3>": the demangled name rocprofv3 / the code-object metadata give the kernel this decoder
// launches (sim: the Monte-Carlo variant) -- the key under which its PMC counters are filed in profiles/.  Empty: no fused kernel.
int fused_kernel_name(const Decoder* d, bool sim, char* buf, size_t len) {
    if (!buf || len == 0) return LDPC_E_ARG;
    buf[0] = 0;
    const FusedPlan* p = d->fused;
    if (!p || !p->ok) return LDPC_OK;
    const ShapeEntry& s = all_shapes()[p->shape];
    if (s.alg == ALG_BEC)  // the bit-sliced erasure kernels have no algorithm parameter
        snprintf(buf, len, "%s<%d, %d, %d, %d, %d, %d, %d>", sim ? "k_fused_becs_mc" : "k_fused_becs", s.DC, s.DV, s.CRW, s.VRW, s.NW, s.VRX, s.DVX);
    else
        snprintf(buf, len, "%s<%d, %d, %d, %d, %d, %d, %s, %d, %d>", s.esz == 8 ? "k_fused_f64" : "k_fused_bp", s.alg, s.DC, s.DV, s.CRW, s.VRW, s.NW,
                 sim ? "true" : "false", s.VRX, s.DVX);
    return LDPC_OK;
}

namespace {

// The kernel shape the fused backend uses for (code, algorithm, arithmetic), or -1 -- host-only logic, no device needed.
struct ShapeChoice {
    int si = -1;
    VarRounds vr;
    bool SYS = false;
    int esz = 4;
};

ShapeChoice choose_shape(const Code* c, int alg, int dtype) {
    ShapeChoice out;
    // fp64 message arithmetic (min-sum, sum-product): the 8-byte kernels; the erasure decoder is bit-sliced (ldpc_bec_kernels.hpp): one
    // kernel family whatever `dtype` says, 8-byte elements = two bit planes of 32 frames
    const int want_esz = (alg == ALG_BEC || dtype == DT_F64) ? 8 : 4;
    const bool full_dv = c->min_dv == c->max_dv;  // every variable has all its DV edges: no zero row needed
    const bool short_rows = c->min_dc != c->max_dc;
    if (c->min_dc < 1) return out;
    int force_nw = 0;
    if (const char* e = std::getenv("LDPC_FUSED_NW")) force_nw = atoi(e);
    const std::vector<ShapeEntry>& kShapes = all_shapes();
    for (int i = 0; i < (int)kShapes.size() && out.si < 0; ++i) {
        const ShapeEntry& s = kShapes[i];
        const int CR = fused_check_rows(s), VR = s.VRW * s.NW;
        const bool big = s.NW > 4 || (s.NW > 1 && s.VRX > 0) || s.esz == 8;  // shapes with a system row: (half) a row of variable slots is reserved
        if (s.esz != want_esz) continue;
        if (s.alg != alg || c->max_dc != s.DC || c->m > CR * 64 || c->n + (short_rows ? 1 : 0) > VR * 64 - (big ? (s.esz == 8 ? 32 : 64) : 0)) continue;
        if (force_nw && s.NW != force_nw) continue;
        if (s.VRX == 0) {
            if (short_rows || c->max_dv > s.DV) continue;
        } else {
            // a variable fits a round that gathers at least as many messages as it has edges: those above DV need a wide round, those above
            // two a wide or a DV round (shapes with pair rounds)
            int wide = 0, above2 = 0;
            for (int v = 0; v < c->n; ++v) {
                wide += (c->col_ptr[v + 1] - c->col_ptr[v]) > s.DV;
                above2 += (c->col_ptr[v + 1] - c->col_ptr[v]) > 2;
            }
            if (c->max_dv > s.DVX || wide > wide_rounds(s.VRX) * s.NW * 64) continue;
            if (pair_rounds(s.VRX) > 0 && above2 > (s.VRW - pair_rounds(s.VRX)) * s.NW * 64) continue;
        }
        if (s.NW > 1 && !big && (!full_dv || c->max_dv != s.DV || CR * 64 - c->m < s.NW)) continue;  // needs padded check slots for the hand-off
        out.si = i;
    }
    if (out.si < 0) return out;
    const ShapeEntry& shape = kShapes[out.si];
    out.esz = shape.esz;
    out.SYS = shape.NW > 4 || (shape.NW > 1 && shape.VRX > 0) || shape.esz == 8;  // system row (see the kernels)
    VarRounds& vr = out.vr;
    vr.VR = shape.VRW * shape.NW; vr.DV = shape.DV; vr.vrx = wide_rounds(shape.VRX); vr.vr2 = pair_rounds(shape.VRX); vr.dvx = shape.DVX; vr.nw = shape.NW; vr.reserved = out.SYS ? 1 : 0;
    vr.reserved_half = shape.esz == 8;  // the fp64 kernels' system words (36 dwords) fit the upper half of the last marginal row
    // fp64 sum-product sums log|tanh| over a row in the reference's order (ascending variable): its plans keep the edge order
    vr.fixed_edge_order = alg == ALG_SPA && dtype == DT_F64;
    return out;
}

// Layout of a chosen shape: a stored plan (a long annealing run done once, ldpc_layout.hpp "plan store") or an annealing run now
// (`moves` <= 0: $LDPC_FUSED_PLAN_MOVES / $LDPC_FUSED_PLAN_MS / the default), which is then kept in `save_dir` (if not empty).
// who holds a plan-store lock: "<pid> <host> <boot id> <pid namespace>\n".  A waiter trusts the pid (kill(pid, 0)) only when everything
// behind it equals its own.
static std::string read_small_file(const std::string& path) {
    std::string out;
    if (FILE* f = fopen(path.c_str(), "r")) {
        char buf[512];
        const size_t got = fread(buf, 1, sizeof(buf), f);
        out.assign(buf, got);
        fclose(f);
    }
    return out;
}
static const std::string& lock_identity() {
    static const std::string id = [] {
        char host[256] = "?";
        (void)gethostname(host, sizeof(host) - 1);
        std::string boot = read_small_file("/proc/sys/kernel/random/boot_id");
        while (!boot.empty() && (boot.back() == '\n' || boot.back() == ' ')) boot.pop_back();
        char ns[128] = "?";
        const ssize_t len = readlink("/proc/self/ns/pid", ns, sizeof(ns) - 1);
        if (len > 0) ns[len] = 0;
        return std::to_string((long)getpid()) + " " + host + " " + (boot.empty() ? "?" : boot) + " " + ns + "\n";
    }();
    return id;
}

bool obtain_layout(const Code* c, const ShapeChoice& ch, bool use_store, long moves, const std::string& save_dir, FusedLayout* L) {
    const ShapeEntry& shape = all_shapes()[ch.si];
    const int CR = fused_check_rows(shape);
    const uint64_t key = layout_key(*c, shape.DC, CR, ch.vr, shape.NW);
    char name[40];
    snprintf(name, sizeof(name), "%016llx.plan", (unsigned long long)key);
    if (use_store)
        for (const std::string& dir : plan_dirs())
            if (layout_load(dir + "/" + name, key, *c, shape.DC, CR, ch.vr, L)) return true;
    if (moves <= 0) {
        moves = kDefaultPlanMoves;
        if (const char* e = std::getenv("LDPC_FUSED_PLAN_MOVES")) moves = atol(e);
        else if (const char* ms = std::getenv("LDPC_FUSED_PLAN_MS")) moves = (long)(atof(ms) * 1700.0);
    }
    // Plan once per node: eight ranks (or run_sims.sh PARA) constructing a decoder for the same un-planned code would each anneal the
    // same plan (2.4 s+, a pure function of H and the shape).  The first process to create <save_dir>/.<key>.lock anneals and
    // publishes the file with an atomic rename; the others wait for it (bounded), then load it -- or anneal themselves if it never
    // comes (a crashed owner; a lock older than ten minutes is ignored).
    std::string lock;
    bool owner = true;
    if (!save_dir.empty() && use_store) {
        make_dirs(save_dir);
        lock = save_dir + "/." + name + ".lock";
        struct stat sb;
        if (stat(lock.c_str(), &sb) == 0 && time(nullptr) - sb.st_mtime > 600) (void)unlink(lock.c_str());
        const int fd = open(lock.c_str(), O_CREAT | O_EXCL | O_WRONLY, 0600);
        if (fd >= 0) {
            const std::string me = lock_identity();
            if (write(fd, me.data(), me.size()) != (ssize_t)me.size()) { /* the lock works without the identity; waiters then fall back to its age */ }
            (void)close(fd);
        } else if (errno == EEXIST) {
            owner = false;
            const double wait_s = 30.0 + (double)moves / 1.0e6;  // the owner needs about moves / 1.7e6 seconds
            for (double waited = 0; waited < wait_s; waited += 0.05) {
                if (layout_load(save_dir + "/" + name, key, *c, shape.DC, CR, ch.vr, L)) return true;
                if (stat(lock.c_str(), &sb) != 0) break;  // the lock is gone: published in between (checked below), or the owner died
                // the owner wrote its pid into the lock: a dead owner is not waited for (it would hold every newcomer for the full wait)
                // -- but a pid only means something in the owner's own pid namespace on the owner's own boot of the owner's own host (the plan
                // directory may be shared over NFS or between containers): anywhere else the waiter falls back to the lock's age
                long pid = 0;
                const std::string theirs = read_small_file(lock);
                const size_t sp = theirs.find(' ');
                if (sp != std::string::npos && theirs.substr(sp) == lock_identity().substr(lock_identity().find(' '))) pid = atol(theirs.c_str());
                if (pid > 0 && kill((pid_t)pid, 0) != 0 && errno == ESRCH) {
                    (void)unlink(lock.c_str());
                    break;
                }
                usleep(50000);
            }
            // the owner may have renamed the plan into place and removed the lock between the load and the stat above
            if (layout_load(save_dir + "/" + name, key, *c, shape.DC, CR, ch.vr, L)) return true;
            lock.clear();  // not ours to remove
        } else {
            lock.clear();  // read-only or missing directory: anneal without publishing
        }
    }
    plan_fused_layout(*c, shape.DC, CR, ch.vr, 0x1200u, moves, L);
    if (!save_dir.empty()) {
        make_dirs(save_dir);
        const std::string tmp = save_dir + "/." + name + "." + std::to_string((long)getpid());
        if (layout_save(tmp, key, *c, *L)) (void)rename(tmp.c_str(), (save_dir + "/" + name).c_str());  // atomic: readers never see half a file
    }
    // remove the lock only if it is still OURS (a waiter that judged us dead may have removed it and a newcomer taken a fresh one)
    if (owner && !lock.empty() && read_small_file(lock) == lock_identity()) (void)unlink(lock.c_str());
    return false;
}

}  // namespace

// Host-only planning (no GPU): which shape the fused backend would use and its layout plan, annealed with `moves` moves and stored
// in `out_dir` -- what tools/plan_codes.py calls to produce the shipped plans.  info4 = {waves per frame (0: no fused shape),
// conflict-free gather cycles, extra cycles of the trivial placement, extra cycles of the plan}.
int fused_plan_host(const Code* c, int alg, int dtype, long moves, const char* out_dir, double* info4) {
    for (int i = 0; i < 4; ++i) info4[i] = 0;
    const ShapeChoice ch = choose_shape(c, alg, dtype);
    if (ch.si < 0) return LDPC_OK;
    FusedLayout L;
    // moves < 0: exactly what ldpc_decoder_create does -- plan store first, otherwise ONE process per node anneals the default run
    const bool found = obtain_layout(c, ch, moves < 0, moves, out_dir ? std::string(out_dir) : (moves < 0 ? plan_save_dir() : std::string()), &L);
    info4[0] = (found ? -1.0 : 1.0) * all_shapes()[ch.si].NW;  // negative: the plan came out of the store (moves < 0 only)
    info4[1] = L.base_cycles;
    info4[2] = L.extra_cycles_identity;
    info4[3] = L.extra_cycles_planned;
    return LDPC_OK;
}


// Tables of the bit-sliced erasure kernels (ldpc_bec_kernels.hpp) from a layout plan.  Slab layout in units of 8-byte elements:
// v2c rows [NW * VNK][64] (variable-major: row = the variable phase's gather index, VarRounds::first_gather), summary rows [CR][64],
// the system row (element 0 = {0,0}, element 1 = a known 0).  Check row R belongs to wave R % NW as its local row R / NW.
static int becs_build_plan(Decoder* d, FusedPlan* p, const ShapeEntry& shape, const VarRounds& vr, const FusedLayout& L) {
    const Code* c = d->code;
    const int DC = shape.DC, NW = shape.NW, CRW = shape.CRW, VRW = shape.VRW;
    const int CR = fused_check_rows(shape), VR = VRW * NW, NPAD = VR * 64;
    const int VNK = vr.per_wave();
    const int CNW = (CRW * DC + 1) / 2, VNW = (VNK + 1) / 2;
    const uint32_t sum_base = (uint32_t)NW * VNK * 64, sys_base = sum_base + (uint32_t)CR * 64;
    const uint32_t zero_e = sys_base, known0_e = sys_base + 1;
    p->lds_bytes = (size_t)(sys_base + 128) * 8;  // system row + the row of the second accumulator slot's histogram (Monte-Carlo kernel)
    if (p->lds_bytes > (size_t)160 * 1024 || sys_base + 128 > 65536u || CR > CRW * NW) return LDPC_OK;  // plan stays !ok
    std::vector<uint32_t> cn_tab((size_t)NW * CNW * 64, 0), vn_tab((size_t)NW * VNW * 64, 0);
    std::vector<int32_t> var_of_slot((size_t)NPAD, -1);
    for (int v = 0; v < c->n; ++v) var_of_slot[L.var_slot[v]] = v;
    // table entries: byte offsets while the slab fits 64 KB, element indices beyond (BecShape::BYTE_TAB, the kernels' rule)
    const uint32_t tab_scale = (sys_base * 8u + 1024u <= 65536u) ? 8u : 1u;
    auto put16 = [tab_scale](std::vector<uint32_t>& tab, int words_per_wave, int wv, int k, int lane, uint32_t val) {
        val *= tab_scale;
        uint32_t& w = tab[((size_t)wv * words_per_wave + (k >> 1)) * 64 + lane];
        w = (k & 1) ? ((w & 0x0000ffffu) | (val << 16)) : ((w & 0xffff0000u) | val);
    };
    // check side: position j of check slot (R, lane) gathers the v2c element its variable writes for this edge; a short row reads a known 0
    std::vector<int64_t> cn_e((size_t)CRW * NW * DC * 64, -1);  // [R][j][lane], -1 = padded check lane
    for (int cc = 0; cc < c->m; ++cc) {
        const int R = L.chk_slot[cc] / 64, lane = L.chk_slot[cc] % 64;
        for (int j = 0; j < DC; ++j) cn_e[((size_t)R * DC + j) * 64 + lane] = known0_e;
        for (int k = c->row_ptr[cc]; k < c->row_ptr[cc + 1]; ++k) {
            const int vs = L.var_slot[c->edge_var[k]];
            cn_e[((size_t)R * DC + L.edge_pos[k]) * 64 + lane] = (int64_t)(vr.first_gather(vs / 64) + L.var_pos[k]) * 64 + vs % 64;
        }
    }
    // padded check lanes may read anything: an address a real lane of their half-wave reads anyway (LDS broadcast, no extra cycle)
    for (size_t g0 = 0; g0 < cn_e.size(); g0 += 32) {
        int64_t rep = zero_e;
        for (int l = 0; l < 32; ++l)
            if (cn_e[g0 + l] >= 0) { rep = cn_e[g0 + l]; break; }
        for (int l = 0; l < 32; ++l)
            if (cn_e[g0 + l] < 0) cn_e[g0 + l] = rep;
    }
    for (int R = 0; R < CRW * NW; ++R)
        for (int j = 0; j < DC; ++j)
            for (int lane = 0; lane < 64; ++lane) put16(cn_tab, CNW, R % NW, (R / NW) * DC + j, lane, (uint32_t)cn_e[((size_t)R * DC + j) * 64 + lane]);
    // variable side: gather position g of slot (Q, lane) reads the summary of its check; missing edges -- and every position of a padded
    // slot, which must stay silent -- read {0,0}: "no message"
    for (int Q = 0; Q < VR; ++Q)
        for (int j = 0; j < vr.width(Q); ++j)
            for (int lane = 0; lane < 64; ++lane) put16(vn_tab, VNW, Q / VRW, vr.first_gather(Q) % VNK + j, lane, zero_e);
    for (int64_t k = 0; k < c->E; ++k) {
        const int vs = L.var_slot[c->edge_var[k]], cs = L.chk_slot[c->edge_chk[k]];
        const int Q = vs / 64;
        put16(vn_tab, VNW, Q / VRW, vr.first_gather(Q) % VNK + L.var_pos[k], vs % 64, sum_base + (uint32_t)cs);
    }
    LDPC_HIP_TRY(hipSetDevice(c->device));
    LDPC_TRY(upload_vec(cn_tab, &p->d_cn_tab));
    LDPC_TRY(upload_vec(vn_tab, &p->d_vn_tab));
    LDPC_TRY(upload_vec(var_of_slot, &p->d_var_of_slot));
    {
        std::vector<int32_t> sov(L.var_slot.begin(), L.var_slot.end());
        while (sov.size() % 4) sov.push_back(0);
        LDPC_TRY(upload_vec(sov, &p->d_slot_of_var));
    }
    LDPC_HIP_TRY(hipMalloc((void**)&p->disp[0].d, 2 * 8 * 64));
    hipDeviceProp_t prop;
    LDPC_HIP_TRY(hipGetDeviceProperties(&prop, c->device));
    p->num_cu = prop.multiProcessorCount;
    const int by_lds = (int)((size_t)160 * 1024 / p->lds_bytes);
    int cap = 8;
    p->groups_per_cu = by_lds < cap ? by_lds : cap;
    LDPC_HIP_TRY(hipFuncSetAttribute(shape.kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds_bytes));
    LDPC_HIP_TRY(hipFuncSetAttribute(shape.kernel_sim, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds_bytes));
    p->ok = p->groups_per_cu >= 1;
    return LDPC_OK;
}

int fused_plan_create(Decoder* d) {
    const Code* c = d->code;
    d->fused = new FusedPlan();
    FusedPlan* p = d->fused;
    const bool short_rows = c->min_dc != c->max_dc;
    if (d->dtype == DT_F16) return LDPC_OK;  // fp16 storage is a mode of the streaming kernels: no LDS-resident plan
    const ShapeChoice ch = choose_shape(c, d->alg, d->dtype);
    const int si = ch.si;
    if (si < 0) return LDPC_OK;
    const std::vector<ShapeEntry>& kShapes = all_shapes();
    const ShapeEntry& shape = kShapes[si];
    const int DC = shape.DC, DV = shape.DV, NW = shape.NW, CRW = shape.CRW, VRW = shape.VRW;
    const int CR = fused_check_rows(shape), VR = VRW * NW, NPAD = VR * 64;  // CR < CRW * NW: the last waves run fewer check rows
    const VarRounds& vr = ch.vr;
    const bool BIG = NW > 4;                         // dword-index tables, 160 KB frame
    const int esz = shape.esz;
    const bool SYS = ch.SYS;
    p->sys_off = SYS ? (NPAD - (vr.reserved_half ? 32 : 64)) * esz : 0;
    p->shape = si; p->DC = DC; p->DV = DV; p->CR = CR; p->VR = VR; p->NW = NW;
    p->zero_row = (NW == 1) ? 1 : 0;

    // ---- layout: check c -> slot (R, lane), variable v -> slot, edge positions (ldpc_layout.hpp)
    FusedLayout L;
    const char* mode = std::getenv("LDPC_FUSED_LAYOUT");
    if (mode && std::string(mode) == "identity") {
        identity_layout(*c, DC, vr, &L);
        if (NW > 1 && !SYS) {  // spread the checks / variables evenly over the waves so that every wave keeps padded slots
            for (int cc = 0; cc < c->m; ++cc) L.chk_slot[cc] = (int)((int64_t)cc * CR * 64 / c->m);
            for (int v = 0; v < c->n; ++v) L.var_slot[v] = (int)((int64_t)v * NPAD / c->n);
        }
        L.base_cycles = 2.0 * (CR * DC + vr.total_gathers());
        L.extra_cycles_identity = L.extra_cycles_planned = layout_extra_cycles(*c, DC, CR, vr, L);
    } else {
        p->plan_from_store = obtain_layout(c, ch, !(mode && std::string(mode) == "replan"), 0, plan_save_dir(), &L);
    }
    p->extra_identity = L.extra_cycles_identity;
    p->extra_planned = L.extra_cycles_planned;
    p->base_cycles = L.base_cycles;
    if (d->alg == ALG_BEC) return becs_build_plan(d, p, shape, vr, L);
    const std::vector<int>&chk_slot = L.chk_slot, &var_slot = L.var_slot, &edge_pos = L.edge_pos, &var_pos = L.var_pos;

    const int VNK = vr.total_gathers() / NW;  // gathers of one wave's variable phase (wide rounds exist only for NW == 1)
    const int CNW = (CRW * DC + 1) / 2, VNW = (VNK + 1) / 2;
    std::vector<uint32_t> cn_tab((size_t)NW * CNW * 64, 0), vn_tab((size_t)NW * VNW * 64, 0);
    std::vector<int32_t> var_of_slot((size_t)NPAD, -1);
    std::vector<u64> cn_active((size_t)CRW * NW, 0);  // indexed (wave, row of the wave); rows a wave does not have stay 0
    // global gather index K = R*DC + j (check side) or Q*DV + j (variable side) -> (wave, packed half-word) of the table
    auto put16 = [](std::vector<uint32_t>& tab, int words_per_wave, int per_wave, int K, int lane, uint32_t val) {
        const int wv = K / per_wave, k = K % per_wave;
        uint32_t& w = tab[((size_t)wv * words_per_wave + (k >> 1)) * 64 + lane];
        w = (k & 1) ? ((w & 0x0000ffffu) | (val << 16)) : ((w & 0xffff0000u) | val);
    };
    const uint32_t c2v_base = (uint32_t)NPAD * esz;
    // gather tables hold 16 bits per entry: byte offsets while the frame fits 64 KB, element indices beyond (fp32: the 16-wave shape;
    // fp64: exactly the kernel's own WIDE rule)
    const size_t frame_bytes = (size_t)(NPAD + CR * DC * 64) * esz;
    const int tab_shift = esz == 8 ? (frame_bytes > 65536 ? 3 : 0) : (BIG ? 2 : 0);
    std::vector<int64_t> cn_addr((size_t)CR * DC * 64, -1), vn_addr((size_t)vr.total_gathers() * 64, -1);  // byte offsets, -1 = padded lane
    for (int v = 0; v < c->n; ++v) var_of_slot[var_slot[v]] = v;
    // Short check rows are padded with reads of a "certain" variable slot (marked -2: +inf marginal, see the kernel); a variable with
    // fewer edges than its round gathers reads a zero word for the missing ones.  Neither costs an LDS cycle when the word sits on a bank
    // that no real lane of the same half-wave gather uses (all such lanes of the half-wave then read ONE address: a broadcast) -- so there
    // is a certain slot and a zero word on as many of the 32 banks as the frame has room for, and every half-wave takes the one whose bank
    // is free.  (Rounds 3-5 had ONE of each: on the irregular n = 10 000 code 770 of the 16-wave kernel's 787 measured bank-conflict
    // cycles per frame-sweep were these two words, the planner's 65 the rest.)
    constexpr int64_t SHORT_ROW = -2, MISSING_EDGE = -3;  // resolved per half-wave below; -1 = padded lane
    std::vector<int64_t> certain_addr, zero_words;
    if (short_rows) {
        bool bank_has[32] = {};
        for (int s = vr.usable_slots() - 1; s >= 0 && certain_addr.size() < 32; --s)
            if (var_of_slot[s] == -1 && !bank_has[s & 31]) {
                bank_has[s & 31] = true;
                var_of_slot[s] = -2;
                p->certain_entry = (uint32_t)(((int64_t)s * esz) >> tab_shift);
                certain_addr.push_back((int64_t)s * esz);
            }
        if (certain_addr.empty()) return LDPC_OK;
    }
    if (SYS) {  // the free words of the system row (the kernels zero them: fp32 words 33..63; fp64 doubles 9..31 in the irregular shapes, double 17 otherwise)
        bool wide_rounds = false;
        for (int q = 0; q < VR; ++q) wide_rounds |= vr.width(q) > DV;
        for (int i = (esz == 8 ? (wide_rounds ? 9 : 17) : 33); i < (esz == 8 ? (wide_rounds ? 32 : 18) : 64); ++i) zero_words.push_back((int64_t)p->sys_off + (int64_t)i * esz);
    } else if (p->zero_row) {  // the always-zero row behind the c2v area
        for (int l = 0; l < 64; ++l) zero_words.push_back((int64_t)c2v_base + (int64_t)(CR * DC * 64 + l) * esz);
    }
    for (int cc = 0; cc < c->m; ++cc) {
        const int R = chk_slot[cc] / 64, lane = chk_slot[cc] % 64;
        cn_active[R] |= 1ull << lane;
        unsigned used = 0;
        for (int k = c->row_ptr[cc]; k < c->row_ptr[cc + 1]; ++k) {
            cn_addr[(size_t)(R * DC + edge_pos[k]) * 64 + lane] = (int64_t)var_slot[c->edge_var[k]] * esz;
            used |= 1u << edge_pos[k];
        }
        for (int j = 0; j < DC; ++j)
            if (!((used >> j) & 1u)) cn_addr[(size_t)(R * DC + j) * 64 + lane] = SHORT_ROW;
    }
    for (int v = 0; v < c->n; ++v) {
        const int Q = var_slot[v] / 64, lane = var_slot[v] % 64;
        // a real variable with fewer edges than its round gathers sums zeros for the missing ones
        if (!zero_words.empty())
            for (int j = 0; j < vr.width(Q); ++j) vn_addr[(size_t)(vr.first_gather(Q) + j) * 64 + lane] = MISSING_EDGE;
        for (int pidx = c->col_ptr[v]; pidx < c->col_ptr[v + 1]; ++pidx) {
            const int k = c->col_edge[pidx], cs = chk_slot[c->edge_chk[k]];
            // row of the message of (check row Rg = cs / 64, edge position j): Rg * DC + j; the 16-wave shape interleaves the waves' rows
            // (local row k of wave w at k * NW + w, see fused_bp_body) so that its c2v stores reach most rows without an address register
            const int Rg = cs / 64, jj = edge_pos[k];
            const int row = (BIG && esz == 4) ? ((Rg % CRW) * DC + jj) * NW + Rg / CRW : Rg * DC + jj;  // (k_fused_bp's 16-wave shape only: the fp64 kernels keep the plain order)
            vn_addr[(size_t)(vr.first_gather(Q) + var_pos[k]) * 64 + lane] = c2v_base + (int64_t)(row * 64 + cs % 64) * esz;
        }
    }
    for (int64_t a : certain_addr) {  // a certain slot sums nothing: every gather reads a zero word
        const int slot = (int)(a / esz), Q = slot / 64, lane = slot % 64;
        for (int j = 0; j < vr.width(Q); ++j) vn_addr[(size_t)(vr.first_gather(Q) + j) * 64 + lane] = MISSING_EDGE;
    }
    // per half-wave: the word (of `words`) on the bank the fewest distinct real addresses of this gather use -- none at all whenever there
    // is a choice, since a half-wave with k such lanes leaves at least k banks free
    auto resolve = [esz](std::vector<int64_t>& addr, int64_t mark, const std::vector<int64_t>& words) {
        for (size_t g0 = 0; g0 < addr.size(); g0 += 32) {
            bool any = false;
            for (int l = 0; l < 32; ++l) any |= addr[g0 + l] == mark;
            if (!any) continue;
            int load[32] = {};
            for (int l = 0; l < 32; ++l) {
                const int64_t a = addr[g0 + l];
                if (a < 0) continue;
                bool dup = false;
                for (int i = 0; i < l; ++i) dup |= addr[g0 + i] == a;
                if (!dup) ++load[(a / esz) & 31];
            }
            int64_t best = words[0];
            for (int64_t wd : words)
                if (load[(wd / esz) & 31] < load[(best / esz) & 31]) best = wd;
            for (int l = 0; l < 32; ++l)
                if (addr[g0 + l] == mark) addr[g0 + l] = best;
        }
    };
    if (!certain_addr.empty()) resolve(cn_addr, SHORT_ROW, certain_addr);
    if (!zero_words.empty()) resolve(vn_addr, MISSING_EDGE, zero_words);
    // padded lanes may read anything: let them repeat an address of their own half-wave (LDS broadcast, no extra cycle)
    auto fill_padding = [](std::vector<int64_t>& addr, int64_t fallback) {
        for (size_t g0 = 0; g0 < addr.size(); g0 += 32) {
            int64_t rep = fallback;
            for (int l = 0; l < 32; ++l)
                if (addr[g0 + l] >= 0) { rep = addr[g0 + l]; break; }
            for (int l = 0; l < 32; ++l)
                if (addr[g0 + l] < 0) addr[g0 + l] = rep;
        }
    };
    // Padded check lanes repeat an address of their own half-wave (broadcast, no extra LDS cycle); the kernels mask them out of the
    // syndrome (cn_valid).  The 16-wave fp32 shape has no register for that mask: there a padded lane reads zero words -- sign bit 0 in
    // every position, invisible in the syndrome, and its messages stay 0 -- again the one on the bank its half-wave leaves free.  (Rounds
    // 3-5 let it read ONE real marginal dc times, on whichever bank collided least over the dc gathers: with 31 real lanes on 31 banks in
    // each of them that was ~3 extra cycles for nearly every half-wave that holds a padded lane.)
    if (BIG && esz == 4) resolve(cn_addr, -1, zero_words);
    fill_padding(cn_addr, 0);
    fill_padding(vn_addr, c2v_base);
    {  // bank-conflict cycles per sweep of the tables AS BUILT (every lane's final address; the planner's figure covers real edges only)
        auto built_cost = [&](const std::vector<int64_t>& addr) {
            long extra = 0;
            for (size_t g0 = 0; g0 < addr.size(); g0 += 32) {
                int mx = 1;
                for (int b = 0; b < 32; ++b) {
                    int distinct = 0;
                    int64_t seen[32];
                    for (int l = 0; l < 32; ++l) {
                        const int64_t a = addr[g0 + l] / esz;
                        if ((a & 31) != b) continue;
                        bool dup = false;
                        for (int i = 0; i < distinct; ++i) dup |= seen[i] == a;
                        if (!dup) seen[distinct++] = a;
                    }
                    mx = std::max(mx, distinct);
                }
                extra += mx - 1;
            }
            return extra;
        };
        p->extra_built = (double)(built_cost(cn_addr) + built_cost(vn_addr));
    }
    for (int K = 0; K < CR * DC; ++K)
        for (int lane = 0; lane < 64; ++lane) put16(cn_tab, CNW, CRW * DC, K, lane, (uint32_t)(cn_addr[(size_t)K * 64 + lane] >> tab_shift));
    for (int K = 0; K < vr.total_gathers(); ++K)
        for (int lane = 0; lane < 64; ++lane) put16(vn_tab, VNW, VNK, K, lane, (uint32_t)(vn_addr[(size_t)K * 64 + lane] >> tab_shift));
    if (NW > 1 && !SYS) {
        // hand-off words: for each wave the c2v slot (position dc-1) of one of its padded check lanes, in its LAST round
        // that has one (so the wave's own garbage write to it precedes the verdict write in program order)
        for (int wv = 0; wv < NW; ++wv) {
            int found = -1;
            for (int R = (wv + 1) * CRW - 1; R >= wv * CRW && found < 0; --R)
                for (int lane = 0; lane < 64 && found < 0; ++lane)
                    if (!((cn_active[R] >> lane) & 1ull)) found = (int)c2v_base + ((R * DC + DC - 1) * 64 + lane) * esz;
            if (found < 0) return LDPC_OK;  // (plan stays !ok -> streaming backend; practically unreachable)
            p->sync_off[wv] = found;
            int mfound = -1;
            for (int s = wv * VRW * 64; s < (wv + 1) * VRW * 64 && mfound < 0; ++s)
                if (var_of_slot[s] < 0) mfound = s * esz;
            if (mfound < 0) return LDPC_OK;
            p->msync_off[wv] = mfound;
        }
    }
    p->lds_bytes = (size_t)(NPAD + (CR * DC + p->zero_row) * 64) * esz;
    if (p->lds_bytes > ((BIG || tab_shift) ? (size_t)160 * 1024 : (size_t)65535)) return LDPC_OK;  // 16-bit byte offsets / element indices
    LDPC_HIP_TRY(hipSetDevice(c->device));
    LDPC_TRY(upload_vec(cn_tab, &p->d_cn_tab));
    LDPC_TRY(upload_vec(vn_tab, &p->d_vn_tab));
    LDPC_TRY(upload_vec(var_of_slot, &p->d_var_of_slot));
    {
        std::vector<int32_t> sov(var_slot.begin(), var_slot.end());
        while (sov.size() % 4) sov.push_back(0);
        LDPC_TRY(upload_vec(sov, &p->d_slot_of_var));
    }
    LDPC_TRY(upload_vec(cn_active, &p->d_cn_active));
    LDPC_HIP_TRY(hipMalloc((void**)&p->disp[0].d, 2 * 8 * 64));  // 2 x 8 frame counters, one cache line apart
    hipDeviceProp_t prop;
    LDPC_HIP_TRY(hipGetDeviceProperties(&prop, c->device));
    p->num_cu = prop.multiProcessorCount;
    const int by_lds = (int)((size_t)160 * 1024 / p->lds_bytes);
    // resident frames per CU (waves: NW x that): 8 for the n = 1200 shapes (LDS-bound anyway); the small one-wave shape
    // (n <= 512, 8 KB of LDS, built for 128 VGPRs) runs 16 -- measured +13 % on 512_3_6_rand_ldpc_2
    int cap = (NW == 1 && CRW <= 4) ? 16 : 8;
    p->groups_per_cu = by_lds < cap ? by_lds : cap;
    LDPC_HIP_TRY(hipFuncSetAttribute(shape.kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds_bytes));
    if (shape.kernel_sim) LDPC_HIP_TRY(hipFuncSetAttribute(shape.kernel_sim, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds_bytes));
    for (const void* kg : {shape.kernel_grid, shape.kernel_sim_grid})
        if (kg) LDPC_HIP_TRY(hipFuncSetAttribute(kg, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds_bytes));
    p->ok = p->groups_per_cu >= 1;
    return LDPC_OK;
}

void fused_plan_destroy(Decoder* d) {
    FusedPlan* p = d->fused;
    if (!p) return;
    for (void* q : {(void*)p->d_cn_tab, (void*)p->d_vn_tab, (void*)p->d_var_of_slot, (void*)p->d_slot_of_var, (void*)p->d_cn_active, (void*)p->disp[0].d, (void*)p->disp[1].d})
        if (q) (void)hipFree(q);
    delete p;
    d->fused = nullptr;
}

static int fused_launch(Decoder* d, FusedArgs& a, bool sim, int64_t B, int32_t max_iter, uint32_t flags, hipStream_t st) {
    FusedPlan* p = d->fused;
    if (B >= ((int64_t)1 << 31)) {
        set_error("fused backend: at most 2^31-1 frames per call");
        return LDPC_E_ARG;
    }
    const ShapeEntry& shape = all_shapes()[p->shape];
    const Code* c = d->code;
    FusedPlan::Dispenser& dsp = p->disp[d->after_kernel_event ? 1 : 0];
    if (!dsp.d) LDPC_HIP_TRY(hipMalloc((void**)&dsp.d, 2 * 8 * 64));
    unsigned long long* next_set = dsp.d + (size_t)dsp.sel * 64;
    if (!(dsp.clean && dsp.stream == st)) LDPC_HIP_TRY(hipMemsetAsync(next_set, 0, 8 * 64, st));
    long long groups = (long long)p->num_cu * p->groups_per_cu;
    if (a.rounds < 1) a.rounds = 1;
    if (a.rounds == 1) a.round_stride = 0;
    const long long units = shape.alg == ALG_BEC ? (B + 31) / 32 * a.rounds : B;  // the bit-sliced erasure kernels hand out slabs of 32 frames
    if (groups > units) groups = units;
    a.B = B;
    a.n = c->n;
    a.max_iter = max_iter > 0 ? max_iter : 100000;
    a.flags = flags;
    a.cn_tab = p->d_cn_tab;
    a.vn_tab = p->d_vn_tab;
    a.var_of_slot = p->d_var_of_slot;
    a.slot_of_var = p->d_slot_of_var;
    a.cn_active = p->d_cn_active;
    a.next_frame = next_set;
    for (int i = 0; i < 4; ++i) {
        a.sync_off[i] = p->sync_off[i];
        a.msync_off[i] = p->msync_off[i];
    }
    a.zero_row = p->zero_row;
    a.sys_off = p->sys_off;
    a.certain_entry = p->certain_entry;
    {  // 32-bit partial counters of a workgroup (sim_count): bit errors <= n and sweeps <= max_iter per frame
        const long long per_frame = a.n > a.max_iter ? a.n : a.max_iter;
        long long fe = ((long long)1 << 31) / (per_frame > 0 ? per_frame : 1);
        a.flush_every = (int)(fe < 1 ? 1 : (fe > BECS_FLUSH_MAX ? BECS_FLUSH_MAX : fe));  // ceiling shared with the kernel's packed 16-bit counters
    }
    // exact-in-fp32 mode: the guarded variant of the kernel, its grid constants and the violation counter
    const int grid_k = LDPC_FLAG_PRIOR_GRID_OF(flags);
    const void* kern = sim ? shape.kernel_sim : shape.kernel;
    if (grid_k >= 0) {
        if (shape.esz == 8) {
            // fp64 arithmetic needs no guard; the Monte-Carlo kernels of the fp64 / erasure decoders draw their noise unquantised
            if (sim) {
                set_error("prior grid: ldpc_simulate quantises in the fp32 min-sum kernels; fp64 decoders take grid priors through ldpc_channel + ldpc_decode");
                return LDPC_E_UNSUPPORTED;
            }
        } else {
            kern = sim ? shape.kernel_sim_grid : shape.kernel_grid;
            if (!kern || grid_k > 23) {
                set_error("prior grid: no guarded kernel for this (code, algorithm) -- fp32 min-sum shapes of n = 1200 and n = 10 000 have one");
                return LDPC_E_UNSUPPORTED;
            }
            if (!d->gridviol.p) {
                LDPC_TRY(d->gridviol.reserve((size_t)(1 + GRID_REDO_CAP) * 8));
                LDPC_HIP_TRY(hipMemsetAsync(d->gridviol.p, 0, (size_t)(1 + GRID_REDO_CAP) * 8, st));
            }
            a.grid_scale = (float)ldexp(1.0, grid_k);
            a.grid_inv = (float)ldexp(1.0, -grid_k);
            // L = 2^(24-k) / (dv_max + 2), rounded down to a power of two: |prior| < L and |c2v| < L keep every sum of the iteration
            // below 2^(24-k), i.e. exact (the kernel's comment at `gmax`)
            // (irregular shapes keep the older scheme: magnitudes AND marginals below 2^(21-k))
            int shift = shape.VRX > 0 ? 3 : 0;
            while ((1 << shift) < c->max_dv + 2 && shape.VRX == 0) ++shift;
            a.grid_limit = (float)ldexp(1.0, 24 - grid_k - shift);
            a.grid_viol = (unsigned long long*)d->gridviol.p;
        }
    }
    void* args[] = {&a};
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (d->profile) {
        LDPC_TRY(prof_event(d, 0, &e0));
        LDPC_TRY(prof_event(d, 1, &e1));
        LDPC_HIP_TRY(hipEventRecord(e0, st));
    }
    LDPC_HIP_TRY(hipLaunchKernel(kern, dim3((unsigned)groups), dim3(64 * shape.NW), args, p->lds_bytes, st));
    if (d->after_kernel_event) LDPC_HIP_TRY(hipEventRecord(d->after_kernel_event, st));  // low-latency host path: wait for the kernel only
    // the other counter set for the next launch, zeroed behind this kernel (off the critical path of both launches)
    dsp.sel ^= 1;
    dsp.clean = hipMemsetAsync(dsp.d + (size_t)dsp.sel * 64, 0, 8 * 64, st) == hipSuccess;
    dsp.stream = st;
    if (d->profile) {
        LDPC_HIP_TRY(hipEventRecord(e1, st));
        LDPC_HIP_TRY(hipStreamSynchronize(st));
        LDPC_TRY(prof_collect(d, {{2, e0, e1}}));
    }
    d->last_sweeps = max_iter;
    d->last_backend = BK_FUSED;
    return LDPC_OK;
}

int fused_decode(Decoder* d, const void* priors, const uint8_t* y0, int64_t B, int32_t max_iter, uint32_t flags,
                 uint8_t* xhat, int32_t* iters, void* soft_out, hipStream_t st) {
    FusedPlan* p = d->fused;
    if (!p || !p->ok) {
        set_error("fused backend not available for this decoder");
        return LDPC_E_UNSUPPORTED;
    }
    if (B <= 0) return LDPC_OK;
    if (d->alg == ALG_BEC ? !y0 : !priors) {
        set_error(d->alg == ALG_BEC ? "erasure decoder needs the received symbols (y0)" : "priors pointer is null");
        return LDPC_E_ARG;
    }
    FusedArgs a{};
    a.priors = priors;
    a.y0 = y0;
    a.xhat = xhat;
    a.iters = iters;
    a.soft = soft_out;
    return fused_launch(d, a, false, B, max_iter, flags, st);
}

// channel -> LLR -> decode -> count in ONE kernel (BI-AWGN, all-`codeword` word): priors never touch HBM.
bool fused_simulate_supported(const Decoder* d, int channel, double param, int hist_bins) {
    if (!fused_supported(d) || hist_bins < 0 || hist_bins > 60) return false;  // 0 bins: counters only (what main.py asks for); lanes 60..63 hold the four counters
    if (d->alg == ALG_BEC) return channel == CH_BEC;
    if (channel == CH_BSC && !(param > 0.0 && param < 0.5)) {
        // the in-kernel BSC needs llr > 0 (the received bit is the sign of the prior); the fp64 kernels can still count on priors
        // generated into HBM by the channel kernel
        return all_shapes()[d->fused->shape].esz == 8;
    }
    return channel == CH_BIAWGN || channel == CH_BSC;
}

bool fused_simulate_rounds_supported(const Decoder* d) {  // one launch for several rounds: the erasure Monte-Carlo kernel (continuous refill)
    return fused_supported(d) && d->alg == ALG_BEC;
}

int fused_simulate(Decoder* d, int channel, double param, int codeword, uint64_t seed, uint64_t stream_id, uint64_t frame0, int64_t B,
                   int32_t max_iter, uint32_t flags, int32_t hist_bins, int64_t* counters, hipStream_t st, int rounds, uint64_t round_stride) {
    if (B <= 0) return LDPC_OK;
    if (rounds > 1 && (!fused_simulate_rounds_supported(d) || (B + 31) / 32 * (int64_t)rounds >= ((int64_t)1 << 31))) {
        set_error("fused_simulate: several rounds per launch are the erasure kernel's (at most 2^31 slabs of 32 frames)");
        return LDPC_E_UNSUPPORTED;
    }
    if (all_shapes()[d->fused->shape].esz == 8 && channel == CH_BSC && !(param > 0.0 && param < 0.5)) {
        // fp64, BSC beyond p = 1/2 (negative LLRs): priors and the received word of a chunk are generated into HBM by the channel kernel,
        // the decode kernel then counts the errors itself (no decisions written, no counting kernel)
        const int64_t step = (int64_t)1 << 17;
        const size_t n = (size_t)d->code->n;
        for (int64_t b0 = 0; b0 < B; b0 += step) {
            const int64_t nb = (B - b0) < step ? (B - b0) : step;
            LDPC_TRY(d->h_in.reserve((size_t)nb * n * sizeof(double)));
            LDPC_TRY(d->h_y0.reserve((size_t)nb * n));
            uint8_t* y = (uint8_t*)d->h_y0.p;
            LDPC_TRY(channel_generate(channel, DT_F64, param, codeword, seed, stream_id, frame0 + (uint64_t)b0, nb, (int32_t)n, d->h_in.p, y, st));
            FusedArgs a{};
            a.priors = d->h_in.p;
            a.y0 = y;
            a.codeword = codeword;
            a.hist_bins = hist_bins;
            a.counters = (unsigned long long*)counters;
            LDPC_TRY(fused_launch(d, a, false, nb, max_iter, flags, st));
        }
        return LDPC_OK;
    }
    const double var = pow(10.0, -param / 10.0);  // src/biawgn.py:10 -- same host arithmetic as channel_generate()
    const double sigma = sqrt(var), k = 2.0 / var;
    FusedArgs a{};
    a.sim_mean = (float)(2 * codeword - 1);
    a.sim_sigma = (float)sigma;
    a.sim_k = (float)k;
    a.sim_mean_d = (double)(2 * codeword - 1);  // k_biawgn<double> computes with exactly these doubles
    a.sim_sigma_d = sigma;
    a.sim_k_d = k;
    a.seed = seed;
    a.frame0 = frame0;
    a.stream = (unsigned)stream_id;
    a.codeword = codeword;
    a.hist_bins = hist_bins;
    a.sim_channel = channel;
    if (channel == CH_BSC || channel == CH_BEC) {
        double t = ceil(param * 4294967296.0 - 0.5);  // same threshold as channel_generate()
        a.bsc_thr = (unsigned long long)(t < 0 ? 0 : t);
        a.bsc_llr = channel == CH_BSC ? (float)(log(1.0 - param) - log(param)) : 0.0f;
        a.bsc_llr_d = channel == CH_BSC ? log(1.0 - param) - log(param) : 0.0;
    }
    a.counters = (unsigned long long*)counters;
    a.rounds = rounds < 1 ? 1 : rounds;
    a.round_stride = round_stride;
    a.counter_stride = 4 + hist_bins;
    return fused_launch(d, a, true, B, max_iter, flags, st);
}

}  // namespace ldpc
