"""The LDS layout planner is host C++ (no GPU needed): build it with AddressSanitizer + UBSan and run it on the golden
(3,6) n=1200 code -- checks memory safety of the annealer / edge colouring and that the plan beats the trivial placement."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ldpc_decoders_amd", "csrc")

HARNESS = r'''
#include "ldpc_layout.hpp"
#include <algorithm>
#include <cstdio>
#include <fstream>
#include <sstream>
using namespace ldpc;
namespace ldpc { void set_error(const char*, ...) {} const char* last_error() { return ""; } }
int main(int argc, char** argv) {
    std::ifstream f(argv[1]); std::string line; Code c; std::vector<std::vector<int>> rows; int n = 0;
    while (std::getline(f, line)) { std::istringstream is(line); std::vector<int> r; int v; while (is >> v) { r.push_back(v - 1); n = std::max(n, v); } if (!r.empty()) rows.push_back(r); }
    c.m = (int)rows.size(); c.n = n; c.row_ptr.assign(c.m + 1, 0); c.col_ptr.assign(c.n + 1, 0);
    for (int i = 0; i < c.m; ++i) { std::sort(rows[i].begin(), rows[i].end()); for (int v : rows[i]) { c.edge_chk.push_back(i); c.edge_var.push_back(v); c.col_ptr[v + 1]++; } c.row_ptr[i + 1] = (int)c.edge_var.size(); }
    c.E = (int64_t)c.edge_var.size(); for (int v = 0; v < c.n; ++v) c.col_ptr[v + 1] += c.col_ptr[v];
    c.col_edge.assign(c.E, 0); std::vector<int> fill(c.col_ptr.begin(), c.col_ptr.end() - 1);
    for (int k = 0; k < c.E; ++k) c.col_edge[fill[c.edge_var[k]]++] = k;
    const int CR = atoi(argv[2]), VR = atoi(argv[3]);
    VarRounds vr; vr.VR = VR; vr.DV = 3; vr.vrx = atoi(argv[4]); vr.dvx = vr.vrx ? 8 : 3;
    if (argc > 8) { vr.nw = atoi(argv[8]); vr.reserved = atoi(argv[9]); }  // per-wave wide rounds / reserved system row
    if (argc > 10) vr.vr2 = atoi(argv[10]);                                 // pair rounds: a wave's last rounds gather two messages
    FusedLayout L; plan_fused_layout(c, 6, CR, vr, 0x1200, 1200000, &L);
    {   // plan store: round trip, and a damaged file is refused
        const uint64_t key = layout_key(c, 6, CR, vr, 1);
        const std::string path = std::string(argv[5]) + "/rt.plan";
        FusedLayout M;
        if (!layout_valid(c, 6, CR, vr, L) || !layout_save(path, key, c, L) || !layout_load(path, key, c, 6, CR, vr, &M)) return 6;
        if (M.chk_slot != L.chk_slot || M.var_slot != L.var_slot || M.edge_pos != L.edge_pos || M.var_pos != L.var_pos ||
            M.extra_cycles_planned != L.extra_cycles_planned) return 7;
        if (layout_load(path, key + 1, c, 6, CR, vr, &M)) return 8;
        FusedLayout D = L; std::swap(D.var_slot[0], D.var_slot[1]); D.var_slot[2] = D.var_slot[3];  // duplicate slot
        if (layout_valid(c, 6, CR, vr, D) || !layout_save(path, key, c, D) || layout_load(path, key, c, 6, CR, vr, &M)) return 9;
    }
    if (argc > 6 && argv[6][0]) {  // a shipped plan for this code / shape: must load and beat the short run
        char name[64]; snprintf(name, sizeof(name), "/%016llx.plan", (unsigned long long)layout_key(c, 6, CR, vr, atoi(argv[7])));
        FusedLayout S;
        if (!layout_load(std::string(argv[6]) + name, layout_key(c, 6, CR, vr, atoi(argv[7])), c, 6, CR, vr, &S)) return 10;
        if (S.extra_cycles_planned >= L.extra_cycles_planned) return 11;
        printf("stored %.0f\n", S.extra_cycles_planned);
    }
    for (int v = 0; v < c.n; ++v)  // placement constraint: more than 3 edges only in the wide rounds
        if (c.col_ptr[v + 1] - c.col_ptr[v] > vr.width(L.var_slot[v] / 64) || !vr.usable_slot(L.var_slot[v])) return 5;
    // the plan must be a permutation of slots and of the positions inside every check
    std::vector<int> seen(CR * 64, 0); for (int s : L.chk_slot) { if (s < 0 || s >= CR * 64 || seen[s]++) return 2; }
    std::vector<int> seenv(VR * 64, 0); for (int s : L.var_slot) { if (s < 0 || s >= VR * 64 || seenv[s]++) return 3; }
    for (int cc = 0; cc < c.m; ++cc) {
        int mask = 0, cnt = 0;
        for (int k = c.row_ptr[cc]; k < c.row_ptr[cc + 1]; ++k) { if (L.edge_pos[k] < 0 || L.edge_pos[k] > 5 || (mask >> L.edge_pos[k]) & 1) return 4; mask |= 1 << L.edge_pos[k]; ++cnt; }
        if (cnt == 6 && mask != 63) return 4;
    }
    printf("%.0f %.0f %.0f\n", L.base_cycles, L.extra_cycles_identity, L.extra_cycles_planned);
    return 0;
}
'''


@pytest.mark.timeout(600)
@pytest.mark.parametrize("code_name,cr,vr,vrx,nw", [("1200_3_6_rand_ldpc_1", 10, 19, 0, 1), ("1200_3_6_rand_ldpc_1", 10, 20, 0, 2),
                                                      ("1200_rho_x5_rand_ldpc_5", 10, 19, 4, 1)])
def test_layout_planner_under_asan(tmp_path, code_name, cr, vr, vrx, nw):
    cxx = shutil.which("g++")
    if cxx is None:
        pytest.skip("no host C++ compiler")
    src = tmp_path / "harness.cpp"
    src.write_text(HARNESS)
    exe = str(tmp_path / "harness")
    # host-only build of the planner: the HIP header is only needed for types in ldpc_common.hpp -> use hipcc's host pass
    hipcc = "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-I", CSRC, "-x", "hip",
           "--offload-arch=gfx950", "--cuda-host-only", str(src), os.path.join(CSRC, "ldpc_layout.hip"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        if "error:" in r.stderr and "ldpc_" in r.stderr:
            pytest.fail("planner harness does not compile: " + r.stderr[-1500:])
        pytest.skip("sanitized host build unavailable here: " + r.stderr[-300:])
    code_file = os.path.join(ROOT, "ldpc_decoders_amd", "data", "codes", code_name + ".txt")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    plans = os.path.join(ROOT, "ldpc_decoders_amd", "plans")
    out = subprocess.run([exe, code_file, str(cr), str(vr), str(vrx), str(tmp_path), plans, str(nw)], capture_output=True, text=True,
                         env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().splitlines()
    stored = float(lines[0].split()[1])  # conflict cycles of the shipped plan (ldpc_decoders_amd/plans), recomputed on load
    base, ident, planned = (float(v) for v in lines[1].split())
    assert base == 2.0 * (cr * 6 + vrx * 8 + (vr - vrx) * 3) and planned < 0.6 * ident and stored < planned


@pytest.mark.timeout(600)
def test_layout_planner_under_asan_pair_rounds(tmp_path):
    # the irregular two-wave shape with three round widths (per wave 2 x 8, 2 x 3, 6 x 2 gathers; last row reserved): every variable in a
    # round at least as wide as its degree, plan-store round trip, and the shipped plan of this code / shape loads and beats a short run
    if shutil.which("g++") is None:
        pytest.skip("no host C++ compiler")
    src = tmp_path / "harness.cpp"
    src.write_text(HARNESS)
    exe = str(tmp_path / "harness")
    cmd = ["/opt/rocm/bin/hipcc", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-I", CSRC, "-x", "hip",
           "--offload-arch=gfx950", "--cuda-host-only", str(src), os.path.join(CSRC, "ldpc_layout.hip"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        if "error:" in r.stderr and "ldpc_" in r.stderr:
            pytest.fail("planner harness does not compile: " + r.stderr[-1500:])
        pytest.skip("sanitized host build unavailable here: " + r.stderr[-300:])
    code_file = os.path.join(ROOT, "ldpc_decoders_amd", "data", "codes", "1200_rho_x5_rand_ldpc_5.txt")
    plans = os.path.join(ROOT, "ldpc_decoders_amd", "plans")
    out = subprocess.run([exe, code_file, "10", "20", "2", str(tmp_path), plans, "2", "2", "1", "6"], capture_output=True, text=True,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"), timeout=500)
    assert out.returncode == 0, out.stdout[-500:] + out.stderr[-2000:]
    lines = out.stdout.strip().splitlines()
    stored = float(lines[0].split()[1])
    base, ident, planned = (float(v) for v in lines[1].split())
    assert base == 2.0 * (10 * 6 + 2 * (2 * 8 + 2 * 3 + 6 * 2)) and stored < planned < 0.6 * ident


@pytest.mark.timeout(900)
def test_layout_planner_under_asan_16_wave_shape(tmp_path):
    # the one-frame-per-CU shape: 80 check rounds, 160 variable rounds over 16 waves (3 wide rounds each), last row reserved,
    # on a generated rate-1/2 irregular n = 10 000 code
    import numpy as np

    from ldpc_decoders_amd import codes

    if shutil.which("g++") is None:
        pytest.skip("no host C++ compiler")
    code = codes.rand_irregular_ldpc(10000, codes.LAMBDA_RHO_X5_HALF_RATE, 6, np.random.RandomState(4))
    code_file = codes.save_parity_mtx(code, "irg10000", str(tmp_path))
    src = tmp_path / "harness.cpp"
    src.write_text(HARNESS.replace("plan_fused_layout(c, 6, CR, vr, 0x1200, 1200000, &L)", "plan_fused_layout(c, 6, CR, vr, 0x1200, 300000, &L)"))
    exe = str(tmp_path / "harness")
    cmd = ["/opt/rocm/bin/hipcc", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-I", CSRC, "-x", "hip",
           "--offload-arch=gfx950", "--cuda-host-only", str(src), os.path.join(CSRC, "ldpc_layout.hip"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        if "error:" in r.stderr and "ldpc_" in r.stderr:
            pytest.fail("planner harness does not compile: " + r.stderr[-1500:])
        pytest.skip("sanitized host build unavailable here: " + r.stderr[-300:])
    out = subprocess.run([exe, code_file, "80", "160", "3", str(tmp_path), "", "16", "16", "1"], capture_output=True, text=True,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"), timeout=800)
    assert out.returncode == 0, out.stdout[-500:] + out.stderr[-2000:]
    base, ident, planned = (float(v) for v in out.stdout.strip().splitlines()[-1].split())
    assert base == 2.0 * (80 * 6 + 16 * (3 * 8 + 7 * 3)) and planned < ident
