// Round-3 microbenchmark: read-only / write-only ceilings and the traffic pattern of a check pass that keeps MARGINALS instead of
// variable-to-check messages:   per check: stream-read dc old c2v lines, gather dc marginal lines (each marginal line is used by dv = 3
// checks of the same tile: the re-reads can come out of the L2 / Infinity Cache), stream-write dc new c2v lines.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/hbm_marg.hip -o build/ab/hbm_marg
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
template <typename V> __device__ __forceinline__ float lane_sum(V v);
template <> __device__ __forceinline__ float lane_sum<float>(float v) { return v; }
template <> __device__ __forceinline__ float lane_sum<f4>(f4 v) { return v.x + v.y + v.z + v.w; }

template <typename V, int K>
__global__ __launch_bounds__(256) void k_read(const V* __restrict__ a, float* __restrict__ out, long nlines, int lpw) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    float acc = 0;
    for (int l = 0; l < lpw; l += K) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const long line = wave * lpw + l + j;
            if (line < nlines) acc += lane_sum(a[line * 64 + lane]);
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}
template <typename V, int K, bool NT>
__global__ __launch_bounds__(256) void k_write(V* __restrict__ a, long nlines, int lpw) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    for (int l = 0; l < lpw; ++l) {
        const long line = wave * lpw + l;
        V v = (V)(float)lane;
        if (line < nlines) {
            if constexpr (NT) __builtin_nontemporal_store(v, a + line * 64 + lane); else a[line * 64 + lane] = v;
        }
    }
}

// check pass on marginals.  c2v: [tiles][E][64] V ; marg: [tiles][n][64] V ; var: [E] (same graph for every tile)
template <typename V, int DC, int UNR, bool NTS, bool NTL>
__global__ __launch_bounds__(256) void k_cn_marg(V* __restrict__ c2v, const V* __restrict__ marg, const int* __restrict__ var, int m, int n, int tiles,
                                                 int chunks, int cpw) {
    const int lane = threadIdx.x & 63;
    const int task = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int tile = task / chunks, chunk = task - tile * chunks;
    if (tile >= tiles) return;
    V* ct = c2v + (long)tile * m * DC * 64 + lane;
    const V* mt = marg + (long)tile * n * 64 + lane;
    const int c_end = min(m, (chunk + 1) * cpw);
    for (int c = chunk * cpw; c < c_end; c += UNR) {
        V o[UNR][DC], g[UNR][DC];
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
            for (int j = 0; j < DC; ++j) {
                const long k = (long)min(c + u, c_end - 1) * DC + j;
                o[u][j] = NTL ? __builtin_nontemporal_load(ct + k * 64) : ct[k * 64];
                g[u][j] = mt[(long)var[k] * 64];
            }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
            for (int j = 0; j < DC; ++j) {
                const long k = (long)(c + u) * DC + j;
                const V r = g[u][j] - o[u][j];
                if (c + u < c_end) {
                    if constexpr (NTS) __builtin_nontemporal_store(r, ct + k * 64); else ct[k * 64] = r;
                }
            }
    }
}
// variable pass on marginals: gather dv c2v lines (read-only), stream prior, stream-write marginal
template <typename V, int DV, int UNR, bool NTS>
__global__ __launch_bounds__(256) void k_vn_marg(const V* __restrict__ c2v, const V* __restrict__ prior, V* __restrict__ marg, const int* __restrict__ col_edge,
                                                 long E, int n, int tiles, int chunks, int vpw) {
    const int lane = threadIdx.x & 63;
    const int task = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int tile = task / chunks, chunk = task - tile * chunks;
    if (tile >= tiles) return;
    const V* ct = c2v + (long)tile * E * 64 + lane;
    const V* pt = prior + (long)tile * n * 64 + lane;
    V* mt = marg + (long)tile * n * 64 + lane;
    const int v_end = min(n, (chunk + 1) * vpw);
    for (int v = chunk * vpw; v < v_end; v += UNR) {
        V c[UNR][DV], p[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int vv = min(v + u, v_end - 1);
            p[u] = pt[(long)vv * 64];
#pragma unroll
            for (int j = 0; j < DV; ++j) c[u][j] = ct[(long)col_edge[vv * DV + j] * 64];
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            V s = p[u];
#pragma unroll
            for (int j = 0; j < DV; ++j) s += c[u][j];
            if (v + u < v_end) {
                if constexpr (NTS) __builtin_nontemporal_store(s, mt + (long)(v + u) * 64); else mt[(long)(v + u) * 64] = s;
            }
        }
    }
}

static hipEvent_t e0, e1;
template <typename F> static double timed(F launch) {
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        for (int i = 0; i < 3; ++i) launch();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    return ms / 3;
}

template <typename V, int UNR, bool NTS, bool NTL>
static void run_marg(const char* tag, int n, int frames, int cpw) {
    const int DC = 6, DV = 3, m = n / 2;
    const long E = (long)m * DC;
    const int W = sizeof(V) / 4;
    const int tiles = frames / (64 * W);
    std::vector<int> var(E), col_edge(E);
    // random (3,6)-regular socket matching
    std::vector<int> sock(E);
    for (long i = 0; i < E; ++i) sock[i] = (int)(i / DV);
    std::mt19937 rng(7);
    std::shuffle(sock.begin(), sock.end(), rng);
    std::vector<int> fill(n, 0);
    for (long k = 0; k < E; ++k) {
        var[k] = sock[k];
        col_edge[(long)sock[k] * DV + fill[sock[k]]++] = (int)k;
    }
    int *dvar, *dcol;
    (void)hipMalloc(&dvar, E * 4);
    (void)hipMalloc(&dcol, E * 4);
    (void)hipMemcpy(dvar, var.data(), E * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dcol, col_edge.data(), E * 4, hipMemcpyHostToDevice);
    V *c2v, *marg, *prior;
    const size_t cb = (size_t)tiles * E * 64 * sizeof(V), mb = (size_t)tiles * n * 64 * sizeof(V);
    (void)hipMalloc(&c2v, cb);
    (void)hipMalloc(&marg, mb);
    (void)hipMalloc(&prior, mb);
    (void)hipMemset(c2v, 0, cb);
    (void)hipMemset(marg, 0, mb);
    (void)hipMemset(prior, 0, mb);
    const int chunks = (m + cpw - 1) / cpw;
    const long tasks = (long)tiles * chunks;
    const double ms_cn = timed([&] { hipLaunchKernelGGL((k_cn_marg<V, DC, UNR, NTS, NTL>), dim3((tasks + 3) / 4), dim3(256), 0, 0, c2v, marg, dvar, m, n, tiles, chunks, cpw); });
    const int vpw = cpw * 2, vchunks = (n + vpw - 1) / vpw;
    const long vtasks = (long)tiles * vchunks;
    const double ms_vn = timed([&] { hipLaunchKernelGGL((k_vn_marg<V, DV, 4 / (sizeof(V) / 4 > 1 ? 2 : 1), NTS>), dim3((vtasks + 3) / 4), dim3(256), 0, 0, c2v, prior, marg, dcol, E, n, tiles, vchunks, vpw); });
    const double alg = (double)frames * 4.0 * (4.0 * E + n);  // SURVEY 8(d) bytes per sweep
    const double cn_min = (double)frames * 4.0 * (2.0 * E + n), cn_max = (double)frames * 4.0 * 3.0 * E;
    const double vn_b = (double)frames * 4.0 * (E + 2.0 * n);
    printf("%-34s n=%d frames=%d line=%zu B cpw=%3d unr=%d nts=%d ntl=%d | check %.3f ms (%.2f..%.2f TB/s actual)  variable %.3f ms (%.2f TB/s actual) | sweep %.3f ms = %.2f TB/s of s(4E+n)\n",
           tag, n, frames, 64 * sizeof(V), cpw, UNR, (int)NTS, (int)NTL, ms_cn, cn_min / ms_cn / 1e9, cn_max / ms_cn / 1e9, ms_vn, vn_b / ms_vn / 1e9, ms_cn + ms_vn,
           alg / (ms_cn + ms_vn) / 1e9);
    fflush(stdout);
    (void)hipFree(c2v); (void)hipFree(marg); (void)hipFree(prior); (void)hipFree(dvar); (void)hipFree(dcol);
}

int main() {
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const long bytes = 1L << 32;
    void* a;
    float* out;
    (void)hipMalloc(&a, bytes);
    (void)hipMalloc(&out, 64);
    (void)hipMemset(a, 0, bytes);
    for (int lpw : {24, 96}) {
        long nl = bytes / 1024, waves = (nl + lpw - 1) / lpw;
        double ms = timed([&] { hipLaunchKernelGGL((k_read<f4, 6>), dim3((waves + 3) / 4), dim3(256), 0, 0, (const f4*)a, out, nl, lpw); });
        printf("read-only  16 B/lane lpw=%d   %.3f ms  %.2f TB/s\n", lpw, ms, bytes / ms / 1e9);
        ms = timed([&] { hipLaunchKernelGGL((k_write<f4, 6, false>), dim3((waves + 3) / 4), dim3(256), 0, 0, (f4*)a, nl, lpw); });
        printf("write-only 16 B/lane lpw=%d   %.3f ms  %.2f TB/s\n", lpw, ms, bytes / ms / 1e9);
        ms = timed([&] { hipLaunchKernelGGL((k_write<f4, 6, true>), dim3((waves + 3) / 4), dim3(256), 0, 0, (f4*)a, nl, lpw); });
        printf("write-only 16 B/lane nt lpw=%d %.3f ms  %.2f TB/s\n", lpw, ms, bytes / ms / 1e9);
        nl = bytes / 256, waves = (nl + lpw * 4 - 1) / (lpw * 4);
        ms = timed([&] { hipLaunchKernelGGL((k_read<float, 12>), dim3((waves + 3) / 4), dim3(256), 0, 0, (const float*)a, out, nl, lpw * 4); });
        printf("read-only   4 B/lane lpw=%d   %.3f ms  %.2f TB/s\n", lpw * 4, ms, bytes / ms / 1e9);
    }
    (void)hipFree(a);
    fflush(stdout);
    // n = 64 800, 8 192 frames (6.4 GB of c2v) ; n = 1200, 65 536 frames
    for (int cpw : {16, 64}) {
        run_marg<float, 2, true, false>("marg scheme 256 B lines", 64800, 8192, cpw);
        run_marg<float, 2, true, true>("marg scheme 256 B lines", 64800, 8192, cpw);
        run_marg<f4, 1, true, false>("marg scheme 1 KiB lines", 64800, 8192, cpw);
        run_marg<f4, 1, true, true>("marg scheme 1 KiB lines", 64800, 8192, cpw);
        run_marg<f4, 1, false, false>("marg scheme 1 KiB lines", 64800, 8192, cpw);
        run_marg<f4, 2, true, true>("marg scheme 1 KiB lines", 64800, 8192, cpw);
    }
    run_marg<float, 2, true, false>("marg scheme 256 B lines", 1200, 65536, 16);
    run_marg<f4, 1, true, false>("marg scheme 1 KiB lines", 1200, 65536, 16);
    run_marg<f4, 1, true, true>("marg scheme 1 KiB lines", 1200, 65536, 16);
    run_marg<float, 2, true, false>("marg scheme 256 B lines", 10000, 32768, 16);
    run_marg<f4, 1, true, true>("marg scheme 1 KiB lines", 10000, 32768, 16);
    return 0;
}
