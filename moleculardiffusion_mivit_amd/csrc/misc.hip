// Error plumbing, small elementwise / reduction kernels and library-level C-ABI entry points.
#include "common.h"
#include <stdarg.h>
#include <algorithm>

static thread_local char g_err[512] = "";

void mivit_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *mivit_last_error(void) { return g_err; }
extern "C" int mivit_abi_version(void) { return MIVIT_ABI_VERSION; }
extern "C" int mivit_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ------------------------------------------------------------------------------------------------
// kernel timing
// ------------------------------------------------------------------------------------------------
#include <mutex>
#include <vector>
namespace {
struct ProfState {
    std::mutex mu;
    uint64_t mask = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[MIVIT_PROF_NUM_TAGS];
    std::vector<hipEvent_t> pool;
    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    }
};
ProfState g_prof;
thread_local int t_tag = MIVIT_PROF_OP;
thread_local int t_override = -1;             // >= 0: the engine pins the tag across helper calls that set their own
thread_local hipEvent_t t_start = nullptr;
const char *kTagNames[MIVIT_PROF_NUM_TAGS] = {"embed_fwd", "embed_wgrad", "linear_fwd", "linear_dgrad", "linear_wgrad",
                                              "attn_fwd", "attn_bwd", "ln_fwd", "ln_bwd", "op", "attn_block_fwd", "mlp_block_fwd",
                                              "mlp_block_bwd", "attn_out_bwd", "attn_core_bwd", "qkv_wgrad", "qkv_dgrad", "qkv_bwd"};
}  // namespace

void prof_set_tag(int tag) {
    if (t_override >= 0) tag = t_override;
    t_tag = (tag >= 0 && tag < MIVIT_PROF_NUM_TAGS) ? tag : MIVIT_PROF_OP;
}
void prof_pin_tag(int tag) { t_override = tag; if (tag >= 0) t_tag = tag; }
bool prof_begin(hipStream_t s) {
    if (!((g_prof.mask >> t_tag) & 1ull)) return false;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    t_start = g_prof.get();
    if (!t_start) return false;
    (void)hipEventRecord(t_start, s);
    return true;
}
void prof_end(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    hipEvent_t stop = g_prof.get();
    if (!stop) { g_prof.pool.push_back(t_start); return; }
    (void)hipEventRecord(stop, s);
    g_prof.ev[t_tag].push_back({t_start, stop});
}
extern "C" int mivit_profile_enable(uint64_t tag_mask) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.mask = tag_mask;
    return 0;
}
extern "C" int mivit_profile_collect(int tag, double *total_ms, int *count) {
    MIVIT_CHECK(tag >= 0 && tag < MIVIT_PROF_NUM_TAGS, "profile_collect: bad tag %d", tag);
    std::lock_guard<std::mutex> lk(g_prof.mu);
    double tot = 0.0;
    int n = 0;
    for (auto &pr : g_prof.ev[tag]) {
        float ms = 0.f;
        if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
            tot += ms;
            ++n;
        }
        g_prof.pool.push_back(pr.first);
        g_prof.pool.push_back(pr.second);
    }
    g_prof.ev[tag].clear();
    if (total_ms) *total_ms = tot;
    if (count) *count = n;
    return 0;
}
extern "C" const char *mivit_profile_tag_name(int tag) {
    return (tag >= 0 && tag < MIVIT_PROF_NUM_TAGS) ? kTagNames[tag] : "?";
}

namespace {

// out[i] (+)= sum_p part[p * n + i].  Block = 32 element-threads (one float4 each) x 8 part-lanes; every part-lane
// sums its parts in a fixed order, the 8 lanes are combined through LDS in a fixed order: deterministic.
__device__ __forceinline__ void slab_reduce_block(const float *part, int nparts, int64_t n, int64_t ps, float *out, int accumulate,
                                                  int vec_ok, int block) {      // ps = distance between parts
    __shared__ float4 red[8][32];
    const int ex = threadIdx.x & 31, pl = threadIdx.x >> 5;
    const int64_t e0 = ((int64_t)block * 32 + ex) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e0 < n) {
        if (vec_ok) {
            int p = pl;
            // 8, then 4 independent 16-byte loads in flight per thread (each iteration is one memory round trip: 256 slabs over 8
            // part-lanes were 8 round trips at 4 loads per iteration)
            for (; p + 56 < nparts; p += 64) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4 *>(part + (int64_t)(p + 8 * u) * ps + e0);
                acc.x += ((v[0].x + v[1].x) + (v[2].x + v[3].x)) + ((v[4].x + v[5].x) + (v[6].x + v[7].x));
                acc.y += ((v[0].y + v[1].y) + (v[2].y + v[3].y)) + ((v[4].y + v[5].y) + (v[6].y + v[7].y));
                acc.z += ((v[0].z + v[1].z) + (v[2].z + v[3].z)) + ((v[4].z + v[5].z) + (v[6].z + v[7].z));
                acc.w += ((v[0].w + v[1].w) + (v[2].w + v[3].w)) + ((v[4].w + v[5].w) + (v[6].w + v[7].w));
            }
            for (; p + 24 < nparts; p += 32) {
                const float4 a = *reinterpret_cast<const float4 *>(part + (int64_t)p * ps + e0);
                const float4 b = *reinterpret_cast<const float4 *>(part + (int64_t)(p + 8) * ps + e0);
                const float4 c = *reinterpret_cast<const float4 *>(part + (int64_t)(p + 16) * ps + e0);
                const float4 d = *reinterpret_cast<const float4 *>(part + (int64_t)(p + 24) * ps + e0);
                acc.x += (a.x + b.x) + (c.x + d.x); acc.y += (a.y + b.y) + (c.y + d.y);
                acc.z += (a.z + b.z) + (c.z + d.z); acc.w += (a.w + b.w) + (c.w + d.w);
            }
            for (; p < nparts; p += 8) {
                const float4 a = *reinterpret_cast<const float4 *>(part + (int64_t)p * ps + e0);
                acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
            }
        } else {
            for (int p = pl; p < nparts; p += 8) {
                const float *q = part + (int64_t)p * ps + e0;
                acc.x += q[0];
                if (e0 + 1 < n) acc.y += q[1];
                if (e0 + 2 < n) acc.z += q[2];
                if (e0 + 3 < n) acc.w += q[3];
            }
        }
    }
    red[pl][ex] = acc;
    __syncthreads();
    if (pl == 0 && e0 < n) {
        float4 t = red[0][ex];
#pragma unroll
        for (int y = 1; y < 8; ++y) { t.x += red[y][ex].x; t.y += red[y][ex].y; t.z += red[y][ex].z; t.w += red[y][ex].w; }
        float *o = out + e0;
        const float v[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (e0 + i < n) o[i] = accumulate ? o[i] + v[i] : v[i];
    }
}
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float *part, int nparts, int64_t n, int64_t ps, float *out,
                                                          int accumulate, int vec_ok) {
    slab_reduce_block(part, nparts, n, ps, out, accumulate, vec_ok, blockIdx.x);
}
// several queued reductions in ONE launch (slab_defer_begin / slab_defer_flush below): block b belongs to the job whose block
// range contains it; the arithmetic of a job is slab_reduce_kernel's
constexpr int MAX_SLAB_JOBS = 8;
struct SlabJobs {
    const float *part[MAX_SLAB_JOBS]; float *out[MAX_SLAB_JOBS];
    int nparts[MAX_SLAB_JOBS], n[MAX_SLAB_JOBS], stride[MAX_SLAB_JOBS], vec_ok[MAX_SLAB_JOBS], blk_end[MAX_SLAB_JOBS];
    int njobs;
};
__global__ __launch_bounds__(256) void slab_reduce_multi_kernel(const SlabJobs j) {
    int k = 0;
    while (k + 1 < j.njobs && (int)blockIdx.x >= j.blk_end[k]) ++k;          // (uniform per block)
    const int b0 = k ? j.blk_end[k - 1] : 0;
    slab_reduce_block(j.part[k], j.nparts[k], j.n[k], j.stride[k], j.out[k], 0, j.vec_ok[k], (int)blockIdx.x - b0);
}

template <typename T>
__global__ __launch_bounds__(256) void reg_token_kernel(T *tokens, int B, int S, int E, const float *reg,
                                                        const T *add, const float *pos) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * E) return;
    const int b = (int)(i / E), e = (int)(i % E);
    float v = reg[e];
    if (add) v += to_f32(add[(int64_t)b * E + e]);
    if (pos) v += pos[e];
    tokens[(int64_t)b * S * E + e] = from_f32<T>(v);
}

template <typename T>
__global__ __launch_bounds__(256) void mean_pool_fwd_kernel(const T *x, int B, int S, int E, T *out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * E) return;
    const int b = (int)(i / E), e = (int)(i % E);
    float acc = 0.f;
    int s = 0;
    for (; s + 8 <= S; s += 8) {          // eight loads in flight (same summation order as the rolled loop)
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = to_f32(x[((int64_t)b * S + s + u) * E + e]);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; s < S; ++s) acc += to_f32(x[((int64_t)b * S + s) * E + e]);
    out[i] = from_f32<T>(acc / (float)S);
}

template <typename T>
__global__ __launch_bounds__(256) void mean_pool_bwd_kernel(const T *dout, int B, int S, int E, T *dx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * S * E) return;
    const int e = (int)(i % E);
    const int b = (int)(i / ((int64_t)S * E));
    dx[i] = from_f32<T>(to_f32(dout[(int64_t)b * E + e]) / (float)S);
}

// part[chunk][r][e] = sum over b in chunk of x[b, s0 + r, e]
template <typename T>
__global__ __launch_bounds__(256) void batch_colsum_kernel(const T *x, int B, int S, int E, int s0, int rows,
                                                           int bchunk, float *part) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)rows * E) return;
    const int r = (int)(i / E), e = (int)(i % E);
    const int bb = blockIdx.y * bchunk, be = min(B, bb + bchunk);
    float acc = 0.f;
    // eight loads in flight (the rolled loop was one memory round trip per batch element: 128 in a row at B = 16384)
    int b = bb;
    for (; b + 8 <= be; b += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = to_f32(x[((int64_t)(b + u) * S + s0 + r) * E + e]);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; b < be; ++b) acc += to_f32(x[((int64_t)b * S + s0 + r) * E + e]);
    part[(int64_t)blockIdx.y * rows * E + i] = acc;
}

template <typename TS, typename TD>
__global__ __launch_bounds__(256) void convert_kernel(const TS *src, int64_t lds_, TD *dst, int64_t ldd, int rows,
                                                      int cols, int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)rows * cols) return;
    const int64_t r = i / cols, c = i % cols;
    float v = to_f32(src[r * lds_ + c]);
    if (accumulate) v += to_f32(dst[r * ldd + c]);
    dst[r * ldd + c] = from_f32<TD>(v);
}

int batch_chunks(int B) {
    int c = ceil_div(B, 64);
    return c < 1 ? 1 : (c > 128 ? 128 : c);
}

}  // namespace

// up to three independent reductions of equally shaped slabs in ONE launch (LayerNorm backward: dgamma, dbeta, bias
// gradient): blockIdx.y selects the (slab, output) pair
struct Reduce3 {
    const float *part[3];
    float *out[3];
};
__global__ __launch_bounds__(1024) void slab_reduce3_kernel(const Reduce3 r, int nparts, int n, int accumulate) {
    __shared__ float red[32][33];
    const float *part = r.part[blockIdx.y];
    float *out = r.out[blockIdx.y];
    const int ex = threadIdx.x & 31, pl = threadIdx.x >> 5;       // 32 columns x 32 part lanes (fixed summation order)
    const int e = blockIdx.x * 32 + ex;
    float acc = 0.f;
    if (e < n) {
        // eight loads in flight per thread (a plain `acc += part[...]` loop is one L2 round trip per part: 64 serialised round
        // trips at the 2048 partials of a LayerNorm backward -- 19 us per launch, 14 launches per step of the 64-wide models);
        // the summation order stays fixed: ((p0 + p1) + ...) per batch, batches in order
        int p = pl;
        for (; p + 7 * 32 < nparts; p += 8 * 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(p + 32 * u) * n + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; p < nparts; p += 32) acc += part[(int64_t)p * n + e];
    }
    red[pl][ex] = acc;
    __syncthreads();
    if (pl == 0 && e < n) {
        float t = red[0][ex];
#pragma unroll
        for (int y = 1; y < 32; ++y) t += red[y][ex];
        out[e] = accumulate ? out[e] + t : t;
    }
}

// ------------------------------------------------------------------------------------------------
// hipGraph cache
// ------------------------------------------------------------------------------------------------
namespace {
struct GraphEntry {
    std::vector<uint64_t> key;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    uint64_t last_use = 0;
};
struct GraphState {
    std::mutex mu;
    std::vector<GraphEntry> entries;
    hipStream_t capture_stream[16] = {};
    uint64_t tick = 0;
    int mode = -1, failures = 0;
    uint64_t replays = 0, captures = 0;
};
GraphState g_graph;
size_t graph_cap() { static const size_t c = [] { const char *e = getenv("MIVIT_GRAPH_CAP"); return e ? (size_t)atoi(e) : (size_t)32; }(); return c; }

void graph_drop(GraphEntry &e) {
    if (e.exec) (void)hipGraphExecDestroy(e.exec);
    if (e.graph) (void)hipGraphDestroy(e.graph);
    e.exec = nullptr; e.graph = nullptr;
}
}  // namespace

int graph_run(const uint64_t *key, int nkey, hipStream_t s, const std::function<int(hipStream_t)> &body) {
    GraphState &g = g_graph;
    std::unique_lock<std::mutex> lk(g.mu);
    if (g.mode < 0) {
        const char *e = getenv("MIVIT_GRAPHS");
        g.mode = (e && e[0] == '0') ? 0 : 1;
    }
    if (!g.mode || g.failures >= 3 || g_prof.mask != 0) { lk.unlock(); return body(s); }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { lk.unlock(); return body(s); }
    std::vector<uint64_t> k(key, key + nkey);
    k.push_back((uint64_t)dev);
    GraphEntry *hit = nullptr;
    for (auto &e : g.entries)
        if (e.key == k) { hit = &e; break; }
    ++g.tick;
    if (hit && hit->exec) {
        hit->last_use = g.tick; ++g.replays;
        MIVIT_HIP(hipGraphLaunch(hit->exec, s));
        return 0;
    }
    if (!hit) {                                   // first sighting: remember, run directly
        if (g.entries.size() >= graph_cap()) {
            size_t lru = 0;
            for (size_t i = 1; i < g.entries.size(); ++i)
                if (g.entries[i].last_use < g.entries[lru].last_use) lru = i;
            graph_drop(g.entries[lru]);
            g.entries.erase(g.entries.begin() + lru);
        }
        GraphEntry e; e.key = k; e.last_use = g.tick;
        g.entries.push_back(e);
        lk.unlock();
        return body(s);
    }
    // second sighting: capture on the internal stream (the caller's stream may be the legacy default stream)
    hit->last_use = g.tick;
    if (!g.capture_stream[dev] && hipStreamCreateWithFlags(&g.capture_stream[dev], hipStreamNonBlocking) != hipSuccess) {
        ++g.failures; (void)hipGetLastError(); lk.unlock(); return body(s);
    }
    hipStream_t cs = g.capture_stream[dev];
    if (hipStreamBeginCapture(cs, hipStreamCaptureModeRelaxed) != hipSuccess) {
        ++g.failures; (void)hipGetLastError(); lk.unlock(); return body(s);
    }
    const int rc = body(cs);
    hipGraph_t graph = nullptr;
    const hipError_t e1 = hipStreamEndCapture(cs, &graph);
    if (rc != 0) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    hipGraphExec_t exec = nullptr;
    if (e1 != hipSuccess || !graph || hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
        if (graph) (void)hipGraphDestroy(graph);
        ++g.failures; (void)hipGetLastError(); lk.unlock();
        return body(s);
    }
    hit->graph = graph; hit->exec = exec; ++g.captures;
    MIVIT_HIP(hipGraphLaunch(exec, s));
    return 0;
}

extern "C" void mivit_graph_stats(uint64_t *replays, uint64_t *captures, int *failures) {
    std::lock_guard<std::mutex> lk(g_graph.mu);
    if (replays) *replays = g_graph.replays;
    if (captures) *captures = g_graph.captures;
    if (failures) *failures = g_graph.failures;
}

int launch_slab_reduce3(const float *p0, float *o0, const float *p1, float *o1, const float *p2, float *o2, int nparts,
                        int n, int accumulate, hipStream_t s) {
    Reduce3 r = {{p0, p1, p2}, {o0, o1, o2}};
    const int cnt = o2 ? 3 : (o1 ? 2 : 1);
    hipLaunchKernelGGL(slab_reduce3_kernel, dim3((n + 31) / 32, cnt), dim3(1024), 0, s, r, nparts, n, accumulate);
    MIVIT_LAUNCH_CHECK();
    return 0;
}

int launch_slab_reduce(const float *part, int nparts, int64_t n, float *out, int accumulate, hipStream_t s) {
    const int blocks = (int)((n + 127) / 128);
    const int vec_ok = (n % 4 == 0) && ((reinterpret_cast<uintptr_t>(part) & 15) == 0);
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, s, part, nparts, n, n, out, accumulate, vec_ok);
    MIVIT_LAUNCH_CHECK();
    return 0;
}
// Deferred reductions: the fused backward blocks of one encoder layer (mlp_block_bwd, attn_out_bwd, qkv_bwd) each end in a slab
// reduction whose result nothing in the layer's backward reads -- three dependent 5-7 us launches per layer on the critical path
// (24 of the ~85 launches of a 64-wide model's step).  Between slab_defer_begin() and slab_defer_flush(s) the strided
// reductions are queued (their slabs must stay untouched until the flush: the engine gives each block its own workspace region)
// and run as ONE launch.
static thread_local bool t_slab_defer = false;
static thread_local SlabJobs t_slab_jobs = {};
void slab_defer_begin() { t_slab_defer = true; t_slab_jobs.njobs = 0; }
void slab_defer_cancel() { t_slab_defer = false; t_slab_jobs.njobs = 0; }          // (a call that failed between begin and flush)
int slab_defer_flush(hipStream_t s) {
    t_slab_defer = false;
    const int nj = t_slab_jobs.njobs;
    t_slab_jobs.njobs = 0;
    if (nj == 0) return 0;
    SlabJobs j = t_slab_jobs;
    j.njobs = nj;
    hipLaunchKernelGGL(slab_reduce_multi_kernel, dim3(j.blk_end[nj - 1]), dim3(256), 0, s, j);
    MIVIT_LAUNCH_CHECK();
    return 0;
}
// the same for parts that are `stride` floats apart (one field of a per-workgroup record)
int launch_slab_reduce_strided(const float *part, int nparts, int64_t stride, int64_t n, float *out, hipStream_t s) {
    const int blocks = (int)((n + 127) / 128);
    const int vec_ok = (n % 4 == 0) && (stride % 4 == 0) && ((reinterpret_cast<uintptr_t>(part) & 15) == 0);
    if (t_slab_defer && t_slab_jobs.njobs < MAX_SLAB_JOBS && n < (1ll << 31) && stride < (1ll << 31)) {
        SlabJobs &q = t_slab_jobs;
        const int k = q.njobs++;
        q.part[k] = part; q.out[k] = out; q.nparts[k] = nparts; q.n[k] = (int)n; q.stride[k] = (int)stride; q.vec_ok[k] = vec_ok;
        q.blk_end[k] = (k ? q.blk_end[k - 1] : 0) + blocks;
        return 0;
    }
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, s, part, nparts, n, stride, out, 0, vec_ok);
    MIVIT_LAUNCH_CHECK();
    return 0;
}

int launch_reg_token_fill(int dtype, void *tokens, int B, int S, int E, const float *reg, const void *add,
                          const float *pos, hipStream_t s) {
    const int blocks = (int)(((int64_t)B * E + 255) / 256);
    if (dtype == MIVIT_F32)
        hipLaunchKernelGGL(reg_token_kernel<float>, dim3(blocks), dim3(256), 0, s, static_cast<float *>(tokens), B, S, E,
                           reg, static_cast<const float *>(add), pos);
    else if (dtype == MIVIT_BF16)
        hipLaunchKernelGGL(reg_token_kernel<bf16>, dim3(blocks), dim3(256), 0, s, static_cast<bf16 *>(tokens), B, S, E,
                           reg, static_cast<const bf16 *>(add), pos);
    else
        hipLaunchKernelGGL(reg_token_kernel<f16>, dim3(blocks), dim3(256), 0, s, static_cast<f16 *>(tokens), B, S, E,
                           reg, static_cast<const f16 *>(add), pos);
    MIVIT_LAUNCH_CHECK();
    return 0;
}

int launch_mean_pool_fwd(int dtype, const void *x, int B, int S, int E, void *out, hipStream_t s) {
    const int blocks = (int)(((int64_t)B * E + 255) / 256);
    if (dtype == MIVIT_F32)
        hipLaunchKernelGGL(mean_pool_fwd_kernel<float>, dim3(blocks), dim3(256), 0, s, static_cast<const float *>(x), B,
                           S, E, static_cast<float *>(out));
    else if (dtype == MIVIT_BF16)
        hipLaunchKernelGGL(mean_pool_fwd_kernel<bf16>, dim3(blocks), dim3(256), 0, s, static_cast<const bf16 *>(x), B, S,
                           E, static_cast<bf16 *>(out));
    else
        hipLaunchKernelGGL(mean_pool_fwd_kernel<f16>, dim3(blocks), dim3(256), 0, s, static_cast<const f16 *>(x), B, S,
                           E, static_cast<f16 *>(out));
    MIVIT_LAUNCH_CHECK();
    return 0;
}

int launch_mean_pool_bwd(int dtype, const void *dout, int B, int S, int E, void *dx, hipStream_t s) {
    const int blocks = (int)(((int64_t)B * S * E + 255) / 256);
    if (dtype == MIVIT_F32)
        hipLaunchKernelGGL(mean_pool_bwd_kernel<float>, dim3(blocks), dim3(256), 0, s, static_cast<const float *>(dout),
                           B, S, E, static_cast<float *>(dx));
    else if (dtype == MIVIT_BF16)
        hipLaunchKernelGGL(mean_pool_bwd_kernel<bf16>, dim3(blocks), dim3(256), 0, s, static_cast<const bf16 *>(dout), B,
                           S, E, static_cast<bf16 *>(dx));
    else
        hipLaunchKernelGGL(mean_pool_bwd_kernel<f16>, dim3(blocks), dim3(256), 0, s, static_cast<const f16 *>(dout), B,
                           S, E, static_cast<f16 *>(dx));
    MIVIT_LAUNCH_CHECK();
    return 0;
}

size_t batch_colsum_ws_bytes(int B, int rows, int E) {
    return align_up((size_t)batch_chunks(B) * rows * E * sizeof(float), 256);
}

int launch_batch_colsum(int dtype, const void *x, int B, int S, int E, int s0, int rows, float *out, void *ws,
                        size_t ws_bytes, hipStream_t s) {
    MIVIT_CHECK(ws_bytes >= batch_colsum_ws_bytes(B, rows, E), "batch_colsum: workspace too small");
    const int chunks = batch_chunks(B);
    const int bchunk = ceil_div(B, chunks);
    dim3 grid((unsigned)(((int64_t)rows * E + 255) / 256), chunks);
    float *part = static_cast<float *>(ws);
    if (dtype == MIVIT_F32)
        hipLaunchKernelGGL(batch_colsum_kernel<float>, grid, dim3(256), 0, s, static_cast<const float *>(x), B, S, E, s0,
                           rows, bchunk, part);
    else if (dtype == MIVIT_BF16)
        hipLaunchKernelGGL(batch_colsum_kernel<bf16>, grid, dim3(256), 0, s, static_cast<const bf16 *>(x), B, S, E, s0,
                           rows, bchunk, part);
    else
        hipLaunchKernelGGL(batch_colsum_kernel<f16>, grid, dim3(256), 0, s, static_cast<const f16 *>(x), B, S, E, s0,
                           rows, bchunk, part);
    MIVIT_LAUNCH_CHECK();
    return launch_slab_reduce(part, chunks, (int64_t)rows * E, out, 0, s);
}

template <typename H>
static void convert_h(int src_is_f32, const void *src, int64_t lds_, int dst_is_f32, void *dst, int64_t ldd, int rows, int cols,
                      int accumulate, int blocks, hipStream_t s) {
    if (src_is_f32)
        hipLaunchKernelGGL((convert_kernel<float, H>), dim3(blocks), dim3(256), 0, s, static_cast<const float *>(src),
                           lds_, static_cast<H *>(dst), ldd, rows, cols, accumulate);
    else if (dst_is_f32)
        hipLaunchKernelGGL((convert_kernel<H, float>), dim3(blocks), dim3(256), 0, s, static_cast<const H *>(src),
                           lds_, static_cast<float *>(dst), ldd, rows, cols, accumulate);
    else
        hipLaunchKernelGGL((convert_kernel<H, H>), dim3(blocks), dim3(256), 0, s, static_cast<const H *>(src),
                           lds_, static_cast<H *>(dst), ldd, rows, cols, accumulate);
}

int launch_convert(int src_is_f32, const void *src, int64_t lds_, int dst_is_f32, void *dst, int64_t ldd, int rows,
                   int cols, int accumulate, hipStream_t s, int dtype16) {
    if (rows <= 0 || cols <= 0) return 0;
    const int blocks = (int)(((int64_t)rows * cols + 255) / 256);
    if (src_is_f32 && dst_is_f32)
        hipLaunchKernelGGL((convert_kernel<float, float>), dim3(blocks), dim3(256), 0, s, static_cast<const float *>(src),
                           lds_, static_cast<float *>(dst), ldd, rows, cols, accumulate);
    else if (dtype16 == MIVIT_F16)
        convert_h<f16>(src_is_f32, src, lds_, dst_is_f32, dst, ldd, rows, cols, accumulate, blocks, s);
    else
        convert_h<bf16>(src_is_f32, src, lds_, dst_is_f32, dst, ldd, rows, cols, accumulate, blocks, s);
    MIVIT_LAUNCH_CHECK();
    return 0;
}

// A kernel, not hipMemsetAsync: this runs inside stream captures, and replayed MEMSET nodes were observed to leave the
// buffer untouched on some replays (ROCm 7.2): training trajectories drifted from step ~30 on with graphs enabled while
// the same launches outside a graph matched the reference (tests/test_training_parity_gpu.py).
namespace {
__global__ __launch_bounds__(256) void fill_zero_kernel(uint4 *p16, size_t n16, unsigned char *tail, size_t ntail) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) p16[i] = make_uint4(0u, 0u, 0u, 0u);
    if (i < ntail) tail[i] = 0;
}
}  // namespace
int launch_fill_zero(void *p, size_t bytes, hipStream_t s) {
    if (!bytes) return 0;
    unsigned char *b = static_cast<unsigned char *>(p);
    const size_t head = (16 - (reinterpret_cast<uintptr_t>(b) & 15)) & 15;          // bytes before the first 16-byte boundary
    if (head >= bytes) {
        hipLaunchKernelGGL(fill_zero_kernel, dim3(1), dim3(256), 0, s, nullptr, (size_t)0, b, bytes);
    } else {
        if (head) hipLaunchKernelGGL(fill_zero_kernel, dim3(1), dim3(256), 0, s, nullptr, (size_t)0, b, head);
        const size_t n16 = (bytes - head) / 16, ntail = (bytes - head) % 16;
        hipLaunchKernelGGL(fill_zero_kernel, dim3((unsigned)std::max<size_t>(1, (n16 + 255) / 256)), dim3(256), 0, s,
                           reinterpret_cast<uint4 *>(b + head), n16, b + head + n16 * 16, ntail);
    }
    MIVIT_LAUNCH_CHECK();
    return 0;
}

// Weight gradient of a Linear whose input was handed over NORMALISED (x = gamma * n + beta applied by the consumer while
// loading, fused layer blocks): the weight-gradient kernel contracted dy with n, so
//   dW = dy^T (n diag(gamma) + 1 beta^T) = (dy^T n) diag(gamma) + db beta^T,   db = column sums of dy
namespace {
__global__ __launch_bounds__(256) void affine_fixup_kernel(float *dW, const float *db, const float *gamma, const float *beta,
                                                           int N, int K) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * K) return;
    const int n = i / K, k = i - n * K;
    dW[i] = dW[i] * gamma[k] + db[n] * beta[k];
}
}  // namespace
int launch_affine_fixup(float *dW, const float *db, const float *gamma, const float *beta, int N, int K, hipStream_t s) {
    MIVIT_CHECK(dW && db && gamma && beta && N > 0 && K > 0, "affine_fixup: null pointer / empty problem");
    hipLaunchKernelGGL(affine_fixup_kernel, dim3(ceil_div(N * K, 256)), dim3(256), 0, s, dW, db, gamma, beta, N, K);
    MIVIT_LAUNCH_CHECK();
    return 0;
}
