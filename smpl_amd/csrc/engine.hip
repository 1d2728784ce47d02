// smpl_amd/csrc/engine.hip -- host side of the C-ABI (include/smpl_amd.h): handles that own HBM,
// kernel launches, the ManipLattice state table with commit-order ids, the speculative successor
// cache and the ARA* caller.  There is no CPU fallback: every query below runs the gfx950 kernels
// and fails with SMPLX_E_HIP when no GPU is present.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/smpl_amd.h"
#include "det_math.h"
#include "device_types.h"
#include "kernels.h"
#include "model_compile.h"
#include "specialize.h"
#include "test_hooks.h"
#ifdef SMPLX_BFS_TRACE
extern __device__ long long g_bfs_trace[16];
#endif

namespace {

thread_local std::string g_error;

int set_error(int code, const std::string& msg)
{
    g_error = msg;
    return code;
}

// input guard of the entry points that take joint values from the caller: a non-finite or absurd value would make the
// limit folding of KDLRobotModel::checkJointLimits (a -= 2*pi until in range) spin forever on the device
bool sane_values(const double* q, size_t n)
{
    for (size_t i = 0; i < n; ++i)
        if (!(q[i] > -1.0e6 && q[i] < 1.0e6)) return false;
    return true;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return set_error(SMPLX_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));          \
    } while (0)

// launch of a model-dependent kernel: the per-robot build when the space has one (specialize.h)
#define KLAUNCH(space, ID, kern, grid, block, lds, stream, ...)                                                   \
    do {                                                                                                          \
        hipError_t le_ = smplx::launch((space)->ks.k[smplx::ID], kern, grid, block, lds, stream, __VA_ARGS__);   \
        if (le_ != hipSuccess) return set_error(SMPLX_E_HIP, std::string(#kern) + ": " + hipGetErrorString(le_)); \
    } while (0)

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    int reserve(size_t n)
    {
        if (n <= cap) return SMPLX_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = std::max(n, (size_t)64);
        HIP_TRY(hipMalloc((void**)&p, want * sizeof(T)));
        cap = want;
        return SMPLX_OK;
    }
    ~DevBuf() { if (p) (void)hipFree(p); }
};

template <class T>
struct PinBuf {
    T* p = nullptr;
    size_t cap = 0;
    int reserve(size_t n)
    {
        if (n <= cap) return SMPLX_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = std::max(n, (size_t)64);
        HIP_TRY(hipHostMalloc((void**)&p, want * sizeof(T), hipHostMallocDefault));
        cap = want;
        return SMPLX_OK;
    }
    ~PinBuf() { if (p) (void)hipHostFree(p); }
};

// coord -> id table: open addressing over the commit-ordered coordinate array.  State ids depend
// only on insertion order (manip_lattice.cpp:1302-1354), never on the hash function.
struct CoordTable {
    int N = 0;
    std::vector<int32_t> slots;   // id + 1, 0 = empty
    size_t mask = 0, used = 0;
    void init(int n)
    {
        N = n;
        slots.assign(1 << 16, 0);
        mask = slots.size() - 1;
        used = 0;
    }
    static uint64_t hash(const int32_t* c, int n)
    {
        uint64_t h = 0x9E3779B97F4A7C15ull;
        for (int i = 0; i < n; ++i) {
            h ^= (uint64_t)(uint32_t)c[i] + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
            h *= 0xFF51AFD7ED558CCDull;
            h ^= h >> 33;
        }
        return h;
    }
    int find(const int32_t* c, const std::vector<int32_t>& coords) const { return find_hashed(c, hash(c, N), coords); }
    // the two dependent cache misses of a lookup (slot, then the coordinate row it names), started ahead of time
    void prefetch_slot(uint64_t h) const { __builtin_prefetch(&slots[h & mask]); }
    void prefetch_row(uint64_t h, const std::vector<int32_t>& coords) const
    {
        const int32_t s = slots[h & mask];
        if (s) __builtin_prefetch(&coords[(size_t)(s - 1) * N]);
    }
    int find_hashed(const int32_t* c, uint64_t h, const std::vector<int32_t>& coords) const
    {
        size_t i = h & mask;
        while (true) {
            const int32_t s = slots[i];
            if (s == 0) return -1;
            if (std::memcmp(&coords[(size_t)(s - 1) * N], c, sizeof(int32_t) * N) == 0) return s - 1;
            i = (i + 1) & mask;
        }
    }
    void insert(int id, const std::vector<int32_t>& coords)
    {
        if ((used + 1) * 2 > slots.size()) {
            std::vector<int32_t> old;
            old.swap(slots);
            slots.assign(old.size() * 2, 0);
            mask = slots.size() - 1;
            for (int32_t s : old) if (s) place(s - 1, coords);
        }
        place(id, coords);
        ++used;
    }
    void place(int id, const std::vector<int32_t>& coords)
    {
        size_t i = hash(&coords[(size_t)id * N], N) & mask;
        while (slots[i]) i = (i + 1) & mask;
        slots[i] = id + 1;
    }
};

// Dense outputs of a planner frontier batch in ONE allocation -- successor joint values | coordinates | heuristic |
// flags -- so that they come back in one DMA copy; the device block and its pinned host twin share the layout.
struct OutView {
    double* sq = nullptr;
    int32_t* coord = nullptr;
    int32_t* h = nullptr;
    int32_t* id = nullptr;      // K5: state id of the successor's coordinate in the device table, -1 = not there
    unsigned char* flags = nullptr;
    size_t bytes = 0;
};

inline OutView carve_out(unsigned char* base, size_t BM, int N)
{
    OutView v;
    size_t o = 0;
    v.sq = (double*)(base + o); o += BM * N * sizeof(double);
    v.coord = (int32_t*)(base + o); o += BM * N * sizeof(int32_t);
    v.h = (int32_t*)(base + o); o += BM * sizeof(int32_t);
    v.id = (int32_t*)(base + o); o += BM * sizeof(int32_t);
    v.flags = base + o; o += BM;
    v.bytes = (o + 15) / 16 * 16;
    return v;
}

}  // namespace

#include "grid_handle.h"

struct smplx_model {
    smplx::HostModel hm;
};

struct DevSearch {
    unsigned char* arena = nullptr;      // one allocation carved into the buffers of SmplxSearchDev
    SmplxSearchDev* d_hdr = nullptr;
    SmplxSearchDev h;                    // host copy of the header: pointers, capacities, and the last state read back
    struct Caps { int states = 0, heap = 0, incons = 0, log = 0, succ = 0, path = 0; } caps;
    int dev_states = 0;                  // ids [0, dev_states) exist on the device
    bool table_fresh = false;            // the device table was just (re)allocated: empty
    bool host_behind = false;            // the device created states / committed lists the host arrays do not hold yet
    bool log_on_device = false;          // the expansion log of the last search has not been read back
    int call_number = 0, n_succ_kept = 0;
    int64_t grows = 0, searches = 0, ticks[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int dup_pushes = 0;
    int test_capacity = 0;               // test hook: first capacity in states
    bool test_no_helper = false;         // test hook: launch k_search without its helper wave
};

struct smplx_space {
    smplx::HostModel model;
    const smplx_grid* grid = nullptr;
    smplx::HostActions actions;
    smplx_params params;
    SmplxSpaceDev hs;
    SmplxSpaceDev* d_space = nullptr;
    hipStream_t stream = nullptr;
    int device = 0;   // HIP device the handle lives on (worker threads select it explicitly)
    int N = 0, M = 0;
    size_t lds_bytes = 0, blob_bytes = 0;
    size_t lds_bytes_valid = 0;      // k_state_valid, k_edge_valid, k_pipe_configs: in the per-robot build they keep the saved link transforms in registers
    int lds_nroot = 0;   // root-position slots per thread in LDS: none in the per-robot build (they live in registers there)
    // BFS
    int32_t* d_bfs = nullptr;
    int32_t* d_queue = nullptr;                 // brick lists of the two passes in flight
    int32_t* d_counts = nullptr;
    int32_t* d_brick_queued = nullptr;          // wave-per-brick mode: 2 x nbricks "queued for the next pass" words
    int bfs_bricks[3] = {0, 0, 0};
    int32_t* d_minus_one = nullptr;   // a device int holding -1 (k_expand: deferred pass without a counter)
    int64_t bfs_total = 0;                      // cells of the padded grid the API hands out (smplx_bfs_copy)
    int64_t bfs_ints = 0;                       // ints of the brick-major records on the device
    bool bfs_reset_due = false;
    int bfs_tag = 0;                            // tag of the last BFS run (device_types.h SmplxBfsDev), 0 before the first
    std::vector<int32_t> bfs_queue_sizes;       // bricks queued in every pass of the last BFS: sizes the next goal's launches
    int bfs_levels = 0;
    int wall_thr = -1;
    bool goal_set = false;
    uint64_t grid_epoch = 0;         // the grid's edit count when the goal was set: the successor caches belong to that field
    double goal_xyz[3] = {0, 0, 0};
    double start_xyz[3] = {0, 0, 0};   // planning-link position of the start state (getMetricStartDistance)
    int status = SMPLX_OK;            // sticky: first error of a call that has no way to report one (smplx_space_status)
    std::string status_msg;
    // scratch
    DevBuf<double> b_q, b_q2, b_sq, b_xyz;
    DevBuf<unsigned char> b_flags, b_work;
    DevBuf<int32_t> b_coord, b_h, b_cost, b_lookups, b_way;
    smplx::KernelSet ks;       // per-robot kernels (specialize.h), generic ones with SMPLX_SPACE_GENERIC_KERNELS or SMPLX_SPECIALIZE=0
    std::string specialize_note;   // why the per-robot build is absent, if it is
    bool fused_mode = false;   // SMPLX_SPACE_FUSED: one thread per edge (reference lookup tallies)
    int work_list_items = 0;   // > 0: test hook (test_hooks.h) -- a work list this small, so that the deferred pass is exercised
    int small_batch_max = 512;     // batches up to this many states take the single-launch kernel (SMPLX_SPACE_NO_SMALL_KERNEL disables)
    double small_latency_limit = 70e-6;   // SMPLX_SMALL_KERNEL=always lifts it, =never disables the single-launch kernel
    DevBuf<unsigned long long> b_counters;
    PinBuf<double> p_q;
    DevBuf<unsigned char> b_out;   // packed outputs of a planner batch (OutView)
    PinBuf<unsigned char> p_out;
    OutView dv, pv;                // views of the batch in flight: device block, pinned host twin
    // lattice: commit-ordered state table (manip_lattice.cpp:1302-1354)
    std::vector<int32_t> coords;
    std::vector<double> qs;
    std::vector<int32_t> h_of_id;
    CoordTable table;
    int start_id = -1;
    // device copy of the state table (K5; SmplxTableDev in hs.table): the states created since the last synchronisation
    // wait in pending_ins as (query slot, id, coord[N]) triples and go up with the next frontier batch
    int32_t* d_table = nullptr;
    size_t table_cap = 0, table_count = 0;
    std::vector<int32_t> pending_ins;
    DevBuf<int32_t> b_ins;
    PinBuf<int32_t> p_ins;
    // speculative successor cache (per state id: evaluated but not yet committed successors)
    struct Rec { int32_t cost; int32_t h; int32_t goal; int32_t known; int32_t prim; };
    std::vector<int64_t> cache_off;     // per id: first record, -1 = not evaluated
    std::vector<int32_t> cache_cnt;
    std::vector<Rec> recs;
    std::vector<int32_t> rec_coord;
    std::vector<double> rec_q;
    // committed successor lists (served on re-expansion in later ARA* iterations)
    std::vector<int64_t> done_off;
    std::vector<int32_t> done_cnt;
    std::vector<int32_t> done_succ, done_cost, done_prim;
    DevSearch ds;                       // device-resident ARA* of this query (search_host.h)
    std::vector<int32_t> hint;
    // Speculation for callers that only know GetSuccs (an unchanged SBPL planner never calls smplx_hint_frontier): the
    // space mirrors the g-values the caller's expansions imply (g[succ] = min(g[succ], g[id] + cost), exactly what
    // ARAStar::expand does, arastar.cpp:546-551) and, on a miss, lets the created-but-unevaluated states with the
    // smallest g + w*h ride along.  Only a guess at the caller's OPEN order: a wrong guess costs GPU work, never results.
    bool plain_mode = false;            // set by the first smplx_get_succs from outside the engine's own search
    int auto_spec = 96;                 // states that ride along per miss (SMPLX_AUTO_SPECULATE, 0 = off)
    double auto_w = 5.0;                // weight of h in the ranking (SMPLX_AUTO_SPECULATE_W)
    std::vector<uint32_t> g_est;        // per id, mirrored g (plain mode only)
    std::vector<std::pair<uint64_t, int32_t>> pool;   // binary min-heap of (rank key, id) of unevaluated states
    // a frontier batch in flight (issued on `stream`, completion signalled by `batch_done`)
    std::vector<int32_t> inflight;
    std::vector<double> inflight_q;     // smplx_plan_multi: the joint values of `inflight`, staged by the query's worker
    hipEvent_t batch_done = nullptr;
    bool inflight_zero_copy = false;   // the batch in flight wrote its results straight into the pinned host buffers
    // Small batches: the single-launch kernel costs the host one launch (27 us issue-to-landing for the handful of states
    // a lone query misses on), the pipeline several launches and copies (~34 us).  Both give the same bytes.  The engine
    // watches the issue-to-landing time of the single-launch path and sits out 2000 batches on the pipeline path whenever
    // its moving average exceeds 70 us: a safety net from the time the kernel checked the snap-to-goal edge of every
    // state ungated (105 us per launch; fixed, see k_small_batch) -- it costs nothing when the kernel behaves.
    std::chrono::steady_clock::time_point t_issue;
    bool inflight_small = false;
    bool adaptive_small = false;  // only a lone query measures: with several queries per thread the landing time includes their turns
    double small_latency = 0.0;   // moving average, seconds
    int small_seen = 0, pipeline_left = 0;
    int64_t small_launches = 0, pipe_launches = 0;
    // stats
    int64_t gpu_batches = 0, cache_hits = 0, cache_misses = 0, committed_evals = 0, gpu_evals = 0;
    // cross-query batches (smplx_plan_multi): query table + per-state query index, owned by the leading space
    DevBuf<const SmplxSpaceDev*> b_stab;
    DevBuf<unsigned short> b_stateq;
    PinBuf<unsigned short> p_stateq;
    std::vector<int32_t> eval_count;    // per id: evaluated (active) primitives, for committed_evals
    std::vector<int32_t> expansion_log;
    // optional per-kernel timing of expand launches (bench.py roofline): 3 events per launch
    std::vector<hipEvent_t> prof_events;
    size_t prof_used = 0;
};

namespace {

int upload_space(smplx_space* s)
{
    HIP_TRY(hipMemcpyAsync(s->d_space, &s->hs, sizeof(SmplxSpaceDev), hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return SMPLX_OK;
}

inline int blocks_for(long long n, int block) { return (int)((n + block - 1) / block); }

// host mirror of ManipLattice::stateToCoord (manip_lattice.cpp:1263-1289) on det_math
void state_to_coord(const SmplxModelDev& m, const double* q, int32_t* c)
{
    for (int v = 0; v < m.nvars; ++v) {
        const double delta = m.coord_delta[v];
        if (m.var_type[v] == SMPLX_JT_CONTINUOUS) {
            const double pos = smplx_normalize_angle_positive(q[v]);
            int k = (int)((pos + delta * 0.5) / delta);
            if (k == m.coord_vals[v]) k = 0;
            c[v] = k;
        } else {
            c[v] = (int)(((q[v] - m.var_min[v]) / delta) + 0.5);
        }
    }
}

// KDLRobotModel::checkJointLimits (kdl_robot_model.cpp:173-189, 210-235)
bool host_check_limits(const SmplxModelDev& m, const double* q)
{
    for (int v = 0; v < m.nvars; ++v) {
        double a = q[v];
        if (std::fabs(a) > SMPLX_2PI) a = std::fmod(a, SMPLX_2PI);
        while (a > m.var_min_norm[v]) a -= SMPLX_2PI;
        while (a < m.var_min[v]) a += SMPLX_2PI;
        if (a < m.var_min[v] || a > m.var_max[v]) return false;
    }
    return true;
}

int run_heuristic(smplx_space* s, const double* q, int n, int32_t* h, double* xyz)
{
    const int N = s->N;
    if (int e = s->b_q.reserve((size_t)n * N)) return e;
    if (int e = s->b_h.reserve(n)) return e;
    if (int e = s->b_xyz.reserve((size_t)n * 3)) return e;
    HIP_TRY(hipMemcpyAsync(s->b_q.p, q, sizeof(double) * n * N, hipMemcpyHostToDevice, s->stream));
    KLAUNCH(s, K_HEURISTIC, k_heuristic, dim3(blocks_for(n, SMPLX_BLOCK)), dim3(SMPLX_BLOCK), s->blob_bytes, s->stream, s->d_space, s->b_q.p, n,
                       s->b_h.p, s->b_xyz.p);
    HIP_TRY(hipGetLastError());
    if (h) HIP_TRY(hipMemcpyAsync(h, s->b_h.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, s->stream));
    if (xyz) HIP_TRY(hipMemcpyAsync(xyz, s->b_xyz.p, sizeof(double) * n * 3, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return SMPLX_OK;
}

constexpr int kBfsHistory = 2048;   // passes whose queue sizes are kept behind the counters (d_counts)

// BFS_3D::run to completion on the device (bfs3d.cpp:156-201, 507-547): passes over the queued 8x8x8 bricks until none is
// queued (kernels.hip k_bfs_brick_wave)
int run_bfs(smplx_space* s, const double xyz[3])
{
    const smplx_grid* g = s->grid;
    int c[3];
    for (int a = 0; a < 3; ++a) c[a] = (int)(g->dev.inv_res * (xyz[a] - g->dev.origin_minus_res[a]) + 0.5) - 1;
    // BFS_3D::run's reset (bfs3d.cpp:162-166): the run's tag makes every other run's distances UNDISCOVERED; a pass over
    // the records only when the tags wrap (finish_goal chose the tag and uploaded it)
    if (s->bfs_reset_due) {
        hipLaunchKernelGGL(k_bfs_reset, dim3(2048), dim3(256), 0, s->stream, s->d_bfs, (size_t)s->bfs_ints);
        HIP_TRY(hipGetLastError());
        s->bfs_reset_due = false;
    }
    const int tag_word = s->hs.bfs.tag_word, tag_mask = s->hs.bfs.tag_mask;
    s->bfs_levels = 0;
    const bool in_bounds = !(c[0] < 0 || c[1] < 0 || c[2] < 0 || c[0] >= g->n[0] || c[1] >= g->n[1] || c[2] >= g->n[2]);
    if (!in_bounds) {   // bfs3d.cpp:169-171: nothing is labelled
        HIP_TRY(hipStreamSynchronize(s->stream));
        return SMPLX_OK;
    }
    const int nbx = s->bfs_bricks[0], nby = s->bfs_bricks[1], nbz = s->bfs_bricks[2];
    const int nbricks = nbx * nby * nbz;
    // two brick lists alternate, each cut into 16 sub-lists of nbricks entries with their own counters on separate
    // lines: d_queue holds the lists, d_counts the 3 x 16 counters (in / next / zeroed for the pass after)
    const int kShards = 16;
    const size_t list_ints = (size_t)kShards * nbricks;
    int32_t* lists = s->d_queue;
    hipLaunchKernelGGL(k_bfs_brick_seed, dim3(1), dim3(64), 0, s->stream, s->d_bfs, c[0], c[1], c[2], nbx, nby, nbz, lists, s->d_counts, tag_word);
    HIP_TRY(hipGetLastError());
    int pass = 0;
    std::vector<int32_t> cnt(3 * kShards * 32 + kBfsHistory);
    int32_t* queued[2] = {s->d_brick_queued, s->d_brick_queued + nbricks};
    int32_t* d_history = s->d_counts + 3 * kShards * 32;
    // Launch sizes.  Every block of a launch reads the counters even when it has no brick (16 384 mostly idle blocks cost
    // ~6 us, 2 048 ~2.4 us), and every look at the counters from the host costs ~40 us (copy, synchronise, the stream
    // running dry).  The passes of two goals in one grid are much alike, so the queue sizes of the last BFS (kept by
    // the kernel behind the counters) size this one: all its passes plus two are enqueued at once, each with twice
    // the blocks its neighbourhood of passes had bricks, and the one look at the end usually finds nothing queued.  A first BFS
    // -- or one that outlives the plan -- goes in chunks: 16 passes while the front is wide, 4 once fewer than 256 bricks
    // are queued (the tail is a narrow front: a chunk of 16 wasted eight passes on average).
    const std::vector<int32_t> plan = s->bfs_queue_sizes;
    int planned = 0;
    for (size_t k = 0; k < plan.size(); ++k) if (plan[k] > 0) planned = (int)k + 1;
    const int wave_grid_max = 16384;
    int wave_grid = planned > 0 ? 2048 : wave_grid_max;    // (past the plan: its tail)
    int chunk = planned > 0 ? planned + 2 : 16;
    const bool dbg = getenv("SMPLX_DEBUG_TIMING") != nullptr;
    if (dbg) chunk = 1;     // one look at the counters per pass: bricks and microseconds of every pass on stderr
    auto grid_of = [&](int p) {
        if (p >= planned) return wave_grid;
        int m = 0;
        for (int k = std::max(0, p - 1); k <= std::min(planned - 1, p + 1); ++k) m = std::max(m, plan[k]);
        return std::min(wave_grid_max, std::max(1024, 2 * m));
    };
    while (true) {
        const auto tp0 = std::chrono::steady_clock::now();
        for (int k = 0; k < chunk; ++k, ++pass) {
            const int in = pass & 1, out = (pass + 1) & 1;
            const int c_in = pass % 3, c_next = (pass + 1) % 3, c_after = (pass + 2) % 3;
            hipLaunchKernelGGL(k_bfs_brick_wave, dim3(std::min(nbricks, grid_of(pass))), dim3(64), 0, s->stream, s->d_bfs, nbx, nby, nbz,
                               lists + in * list_ints, s->d_counts + c_in * kShards * 32, lists + out * list_ints,
                               s->d_counts + c_next * kShards * 32, s->d_counts + c_after * kShards * 32, nbricks,
                               queued[in], queued[out], pass < kBfsHistory ? d_history + pass : (int32_t*)nullptr, tag_word, tag_mask);
        }
        HIP_TRY(hipGetLastError());
        const size_t look = 3 * kShards * 32 + (size_t)std::min(pass, kBfsHistory);
        HIP_TRY(hipMemcpyAsync(cnt.data(), s->d_counts, sizeof(int32_t) * look, hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));
        long pending = 0;
        const int set = pass % 3;   // the "in" counters of the pass that would come next
        for (int k = 0; k < kShards; ++k) pending += cnt[(size_t)set * kShards * 32 + 32 * k];
        if (dbg) {
            fprintf(stderr, "[smplx bfs] pass %d: %.1f us (launch + sync), %ld bricks queued for the next\n", pass - 1,
                    1e6 * std::chrono::duration<double>(std::chrono::steady_clock::now() - tp0).count(), pending);
#ifdef SMPLX_BFS_TRACE
            {
                long long tr[16], zero[16] = {0};
                if (hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_bfs_trace), sizeof(tr)) == hipSuccess && tr[7] > 0) {
                    static const char* names[7] = {"", "list", "tile load", "sweeps", "stores", "requeue test", "claim"};
                    fprintf(stderr, "[smplx bfs]   %lld visits, longest / mean (us):", tr[7]);
                    for (int k = 1; k < 7; ++k) fprintf(stderr, " %s %.2f / %.2f%s", names[k], 0.01 * tr[k], 0.01 * tr[8 + k] / tr[7], k < 6 ? "," : "\n");
                }
                (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bfs_trace), zero, sizeof(zero));
            }
#endif
        }
        if (pending == 0) break;
        if (!dbg) chunk = pending < 256 ? 4 : 16;
        wave_grid = pending < 256 ? std::min(wave_grid_max, 2048) : wave_grid_max;
        // (a label-correcting brick sweep can legitimately need on the order of nbricks passes on maze-like free space)
        if (pass > 4 * nbricks + 1024) return set_error(SMPLX_E_HIP, "BFS did not terminate");
    }
    s->bfs_queue_sizes.assign(cnt.begin() + 3 * kShards * 32, cnt.begin() + 3 * kShards * 32 + std::min(pass, kBfsHistory));
    s->bfs_levels = pass;
    return SMPLX_OK;
}

// per-block tallies: 4 uint64 per block of the (state x primitive) grid (kernels.hip tally_block)
inline size_t counter_words(int B, int M) { return (size_t)blocks_for((long long)B * M, SMPLX_BLOCK) * SMPLX_TALLIES; }

// carve of the per-batch device scratch (smplx_expand_work_bytes)
struct ExpandWork {
    double* goal_dist;
    int32_t* state_lookups;
    unsigned char* state_bad;
    int32_t* edge_w;
    int32_t* edge_lookups;
    unsigned char* edge_bad;
    int32_t* work_count;
    unsigned long long* work;   // 64-bit items: edge | waypoint << 32 | waypoint count << 48
    int capacity;
};

inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }

size_t expand_work_bytes(int B, int M)
{
    const size_t b = (size_t)B, bm = (size_t)B * M;
    return align256(b * 8) + align256(b * 4) + align256(b) + align256(bm * 4) + align256(bm * 4) + align256(bm) + 2048 +
           align256(bm * 16 * 8);
}

ExpandWork carve_work(void* base, int B, int M)
{
    unsigned char* w = (unsigned char*)base;
    const size_t b = (size_t)B, bm = (size_t)B * M;
    ExpandWork k;
    k.goal_dist = (double*)w; w += align256(b * 8);
    k.state_lookups = (int32_t*)w; w += align256(b * 4);
    k.state_bad = w; w += align256(b);
    k.edge_w = (int32_t*)w; w += align256(bm * 4);
    k.edge_lookups = (int32_t*)w; w += align256(bm * 4);
    k.edge_bad = w; w += align256(bm);
    k.work_count = (int32_t*)w; w += 2048;   // 8 shard counters + deferred count, one 128-byte line each
    k.work = (unsigned long long*)w;
    k.capacity = (int)std::min<size_t>(bm * 16, (size_t)1 << 30) / 8 * 8;
    return k;
}

// ---- device copy of the state table (K5) ------------------------------------------------------------------------
int table_alloc(smplx_space* s, size_t cap)
{
    if (s->d_table) (void)hipFree(s->d_table);
    s->d_table = nullptr;
    const int stride = smplx_table_stride(s->N);
    HIP_TRY(hipMalloc((void**)&s->d_table, cap * (size_t)stride * sizeof(int32_t)));
    HIP_TRY(hipMemsetAsync(s->d_table, 0, cap * (size_t)stride * sizeof(int32_t), s->stream));
    s->table_cap = cap;
    s->hs.table.slots = s->d_table;
    s->hs.table.mask = (uint32_t)(cap - 1);
    s->hs.table.stride = stride;
    s->hs.table.pad = 0;
    return SMPLX_OK;
}

// load factor above 1/2: a table four times the size, every committed state re-inserted with the next batch
int table_grow_if_needed(smplx_space* s)
{
    if (!s->d_table || s->table_count * 2 <= s->table_cap) return SMPLX_OK;
    size_t cap = s->table_cap;
    while (s->table_count * 2 > cap) cap *= 4;
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (int e = table_alloc(s, cap)) return e;
    if (int e = upload_space(s)) return e;
    const int nstates = (int)s->h_of_id.size();
    s->pending_ins.clear();
    s->pending_ins.reserve((size_t)nstates * (s->N + 2));
    for (int id = 1; id < nstates; ++id) {
        s->pending_ins.push_back(0);
        s->pending_ins.push_back(id);
        s->pending_ins.insert(s->pending_ins.end(), &s->coords[(size_t)id * s->N], &s->coords[(size_t)id * s->N] + s->N);
    }
    return SMPLX_OK;
}

// append the space's pending inserts to a staging array, tagged with its slot in the batch's query table
void table_take_pending(smplx_space* s, int slot, std::vector<int32_t>& items)
{
    const size_t w = (size_t)s->N + 2;
    const size_t o = items.size();
    items.insert(items.end(), s->pending_ins.begin(), s->pending_ins.end());
    for (size_t k = o; k < items.size(); k += w) items[k] = slot;
    s->pending_ins.clear();
}

// upload staged inserts and run k_table_insert on `stream` (before the expansion kernels of the same stream)
int table_upload(smplx_space* lead, const std::vector<int32_t>& items, DevBuf<int32_t>& dbuf, PinBuf<int32_t>& pbuf, hipStream_t stream,
                 const SmplxSpaceDev* const* stab)
{
    if (items.empty()) return SMPLX_OK;
    const int n = (int)(items.size() / ((size_t)lead->N + 2));
    if (int e = dbuf.reserve(items.size())) return e;
    if (int e = pbuf.reserve(items.size())) return e;
    std::memcpy(pbuf.p, items.data(), items.size() * sizeof(int32_t));
    HIP_TRY(hipMemcpyAsync(dbuf.p, pbuf.p, items.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_table_insert, dim3(blocks_for(n, SMPLX_BLOCK)), dim3(SMPLX_BLOCK), 0, stream, lead->d_space, stab, dbuf.p, n, lead->N);
    HIP_TRY(hipGetLastError());
    return SMPLX_OK;
}

// first use of the device table on a space that was created without one (smplx_table_sync, the K5 entry points):
// allocate it for the states there are and queue them all
int table_ensure(smplx_space* s)
{
    if (s->d_table) return SMPLX_OK;
    size_t cap = (size_t)1 << 18;
    const size_t nstates = s->h_of_id.size();
    while (nstates * 2 > cap) cap *= 4;
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (int e = table_alloc(s, cap)) return e;
    if (int e = upload_space(s)) return e;
    s->pending_ins.clear();
    for (int id = 1; id < (int)nstates; ++id) {
        s->pending_ins.push_back(0);
        s->pending_ins.push_back(id);
        s->pending_ins.insert(s->pending_ins.end(), &s->coords[(size_t)id * s->N], &s->coords[(size_t)id * s->N] + s->N);
    }
    s->table_count = nstates > 0 ? nstates - 1 : 0;
    return SMPLX_OK;
}

// the space's own batches: everything pending goes up on its stream
int table_flush(smplx_space* s, hipStream_t stream)
{
    if (!s->d_table) return SMPLX_OK;
    if (int e = table_grow_if_needed(s)) return e;
    if (s->pending_ins.empty()) return SMPLX_OK;
    std::vector<int32_t> items;
    table_take_pending(s, 0, items);
    return table_upload(s, items, s->b_ins, s->p_ins, stream, nullptr);
}

// optional K5 outputs of an expansion launch
struct K5Out {
    int32_t* d_id = nullptr;                 // dense [B][M] ids (-1 = unknown)
    const SmplxCompactDev* cmp = nullptr;    // compact stream (device pointers), or null
    const int32_t* items = nullptr;          // states to insert at the head of the batch's first kernel: n_items x (N + 2)
    int n_items = 0;                         //   int32 (device memory, or pinned host memory for the zero-copy launch)
};

// The pending inserts of a batch travel in the same upload as its parents: they sit behind the B x N doubles of the
// pinned parent buffer.  Returns the doubles the items occupy; *items_at = their offset in doubles.
size_t stage_items(PinBuf<double>& p_q, size_t parent_doubles, const std::vector<int32_t>& items)
{
    if (items.empty()) return 0;
    std::memcpy((void*)(p_q.p + parent_doubles), items.data(), items.size() * sizeof(int32_t));
    return (items.size() + 1) / 2;
}

// pinned host buffers of a zero-copy small batch: the kernel reads the parents from, and also writes the results to, host
// memory (a few KB of PCIe traffic instead of DMA copies with their fixed latency)
struct ZeroCopy {
    const double* q = nullptr;
    unsigned char* flags = nullptr;
    int32_t* coord = nullptr;
    double* sq = nullptr;
    int32_t* h = nullptr;
    int32_t* id = nullptr;
};

bool small_kernel_fits(const smplx_space* s, int B)
{
    const int small_block = smplx_small_block(s->M);
    const size_t small_lds = smplx_lds_bytes_n(s->blob_bytes, s->lds_nroot, s->model.dev.nslots, s->model.dev.nvars, s->model.dev.stack_bytes, small_block);
    return !s->fused_mode && B <= s->small_batch_max && small_block <= 512 && small_lds <= 150 * 1024 && s->work_list_items == 0 &&
           s->pipeline_left == 0;
}

int launch_expand(smplx_space* s, const double* d_q, int B, unsigned char* d_flags, int32_t* d_coord, double* d_sq,
                  int32_t* d_h, int32_t* d_cost, int32_t* d_lookups, void* d_work, unsigned long long* d_counters,
                  hipStream_t stream, const SmplxSpaceDev* const* stab = nullptr, const unsigned short* state_q = nullptr,
                  const ZeroCopy* zero_copy = nullptr, bool force_pipeline = false, const K5Out* k5 = nullptr)
{
    ExpandWork k = carve_work(d_work, B, s->M);
    int32_t* d_id = k5 ? k5->d_id : nullptr;
    SmplxCompactDev cmp;
    std::memset(&cmp, 0, sizeof(cmp));
    if (k5 && k5->cmp) cmp = *k5->cmp;
    if (cmp.rec_a) force_pipeline = true;   // the compact stream is produced by k_pipe_finish
    const int32_t* ins_items = k5 && s->d_table ? k5->items : nullptr;
    const int n_ins = ins_items ? k5->n_items : 0;
    if (s->work_list_items > 0) k.capacity = s->work_list_items;   // test hook: almost every edge overflows into the deferred pass
    hipEvent_t* ev = nullptr;
    if (s->prof_used + 3 <= s->prof_events.size()) { ev = &s->prof_events[s->prof_used]; s->prof_used += 3; }
    const int bs = blocks_for(B, SMPLX_BLOCK);
    const int be = blocks_for((long long)B * s->M, SMPLX_BLOCK);
    const int64_t* norefs = nullptr;
    const int small_block = smplx_small_block(s->M);
    const size_t small_lds = smplx_lds_bytes_n(s->blob_bytes, s->lds_nroot, s->model.dev.nslots, s->model.dev.nvars, s->model.dev.stack_bytes, small_block);
    if (!force_pipeline && !s->fused_mode && !ev && B <= s->small_batch_max && small_block <= 512 && small_lds <= 150 * 1024 &&
        s->work_list_items == 0 && s->pipeline_left == 0) {
        ++s->small_launches;
        // a handful of states: ONE launch, all FK chains side by side (kernels.hip k_small_batch)
        // zero_copy: parents are read from, and results also written to, that space's pinned host buffers
        const double* qsrc = zero_copy ? zero_copy->q : d_q;
        KLAUNCH(s, K_SMALL_BATCH, k_small_batch, dim3(B + blocks_for(n_ins, small_block)), dim3(small_block), small_lds, stream, s->d_space, qsrc, norefs, B, k.goal_dist,
                           k.state_bad, k.state_lookups, d_flags, d_coord, d_sq, d_h, d_cost, d_lookups, stab, state_q,
                           zero_copy ? zero_copy->flags : (unsigned char*)nullptr, zero_copy ? zero_copy->coord : (int32_t*)nullptr,
                           zero_copy ? zero_copy->sq : (double*)nullptr, zero_copy ? zero_copy->h : (int32_t*)nullptr, d_id,
                           zero_copy ? zero_copy->id : (int32_t*)nullptr, ins_items, n_ins);
    } else if (s->fused_mode) {
        // one thread walks a whole edge: exact reference early-exit order (and lookup tallies)
        if (d_id) HIP_TRY(hipMemsetAsync(d_id, 0xFF, sizeof(int32_t) * (size_t)B * s->M, stream));   // fused mode: no table lookups
        if (n_ins > 0) {
            hipLaunchKernelGGL(k_table_insert, dim3(blocks_for(n_ins, SMPLX_BLOCK)), dim3(SMPLX_BLOCK), 0, stream, s->d_space, stab, ins_items, n_ins, s->N);
        }
        if (ev) (void)hipEventRecord(ev[0], stream);
        KLAUNCH(s, K_STATE_PREP, k_state_prep, dim3(bs), dim3(SMPLX_BLOCK), s->lds_bytes, stream, s->d_space, d_q, norefs, B,
                           k.goal_dist, k.state_bad, k.state_lookups, stab, state_q);
        if (ev) (void)hipEventRecord(ev[1], stream);
        KLAUNCH(s, K_EXPAND, k_expand, dim3(be), dim3(SMPLX_BLOCK), s->lds_bytes, stream, s->d_space, d_q, norefs, B,
                           k.goal_dist, k.state_bad, k.state_lookups, d_flags, d_coord, d_sq, d_h, d_cost, d_lookups,
                           d_counters, (const int*)nullptr, stab, state_q);
        if (ev) (void)hipEventRecord(ev[2], stream);
    } else {
        const size_t lm = s->blob_bytes;
        ++s->pipe_launches;
        KLAUNCH(s, K_PIPE_PREP, k_pipe_prep, dim3(bs + blocks_for(n_ins, SMPLX_BLOCK)), dim3(SMPLX_BLOCK), lm, stream, s->d_space, d_q, norefs, B,
                           k.goal_dist, k.work_count, stab, state_q, cmp.totals, ins_items, n_ins);
        KLAUNCH(s, K_PIPE_SETUP, k_pipe_setup, dim3(be), dim3(SMPLX_BLOCK), lm, stream, s->d_space, d_q, norefs, B,
                           k.goal_dist, d_flags, d_sq, k.edge_w, k.edge_lookups, k.edge_bad, k.state_lookups, k.state_bad,
                           k.work, k.work_count, k.capacity, stab, state_q);
        if (ev) (void)hipEventRecord(ev[0], stream);
        // (a smaller grid was tried -- idle blocks cost next to nothing: 22.0 us at 3 configurations per edge, 21.7 at 1.35)
        const int bc = blocks_for((long long)B + (long long)B * s->M * 3, SMPLX_BLOCK);
        KLAUNCH(s, K_PIPE_CONFIGS, k_pipe_configs, dim3(bc), dim3(SMPLX_BLOCK), s->lds_bytes_valid, stream, s->d_space, d_q, norefs, B,
                           d_sq, k.edge_w, k.edge_lookups, k.edge_bad, k.state_lookups, k.state_bad, k.work, k.work_count,
                           k.capacity);
        if (ev) (void)hipEventRecord(ev[1], stream);
        // edges whose waypoints did not fit the work list (normally none) are walked whole by their finish thread
        KLAUNCH(s, K_PIPE_FINISH, k_pipe_finish, dim3(be), dim3(SMPLX_BLOCK), s->lds_bytes, stream, s->d_space, d_q, norefs, B,
                           k.edge_w, k.edge_lookups, k.edge_bad, k.state_lookups, k.state_bad, d_flags, d_coord, d_sq, d_h,
                           d_cost, d_lookups, d_counters, k.goal_dist, stab, state_q, d_id, cmp);
        if (ev) (void)hipEventRecord(ev[2], stream);
    }
    HIP_TRY(hipGetLastError());
    return SMPLX_OK;
}

int reserve_expand(smplx_space* s, int B)
{
    const size_t BM = (size_t)B * s->M;
    int e;
    if ((e = s->b_q.reserve((size_t)B * s->N))) return e;
    if ((e = s->b_work.reserve(expand_work_bytes(B, s->M)))) return e;
    if ((e = s->b_flags.reserve(BM))) return e;
    if ((e = s->b_coord.reserve(BM * s->N))) return e;
    if ((e = s->b_sq.reserve(BM * s->N))) return e;
    if ((e = s->b_h.reserve(BM))) return e;
    if ((e = s->b_cost.reserve(BM))) return e;
    if ((e = s->b_lookups.reserve(BM))) return e;
    if ((e = s->b_counters.reserve(counter_words(B, s->M)))) return e;
    return SMPLX_OK;
}

int new_state(smplx_space* s, const int32_t* coord, const double* q, int32_t h)
{
    const int id = (int)(s->coords.size() / s->N);
    s->coords.insert(s->coords.end(), coord, coord + s->N);
    s->qs.insert(s->qs.end(), q, q + s->N);
    s->h_of_id.push_back(h);
    s->cache_off.push_back(-1);
    s->cache_cnt.push_back(0);
    s->done_off.push_back(-1);
    s->done_cnt.push_back(0);
    s->eval_count.push_back(0);
    s->table.insert(id, s->coords);
    if (s->plain_mode) s->g_est.push_back(1000000000u);
    if (s->d_table) {
        s->pending_ins.push_back(0);
        s->pending_ins.push_back(id);
        s->pending_ins.insert(s->pending_ins.end(), coord, coord + s->N);
        ++s->table_count;
    }
    return id;
}

// plain-GetSuccs speculation pool (see smplx_space::plain_mode)
inline uint64_t pool_key(const smplx_space* s, int id)
{
    const int32_t h = s->h_of_id[id];
    const double k = (double)s->g_est[id] + s->auto_w * (double)(h < 0 ? 0 : h);
    return k >= 1.8e19 ? ~0ull : (uint64_t)k;
}

inline void pool_push(smplx_space* s, int id)
{
    s->pool.emplace_back(pool_key(s, id), id);
    std::push_heap(s->pool.begin(), s->pool.end(), std::greater<std::pair<uint64_t, int32_t>>());
}

// fill s->hint with the best-ranked states that are neither evaluated nor committed
void auto_hint(smplx_space* s, int miss_id)
{
    s->hint.clear();
    const auto cmp = std::greater<std::pair<uint64_t, int32_t>>();
    while (!s->pool.empty() && (int)s->hint.size() < s->auto_spec) {
        std::pop_heap(s->pool.begin(), s->pool.end(), cmp);
        const std::pair<uint64_t, int32_t> top = s->pool.back();
        s->pool.pop_back();
        const int id = top.second;
        if (id == miss_id || s->cache_off[id] != -1 || s->done_off[id] >= 0) continue;   // evaluated meanwhile
        if (top.first != pool_key(s, id)) continue;                                        // a better-ranked copy exists
        s->hint.push_back(id);
    }
}

void reset_lattice(smplx_space* s)
{
    s->coords.clear(); s->qs.clear(); s->h_of_id.clear();
    s->cache_off.clear(); s->cache_cnt.clear(); s->recs.clear(); s->rec_coord.clear(); s->rec_q.clear();
    s->done_off.clear(); s->done_cnt.clear(); s->done_succ.clear(); s->done_cost.clear(); s->done_prim.clear();
    s->ds.dev_states = 0; s->ds.host_behind = false; s->ds.log_on_device = false; s->ds.n_succ_kept = 0;
    s->ds.table_fresh = s->d_table != nullptr;   // (emptied below)
    s->eval_count.clear();
    s->hint.clear();
    s->pool.clear();
    s->g_est.clear();
    s->plain_mode = false;
    s->table.init(s->N);
    s->start_id = -1;
    s->pending_ins.clear();
    s->table_count = 0;
    if (s->d_table) (void)hipMemsetAsync(s->d_table, 0, s->table_cap * (size_t)s->hs.table.stride * sizeof(int32_t), s->stream);
    // id 0 is reserved for the goal (manip_lattice.cpp:122); it has no coordinate and is never hashed
    s->coords.assign(s->N, 0);
    s->qs.assign(s->N, 0.0);
    s->h_of_id.push_back(0);
    s->cache_off.push_back(-1); s->cache_cnt.push_back(0);
    s->done_off.push_back(-1); s->done_cnt.push_back(0);
    s->eval_count.push_back(0);
}

// evaluate the successors of `id` plus hinted frontier states in one frontier batch
// the states of the next frontier batch: `id` plus the hinted frontier states that are neither cached nor in flight
void select_batch(smplx_space* s, int id, int cap)
{
    std::vector<int32_t>& batch = s->inflight;
    batch.clear();
    batch.push_back(id);
    s->cache_off[id] = -2;   // mark as "in this batch"
    for (int32_t hId : s->hint) {
        if ((int)batch.size() >= cap) break;
        if (hId <= 0 || hId >= (int)s->cache_off.size()) continue;
        if (s->cache_off[hId] != -1 || s->done_off[hId] >= 0) continue;
        s->cache_off[hId] = -2;
        batch.push_back(hId);
    }
    s->hint.clear();
}

// A frontier batch takes tens of microseconds; an interrupt-driven hipEventSynchronize adds about as much again to
// wake the thread up.  The search thread has nothing else to do, so it polls -- with a deadline: a batch that has not
// landed after SMPLX_BATCH_TIMEOUT_S seconds (default 30; a batch takes well under a millisecond) is a hung kernel,
// and the caller gets SMPLX_E_HIP instead of a thread that never returns (include/smpl_amd.h: every function returns).
double batch_timeout_seconds()
{
    static const double t = [] {
        const char* e = getenv("SMPLX_BATCH_TIMEOUT_S");
        const double v = e ? atof(e) : 0.0;
        return v > 0.0 ? v : 30.0;
    }();
    return t;
}

int wait_event_polling(hipEvent_t ev)
{
    std::chrono::steady_clock::time_point t0;
    bool timing = false;
    for (unsigned spins = 0;; ++spins) {
        const hipError_t st = hipEventQuery(ev);
        if (st == hipSuccess) return SMPLX_OK;
        if (st != hipErrorNotReady) return set_error(SMPLX_E_HIP, std::string("hipEventQuery: ") + hipGetErrorString(st));
        if ((spins & 0x3FFF) == 0x3FFF) {   // look at the clock every 16k polls (a few milliseconds)
            const auto now = std::chrono::steady_clock::now();
            if (!timing) { t0 = now; timing = true; }
            else if (std::chrono::duration<double>(now - t0).count() > batch_timeout_seconds())
                return set_error(SMPLX_E_HIP, "frontier batch did not complete within SMPLX_BATCH_TIMEOUT_S: kernel hung?");
        }
    }
}

// small batches skip the DMA copies: the kernel reads the parents from, and writes the results to, pinned host memory
bool takes_small_kernel(const smplx_space* s, int B)
{
    const int small_block = smplx_small_block(s->M);
    const size_t small_lds = smplx_lds_bytes_n(s->blob_bytes, s->lds_nroot, s->model.dev.nslots, s->model.dev.nvars, s->model.dev.stack_bytes, small_block);
    return !s->fused_mode && s->prof_events.empty() && B <= s->small_batch_max && small_block <= 512 && small_lds <= 150 * 1024 &&
           s->work_list_items == 0 && s->pipeline_left == 0;
}

// enqueue one frontier batch (state `id` plus hinted frontier states) on the space's stream: upload, the
// expansion pipeline, download, completion event.  Returns without waiting.
int issue_batch(smplx_space* s, int id)
{
    const int N = s->N, M = s->M;
    const int cap = s->params.batch_states > 0 ? s->params.batch_states : 4096;
    select_batch(s, id, cap);
    std::vector<int32_t>& batch = s->inflight;
    const int B = (int)batch.size();
    const size_t BM = (size_t)B * M;
    if (int e = reserve_expand(s, B)) return e;
    int e;
    if ((e = s->p_q.reserve((size_t)B * N))) return e;
    const size_t out_bytes = carve_out(nullptr, BM, N).bytes;
    if ((e = s->b_out.reserve(out_bytes))) return e;
    if ((e = s->p_out.reserve(out_bytes))) return e;
    s->dv = carve_out(s->b_out.p, BM, N);
    s->pv = carve_out(s->p_out.p, BM, N);
    auto pack_parents = [&]() {
        for (int i = 0; i < B; ++i) std::memcpy(&s->p_q.p[(size_t)i * N], &s->qs[(size_t)batch[i] * N], sizeof(double) * N);
    };
    pack_parents();
    if (s->pipeline_left > 0 && B <= s->small_batch_max) --s->pipeline_left;   // sitting out on the pipeline path (see smplx_space)
    s->inflight_small = takes_small_kernel(s, B);
    s->inflight_zero_copy = s->inflight_small;
    s->t_issue = std::chrono::steady_clock::now();
    // K5: the states committed since the last batch join the device table at the head of this batch's first kernel;
    // their (id, coordinate) triples ride in the parents' upload
    K5Out k5;
    k5.d_id = s->dv.id;
    size_t item_doubles = 0;
    if (s->d_table) {
        if ((e = table_grow_if_needed(s))) return e;
        if (!s->pending_ins.empty()) {
            if ((e = s->p_q.reserve((size_t)B * N + s->pending_ins.size() / 2 + 1))) return e;
            if ((e = s->b_q.reserve((size_t)B * N + s->pending_ins.size() / 2 + 1))) return e;
            pack_parents();   // the buffer may have moved
            k5.n_items = (int)(s->pending_ins.size() / ((size_t)N + 2));
            item_doubles = stage_items(s->p_q, (size_t)B * N, s->pending_ins);
            s->pending_ins.clear();
        }
    }
    if (s->inflight_zero_copy) {
        k5.items = (const int32_t*)(s->p_q.p + (size_t)B * N);
        ZeroCopy zc;
        zc.q = s->p_q.p; zc.flags = s->pv.flags; zc.coord = s->pv.coord; zc.sq = s->pv.sq; zc.h = s->pv.h; zc.id = s->pv.id;
        if ((e = launch_expand(s, s->b_q.p, B, s->dv.flags, s->dv.coord, s->dv.sq, s->dv.h, s->b_cost.p, s->b_lookups.p,
                               s->b_work.p, s->b_counters.p, s->stream, nullptr, nullptr, &zc, false, &k5))) return e;
        HIP_TRY(hipEventRecord(s->batch_done, s->stream));
        ++s->gpu_batches;
        return SMPLX_OK;
    }
    HIP_TRY(hipMemcpyAsync(s->b_q.p, s->p_q.p, sizeof(double) * ((size_t)B * N + item_doubles), hipMemcpyHostToDevice, s->stream));
    k5.items = (const int32_t*)(s->b_q.p + (size_t)B * N);
    if ((e = launch_expand(s, s->b_q.p, B, s->dv.flags, s->dv.coord, s->dv.sq, s->dv.h, s->b_cost.p, s->b_lookups.p,
                           s->b_work.p, s->b_counters.p, s->stream, nullptr, nullptr, nullptr, false, &k5))) return e;
    HIP_TRY(hipMemcpyAsync(s->p_out.p, s->b_out.p, out_bytes, hipMemcpyDeviceToHost, s->stream));   // one copy for all five outputs
    HIP_TRY(hipEventRecord(s->batch_done, s->stream));
    ++s->gpu_batches;
    return SMPLX_OK;
}

// one dense output row -> cached successor records (appended to recs); returns the record count
int ingest_row(smplx_space* s, const OutView& pv, size_t row, int* evals_out)
{
    const int N = s->N, M = s->M;
    // the flags first (25 bytes): how many records, then ONE growth of each array and plain copies into it
    int cnt = 0, evals = 0;
    const unsigned char* fl = &pv.flags[row * M];
    for (int p = 0; p < M; ++p) {
        evals += (fl[p] & SMPLX_F_INACTIVE) ? 0 : 1;
        cnt += (fl[p] & SMPLX_F_VALID) ? 1 : 0;
    }
    *evals_out = evals;
    if (cnt == 0) return 0;
    const size_t r0 = s->recs.size();
    s->recs.resize(r0 + cnt);
    s->rec_coord.resize((r0 + cnt) * (size_t)N);
    s->rec_q.resize((r0 + cnt) * (size_t)N);
    size_t r = r0;
    for (int p = 0; p < M; ++p) {
        const unsigned char f = fl[p];
        if (!(f & SMPLX_F_VALID)) continue;
        const size_t k = row * M + p;
        smplx_space::Rec& rec = s->recs[r];
        rec.cost = s->actions.dev.cost[p];
        rec.h = pv.h[k];
        rec.goal = (f & SMPLX_F_GOAL) ? 1 : 0;
        rec.known = s->d_table ? pv.id[k] : -1;
        rec.prim = p;
        std::memcpy(&s->rec_coord[r * N], &pv.coord[k * N], sizeof(int32_t) * N);
        std::memcpy(&s->rec_q[r * N], &pv.sq[k * N], sizeof(double) * N);
        ++r;
    }
    return cnt;
}

// the batch in flight has completed: turn its dense outputs into cached successor records
int collect_batch(smplx_space* s, const smplx_space* src = nullptr, size_t first = 0, const OutView* view = nullptr)
{
    if (!src) src = s;   // a cross-query batch lands in the leading space's buffers (or in `view`), at row `first`
    const OutView& pv = view ? *view : src->pv;
    const std::vector<int32_t>& batch = s->inflight;
    const int B = (int)batch.size();
    if (src == s && s->inflight_small && s->adaptive_small && B <= 16) {
        // issue-to-landing time of the single-launch path (the search thread has been polling since the issue); only the
        // handful-of-states batches are watched: a batch of hundreds of states legitimately takes longer
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - s->t_issue).count();
        s->small_latency = s->small_seen == 0 ? dt : 0.8 * s->small_latency + 0.2 * dt;
        if (++s->small_seen >= 16 && s->small_latency > s->small_latency_limit) { s->pipeline_left = 2000; s->small_seen = 0; }
    }
    if (src == s) { s->inflight_small = false; s->inflight_zero_copy = false; }
    for (int i = 0; i < B; ++i) {
        const int sid = batch[i];
        s->cache_off[sid] = (int64_t)s->recs.size();
        int evals = 0;
        s->cache_cnt[sid] = ingest_row(s, pv, first + (size_t)i, &evals);
        s->eval_count[sid] = evals;
        s->gpu_evals += evals;
    }
    s->inflight.clear();
    return SMPLX_OK;
}

int run_batch(smplx_space* s, int id)
{
    s->adaptive_small = true;   // synchronous: the landing time is the GPU's
    if (int e = issue_batch(s, id)) return e;
    if (int e = wait_event_polling(s->batch_done)) return e;
    return collect_batch(s);
}

// GetSuccs (manip_lattice.cpp:219-313): ids are assigned here, in the caller's sequential order
int get_succs(smplx_space* s, int id, const int32_t** succs, const int32_t** costs, int* n)
{
    if (id == 0) { *n = 0; *succs = nullptr; *costs = nullptr; return SMPLX_OK; }   // goal is absorbing (:231)
    if (id < 0 || id >= (int)s->cache_off.size()) return set_error(SMPLX_E_STATE, "unknown state id");
    if (s->done_off[id] < 0) {
        if (s->cache_off[id] < 0) {
            ++s->cache_misses;   // plain GetSuccs callers (the unchanged ARA* of smpl): synchronous batch
            if (s->plain_mode && s->hint.empty() && s->auto_spec > 0) auto_hint(s, id);
            if (int e = run_batch(s, id)) return e;
        } else {
            ++s->cache_hits;
        }
        const int64_t off = s->cache_off[id];
        const int cnt = s->cache_cnt[id];
        const int64_t dof = (int64_t)s->done_succ.size();
        // With many queries per core the tables live in DRAM: a lookup is two dependent misses (slot, coordinate row).
        // The hashes of all records first, their slots prefetched together, then the rows the slots name: the ~14
        // lookups of an expansion overlap instead of queueing (commit is the host's largest share of an expansion).
        uint64_t hashes[SMPLX_MAX_PRIMS];
        const int npre = cnt <= SMPLX_MAX_PRIMS ? cnt : 0;
        for (int k = 0; k < npre; ++k) {
            hashes[k] = CoordTable::hash(&s->rec_coord[(size_t)(off + k) * s->N], s->N);
            s->table.prefetch_slot(hashes[k]);
        }
        for (int k = 0; k < npre; ++k) s->table.prefetch_row(hashes[k], s->coords);
        for (int k = 0; k < cnt; ++k) {
            const smplx_space::Rec r = s->recs[off + k];
            const int32_t* c = &s->rec_coord[(size_t)(off + k) * s->N];
            // K5: the device table already named the state when the batch was evaluated (it only holds committed
            // states, so a hit is final); otherwise getOrCreateState on the host table
            int sid = r.known >= 0 ? r.known : (k < npre ? s->table.find_hashed(c, hashes[k], s->coords) : s->table.find(c, s->coords));
            if (sid < 0) {
                sid = new_state(s, c, &s->rec_q[(size_t)(off + k) * s->N], r.h);
            }
            s->done_succ.push_back(r.goal ? 0 : sid);
            s->done_cost.push_back(r.cost);
            s->done_prim.push_back(r.prim);
        }
        s->done_off[id] = dof;
        s->done_cnt[id] = cnt;
    }
    // every GetSuccs call of the reference runs the whole loop body again (a state re-expanded in a later ARA*
    // iteration is re-evaluated, manip_lattice.cpp:263-305); here the repeat is served from the committed list, but it
    // counts as the same number of successor evaluations, so that the figure compares with the CPU planner's
    s->committed_evals += s->eval_count[id];
    if (s->plain_mode) {
        // the caller is expanding `id` now: mirror its g-updates and (re)rank the successors not yet evaluated
        const uint32_t gp = s->g_est[id];
        for (int k = 0; k < s->done_cnt[id]; ++k) {
            const int sid = s->done_succ[s->done_off[id] + k];
            if (sid == 0 || gp >= 1000000000u) continue;
            const uint32_t g = gp + (uint32_t)s->done_cost[s->done_off[id] + k];
            if (g < s->g_est[sid]) {
                s->g_est[sid] = g;
                if (s->cache_off[sid] == -1 && s->done_off[sid] < 0) pool_push(s, sid);
            }
        }
    }
    *n = s->done_cnt[id];
    *succs = s->done_succ.data() + s->done_off[id];
    *costs = s->done_cost.data() + s->done_off[id];
    return SMPLX_OK;
}

// host mirror of applyMotionPrimitive (kernels.hip; manip_lattice_action_space.cpp:575-621) -- the same expressions in the
// same order, compiled with -ffp-contract=off like the kernels: bit-identical successor joint values
void host_apply_prim(const SmplxActionsDev& A, const double* parent, int pi, int nv, double* out)
{
    double d0 = A.delta[pi][0], d1 = nv > 1 ? A.delta[pi][1] : 0.0;
    if (A.xy_rotate_by_var3 && nv > 3) {
        double sn, cs;
        smplx_sincos(parent[3], &sn, &cs);
        const double a0 = d0, a1 = d1;
        d0 = cs * a0 + (-sn) * a1;
        d1 = sn * a0 + cs * a1;
    }
    for (int v = 0; v < nv; ++v) {
        const double d = v == 0 ? d0 : (v == 1 ? d1 : A.delta[pi][v]);
        out[v] = d + parent[v];
    }
}

#include "search_host.h"

}  // namespace

// =================================================================================================
// C-ABI
// =================================================================================================

extern "C" {

const char* smplx_last_error(void) { return g_error.c_str(); }

int smplx_shard_range(int rank, int world, int total, int per_rank, int* first, int* count)
{
    if (!first || !count || world <= 0 || rank < 0 || rank >= world || total < 0 || per_rank <= 0) return set_error(SMPLX_E_ARG, "bad shard arguments");
    const long long f = (long long)rank * per_rank;
    *first = (int)std::min<long long>(f, total);
    *count = (int)std::max<long long>(0, std::min<long long>(f + per_rank, total) - *first);
    return SMPLX_OK;
}

int smplx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int smplx_grid_create(const double origin[3], int nx, int ny, int nz, double res, double max_dist, const int32_t* d2,
                      smplx_grid** out)
{
    if (!origin || !d2 || !out || nx <= 0 || ny <= 0 || nz <= 0 || !(res > 0.0)) return set_error(SMPLX_E_ARG, "bad grid arguments");
    smplx_grid* g = new smplx_grid;
    const double inv_res = 1.0 / res;
    g->dmax_int = (int)std::ceil(max_dist * inv_res);   // distance_map.hpp:126
    g->dmax_sqrd = g->dmax_int * g->dmax_int;
    if (g->dmax_sqrd > 65535) { delete g; return set_error(SMPLX_E_LIMIT, "max_dist/res exceeds 255 cells (16-bit squared distances)"); }
    g->res = res; g->max_dist = max_dist;
    g->n[0] = nx; g->n[1] = ny; g->n[2] = nz;
    const int bx = (nx + 3) / 4, by = (ny + 3) / 4, bz = (nz + 3) / 4;
    std::vector<uint16_t> tiled((size_t)bx * by * bz * 64, 0);
    for (int x = 0; x < nx; ++x)
        for (int y = 0; y < ny; ++y)
            for (int z = 0; z < nz; ++z) {
                const int v = d2[((size_t)x * ny + y) * nz + z];
                if (v < 0 || v > g->dmax_sqrd) { delete g; return set_error(SMPLX_E_ARG, "squared distance outside [0, dmax^2]"); }
                const size_t brick = ((size_t)(x >> 2) * by + (y >> 2)) * bz + (z >> 2);
                tiled[brick * 64 + ((x & 3) << 4) + ((y & 3) << 2) + (z & 3)] = (uint16_t)v;
            }
    hipError_t e = hipMalloc((void**)&g->d_d2, tiled.size() * sizeof(uint16_t));
    if (e == hipSuccess) e = hipMemcpy(g->d_d2, tiled.data(), tiled.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) { delete g; return set_error(SMPLX_E_HIP, std::string("grid upload: ") + hipGetErrorString(e)); }
    for (int a = 0; a < 3; ++a) { g->origin[a] = origin[a]; g->dev.origin_minus_res[a] = origin[a] - res; g->dev.n[a] = g->n[a]; }
    g->dev.res = res; g->dev.inv_res = inv_res;
    g->dev.bricks[0] = bx; g->dev.bricks[1] = by; g->dev.bricks[2] = bz;
    g->dev.dmax_sqrd = g->dmax_sqrd; g->dev.pad = 0;
    g->dev.d2 = g->d_d2;
    *out = g;
    return SMPLX_OK;
}

void smplx_grid_destroy(smplx_grid* g)
{
    if (!g) return;
    if (g->d_d2) (void)hipFree(g->d_d2);
    if (g->d_occ) (void)hipFree(g->d_occ);
    if (g->d_tmp) (void)hipFree(g->d_tmp);
    if (g->d_counts) (void)hipFree(g->d_counts);
    delete g;
}

// error text for the other translation units of the library (field.hip)
int smplx_internal_set_error(int code, const char* msg) { return set_error(code, msg ? msg : ""); }

int smplx_model_create(const char* robot_text, smplx_model** out)
{
    if (!robot_text || !out) return set_error(SMPLX_E_ARG, "null argument");
    smplx_model* m = new smplx_model;
    if (!smplx::compile_robot_text(robot_text, m->hm)) {
        const std::string err = m->hm.error;
        delete m;
        return set_error(err.find("too many") != std::string::npos ? SMPLX_E_LIMIT : SMPLX_E_PARSE, err);
    }
    *out = m;
    return SMPLX_OK;
}

void smplx_model_destroy(smplx_model* m) { delete m; }

int smplx_model_counts(const smplx_model* m, int* njoints, int* nvars, int* ntrees, int* nnodes, int* npairs, int* nslots)
{
    if (!m) return set_error(SMPLX_E_ARG, "null model");
    const SmplxModelDev& d = m->hm.dev;
    if (njoints) *njoints = d.njoints;
    if (nvars) *nvars = d.nvars;
    if (ntrees) *ntrees = d.ntrees;
    if (nnodes) *nnodes = d.nnodes;
    if (npairs) *npairs = d.npairs;
    if (nslots) *nslots = d.nslots;
    return SMPLX_OK;
}

int smplx_model_joints(const smplx_model* m, double* origins, double* k, int* file_index)
{
    if (!m) return set_error(SMPLX_E_ARG, "null model");
    const SmplxModelDev& d = m->hm.dev;
    for (int j = 0; j < d.njoints; ++j) {
        if (origins) std::memcpy(origins + 12 * (size_t)j, d.joints[j].origin, sizeof(double) * 12);
        if (k) k[j] = m->hm.joint_k[j];
        if (file_index) file_index[j] = m->hm.file_joint_index[j];
    }
    return SMPLX_OK;
}

int smplx_model_nodes(const smplx_model* m, double* xyzr, int* left, int* right, int* tree_first)
{
    if (!m) return set_error(SMPLX_E_ARG, "null model");
    const SmplxModelDev& d = m->hm.dev;
    for (int i = 0; i < d.nnodes; ++i) {
        if (xyzr) { xyzr[4 * i] = d.nodes[i].c[0]; xyzr[4 * i + 1] = d.nodes[i].c[1]; xyzr[4 * i + 2] = d.nodes[i].c[2]; xyzr[4 * i + 3] = d.nodes[i].r; }
        if (left) left[i] = d.nodes[i].left;
        if (right) right[i] = d.nodes[i].right;
    }
    if (tree_first) for (int t = 0; t <= d.ntrees; ++t) tree_first[t] = d.tree_first[t];
    return SMPLX_OK;
}

int smplx_model_pairs(const smplx_model* m, int* pairs)
{
    if (!m || !pairs) return set_error(SMPLX_E_ARG, "null argument");
    const SmplxModelDev& d = m->hm.dev;
    for (int i = 0; i < d.npairs; ++i) { pairs[2 * i] = d.pair_a[i]; pairs[2 * i + 1] = d.pair_b[i]; }
    return SMPLX_OK;
}

int smplx_model_const_header(const smplx_model* m, char* out, int cap)
{
    if (!m) return set_error(SMPLX_E_ARG, "null argument");
    const std::string h = smplx::model_const_header(m->hm.dev);
    if (out && cap > 0) { std::strncpy(out, h.c_str(), cap - 1); out[cap - 1] = 0; }
    return (int)h.size() + 1;
}

int smplx_space_create(const smplx_model* model, const smplx_grid* grid, const char* mprim_text, const smplx_params* params,
                       smplx_space** out)
{
    if (!model || !grid || !mprim_text || !params || !out) return set_error(SMPLX_E_ARG, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return set_error(SMPLX_E_HIP, "no HIP device: the engine has no CPU path");
    smplx_space* s = new smplx_space;
    s->model = model->hm;
    s->grid = grid;
    s->params = *params;
    s->fused_mode = (params->flags & SMPLX_SPACE_FUSED) != 0;
    if (params->flags & SMPLX_SPACE_NO_SMALL_KERNEL) s->small_batch_max = 0;
    if (const char* e = getenv("SMPLX_AUTO_SPECULATE")) s->auto_spec = std::max(0, atoi(e));
    s->N = s->model.dev.nvars;
    if (!smplx::load_mprim_text(mprim_text, params->resolutions, s->N, s->actions)) {
        const std::string err = s->actions.error;
        delete s;
        return set_error(SMPLX_E_PARSE, "mprim: " + err);
    }
    s->M = s->actions.dev.nprims;
    SmplxActionsDev& A = s->actions.dev;
    A.use_long_and_short = params->use_long_and_short;
    A.xy_rotate_by_var3 = params->xy_rotate_by_var3;
    // defaults of ManipLatticeActionSpace::init (manip_lattice_action_space.cpp:75-84), then the caller's overrides
    for (int i = 0; i < 4; ++i) { A.enabled[i] = 0; A.thresh[i] = 0.4; }
    A.enabled[SMPLX_MP_SHORT] = params->use_short_dist_mprims;
    A.thresh[SMPLX_MP_SHORT] = params->short_dist_mprims_thresh;
    A.enabled[SMPLX_MP_SNAP_XYZ_RPY] = params->use_xyzrpy_snap_mprim;
    A.thresh[SMPLX_MP_SNAP_XYZ_RPY] = params->xyzrpy_snap_dist_thresh;
    smplx::fill_discretization(s->model.dev, params->resolutions);
    for (int i = 0; i < s->model.dev.nnodes; ++i)
        s->model.dev.nodes[i].thr = smplx::sphere_threshold(s->model.dev.nodes[i].r, params->padding, grid->res, grid->dmax_sqrd);
    s->wall_thr = smplx::wall_threshold(params->bfs_inflation_radius, grid->res, grid->dmax_sqrd);
    std::memset(&s->hs, 0, sizeof(s->hs));
    s->hs.model = s->model.dev;
    s->blob_bytes = smplx::pack_model_blob(s->model.dev, s->hs.model_blob, sizeof(s->hs.model_blob));
    if (s->blob_bytes == 0) { delete s; return set_error(SMPLX_E_LIMIT, "model does not fit the packed LDS image"); }
    s->hs.grid = grid->dev;
    s->hs.actions = A;
    s->hs.goal.type = SMPLX_GOAL_JOINT;

    auto bail = [&](hipError_t e, const char* what) {
        const std::string msg = std::string(what) + ": " + hipGetErrorString(e);
        smplx_space_destroy(s);
        return set_error(SMPLX_E_HIP, msg);
    };
    hipError_t e;
    if ((e = hipGetDevice(&s->device)) != hipSuccess) return bail(e, "hipGetDevice");
    if ((e = hipStreamCreate(&s->stream)) != hipSuccess) return bail(e, "hipStreamCreate");
    {
        const char* env = getenv("SMPLX_SPECIALIZE");
        smplx::generic_kernels(s->ks);
        if (params->flags & SMPLX_SPACE_GENERIC_KERNELS) s->specialize_note = "disabled by SMPLX_SPACE_GENERIC_KERNELS";
        else if (env && env[0] == '0') s->specialize_note = "disabled by SMPLX_SPECIALIZE=0";
        else if (!smplx::specialized_kernels(s->model.dev, s->ks, s->specialize_note) && env && env[0] == '2') {
            // SMPLX_SPECIALIZE=2: the per-robot build is required
            const std::string msg = "kernel specialisation failed: " + s->specialize_note;
            smplx_space_destroy(s);
            return set_error(SMPLX_E_HIP, msg);
        }
    }
    s->lds_nroot = s->ks.specialized ? 0 : s->model.dev.nroot;
    s->lds_bytes = smplx_lds_bytes(s->blob_bytes, s->lds_nroot, s->model.dev.nslots, s->model.dev.nvars, s->model.dev.stack_bytes);
    s->lds_bytes_valid = smplx_lds_bytes(s->blob_bytes, s->lds_nroot, s->ks.specialized ? 0 : s->model.dev.nslots, s->model.dev.nvars, s->model.dev.stack_bytes);
    if (s->lds_bytes > 160 * 1024) { smplx_space_destroy(s); return set_error(SMPLX_E_LIMIT, "model needs more LDS per block than a CU has (160 KB)"); }
    if ((e = hipEventCreateWithFlags(&s->batch_done, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipMalloc((void**)&s->d_space, sizeof(SmplxSpaceDev))) != hipSuccess) return bail(e, "hipMalloc space");
    const int dx = grid->n[0] + 2, dy = grid->n[1] + 2, dz = grid->n[2] + 2;
    s->bfs_total = (int64_t)dx * dy * dz;
    for (int a = 0; a < 3; ++a) s->bfs_bricks[a] = (grid->n[a] + 7) / 8;
    const size_t nbricks = (size_t)s->bfs_bricks[0] * s->bfs_bricks[1] * s->bfs_bricks[2];
    s->bfs_ints = (int64_t)nbricks * SMPLX_BFS_REC;
    if ((e = hipMalloc((void**)&s->d_bfs, sizeof(int32_t) * s->bfs_ints)) != hipSuccess) return bail(e, "hipMalloc bfs");
    if ((e = hipMalloc((void**)&s->d_brick_queued, sizeof(int32_t) * 2 * nbricks + 64)) != hipSuccess) return bail(e, "hipMalloc bfs queued");
    if ((e = hipMemset(s->d_brick_queued, 0, sizeof(int32_t) * 2 * nbricks + 64)) != hipSuccess) return bail(e, "hipMemset bfs queued");
    if ((e = hipMalloc((void**)&s->d_queue, sizeof(int32_t) * (2 * 16 * nbricks + 64))) != hipSuccess) return bail(e, "hipMalloc bfs queue");
    if ((e = hipMalloc((void**)&s->d_counts, sizeof(int32_t) * (3 * 16 * 32 + kBfsHistory))) != hipSuccess) return bail(e, "hipMalloc bfs counts");
    if ((e = hipMalloc((void**)&s->d_minus_one, sizeof(int32_t))) != hipSuccess) return bail(e, "hipMalloc");
    { const int32_t m1 = -1; if ((e = hipMemcpy(s->d_minus_one, &m1, sizeof(m1), hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy"); }
    s->hs.bfs.dim_x = dx; s->hs.bfs.dim_y = dy; s->hs.bfs.dim_z = dz; s->hs.bfs.dim_xy = dx * dy;
    s->hs.bfs.cost_per_cell = params->cost_per_cell;
    s->hs.bfs.nbx = s->bfs_bricks[0]; s->hs.bfs.nby = s->bfs_bricks[1]; s->hs.bfs.nbz = s->bfs_bricks[2];
    s->hs.bfs.dist = s->d_bfs;
    s->hs.bfs.tag_mask = (int64_t)grid->n[0] * grid->n[1] * grid->n[2] < ((int64_t)1 << 28) ? (int32_t)0xF0000000u : 0;
    s->hs.bfs.tag_word = 0;
    // BfsHeuristic::syncGridAndBfs (bfs_heuristic.cpp:331-353), once, at init
    hipLaunchKernelGGL(k_bfs_init, dim3(2048), dim3(256), 0, s->stream, grid->dev, s->wall_thr, s->bfs_bricks[0], s->bfs_bricks[1], s->bfs_bricks[2], s->d_bfs);
    if ((e = hipGetLastError()) != hipSuccess) return bail(e, "k_bfs_init");
    {
        // Device copy of the state table (K5).  The K5 entry points and smplx_table_sync create it on first use.  The
        // planner's own batches use it only with SMPLX_DEVICE_TABLE=1: measured on MI355X (cfg 2 / cfg 4, bench.py) the
        // ids it hands back save the host ~0.5 us per expansion, but the lookups and the rides of the inserts add more
        // than that to the latency of every batch (single query 79k -> 70k states/s, 128-query shard 1.26M -> 1.03M).
        const char* env = getenv("SMPLX_DEVICE_TABLE");
        if (env && env[0] == '1' && table_alloc(s, (size_t)1 << 18) != SMPLX_OK) {
            const std::string m = g_error; smplx_space_destroy(s); return set_error(SMPLX_E_HIP, m);
        }
    }
    if (upload_space(s) != SMPLX_OK) { const std::string m = g_error; smplx_space_destroy(s); return set_error(SMPLX_E_HIP, m); }
    reset_lattice(s);
    *out = s;
    return SMPLX_OK;
}

void smplx_space_destroy(smplx_space* s)
{
    if (!s) return;
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    for (hipEvent_t e : s->prof_events) (void)hipEventDestroy(e);
    if (s->batch_done) (void)hipEventDestroy(s->batch_done);
    if (s->d_space) (void)hipFree(s->d_space);
    if (s->d_bfs) (void)hipFree(s->d_bfs);
    if (s->d_queue) (void)hipFree(s->d_queue);
    if (s->d_counts) (void)hipFree(s->d_counts);
    if (s->d_brick_queued) (void)hipFree(s->d_brick_queued);
    if (s->d_minus_one) (void)hipFree(s->d_minus_one);
    if (s->d_table) (void)hipFree(s->d_table);
    (void)search_free(s);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

int smplx_space_specialized(const smplx_space* s, char* note, int cap)
{
    if (!s) return 0;
    if (note && cap > 0) { std::strncpy(note, s->specialize_note.c_str(), cap - 1); note[cap - 1] = 0; }
    return s->ks.specialized ? 1 : 0;
}

int smplx_test_set_work_list_items(smplx_space* s, int items)
{
    if (!s || items < 0) return set_error(SMPLX_E_ARG, "bad argument");
    s->work_list_items = items / 8 * 8;
    return SMPLX_OK;
}

int smplx_test_set_search_helper(smplx_space* s, int on)
{
    if (!s) return set_error(SMPLX_E_ARG, "null space");
    s->ds.test_no_helper = on == 0;
    return SMPLX_OK;
}

int smplx_test_set_search_capacity(smplx_space* s, int states)
{
    if (!s || states < 0) return set_error(SMPLX_E_ARG, "bad argument");
    s->ds.test_capacity = states;
    return SMPLX_OK;
}

int smplx_search_counters(const smplx_space* s, int64_t out[16])
{
    if (!s || !out) return set_error(SMPLX_E_ARG, "null argument");
    for (int k = 0; k < 16; ++k) out[k] = 0;
    const DevSearch& D = s->ds;
    out[0] = D.searches; out[1] = D.grows; out[2] = D.dup_pushes;
    for (int k = 0; k < 7; ++k) out[3 + k] = D.ticks[k];
    out[10] = D.h.nstates;
    out[11] = search_heap_cache_entries(s, nullptr);
    out[12] = D.ticks[7] >> 32;            // evaluation rounds opened on a guess of the next pop
    out[13] = D.ticks[7] & 0xFFFFFFFFll;   // ... that the pop confirmed
    return SMPLX_OK;
}

int smplx_test_heap_ops(const int32_t* ops, int nops, int lds_entries, int32_t* top_after)
{
    if (!ops || !top_after || nops <= 0 || lds_entries < 1 || lds_entries > 4096) return set_error(SMPLX_E_ARG, "bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return set_error(SMPLX_E_HIP, "no HIP device");
    DevBuf<int32_t> d_ops, d_top;
    DevBuf<unsigned long long> d_heap;
    DevBuf<SmplxSState> d_st;
    int e;
    if ((e = d_ops.reserve(2 * (size_t)nops))) return e;
    if ((e = d_top.reserve((size_t)nops))) return e;
    if ((e = d_heap.reserve((size_t)nops + 2))) return e;
    if ((e = d_st.reserve((size_t)nops + 1))) return e;
    HIP_TRY(hipMemcpy(d_ops.p, ops, sizeof(int32_t) * 2 * (size_t)nops, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(d_st.p, 0, sizeof(SmplxSState) * ((size_t)nops + 1)));
    hipLaunchKernelGGL(k_heap_ops, dim3(1), dim3(64), (size_t)lds_entries * 8, 0, d_ops.p, nops, lds_entries, d_heap.p, d_st.p, d_top.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(top_after, d_top.p, sizeof(int32_t) * (size_t)nops, hipMemcpyDeviceToHost));
    return SMPLX_OK;
}

int smplx_space_num_vars(const smplx_space* s) { return s ? s->N : 0; }
int smplx_space_num_prims(const smplx_space* s) { return s ? s->M : 0; }

int smplx_space_discretization(const smplx_space* s, int32_t* coord_vals, double* coord_deltas)
{
    if (!s) return set_error(SMPLX_E_ARG, "null space");
    for (int v = 0; v < s->N; ++v) {
        if (coord_vals) coord_vals[v] = s->model.dev.coord_vals[v];
        if (coord_deltas) coord_deltas[v] = s->model.dev.coord_delta[v];
    }
    return SMPLX_OK;
}

int smplx_check_joint_limits(const smplx_space* s, const double* q, int n, uint8_t* ok)
{
    if (!s || !q || !ok || n < 0) return set_error(SMPLX_E_ARG, "bad argument");
    if (!sane_values(q, (size_t)n * s->N)) return set_error(SMPLX_E_ARG, "joint values must be finite (|q| < 1e6)");
    for (int i = 0; i < n; ++i) ok[i] = host_check_limits(s->model.dev, q + (size_t)i * s->N) ? 1 : 0;
    return SMPLX_OK;
}

int smplx_cc_state_valid_batch(smplx_space* s, const double* q, int n, uint8_t* valid, int32_t* lookups)
{
    if (!s || !q || !valid || n < 0) return set_error(SMPLX_E_ARG, "bad argument");
    if (n == 0) return SMPLX_OK;
    if (!sane_values(q, (size_t)n * s->N)) return set_error(SMPLX_E_ARG, "joint values must be finite (|q| < 1e6)");
    int e;
    if ((e = s->b_q.reserve((size_t)n * s->N))) return e;
    if ((e = s->b_flags.reserve(n))) return e;
    if ((e = s->b_lookups.reserve(n))) return e;
    HIP_TRY(hipMemcpyAsync(s->b_q.p, q, sizeof(double) * n * s->N, hipMemcpyHostToDevice, s->stream));
    KLAUNCH(s, K_STATE_VALID, k_state_valid, dim3(blocks_for(n, SMPLX_BLOCK)), dim3(SMPLX_BLOCK), s->lds_bytes_valid, s->stream, s->d_space,
                       s->b_q.p, n, s->b_flags.p, s->b_lookups.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(valid, s->b_flags.p, n, hipMemcpyDeviceToHost, s->stream));
    if (lookups) HIP_TRY(hipMemcpyAsync(lookups, s->b_lookups.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return SMPLX_OK;
}

int smplx_cc_state_valid_batch_device(smplx_space* s, const double* d_q, int n, uint8_t* d_valid, int32_t* d_lookups, void* stream)
{
    if (!s || !d_q || !d_valid || n <= 0) return set_error(SMPLX_E_ARG, "bad argument");
    KLAUNCH(s, K_STATE_VALID, k_state_valid, dim3(blocks_for(n, SMPLX_BLOCK)), dim3(SMPLX_BLOCK), s->lds_bytes_valid, (hipStream_t)stream,
            s->d_space, d_q, n, d_valid, d_lookups);
    HIP_TRY(hipGetLastError());
    return SMPLX_OK;
}

int smplx_cc_edge_valid_batch(smplx_space* s, const double* a, const double* b, int n, uint8_t* valid, int32_t* lookups,
                              int32_t* waypoints)
{
    if (!s || !a || !b || !valid || n < 0) return set_error(SMPLX_E_ARG, "bad argument");
    if (n == 0) return SMPLX_OK;
    if (!sane_values(a, (size_t)n * s->N) || !sane_values(b, (size_t)n * s->N))
        return set_error(SMPLX_E_ARG, "joint values must be finite (|q| < 1e6)");
    int e;
    if ((e = s->b_q.reserve((size_t)n * s->N))) return e;
    if ((e = s->b_q2.reserve((size_t)n * s->N))) return e;
    if ((e = s->b_flags.reserve(n))) return e;
    if ((e = s->b_lookups.reserve(n))) return e;
    if ((e = s->b_way.reserve(n))) return e;
    HIP_TRY(hipMemcpyAsync(s->b_q.p, a, sizeof(double) * n * s->N, hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipMemcpyAsync(s->b_q2.p, b, sizeof(double) * n * s->N, hipMemcpyHostToDevice, s->stream));
    KLAUNCH(s, K_EDGE_VALID, k_edge_valid, dim3(blocks_for(n, SMPLX_BLOCK)), dim3(SMPLX_BLOCK), s->lds_bytes_valid, s->stream, s->d_space,
                       s->b_q.p, s->b_q2.p, n, s->b_flags.p, s->b_lookups.p, s->b_way.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(valid, s->b_flags.p, n, hipMemcpyDeviceToHost, s->stream));
    if (lookups) HIP_TRY(hipMemcpyAsync(lookups, s->b_lookups.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, s->stream));
    if (waypoints) HIP_TRY(hipMemcpyAsync(waypoints, s->b_way.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return SMPLX_OK;
}

int smplx_cc_interpolate(smplx_space* s, const double* a, const double* b, double* out, int cap, int* n)
{
    if (!s || !a || !b || !n) return set_error(SMPLX_E_ARG, "bad argument");
    if (!sane_values(a, s->N) || !sane_values(b, s->N)) return set_error(SMPLX_E_ARG, "joint values must be finite (|q| < 1e6)");
    // waypoint count and interpolation are host arithmetic on det_math (collision_space.cpp:583-640,
    // robot_motion_collision_model.h:297-320); no grid or sphere data is involved
    const SmplxModelDev& M = s->model.dev;
    double motion = 0.0;
    for (int v = 0; v < M.nvars; ++v) {
        if (M.var_type[v] == SMPLX_JT_CONTINUOUS) motion += M.var_k[v] * std::fabs(smplx_shortest_angle_diff(b[v], a[v]));
        else if (M.var_type[v] == SMPLX_JT_REVOLUTE) motion += M.var_k[v] * std::fabs(b[v] - a[v]);
        else if (M.var_type[v] == SMPLX_JT_PRISMATIC) motion += std::fabs(b[v] - a[v]);
    }
    int W = 0;
    if (motion != 0.0) W = std::max(2, (int)std::ceil(motion / 0.05) + 1);
    *n = W;
    if (!out) return SMPLX_OK;
    const double inv = W > 0 ? 1.0 / (double)(W - 1) : 0.0;
    for (int w = 0; w < W && w < cap; ++w) {
        const double alpha = (double)w * inv;
        for (int v = 0; v < M.nvars; ++v) {
            const double d = M.var_type[v] == SMPLX_JT_CONTINUOUS ? smplx_shortest_angle_diff(b[v], a[v]) : b[v] - a[v];
            out[(size_t)w * M.nvars + v] = a[v] + alpha * d;
        }
    }
    return SMPLX_OK;
}

int smplx_cc_sphere_positions(smplx_space* s, const double* q, int n, double* out)
{
    if (!s || !q || !out || n < 0) return set_error(SMPLX_E_ARG, "bad argument");
    if (n == 0) return SMPLX_OK;
    if (!sane_values(q, (size_t)n * s->N)) return set_error(SMPLX_E_ARG, "joint values must be finite (|q| < 1e6)");
    int e;
    const size_t cnt = (size_t)n * s->model.dev.nnodes * 3;
    if ((e = s->b_q.reserve((size_t)n * s->N))) return e;
    if ((e = s->b_sq.reserve(cnt))) return e;
    HIP_TRY(hipMemcpyAsync(s->b_q.p, q, sizeof(double) * n * s->N, hipMemcpyHostToDevice, s->stream));
    KLAUNCH(s, K_SPHERE_POSITIONS, k_sphere_positions, dim3(blocks_for(n, SMPLX_BLOCK)), dim3(SMPLX_BLOCK), s->lds_bytes, s->stream,
                       s->d_space, s->b_q.p, n, s->b_sq.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, s->b_sq.p, sizeof(double) * cnt, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return SMPLX_OK;
}

static int finish_goal(smplx_space* s)
{
    for (int a = 0; a < 3; ++a) s->hs.goal.xyz[a] = s->goal_xyz[a];
    // the tag of this goal's BFS run (device_types.h SmplxBfsDev): 1..7, a reset of the records when they wrap
    if (s->hs.bfs.tag_mask != 0) {
        s->bfs_reset_due = s->bfs_tag == 7;
        s->bfs_tag = s->bfs_tag % 7 + 1;
        s->hs.bfs.tag_word = s->bfs_tag << 28;
    } else {
        s->bfs_reset_due = s->bfs_tag != 0;
        s->bfs_tag = 1;
        s->hs.bfs.tag_word = 0;
    }
    if (int e = upload_space(s)) return e;
    if (int e = run_bfs(s, s->goal_xyz)) return e;
    s->goal_set = true;
    s->grid_epoch = s->grid->epoch;
    // a new goal starts a new query: the state table restarts (ids are per query)
    reset_lattice(s);
    // heuristic of the goal id = BFS cost at the goal pose's cell (manip_lattice.cpp:1176-1190)
    const smplx_grid* g = s->grid;
    int c[3];
    for (int a = 0; a < 3; ++a) c[a] = (int)(g->dev.inv_res * (s->goal_xyz[a] - g->dev.origin_minus_res[a]) + 0.5) - 1;
    const bool in_bounds = !(c[0] < 0 || c[1] < 0 || c[2] < 0 || c[0] >= g->n[0] || c[1] >= g->n[1] || c[2] >= g->n[2]);
    s->h_of_id[0] = in_bounds ? 0 : 32767;   // the seeded cell has distance 0 (bfs3d.cpp:178)
    return SMPLX_OK;
}

int smplx_set_goal_joint(smplx_space* s, const double* angles, const double* tolerances)
{
    if (!s || !angles || !tolerances) return set_error(SMPLX_E_ARG, "null argument");
    if (!sane_values(angles, s->N)) return set_error(SMPLX_E_ARG, "goal angles must be finite (|q| < 1e6)");
    SmplxGoalDev& G = s->hs.goal;
    G.type = SMPLX_GOAL_JOINT;
    for (int v = 0; v < s->N; ++v) { G.angles[v] = angles[v]; G.angle_tol[v] = tolerances[v]; }
    state_to_coord(s->model.dev, angles, G.coord);
    // goal pose = planning-link FK of the goal angles (planner_interface.cpp:1232-1235)
    int32_t h;
    if (int e = run_heuristic(s, angles, 1, &h, s->goal_xyz)) return e;
    return finish_goal(s);
}

int smplx_set_goal_xyz(smplx_space* s, const double xyz[3], const double tol[3])
{
    if (!s || !xyz || !tol) return set_error(SMPLX_E_ARG, "null argument");
    if (!sane_values(xyz, 3)) return set_error(SMPLX_E_ARG, "goal position must be finite");
    SmplxGoalDev& G = s->hs.goal;
    G.type = SMPLX_GOAL_XYZ;
    for (int a = 0; a < 3; ++a) { s->goal_xyz[a] = xyz[a]; G.xyz_tol[a] = tol[a]; }
    return finish_goal(s);
}

int smplx_goal_pose(const smplx_space* s, double xyz[3])
{
    if (!s || !xyz) return set_error(SMPLX_E_ARG, "null argument");
    for (int a = 0; a < 3; ++a) xyz[a] = s->goal_xyz[a];
    return SMPLX_OK;
}

int smplx_heuristic_batch(smplx_space* s, const double* q, int n, int32_t* h, double* xyz)
{
    if (!s || !q || n < 0) return set_error(SMPLX_E_ARG, "bad argument");
    if (n == 0) return SMPLX_OK;
    if (!sane_values(q, (size_t)n * s->N)) return set_error(SMPLX_E_ARG, "joint values must be finite (|q| < 1e6)");
    return run_heuristic(s, q, n, h, xyz);
}

int64_t smplx_bfs_size(const smplx_space* s) { return s ? s->bfs_total : 0; }
int smplx_bfs_levels(const smplx_space* s) { return s ? s->bfs_levels : 0; }

int smplx_bfs_copy(smplx_space* s, int32_t* out)
{
    if (!s || !out) return set_error(SMPLX_E_ARG, "null argument");
    // the device keeps brick-major records (device_types.h SmplxBfsDev); what goes out is the reference's padded array
    int32_t* tmp = nullptr;
    HIP_TRY(hipMalloc((void**)&tmp, sizeof(int32_t) * s->bfs_total));
    hipLaunchKernelGGL(k_bfs_export, dim3(2048), dim3(256), 0, s->stream, s->hs.bfs, tmp);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, tmp, sizeof(int32_t) * s->bfs_total, hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    (void)hipFree(tmp);
    if (e != hipSuccess) return set_error(SMPLX_E_HIP, std::string("smplx_bfs_copy: ") + hipGetErrorString(e));
    return SMPLX_OK;
}

int smplx_bfs_metric_goal_distance(smplx_space* s, const double* xyz, int n, double* out)
{
    if (!s || !xyz || !out || n < 0) return set_error(SMPLX_E_ARG, "bad argument");
    if (!s->goal_set) return set_error(SMPLX_E_STATE, "goal not set");
    if (n == 0) return SMPLX_OK;
    if (!sane_values(xyz, (size_t)n * 3)) return set_error(SMPLX_E_ARG, "positions must be finite");
    int e;
    if ((e = s->b_xyz.reserve((size_t)n * 3))) return e;
    if ((e = s->b_q2.reserve(n))) return e;
    HIP_TRY(hipMemcpyAsync(s->b_xyz.p, xyz, sizeof(double) * n * 3, hipMemcpyHostToDevice, s->stream));
    hipLaunchKernelGGL(k_bfs_metric, dim3(blocks_for(n, SMPLX_BLOCK)), dim3(SMPLX_BLOCK), 0, s->stream, s->hs.grid, s->hs.bfs, s->b_xyz.p, n,
                       s->b_q2.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, s->b_q2.p, sizeof(double) * n, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return SMPLX_OK;
}

int smplx_bfs_metric_start_distance(smplx_space* s, const double* xyz, int n, double* out)
{
    if (!s || !xyz || !out || n < 0) return set_error(SMPLX_E_ARG, "bad argument");
    if (s->start_id < 0) return set_error(SMPLX_E_STATE, "start not set");
    // bfs_heuristic.cpp:103-127: Manhattan distance in cells between the start's planning-link cell and the point's
    const smplx_grid* g = s->grid;
    auto cell = [&](const double* p, int c[3]) {
        for (int a = 0; a < 3; ++a) c[a] = (int)(g->dev.inv_res * (p[a] - g->dev.origin_minus_res[a]) + 0.5) - 1;
    };
    int sc[3];
    cell(s->start_xyz, sc);
    for (int i = 0; i < n; ++i) {
        int c[3];
        cell(xyz + 3 * (size_t)i, c);
        out[i] = g->res * (double)(std::abs(sc[0] - c[0]) + std::abs(sc[1] - c[1]) + std::abs(sc[2] - c[2]));
    }
    return SMPLX_OK;
}

int smplx_space_status(const smplx_space* s, char* msg, int cap)
{
    if (!s) return SMPLX_E_ARG;
    if (msg && cap > 0) { std::strncpy(msg, s->status_msg.c_str(), cap - 1); msg[cap - 1] = 0; }
    return s->status;
}

void smplx_space_clear_status(smplx_space* s)
{
    if (!s) return;
    s->status = SMPLX_OK;
    s->status_msg.clear();
}

int smplx_expand_batch(smplx_space* s, const double* q, int B, uint8_t* flags, int32_t* coord, double* succ_q, int32_t* h,
                       int32_t* cost, int32_t* lookups)
{
    if (!s || !q || B < 0) return set_error(SMPLX_E_ARG, "bad argument");
    if (!s->goal_set) return set_error(SMPLX_E_STATE, "set a goal first (the primitives are gated by goal distance)");
    if (B == 0) return SMPLX_OK;
    if (!sane_values(q, (size_t)B * s->N)) return set_error(SMPLX_E_ARG, "joint values must be finite (|q| < 1e6)");
    if (int e = reserve_expand(s, B)) return e;
    const size_t BM = (size_t)B * s->M;
    HIP_TRY(hipMemsetAsync(s->b_coord.p, 0, sizeof(int32_t) * BM * s->N, s->stream));
    HIP_TRY(hipMemsetAsync(s->b_sq.p, 0, sizeof(double) * BM * s->N, s->stream));
    HIP_TRY(hipMemcpyAsync(s->b_q.p, q, sizeof(double) * B * s->N, hipMemcpyHostToDevice, s->stream));
    if (int e = launch_expand(s, s->b_q.p, B, s->b_flags.p, s->b_coord.p, s->b_sq.p, s->b_h.p, s->b_cost.p, s->b_lookups.p,
                              s->b_work.p, nullptr, s->stream)) return e;
    if (flags) HIP_TRY(hipMemcpyAsync(flags, s->b_flags.p, BM, hipMemcpyDeviceToHost, s->stream));
    if (coord) HIP_TRY(hipMemcpyAsync(coord, s->b_coord.p, sizeof(int32_t) * BM * s->N, hipMemcpyDeviceToHost, s->stream));
    if (succ_q) HIP_TRY(hipMemcpyAsync(succ_q, s->b_sq.p, sizeof(double) * BM * s->N, hipMemcpyDeviceToHost, s->stream));
    if (h) HIP_TRY(hipMemcpyAsync(h, s->b_h.p, sizeof(int32_t) * BM, hipMemcpyDeviceToHost, s->stream));
    if (cost) HIP_TRY(hipMemcpyAsync(cost, s->b_cost.p, sizeof(int32_t) * BM, hipMemcpyDeviceToHost, s->stream));
    if (lookups) HIP_TRY(hipMemcpyAsync(lookups, s->b_lookups.p, sizeof(int32_t) * BM, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return SMPLX_OK;
}

size_t smplx_expand_work_bytes(const smplx_space* s, int B)
{
    if (!s || B <= 0) return 0;
    return expand_work_bytes(B, s->M);
}

int smplx_expand_batch_device(smplx_space* s, const double* d_q, int B, uint8_t* d_flags, int32_t* d_coord, double* d_succ_q,
                              int32_t* d_h, int32_t* d_cost, int32_t* d_lookups, void* d_work, uint64_t* d_counters, void* stream)
{
    if (!s || !d_q || !d_flags || !d_coord || !d_succ_q || !d_h || !d_cost || !d_lookups || !d_work || B <= 0)
        return set_error(SMPLX_E_ARG, "bad argument");
    if (!s->goal_set) return set_error(SMPLX_E_STATE, "set a goal first");
    return launch_expand(s, d_q, B, d_flags, d_coord, d_succ_q, d_h, d_cost, d_lookups, d_work,
                         (unsigned long long*)d_counters, (hipStream_t)stream);
}

int smplx_table_sync(smplx_space* s)
{
    if (!s) return set_error(SMPLX_E_ARG, "null argument");
    if (int e = table_ensure(s)) return e;
    if (int e = table_flush(s, s->stream)) return e;
    HIP_TRY(hipStreamSynchronize(s->stream));
    return SMPLX_OK;
}

size_t smplx_compact_rec_b_bytes(const smplx_space* s) { return s ? (size_t)smplx_rec_b_bytes(s->N) : 0; }
int smplx_compact_blocks(const smplx_space* s, int B) { return s && B > 0 ? blocks_for((long long)B * s->M, SMPLX_BLOCK) : 0; }
int smplx_compact_totals_len(void) { return SMPLX_CMP_TOTALS; }
int smplx_compact_capacity(const smplx_space* s, int B)
{
    if (!s || B <= 0) return 0;
    const int nblocks = blocks_for((long long)B * s->M, SMPLX_BLOCK);
    return SMPLX_CMP_SHARDS * ((nblocks + SMPLX_CMP_SHARDS - 1) / SMPLX_CMP_SHARDS) * SMPLX_BLOCK;   // no sub-region can overflow
}

int smplx_expand_batch_k5_device(smplx_space* s, const double* d_q, int B, uint8_t* d_flags, int32_t* d_coord, double* d_succ_q,
                                 int32_t* d_h, int32_t* d_cost, int32_t* d_lookups, int32_t* d_succ_id, int32_t* d_rec_a, int cap_a,
                                 void* d_rec_b, int cap_b, int32_t* d_block_tab, int32_t* d_totals, void* d_work,
                                 uint64_t* d_counters, void* stream)
{
    if (!s || !d_q || !d_flags || !d_coord || !d_succ_q || !d_h || !d_cost || !d_lookups || !d_work || B <= 0)
        return set_error(SMPLX_E_ARG, "bad argument");
    if (d_rec_a && (!d_rec_b || !d_block_tab || !d_totals || cap_a < SMPLX_CMP_SHARDS || cap_b < SMPLX_CMP_SHARDS))
        return set_error(SMPLX_E_ARG, "incomplete compact-stream arguments");
    cap_a = cap_a / SMPLX_CMP_SHARDS * SMPLX_CMP_SHARDS;
    cap_b = cap_b / SMPLX_CMP_SHARDS * SMPLX_CMP_SHARDS;
    if (!s->goal_set) return set_error(SMPLX_E_STATE, "set a goal first");
    if (s->fused_mode && d_rec_a) return set_error(SMPLX_E_STATE, "the compact stream needs the pipeline kernels (not fused mode)");
    SmplxCompactDev cmp;
    std::memset(&cmp, 0, sizeof(cmp));
    cmp.rec_a = d_rec_a; cmp.rec_b = (unsigned char*)d_rec_b; cmp.block_tab = d_block_tab; cmp.totals = d_totals;
    cmp.cap_a = cap_a; cmp.cap_b = cap_b; cmp.rec_b_bytes = smplx_rec_b_bytes(s->N);
    K5Out k5;
    k5.d_id = d_succ_id;
    k5.cmp = d_rec_a ? &cmp : nullptr;
    return launch_expand(s, d_q, B, d_flags, d_coord, d_succ_q, d_h, d_cost, d_lookups, d_work, (unsigned long long*)d_counters,
                         (hipStream_t)stream, nullptr, nullptr, nullptr, false, &k5);
}

int smplx_expand_batch_k5(smplx_space* s, const double* q, int B, uint8_t* flags, int32_t* coord, double* succ_q, int32_t* h,
                          int32_t* succ_id, int32_t* rec_a, int cap_a, void* rec_b, int cap_b, int32_t* block_tab, int32_t totals[3])
{
    if (!s || !q || B <= 0 || !rec_a || !rec_b || !block_tab || !totals || cap_a <= 0 || cap_b <= 0) return set_error(SMPLX_E_ARG, "bad argument");
    if (!s->goal_set) return set_error(SMPLX_E_STATE, "set a goal first");
    if (!sane_values(q, (size_t)B * s->N)) return set_error(SMPLX_E_ARG, "joint values must be finite (|q| < 1e6)");
    if (int e = reserve_expand(s, B)) return e;
    const size_t BM = (size_t)B * s->M;
    const size_t rb = (size_t)smplx_rec_b_bytes(s->N);
    const int nblocks = blocks_for((long long)BM, SMPLX_BLOCK);
    int e;
    if ((e = s->b_way.reserve(BM))) return e;                                       // dense ids
    if ((e = s->b_ins.reserve(2 * (size_t)cap_a + 4 * (size_t)nblocks + SMPLX_CMP_TOTALS))) return e;   // A records | block table | totals
    if ((e = s->b_out.reserve(rb * (size_t)cap_b))) return e;                       // B records
    if ((e = table_ensure(s))) return e;
    if ((e = table_flush(s, s->stream))) return e;
    int32_t* d_a = s->b_ins.p;
    int32_t* d_bt = d_a + 2 * (size_t)cap_a;
    int32_t* d_tot = d_bt + 4 * (size_t)nblocks;
    HIP_TRY(hipMemcpyAsync(s->b_q.p, q, sizeof(double) * B * s->N, hipMemcpyHostToDevice, s->stream));
    if ((e = smplx_expand_batch_k5_device(s, s->b_q.p, B, s->b_flags.p, s->b_coord.p, s->b_sq.p, s->b_h.p, s->b_cost.p, s->b_lookups.p,
                                          s->b_way.p, d_a, cap_a, s->b_out.p, cap_b, d_bt, d_tot, s->b_work.p, nullptr, s->stream))) return e;
    if (flags) HIP_TRY(hipMemcpyAsync(flags, s->b_flags.p, BM, hipMemcpyDeviceToHost, s->stream));
    if (coord) HIP_TRY(hipMemcpyAsync(coord, s->b_coord.p, sizeof(int32_t) * BM * s->N, hipMemcpyDeviceToHost, s->stream));
    if (succ_q) HIP_TRY(hipMemcpyAsync(succ_q, s->b_sq.p, sizeof(double) * BM * s->N, hipMemcpyDeviceToHost, s->stream));
    if (h) HIP_TRY(hipMemcpyAsync(h, s->b_h.p, sizeof(int32_t) * BM, hipMemcpyDeviceToHost, s->stream));
    if (succ_id) HIP_TRY(hipMemcpyAsync(succ_id, s->b_way.p, sizeof(int32_t) * BM, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipMemcpyAsync(rec_a, d_a, sizeof(int32_t) * 2 * (size_t)cap_a, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipMemcpyAsync(rec_b, s->b_out.p, rb * (size_t)cap_b, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipMemcpyAsync(block_tab, d_bt, sizeof(int32_t) * 4 * (size_t)nblocks, hipMemcpyDeviceToHost, s->stream));
    int32_t raw[SMPLX_CMP_TOTALS];
    HIP_TRY(hipMemcpyAsync(raw, d_tot, sizeof(raw), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    totals[0] = totals[1] = 0;
    for (int k = 0; k < SMPLX_CMP_SHARDS; ++k) { totals[0] += raw[32 * k]; totals[1] += raw[32 * k + 1]; }
    totals[2] = raw[32 * SMPLX_CMP_SHARDS];
    return SMPLX_OK;
}

size_t smplx_counters_bytes(const smplx_space* s, int B)
{
    if (!s || B <= 0) return 0;
    return counter_words(B, s->M) * sizeof(unsigned long long);
}

int smplx_counters_read(const smplx_space* s, const uint64_t* d_counters, int B, uint64_t out[6])
{
    if (!s || !d_counters || !out || B <= 0) return set_error(SMPLX_E_ARG, "bad argument");
    const size_t cw = counter_words(B, s->M);
    std::vector<unsigned long long> part(cw);
    HIP_TRY(hipMemcpy(part.data(), d_counters, sizeof(unsigned long long) * cw, hipMemcpyDeviceToHost));
    for (int k = 0; k < SMPLX_TALLIES; ++k) out[k] = 0;
    for (size_t i = 0; i < cw; ++i) out[i % SMPLX_TALLIES] += part[i];
    return SMPLX_OK;
}

int smplx_profile_begin(smplx_space* s, int max_launches)
{
    if (!s || max_launches < 0) return set_error(SMPLX_E_ARG, "bad argument");
    for (hipEvent_t e : s->prof_events) (void)hipEventDestroy(e);
    s->prof_events.clear();
    s->prof_used = 0;
    for (int i = 0; i < 3 * max_launches; ++i) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        s->prof_events.push_back(e);
    }
    return SMPLX_OK;
}

int smplx_profile_end(smplx_space* s, double* prep_ms, double* expand_ms, int* launches)
{
    if (!s || !prep_ms || !expand_ms || !launches) return set_error(SMPLX_E_ARG, "null argument");
    double a = 0.0, b = 0.0;
    const int n = (int)(s->prof_used / 3);
    for (int i = 0; i < n; ++i) {
        float t = 0.f;
        HIP_TRY(hipEventSynchronize(s->prof_events[3 * i + 2]));
        HIP_TRY(hipEventElapsedTime(&t, s->prof_events[3 * i], s->prof_events[3 * i + 1]));
        a += t;
        HIP_TRY(hipEventElapsedTime(&t, s->prof_events[3 * i + 1], s->prof_events[3 * i + 2]));
        b += t;
    }
    *prep_ms = a; *expand_ms = b; *launches = n;
    for (hipEvent_t e : s->prof_events) (void)hipEventDestroy(e);
    s->prof_events.clear();
    s->prof_used = 0;
    return SMPLX_OK;
}

int smplx_set_start(smplx_space* s, const double* q, int* id)
{
    if (!s || !q) return set_error(SMPLX_E_ARG, "null argument");
    if (!s->goal_set) return set_error(SMPLX_E_STATE, "set the goal before the start (planner_interface.cpp:1469-1500 order)");
    if (!sane_values(q, s->N)) return set_error(SMPLX_E_ARG, "joint values must be finite (|q| < 1e6)");
    if (!host_check_limits(s->model.dev, q)) return set_error(SMPLX_E_INVALID, "start state violates joint limits");
    if (int e = pull_lattice(s)) return e;
    uint8_t ok = 0;
    if (int e = smplx_cc_state_valid_batch(s, q, 1, &ok, nullptr)) return e;
    if (!ok) return set_error(SMPLX_E_INVALID, "start state is in collision");
    std::vector<int32_t> c(s->N);
    state_to_coord(s->model.dev, q, c.data());
    int sid = s->table.find(c.data(), s->coords);
    int32_t h = 0;
    if (int e = run_heuristic(s, q, 1, &h, s->start_xyz)) return e;   // also the start's planning-link position
    if (sid < 0) sid = new_state(s, c.data(), q, h);
    s->start_id = sid;
    if (s->plain_mode) { s->g_est.assign(s->h_of_id.size(), 1000000000u); s->g_est[sid] = 0; s->pool.clear(); }
    if (id) *id = sid;
    return SMPLX_OK;
}

int smplx_start_id(const smplx_space* s) { return s ? s->start_id : -1; }
int smplx_goal_id(const smplx_space* s) { (void)s; return 0; }

int smplx_get_succs(smplx_space* s, int id, int32_t* succs, int32_t* costs, int cap, int* n)
{
    if (!s || !n) return set_error(SMPLX_E_ARG, "null argument");
    if (!s->goal_set) return set_error(SMPLX_E_STATE, "goal not set");
    if (s->grid->epoch != s->grid_epoch) return set_error(SMPLX_E_STATE, "the grid was edited after the goal was set: cached successors are stale, set the goal again");
    if (int e = pull_lattice(s)) return e;
    if (!s->plain_mode) {
        // first GetSuccs from outside: start mirroring the caller's g-values (the start has g = 0, arastar.cpp:172-176)
        s->plain_mode = true;
        s->g_est.assign(s->h_of_id.size(), 1000000000u);
        if (s->start_id > 0) s->g_est[s->start_id] = 0;
        s->pool.clear();
    }
    const int32_t *ps, *pc;
    int cnt = 0;
    if (int e = get_succs(s, id, &ps, &pc, &cnt)) {
        // the SBPL-side caller (GetSuccs has no return value) sees an empty list; the error stays readable here
        if (s->status == SMPLX_OK) { s->status = e; s->status_msg = g_error; }
        return e;
    }
    *n = cnt;
    for (int i = 0; i < cnt && i < cap; ++i) {
        if (succs) succs[i] = ps[i];
        if (costs) costs[i] = pc[i];
    }
    return SMPLX_OK;
}

int smplx_hint_frontier(smplx_space* s, const int32_t* ids, int n)
{
    if (!s || (!ids && n > 0)) return set_error(SMPLX_E_ARG, "null argument");
    if (int e = pull_lattice(s)) return e;
    s->hint.assign(ids, ids + n);
    return SMPLX_OK;
}

int smplx_get_goal_heuristic(smplx_space* s, int id, int32_t* h)
{
    if (!s || !h) return set_error(SMPLX_E_ARG, "null argument");
    if (int e = pull_lattice(s)) return e;
    if (id < 0 || id >= (int)s->h_of_id.size()) return set_error(SMPLX_E_STATE, "unknown state id");
    *h = s->h_of_id[id];
    return SMPLX_OK;
}

int smplx_num_states(const smplx_space* s)
{
    if (!s) return 0;
    if (s->ds.host_behind) return s->ds.h.nstates;     // the device-resident search created states the host has not fetched yet
    return (int)s->h_of_id.size();
}

int smplx_space_counters(const smplx_space* s, int64_t out[6])
{
    if (!s || !out) return set_error(SMPLX_E_ARG, "null argument");
    out[0] = s->gpu_batches; out[1] = s->cache_hits; out[2] = s->cache_misses; out[3] = s->committed_evals;
    out[4] = s->gpu_evals; out[5] = (int64_t)smplx_num_states(s);
    return SMPLX_OK;
}

int smplx_get_state(const smplx_space* cs, int id, double* q, int32_t* coord)
{
    if (!cs) return set_error(SMPLX_E_ARG, "null argument");
    smplx_space* s = const_cast<smplx_space*>(cs);      // (fetching what the device created does not change the lattice)
    if (int e = pull_lattice(s)) return e;
    if (id < 0 || id >= (int)s->h_of_id.size()) return set_error(SMPLX_E_STATE, "unknown state id");
    if (q) std::memcpy(q, &s->qs[(size_t)id * s->N], sizeof(double) * s->N);
    if (coord) std::memcpy(coord, &s->coords[(size_t)id * s->N], sizeof(int32_t) * s->N);
    return SMPLX_OK;
}

// -------------------------------------------------------------------------------------------------
// ARA* -- the caller (smpl/src/search/arastar.cpp).  Sequential and bit-faithful: OPEN is the
// intrusive binary heap of smpl/include/smpl/detail/intrusive_heap.hpp (strict '<' sift rules), keys are
// g + (unsigned)(eps*h).  The only addition is the frontier hint before a cache miss.
// -------------------------------------------------------------------------------------------------

namespace {

const unsigned int kInfiniteCost = 1000000000u;   // SBPL INFINITECOST

struct SearchState {
    unsigned int g, h, f, eg;
    unsigned short iteration_closed, call_number;
    int bp;
    int heap_index;
    bool incons;
    bool made;
};

struct Search {
    smplx_space* sp;
    std::vector<SearchState> st;
    std::vector<int> heap;      // heap[0] unused
    std::vector<int> incons;
    double curr_eps = 1.0, initial_eps = 1.0, final_eps = 1.0, delta_eps = 1.0;
    bool improve = true, bounded = false;
    int max_init = 0, max_rep = 0;
    int iteration = 1, call_number = 0;
    double satisfied_eps = std::numeric_limits<double>::infinity();
    int expand_count = 0, expand_count_init = 0;
    int start_id = -1, goal_id = 0;
    int error = SMPLX_OK;

    bool less(int a, int b) const { return st[a].f < st[b].f; }
    bool heap_empty() const { return heap.size() == 1; }
    void heap_clear() { for (size_t i = 1; i < heap.size(); ++i) st[heap[i]].heap_index = 0; heap.resize(1); }
    void percolate_down(size_t pivot)   // intrusive_heap.hpp:346-377
    {
        if (pivot >= heap.size()) return;
        size_t left = pivot << 1, right = (pivot << 1) + 1;
        const int tmp = heap[pivot];
        while (left < heap.size()) {
            size_t c = right;
            if (right >= heap.size() || less(heap[left], heap[right])) c = left;
            if (less(heap[c], tmp)) {
                heap[pivot] = heap[c];
                st[heap[pivot]].heap_index = (int)pivot;
                pivot = c;
            } else break;
            left = pivot << 1; right = (pivot << 1) + 1;
        }
        heap[pivot] = tmp;
        st[tmp].heap_index = (int)pivot;
    }
    void percolate_up(size_t pivot)     // intrusive_heap.hpp:379-395
    {
        const int tmp = heap[pivot];
        while (pivot != 1) {
            const size_t p = pivot >> 1;
            if (less(heap[p], tmp)) break;
            heap[pivot] = heap[p];
            st[heap[pivot]].heap_index = (int)pivot;
            pivot = p;
        }
        heap[pivot] = tmp;
        st[tmp].heap_index = (int)pivot;
    }
    void push(int e) { st[e].heap_index = (int)heap.size(); heap.push_back(e); percolate_up(heap.size() - 1); }
    void pop()
    {
        st[heap[1]].heap_index = 0;
        heap[1] = heap.back();
        heap.pop_back();
        percolate_down(1);
    }
    void make() { for (size_t i = (heap.size() - 1) >> 1; i >= 1; --i) percolate_down(i); }

    SearchState& get(int id)
    {
        if ((int)st.size() <= id) {
            SearchState z;
            std::memset(&z, 0, sizeof(z));
            st.resize(id + 1, z);
        }
        if (!st[id].made) { st[id].made = true; st[id].call_number = 0; st[id].heap_index = 0; }
        return st[id];
    }
    void reinit(int id)   // arastar.cpp:613-627
    {
        SearchState& s = get(id);
        if (s.call_number != (unsigned short)call_number) {
            int32_t h = 0;
            smplx_get_goal_heuristic(sp, id, &h);
            s.g = kInfiniteCost;
            s.h = (unsigned int)h;
            s.f = kInfiniteCost;
            s.eg = kInfiniteCost;
            s.iteration_closed = 0;
            s.call_number = (unsigned short)call_number;
            s.bp = -1;
            s.incons = false;
        }
    }
    unsigned int key(const SearchState& s) const { return s.g + (unsigned int)(long long)(curr_eps * s.h); }   // :579-582
    void reorder_open()
    {
        for (size_t i = 1; i < heap.size(); ++i) st[heap[i]].f = key(st[heap[i]]);
        make();
    }
    bool timed_out(int elapsed) const
    {
        if (!bounded) return false;
        if (satisfied_eps == std::numeric_limits<double>::infinity()) return elapsed >= max_init;
        return elapsed >= max_rep;
    }
    void expand(int sid)   // arastar.cpp:531-568
    {
        const int32_t *succs, *costs;
        int n = 0;
        error = get_succs(sp, sid, &succs, &costs, &n);   // served from the cache: improve_path checked ready(sid)
        if (error) return;
        // succs/costs point into the committed arrays, which only get_succs grows: nothing below calls it
        const int32_t* ss = succs;
        const int32_t* cc = costs;
        const unsigned int eg = st[sid].eg;
        for (int i = 0; i < n; ++i)
            if ((size_t)ss[i] < st.size()) __builtin_prefetch(&st[ss[i]]);   // the successors' search states, all misses at once
        for (int i = 0; i < n; ++i) {
            const int nid = ss[i];
            reinit(nid);
            SearchState& t = st[nid];
            const int new_cost = (int)(eg + (unsigned int)cc[i]);
            if ((unsigned int)new_cost < t.g) {
                t.g = (unsigned int)new_cost;
                t.bp = sid;
                if (t.iteration_closed != (unsigned short)iteration) {
                    t.f = key(t);
                    if (t.heap_index != 0) percolate_up(t.heap_index);
                    else push(nid);
                } else if (!t.incons) {
                    incons.push_back(nid);
                }
            }
        }
    }
    // states whose successors are already on the host (cached or committed); the goal never expands
    bool ready(int sid) const { return sid == 0 || sp->done_off[sid] >= 0 || sp->cache_off[sid] >= 0; }

    enum { R_DONE = 0, R_YIELD = 100 };
    int hint_scan = 1024;  // entries of OPEN's array examined for the hint of a miss (SMPLX_HINT_SCAN)
    int pause_after = 0;   // > 0: hand control back after that many expansions without a miss (miss_id = -1): keeps the
                           // rounds of the pipelined multi-query driver even; the search resumes at exactly this point
    int improve_path(int& elapsed)   // arastar.cpp:486-527; returns R_YIELD when a frontier batch was issued
    {
        int since_entry = 0;
        while (!heap_empty()) {
            const int m = heap[1];
            if (st[m].f >= st[goal_id].f || m == goal_id) return 0;
            if (timed_out(elapsed)) return 4;
            if (pause_after > 0 && since_entry >= pause_after) { miss_id = -1; return R_YIELD; }
            ++since_entry;
            if (!ready(m)) {
                // cache miss: the state and the top of OPEN go to the GPU as one frontier batch; the search
                // resumes from exactly this point when the batch has landed (nothing has been popped yet)
                // the hint: the states near the top of OPEN that have not been evaluated yet.  Only the first `hint_scan`
                // entries of the heap array are looked at (the array is only roughly sorted, and what sits deep in it is not
                // expanded soon): scanning all of a 30 000-entry OPEN on every miss cost the single-query search ~10 us per miss
                const int cap = sp->params.batch_states > 0 ? sp->params.batch_states : 4096;
                sp->hint.clear();
                const size_t scan_end = std::min(heap.size(), (size_t)2 + (size_t)hint_scan);
                for (size_t i = 2; i < scan_end && (int)sp->hint.size() < cap - 1; ++i) {
                    const int hid = heap[i];
                    if (hid > 0 && sp->cache_off[hid] == -1 && sp->done_off[hid] < 0) sp->hint.push_back(hid);
                }
                ++sp->cache_misses;
                miss_id = m;
                if (!defer_issue) {
                    error = issue_batch(sp, m);
                    if (error) return 99;
                }
                return R_YIELD;
            }
            pop();
            st[m].iteration_closed = (unsigned short)iteration;
            st[m].eg = st[m].g;
            sp->expansion_log.push_back(m);
            expand(m);
            if (error) return 99;
            ++elapsed;
        }
        return 5;
    }

    // arastar.cpp:107-215 (always from scratch) as a resumable state machine: resume() runs until the search
    // finishes (R_DONE) or a frontier batch is in flight (R_YIELD)
    int phase = 0, num = 0, err = 0, solved = 0, cost = 0;
    std::vector<int> solution;
    int resume()
    {
        if (phase == 0) {
            heap.assign(1, 0);
            incons.clear();
            ++call_number;
            reinit(start_id);
            reinit(goal_id);
            st[start_id].g = 0;
            st[start_id].f = key(st[start_id]);
            push(start_id);
            iteration = 1;
            curr_eps = initial_eps;
            satisfied_eps = std::numeric_limits<double>::infinity();
            // goal id "changed" on a fresh search: recompute h of existing states and reorder (:155-162)
            for (size_t i = 0; i < st.size(); ++i) {
                if (st[i].made) { int32_t h = 0; smplx_get_goal_heuristic(sp, (int)i, &h); st[i].h = (unsigned int)h; }
            }
            reorder_open();
            num = 0; err = 0;
            phase = 1;
        }
        while (true) {
            if (phase == 1) {
                if (!(satisfied_eps > final_eps)) break;
                if (curr_eps == satisfied_eps) {
                    if (!improve) break;
                    ++iteration;
                    curr_eps -= delta_eps;
                    curr_eps = std::max(curr_eps, final_eps);
                    for (int s : incons) { st[s].incons = false; push(s); }
                    reorder_open();
                    incons.clear();
                }
                phase = 2;
                num_before = num;
            }
            err = improve_path(num);
            if (err == R_YIELD) return R_YIELD;
            if (curr_eps == initial_eps) expand_count_init += num;
            phase = 1;
            if (err) break;
            satisfied_eps = curr_eps;
        }
        expand_count += num;
        phase = 3;
        if (satisfied_eps == std::numeric_limits<double>::infinity()) { solved = 0; return R_DONE; }
        for (int s = goal_id; s >= 0; s = st[s].bp) solution.push_back(s);
        std::reverse(solution.begin(), solution.end());
        cost = (int)st[goal_id].g;
        solved = 1;
        return R_DONE;
    }
    int num_before = 0;
    int miss_id = -1;
    bool defer_issue = false;   // cross-query batching: the caller gathers the misses of many queries into one launch
};

void fill_search(Search& S, smplx_space* s, const smplx_search_params* p)
{
    S.sp = s;
    S.initial_eps = p->initial_eps;
    S.final_eps = std::max(p->final_eps, 1.0);   // ARAStar::setTargetEpsilon (arastar.h:112-114)
    S.delta_eps = p->delta_eps;
    S.improve = p->improve != 0;
    S.bounded = p->bounded != 0;
    S.max_init = p->max_expansions_init;
    S.max_rep = p->max_expansions;
    S.start_id = s->start_id;
    S.goal_id = 0;
}

}  // namespace

// One slice [q0, q1) of a set of queries that share scene, robot and primitives, driven by the calling thread:
// every sweep runs each live query until it misses, gathers the misses into ONE cross-query frontier batch
// (per-state query index -> that query's goal and BFS grid), and hands the results back.
int run_group(smplx_space** spaces, Search* S, int q0, int q1, char* done, double* t_done,
              std::chrono::steady_clock::time_point t0)
{
    smplx_space* lead = spaces[q0];
    HIP_TRY(hipSetDevice(lead->device));   // a fresh host thread starts on device 0
    const int nq = q1 - q0;
    int remaining = nq;
        const int N = lead->N, M = lead->M;
        {
            std::vector<const SmplxSpaceDev*> tab(nq);
            for (int q = 0; q < nq; ++q) tab[q] = spaces[q0 + q]->d_space;
            if (int e = lead->b_stab.reserve(nq)) return e;
            HIP_TRY(hipMemcpy(lead->b_stab.p, tab.data(), sizeof(void*) * nq, hipMemcpyHostToDevice));
        }
        for (int q = q0; q < q1; ++q) S[q].defer_issue = true;
        // hinted frontier states per query and sweep: enough to keep a query fed, small enough that the dense
        // download of a sweep stays in the hundreds of kilobytes
        const int cap_q = std::max(16, std::min(512, (lead->params.batch_states > 0 ? lead->params.batch_states : 4096) / std::max(1, nq / 4)));
        std::vector<int> reqs;
        std::vector<int32_t> ins_items;
        const bool dbg = getenv("SMPLX_DEBUG_TIMING") != nullptr;
        double t_resume = 0, t_pack = 0, t_gpu = 0, t_collect = 0;
        long sweeps = 0, swept_states = 0;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
        while (remaining > 0) {
            reqs.clear();
            const auto tr0 = now();
            for (int q = q0; q < q1; ++q) {
                if (done[q]) continue;
                const int r = S[q].resume();
                if (S[q].error) return S[q].error;
                if (r == Search::R_YIELD) { reqs.push_back(q); continue; }
                done[q] = 1;
                --remaining;
                t_done[q] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            }
            const auto tr1 = now();
            t_resume += secs(tr0, tr1);
            if (reqs.empty()) break;
            size_t total = 0;
            for (int q : reqs) { select_batch(spaces[q], S[q].miss_id, cap_q); total += spaces[q]->inflight.size(); }
            ++sweeps; swept_states += (long)total;
            const int B = (int)total;
            const size_t BM = total * M;
            int e;
            if ((e = reserve_expand(lead, B))) return e;
            if ((e = lead->b_stateq.reserve(total))) return e;
            if ((e = lead->p_stateq.reserve(total))) return e;
            if ((e = lead->p_q.reserve(total * N))) return e;
            const size_t out_bytes = carve_out(nullptr, BM, N).bytes;
            if ((e = lead->b_out.reserve(out_bytes))) return e;
            if ((e = lead->p_out.reserve(out_bytes))) return e;
            lead->dv = carve_out(lead->b_out.p, BM, N);
            lead->pv = carve_out(lead->p_out.p, BM, N);
            size_t row = 0;
            for (int q : reqs) {
                const smplx_space* sq = spaces[q];
                for (int32_t id : sq->inflight) {
                    std::memcpy(&lead->p_q.p[row * N], &sq->qs[(size_t)id * N], sizeof(double) * N);
                    lead->p_stateq.p[row] = (unsigned short)(q - q0);
                    ++row;
                }
            }
            const auto tp1 = now();
            t_pack += secs(tr1, tp1);
            K5Out k5;
            k5.d_id = lead->dv.id;
            size_t item_doubles = 0;
            if (lead->d_table) {
                std::vector<int32_t>& items = ins_items;
                items.clear();
                for (int q : reqs) {
                    if ((e = table_grow_if_needed(spaces[q]))) return e;
                    table_take_pending(spaces[q], q - q0, items);
                }
                if (!items.empty()) {
                    if ((e = lead->p_q.reserve(total * N + items.size() / 2 + 1))) return e;   // (may move the buffer: packed below)
                    if ((e = lead->b_q.reserve(total * N + items.size() / 2 + 1))) return e;
                    k5.n_items = (int)(items.size() / ((size_t)N + 2));
                }
            }
            {   // parents (packed again here: the pinned buffer may just have been re-allocated)
                size_t r2 = 0;
                for (int q : reqs) {
                    const smplx_space* sq = spaces[q];
                    for (int32_t id : sq->inflight) { std::memcpy(&lead->p_q.p[r2 * N], &sq->qs[(size_t)id * N], sizeof(double) * N); ++r2; }
                }
                item_doubles = stage_items(lead->p_q, total * N, ins_items);
            }
            HIP_TRY(hipMemcpyAsync(lead->b_q.p, lead->p_q.p, sizeof(double) * (total * N + item_doubles), hipMemcpyHostToDevice, lead->stream));
            HIP_TRY(hipMemcpyAsync(lead->b_stateq.p, lead->p_stateq.p, sizeof(unsigned short) * total, hipMemcpyHostToDevice, lead->stream));
            k5.items = (const int32_t*)(lead->b_q.p + total * N);
            if ((e = launch_expand(lead, lead->b_q.p, B, lead->dv.flags, lead->dv.coord, lead->dv.sq, lead->dv.h, lead->b_cost.p,
                                   lead->b_lookups.p, lead->b_work.p, nullptr, lead->stream, lead->b_stab.p, lead->b_stateq.p,
                                   nullptr, false, &k5))) return e;
            HIP_TRY(hipMemcpyAsync(lead->p_out.p, lead->b_out.p, out_bytes, hipMemcpyDeviceToHost, lead->stream));
            HIP_TRY(hipStreamSynchronize(lead->stream));
            const auto tg1 = now();
            t_gpu += secs(tp1, tg1);
            ++lead->gpu_batches;
            row = 0;
            for (int q : reqs) {
                const size_t nb = spaces[q]->inflight.size();
                if ((e = collect_batch(spaces[q], lead, row))) return e;
                row += nb;
            }
            t_collect += secs(tg1, now());
        }
        if (dbg) fprintf(stderr, "[smplx timing] slice [%d,%d): %ld sweeps, %.1f states/sweep; search+commit %.3fs pack %.3fs gpu(issue..sync) %.3fs collect %.3fs\n",
                         q0, q1, sweeps, sweeps ? (double)swept_states / sweeps : 0.0, t_resume, t_pack, t_gpu, t_collect);
    return SMPLX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Asynchronous multi-query driver (host_threads > 1).  Measured on MI355X with the 128 queries of the config-4 shard:
// the sequential host work per expansion (heap, commit, hashing, record ingestion: about 4.7 us) outweighs the GPU time
// of a sweep 15:1 on one thread; several threads that each launch their own sweeps queue up behind each other in the
// runtime (8 threads: 800 us per sweep); and a round barrier between "all searches" and "one batch" makes every round
// as long as its slowest query.  So there are no rounds:
//   * T worker threads own the queries (static ownership: a query's heap and tables stay in one core's caches).  A worker
//     runs a query until it misses, leaves the request in the query's slot and turns to its next query; it makes no
//     HIP call at all.
//   * ONE submitter thread owns the GPU.  Whenever a buffer set is free it takes every request pending at that moment
//     into one cross-query frontier batch (per-state query index -> that query's goal and BFS grid), launches it and
//     moves on; up to kInFlight batches are in flight on their own streams, so the launch and copy overhead of one
//     hides behind the kernels of the other.  A landed batch is announced per query; the owner ingests it when it
//     comes round.
// Every query sees only its own successor records, in its own sequential order: results are those of a solo run.
// ---------------------------------------------------------------------------------------------------------------
struct BatchBuffers {
    DevBuf<double> b_q;
    DevBuf<unsigned short> b_stateq;
    DevBuf<unsigned char> b_work, b_out;
    DevBuf<int32_t> b_cost, b_lookups;
    PinBuf<double> p_q;
    PinBuf<unsigned short> p_stateq;
    PinBuf<unsigned char> p_out;
    DevBuf<int32_t> b_ins;
    PinBuf<int32_t> p_ins;
    std::vector<int32_t> ins_items;
    OutView dv, pv;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    std::atomic<int> uncollected{0};   // queries of the batch that landed in this set and have not been ingested yet
    std::vector<int> queries;          // the queries of the batch in flight
    bool in_flight = false;
    size_t total = 0;
};

enum { QS_RUNNABLE = 0, QS_REQUESTED = 1, QS_LANDED = 2, QS_IN_FLIGHT = 3 };

static inline void cpu_relax() { __builtin_ia32_pause(); }

int run_pipelined(smplx_space** spaces, Search* S, int nq, int nworkers, char* done, double* t_done,
                  std::chrono::steady_clock::time_point t0)
{
    enum { kSets = 8, kInFlight = 4 };
    const int small_zero_copy_max = 512;   // batches up to this size: one launch, results written straight to pinned host memory
    smplx_space* lead = spaces[0];
    const int N = lead->N, M = lead->M;
    // one cache line per query state and per counter: the submitter polls them while the workers write them (with the
    // states packed 16 to a line, a scan of all queries cost the submitter 15-50 us per batch and slowed every worker store)
    struct alignas(64) PaddedInt { std::atomic<int> v{0}; };
    std::vector<PaddedInt> qstate_store(nq);
    auto qstate = [&](int q) -> std::atomic<int>& { return qstate_store[q].v; };
    PaddedInt pend_cnt[8], live_cnt[8];   // per issue group: requests waiting / queries not finished
    std::vector<long> row_of(nq, -1);
    std::vector<int> set_of(nq, -1);
    std::atomic<int> remaining{nq}, error{0};
    std::string error_msg;
    const int pause_after = 16;   // expansions without a miss before a query hands its worker to the next one (measured flat between 4 and 1000)
    for (int q = 0; q < nq; ++q) { qstate(q).store(QS_RUNNABLE); S[q].defer_issue = true; S[q].pause_after = pause_after; }
    {
        std::vector<const SmplxSpaceDev*> tab(nq);
        for (int q = 0; q < nq; ++q) tab[q] = spaces[q]->d_space;
        if (int e = lead->b_stab.reserve(nq)) return e;
        HIP_TRY(hipMemcpy(lead->b_stab.p, tab.data(), sizeof(void*) * nq, hipMemcpyHostToDevice));
    }
    const int cap_q = std::max(16, std::min(512, (lead->params.batch_states > 0 ? lead->params.batch_states : 4096) / std::max(1, nq / 8)));
    const bool dbg = getenv("SMPLX_DEBUG_TIMING") != nullptr;
    const int device = lead->device;
    const int issue_percent = 45;   // a batch is issued when this share of the live queries waits (1 %: 9.9e5 states/s, 45 %: 1.22e6, 70 %: 1.17e6)
    const int groups = 1;           // (forming batches within 2-3 independent groups of queries was measured: 1.37-1.39e6 against 1.42e6)
    for (int q = 0; q < nq; ++q) live_cnt[(q / nworkers) % groups].v.fetch_add(1, std::memory_order_relaxed);
    std::vector<BatchBuffers> sets(kSets);

    auto fail = [&](int code, const std::string& msg) {
        int expect = 0;
        if (error.compare_exchange_strong(expect, code)) error_msg = msg;
    };

    // worker w owns the queries q with q % nworkers == w
    auto worker = [&](int w) {
        double t_work = 0, t_ingest = 0;
        long n_ingest = 0, n_resume = 0;
        const auto w_begin = std::chrono::steady_clock::now();
        while (remaining.load(std::memory_order_acquire) > 0 && error.load(std::memory_order_relaxed) == 0) {
            bool progressed = false;
            for (int q = w; q < nq; q += nworkers) {
                if (done[q]) continue;
                int st = qstate(q).load(std::memory_order_acquire);
                if (st == QS_REQUESTED || st == QS_IN_FLIGHT) continue;
                const auto a0 = std::chrono::steady_clock::now();
                if (st == QS_LANDED) {
                    BatchBuffers& Bf = sets[set_of[q]];
                    if (int e = collect_batch(spaces[q], lead, (size_t)row_of[q], &Bf.pv)) { fail(e, g_error); return; }
                    Bf.uncollected.fetch_sub(1, std::memory_order_acq_rel);
                    qstate(q).store(QS_RUNNABLE, std::memory_order_relaxed);
                    if (dbg) { t_ingest += std::chrono::duration<double>(std::chrono::steady_clock::now() - a0).count(); ++n_ingest; }
                }
                ++n_resume;
                const int r = S[q].resume();
                progressed = true;
                if (S[q].error) { fail(S[q].error, g_error); return; }
                if (r == Search::R_YIELD) {
                    if (S[q].miss_id >= 0) {
                        select_batch(spaces[q], S[q].miss_id, cap_q);
                        {   // stage the parents' joint values for the submitter
                            smplx_space* sq = spaces[q];
                            sq->inflight_q.resize(sq->inflight.size() * (size_t)N);
                            size_t r = 0;
                            for (int32_t id : sq->inflight) { std::memcpy(&sq->inflight_q[r * N], &sq->qs[(size_t)id * N], sizeof(double) * N); ++r; }
                        }
                        qstate(q).store(QS_REQUESTED, std::memory_order_release);
                        pend_cnt[(q / nworkers) % groups].v.fetch_add(1, std::memory_order_release);
                    }
                } else {
                    done[q] = 1;
                    t_done[q] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    live_cnt[(q / nworkers) % groups].v.fetch_sub(1, std::memory_order_acq_rel);
                    remaining.fetch_sub(1, std::memory_order_acq_rel);
                }
                t_work += std::chrono::duration<double>(std::chrono::steady_clock::now() - a0).count();
            }
            if (!progressed) cpu_relax();
        }
        if (dbg) {
            const double tot = std::chrono::duration<double>(std::chrono::steady_clock::now() - w_begin).count();
            fprintf(stderr, "[smplx timing] worker %d: search+commit+ingest %.3fs of %.3fs (ingest %.3fs in %ld landings; %ld resumes)\n", w, t_work, tot,
                    t_ingest, n_ingest, n_resume);
        }
    };

    auto submitter = [&]() -> int {
        HIP_TRY(hipSetDevice(device));
        for (BatchBuffers& Bf : sets) {
            HIP_TRY(hipStreamCreate(&Bf.stream));
            HIP_TRY(hipEventCreateWithFlags(&Bf.done, hipEventDisableTiming));
        }
        long sweeps = 0, states = 0;
        double t_issue = 0;
        int in_flight = 0, next_set = 0, oldest = 0;
        // SMPLX_DEBUG_TIMING: how long the GPU had nothing of this shard, issue-to-landing time, depth at issue
        double t_gpu_idle = 0, lat_sum = 0, t_pack = 0;
        long depth_sum = 0;
        auto idle_since = std::chrono::steady_clock::now();
        std::chrono::steady_clock::time_point issued_at[kSets];
        unsigned poll_spins = 0;
        while (remaining.load(std::memory_order_acquire) > 0 && error.load(std::memory_order_relaxed) == 0) {
            bool did = false;
            // retire landed batches in issue order
            while (in_flight > 0) {
                BatchBuffers& Bf = sets[oldest];
                const hipError_t st = hipEventQuery(Bf.done);
                if (st == hipErrorNotReady) {
                    // a batch takes well under a millisecond: one that has not landed after SMPLX_BATCH_TIMEOUT_S is a hung
                    // kernel; the workers leave through `error` and the call returns (include/smpl_amd.h: every function returns)
                    if ((++poll_spins & 0x3FFF) == 0 &&
                        std::chrono::duration<double>(std::chrono::steady_clock::now() - issued_at[oldest]).count() > batch_timeout_seconds())
                        return set_error(SMPLX_E_HIP, "frontier batch did not complete within SMPLX_BATCH_TIMEOUT_S: kernel hung?");
                    break;
                }
                if (st != hipSuccess) return set_error(SMPLX_E_HIP, std::string("hipEventQuery: ") + hipGetErrorString(st));
                Bf.uncollected.store((int)Bf.queries.size(), std::memory_order_relaxed);
                for (int q : Bf.queries) qstate(q).store(QS_LANDED, std::memory_order_release);
                Bf.in_flight = false;
                if (dbg) {
                    const auto nowt = std::chrono::steady_clock::now();
                    lat_sum += std::chrono::duration<double>(nowt - issued_at[oldest]).count();
                    if (in_flight == 1) idle_since = nowt;
                }
                oldest = (oldest + 1) % kSets;
                --in_flight;
                did = true;
            }
            // issue: every request pending right now, if a buffer set is free
            BatchBuffers& Nf = sets[next_set];
            if (in_flight < kInFlight && !Nf.in_flight && Nf.uncollected.load(std::memory_order_acquire) == 0) {
                const auto i0 = std::chrono::steady_clock::now();
                // A batch has a fixed cost (issuing ~15 us, ~40 us on the GPU whatever its size).  Taking every request the
                // moment it appears gives many small batches and a query then waits for several batch times per miss, so a batch
                // is issued when issue_percent (45 %) of the live queries are waiting; the two counters are kept by the workers.
                int live_g[8], pend_g[8];
                for (int g = 0; g < groups; ++g) {
                    live_g[g] = live_cnt[g].v.load(std::memory_order_acquire);
                    pend_g[g] = pend_cnt[g].v.load(std::memory_order_acquire);
                }
                // the group closest to its threshold (one group: every live query)
                int pick = -1;
                for (int g = 0; g < groups; ++g) {
                    if (live_g[g] == 0 || pend_g[g] == 0) continue;
                    if (pend_g[g] < std::max(1, (live_g[g] * issue_percent + 99) / 100)) continue;
                    if (pick < 0 || (long)pend_g[g] * live_g[pick] > (long)pend_g[pick] * live_g[g]) pick = g;
                }
                Nf.queries.clear();
                size_t total = 0;
                if (pick >= 0) {
                    for (int q = 0; q < nq; ++q) {
                        if ((q / nworkers) % groups != pick) continue;
                        if (qstate(q).load(std::memory_order_acquire) != QS_REQUESTED) continue;
                        row_of[q] = (long)total;
                        set_of[q] = next_set;
                        total += spaces[q]->inflight.size();
                        Nf.queries.push_back(q);
                    }
                    pend_cnt[pick].v.fetch_sub((int)Nf.queries.size(), std::memory_order_acq_rel);
                }
                if (total > 0) {
                    const int B = (int)total;
                    const size_t BM = total * M;
                    int e;
                    if ((e = Nf.b_q.reserve(total * N))) return e;
                    if ((e = Nf.b_work.reserve(expand_work_bytes(B, M)))) return e;
                    if ((e = Nf.b_cost.reserve(BM))) return e;
                    if ((e = Nf.b_lookups.reserve(BM))) return e;
                    if ((e = Nf.b_stateq.reserve(total))) return e;
                    if ((e = Nf.p_stateq.reserve(total))) return e;
                    if ((e = Nf.p_q.reserve(total * N))) return e;
                    const size_t out_bytes = carve_out(nullptr, BM, N).bytes;
                    if ((e = Nf.b_out.reserve(out_bytes))) return e;
                    if ((e = Nf.p_out.reserve(out_bytes))) return e;
                    Nf.dv = carve_out(Nf.b_out.p, BM, N);
                    Nf.pv = carve_out(Nf.p_out.p, BM, N);
                    size_t row = 0;
                    Nf.ins_items.clear();
                    for (int q : Nf.queries) {
                        smplx_space* sq = spaces[q];
                        // the parents' joint values were staged by the query's worker when it made the request (they were in
                        // its cache then; gathering 270 rows from 58 queries' state arrays here cost the submitter ~20 us
                        // of cache misses per batch)
                        const size_t nrows = sq->inflight.size();
                        std::memcpy(&Nf.p_q.p[row * N], sq->inflight_q.data(), sizeof(double) * N * nrows);
                        for (size_t k = 0; k < nrows; ++k) Nf.p_stateq.p[row + k] = (unsigned short)q;
                        row += nrows;
                        // a requesting query is not being touched by its worker: its committed states join the device table
                        if ((e = table_grow_if_needed(sq))) return e;
                        table_take_pending(sq, q, Nf.ins_items);
                        qstate(q).store(QS_IN_FLIGHT, std::memory_order_relaxed);
                    }
                    K5Out k5;
                    k5.d_id = Nf.dv.id;
                    size_t item_doubles = 0;
                    if (!Nf.ins_items.empty()) {
                        const size_t need = total * N + Nf.ins_items.size() / 2 + 1;
                        if (need > Nf.p_q.cap) {   // re-allocate and pack the parents again
                            if ((e = Nf.p_q.reserve(need))) return e;
                            size_t r2 = 0;
                            for (int q : Nf.queries)
                            {
                                const size_t nrows = spaces[q]->inflight.size();
                                std::memcpy(&Nf.p_q.p[r2 * N], spaces[q]->inflight_q.data(), sizeof(double) * N * nrows);
                                r2 += nrows;
                            }
                        }
                        if ((e = Nf.b_q.reserve(need))) return e;
                        k5.n_items = (int)(Nf.ins_items.size() / ((size_t)N + 2));
                        item_doubles = stage_items(Nf.p_q, total * N, Nf.ins_items);
                    }
                    // Batches of up to 512 states: ONE launch, results written straight to pinned host memory.  Against the
                    // pipeline (two uploads, four kernels, one download: seven runtime calls) the submitter spends 34
                    // instead of 48 us per batch and a batch lands after 78 instead of 113 us: shard +8..17 % (same box,
                    // A/B).  (Round 2 first measured the opposite -- 152 us per launch at ~100 states -- because the
                    // kernel then checked the snap-to-goal edge of every state ungated, see k_small_batch.)
                    if (B <= small_zero_copy_max && small_kernel_fits(lead, B) && lead->prof_events.empty()) {
                        // one launch, no copies: parents, query indices and results live in pinned host memory
                        if (dbg) t_pack += std::chrono::duration<double>(std::chrono::steady_clock::now() - i0).count();
                        ZeroCopy zc;
                        zc.q = Nf.p_q.p; zc.flags = Nf.pv.flags; zc.coord = Nf.pv.coord; zc.sq = Nf.pv.sq; zc.h = Nf.pv.h; zc.id = Nf.pv.id;
                        k5.items = (const int32_t*)(Nf.p_q.p + total * N);
                        if ((e = launch_expand(lead, Nf.b_q.p, B, Nf.dv.flags, Nf.dv.coord, Nf.dv.sq, Nf.dv.h, Nf.b_cost.p, Nf.b_lookups.p,
                                               Nf.b_work.p, nullptr, Nf.stream, lead->b_stab.p, Nf.p_stateq.p, &zc, false, &k5))) return e;
                    } else {
                        HIP_TRY(hipMemcpyAsync(Nf.b_q.p, Nf.p_q.p, sizeof(double) * (total * N + item_doubles), hipMemcpyHostToDevice, Nf.stream));
                        HIP_TRY(hipMemcpyAsync(Nf.b_stateq.p, Nf.p_stateq.p, sizeof(unsigned short) * total, hipMemcpyHostToDevice, Nf.stream));
                        k5.items = (const int32_t*)(Nf.b_q.p + total * N);
                        if ((e = launch_expand(lead, Nf.b_q.p, B, Nf.dv.flags, Nf.dv.coord, Nf.dv.sq, Nf.dv.h, Nf.b_cost.p, Nf.b_lookups.p,
                                               Nf.b_work.p, nullptr, Nf.stream, lead->b_stab.p, Nf.b_stateq.p, nullptr, true, &k5))) return e;
                        HIP_TRY(hipMemcpyAsync(Nf.p_out.p, Nf.b_out.p, out_bytes, hipMemcpyDeviceToHost, Nf.stream));
                    }
                    HIP_TRY(hipEventRecord(Nf.done, Nf.stream));
                    issued_at[next_set] = i0;
                    if (dbg) {
                        depth_sum += in_flight;
                        if (in_flight == 0) t_gpu_idle += std::chrono::duration<double>(i0 - idle_since).count();
                    }
                    Nf.in_flight = true;
                    Nf.total = total;
                    ++in_flight;
                    next_set = (next_set + 1) % kSets;
                    ++lead->gpu_batches;
                    ++sweeps; states += (long)total;
                    did = true;
                    t_issue += std::chrono::duration<double>(std::chrono::steady_clock::now() - i0).count();
                }
            }
            if (!did) cpu_relax();
        }
        // drain what is still in flight (only on error paths: with no live query nothing is pending)
        for (BatchBuffers& Bf : sets) if (Bf.stream) (void)hipStreamSynchronize(Bf.stream);
        if (dbg) fprintf(stderr, "[smplx timing] submitter: %ld batches, %.1f states/batch; issuing %.3fs (pack + enqueue); GPU without a batch %.3fs; "
                                 "issue-to-landing %.1f us on average; %.2f batches already in flight at issue; of the issuing, %.3fs before the launch call\n",
                         sweeps, sweeps ? (double)states / sweeps : 0.0, t_issue, t_gpu_idle, sweeps ? 1e6 * lat_sum / sweeps : 0.0,
                         sweeps ? (double)depth_sum / sweeps : 0.0, t_pack);
        return SMPLX_OK;
    };

    std::vector<std::thread> th;
    for (int w = 0; w < nworkers; ++w) th.emplace_back(worker, w);
    int rc = submitter();
    if (rc != SMPLX_OK) fail(rc, g_error);
    for (auto& x : th) x.join();
    for (BatchBuffers& Bf : sets) {
        if (Bf.stream) { (void)hipStreamSynchronize(Bf.stream); (void)hipStreamDestroy(Bf.stream); }
        if (Bf.done) (void)hipEventDestroy(Bf.done);
    }
    if (error.load() != 0) return set_error(error.load(), error_msg);
    return SMPLX_OK;
}

static int read_counters(smplx_space* s, size_t cw, unsigned long long counters[4])
{
    std::vector<unsigned long long> part(cw);
    HIP_TRY(hipMemcpy(part.data(), s->b_counters.p, sizeof(unsigned long long) * cw, hipMemcpyDeviceToHost));
    for (int k = 0; k < 4; ++k) counters[k] = 0;
    for (size_t i = 0; i < cw; ++i) if (i % SMPLX_TALLIES < 4) counters[i % SMPLX_TALLIES] += part[i];
    return SMPLX_OK;
}

int smplx_plan_multi(smplx_space** spaces, int nq, const smplx_search_params* p, int32_t* path_ids, int cap,
                     smplx_search_stats* stats, double* wall_seconds, int host_threads)
{
    if (!spaces || nq <= 0 || !p || !stats) return set_error(SMPLX_E_ARG, "bad argument");
    std::vector<Search> S(nq);
    std::vector<size_t> cw(nq);
    struct Base { int64_t b, h, m, c, g; };
    std::vector<Base> base(nq);
    for (int q = 0; q < nq; ++q) {
        smplx_space* s = spaces[q];
        if (!s) return set_error(SMPLX_E_ARG, "null space");
        if (!s->goal_set) return set_error(SMPLX_E_STATE, "goal not set");
        if (s->start_id < 0) return set_error(SMPLX_E_STATE, "start not set");
        if (s->grid->epoch != s->grid_epoch) return set_error(SMPLX_E_STATE, "the grid was edited after the goal was set: cached successors are stale, set the goal again");
        if (int e = pull_lattice(s)) return e;
        if (int e = pull_log(s)) return e;
        fill_search(S[q], s, p);
        s->adaptive_small = nq == 1;
        if (nq > 1) s->pipeline_left = 0;
        s->expansion_log.clear();
        const int capB = s->params.batch_states > 0 ? s->params.batch_states : 4096;
        cw[q] = counter_words(capB, s->M);
        if (int e = s->b_counters.reserve(cw[q])) return e;
        HIP_TRY(hipMemsetAsync(s->b_counters.p, 0, sizeof(unsigned long long) * cw[q], s->stream));
        base[q] = {s->gpu_batches, s->cache_hits, s->cache_misses, s->committed_evals, s->gpu_evals};
    }
    std::vector<char> done(nq, 0), waiting(nq, 0);
    std::vector<double> t_done(nq, 0.0);
    int remaining = nq;
    // Queries that share the scene (same grid handle), robot and primitives can share launches: their misses are
    // gathered into ONE cross-query frontier batch per sweep (per-state query index -> that query's goal and BFS
    // grid).  Otherwise each query issues its own batches on its own stream.
    bool grouped = nq > 1;
    for (int q = 1; q < nq && grouped; ++q) {
        const smplx_space* a = spaces[0];
        const smplx_space* b = spaces[q];
        grouped = a->grid == b->grid && a->blob_bytes == b->blob_bytes &&
                  std::memcmp(a->hs.model_blob, b->hs.model_blob, a->blob_bytes) == 0 &&
                  std::memcmp(&a->hs.actions, &b->hs.actions, sizeof(SmplxActionsDev)) == 0 && a->fused_mode == b->fused_mode;
    }
    const auto t0 = std::chrono::steady_clock::now();
    // ---- the device-resident search: one persistent workgroup per query, no host round trips.  Taken
    // whenever the kernel fits the robot (search_host.h); SMPLX_SEARCH=host selects the host-driven loop below, which is
    // also what serves an external SBPL planner through smplx_get_succs ----
    {
        bool device = true;
        for (int q = 0; q < nq && device; ++q) device = search_on_device(spaces[q]);
        const char* mode = getenv("SMPLX_SEARCH");
        if (mode && !std::strcmp(mode, "device")) {
            if (!device) return set_error(SMPLX_E_LIMIT, "SMPLX_SEARCH=device: the search kernel does not fit this robot / space");
        } else if (nq == 1) {
            // A lone query is a chain of dependent expansions: measured on MI355X (cfg 2) the host-driven loop with its
            // speculative frontier batches expands 1.1e5 states/s, the single workgroup 3.9e4; the device-resident search
            // wins where it has queries to run side by side (cfg 4: 4.5e6 against 1.5e6 states/s).
            device = false;
        }
        if (device) {
            if (grouped || nq == 1) {
                if (int e = search_run(spaces, nq, p, path_ids, cap, stats, t_done.data(), t0)) return e;
            } else {
                for (int q = 0; q < nq; ++q)
                    if (int e = search_run(spaces + q, 1, p, path_ids ? path_ids + (size_t)q * cap : nullptr, cap, stats + q, t_done.data() + q, t0)) return e;
            }
            if (wall_seconds) *wall_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            return SMPLX_OK;
        }
    }
    if (grouped) {
        // host threads: each drives a contiguous slice of the queries with its own leading space / stream, so the
        // commit work (hashing, heap, record ingestion) of different slices overlaps; the GPU serves all of them
        int nthreads = host_threads > 0 ? host_threads : 1;
        nthreads = std::max(1, std::min(nthreads, nq));
        std::vector<int> rc(nthreads, SMPLX_OK);
        std::vector<std::string> msg(nthreads);
        auto worker = [&](int t) {
            const int nt = (nthreads == 1 || nq < 4) ? 1 : nthreads;
            const int q0 = (int)((long long)nq * t / nt), q1 = (int)((long long)nq * (t + 1) / nt);
            rc[t] = run_group(spaces, S.data(), q0, q1, done.data(), t_done.data(), t0);
            if (rc[t] != SMPLX_OK) msg[t] = g_error;
        };
        if (nthreads == 1 || nq < 4) {
            worker(0);
            if (rc[0] != SMPLX_OK) return set_error(rc[0], msg[0]);
        } else {
            // host_threads worker threads + this thread as the only GPU submitter (run_pipelined)
            if (int e = run_pipelined(spaces, S.data(), nq, std::min(nthreads, nq), done.data(), t_done.data(), t0)) return e;
        }
        remaining = 0;
    } else {
        // One host thread drives every query: a query runs until it misses, its frontier batch goes to its own
        // stream, and the thread moves on to the next query; a landed batch is collected when its turn comes again.
        double t_resume = 0, t_wait = 0, t_collect = 0;
        const bool dbg = getenv("SMPLX_DEBUG_TIMING") != nullptr;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
        while (remaining > 0) {
            bool progressed = false;
            for (int q = 0; q < nq; ++q) {
                if (done[q]) continue;
                smplx_space* s = spaces[q];
                if (waiting[q]) {
                    const hipError_t st = hipEventQuery(s->batch_done);
                    if (st == hipErrorNotReady) continue;
                    if (st != hipSuccess) return set_error(SMPLX_E_HIP, std::string("hipEventQuery: ") + hipGetErrorString(st));
                    const auto c0 = now();
                    if (int e = collect_batch(s)) return e;
                    t_collect += secs(c0, now());
                    waiting[q] = 0;
                }
                const auto r0 = now();
                const int r = S[q].resume();
                t_resume += secs(r0, now());
                progressed = true;
                if (S[q].error) return S[q].error;
                if (r == Search::R_YIELD) { waiting[q] = 1; continue; }
                done[q] = 1;
                --remaining;
                t_done[q] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            }
            if (!progressed) {
                // every live query is waiting on the GPU: block on one of them instead of spinning
                const auto w0 = now();
                for (int q = 0; q < nq; ++q)
                    if (!done[q] && waiting[q]) { if (int e = wait_event_polling(spaces[q]->batch_done)) return e; break; }
                t_wait += secs(w0, now());
            }
        }
        if (dbg) fprintf(stderr, "[smplx timing] resume(search+issue) %.3fs wait %.3fs collect %.3fs; launches: single-kernel %lld pipeline %lld\n",
                         t_resume, t_wait, t_collect, (long long)spaces[0]->small_launches, (long long)spaces[0]->pipe_launches);
    }
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (wall_seconds) *wall_seconds = wall;
    for (int q = 0; q < nq; ++q) {
        smplx_space* s = spaces[q];
        unsigned long long counters[4] = {0, 0, 0, 0};
        if (!grouped) { if (int e = read_counters(s, cw[q], counters)) return e; }
        smplx_search_stats& st = stats[q];
        std::memset(&st, 0, sizeof(st));
        st.solved = S[q].solved;
        st.path_len = (int)S[q].solution.size();
        st.cost = S[q].cost;
        st.expansions = S[q].expand_count;
        st.expansions_init = S[q].expand_count_init;
        st.satisfied_eps = S[q].satisfied_eps;
        st.seconds = t_done[q];
        st.gpu_succ_evals = s->gpu_evals - base[q].g;
        st.grid_lookups = (int64_t)counters[2];
        st.committed_succ_evals = s->committed_evals - base[q].c;
        st.gpu_batches = s->gpu_batches - base[q].b;
        st.cache_misses = s->cache_misses - base[q].m;
        st.cache_hits = (s->cache_hits - base[q].h) - st.cache_misses;   // expansions served without waiting for the GPU
        if (path_ids)
            for (int i = 0; i < (int)S[q].solution.size() && i < cap; ++i) path_ids[(size_t)q * cap + i] = S[q].solution[i];
    }
    return SMPLX_OK;
}

int smplx_plan(smplx_space* s, const smplx_search_params* p, int32_t* path_ids, int cap, smplx_search_stats* stats)
{
    if (!s) return set_error(SMPLX_E_ARG, "null argument");
    return smplx_plan_multi(&s, 1, p, path_ids, cap, stats, nullptr, 1);
}

int smplx_expansion_log_size(const smplx_space* s)
{
    if (!s) return 0;
    return s->ds.log_on_device ? s->ds.h.n_log : (int)s->expansion_log.size();
}

int smplx_expansion_log(const smplx_space* cs, int32_t* out)
{
    if (!cs || !out) return set_error(SMPLX_E_ARG, "null argument");
    smplx_space* s = const_cast<smplx_space*>(cs);
    if (int e = pull_log(s)) return e;
    std::copy(s->expansion_log.begin(), s->expansion_log.end(), out);
    return SMPLX_OK;
}

int smplx_extract_path(smplx_space* s, const int32_t* ids, int len, double* q)
{
    if (!s || !ids || !q) return set_error(SMPLX_E_ARG, "null argument");
    if (int e = pull_lattice(s)) return e;
    // manip_lattice.cpp:2018-2155: every id maps to its stored state; a trailing goal id (0) maps to the
    // cheapest goal-satisfying successor of its predecessor
    for (int i = 0; i < len; ++i) {
        int id = ids[i];
        if (id == 0) {
            if (i == 0) return set_error(SMPLX_E_STATE, "path cannot start at the goal id");
            const int prev = ids[i - 1];
            if (prev > 0 && prev < (int)s->cache_off.size() && s->cache_off[prev] < 0 && s->done_off[prev] >= 0 &&
                s->done_prim.size() == s->done_succ.size()) {
                // expanded by the device-resident search: the committed list names the primitive of every successor; the
                // first goal successor in primitive order is the cheapest (every edge costs 1000 here, manip_lattice.cpp:
                // 1388-1412) and its joint values are recomputed with the device's arithmetic
                int prim = -1;
                for (int k = 0; k < s->done_cnt[prev] && prim < 0; ++k)
                    if (s->done_succ[s->done_off[prev] + k] == 0) prim = s->done_prim[s->done_off[prev] + k];
                if (prim < 0) return set_error(SMPLX_E_STATE, "no goal successor found during path extraction");
                const SmplxActionsDev& A = s->actions.dev;
                if (A.type[prim] == SMPLX_MP_LONG || A.type[prim] == SMPLX_MP_SHORT)
                    host_apply_prim(A, &s->qs[(size_t)prev * s->N], prim, s->N, q + (size_t)i * s->N);
                else
                    std::memcpy(q + (size_t)i * s->N, s->hs.goal.angles, sizeof(double) * s->N);   // snap to a joint goal (:551-559)
                continue;
            }
            if (prev <= 0 || prev >= (int)s->cache_off.size() || s->cache_off[prev] < 0)
                return set_error(SMPLX_E_STATE, "goal predecessor was never expanded");
            int best = -1, best_cost = std::numeric_limits<int>::max();
            for (int k = 0; k < s->cache_cnt[prev]; ++k) {
                const smplx_space::Rec& r = s->recs[s->cache_off[prev] + k];
                if (!r.goal) continue;
                const int edge_cost = 1000;   // 3-argument cost() (manip_lattice.cpp:1388-1412)
                if (edge_cost < best_cost) { best_cost = edge_cost; best = k; }
            }
            if (best < 0) return set_error(SMPLX_E_STATE, "no goal successor found during path extraction");
            std::memcpy(q + (size_t)i * s->N, &s->rec_q[(size_t)(s->cache_off[prev] + best) * s->N], sizeof(double) * s->N);
            continue;
        }
        if (id < 0 || id >= (int)s->h_of_id.size()) return set_error(SMPLX_E_STATE, "unknown state id in path");
        std::memcpy(q + (size_t)i * s->N, &s->qs[(size_t)id * s->N], sizeof(double) * s->N);
    }
    return SMPLX_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// path post-processing.  The greedy loops are the reference's, run on the host; what they ask of
// the collision checker -- isStateToStateValid for the shortcut generator, isStateValid for every interpolated
// point -- is answered from waypoint-parallel GPU batches (k_state_valid over all waypoints of all candidate
// edges of the current segment start), so the answers are those of the sequential checker (an edge is valid iff
// all its waypoints are, collision_space.cpp:538-581).
// ---------------------------------------------------------------------------------------------------------------
}  // extern "C"

namespace {

struct PathTools {
    smplx_space* s;
    int N;
    const SmplxModelDev& M;
    long edge_batches = 0, configs = 0;
    explicit PathTools(smplx_space* sp) : s(sp), N(sp->N), M(sp->model.dev) {}

    int waypoint_count(const double* a, const double* b) const
    {
        int W = 0;
        (void)smplx_cc_interpolate(s, a, b, nullptr, 0, &W);
        return W;
    }
    void append_waypoints(const double* a, const double* b, int W, std::vector<double>& out) const
    {
        const size_t o = out.size();
        out.resize(o + (size_t)W * N);
        int n = 0;
        (void)smplx_cc_interpolate(s, a, b, out.data() + o, W, &n);
    }
    // post_processing.cpp:52-67
    double distance(const double* from, const double* to) const
    {
        double dist = 0.0;
        for (int v = 0; v < N; ++v) {
            if (M.var_type[v] == SMPLX_JT_CONTINUOUS) dist += std::fabs(smplx_shortest_angle_diff(to[v], from[v]));
            else dist += std::fabs(to[v] - from[v]);
        }
        return dist;
    }
    // collision_space.cpp:776-793
    bool within_limits(const double* q) const
    {
        for (int v = 0; v < N; ++v) {
            if (M.var_type[v] == SMPLX_JT_CONTINUOUS) continue;
            if (!(q[v] >= M.var_min[v] && q[v] <= M.var_max[v])) return false;
        }
        return true;
    }
    int states_valid(const std::vector<double>& q, std::vector<uint8_t>& valid)
    {
        const int n = (int)(q.size() / N);
        valid.assign(n, 0);
        configs += n;
        return smplx_cc_state_valid_batch(s, q.data(), n, valid.data(), nullptr);
    }
};

// validity of path edges (a, b).  An unknown edge is answered optimistically ("valid") and queued; the caller
// re-runs its loop after resolve() until a run asks nothing new, so the accepted run saw only checked answers.
struct EdgeOracle {
    PathTools& T;
    const double* path;
    std::unordered_map<uint64_t, int8_t> known;
    std::vector<std::pair<int, int>> pending;

    bool query(int a, int b)
    {
        const uint64_t key = ((uint64_t)(uint32_t)a << 32) | (uint32_t)b;
        auto it = known.find(key);
        if (it != known.end()) {
            if (it->second >= 0) return it->second == 1;
            return true;                       // already queued in this run
        }
        known.emplace(key, (int8_t)-1);
        pending.emplace_back(a, b);
        return true;
    }
    // one waypoint-parallel batch for every queued edge
    int resolve()
    {
        std::vector<double> q;
        std::vector<int> first(1, 0);
        for (const auto& e : pending) {
            const double* a = path + (size_t)e.first * T.N;
            const double* b = path + (size_t)e.second * T.N;
            const int W = T.waypoint_count(a, b);
            T.append_waypoints(a, b, W, q);
            first.push_back(first.back() + W);
        }
        std::vector<uint8_t> valid;
        if (!q.empty()) { if (int e = T.states_valid(q, valid)) return e; }
        ++T.edge_batches;
        for (size_t k = 0; k < pending.size(); ++k) {
            bool all = true;
            for (int w = first[k]; w < first[k + 1]; ++w) all = all && valid[w] != 0;
            known[((uint64_t)(uint32_t)pending[k].first << 32) | (uint32_t)pending[k].second] = all ? 1 : 0;
        }
        pending.clear();
        return SMPLX_OK;
    }
};

// shortcut.hpp:110-286 with the joint-space generator (post_processing.cpp:100-127), granularity 1
void shortcut_run(PathTools& T, EdgeOracle& E, const std::vector<double>& pin, const std::vector<double>& accum,
                  std::vector<double>& pout)
{
    const int N = T.N;
    const int P = (int)(pin.size() / N);
    pout.clear();
    auto pt = [&](int i) { return pin.data() + (size_t)i * N; };
    auto push = [&](int i) { pout.insert(pout.end(), pt(i), pt(i) + N); };
    auto generate = [&](int a, int b, double& cost) {
        if (!E.query(a, b)) return false;
        cost = T.distance(pt(a), pt(b));
        return true;
    };
    int start = 0, end = 1;
    bool best_direct = false;
    int best_last = end;
    double best_cost = accum[end] - accum[start], cost = 0.0;
    if (generate(start, end, cost) && cost <= best_cost) { best_direct = true; best_cost = cost; }
    push(0);
    auto emit_best = [&]() {
        if (best_direct) push(best_last);
        else for (int i = start + 1; i <= best_last; ++i) push(i);
    };
    while (end != P) {
        bool improved = false;
        const int look = std::min(1, P - end - 1);
        if (look != 0) {
            double new_cost = best_cost + (accum[end + look] - accum[end]);
            if (generate(start, end + look, cost) && cost <= new_cost) {
                improved = true;
                best_direct = true;
                best_last = end + look;
                new_cost = cost;
            }
            best_cost = new_cost;
        }
        if (improved) {
            end += look;
        } else if (look == 0) {
            end = P;
        } else {
            emit_best();
            start = end;
            end += look;
            best_direct = false;
            best_last = end;
            best_cost = accum[end] - accum[start];
            if (generate(start, end, cost) && cost <= best_cost) { best_direct = true; best_cost = cost; }
        }
    }
    emit_best();
}

int shortcut_path(PathTools& T, const std::vector<double>& pin, std::vector<double>& pout)
{
    const int N = T.N;
    const int P = (int)(pin.size() / N);
    if (P < 2) { pout = pin; return SMPLX_OK; }
    std::vector<double> accum(P);
    accum[0] = 0.0;
    for (int i = 1; i < P; ++i) accum[i] = accum[i - 1] + T.distance(pin.data() + (size_t)(i - 1) * N, pin.data() + (size_t)i * N);
    EdgeOracle E{T, pin.data()};
    for (;;) {
        shortcut_run(T, E, pin, accum, pout);
        if (E.pending.empty()) return SMPLX_OK;     // every answer this run used was a checked one
        if (int e = E.resolve()) return e;
    }
}

// post_processing.cpp:464-523 over CollisionSpace::interpolatePath (collision_space.cpp:583-612); *done = the
// reference's return value (false leaves the path as it was)
int interpolate_path(PathTools& T, std::vector<double>& path, bool fork_limits_test, bool* done)
{
    const int N = T.N;
    const int P = (int)(path.size() / N);
    *done = true;
    if (P == 0) return SMPLX_OK;
    std::vector<double> q;
    std::vector<int> first(1, 0);
    for (int i = 0; i + 1 < P; ++i) {
        const double* a = path.data() + (size_t)i * N;
        const double* b = a + N;
        const bool wa = T.within_limits(a), wb = T.within_limits(b);
        // [FORK] :592-597 reports "Joint limits violated" when either end IS within its limits
        if (fork_limits_test ? (wa || wb) : (!wa || !wb)) { *done = false; return SMPLX_OK; }
        const int W = T.waypoint_count(a, b);
        T.append_waypoints(a, b, W, q);
        first.push_back(first.back() + W);
    }
    std::vector<uint8_t> valid;
    if (!q.empty()) { if (int e = T.states_valid(q, valid)) return e; }
    std::vector<double> out(path.begin(), path.begin() + N);
    for (int i = 0; i + 1 < P; ++i) {
        bool collision = false;
        for (int w = first[i]; w < first[i + 1]; ++w) collision = collision || valid[w] == 0;
        if (collision) {
            out.insert(out.end(), path.begin() + (size_t)(i + 1) * N, path.begin() + (size_t)(i + 2) * N);
        } else if (first[i + 1] > first[i]) {
            out.insert(out.end(), q.begin() + (size_t)(first[i] + 1) * N, q.begin() + (size_t)first[i + 1] * N);
        }
    }
    path.swap(out);
    return SMPLX_OK;
}

}  // namespace

extern "C" {

int smplx_post_process_path(smplx_space* s, const double* path, int n, int flags, double* out, int cap, int* nout,
                            int64_t* stats)
{
    if (!s || (!path && n > 0) || n < 0 || !nout) return set_error(SMPLX_E_ARG, "bad argument");
    if (n > 0 && !sane_values(path, (size_t)n * s->N)) return set_error(SMPLX_E_ARG, "joint values must be finite (|q| < 1e6)");
    PathTools T(s);
    std::vector<double> p(path, path + (size_t)n * s->N);
    const bool fork_test = !(flags & SMPLX_PP_UPSTREAM_LIMITS);
    bool done = false;
    // PlannerInterface::postProcessPath (planner_interface.cpp:2651-2697)
    if (flags & SMPLX_PP_SHORTCUT) {
        if (int e = interpolate_path(T, p, fork_test, &done)) return e;   // failure leaves the path as it was
        std::vector<double> in = p;
        if (int e = shortcut_path(T, in, p)) return e;
    }
    if (flags & SMPLX_PP_INTERPOLATE) {
        if (int e = interpolate_path(T, p, fork_test, &done)) return e;
    }
    const int np = (int)(p.size() / s->N);
    *nout = np;
    if (stats) { stats[0] = T.edge_batches; stats[1] = T.configs; }
    if (out) {
        if (np > cap) return set_error(SMPLX_E_LIMIT, "output path does not fit");
        std::memcpy(out, p.data(), sizeof(double) * p.size());
    }
    return SMPLX_OK;
}

}  // extern "C"
