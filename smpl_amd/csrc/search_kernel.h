// smpl_amd/csrc/search_kernel.h -- device-resident ARA* (SURVEY row N2), included at the end of kernels.hip.
//
//   k_search      one persistent workgroup per query: pop -> expand on the lanes -> getOrCreateState -> push / decrease.
//                 Follows smpl/src/search/arastar.cpp:107-215 (replan), 486-527 (improvePath), 531-568 (expand),
//                 571-582 (reorderOpen, computeKey), 613-627 (reinitSearchState); OPEN is the reference's binary heap
//                 (smpl/include/smpl/detail/intrusive_heap.hpp:145-166 push / pop, 346-395 the sift rules: strict '<',
//                 the left child only if left < right), so ties pop in the reference's order and state ids -- assigned
//                 here, by the workgroup, in its own commit order (manip_lattice.cpp:1302-1354) -- are the reference's.
//
// Who does what inside the workgroup (blockDim.x = smplx_small_block(nprims)):
//   thread 0      the sequential part: the replan state machine, pop, the relaxation of the successors in primitive order
//                 (the order decides ties in OPEN), INCONS
//   all threads   the GetSuccs loop body of the popped state (expand_state_block: one configuration per lane), the state
//                 table lookups / inserts of its successors (one lane per primitive), the prefetch of the successors'
//                 search states, the recomputation of f and the level-parallel make() of an epsilon step
// The first `LH` entries of the heap array live in LDS while the kernel runs (a sift is a chain of dependent reads: tens of
// nanoseconds each in LDS, most of a microsecond in HBM); they are written back when the launch ends.
//
// A launch runs at most `max_steps` expansions and then stores its state in the query's SmplxSearchDev, so that the host
// sees progress, can stop a search, and can enlarge buffers (SMPLX_SS_GROW) between launches.

struct SearchLds {
    // working copy of the header fields that change
    double curr_eps, satisfied_eps;
    int heap_size, nstates, n_incons, n_log, n_succ;
    int iteration, call_number, phase, num, expand_count, expand_count_init, err;
    int dup_pushes, status, grow_what;
    unsigned int goal_f;
    long long committed_evals, gpu_evals, lookups;
    long long ticks[8];
    // the step in hand
    int action, m, cnt, evals, from_cache;
    unsigned int eg;
    int succ_id[SMPLX_MAX_PRIMS], succ_cost[SMPLX_MAX_PRIMS], succ_prim[SMPLX_MAX_PRIMS];
    int alias[SMPLX_MAX_PRIMS];
    int lane_id[SMPLX_MAX_PRIMS];          // commit: state id of primitive p's successor (-1 = not in the table yet)
    unsigned int lane_hash[SMPLX_MAX_PRIMS];
    SmplxSState sst[SMPLX_MAX_PRIMS];      // the successors' search states, fetched together
};

enum { SA_EXPAND = 0, SA_REORDER = 1, SA_EXIT = 2 };
#define SMPLX_INFINITECOST 1000000000u     // SBPL INFINITECOST


// a heap entry as one 64-bit word (the layout of SmplxHeapEntry: f in the low half, id in the high half): LDS and HBM
// copies are single loads / stores
typedef unsigned long long hent_t;
__device__ __forceinline__ unsigned int hent_f(hent_t e) { return (unsigned int)e; }
__device__ __forceinline__ int hent_id(hent_t e) { return (int)(e >> 32); }
__device__ __forceinline__ hent_t hent_make(unsigned int f, int id) { return (hent_t)f | ((hent_t)(unsigned int)id << 32); }

struct HeapRef {
    LDS_AS hent_t* lds;                    // entries [0, lh)
    hent_t* hbm;                           // entries [lh, ...)
    SmplxSState* st;
    int lh;
};

__device__ __forceinline__ hent_t hget(const HeapRef& H, int i)
{
    if (i < H.lh) return H.lds[i];
    return H.hbm[i];
}
// the entry alone (the state's heap_index is not touched)
__device__ __forceinline__ void hput(const HeapRef& H, int i, hent_t e)
{
    if (i < H.lh) H.lds[i] = e;
    else H.hbm[i] = e;
}
// place an entry and record its position in the state (intrusive_heap.hpp: m_data[i] = e; e->m_heap_index = i)
__device__ __forceinline__ void hset(const HeapRef& H, int i, hent_t e)
{
    hput(H, i, e);
    H.st[hent_id(e)].heap_index = i;
}

// intrusive_heap.hpp:346-377; size = number of elements (the reference's m_data.size() - 1)
__device__ __forceinline__ void heap_percolate_down(const HeapRef& H, int pivot, int size)
{
    if (pivot > size) return;
    int left = pivot << 1, right = left + 1;
    const hent_t tmp = hget(H, pivot);
    while (left <= size) {
        hent_t es = hget(H, left);
        int s = left;
        if (right <= size) {
            const hent_t er = hget(H, right);
            if (!(hent_f(es) < hent_f(er))) { es = er; s = right; }
        }
        if (hent_f(es) < hent_f(tmp)) {
            hset(H, pivot, es);
            pivot = s;
        } else break;
        left = pivot << 1; right = left + 1;
    }
    hset(H, pivot, tmp);
}

// intrusive_heap.hpp:379-395
__device__ __forceinline__ void heap_percolate_up(const HeapRef& H, int pivot)
{
    const hent_t tmp = hget(H, pivot);
    while (pivot != 1) {
        const int p = pivot >> 1;
        const hent_t ep = hget(H, p);
        if (hent_f(ep) < hent_f(tmp)) break;
        hset(H, pivot, ep);
        pivot = p;
    }
    hset(H, pivot, tmp);
}

// intrusive_heap.hpp:208-213 make(): percolate_down(i) for i = size/2 .. 1, by ALL threads of the block.  Nodes of one depth
// own disjoint subtrees, so a depth is sifted by all threads at once, deepest first -- the result is the sequential
// loop's.  With the same element twice in the array (`duplicates`) two sifts could race on its heap_index: then thread 0
// runs the loop alone.  Ends with a barrier.
__device__ __forceinline__ void heap_make_block(const HeapRef& H, int size, bool duplicates)
{
    const int t = threadIdx.x;
    if (duplicates) {
        if (t == 0) for (int i = size >> 1; i >= 1; --i) heap_percolate_down(H, i, size);
        __syncthreads();
        return;
    }
    int top_depth = 0;
    while ((2 << top_depth) <= (size >> 1)) ++top_depth;      // depth of node size/2
    for (int d = top_depth; d >= 0; --d) {
        const int first = 1 << d;
        int lastn = (2 << d) - 1;
        if (lastn > (size >> 1)) lastn = size >> 1;
        for (int i = first + t; i <= lastn; i += blockDim.x) heap_percolate_down(H, i, size);
        __syncthreads();
    }
}

// ARAStar::computeKey (arastar.cpp:579-582)
__device__ __forceinline__ unsigned int search_key(double eps, unsigned int g, unsigned int h)
{
    return g + (unsigned int)(long long)(eps * (double)h);
}

// every OPEN entry of state `id` takes the state's new f.  Only needed once a state has been pushed while already in OPEN
// (the reference appends a state to INCONS once per improvement, arastar.cpp:563-565, and pushes every INCONS entry,
// :180-184): its heap then holds the same element twice, and both see an f change because both point to it.
__device__ __forceinline__ void heap_refresh_duplicates(const HeapRef& H, int size, int id, unsigned int f)
{
    for (int i = 1; i <= size; ++i) {
        const hent_t e = hget(H, i);
        if (hent_id(e) == id && hent_f(e) != f) hput(H, i, hent_make(f, id));
    }
}

// the L1-bypassing form of table_lookup for the workgroup that also inserts (its own inserts were made with atomics and
// plain stores in an earlier phase, separated from this one by a barrier): 64-bit relaxed atomic loads, all in flight at once
__device__ __forceinline__ int table_lookup_own(const SmplxTableDev& T, const LDS_AS int* c, int nv, unsigned int hash)
{
    unsigned int i = hash & T.mask;
    while (true) {
        const int* sl = T.slots + (size_t)i * T.stride;
        const int tag = __atomic_load_n(&sl[0], __ATOMIC_RELAXED);
        if (tag <= 0) return -1;
        bool same = true;
        for (int v = 0; v < nv; ++v) same = same && __atomic_load_n(&sl[1 + v], __ATOMIC_RELAXED) == c[v];
        if (same) return tag - 1;
        i = (i + 1) & T.mask;
    }
}

__device__ __forceinline__ void table_insert_own(const SmplxTableDev& T, const LDS_AS int* c, int nv, unsigned int hash, int id)
{
    unsigned int k = hash & T.mask;
    while (true) {
        int* sl = T.slots + (size_t)k * T.stride;
        if (atomicCAS(&sl[0], 0, -(id + 1)) == 0) {
            for (int v = 0; v < nv; ++v) __atomic_store_n(&sl[1 + v], c[v], __ATOMIC_RELAXED);
            __threadfence();
            __atomic_store_n(&sl[0], id + 1, __ATOMIC_RELAXED);
            return;
        }
        k = (k + 1) & T.mask;
    }
}

__device__ __forceinline__ unsigned int coord_hash_lds(const LDS_AS int* c, int n)
{
    unsigned int h = 2166136261u;
    for (int i = 0; i < n; ++i) h = (h ^ (unsigned int)c[i]) * 16777619u;
    h ^= h >> 15; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

// ARAStar::reinitSearchState (arastar.cpp:613-627) on a copy of the state; h stays (GetGoalHeuristic of a state is fixed
// for a goal).  A state that was not touched in this call is not in OPEN: OPEN is emptied when a call starts.
__device__ __forceinline__ void sstate_reinit(SmplxSState& s, int call_number)
{
    s.g = SMPLX_INFINITECOST;
    s.f = SMPLX_INFINITECOST;
    s.eg = SMPLX_INFINITECOST;
    s.iteration_closed = 0;
    s.call_number = (unsigned short)call_number;
    s.bp = -1;
    s.heap_index = 0;
    s.flags = 0;
}

typedef int __attribute__((ext_vector_type(4))) sk_int4;

// write a state's fields back; heap_index only when the caller owns it (it is otherwise kept current by hset)
__device__ __forceinline__ void sstate_store(SmplxSState* dst, const SmplxSState& s, bool with_heap_index)
{
    sk_int4 a;
    a.x = (int)s.g; a.y = (int)s.h; a.z = (int)s.f; a.w = (int)s.eg;
    *reinterpret_cast<sk_int4*>(dst) = a;
    dst->bp = s.bp;
    if (with_heap_index) dst->heap_index = s.heap_index;
    *reinterpret_cast<unsigned int*>(&dst->iteration_closed) = (unsigned int)s.iteration_closed | ((unsigned int)s.call_number << 16);
    dst->flags = s.flags;
}

__device__ __forceinline__ bool search_timed_out(const SmplxSearchDev& P, const SearchLds& W)
{
    if (!P.bounded) return false;
    if (W.satisfied_eps == __builtin_inf()) return W.num >= P.max_init;
    return W.num >= P.max_rep;
}

extern "C" __global__ void __launch_bounds__(512)
k_search(const SmplxSpaceDev* const* __restrict__ stab, int max_steps, int lh, int* __restrict__ status_out)
{
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ ExpandLds X;
    __shared__ SearchLds W;
    __shared__ SmplxSearchDev Ph;                // the query's header as the launch found it: pointers, capacities, parameters
    static_assert(sizeof(ExpandLds) + sizeof(SearchLds) + sizeof(SmplxSearchDev) <= SMPLX_SEARCH_STATIC_LDS, "engine.hip budgets this much static LDS");
    const SmplxSpaceDev* Sq = stab[blockIdx.x];
    const SmplxSpaceDev* S = stab[0];            // scene, robot and primitives are shared by the queries of a launch
    SmplxSearchDev* const Pd = Sq->search;
    const int t = threadIdx.x;
    {
        const int st0 = Pd->status;
        if (st0 == SMPLX_SS_DONE || st0 == SMPLX_SS_ERROR) {   // uniform: this query needs nothing more
            if (t == 0 && status_out) status_out[blockIdx.x] = st0;
            return;
        }
    }
    for (int i = t; i < (int)(sizeof(SmplxSearchDev) / 4); i += blockDim.x) ((int*)&Ph)[i] = ((const int*)Pd)[i];
    __syncthreads();
    const SmplxSearchDev* const P = &Ph;         // read-only view; what changes goes through W and back to Pd at the end
    ModelLds Mv;
    ThreadLds L = setup_lds(S, smem, &Mv, blockDim.x);
    const ModelLds* M = &Mv;
    const SmplxActionsDev& A = S->actions;
    const SmplxGridDev grid = S->grid;
    const SmplxTableDev table = Sq->table;
    const int nprims = A.nprims, nv = MV_NVARS(M);
    HeapRef H;
    {
        // the heap cache sits behind the model and the per-thread scratch of the expansion (setup_lds)
#ifdef SMPLX_CONST_MODEL
        const int nroot_lds = 0;
#else
        const int nroot_lds = Mv.nroot;
#endif
        const int* hdr = reinterpret_cast<const int*>(S->model_blob);
        unsigned int off = (unsigned int)hdr[SMPLX_BH_BYTES] +
                           (unsigned int)((3 * nroot_lds + 12 * Mv.nslots + Mv.nvars) * 8 + SMPLX_STACK_BYTES) * blockDim.x;
        off = (off + 15u) & ~15u;
        H.lds = (LDS_AS hent_t*)((LDS_AS unsigned char*)smem + off);
        H.hbm = (hent_t*)P->heap;
        H.st = P->st;
        H.lh = lh;
    }
    // ---- load the working state ----
    if (t == 0) {
        W.curr_eps = P->curr_eps; W.satisfied_eps = P->satisfied_eps;
        W.heap_size = P->heap_size; W.nstates = P->nstates; W.n_incons = P->n_incons; W.n_log = P->n_log; W.n_succ = P->n_succ;
        W.iteration = P->iteration; W.call_number = P->call_number; W.phase = P->phase; W.num = P->num;
        W.expand_count = P->expand_count; W.expand_count_init = P->expand_count_init; W.err = P->err;
        W.dup_pushes = P->dup_pushes; W.goal_f = P->goal_f;
        W.committed_evals = P->committed_evals; W.gpu_evals = P->gpu_evals; W.lookups = P->lookups;
        for (int k = 0; k < 8; ++k) W.ticks[k] = P->ticks[k];
        W.action = SA_EXPAND; W.status = SMPLX_SS_RUNNING; W.grow_what = 0;
    }
    __syncthreads();
    {
        const int n = W.heap_size + 1 < lh ? W.heap_size + 1 : lh;
        for (int i = t; i < n; i += blockDim.x) H.lds[i] = H.hbm[i];
    }
    __syncthreads();
    if (t == 0 && W.phase == 0) {
        // ---- ARAStar::replan, from scratch (arastar.cpp:107-167): empty OPEN and INCONS, new call number, the start state
        // with g = 0 into OPEN ----
        W.heap_size = 0; W.n_incons = 0; W.n_log = 0;
        W.call_number = (W.call_number + 1) & 0xFFFF;
        if (W.call_number == 0) W.call_number = 1;
        SmplxSState ss = P->st[P->start_id], gs = P->st[0];
        sstate_reinit(ss, W.call_number);
        sstate_reinit(gs, W.call_number);
        W.iteration = 1;
        W.curr_eps = P->initial_eps;
        W.satisfied_eps = __builtin_inf();
        ss.g = 0;
        ss.f = search_key(W.curr_eps, ss.g, ss.h);
        sstate_store(&P->st[P->start_id], ss, true);
        if (P->start_id != 0) sstate_store(&P->st[0], gs, true);
        W.goal_f = P->start_id == 0 ? ss.f : gs.f;
        W.heap_size = 1;
        hset(H, 1, hent_make(ss.f, P->start_id));
        W.num = 0; W.err = 0; W.expand_count = 0; W.expand_count_init = 0; W.dup_pushes = 0;
        W.phase = 1;
    }
    __syncthreads();

    long long tick = 0;
    if (t == 0) tick = (long long)wall_clock64();
#define SK_TICK(k) do { const long long now_ = (long long)wall_clock64(); W.ticks[k] += now_ - tick; tick = now_; } while (0)

    for (int step = 0; step < max_steps;) {
        // =========================== thread 0: what happens next (Search::resume / improve_path) ===========================
        if (t == 0) {
            int action = -1;
            while (action < 0) {
                if (W.phase == 1) {
                    // arastar.cpp:169-186
                    if (!(W.satisfied_eps > P->final_eps)) { W.phase = 3; action = SA_EXIT; break; }
                    if (W.curr_eps == W.satisfied_eps) {
                        if (!P->improve) { W.phase = 3; action = SA_EXIT; break; }
                        if (W.heap_size + W.n_incons > P->cap_heap) { W.status = SMPLX_SS_GROW; W.grow_what = 1; action = SA_EXIT; break; }
                        action = SA_REORDER;
                        break;
                    }
                    W.phase = 2;
                }
                // ---- one step of improvePath (arastar.cpp:486-527) ----
                int err = -1;
                hent_t top = 0;
                if (W.heap_size == 0) err = 5;                                          // EXHAUSTED_OPEN_LIST
                else {
                    top = hget(H, 1);
                    if (hent_f(top) >= W.goal_f || hent_id(top) == 0) err = 0;          // SUCCESS
                    else if (search_timed_out(*P, W)) err = 4;                          // TIMED_OUT
                }
                if (err >= 0) {
                    // back in replan (arastar.cpp:188-197)
                    if (W.curr_eps == P->initial_eps) W.expand_count_init += W.num;
                    W.phase = 1;
                    W.err = err;
                    if (err != 0) { W.phase = 3; action = SA_EXIT; break; }
                    W.satisfied_eps = W.curr_eps;
                    continue;
                }
                // room for one more expansion?  (checked before anything is popped: the host enlarges and launches again)
                if (W.nstates + nprims > P->cap_states || W.heap_size + nprims > P->cap_heap || W.n_incons + nprims > P->cap_incons ||
                    W.n_log + 1 > P->cap_log || W.n_succ + nprims > P->cap_succ || (unsigned int)(2 * (W.nstates + nprims)) > table.mask + 1u) {
                    W.status = SMPLX_SS_GROW; W.grow_what = 2;
                    action = SA_EXIT;
                    break;
                }
                // ---- pop (intrusive_heap.hpp:155-166) ----
                const int m = hent_id(top);
                SmplxSState* sm = &P->st[m];
                const unsigned int g = sm->g;            // in flight while the heap is sifted
                sm->heap_index = 0;
                const hent_t last = hget(H, W.heap_size);
                --W.heap_size;
                if (W.heap_size >= 1) {
                    hput(H, 1, last);
                    heap_percolate_down(H, 1, W.heap_size);
                }
                sm->iteration_closed = (unsigned short)W.iteration;
                sm->eg = g;
                P->log[W.n_log++] = m;
                W.m = m;
                W.eg = g;
                action = SA_EXPAND;
            }
            W.action = action;
            SK_TICK(1);
        }
        __syncthreads();
        const int action = W.action;
        if (action == SA_EXIT) break;

        if (action == SA_REORDER) {
            // =========================== a new epsilon (arastar.cpp:174-186, 571-577) ===========================
            if (t == 0) {
                ++W.iteration;
                W.curr_eps -= P->delta_eps;
                W.curr_eps = W.curr_eps > P->final_eps ? W.curr_eps : P->final_eps;
                for (int i = 0; i < W.n_incons; ++i) {
                    const int sid = P->incons[i];
                    SmplxSState* ss = &P->st[sid];
                    if (ss->heap_index != 0) { ss->flags |= 1u; ++W.dup_pushes; }   // already in OPEN: the same element twice
                    ++W.heap_size;
                    hset(H, W.heap_size, hent_make(ss->f, sid));
                    heap_percolate_up(H, W.heap_size);
                }
                W.n_incons = 0;
            }
            __syncthreads();
            // f of every OPEN entry under the new epsilon
            const int size = W.heap_size;
            const double eps = W.curr_eps;
            for (int i = 1 + t; i <= size; i += blockDim.x) {
                const int eid = hent_id(hget(H, i));
                SmplxSState* ss = &P->st[eid];
                const unsigned int f = search_key(eps, ss->g, ss->h);
                ss->f = f;
                hput(H, i, hent_make(f, eid));
            }
            __syncthreads();
            heap_make_block(H, size, W.dup_pushes > 0);
            if (t == 0) {
                const SmplxSState* gs = &P->st[0];
                W.goal_f = gs->f;     // (the goal state is re-initialised when a call starts: its f is this call's)
                W.phase = 2;
                SK_TICK(5);
            }
            __syncthreads();
            continue;
        }

        // =========================== expand the popped state (arastar.cpp:531-568) ===========================
        ++step;
        const int m = W.m;
        const int off = P->done_off[m];
        if (off >= 0) {
            // GetSuccs of a state expanded before (a later ARA* iteration): the committed list
            const int dc = P->done_cnt[m];
            const int cnt = dc & 0xFF;
            if (t < cnt) {
                const SmplxSucc sc = P->succ[off + t];
                W.succ_id[t] = sc.id;
                W.succ_cost[t] = sc.cost_prim & 0xFFFFFF;
                W.succ_prim[t] = (int)((unsigned int)sc.cost_prim >> 24);
            }
            if (t == 0) { W.cnt = cnt; W.evals = dc >> 8; W.from_cache = 1; }
            __syncthreads();
        } else {
            // ---- GetSuccs loop body (manip_lattice.cpp:254-305) on the lanes ----
            if (t < nv) X.parent[t] = P->q[(size_t)m * nv + t];
            expand_state_block(M, L, S, Sq, grid, X);
            if (t == 0) SK_TICK(2);
            // ---- getOrCreateState for every valid successor (manip_lattice.cpp:1302-1354): lane p = primitive p ----
            const int p = t;
            const bool lane = p < nprims;
            const int flags = lane ? X.flags[p] : SMPLX_F_INACTIVE;
            const bool valid = (flags & SMPLX_F_VALID) != 0;
            int id = -1;
            unsigned int hash = 0;
            if (valid) {
                hash = coord_hash_lds((const LDS_AS int*)X.coord[p], nv);
                id = table_lookup_own(table, (const LDS_AS int*)X.coord[p], nv, hash);
            }
            if (lane) { W.lane_id[p] = id; W.lane_hash[p] = hash; }
            __syncthreads();
            // two successors of this expansion with the same new coordinate: the lower primitive creates the state
            int dup_of = -1;
            if (valid && id < 0) {
                for (int k = 0; k < p && dup_of < 0; ++k) {
                    if (W.lane_id[k] >= 0 || !(X.flags[k] & SMPLX_F_VALID) || W.lane_hash[k] != hash) continue;
                    bool same = true;
                    for (int v = 0; v < nv; ++v) same = same && X.coord[k][v] == X.coord[p][v];
                    if (same) dup_of = k;
                }
            }
            const bool is_new = valid && id < 0 && dup_of < 0;
            // lanes p < nprims <= 64 are all in wave 0: ballots give the ranks
            const unsigned long long m_new = __ballot(is_new), m_valid = __ballot(valid), m_eval = __ballot(lane && !(flags & SMPLX_F_INACTIVE));
            const unsigned long long below = (t & 63) == 0 ? 0ull : (~0ull >> (64 - (t & 63)));
            if (is_new) {
                id = W.nstates + __popcll(m_new & below);
                table_insert_own(table, (const LDS_AS int*)X.coord[p], nv, hash, id);
                for (int v = 0; v < nv; ++v) { P->coord[(size_t)id * nv + v] = X.coord[p][v]; P->q[(size_t)id * nv + v] = X.sq[p][v]; }
                SmplxSState ns;
                ns.h = (unsigned int)X.h[p];
                sstate_reinit(ns, W.call_number);
                sstate_store(&P->st[id], ns, true);
                P->done_off[id] = -1;
            }
            __syncthreads();   // W.lane_id was read by the duplicate scan above
            if (lane) W.lane_id[p] = id;
            __syncthreads();
            if (dup_of >= 0) id = W.lane_id[dup_of];
            if (valid) {
                const int k = __popcll(m_valid & below);
                const int sid = (flags & SMPLX_F_GOAL) ? 0 : id;      // a goal successor is reported as the goal id (manip_lattice.cpp:283-296)
                W.succ_id[k] = sid;
                W.succ_cost[k] = A.cost[p];
                W.succ_prim[k] = p;
                SmplxSucc sc;
                sc.id = sid;
                sc.cost_prim = A.cost[p] | (p << 24);
                P->succ[W.n_succ + k] = sc;
            }
            if (t == 0) {
                const int cnt = __popcll(m_valid), evals = __popcll(m_eval);
                P->done_off[m] = W.n_succ;
                P->done_cnt[m] = cnt | (evals << 8);
                W.n_succ += cnt;
                W.nstates += __popcll(m_new);
                W.cnt = cnt; W.evals = evals; W.from_cache = 0;
                W.gpu_evals += evals;
                long long lk = 0;
                for (int k = 0; k < nprims; ++k) lk += X.lookups[k];
                W.lookups += lk;
                SK_TICK(3);
            }
            __syncthreads();
        }
        // ---- the successors' search states, all misses at once; and which of them name the same state ----
        const int cnt = W.cnt;
        if (t < cnt) {
            const int sid = W.succ_id[t];
            W.sst[t] = P->st[sid];
            int al = -1;
            for (int k = 0; k < t; ++k) if (W.succ_id[k] == sid) { al = k; break; }
            W.alias[t] = al;
        }
        __syncthreads();
        if (t == 0) {
            W.committed_evals += W.evals;
            const unsigned int eg = W.eg;
            for (int k = 0; k < cnt; ++k) {
                const int sid = W.succ_id[k];
                SmplxSState& ss = W.sst[W.alias[k] >= 0 ? W.alias[k] : k];
                bool dirty = false;
                if (ss.call_number != (unsigned short)W.call_number) { sstate_reinit(ss, W.call_number); P->st[sid].heap_index = 0; dirty = true; }
                const int new_cost = (int)(eg + (unsigned int)W.succ_cost[k]);
                if ((unsigned int)new_cost < ss.g) {
                    const bool reached_before = ss.g != SMPLX_INFINITECOST;
                    ss.g = (unsigned int)new_cost;
                    ss.bp = m;
                    dirty = true;
                    if (ss.iteration_closed != (unsigned short)W.iteration) {
                        ss.f = search_key(W.curr_eps, ss.g, ss.h);
                        if (sid == 0) W.goal_f = ss.f;
                        // a state reached for the first time in this call is not in OPEN; otherwise its position is read
                        // where the sifts keep it current
                        const int hi = reached_before ? P->st[sid].heap_index : 0;
                        if (hi != 0) {
                            hput(H, hi, hent_make(ss.f, sid));
                            if (ss.flags & 1u) heap_refresh_duplicates(H, W.heap_size, sid, ss.f);
                            heap_percolate_up(H, hi);
                        } else {
                            if ((ss.flags & 1u) && W.dup_pushes > 0) heap_refresh_duplicates(H, W.heap_size, sid, ss.f);
                            ++W.heap_size;
                            hset(H, W.heap_size, hent_make(ss.f, sid));
                            heap_percolate_up(H, W.heap_size);
                        }
                    } else {
                        P->incons[W.n_incons++] = sid;      // (never marked: arastar.cpp:563-565)
                    }
                }
                if (dirty) sstate_store(&P->st[sid], ss, false);
            }
            ++W.num;
            SK_TICK(4);
        }
        __syncthreads();
    }

    // ---- the launch ends: results, or the state the next launch picks up ----
    __syncthreads();
    {
        const int n = W.heap_size + 1 < lh ? W.heap_size + 1 : lh;
        for (int i = t; i < n; i += blockDim.x) H.hbm[i] = H.lds[i];
    }
    if (t == 0) {
        int solved = Ph.solved, cost = Ph.cost, n_path = Ph.n_path;
        if (W.phase == 3) {
            // arastar.cpp:199-214
            W.expand_count += W.num;
            W.status = SMPLX_SS_DONE;
            if (W.satisfied_eps == __builtin_inf()) {
                solved = 0; cost = 0; n_path = 0;
            } else {
                int n = 0;
                for (int sid = 0; sid >= 0 && n < P->cap_path; sid = P->st[sid].bp) P->path[n++] = sid;
                n_path = n;
                cost = (int)P->st[0].g;
                solved = 1;
            }
            W.phase = 4;
        }
        Pd->solved = solved; Pd->cost = cost; Pd->n_path = n_path;
        Pd->curr_eps = W.curr_eps; Pd->satisfied_eps = W.satisfied_eps;
        Pd->heap_size = W.heap_size; Pd->nstates = W.nstates; Pd->n_incons = W.n_incons; Pd->n_log = W.n_log; Pd->n_succ = W.n_succ;
        Pd->iteration = W.iteration; Pd->call_number = W.call_number; Pd->phase = W.phase; Pd->num = W.num;
        Pd->expand_count = W.expand_count; Pd->expand_count_init = W.expand_count_init; Pd->err = W.err;
        Pd->dup_pushes = W.dup_pushes; Pd->goal_f = W.goal_f;
        Pd->status = W.status; Pd->grow_what = W.grow_what;
        Pd->committed_evals = W.committed_evals; Pd->gpu_evals = W.gpu_evals; Pd->lookups = W.lookups;
        SK_TICK(6);
        for (int k = 0; k < 8; ++k) Pd->ticks[k] = W.ticks[k];
        if (status_out) status_out[blockIdx.x] = W.status;
    }
#undef SK_TICK
}

// getOrCreateState for states the device table does not hold yet (the start state the host created; every state after
// the table was enlarged): inserts ids [first, n) of the query's coordinate array
extern "C" __global__ void __launch_bounds__(256)
k_search_table_fill(const SmplxSpaceDev* __restrict__ Sq, const int* __restrict__ coord, int first, int n, int nvars)
{
    const SmplxTableDev T = Sq->table;
    for (int id = first + blockIdx.x * blockDim.x + threadIdx.x; id < n; id += gridDim.x * blockDim.x) {
        if (id == 0) continue;      // the goal id has no coordinate (manip_lattice.cpp:122)
        const int* c = coord + (size_t)id * nvars;
        unsigned int k = smplx_coord_hash(c, nvars) & T.mask;
        while (true) {
            int* sl = T.slots + (size_t)k * T.stride;
            if (atomicCAS(&sl[0], 0, -(id + 1)) == 0) {
                for (int v = 0; v < nvars; ++v) sl[1 + v] = c[v];
                __threadfence();
                __atomic_store_n(&sl[0], id + 1, __ATOMIC_RELAXED);
                break;
            }
            k = (k + 1) & T.mask;
        }
    }
}

// Parity-test kernel (test_hooks.h): the heap primitives of k_search driven by an op sequence in the language of
// oracle/heap_ref_driver.cpp -- 0 push(priority), 1 pop, 2 / 5 decrease / increase(element << 20 | priority), 3 erase(element),
// 4 re-prioritise everything and make() -- so that they can be compared with the reference's own intrusive_heap
// (tests/golden/heap_ref.json).  Element e is state e; the first `lh` heap entries live in LDS, the rest in HBM.
extern "C" __global__ void __launch_bounds__(256)
k_heap_ops(const int* __restrict__ ops, int nops, int lh, unsigned long long* __restrict__ heap_hbm, SmplxSState* __restrict__ st,
           int* __restrict__ top_after)
{
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int s_size, s_make;
    HeapRef H;
    H.lds = (LDS_AS hent_t*)smem;
    H.hbm = heap_hbm;
    H.st = st;
    H.lh = lh;
    const int t = threadIdx.x;
    if (t == 0) { s_size = 0; s_make = 0; }
    __syncthreads();
    int nelem = 0;
    for (int i = 0; i < nops; ++i) {
        const int code = ops[2 * i], key = ops[2 * i + 1];
        if (t == 0) {
            int size = s_size;
            if (code == 0) {
                st[nelem].f = (unsigned int)key;
                st[nelem].heap_index = 0;
                ++size;
                hset(H, size, hent_make((unsigned int)key, nelem));
                heap_percolate_up(H, size);
            } else if (code == 1) {
                if (size > 0) {
                    st[hent_id(hget(H, 1))].heap_index = 0;
                    const hent_t last = hget(H, size);
                    --size;
                    if (size >= 1) { hput(H, 1, last); heap_percolate_down(H, 1, size); }
                }
            } else if (code == 2 || code == 5) {
                const int e = key >> 20, p = key & 0xFFFFF;
                if (e < nelem && st[e].heap_index != 0) {
                    const int hi = st[e].heap_index;
                    st[e].f = (unsigned int)p;
                    hput(H, hi, hent_make((unsigned int)p, e));
                    if (code == 2) heap_percolate_up(H, hi); else heap_percolate_down(H, hi, size);
                }
            } else if (code == 3) {
                if (key < nelem && st[key].heap_index != 0) {      // intrusive_heap.hpp:197-206
                    const int pos = st[key].heap_index;
                    const hent_t last = hget(H, size);
                    hset(H, pos, last);
                    st[key].heap_index = 0;
                    --size;
                    heap_percolate_down(H, pos, size);
                }
            }
            s_size = size;
            s_make = code == 4;
        }
        if (code == 0) ++nelem;
        __syncthreads();
        if (s_make) {
            const int size = s_size;
            for (int k = 1 + t; k <= size; k += blockDim.x) {
                const int e = hent_id(hget(H, k));
                const unsigned int p = (st[e].f * 7919u + 13u) % 1000u;
                st[e].f = p;
                hput(H, k, hent_make(p, e));
            }
            __syncthreads();
            heap_make_block(H, size, false);
        }
        if (t == 0) top_after[i] = s_size > 0 ? hent_id(hget(H, 1)) : -1;
        __syncthreads();
    }
}
