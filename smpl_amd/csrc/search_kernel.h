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

#define SMPLX_AC_LEVELS 12
#define SMPLX_AC_SLOTS 128

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
    int steps_left;
    // ancestors of the slots the pushes of this relaxation will take, as far as they lie in HBM: level j (1 = parents)
    // holds heap indices ac_lo[j] .. ac_hi[j] at ac_val[ac_base[j] ...]; written through by every sift of the relaxation
    int ac_nlev;
    int ac_lo[SMPLX_AC_LEVELS + 1], ac_hi[SMPLX_AC_LEVELS + 1], ac_base[SMPLX_AC_LEVELS + 1];
    unsigned long long ac_val[SMPLX_AC_SLOTS];
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

// ---- sifts of the relaxation: reads of HBM-resident ancestors come from the cache built before it, writes go to both ----
__device__ __forceinline__ void ac_update(SearchLds& W, int i, hent_t e)
{
    for (int j = 1; j <= W.ac_nlev; ++j)
        if (i >= W.ac_lo[j] && i <= W.ac_hi[j]) { W.ac_val[W.ac_base[j] + i - W.ac_lo[j]] = e; return; }
}
__device__ __forceinline__ void hset_r(const HeapRef& H, SearchLds& W, int i, hent_t e)
{
    if (i < H.lh) H.lds[i] = e;
    else { H.hbm[i] = e; ac_update(W, i, e); }
    H.st[hent_id(e)].heap_index = i;
}
// Every OPEN entry of state `id` takes the state's new f.  Only needed once a state has been pushed while already in OPEN
// (the reference appends a state to INCONS once per improvement, arastar.cpp:563-565, and pushes every INCONS entry,
// :180-184): its heap then holds the same element twice, and both see an f change because both point to it.
__device__ __forceinline__ void heap_refresh_duplicates_r(const HeapRef& H, SearchLds& W, int size, int id, unsigned int f)
{
    for (int i = 1; i <= size; ++i) {
        const hent_t e = hget(H, i);
        if (hent_id(e) == id && hent_f(e) != f) {
            const hent_t ne = hent_make(f, id);
            if (i < H.lh) H.lds[i] = ne; else { H.hbm[i] = ne; ac_update(W, i, ne); }
        }
    }
}
// percolate_up from an arbitrary position (decrease-key)
__device__ __forceinline__ void heap_percolate_up_r(const HeapRef& H, SearchLds& W, int pivot)
{
    const hent_t tmp = hget(H, pivot);
    while (pivot != 1) {
        const int p = pivot >> 1;
        const hent_t ep = hget(H, p);
        if (hent_f(ep) < hent_f(tmp)) break;
        hset_r(H, W, pivot, ep);
        pivot = p;
    }
    hset_r(H, W, pivot, tmp);
}
// push (intrusive_heap.hpp:145-153) into slot `pivot` = size + 1: the walk towards the root reads LDS only
__device__ __forceinline__ void heap_push_r(const HeapRef& H, SearchLds& W, int pivot, hent_t e)
{
    int j = 0;    // level above the slot
    while (pivot != 1) {
        const int p = pivot >> 1;
        hent_t ep;
        if (p < H.lh) ep = H.lds[p];
        else if (j + 1 <= W.ac_nlev && p >= W.ac_lo[j + 1] && p <= W.ac_hi[j + 1]) ep = W.ac_val[W.ac_base[j + 1] + p - W.ac_lo[j + 1]];
        else ep = H.hbm[p];
        if (hent_f(ep) < hent_f(e)) break;
        // the ancestor moves down one level
        if (pivot < H.lh) H.lds[pivot] = ep;
        else {
            H.hbm[pivot] = ep;
            if (j >= 1 && j <= W.ac_nlev && pivot >= W.ac_lo[j] && pivot <= W.ac_hi[j]) W.ac_val[W.ac_base[j] + pivot - W.ac_lo[j]] = ep;
        }
        H.st[hent_id(ep)].heap_index = pivot;
        pivot = p;
        ++j;
    }
    if (pivot < H.lh) H.lds[pivot] = e;
    else {
        H.hbm[pivot] = e;
        if (j >= 1 && j <= W.ac_nlev && pivot >= W.ac_lo[j] && pivot <= W.ac_hi[j]) W.ac_val[W.ac_base[j] + pivot - W.ac_lo[j]] = e;
    }
    H.st[hent_id(e)].heap_index = pivot;
}

// percolate_down for the pop (intrusive_heap.hpp:346-377).  Below the LDS part of the array every level would be a
// dependent HBM round trip: the three levels under the pivot (2 + 4 + 8 entries) are requested together instead and the
// walk continues in registers.
__device__ __forceinline__ hent_t hbm_or_absent(const HeapRef& H, int i, int size) { return i <= size ? H.hbm[i] : ~0ull; }
__device__ __forceinline__ void heap_percolate_down_pf(const HeapRef& H, int pivot, int size)
{
    if (pivot > size) return;
    const hent_t tmp = hget(H, pivot);
    bool done = false;
    while (!done) {
        const int left = pivot << 1;
        if (left > size) break;
        if (left + 1 < H.lh) {
            hent_t es = H.lds[left];
            int s = left;
            if (left + 1 <= size) {
                const hent_t er = H.lds[left + 1];
                if (!(hent_f(es) < hent_f(er))) { es = er; s = left + 1; }
            }
            if (hent_f(es) < hent_f(tmp)) { hset(H, pivot, es); pivot = s; }
            else done = true;
            continue;
        }
        // three levels below `pivot`, all loads in flight at once (an index beyond the heap reads as absent)
        const int b1 = left, b2 = left << 1, b3 = left << 2;
        hent_t c1[2], c2[4], c3[8];
#pragma unroll
        for (int k = 0; k < 2; ++k) c1[k] = b1 + k <= size ? hget(H, b1 + k) : ~0ull;
#pragma unroll
        for (int k = 0; k < 4; ++k) c2[k] = hbm_or_absent(H, b2 + k, size);
#pragma unroll
        for (int k = 0; k < 8; ++k) c3[k] = hbm_or_absent(H, b3 + k, size);
        // level 1
        int s1 = 0;
        hent_t e1 = c1[0];
        if (b1 + 1 <= size && !(hent_f(c1[0]) < hent_f(c1[1]))) { s1 = 1; e1 = c1[1]; }
        if (!(hent_f(e1) < hent_f(tmp))) { done = true; continue; }
        hset(H, pivot, e1);
        pivot = b1 + s1;
        // level 2: children of b1 + s1 are b2 + 2 s1 + {0, 1}
        const hent_t l2 = s1 ? c2[2] : c2[0], r2 = s1 ? c2[3] : c2[1];
        const int i2 = b2 + 2 * s1;
        if (i2 > size) { done = true; continue; }
        int s2 = 0;
        hent_t e2 = l2;
        if (i2 + 1 <= size && !(hent_f(l2) < hent_f(r2))) { s2 = 1; e2 = r2; }
        if (!(hent_f(e2) < hent_f(tmp))) { done = true; continue; }
        hset(H, pivot, e2);
        pivot = i2 + s2;
        // level 3: children of i2 + s2 are b3 + 4 s1 + 2 s2 + {0, 1}
        const int q = 2 * s1 + s2;
        const hent_t l3 = q == 0 ? c3[0] : (q == 1 ? c3[2] : (q == 2 ? c3[4] : c3[6]));
        const hent_t r3 = q == 0 ? c3[1] : (q == 1 ? c3[3] : (q == 2 ? c3[5] : c3[7]));
        const int i3 = b3 + 2 * q;
        if (i3 > size) { done = true; continue; }
        int s3 = 0;
        hent_t e3 = l3;
        if (i3 + 1 <= size && !(hent_f(l3) < hent_f(r3))) { s3 = 1; e3 = r3; }
        if (!(hent_f(e3) < hent_f(tmp))) { done = true; continue; }
        hset(H, pivot, e3);
        pivot = i3 + s3;
    }
    hset(H, pivot, tmp);
}

// one probe sequence of the state table with plain 16-byte loads: the id of the coordinate, or -1 and the empty slot the
// probing ended at.  Only the workgroup that owns the query writes its table while the kernel runs (plain stores, earlier
// in program order or before a barrier), so what it reads is current.
struct TableProbe { int id; unsigned int free_slot; };
__device__ __forceinline__ TableProbe table_probe_own(const SmplxTableDev& T, const LDS_AS int* c, int nv, unsigned int hash)
{
    TableProbe r;
    unsigned int i = hash & T.mask;
    while (true) {
        const sk_int4* sl = reinterpret_cast<const sk_int4*>(T.slots + (size_t)i * T.stride);
        sk_int4 w[4];
        const int nw = (nv + 1 + 3) / 4;       // 16-byte words that hold the tag and the coordinate (stride is a multiple of 8 ints)
#pragma unroll
        for (int k = 0; k < 4; ++k) if (k < nw) w[k] = sl[k];
        const int tag = w[0].x;
        if (tag == 0) { r.id = -1; r.free_slot = i; return r; }
        bool same = tag > 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k >= nw) continue;
            const int base = 4 * k - 1;        // coordinate index of .x
            if (k > 0 && base < nv) same = same && w[k].x == c[base];
            if (base + 1 < nv) same = same && w[k].y == c[base + 1];
            if (base + 2 < nv) same = same && w[k].z == c[base + 2];
            if (base + 3 < nv) same = same && w[k].w == c[base + 3];
        }
        if (same) { r.id = tag - 1; r.free_slot = 0; return r; }
        i = (i + 1) & T.mask;
    }
}
__device__ __forceinline__ void table_store_own(const SmplxTableDev& T, unsigned int slot, const LDS_AS int* c, int nv, int id)
{
    int* sl = T.slots + (size_t)slot * T.stride;
    for (int v = 0; v < nv; ++v) sl[1 + v] = c[v];
    sl[0] = id + 1;
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
        W.steps_left = max_steps;
        W.ac_nlev = 0;
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

    long long tick = 0;
    if (t == 0) tick = (long long)wall_clock64();
#define SK_TICK(k) do { const long long now_ = (long long)wall_clock64(); W.ticks[k] += now_ - tick; tick = now_; } while (0)

    // =========================== thread 0: what happens next (Search::resume / improve_path) ===========================
    // Decides the step -- expand state m, start a new epsilon, or leave -- WITHOUT popping yet: the pop's sift runs after the
    // barrier that publishes m, beside the first phase of the expansion.
    auto select = [&]() {
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
            if (W.steps_left <= 0) { action = SA_EXIT; break; }                      // this launch has done its share
            // room for one more expansion?  (checked before anything is popped: the host enlarges and launches again)
            if (W.nstates + nprims > P->cap_states || W.heap_size + nprims > P->cap_heap || W.n_incons + nprims > P->cap_incons ||
                W.n_log + 1 > P->cap_log || W.n_succ + nprims > P->cap_succ || (unsigned int)(2 * (W.nstates + nprims)) > table.mask + 1u) {
                W.status = SMPLX_SS_GROW; W.grow_what = 2;
                action = SA_EXIT;
                break;
            }
            --W.steps_left;
            W.m = hent_id(top);
            action = SA_EXPAND;
        }
        W.action = action;
    };
    if (t == 0) { select(); SK_TICK(1); }

    while (true) {
        __syncthreads();       // the step is decided; everything the previous step wrote is visible
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
                select();
                SK_TICK(1);
            }
            continue;
        }

        // =========================== expand state m (arastar.cpp:513-519, 531-568) ===========================
        const int m = W.m;
        const int off = P->done_off[m];              // >= 0: expanded before (a later ARA* iteration): the committed list serves
        if (t == 0) {
            // ---- pop (intrusive_heap.hpp:155-166); meanwhile the last wave forms the successors' joint values ----
            SmplxSState* sm = &P->st[m];
            const unsigned int g = sm->g;            // in flight together with the heap's last entry
            const hent_t last = hget(H, W.heap_size);
            sm->heap_index = 0;
            --W.heap_size;
            if (W.heap_size >= 1) {
                hput(H, 1, last);
                heap_percolate_down_pf(H, 1, W.heap_size);
            }
            sm->iteration_closed = (unsigned short)W.iteration;
            sm->eg = g;
            P->log[W.n_log++] = m;
            W.eg = g;
            SK_TICK(1);
        }
        if (off < 0) {
            // ---- GetSuccs loop body (manip_lattice.cpp:254-305) on the lanes ----
            expand_state_block(M, L, S, Sq, grid, X, P->q + (size_t)m * nv);
            if (t == 0) SK_TICK(2);
        }
        if (t >= 64) continue;     // the rest of the step is the first wave's; the others wait at the barrier on top
        // what thread 0 stored while popping (the state's closing, moved heap entries) is read below by OTHER lanes of this wave:
        // wait until those stores have landed (nothing is outstanding when the expansion's barriers lie in between)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");

        // ---- wave 0, lane p = primitive p: getOrCreateState for every valid successor (manip_lattice.cpp:1302-1354), then
        // the successors' search states and the HBM-resident ancestors of the slots their pushes will take ----
        const int lane = t;
        const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
        int cnt, sid = -1, k = -1;                   // this lane's successor: state id, position in the list
        bool valid, fresh = false;                   // fresh: the state was created by this expansion
        unsigned int fresh_h = 0;
        if (off >= 0) {
            const int dc = P->done_cnt[m];
            cnt = dc & 0xFF;
            valid = lane < cnt;
            k = lane;
            if (valid) {
                const SmplxSucc sc = P->succ[off + lane];
                sid = sc.id;
                W.succ_id[lane] = sid;
                W.succ_cost[lane] = sc.cost_prim & 0xFFFFFF;
            }
            if (lane == 0) { W.cnt = cnt; W.evals = dc >> 8; }
        } else {
            const bool in = lane < nprims;
            const int flags = in ? X.flags[lane] : SMPLX_F_INACTIVE;
            valid = (flags & SMPLX_F_VALID) != 0;
            int id = -1;
            unsigned int hash = 0, free_slot = 0;
            if (valid) {
                hash = coord_hash_lds((const LDS_AS int*)X.coord[lane], nv);
                const TableProbe pr = table_probe_own(table, (const LDS_AS int*)X.coord[lane], nv, hash);
                id = pr.id;
                free_slot = pr.free_slot;
            }
            // two successors of this expansion with the same new coordinate: the lower primitive creates the state
            const bool unknown = valid && id < 0;
            const unsigned long long m_unknown = __ballot(unknown);
            int dup_of = -1;
            for (int j = 0; j < nprims; ++j) {
                const unsigned int hj = (unsigned int)__shfl((int)hash, j);
                if (j < lane && unknown && dup_of < 0 && ((m_unknown >> j) & 1ull) && hj == hash) {
                    bool same = true;
                    for (int v = 0; v < nv; ++v) same = same && X.coord[j][v] == X.coord[lane][v];
                    if (same) dup_of = j;
                }
            }
            const bool is_new = unknown && dup_of < 0;
            const unsigned long long m_new = __ballot(is_new), m_valid = __ballot(valid), m_eval = __ballot(in && !(flags & SMPLX_F_INACTIVE));
            // two new coordinates whose probing ended at the same empty slot (rare): the later one probes again below
            bool clash = false;
            for (int j = 0; j < nprims; ++j) {
                const unsigned int fj = (unsigned int)__shfl((int)free_slot, j);
                if (j < lane && is_new && ((m_new >> j) & 1ull) && fj == free_slot) clash = true;
            }
            if (is_new) {
                id = W.nstates + __popcll(m_new & below);
                if (!clash) table_store_own(table, free_slot, (const LDS_AS int*)X.coord[lane], nv, id);
                for (int v = 0; v < nv; ++v) { P->coord[(size_t)id * nv + v] = X.coord[lane][v]; P->q[(size_t)id * nv + v] = X.sq[lane][v]; }
                SmplxSState ns;
                ns.h = (unsigned int)X.h[lane];
                sstate_reinit(ns, W.call_number);
                sstate_store(&P->st[id], ns, true);
                P->done_off[id] = -1;
                fresh_h = ns.h;
            }
            unsigned long long m_clash = __ballot(clash);
            while (m_clash) {      // uniform loop: one clashing lane at a time probes behind what the others have stored
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                const int j = __ffsll((long long)m_clash) - 1;
                m_clash &= m_clash - 1;
                if (lane == j) {
                    const TableProbe pr = table_probe_own(table, (const LDS_AS int*)X.coord[lane], nv, hash);
                    table_store_own(table, pr.free_slot, (const LDS_AS int*)X.coord[lane], nv, id);
                }
            }
            const int id_of_dup = __shfl(id, dup_of >= 0 ? dup_of : 0);
            if (dup_of >= 0) id = id_of_dup;
            fresh = is_new || dup_of >= 0;
            if (dup_of >= 0) fresh_h = (unsigned int)X.h[dup_of];
            cnt = __popcll(m_valid);
            if (valid) {
                k = __popcll(m_valid & below);
                sid = (flags & SMPLX_F_GOAL) ? 0 : id;      // a goal successor is reported as the goal id (manip_lattice.cpp:283-296)
                if (flags & SMPLX_F_GOAL) fresh = false;
                W.succ_id[k] = sid;
                W.succ_cost[k] = A.cost[lane];
                SmplxSucc sc;
                sc.id = sid;
                sc.cost_prim = A.cost[lane] | (lane << 24);
                P->succ[W.n_succ + k] = sc;
            }
            if (lane == 0) {
                const int evals = __popcll(m_eval);
                P->done_off[m] = W.n_succ;
                P->done_cnt[m] = cnt | (evals << 8);
                W.n_succ += cnt;
                W.nstates += __popcll(m_new);
                W.cnt = cnt; W.evals = evals;
                W.gpu_evals += evals;
            }
            // (grid lookups of this expansion, as the reference would count them)
            int lk = in ? X.lookups[lane] : 0;
            for (int o = 32; o > 0; o >>= 1) lk += __shfl_down(lk, o);
            if (lane == 0) W.lookups += lk;
        }
        // which successors name the same state (two goal successors; a primitive that lands in its parent's cell)
        int alias = -1;
        {
            const unsigned long long m_valid = __ballot(valid);
            for (int j = 0; j < 64; ++j) {
                if (!((m_valid >> j) & 1ull)) continue;      // uniform
                const int sj = __shfl(sid, j), kj = __shfl(k, j);
                if (valid && j < lane && alias < 0 && sj == sid) alias = kj;
            }
        }
        if (valid) {
            W.alias[k] = alias;
            if (fresh && alias < 0) {
                SmplxSState ns;                                 // what this expansion has just stored for the new state
                ns.h = fresh_h;
                sstate_reinit(ns, W.call_number);
                W.sst[k] = ns;
            } else if (alias < 0) {
                W.sst[k] = P->st[sid];
            }
        }
        {
            // ancestors of the slots n + 1 .. n + cnt (where this relaxation's pushes go) that lie beyond the LDS part
            const int n = W.heap_size;
            int nlev = 0, total = 0;
            int my_idx[2] = {-1, -1}, my_slot[2] = {0, 0};
            for (int j = 1; j <= SMPLX_AC_LEVELS; ++j) {
                int lo = (n + 1) >> j, hi = (n + cnt) >> j;
                if (hi > n) hi = n;
                if (hi < H.lh || cnt == 0) break;
                if (lo < H.lh) lo = H.lh;
                const int len = hi - lo + 1;
                if (total + len > SMPLX_AC_SLOTS) break;
                if (lane == 0) { W.ac_lo[j] = lo; W.ac_hi[j] = hi; W.ac_base[j] = total; }
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int e = lane + 64 * r - total;
                    if (e >= 0 && e < len) { my_idx[r] = lo + e; my_slot[r] = lane + 64 * r; }
                }
                total += len;
                nlev = j;
            }
            if (lane == 0) W.ac_nlev = nlev;
#pragma unroll
            for (int r = 0; r < 2; ++r)
                if (my_idx[r] >= 0) W.ac_val[my_slot[r]] = H.hbm[my_idx[r]];
        }
        SMPLX_WAVE_SYNC();
        if (t == 0) {
            SK_TICK(3);
            // ---- ARAStar::expand's loop over the successors, in primitive order (arastar.cpp:540-567) ----
            W.committed_evals += W.evals;
            const unsigned int eg = W.eg;
            const int n_succ = W.cnt;
            for (int i = 0; i < n_succ; ++i) {
                const int id = W.succ_id[i];
                const int al = W.alias[i];
                SmplxSState& ss = W.sst[al >= 0 ? al : i];
                bool dirty = false;
                if (ss.call_number != (unsigned short)W.call_number) { sstate_reinit(ss, W.call_number); P->st[id].heap_index = 0; dirty = true; }
                const int new_cost = (int)(eg + (unsigned int)W.succ_cost[i]);
                if ((unsigned int)new_cost < ss.g) {
                    const bool reached_before = ss.g != SMPLX_INFINITECOST;
                    ss.g = (unsigned int)new_cost;
                    ss.bp = m;
                    dirty = true;
                    if (ss.iteration_closed != (unsigned short)W.iteration) {
                        ss.f = search_key(W.curr_eps, ss.g, ss.h);
                        if (id == 0) W.goal_f = ss.f;
                        // a state reached for the first time in this call is not in OPEN; otherwise its position is read
                        // where the sifts keep it current
                        const int hi = reached_before ? P->st[id].heap_index : 0;
                        if (hi != 0) {
                            const hent_t e = hent_make(ss.f, id);
                            if (hi < H.lh) H.lds[hi] = e; else { H.hbm[hi] = e; ac_update(W, hi, e); }
                            if (ss.flags & 1u) heap_refresh_duplicates_r(H, W, W.heap_size, id, ss.f);
                            heap_percolate_up_r(H, W, hi);
                        } else {
                            if ((ss.flags & 1u) && W.dup_pushes > 0) heap_refresh_duplicates_r(H, W, W.heap_size, id, ss.f);
                            ++W.heap_size;
                            heap_push_r(H, W, W.heap_size, hent_make(ss.f, id));
                        }
                    } else {
                        P->incons[W.n_incons++] = id;      // (never marked: arastar.cpp:563-565)
                    }
                }
                if (dirty) sstate_store(&P->st[id], ss, false);
            }
            W.ac_nlev = 0;
            ++W.num;
            SK_TICK(4);
            select();
            SK_TICK(1);
        }
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
