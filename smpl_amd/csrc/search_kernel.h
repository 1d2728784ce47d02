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
//   the LAST wave   ("search wave") runs the search: the replan state machine, pop, the successors' joint values, limits,
//                   coordinates, state-table probes, goal test and heuristic (lane p = primitive p), getOrCreateState, the
//                   relaxation in primitive order (the order decides ties in OPEN), INCONS.  It works as ONE unit: what a
//                   sequential program keeps in variables lives in its registers (uniform over the lanes), per-successor data
//                   in the lane of the primitive, and a value another lane holds is read with v_readlane -- a single lane
//                   walking LDS-resident structures costs a ~100-cycle round trip per step, which is what round 3 first
//                   measured (13 us of relaxation per expansion).
//                   The sifts of the heap are wave-parallel and exact: sift-up reads ALL ancestors of the slot at once
//                   (lane j: index >> j), a ballot finds the level the sequential loop would stop at, the ancestors below it
//                   move down in one scatter; sift-down fetches the five levels under the pivot at once (62 lanes) and
//                   resolves the path in registers.
//   the other waves check configurations: one waypoint of one edge per lane (expand_config_lane), between the two barriers
//                   of a step.
// The first `lh` entries of the heap array live in LDS while the kernel runs, the rest in HBM; the HBM-resident ancestors of
// the slots a relaxation's pushes will take are fetched into LDS in one round trip before it (written through by every sift).
//
// A launch runs at most `max_steps` expansions and then stores its state in the query's SmplxSearchDev, so that the host
// sees progress, can stop a search, and can enlarge buffers (SMPLX_SS_GROW) between launches.

#define SMPLX_AC_LEVELS 16
#define SMPLX_AC_SLOTS 128
#define SMPLX_INFINITECOST 1000000000u     // SBPL INFINITECOST

enum { SA_EVAL = 0, SA_SKIP = 1, SA_REORDER = 2, SA_EXIT = 3 };

// what the waves of the block share
struct SearchLds {
    int action;                            // the round, published before its first barrier
    int buf;                               // ... and the ExpandLds it works on
    int seq;                               // ... and its number
    int helper_gave_up;                    // the helper wave's wait for `go` ran out (never seen; turns into SMPLX_SS_ERROR)
    int go;                                // = seq once the state table holds everything committed before this round's state was
                                           //   popped: the helper wave may probe it (the search wave sets it before it joins the round's end)
    double reorder_eps;
    int reorder_size, reorder_dups;
    // ancestors of the slots the pushes of a relaxation will take, as far as they lie in HBM: level j (1 = parents) holds
    // heap indices ac_lo[j] .. ac_hi[j] at ac_val[ac_base[j] ...]
    int ac_nlev;
    int ac_lo[SMPLX_AC_LEVELS + 1], ac_hi[SMPLX_AC_LEVELS + 1], ac_base[SMPLX_AC_LEVELS + 1];
    unsigned long long ac_val[SMPLX_AC_SLOTS];
};

// a heap entry as one 64-bit word (the layout of SmplxHeapEntry: f in the low half, id in the high half): LDS and HBM
// copies are single loads / stores
typedef unsigned long long hent_t;
#define HENT_ABSENT (~0ull)
__device__ __forceinline__ unsigned int hent_f(hent_t e) { return (unsigned int)e; }
__device__ __forceinline__ int hent_id(hent_t e) { return (int)(e >> 32); }
__device__ __forceinline__ hent_t hent_make(unsigned int f, int id) { return (hent_t)f | ((hent_t)(unsigned int)id << 32); }

// what the helper wave leaves for the search wave at the end of a round: lane = primitive
struct BookOut {
    int limits_ok[64], wcount[64], h[64], is_goal[64];
    unsigned int hash[64], free_slot[64];
    int pr_id[64], dup_of[64], clash[64];
    int ss[8][64];                         // the successor's search state as stored (sstate_load's two 16-byte words)
};

struct HeapRef {
    LDS_AS hent_t* lds;                    // entries [0, lh)
    SMPLX_GLOBAL_AS hent_t* hbm;           // entries [lh, ...)
    SMPLX_GLOBAL_AS SmplxSState* st;
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

// ---- the sequential forms (one thread): the reference's loops as they stand.  Used by the level-parallel make() and as the
// specification of the wave-parallel forms below ----

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

// ---- wave-level helpers: every lane of the wave is active and `lane_index` arguments are uniform ----
__device__ __forceinline__ int wave_rl(int v, int lane_index) { return __builtin_amdgcn_readlane(v, lane_index); }
__device__ __forceinline__ unsigned int wave_rlu(unsigned int v, int lane_index) { return (unsigned int)__builtin_amdgcn_readlane((int)v, lane_index); }
__device__ __forceinline__ hent_t wave_rl64(hent_t v, int lane_index)
{
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)v, lane_index);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(v >> 32), lane_index);
    return (hent_t)lo | ((hent_t)hi << 32);
}

// a write to the HBM part of the heap also refreshes the ancestor cache where it holds that index
__device__ __forceinline__ void ac_update(SearchLds& W, int i, hent_t e)
{
    const int nlev = W.ac_nlev;
    for (int j = 1; j <= nlev; ++j)
        if (i >= W.ac_lo[j] && i <= W.ac_hi[j]) { W.ac_val[W.ac_base[j] + i - W.ac_lo[j]] = e; return; }
}
__device__ __forceinline__ void hset_ac(const HeapRef& H, SearchLds& W, int i, hent_t e)
{
    if (i < H.lh) H.lds[i] = e;
    else { H.hbm[i] = e; ac_update(W, i, e); }
    H.st[hent_id(e)].heap_index = i;
}

// percolate_up by the whole wave (intrusive_heap.hpp:379-395): entry e sifts up from slot `start` (a push: the new last slot;
// decrease: the entry's slot).  Lane j reads the ancestor start >> j; the sequential loop compares every ancestor IN ITS
// ORIGINAL PLACE with e and stops at the first that is smaller, so a ballot finds that level; the ancestors below it move
// down one level each, e takes the freed slot.  `cached` = the ancestor cache covers the walk (a push of the relaxation it
// was built for): HBM-resident ancestors are read from LDS.
__device__ __forceinline__ void heap_sift_up_wave(const HeapRef& H, SearchLds& W, int lane, int start, hent_t e, bool cached)
{
    const int j = lane;
    const int idx = j < 31 ? (start >> j) : 0;
    const bool anc = j >= 1 && idx >= 1;
    hent_t ent = 0;
    if (anc) {
        if (idx < H.lh) ent = H.lds[idx];
        else if (cached && j <= W.ac_nlev && idx >= W.ac_lo[j] && idx <= W.ac_hi[j]) ent = W.ac_val[W.ac_base[j] + idx - W.ac_lo[j]];
        else ent = H.hbm[idx];
    }
    const unsigned long long m_anc = __ballot(anc), m_stop = __ballot(anc && hent_f(ent) < hent_f(e));
    const int jstop = m_stop ? __ffsll((long long)m_stop) - 1 : __popcll(m_anc) + 1;   // ancestors are lanes 1 .. popc(m_anc)
    if (!cached) {
        if (anc && j < jstop) hset_ac(H, W, start >> (j - 1), ent);
        if (j == 0) hset_ac(H, W, start >> (jstop - 1), e);
        return;
    }
    // a walk the cache was built for: a write to level l of the walk can only concern the cache's level l (level 0, the new
    // slots themselves, is not cached)
    const bool mover = anc && j < jstop;
    if (mover || j == 0) {
        const int l = mover ? j - 1 : jstop - 1;
        const int dst = start >> l;
        const hent_t v = mover ? ent : e;
        if (dst < H.lh) H.lds[dst] = v;
        else {
            H.hbm[dst] = v;
            if (l >= 1 && l <= W.ac_nlev && dst >= W.ac_lo[l] && dst <= W.ac_hi[l]) W.ac_val[W.ac_base[l] + dst - W.ac_lo[l]] = v;
        }
        H.st[hent_id(v)].heap_index = dst;
    }
}

// percolate_down by the whole wave (intrusive_heap.hpp:346-377): entry tmp sifts down from `pivot`.  Lanes 0 .. 61 fetch the
// five levels under the pivot at once; the path is resolved in registers (the loop reads every child in its original place:
// moves only ever write to slots ABOVE what is read next); the chosen entries move up in one scatter.
__device__ __forceinline__ void heap_sift_down_wave(const HeapRef& H, int lane, int pivot, int size, hent_t tmp)
{
    const int lv = 31 - __clz(lane + 2);                    // level below the pivot: 1 .. 5 for lanes 0 .. 61
    const int off = lane + 2 - (1 << lv);
    while (true) {
        if ((pivot << 1) > size) break;
        const int idx = (pivot << lv) + off;
        hent_t ent = HENT_ABSENT;
        if (lane < 62 && idx <= size) ent = hget(H, idx);
        int c = 0, depth = 0;
        unsigned long long chosen = 0;
        bool stopped = false;
#pragma unroll
        for (int l = 1; l <= 5; ++l) {
            if (stopped) continue;
            const int li = (1 << l) - 2 + 2 * c;
            const hent_t el = wave_rl64(ent, li), er = wave_rl64(ent, li + 1);
            if (el == HENT_ABSENT) { stopped = true; continue; }
            const bool right = er != HENT_ABSENT && !(hent_f(el) < hent_f(er));
            const hent_t es = right ? er : el;
            if (!(hent_f(es) < hent_f(tmp))) { stopped = true; continue; }
            chosen |= 1ull << (li + (right ? 1 : 0));
            c = 2 * c + (right ? 1 : 0);
            depth = l;
        }
        if ((chosen >> lane) & 1ull) hset(H, idx >> 1, ent);
        pivot = (pivot << depth) + c;
        if (stopped) break;
        // the next round reads below what this one moved: nothing it reads was written
    }
    if (lane == 0) hset(H, pivot, tmp);
}

// ARAStar::computeKey (arastar.cpp:579-582)
__device__ __forceinline__ unsigned int search_key(double eps, unsigned int g, unsigned int h)
{
    return g + (unsigned int)(long long)(eps * (double)h);
}

// Every OPEN entry of state `id` takes the state's new f (whole wave).  Only needed once a state has been pushed while
// already in OPEN (the reference appends a state to INCONS once per improvement, arastar.cpp:563-565, and pushes every
// INCONS entry, :180-184): its heap then holds the same element twice, and both see an f change because both point to it.
__device__ __forceinline__ void heap_refresh_duplicates_wave(const HeapRef& H, SearchLds& W, int lane, int size, int id, unsigned int f)
{
    for (int i = 1 + lane; i <= size; i += 64) {
        const hent_t e = hget(H, i);
        if (hent_id(e) == id && hent_f(e) != f) {
            const hent_t ne = hent_make(f, id);
            if (i < H.lh) H.lds[i] = ne; else { H.hbm[i] = ne; ac_update(W, i, ne); }
        }
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

// write a state's fields back; heap_index only when the caller owns it (it is otherwise kept current by the sifts)
__device__ __forceinline__ void sstate_store(SMPLX_GLOBAL_AS SmplxSState* dst, const SmplxSState& s, bool with_heap_index)
{
    sk_int4 a;
    a.x = (int)s.g; a.y = (int)s.h; a.z = (int)s.f; a.w = (int)s.eg;
    *(SMPLX_GLOBAL_AS sk_int4*)dst = a;
    dst->bp = s.bp;
    if (with_heap_index) dst->heap_index = s.heap_index;
    *(SMPLX_GLOBAL_AS unsigned int*)&dst->iteration_closed = (unsigned int)s.iteration_closed | ((unsigned int)s.call_number << 16);
    dst->flags = s.flags;
}
__device__ __forceinline__ SmplxSState sstate_load(const SMPLX_GLOBAL_AS SmplxSState* src)
{
    const sk_int4 a = ((const SMPLX_GLOBAL_AS sk_int4*)src)[0], b = ((const SMPLX_GLOBAL_AS sk_int4*)src)[1];
    SmplxSState s;
    s.g = (unsigned int)a.x; s.h = (unsigned int)a.y; s.f = (unsigned int)a.z; s.eg = (unsigned int)a.w;
    s.bp = b.x; s.heap_index = b.y;
    s.iteration_closed = (unsigned short)((unsigned int)b.z & 0xFFFFu); s.call_number = (unsigned short)((unsigned int)b.z >> 16);
    s.flags = (unsigned int)b.w;
    return s;
}

// One probe sequence of the state table with plain 16-byte loads.  Only the workgroup that owns the query writes its table
// while the kernel runs, so what it reads is current.  table_probe_start issues the loads of the home slot (they land
// behind the planning-link FK); table_probe_finish looks at them and walks on if it has to.
struct TableProbe { int id; unsigned int free_slot; };
struct TableProbeLoads { sk_int4 w[4]; unsigned int slot; };
__device__ __forceinline__ void table_slot_load(const SmplxTableDev& T, unsigned int slot, int nv, sk_int4 w[4])
{
    const SMPLX_GLOBAL_AS sk_int4* sl = (const SMPLX_GLOBAL_AS sk_int4*)(as_global(T.slots) + (size_t)slot * T.stride);
    const int nw = (nv + 1 + 3) / 4;       // 16-byte words that hold the tag and the coordinate (stride is a multiple of 8 ints)
#pragma unroll
    for (int k = 0; k < 4; ++k) if (k < nw) w[k] = sl[k];
}
__device__ __forceinline__ bool table_slot_match(const sk_int4 w[4], const LDS_AS int* c, int nv)
{
    const int nw = (nv + 1 + 3) / 4;
    bool same = w[0].x > 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= nw) continue;
        const int base = 4 * k - 1;        // coordinate index of .x
        if (k > 0 && base < nv) same = same && w[k].x == c[base];
        if (base + 1 < nv) same = same && w[k].y == c[base + 1];
        if (base + 2 < nv) same = same && w[k].z == c[base + 2];
        if (base + 3 < nv) same = same && w[k].w == c[base + 3];
    }
    return same;
}
__device__ __forceinline__ TableProbeLoads table_probe_start(const SmplxTableDev& T, int nv, unsigned int hash)
{
    TableProbeLoads r;
    r.slot = hash & T.mask;
    table_slot_load(T, r.slot, nv, r.w);
    return r;
}
__device__ __forceinline__ TableProbe table_probe_finish(const SmplxTableDev& T, TableProbeLoads& ld, const LDS_AS int* c, int nv)
{
    TableProbe r;
    while (true) {
        if (ld.w[0].x == 0) { r.id = -1; r.free_slot = ld.slot; return r; }
        if (table_slot_match(ld.w, c, nv)) { r.id = ld.w[0].x - 1; r.free_slot = 0; return r; }
        ld.slot = (ld.slot + 1) & T.mask;
        table_slot_load(T, ld.slot, nv, ld.w);
    }
}
__device__ __forceinline__ void table_store_own(const SmplxTableDev& T, unsigned int slot, const LDS_AS int* c, int nv, int id)
{
    SMPLX_GLOBAL_AS int* sl = as_global(T.slots) + (size_t)slot * T.stride;
    for (int v = 0; v < nv; ++v) sl[1 + v] = c[v];
    sl[0] = id + 1;
}

// Ancestors of the slots n + 1 .. n + cnt (where a relaxation's pushes go) that lie beyond the LDS part of the heap: one
// round trip, all lanes of the search wave.  Returns whether every such ancestor fits the cache.
__device__ __forceinline__ bool ancestor_cache_fill(const HeapRef& H, SearchLds& W, int lane, int n, int cnt)
{
    int nlev = 0, total = 0;
    int my_idx[2] = {-1, -1};
    bool complete = true;
    for (int j = 1; j <= SMPLX_AC_LEVELS; ++j) {
        int lo = (n + 1) >> j, hi = (n + cnt) >> j;
        if (hi > n) hi = n;
        if (hi < H.lh || cnt == 0) break;
        if (lo < H.lh) lo = H.lh;
        const int len = hi - lo + 1;
        if (total + len > SMPLX_AC_SLOTS || j == SMPLX_AC_LEVELS) { complete = false; break; }
        if (lane == 0) { W.ac_lo[j] = lo; W.ac_hi[j] = hi; W.ac_base[j] = total; }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int e = lane + 64 * r - total;
            if (e >= 0 && e < len) my_idx[r] = lo + e;
        }
        total += len;
        nlev = j;
    }
    if (lane == 0) W.ac_nlev = nlev;
#pragma unroll
    for (int r = 0; r < 2; ++r)
        if (my_idx[r] >= 0) W.ac_val[lane + 64 * r] = H.hbm[my_idx[r]];
    return complete;
}

// the search's variables: registers of the search wave, the same value in every lane
struct SearchRegs {
    double curr_eps, satisfied_eps;
    int heap_size, nstates, n_incons, n_log, n_succ;
    int iteration, call_number, phase, num, expand_count, expand_count_init, err;
    int dup_pushes, status, grow_what, steps_left;
    unsigned int goal_f;
    long long committed_evals, gpu_evals, lookups;
};

// getOrCreateState's table side (manip_lattice.cpp:1302-1354) for the successors of one expansion, by a whole wave (lane =
// primitive; every lane calls): hash and probe of each candidate's coordinate, the search state of a coordinate the table
// knows, and among the unknown ones the pairs that share a coordinate (the lower primitive creates the state) or whose
// probing ended at the same empty slot.  WithGoalFK: the planning-link FK / BFS cell / goal test of the successor is done
// here, between the probe's loads and their use (the search wave without a helper wave); otherwise b holds them already.
struct BookRegs { unsigned int hash; TableProbe pr; int dup_of; bool clash; };
template <bool WithGoalFK>
__device__ __forceinline__ void search_book_table(const ModelLds* __restrict__ M, const SmplxGridDev& grid, const SmplxBfsDev& bfs,
                                                  const SmplxGoalDev& G, const SmplxTableDev& table, SMPLX_GLOBAL_AS SmplxSState* st,
                                                  ExpandLds& X, int lane, int nv, bool act, BookLane& b, BookRegs& o, SmplxSState& ss)
{
    o.hash = 0; o.pr.id = -1; o.pr.free_slot = 0; o.dup_of = -1; o.clash = false;
    if (act && b.limits_ok) {
        // getHashEntry (manip_lattice.cpp:1302-1316): the home slot's loads travel behind the planning-link FK; a coordinate the
        // table knows has its search state requested at once (it lands behind the waypoint lanes)
        o.hash = coord_hash_lds((const LDS_AS int*)X.coord[lane], nv);
        TableProbeLoads ld = table_probe_start(table, nv, o.hash);
        if constexpr (WithGoalFK) expand_book_goal(M, grid, bfs, G, X, lane, b);
        o.pr = table_probe_finish(table, ld, (const LDS_AS int*)X.coord[lane], nv);
        if (o.pr.id >= 0 && (!WithGoalFK || !b.is_goal)) ss = sstate_load(&st[o.pr.id]);   // (a goal successor takes the goal's state)
    }
    const bool c_unknown = act && b.limits_ok && o.pr.id < 0;
    {
        unsigned long long members = __ballot(c_unknown);
        while (members) {                                // uniform
            const int j = __ffsll((long long)members) - 1;
            members &= members - 1;
            const unsigned int hj = wave_rlu(o.hash, j);
            if (j < lane && c_unknown && o.dup_of < 0 && hj == o.hash) {
                bool same = true;
                for (int v = 0; v < nv; ++v) same = same && X.coord[j][v] == X.coord[lane][v];
                if (same) o.dup_of = j;
            }
        }
    }
    {
        const bool mine = c_unknown && o.dup_of < 0;
        unsigned long long members = __ballot(mine);
        while (members) {
            const int j = __ffsll((long long)members) - 1;
            members &= members - 1;
            const unsigned int fj = wave_rlu(o.pr.free_slot, j);
            if (j < lane && mine && fj == o.pr.free_slot) o.clash = true;
        }
    }
}

extern "C" __global__ void __launch_bounds__(512)
k_search(const SmplxSpaceDev* const* __restrict__ stab, int max_steps, int lh, int* __restrict__ status_out)
{
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ ExpandLds Xb[2];                  // two evaluations can be open: the one being committed and the next, speculative one
    __shared__ SearchLds W;
    __shared__ SmplxSearchDev Ph;                // the query's header as the launch found it: pointers, capacities, parameters
    __shared__ SmplxGoalDev Gh;                  // ... and the goal
    __shared__ BookOut Bo;                       // helper wave -> search wave, once per round
    __shared__ SmplxActionsDev Ah;               // the primitives: read by the search wave several times per expansion, each read a round trip to L2 otherwise
    static_assert(2 * sizeof(ExpandLds) + sizeof(SearchLds) + sizeof(SmplxSearchDev) + sizeof(SmplxActionsDev) + sizeof(SmplxGoalDev) + sizeof(BookOut) <= SMPLX_SEARCH_STATIC_LDS, "engine.hip budgets this much static LDS");
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
    for (int i = t; i < (int)(sizeof(SmplxActionsDev) / 4); i += blockDim.x) ((int*)&Ah)[i] = ((const int*)&S->actions)[i];
    for (int i = t; i < (int)(sizeof(SmplxGoalDev) / 4); i += blockDim.x) ((int*)&Gh)[i] = ((const int*)&Sq->goal)[i];
    if (t == 0) { W.action = SA_SKIP; W.buf = 0; W.seq = 0; W.go = 0; W.helper_gave_up = 0; W.ac_nlev = 0; }
    __syncthreads();
    const SmplxSearchDev* const P = &Ph;         // read-only view; what changes lives in the search wave and goes back to Pd at the end
    const SmplxActionsDev& A = Ah;
    const SmplxGoalDev& G = Gh;
    const int nprims = A.nprims;
    const int ncfg = nprims * SMPLX_SMALL_LANES + 1;          // config lanes (the last one: the state itself)
    const int book0 = (ncfg + 63) / 64 * 64;                  // first thread of the search wave
    ModelLds Mv;
#ifdef SMPLX_CONST_MODEL
    constexpr bool RS = true;        // the waypoint lanes keep the saved link transforms in registers (kernels.hip const_chain<.., true>)
#else
    constexpr bool RS = false;
#endif
    ThreadLds L = setup_lds(S, smem, &Mv, book0, !RS);        // per-thread scratch for the config waves only (nobody else walks a chain through LDS)
    const ModelLds* M = &Mv;
    const SmplxGridDev grid = S->grid;
    const SmplxBfsDev bfs = Sq->bfs;
    const SmplxTableDev table = Sq->table;
    const int nv = MV_NVARS(M);
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
                           (unsigned int)((3 * nroot_lds + 12 * (RS ? 0 : Mv.nslots) + Mv.nvars) * 8 + hdr[SMPLX_BH_STACK]) * (unsigned int)book0;
        off = (off + 15u) & ~15u;
        H.lds = (LDS_AS hent_t*)((LDS_AS unsigned char*)smem + off);
        H.hbm = (SMPLX_GLOBAL_AS hent_t*)as_global(P->heap);
        H.st = as_global(P->st);
        H.lh = lh;
    }
    {
        const int n = P->heap_size + 1 < lh ? P->heap_size + 1 : lh;
        for (int i = t; i < n; i += blockDim.x) H.lds[i] = H.hbm[i];
    }
    __syncthreads();

    // The wave behind the search wave, where the block has one (smplx_search_block): between the two barriers of a round it
    // does the bookkeeping of the round's successors that does not touch OPEN -- coordinates, heuristic (planning-link FK + BFS
    // cell), goal test, the state table's side of getOrCreateState -- and leaves it in Bo for the search wave, whose own
    // sequential work per expansion (pop, commit, relax) is what bounds the kernel.
    const int help0 = book0 + 64;
    const bool has_helper = (int)blockDim.x > help0;
    if (t < book0 || t >= help0) {
        // =============================== the config waves (and the helper wave) ===============================
        while (true) {
            __syncthreads();                                       // A: the step is published
            const int action = W.action;
            if (action == SA_EXIT) break;
            if (action == SA_EVAL) {
                ExpandLds& Xr = Xb[W.buf];
                if (t < book0) expand_config_lane<RS>(M, L, A, G, grid, Xr, t, ncfg);
                else {
                    // lane p = primitive p, as in the search wave.  First what needs no table: coordinate, planning-link FK, BFS
                    // cell, goal test.  Then -- once the search wave says the table holds everything committed before this round's
                    // state was popped (W.go) -- hash, probe, the known successor's search state, duplicates among the new ones.
                    const int p = t - help0;
                    const bool act = p < nprims && Xr.lookups[p] != 0;   // (lookups[p]: the search wave's "primitive p has an action here")
                    BookLane r;
                    r.limits_ok = false; r.W = 0; r.h = 0; r.is_goal = 0;
                    if (act) {
                        expand_book_coords(M, Xr, p, r);
                        expand_book_goal(M, grid, bfs, G, Xr, p, r);
                    }
                    const int my_round = W.seq;
                    // (bounded: ~a second; the search wave sets go before it joins the round's end on every path, so the bound is
                    // never met -- it is there so that no wave of this kernel can wait for ever)
                    for (int spins = 0; __atomic_load_n(&W.go, __ATOMIC_RELAXED) != my_round; ++spins) {
                        __builtin_amdgcn_s_sleep(2);
                        if (spins > (1 << 24)) { W.helper_gave_up = 1; break; }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    BookRegs o;
                    SmplxSState ss;
                    ss.g = ss.h = ss.f = ss.eg = 0; ss.bp = -1; ss.heap_index = 0; ss.iteration_closed = 0; ss.call_number = 0; ss.flags = 0;
                    search_book_table<false>(M, grid, bfs, G, table, as_global(P->st), Xr, p, nv, act, r, o, ss);
                    Bo.limits_ok[p] = r.limits_ok ? 1 : 0; Bo.wcount[p] = r.W; Bo.h[p] = r.h; Bo.is_goal[p] = r.is_goal;
                    Bo.hash[p] = o.hash; Bo.free_slot[p] = o.pr.free_slot; Bo.pr_id[p] = o.pr.id; Bo.dup_of[p] = o.dup_of; Bo.clash[p] = o.clash ? 1 : 0;
                    Bo.ss[0][p] = (int)ss.g; Bo.ss[1][p] = (int)ss.h; Bo.ss[2][p] = (int)ss.f; Bo.ss[3][p] = (int)ss.eg;
                    Bo.ss[4][p] = ss.bp; Bo.ss[5][p] = ss.heap_index;
                    Bo.ss[6][p] = (int)((unsigned int)ss.iteration_closed | ((unsigned int)ss.call_number << 16)); Bo.ss[7][p] = (int)ss.flags;
                }
            }
            if (action == SA_REORDER) {
                const int size = W.reorder_size;
                const double eps = W.reorder_eps;
                for (int i = 1 + t; i <= size; i += blockDim.x) {     // f of every OPEN entry under the new epsilon (arastar.cpp:571-577)
                    const int eid = hent_id(hget(H, i));
                    SMPLX_GLOBAL_AS SmplxSState* ss = &as_global(P->st)[eid];
                    const unsigned int f = search_key(eps, ss->g, ss->h);
                    ss->f = f;
                    hput(H, i, hent_make(f, eid));
                }
                __syncthreads();
                heap_make_block(H, size, W.reorder_dups > 0);
            }
            __syncthreads();                                       // B: the waypoint verdicts have landed
        }
    } else {
        // =============================== the search wave ===============================
        const int lane = t - book0;
        const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
        SearchRegs R;
        R.curr_eps = P->curr_eps; R.satisfied_eps = P->satisfied_eps;
        R.heap_size = P->heap_size; R.nstates = P->nstates; R.n_incons = P->n_incons; R.n_log = P->n_log; R.n_succ = P->n_succ;
        R.iteration = P->iteration; R.call_number = P->call_number; R.phase = P->phase; R.num = P->num;
        R.expand_count = P->expand_count; R.expand_count_init = P->expand_count_init; R.err = P->err;
        R.dup_pushes = P->dup_pushes; R.goal_f = P->goal_f;
        R.committed_evals = P->committed_evals; R.gpu_evals = P->gpu_evals; R.lookups = P->lookups;
        R.status = SMPLX_SS_RUNNING; R.grow_what = 0; R.steps_left = max_steps;
        long long ticks[8];
        for (int k = 0; k < 8; ++k) ticks[k] = P->ticks[k];
        long long tick = (long long)wall_clock64();
#define SK_TICK(k) do { const long long now_ = (long long)wall_clock64(); ticks[k] += now_ - tick; tick = now_; } while (0)

        if (R.phase == 0) {
            // ---- ARAStar::replan, from scratch (arastar.cpp:107-167): empty OPEN and INCONS, new call number, the start
            // state with g = 0 into OPEN ----
            R.heap_size = 0; R.n_incons = 0; R.n_log = 0;
            R.call_number = (R.call_number + 1) & 0xFFFF;
            if (R.call_number == 0) R.call_number = 1;
            SmplxSState ss = sstate_load(&as_global(P->st)[P->start_id]), gs = sstate_load(&as_global(P->st)[0]);
            sstate_reinit(ss, R.call_number);
            sstate_reinit(gs, R.call_number);
            R.iteration = 1;
            R.curr_eps = P->initial_eps;
            R.satisfied_eps = __builtin_inf();
            ss.g = 0;
            ss.f = search_key(R.curr_eps, ss.g, ss.h);
            R.goal_f = P->start_id == 0 ? ss.f : gs.f;
            R.heap_size = 1;
            if (lane == 0) {
                sstate_store(&as_global(P->st)[P->start_id], ss, true);
                if (P->start_id != 0) sstate_store(&as_global(P->st)[0], gs, true);
                hset(H, 1, hent_make(ss.f, P->start_id));
            }
            R.num = 0; R.err = 0; R.expand_count = 0; R.expand_count_init = 0; R.dup_pushes = 0;
            R.phase = 1;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        }
        // the goal distance of a state follows from its heuristic when h = cost_per_cell * BFS distance is invertible
        // (bfs_heuristic.cpp:129-138 / 355-366: both read the BFS cell of the same planning-link position)
        const int cpc = bfs.cost_per_cell;
        const bool gd_from_h = cpc > 0 && (32767 % cpc) != 0;
        // The next expansion, started early.  An evaluation round (barrier A .. barrier B) keeps the config waves busy for
        // ~14 us while this wave only fills in the bookkeeping, and committing and relaxing a step (~7 us) plus the next pop
        // (~4 us) kept THEM idle.  GetSuccs is a pure function of a state's joint values, so as soon as the verdicts of a
        // step are in, this wave guesses the state the next pop will return -- the best new successor, if it beats the top
        // of OPEN -- and opens the round for it in the other ExpandLds; the relaxation and the pop run beside it.  A right
        // guess is adopted (the round is already under way), a wrong one is waited out and dropped: results never depend
        // on it.  pend: the state whose round is open and not joined yet (-1: none), pbuf its buffer, p_act its lanes.
        int pend = -1, pbuf = 0;
        int round_seq = 0;               // number of the last round opened (W.seq)
        bool p_act = false;
        long long spec_issued = 0, spec_hits = 0;

        while (true) {
            // =========================== what happens next (ARAStar::replan / improvePath) ===========================
            int action = -1, m = 0;
            while (action < 0) {
                if (R.phase == 1) {
                    // arastar.cpp:169-186
                    if (!(R.satisfied_eps > P->final_eps)) { R.phase = 3; action = SA_EXIT; break; }
                    if (R.curr_eps == R.satisfied_eps) {
                        if (!P->improve) { R.phase = 3; action = SA_EXIT; break; }
                        if (R.heap_size + R.n_incons > P->cap_heap) { R.status = SMPLX_SS_GROW; R.grow_what = 1; action = SA_EXIT; break; }
                        action = SA_REORDER;
                        break;
                    }
                    R.phase = 2;
                }
                // ---- one step of improvePath (arastar.cpp:486-527) ----
                int err = -1;
                hent_t top = 0;
                if (R.heap_size == 0) err = 5;                                          // EXHAUSTED_OPEN_LIST
                else {
                    top = hget(H, 1);
                    bool timed_out = false;
                    if (P->bounded) timed_out = R.satisfied_eps == __builtin_inf() ? R.num >= P->max_init : R.num >= P->max_rep;
                    if (hent_f(top) >= R.goal_f || hent_id(top) == 0) err = 0;          // SUCCESS
                    else if (timed_out) err = 4;                                        // TIMED_OUT
                }
                if (err >= 0) {
                    // back in replan (arastar.cpp:188-197)
                    if (R.curr_eps == P->initial_eps) R.expand_count_init += R.num;
                    R.phase = 1;
                    R.err = err;
                    if (err != 0) { R.phase = 3; action = SA_EXIT; break; }
                    R.satisfied_eps = R.curr_eps;
                    continue;
                }
                if (R.steps_left <= 0) { action = SA_EXIT; break; }                      // this launch has done its share
                // room for one more expansion?  (checked before anything is popped: the host enlarges and launches again)
                if (R.nstates + nprims > P->cap_states || R.heap_size + nprims > P->cap_heap || R.n_incons + nprims > P->cap_incons ||
                    R.n_log + 1 > P->cap_log || R.n_succ + nprims > P->cap_succ || (unsigned int)(2 * (R.nstates + nprims)) > table.mask + 1u) {
                    R.status = SMPLX_SS_GROW; R.grow_what = 2;
                    action = SA_EXIT;
                    break;
                }
                --R.steps_left;
                m = hent_id(top);
                action = SA_EVAL;
            }

            if (action != SA_EVAL && pend >= 0) {                                    // B of a round nobody will use
                if (lane == 0) __atomic_store_n(&W.go, round_seq, __ATOMIC_RELAXED);
                __syncthreads();
                pend = -1;
            }
            if (action == SA_EXIT) {
                if (lane == 0) W.action = SA_EXIT;
                __syncthreads();                                   // A
                break;
            }

            if (action == SA_REORDER) {
                // =========================== a new epsilon (arastar.cpp:174-186, 571-577) ===========================
                ++R.iteration;
                R.curr_eps -= P->delta_eps;
                R.curr_eps = R.curr_eps > P->final_eps ? R.curr_eps : P->final_eps;
                for (int i = 0; i < R.n_incons; ++i) {
                    const int sid = as_global(P->incons)[i];
                    const SmplxSState ss = sstate_load(&as_global(P->st)[sid]);
                    if (ss.heap_index != 0) {                       // already in OPEN: the same element twice from now on
                        if (lane == 0) as_global(P->st)[sid].flags = ss.flags | 1u;
                        ++R.dup_pushes;
                    }
                    ++R.heap_size;
                    heap_sift_up_wave(H, W, lane, R.heap_size, hent_make(ss.f, sid), false);
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   // the next push reads what this one stored
                }
                R.n_incons = 0;
                if (lane == 0) { W.action = SA_REORDER; W.reorder_size = R.heap_size; W.reorder_eps = R.curr_eps; W.reorder_dups = R.dup_pushes; }
                __syncthreads();                                   // A
                {
                    const int size = R.heap_size;
                    for (int i = 1 + t; i <= size; i += blockDim.x) {
                        const int eid = hent_id(hget(H, i));
                        SMPLX_GLOBAL_AS SmplxSState* ss = &as_global(P->st)[eid];
                        const unsigned int f = search_key(R.curr_eps, ss->g, ss->h);
                        ss->f = f;
                        hput(H, i, hent_make(f, eid));
                    }
                    __syncthreads();
                    heap_make_block(H, size, R.dup_pushes > 0);
                }
                __syncthreads();                                   // B
                R.goal_f = as_global(P->st)[0].f;      // (the goal state is re-initialised when a call starts: its f is this call's)
                R.phase = 2;
                SK_TICK(5);
                continue;
            }

            // =========================== expand state m (arastar.cpp:513-519, 531-568) ===========================
            // ---- pop (intrusive_heap.hpp:155-166).  The state popped is very often one the previous relaxation has just
            // stored (another lane of this wave did): those stores have landed before it is read ----
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            // ... and so has everything the previous step committed: the helper wave of an open round may read the table now
            if (pend >= 0 && lane == 0) __atomic_store_n(&W.go, round_seq, __ATOMIC_RELAXED);
            const SmplxSState sm = sstate_load(&as_global(P->st)[m]);           // g, h: in flight together with the loads below
            const hent_t last = hget(H, R.heap_size);
            const int off = as_global(P->done_off)[m];                          // >= 0: expanded before (a later ARA* iteration)
            const int dcnt = as_global(P->done_cnt)[m];
            const bool use_spec = pend == m && off < 0;                         // (only states never expanded are guessed)
            const int cur = use_spec ? pbuf : (pend >= 0 ? pbuf ^ 1 : 0);
            ExpandLds& X = Xb[cur];
            if (!use_spec) {
                if (lane < nv) X.parent[lane] = as_global(P->q)[(size_t)m * nv + lane];
                if (lane < nprims) { X.edge_bad[lane] = 0; X.edge_lk[lane] = 0; }
                if (lane == 0) { X.state_bad = 0; X.state_lookups = 0; }
            }
            if (lane == 0) as_global(P->st)[m].heap_index = 0;
            --R.heap_size;
            if (R.heap_size >= 1) heap_sift_down_wave(H, lane, 1, R.heap_size, last);
            // the new top of OPEN is what the NEXT pop returns unless this step puts something above it: its joint values and
            // heuristic are requested now and looked at when the guess is made (below), a whole evaluation later
            int top_id = -1, top_off = 0;
            unsigned int top_f = 0xFFFFFFFFu, top_h = 0;
            double top_q = 0.0;
            if (R.heap_size >= 1) {
                const hent_t tp = hget(H, 1);
                top_id = hent_id(tp);
                top_f = hent_f(tp);
                top_off = as_global(P->done_off)[top_id];
                top_h = as_global(P->st)[top_id].h;
                if (lane < nv) top_q = as_global(P->q)[(size_t)top_id * nv + lane];
            }
            const unsigned int eg = sm.g;
            if (lane == 0) {
                *(SMPLX_GLOBAL_AS unsigned int*)&as_global(P->st)[m].iteration_closed = ((unsigned int)R.iteration & 0xFFFFu) | ((unsigned int)sm.call_number << 16);
                as_global(P->st)[m].eg = eg;
                as_global(P->log)[R.n_log] = m;
            }
            ++R.n_log;
            SK_TICK(1);

            // per-lane data of this step: lane p < nprims = primitive p (evaluation), lane k < cnt = list entry k (cached)
            bool valid = false;
            bool ac_complete = true;     // the ancestor cache holds every HBM-resident ancestor the pushes will read
            int sid = -1, cost = 0, cnt = 0, evals = 0;
            SmplxSState ss;                                          // the successor's search state
            ss.g = ss.h = ss.f = ss.eg = 0; ss.bp = -1; ss.heap_index = 0; ss.iteration_closed = 0; ss.call_number = 0; ss.flags = 0;

            if (off < 0) {
                // ---- GetSuccs loop body (manip_lattice.cpp:254-305): successors' joint values, then the waypoint lanes ----
                const bool in = lane < nprims;
                bool act;
                if (use_spec) {
                    // the round for this state is open already: its successor values are in X, its lanes at work
                    act = p_act;
                    pend = -1;
                    ++spec_hits;
                } else {
                    SMPLX_WAVE_SYNC();                               // X.parent is complete
                    double gd;
                    if (gd_from_h) {
                        const int hh = (int)sm.h;
                        gd = hh == 32767 ? (double)0x7FFFFFFF * grid.res : (double)(hh / cpc) * grid.res;
                    } else {
                        double g1 = 0.0;
                        if (lane == 0) g1 = expand_goal_distance(M, grid, bfs, X);
                        const unsigned long long bits = wave_rl64((unsigned long long)__double_as_longlong(g1), 0);
                        gd = __longlong_as_double((long long)bits);
                    }
                    act = in && prim_has_action(A, G, lane) && mprim_active(A, gd, A.type[lane]);
                    if (act) expand_successor_values(M, A, G, X, lane);
                    if (pend >= 0) { __syncthreads(); pend = -1; }   // B of the round that guessed wrong
                    if (in) X.lookups[lane] = act ? 1 : 0;
                    ++round_seq;
                    if (lane == 0) { X.goal_dist = gd; W.action = SA_EVAL; W.buf = cur; W.seq = round_seq; W.go = round_seq; }
                    __syncthreads();                                 // A
                }
                BookLane b;
                b.limits_ok = false; b.W = 0; b.h = 0; b.is_goal = 0;
                BookRegs bk;
                bk.hash = 0; bk.pr.id = -1; bk.pr.free_slot = 0; bk.dup_of = -1; bk.clash = false;
                if (!has_helper) {
                    // While the waypoint lanes work: everything of getOrCreateState that does not need their verdict
                    if (act) expand_book_coords(M, X, lane, b);
                    search_book_table<true>(M, grid, bfs, G, table, as_global(P->st), X, lane, nv, act, b, bk, ss);
                }
                const SmplxSState goal_ss = sstate_load(&as_global(P->st)[0]);   // (a goal successor relaxes the goal state)
                ac_complete = ancestor_cache_fill(H, W, lane, R.heap_size, __popcll(__ballot(act)));       // (an upper bound of the pushes to come)
                // ---- while the waypoint lanes finish: the round of the most likely next state, the top of OPEN, is got ready in
                // the other buffer (its successors' joint values), so that it can be opened the moment this round closes ----
                const int nxt = cur ^ 1;
                bool top_ready = false, top_act = false;
                if (top_id > 0 && top_off < 0 && top_f < R.goal_f && R.steps_left > 0) {
                    ExpandLds& Y = Xb[nxt];
                    if (lane < nv) Y.parent[lane] = top_q;
                    if (lane < nprims) { Y.edge_bad[lane] = 0; Y.edge_lk[lane] = 0; }
                    if (lane == 0) { Y.state_bad = 0; Y.state_lookups = 0; }
                    SMPLX_WAVE_SYNC();                               // Y.parent is complete
                    double gd2;
                    if (gd_from_h) {
                        const int hh = (int)top_h;
                        gd2 = hh == 32767 ? (double)0x7FFFFFFF * grid.res : (double)(hh / cpc) * grid.res;
                    } else {
                        double g1 = 0.0;
                        if (lane == 0) g1 = expand_goal_distance(M, grid, bfs, Y);
                        const unsigned long long bits = wave_rl64((unsigned long long)__double_as_longlong(g1), 0);
                        gd2 = __longlong_as_double((long long)bits);
                    }
                    top_act = in && prim_has_action(A, G, lane) && mprim_active(A, gd2, A.type[lane]);
                    if (top_act) expand_successor_values(M, A, G, Y, lane);
                    if (in) Y.lookups[lane] = top_act ? 1 : 0;
                    if (lane == 0) Y.goal_dist = gd2;
                    top_ready = true;
                }
                __syncthreads();                                     // B: the waypoint verdicts have landed
                SK_TICK(2);
                if (has_helper && W.helper_gave_up) R.steps_left = 0;      // leave at the next selection; the status says why (below)
                if (has_helper && in) {
                    b.limits_ok = Bo.limits_ok[lane] != 0; b.W = Bo.wcount[lane]; b.h = Bo.h[lane]; b.is_goal = Bo.is_goal[lane];
                    bk.hash = Bo.hash[lane]; bk.pr.free_slot = Bo.free_slot[lane]; bk.pr.id = Bo.pr_id[lane];
                    bk.dup_of = Bo.dup_of[lane]; bk.clash = Bo.clash[lane] != 0;
                    ss.g = (unsigned int)Bo.ss[0][lane]; ss.h = (unsigned int)Bo.ss[1][lane]; ss.f = (unsigned int)Bo.ss[2][lane];
                    ss.eg = (unsigned int)Bo.ss[3][lane]; ss.bp = Bo.ss[4][lane]; ss.heap_index = Bo.ss[5][lane];
                    const unsigned int itc = (unsigned int)Bo.ss[6][lane];
                    ss.iteration_closed = (unsigned short)(itc & 0xFFFFu); ss.call_number = (unsigned short)(itc >> 16);
                    ss.flags = (unsigned int)Bo.ss[7][lane];
                }
                const bool cand = act && b.limits_ok;
                unsigned int hash = bk.hash;
                TableProbe pr = bk.pr;
                int dup_of = bk.dup_of;
                bool clash = bk.clash;
                {
                    // No edge that could create a state has a key below the top's (whatever the verdicts say): the next pop
                    // returns that top, and its round opens at once.  Otherwise the guess waits for the verdicts (below).
                    const bool maybe_new = cand && pr.id < 0 && !b.is_goal;
                    const unsigned int fj = maybe_new ? search_key(R.curr_eps, eg + (unsigned int)A.cost[lane], (unsigned int)b.h) : 0xFFFFFFFFu;
                    if (top_ready && __ballot(maybe_new && fj < top_f) == 0ull) {
                        ++round_seq;
                        if (lane == 0) { W.action = SA_EVAL; W.buf = nxt; W.seq = round_seq; }
                        __syncthreads();                             // A of the guessed round
                        pend = top_id;
                        pbuf = nxt;
                        p_act = top_act;
                        ++spec_issued;
                    }
                }
                int lookups = 0;
                const int flags = in ? expand_verdict(A, X, lane, act, b, lookups) : SMPLX_F_INACTIVE;
                valid = (flags & SMPLX_F_VALID) != 0;
                const bool goal_succ = valid && (flags & SMPLX_F_GOAL) != 0;
                int id = pr.id;
                // ---- getOrCreateState (manip_lattice.cpp:1318-1354) ----
                const bool unknown = valid && id < 0;
                if (!unknown) dup_of = -1;
                {
                    // a candidate that was to create the state for a later one turned out to collide (rare): the pairs again,
                    // among the valid successors only
                    const bool root_valid = __shfl((int)valid, dup_of >= 0 ? dup_of : lane) != 0;
                    if (__ballot(unknown && dup_of >= 0 && !root_valid)) {
                        dup_of = -1;
                        unsigned long long members = __ballot(unknown);
                        while (members) {                            // uniform
                            const int j = __ffsll((long long)members) - 1;
                            members &= members - 1;
                            const unsigned int hj = wave_rlu(hash, j);
                            if (j < lane && unknown && dup_of < 0 && hj == hash) {
                                bool same = true;
                                for (int v = 0; v < nv; ++v) same = same && X.coord[j][v] == X.coord[lane][v];
                                if (same) dup_of = j;
                            }
                        }
                        const bool mine = unknown && dup_of < 0;
                        clash = false;
                        members = __ballot(mine);
                        while (members) {
                            const int j = __ffsll((long long)members) - 1;
                            members &= members - 1;
                            const unsigned int fj = wave_rlu(pr.free_slot, j);
                            if (j < lane && mine && fj == pr.free_slot) clash = true;
                        }
                    }
                }
                const bool is_new = unknown && dup_of < 0;
                if (!is_new) clash = false;
                const unsigned long long m_new = __ballot(is_new), m_valid = __ballot(valid), m_eval = __ballot(in && !(flags & SMPLX_F_INACTIVE));
                if (is_new) {
                    id = R.nstates + __popcll(m_new & below);
                    if (!clash) table_store_own(table, pr.free_slot, (const LDS_AS int*)X.coord[lane], nv, id);
                    for (int v = 0; v < nv; ++v) { as_global(P->coord)[(size_t)id * nv + v] = X.coord[lane][v]; as_global(P->q)[(size_t)id * nv + v] = X.sq[lane][v]; }
                    ss.h = (unsigned int)b.h;
                    sstate_reinit(ss, R.call_number);
                    sstate_store(&as_global(P->st)[id], ss, true);              // (what the relaxation changes is stored again behind it)
                    as_global(P->done_off)[id] = -1;
                }
                {
                    unsigned long long m_clash = __ballot(clash);
                    while (m_clash) {      // uniform loop: one clashing lane at a time probes behind what the others have stored
                        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                        const int j = __ffsll((long long)m_clash) - 1;
                        m_clash &= m_clash - 1;
                        if (lane == j) {
                            TableProbeLoads ld = table_probe_start(table, nv, hash);
                            const TableProbe p2 = table_probe_finish(table, ld, (const LDS_AS int*)X.coord[lane], nv);
                            table_store_own(table, p2.free_slot, (const LDS_AS int*)X.coord[lane], nv, id);
                        }
                    }
                }
                {
                    const int id_of_dup = __shfl(id, dup_of >= 0 ? dup_of : lane);
                    if (dup_of >= 0) id = id_of_dup;
                }
                cnt = __popcll(m_valid);
                evals = __popcll(m_eval);
                if (valid) {
                    sid = goal_succ ? 0 : id;      // a goal successor is reported as the goal id (manip_lattice.cpp:283-296)
                    cost = A.cost[lane];
                    if (goal_succ) ss = goal_ss;
                    // (id, cost | primitive << 24) as one 8-byte store
                    ((SMPLX_GLOBAL_AS unsigned long long*)as_global(P->succ))[R.n_succ + __popcll(m_valid & below)] =
                        (unsigned long long)(unsigned int)sid | ((unsigned long long)(unsigned int)(cost | (lane << 24)) << 32);
                }
                if (lane == 0) { as_global(P->done_off)[m] = R.n_succ; as_global(P->done_cnt)[m] = cnt | (evals << 8); }
                if (pend < 0) {
                    // ---- the guess (see above): the new successor with the least f if it beats the top of OPEN (a push does
                    // not pass an equal key: intrusive_heap.hpp:346-365), else that top ----
                    const bool cand = valid && is_new && !goal_succ;
                    const unsigned int fj = cand ? search_key(R.curr_eps, eg + (unsigned int)cost, ss.h) : 0xFFFFFFFFu;
                    unsigned long long key = ((unsigned long long)fj << 8) | (unsigned long long)lane;      // ties: the lower primitive
                    for (int o = 32; o > 0; o >>= 1) {
                        const unsigned long long other = __shfl_xor(key, o);
                        key = other < key ? other : key;
                    }
                    const unsigned int fbest = (unsigned int)(key >> 8);
                    const int jbest = (int)(key & 0xFFull);
                    const bool guess_succ = fbest != 0xFFFFFFFFu && fbest < top_f && fbest < R.goal_f;
                    const bool guess_top = !guess_succ && top_id > 0 && top_off < 0 && top_f < R.goal_f;
                    if ((guess_succ || guess_top) && R.steps_left > 0) {
                        ExpandLds& Y = Xb[nxt];
                        if (lane < nv) Y.parent[lane] = guess_succ ? X.sq[jbest][lane] : top_q;
                        if (lane < nprims) { Y.edge_bad[lane] = 0; Y.edge_lk[lane] = 0; }
                        if (lane == 0) { Y.state_bad = 0; Y.state_lookups = 0; }
                        SMPLX_WAVE_SYNC();                           // Y.parent is complete
                        double gd2;
                        if (gd_from_h) {
                            const int hh = guess_succ ? wave_rl((int)ss.h, jbest) : (int)top_h;
                            gd2 = hh == 32767 ? (double)0x7FFFFFFF * grid.res : (double)(hh / cpc) * grid.res;
                        } else {
                            double g1 = 0.0;
                            if (lane == 0) g1 = expand_goal_distance(M, grid, bfs, Y);
                            const unsigned long long bits = wave_rl64((unsigned long long)__double_as_longlong(g1), 0);
                            gd2 = __longlong_as_double((long long)bits);
                        }
                        p_act = in && prim_has_action(A, G, lane) && mprim_active(A, gd2, A.type[lane]);
                        if (p_act) expand_successor_values(M, A, G, Y, lane);
                        if (in) Y.lookups[lane] = p_act ? 1 : 0;
                        ++round_seq;
                        if (lane == 0) { Y.goal_dist = gd2; W.action = SA_EVAL; W.buf = nxt; W.seq = round_seq; }
                        __syncthreads();                             // A of the guessed round
                        pend = guess_succ ? wave_rl(sid, jbest) : top_id;
                        pbuf = nxt;
                        ++spec_issued;
                    }
                }
                R.n_succ += cnt;
                R.nstates += __popcll(m_new);
                R.gpu_evals += evals;
                {
                    int lk = in ? lookups : 0;       // grid lookups of this expansion, as the reference would count them
                    for (int o = 32; o > 0; o >>= 1) lk += __shfl_down(lk, o);
                    R.lookups += wave_rl(lk, 0);
                }
            } else {
                // ---- GetSuccs of a state expanded before: the committed list (no evaluation round) ----
                if (pend >= 0) { __syncthreads(); pend = -1; }       // B of the round that guessed wrong
                cnt = dcnt & 0xFF;
                evals = dcnt >> 8;
                valid = lane < cnt;
                if (valid) {
                    const unsigned long long sc = ((const SMPLX_GLOBAL_AS unsigned long long*)as_global(P->succ))[off + lane];
                    const int sc_cost_prim = (int)(unsigned int)(sc >> 32);
                    sid = (int)(unsigned int)sc;
                    cost = sc_cost_prim & 0xFFFFFF;
                    ss = sstate_load(&as_global(P->st)[sid]);
                }
                ac_complete = ancestor_cache_fill(H, W, lane, R.heap_size, cnt);
                SK_TICK(2);
            }

            // ---- which successors name the same state (two goal successors; a primitive that lands in an earlier one's
            // cell): the earliest lane keeps the state, the others refer to it ----
            const unsigned long long m_succ = __ballot(valid);
            int alias = -1;
            {
                unsigned long long rest = m_succ;
                while (rest) {
                    const int j = __ffsll((long long)rest) - 1;
                    rest &= rest - 1;
                    const int sj = wave_rl(sid, j);
                    if (valid && j < lane && alias < 0 && sj == sid) alias = j;
                }
            }
            // what earlier lanes of this wave stored (heap entries moved by the pop, the popped state's closing, new states'
            // rows) and others read below (a self-loop successor, the ancestor cache) has landed
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            SK_TICK(3);

            // ---- ARAStar::expand's loop over the successors, in primitive order (arastar.cpp:540-567): uniform control
            // flow, the state of successor j in the registers of lane j ----
            R.committed_evals += evals;
            bool dirty = false;
            if (__ballot(valid && alias >= 0) == 0ull) {
                // No two successors name the same state (nearly always): what expand's loop decides for a successor then depends
                // on that successor alone, so every lane works it out for its own -- reinitSearchState, the cost test, the key --
                // and the loop below only performs the OPEN / INCONS operations of the successors that improved, in
                // primitive order.  (The general loop further down costs ~0.37 us per successor, improved or not.)
                const bool fresh = valid && ss.call_number != (unsigned short)R.call_number;
                if (fresh) {
                    // reinitSearchState: not touched in this call (and so not in OPEN)
                    sstate_reinit(ss, R.call_number);
                    dirty = true;
                    as_global(P->st)[sid].heap_index = 0;
                }
                const unsigned int new_cost = eg + (unsigned int)cost;
                const bool improve = valid && new_cost < ss.g;
                const bool reached_before = ss.g != SMPLX_INFINITECOST;
                const bool closed_now = (unsigned int)ss.iteration_closed == ((unsigned int)R.iteration & 0xFFFFu);
                unsigned int f = 0;
                if (improve) {
                    ss.g = new_cost; ss.bp = m; dirty = true;
                    if (!closed_now) { f = search_key(R.curr_eps, new_cost, ss.h); ss.f = f; }
                }
                unsigned long long rest = __ballot(improve);
                while (rest) {
                    const int j = __ffsll((long long)rest) - 1;
                    rest &= rest - 1;
                    const int id = wave_rl(sid, j);
                    if (wave_rl((int)closed_now, j) == 0) {
                        const unsigned int fj = wave_rlu(f, j);
                        const unsigned int flags_j = wave_rlu(ss.flags, j);
                        const bool reached_j = wave_rl((int)reached_before, j) != 0;
                        if (id == 0) R.goal_f = fj;
                        int hi = 0;
                        if (reached_j || !ac_complete) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   // earlier sifts' stores have landed
                        if (reached_j) hi = as_global(P->st)[id].heap_index;
                        const hent_t e = hent_make(fj, id);
                        if (hi != 0) {
                            if (flags_j & 1u) heap_refresh_duplicates_wave(H, W, lane, R.heap_size, id, fj);
                            heap_sift_up_wave(H, W, lane, hi, e, false);
                        } else {
                            if ((flags_j & 1u) && R.dup_pushes > 0) heap_refresh_duplicates_wave(H, W, lane, R.heap_size, id, fj);
                            ++R.heap_size;
                            heap_sift_up_wave(H, W, lane, R.heap_size, e, true);
                        }
                    } else {
                        if (lane == 0) as_global(P->incons)[R.n_incons] = id;      // (never marked: arastar.cpp:563-565)
                        ++R.n_incons;
                    }
                }
            } else {
                unsigned long long rest = m_succ;
                while (rest) {
                    const int j = __ffsll((long long)rest) - 1;
                    rest &= rest - 1;
                    const int aj = wave_rl(alias, j);
                    const int r = aj >= 0 ? aj : j;                  // the lane that keeps this successor's state
                    const int id = wave_rl(sid, j);
                    const int cj = wave_rl(cost, j);
                    unsigned int g_r = wave_rlu(ss.g, r);
                    const unsigned int h_r = wave_rlu(ss.h, r);
                    unsigned int itc_r = (unsigned int)wave_rl((int)ss.iteration_closed | ((int)ss.call_number << 16), r);
                    unsigned int flags_r = wave_rlu(ss.flags, r);
                    if ((itc_r >> 16) != (unsigned int)R.call_number) {
                        // reinitSearchState: not touched in this call (and so not in OPEN)
                        if (lane == r) { sstate_reinit(ss, R.call_number); dirty = true; }
                        if (lane == 0) as_global(P->st)[id].heap_index = 0;
                        g_r = SMPLX_INFINITECOST; itc_r = (unsigned int)R.call_number << 16; flags_r = 0;
                    }
                    const int new_cost = (int)(eg + (unsigned int)cj);
                    if (!((unsigned int)new_cost < g_r)) continue;
                    const bool reached_before = g_r != SMPLX_INFINITECOST;
                    if (lane == r) { ss.g = (unsigned int)new_cost; ss.bp = m; dirty = true; }
                    if ((itc_r & 0xFFFFu) != ((unsigned int)R.iteration & 0xFFFFu)) {
                        const unsigned int f = search_key(R.curr_eps, (unsigned int)new_cost, h_r);
                        if (lane == r) ss.f = f;
                        if (id == 0) R.goal_f = f;
                        // a state reached for the first time in this call is not in OPEN; otherwise its position is read where
                        // the sifts keep it current
                        int hi = 0;
                        if (reached_before || !ac_complete) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   // earlier sifts' stores have landed
                        if (reached_before) hi = as_global(P->st)[id].heap_index;
                        const hent_t e = hent_make(f, id);
                        if (hi != 0) {
                            if (flags_r & 1u) heap_refresh_duplicates_wave(H, W, lane, R.heap_size, id, f);
                            heap_sift_up_wave(H, W, lane, hi, e, false);
                        } else {
                            if ((flags_r & 1u) && R.dup_pushes > 0) heap_refresh_duplicates_wave(H, W, lane, R.heap_size, id, f);
                            ++R.heap_size;
                            heap_sift_up_wave(H, W, lane, R.heap_size, e, true);
                        }
                    } else {
                        if (lane == 0) as_global(P->incons)[R.n_incons] = id;      // (never marked: arastar.cpp:563-565)
                        ++R.n_incons;
                    }
                }
            }
            if (dirty && valid && alias < 0) sstate_store(&as_global(P->st)[sid], ss, false);
            if (lane == 0) W.ac_nlev = 0;
            ++R.num;
            SK_TICK(4);
        }

        // ---- the launch ends: results, or the state the next launch picks up ----
        if (lane == 0) {
            int solved = Ph.solved, cost = Ph.cost, n_path = Ph.n_path;
            if (R.phase == 3) {
                // arastar.cpp:199-214
                R.expand_count += R.num;
                R.status = SMPLX_SS_DONE;
                if (R.satisfied_eps == __builtin_inf()) {
                    solved = 0; cost = 0; n_path = 0;
                } else {
                    int n = 0;
                    for (int sid = 0; sid >= 0 && n < P->cap_path; sid = as_global(P->st)[sid].bp) as_global(P->path)[n++] = sid;
                    n_path = n;
                    cost = (int)as_global(P->st)[0].g;
                    solved = 1;
                }
                R.phase = 4;
            }
            if (W.helper_gave_up) R.status = SMPLX_SS_ERROR;
            Pd->solved = solved; Pd->cost = cost; Pd->n_path = n_path;
            Pd->curr_eps = R.curr_eps; Pd->satisfied_eps = R.satisfied_eps;
            Pd->heap_size = R.heap_size; Pd->nstates = R.nstates; Pd->n_incons = R.n_incons; Pd->n_log = R.n_log; Pd->n_succ = R.n_succ;
            Pd->iteration = R.iteration; Pd->call_number = R.call_number; Pd->phase = R.phase; Pd->num = R.num;
            Pd->expand_count = R.expand_count; Pd->expand_count_init = R.expand_count_init; Pd->err = R.err;
            Pd->dup_pushes = R.dup_pushes; Pd->goal_f = R.goal_f;
            Pd->status = R.status; Pd->grow_what = R.grow_what;
            Pd->committed_evals = R.committed_evals; Pd->gpu_evals = R.gpu_evals; Pd->lookups = R.lookups;
            SK_TICK(6);
            ticks[7] += (spec_issued << 32) + spec_hits;            // rounds opened on a guess | guesses the next pop confirmed
            for (int k = 0; k < 8; ++k) Pd->ticks[k] = ticks[k];
            if (status_out) status_out[blockIdx.x] = R.status;
            W.reorder_size = R.heap_size;                           // (for the write-back of the LDS part below)
        }
#undef SK_TICK
    }
    __syncthreads();
    {
        const int n = W.reorder_size + 1 < lh ? W.reorder_size + 1 : lh;
        for (int i = t; i < n; i += blockDim.x) H.hbm[i] = H.lds[i];
    }
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
            SMPLX_GLOBAL_AS int* sl = as_global(T.slots) + (size_t)k * T.stride;
            if (atomicCAS((int*)&sl[0], 0, -(id + 1)) == 0) {
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
// (tests/golden/heap_ref.json).  Element e is state e; the first `lh` heap entries live in LDS, the rest in HBM.  One wave:
// push / decrease run the wave-parallel sift-up, pop / increase / erase the wave-parallel sift-down, make() the level-parallel
// form -- the code paths of the search.
extern "C" __global__ void __launch_bounds__(64)
k_heap_ops(const int* __restrict__ ops, int nops, int lh, unsigned long long* __restrict__ heap_hbm, SmplxSState* __restrict__ st,
           int* __restrict__ top_after)
{
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ SearchLds W;
    HeapRef H;
    H.lds = (LDS_AS hent_t*)smem;
    H.hbm = as_global(heap_hbm);
    H.st = as_global(st);
    H.lh = lh;
    const int lane = threadIdx.x;
    if (lane == 0) W.ac_nlev = 0;
    __syncthreads();
    int nelem = 0, size = 0;
    for (int i = 0; i < nops; ++i) {
        const int code = ops[2 * i], key = ops[2 * i + 1];
        if (code == 0) {
            if (lane == 0) { st[nelem].f = (unsigned int)key; st[nelem].heap_index = 0; }
            ++size;
            heap_sift_up_wave(H, W, lane, size, hent_make((unsigned int)key, nelem), false);
            ++nelem;
        } else if (code == 1) {
            if (size > 0) {
                const hent_t top = hget(H, 1), last = hget(H, size);
                if (lane == 0) st[hent_id(top)].heap_index = 0;
                --size;
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                if (size >= 1) heap_sift_down_wave(H, lane, 1, size, last);
            }
        } else if (code == 2 || code == 5) {
            const int e = key >> 20, p = key & 0xFFFFF;
            const int hi = e < nelem ? st[e].heap_index : 0;
            if (hi != 0) {
                if (lane == 0) st[e].f = (unsigned int)p;
                if (code == 2) heap_sift_up_wave(H, W, lane, hi, hent_make((unsigned int)p, e), false);
                else heap_sift_down_wave(H, lane, hi, size, hent_make((unsigned int)p, e));
            }
        } else if (code == 3) {
            const int pos = key < nelem ? st[key].heap_index : 0;
            if (pos != 0) {                                      // intrusive_heap.hpp:197-206
                const hent_t last = hget(H, size);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                if (lane == 0) { hset(H, pos, last); st[key].heap_index = 0; }
                --size;
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                if (pos <= size) heap_sift_down_wave(H, lane, pos, size, last);
            }
        } else if (code == 4) {
            for (int k = 1 + lane; k <= size; k += 64) {
                const int e = hent_id(hget(H, k));
                const unsigned int p = (st[e].f * 7919u + 13u) % 1000u;
                st[e].f = p;
                hput(H, k, hent_make(p, e));
            }
            __syncthreads();
            heap_make_block(H, size, false);
        }
        __syncthreads();      // (one wave: orders what the lanes stored with what the next op reads)
        if (lane == 0) top_after[i] = size > 0 ? hent_id(hget(H, 1)) : -1;
    }
}
