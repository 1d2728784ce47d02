// smpl_amd/csrc/search_host.h -- host side of the device-resident ARA* (kernel: search_kernel.h).
// Included by engine.hip inside its anonymous namespace, after smplx_space is complete.
//
// The host launches k_search (one workgroup per query), enlarges a query's buffers when its workgroup asks
// (SMPLX_SS_GROW), and reads results.  The lattice the device builds -- coordinates, joint values, heuristics, committed
// successor lists, expansion log -- stays in HBM until somebody asks for it through the C-ABI (smplx_get_state,
// smplx_expansion_log, smplx_get_succs ...): pull_lattice() then brings the host's arrays up to date, so that every entry
// point sees one lattice whichever side created a state.

// one allocation per query, carved into the buffers of SmplxSearchDev
typedef DevSearch::Caps SearchCaps;

inline size_t search_arena_bytes(const SearchCaps& c, int N)
{
    size_t b = 0;
    b += align256((size_t)c.states * N * sizeof(int32_t));
    b += align256((size_t)c.states * N * sizeof(double));
    b += align256((size_t)c.states * sizeof(SmplxSState));
    b += align256(((size_t)c.heap + 1) * sizeof(SmplxHeapEntry));
    b += align256((size_t)c.incons * sizeof(int32_t));
    b += align256((size_t)c.log * sizeof(int32_t));
    b += align256((size_t)c.states * sizeof(int32_t)) * 2;
    b += align256((size_t)c.succ * sizeof(SmplxSucc));
    b += align256((size_t)c.path * sizeof(int32_t));
    return b;
}

inline void search_carve(unsigned char* base, const SearchCaps& c, int N, SmplxSearchDev& h)
{
    unsigned char* w = base;
    h.coord = (int32_t*)w; w += align256((size_t)c.states * N * sizeof(int32_t));
    h.q = (double*)w; w += align256((size_t)c.states * N * sizeof(double));
    h.st = (SmplxSState*)w; w += align256((size_t)c.states * sizeof(SmplxSState));
    h.heap = (SmplxHeapEntry*)w; w += align256(((size_t)c.heap + 1) * sizeof(SmplxHeapEntry));
    h.incons = (int32_t*)w; w += align256((size_t)c.incons * sizeof(int32_t));
    h.log = (int32_t*)w; w += align256((size_t)c.log * sizeof(int32_t));
    h.done_off = (int32_t*)w; w += align256((size_t)c.states * sizeof(int32_t));
    h.done_cnt = (int32_t*)w; w += align256((size_t)c.states * sizeof(int32_t));
    h.succ = (SmplxSucc*)w; w += align256((size_t)c.succ * sizeof(SmplxSucc));
    h.path = (int32_t*)w;
    h.cap_states = c.states; h.cap_heap = c.heap; h.cap_incons = c.incons; h.cap_log = c.log; h.cap_succ = c.succ; h.cap_path = c.path;
}

// LDS of a k_search block for this space: model + per-thread scratch of the expansion, then the heap cache.  Returns the
// heap-cache entries that fit (0 = the kernel does not fit this robot: the host-driven search serves it).
int search_heap_cache_entries(const smplx_space* s, size_t* dynamic_bytes)
{
    const int block = s->ds.test_no_helper ? smplx_small_block(s->M) : smplx_search_block(s->M);
    if (block > 512 || s->M > 64) return 0;
    const int config_threads = smplx_small_block(s->M) - 64;      // per-thread scratch: the config waves only (k_search)
    const size_t base = (smplx_lds_bytes_n(s->blob_bytes, s->lds_nroot, s->ks.specialized ? 0 : s->model.dev.nslots, s->model.dev.nvars, s->model.dev.stack_bytes, config_threads) + 15) / 16 * 16;
    const size_t limit = 160 * 1024 - SMPLX_SEARCH_STATIC_LDS;
    if (base + 64 * sizeof(SmplxHeapEntry) > limit) return 0;
    size_t lh = (limit - base) / sizeof(SmplxHeapEntry);
    lh = std::min<size_t>(lh, 8192) / 64 * 64;
    if (dynamic_bytes) *dynamic_bytes = base + lh * sizeof(SmplxHeapEntry);
    return (int)lh;
}

bool search_on_device(const smplx_space* s)
{
    const char* e = getenv("SMPLX_SEARCH");
    if (e && !std::strcmp(e, "host")) return false;
    if (s->fused_mode || s->work_list_items > 0 || s->small_batch_max == 0) return false;   // spaces set up to exercise a particular host path
    return search_heap_cache_entries(s, nullptr) > 0;
}

int search_free(smplx_space* s)
{
    DevSearch& D = s->ds;
    if (D.arena) (void)hipFree(D.arena);
    if (D.d_hdr) (void)hipFree(D.d_hdr);
    D = DevSearch();
    return SMPLX_OK;
}

// (re)allocate the arena with at least the given capacities, keeping what the device holds
int search_reserve(smplx_space* s, const SearchCaps& want)
{
    DevSearch& D = s->ds;
    const int N = s->N;
    SearchCaps c = D.caps;
    bool grow = !D.arena;
    auto up = [&](int& have, int need) { if (need > have) { have = need; grow = true; } };
    up(c.states, want.states); up(c.heap, want.heap); up(c.incons, want.incons); up(c.log, want.log); up(c.succ, want.succ); up(c.path, want.path);
    if (!D.d_hdr) {
        HIP_TRY(hipMalloc((void**)&D.d_hdr, sizeof(SmplxSearchDev)));
        std::memset(&D.h, 0, sizeof(D.h));
    }
    if (grow) {
        unsigned char* fresh = nullptr;
        HIP_TRY(hipMalloc((void**)&fresh, search_arena_bytes(c, N)));
        SmplxSearchDev nh = D.h;
        search_carve(fresh, c, N, nh);
        if (D.arena) {
            // what the device holds moves over (the search may be in the middle of a replan)
            const SmplxSearchDev& o = D.h;
            const size_t ns = (size_t)std::max(D.dev_states, o.nstates);
            HIP_TRY(hipMemcpyAsync(nh.coord, o.coord, ns * N * sizeof(int32_t), hipMemcpyDeviceToDevice, s->stream));
            HIP_TRY(hipMemcpyAsync(nh.q, o.q, ns * N * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
            HIP_TRY(hipMemcpyAsync(nh.st, o.st, ns * sizeof(SmplxSState), hipMemcpyDeviceToDevice, s->stream));
            HIP_TRY(hipMemcpyAsync(nh.heap, o.heap, ((size_t)o.heap_size + 1) * sizeof(SmplxHeapEntry), hipMemcpyDeviceToDevice, s->stream));
            if (o.n_incons) HIP_TRY(hipMemcpyAsync(nh.incons, o.incons, (size_t)o.n_incons * sizeof(int32_t), hipMemcpyDeviceToDevice, s->stream));
            if (o.n_log) HIP_TRY(hipMemcpyAsync(nh.log, o.log, (size_t)o.n_log * sizeof(int32_t), hipMemcpyDeviceToDevice, s->stream));
            HIP_TRY(hipMemcpyAsync(nh.done_off, o.done_off, ns * sizeof(int32_t), hipMemcpyDeviceToDevice, s->stream));
            HIP_TRY(hipMemcpyAsync(nh.done_cnt, o.done_cnt, ns * sizeof(int32_t), hipMemcpyDeviceToDevice, s->stream));
            if (o.n_succ) HIP_TRY(hipMemcpyAsync(nh.succ, o.succ, (size_t)o.n_succ * sizeof(SmplxSucc), hipMemcpyDeviceToDevice, s->stream));
            HIP_TRY(hipStreamSynchronize(s->stream));
            (void)hipFree(D.arena);
        }
        D.arena = fresh;
        D.h = nh;
        D.caps = c;
        ++D.grows;
    }
    // the state table: at most half full with cap_states states
    size_t tcap = s->table_cap ? s->table_cap : ((size_t)1 << 16);
    while (tcap < 2 * (size_t)c.states) tcap *= 2;
    if (!s->d_table || tcap != s->table_cap) {
        HIP_TRY(hipStreamSynchronize(s->stream));
        if (int e = table_alloc(s, tcap)) return e;
        D.table_fresh = true;   // empty: every state is inserted again (k_search_table_fill)
        s->pending_ins.clear();
    }
    if (s->hs.search != D.d_hdr || grow || D.table_fresh) {
        s->hs.search = D.d_hdr;
        if (int e = upload_space(s)) return e;
    }
    return SMPLX_OK;
}

// states the HOST created and the device does not hold yet (the start state; the lattice of an earlier host-driven search)
int search_push_lattice(smplx_space* s)
{
    DevSearch& D = s->ds;
    const int N = s->N;
    const int have = (int)s->h_of_id.size();
    if (D.dev_states < have) {
        const int first = D.dev_states, n = have - first;
        HIP_TRY(hipMemcpyAsync(D.h.coord + (size_t)first * N, &s->coords[(size_t)first * N], sizeof(int32_t) * (size_t)n * N, hipMemcpyHostToDevice, s->stream));
        HIP_TRY(hipMemcpyAsync(D.h.q + (size_t)first * N, &s->qs[(size_t)first * N], sizeof(double) * (size_t)n * N, hipMemcpyHostToDevice, s->stream));
        std::vector<SmplxSState> st(n);
        std::vector<int32_t> off(n, -1), cnt(n, 0);
        for (int i = 0; i < n; ++i) {
            std::memset(&st[i], 0, sizeof(SmplxSState));
            st[i].h = (uint32_t)s->h_of_id[first + i];
            st[i].bp = -1;
        }
        HIP_TRY(hipMemcpyAsync(D.h.st + first, st.data(), sizeof(SmplxSState) * n, hipMemcpyHostToDevice, s->stream));
        HIP_TRY(hipMemcpyAsync(D.h.done_off + first, off.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, s->stream));
        HIP_TRY(hipMemcpyAsync(D.h.done_cnt + first, cnt.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));   // st / off / cnt are locals
        D.dev_states = have;
    }
    return SMPLX_OK;
}

// The device table holds every state: ids [1, nstates).  `first_missing` = the first id that may not be in it yet (states
// the host created since its last batch wait in pending_ins, in id order); a fresh table gets them all.
int search_fill_table(smplx_space* s, int first_missing, int nstates)
{
    DevSearch& D = s->ds;
    const int first = D.table_fresh ? 0 : first_missing;
    D.table_fresh = false;
    s->pending_ins.clear();
    s->table_count = nstates > 0 ? (size_t)nstates - 1 : 0;
    if (first >= nstates) return SMPLX_OK;
    hipLaunchKernelGGL(k_search_table_fill, dim3(std::max(1, std::min(1024, blocks_for(nstates - first, 256)))), dim3(256), 0, s->stream,
                       s->d_space, (const int32_t*)D.h.coord, first, nstates, s->N);
    HIP_TRY(hipGetLastError());
    return SMPLX_OK;
}

// the host's arrays catch up with what the device created: states [host count, device count), the committed successor
// lists of the states the device expanded, the expansion log of the last search
int pull_lattice(smplx_space* s)
{
    DevSearch& D = s->ds;
    if (!D.host_behind) return SMPLX_OK;
    const int N = s->N;
    HIP_TRY(hipStreamSynchronize(s->stream));
    SmplxSearchDev h;
    HIP_TRY(hipMemcpy(&h, D.d_hdr, sizeof(h), hipMemcpyDeviceToHost));
    const int have = (int)s->h_of_id.size(), total = h.nstates;
    if (total > have) {
        const int n = total - have;
        s->coords.resize((size_t)total * N);
        s->qs.resize((size_t)total * N);
        HIP_TRY(hipMemcpy(&s->coords[(size_t)have * N], h.coord + (size_t)have * N, sizeof(int32_t) * (size_t)n * N, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(&s->qs[(size_t)have * N], h.q + (size_t)have * N, sizeof(double) * (size_t)n * N, hipMemcpyDeviceToHost));
        std::vector<SmplxSState> st(n);
        HIP_TRY(hipMemcpy(st.data(), h.st + have, sizeof(SmplxSState) * n, hipMemcpyDeviceToHost));
        s->h_of_id.resize(total);
        for (int i = 0; i < n; ++i) s->h_of_id[have + i] = (int32_t)st[i].h;
        s->cache_off.resize(total, -1);
        s->cache_cnt.resize(total, 0);
        s->done_off.resize(total, -1);
        s->done_cnt.resize(total, 0);
        s->eval_count.resize(total, 0);
        if (s->plain_mode) s->g_est.resize(total, 1000000000u);
        for (int id = have; id < total; ++id) s->table.insert(id, s->coords);
        s->table_count = (size_t)total - 1;
    }
    // committed successor lists
    {
        std::vector<int32_t> off(total), cnt(total);
        HIP_TRY(hipMemcpy(off.data(), h.done_off, sizeof(int32_t) * total, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(cnt.data(), h.done_cnt, sizeof(int32_t) * total, hipMemcpyDeviceToHost));
        std::vector<SmplxSucc> succ((size_t)h.n_succ);
        if (h.n_succ) HIP_TRY(hipMemcpy(succ.data(), h.succ, sizeof(SmplxSucc) * (size_t)h.n_succ, hipMemcpyDeviceToHost));
        for (int id = 1; id < total; ++id) {
            if (off[id] < 0 || s->done_off[id] >= 0) continue;
            const int c = cnt[id] & 0xFF;
            s->done_off[id] = (int64_t)s->done_succ.size();
            s->done_cnt[id] = c;
            s->eval_count[id] = cnt[id] >> 8;
            for (int k = 0; k < c; ++k) {
                const SmplxSucc& e = succ[(size_t)off[id] + k];
                s->done_succ.push_back(e.id);
                s->done_cost.push_back(e.cost_prim & 0xFFFFFF);
                s->done_prim.push_back((int32_t)((uint32_t)e.cost_prim >> 24));
            }
        }
    }
    D.dev_states = std::max(D.dev_states, total);
    D.host_behind = false;
    return SMPLX_OK;
}

int pull_log(smplx_space* s)
{
    DevSearch& D = s->ds;
    if (!D.log_on_device) return SMPLX_OK;
    HIP_TRY(hipStreamSynchronize(s->stream));
    SmplxSearchDev h;
    HIP_TRY(hipMemcpy(&h, D.d_hdr, sizeof(h), hipMemcpyDeviceToHost));
    s->expansion_log.resize((size_t)h.n_log);
    if (h.n_log) HIP_TRY(hipMemcpy(s->expansion_log.data(), h.log, sizeof(int32_t) * (size_t)h.n_log, hipMemcpyDeviceToHost));
    D.log_on_device = false;
    return SMPLX_OK;
}

// set a query up for a replan from scratch on the device
int search_begin(smplx_space* s, const smplx_search_params* p)
{
    DevSearch& D = s->ds;
    if (int e = pull_lattice(s)) return e;       // a search the device ran earlier on this goal: one lattice
    const int have = (int)s->h_of_id.size();
    SearchCaps c;
    // a bounded search creates at most `bound * primitives` states; sized for ~10 per expansion so that the ordinary query
    // never has to stop for an enlargement (each costs its workgroup the rest of a launch)
    const long long bound = p->bounded ? (long long)std::max(p->max_expansions_init, p->max_expansions) : -1;
    const long long est_states = bound >= 0 ? std::min<long long>(bound * 10 + 4096, 1LL << 22) : 1LL << 18;
    c.states = (int)std::max<long long>(est_states, (long long)have + 2 * s->M + 64);
    if (D.test_capacity > 0) c.states = std::max(D.test_capacity, have + 2 * s->M + 64);
    c.heap = c.states + c.states / 4;
    c.incons = std::max(4096, c.states / 4);
    c.log = (int)(bound >= 0 ? std::min<long long>(bound + 1, 1LL << 26) : 1LL << 17);
    if (D.test_capacity > 0) c.log = std::min(c.log, std::max(64, D.test_capacity / 8));
    c.succ = (int)std::min<long long>((long long)c.log * std::min(s->M, 24), 1LL << 28);
    c.path = 1 << 16;
    if (int e = search_reserve(s, c)) return e;
    if (int e = search_push_lattice(s)) return e;
    if (int e = search_fill_table(s, have - (int)(s->pending_ins.size() / ((size_t)s->N + 2)), have)) return e;
    SmplxSearchDev& h = D.h;
    h.initial_eps = p->initial_eps;
    h.final_eps = std::max(p->final_eps, 1.0);   // ARAStar::setTargetEpsilon (arastar.h:112-114)
    h.delta_eps = p->delta_eps;
    h.improve = p->improve != 0; h.bounded = p->bounded != 0;
    h.max_init = p->max_expansions_init; h.max_rep = p->max_expansions;
    h.start_id = s->start_id;
    h.curr_eps = p->initial_eps; h.satisfied_eps = std::numeric_limits<double>::infinity();
    h.nstates = have;
    h.heap_size = 0; h.n_incons = 0; h.n_log = 0; h.n_path = 0;
    h.n_succ = D.n_succ_kept;
    h.iteration = 1; h.call_number = D.call_number; h.phase = 0; h.status = SMPLX_SS_RUNNING;
    h.num = 0; h.expand_count = 0; h.expand_count_init = 0; h.err = 0; h.solved = 0; h.cost = 0; h.dup_pushes = 0; h.grow_what = 0;
    h.goal_f = 1000000000u;
    h.committed_evals = 0; h.gpu_evals = 0; h.lookups = 0;
    for (int k = 0; k < 8; ++k) h.ticks[k] = 0;
    HIP_TRY(hipMemcpyAsync(D.d_hdr, &h, sizeof(h), hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    s->expansion_log.clear();
    return SMPLX_OK;
}

// the workgroup of this query stopped for room: twice of everything that can run out
int search_grow(smplx_space* s)
{
    DevSearch& D = s->ds;
    HIP_TRY(hipMemcpy(&D.h, D.d_hdr, sizeof(D.h), hipMemcpyDeviceToHost));
    SearchCaps c = D.caps;
    const SmplxSearchDev& h = D.h;
    const int M = s->M;
    if (h.nstates + 2 * M > c.states / 2) c.states = c.states * 2;
    if (h.heap_size + h.n_incons + 2 * M > c.heap / 2) c.heap = c.heap * 2;
    c.heap = std::max(c.heap, c.states + c.states / 4);
    if (h.n_incons + 2 * M > c.incons / 2) c.incons = c.incons * 2;
    if (h.n_log + 2 > c.log / 2) c.log = c.log * 2;
    if (h.n_succ + 2 * M > c.succ / 2) c.succ = c.succ * 2;
    const int nstates = h.nstates;
    if (int e = search_reserve(s, c)) return e;
    if (int e = search_fill_table(s, nstates, nstates)) return e;      // (only a fresh table needs anything)
    D.h.status = SMPLX_SS_RUNNING;
    D.h.grow_what = 0;
    HIP_TRY(hipMemcpyAsync(D.d_hdr, &D.h, sizeof(D.h), hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return SMPLX_OK;
}

// ARAStar::replan for nq queries that share scene, robot and primitives: one workgroup each, launched until all are done
int search_run(smplx_space** spaces, int nq, const smplx_search_params* p, int32_t* path_ids, int cap, smplx_search_stats* stats,
               double* t_done, std::chrono::steady_clock::time_point t0)
{
    smplx_space* lead = spaces[0];
    HIP_TRY(hipSetDevice(lead->device));
    for (int q = 0; q < nq; ++q)
        if (int e = search_begin(spaces[q], p)) return e;
    {
        std::vector<const SmplxSpaceDev*> tab(nq);
        for (int q = 0; q < nq; ++q) tab[q] = spaces[q]->d_space;
        if (int e = lead->b_stab.reserve(nq)) return e;
        HIP_TRY(hipMemcpy(lead->b_stab.p, tab.data(), sizeof(void*) * nq, hipMemcpyHostToDevice));
    }
    PinBuf<int32_t>& status = lead->p_ins;      // nq ints the workgroups write when they leave
    if (int e = status.reserve((size_t)nq)) return e;
    size_t lds = 0;
    const int lh = search_heap_cache_entries(lead, &lds);
    const int block = lead->ds.test_no_helper ? smplx_small_block(lead->M) : smplx_search_block(lead->M);
    int max_steps = 8192;
    std::vector<char> done(nq, 0);
    int remaining = nq;
    int64_t launches = 0;
    while (remaining > 0) {
        for (int q = 0; q < nq; ++q) status.p[q] = -1;
        KLAUNCH(lead, K_SEARCH, k_search, dim3(nq), dim3(block), lds, lead->stream, (const SmplxSpaceDev* const*)lead->b_stab.p, max_steps, lh, status.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(lead->batch_done, lead->stream));
        if (int e = wait_event_polling(lead->batch_done)) return e;
        ++launches;
        const auto now = std::chrono::steady_clock::now();
        for (int q = 0; q < nq; ++q) {
            if (done[q]) continue;
            const int st = status.p[q];
            if (st == SMPLX_SS_DONE) {
                done[q] = 1; --remaining;
                t_done[q] = std::chrono::duration<double>(now - t0).count();
            } else if (st == SMPLX_SS_GROW) {
                if (int e = search_grow(spaces[q])) return e;
            } else if (st == SMPLX_SS_ERROR) {
                return set_error(SMPLX_E_HIP, "device search: a workgroup gave up waiting for its own search wave (internal error)");
            } else if (st != SMPLX_SS_RUNNING) {
                return set_error(SMPLX_E_HIP, "device search: workgroup left no status (kernel fault?)");
            }
        }
    }
    // results
    for (int q = 0; q < nq; ++q) {
        smplx_space* s = spaces[q];
        DevSearch& D = s->ds;
        HIP_TRY(hipMemcpy(&D.h, D.d_hdr, sizeof(D.h), hipMemcpyDeviceToHost));
        const SmplxSearchDev& h = D.h;
        smplx_search_stats& st = stats[q];
        std::memset(&st, 0, sizeof(st));
        st.solved = h.solved;
        st.path_len = h.n_path;
        st.cost = h.cost;
        st.expansions = h.expand_count;
        st.expansions_init = h.expand_count_init;
        st.satisfied_eps = h.satisfied_eps;
        st.seconds = t_done[q];
        st.gpu_succ_evals = h.gpu_evals;
        st.committed_succ_evals = h.committed_evals;
        st.grid_lookups = h.lookups;
        st.gpu_batches = launches;
        st.cache_hits = h.expand_count;      // every expansion was served where the search runs
        st.cache_misses = 0;
        if (path_ids && h.n_path > 0) {
            std::vector<int32_t> rev((size_t)h.n_path);
            HIP_TRY(hipMemcpy(rev.data(), h.path, sizeof(int32_t) * (size_t)h.n_path, hipMemcpyDeviceToHost));
            for (int i = 0; i < h.n_path && i < cap; ++i) path_ids[(size_t)q * cap + i] = rev[(size_t)h.n_path - 1 - i];
        }
        D.call_number = h.call_number;
        D.n_succ_kept = h.n_succ;
        D.host_behind = true;
        D.log_on_device = true;
        D.searches += 1;
        for (int k = 0; k < 8; ++k) D.ticks[k] = h.ticks[k];
        D.dup_pushes = h.dup_pushes;
        s->committed_evals += h.committed_evals;
        s->gpu_evals += h.gpu_evals;
        s->gpu_batches += launches;
    }
    return SMPLX_OK;
}
