// One-launch windows (include/maniac_gpu.h): a window of speculative steps of ONE chain (mgpu_chain_window) and one step of a
// FARM of chains (mgpu_farm_window_*): evaluation, acceptance and commit in a single kernel, results through pinned host
// memory the host polls.
#include "mgpu_engine.h"

namespace mgpu {

// The engine's Topo in device memory (the window kernels index it by residue types they LOAD: as a by-value kernel argument
// the compiler would copy all of it into every lane's scratch); refreshed when the frames' buffers appear.
int chain_topo(mgpu_engine *e, const Topo **d_topo) {
    mgpu_engine::Chain &ch = e->chain;
    if (!ch.d_topo) {
        HIP_TRY(hipMalloc((void **)&ch.d_topo, sizeof(Topo)));
        ch.topo_stale = true;
    }
    if (ch.topo_stale) {
        for (auto &ln : e->lanes) HIP_TRY(hipStreamSynchronize(ln.stream));
        HIP_TRY(hipMemcpy(ch.d_topo, &e->tp, sizeof(Topo), hipMemcpyHostToDevice));
        ch.topo_stale = false;
    }
    *d_topo = ch.d_topo;
    return MGPU_OK;
}

// Every replica's current A(k) back into the primary buffer d_A (farm windows switch a replica between d_A and d_A_alt):
// drains the lanes, one small launch, a synchronise.  Windows still un-waited keep their results in host memory.
int farm_window_normalize(mgpu_engine *e) {
    e->farm.dirty = false;
    if (!e->farm.d_A_alt) return MGPU_OK;
    for (auto &ln : e->lanes) HIP_TRY(hipStreamSynchronize(ln.stream));
    hipLaunchKernelGGL(farm_normalize_kernel, dim3(e->n_replicas), dim3(kBlock), 0, e->lanes[0].stream, e->farm.d_acur, e->d_A,
                       (const double2 *)e->farm.d_A_alt, e->n_slots);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(e->lanes[0].stream));
    return MGPU_OK;
}

}  // namespace mgpu

extern "C" {

// ---- single-chain windows --------------------------------------------------------------------

// largest window the engine accepts, 0 where the one-launch path does not apply (molecules of more than kMaxFusedSitesWide sites
// among the active types, per-k reciprocal form)
static int chain_max_candidates(const mgpu_engine *e) {
    int n1_max = 1;
    for (int t = 0; t < e->tp.n_res; ++t) {
        if (!e->is_active[t]) continue;
        if (e->tp.n1[t] > kMaxFusedSitesWide || e->tp.site_major[t]) return 0;
        n1_max = std::max(n1_max, e->tp.n1[t]);
    }
    if (!recip_by_rows(e, n1_max)) return 0;
    if (e->coul_bytes > 64 * 1024) return 0;
    // the resolving workgroup stages every split partial of the window in LDS: 2 entries per candidate at most
    const int by_lds = (int)((size_t)64 * 1024 / ((size_t)2 * e->pair_nsplit * sizeof(double2)));
    return std::max(0, std::min(kChainMaxCand, by_lds));
}

int mgpu_chain_window_capacity(const mgpu_engine *e, int *max_candidates) {
    if (!e || !max_candidates) return set_error(MGPU_ERR_INVALID_ARG, "chain_window_capacity: null argument");
    *max_candidates = chain_max_candidates(e);
    return MGPU_OK;
}

int mgpu_chain_set_margin(mgpu_engine *e, double relative_margin) {
    if (!e || !(relative_margin >= 0.0)) return set_error(MGPU_ERR_INVALID_ARG, "chain_set_margin: bad argument");
    e->chain.margin = relative_margin;
    return MGPU_OK;
}

int mgpu_chain_set_timing(mgpu_engine *e, int on) {
    if (!e) return set_error(MGPU_ERR_INVALID_ARG, "null engine");
    e->chain.timing = on != 0;
    return MGPU_OK;
}

// Stage times of the LAST window in microseconds since its first workgroup started (100 MHz wall clock of the device):
//   us[0..3]   k role of candidate 0: start, phase tables built, k sweep summed, at the ticket
//   us[4..7]   first pair workgroup:  start, Coulomb table staged, its work units swept, at the ticket
//   us[8..14]  resolving workgroup:   last ticket drawn, acquire fence, partials reduced, decided, tag published,
//                                     commit tables built, commit done (the last two 0 when nothing was accepted)
int mgpu_chain_get_timing(mgpu_engine *e, double us[15]) {
    if (!e || !us) return set_error(MGPU_ERR_INVALID_ARG, "chain_get_timing: null argument");
    if (!e->chain.h_out) return set_error(MGPU_ERR_STATE, "chain_get_timing: no window has run");
    int rc = use_device(e);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(e->lanes[0].stream));      // the commit's stamps are written behind the tag
    const long long *ts = (const long long *)(e->chain.h_out + 10 * kChainMaxCand + 2);
    const long long t0 = std::min(ts[0], ts[kChainStamps]);
    const int first = ((const int *)(e->chain.h_out + 10 * (size_t)kChainMaxCand))[0];
    int k = 0;
    for (int i = 0; i < 4; ++i) us[k++] = (double)(ts[i] - t0) * 0.01;
    for (int i = 0; i < 4; ++i) us[k++] = (double)(ts[kChainStamps + i] - t0) * 0.01;
    for (int i = 0; i < 7; ++i) us[k++] = (i >= 5 && first < 0) ? 0.0 : (double)(ts[2 * kChainStamps + i] - t0) * 0.01;
    return MGPU_OK;
}

int mgpu_chain_get_stats(const mgpu_engine *e, long long *windows, long long *undecided) {
    if (!e) return set_error(MGPU_ERR_INVALID_ARG, "null engine");
    if (windows) *windows = e->chain.windows;
    if (undecided) *undecided = e->chain.undecided;
    return MGPU_OK;
}

int mgpu_chain_window(mgpu_engine *e, int replica, int n, const int *t, const int *m, const int *kind, const int *link,
                      const double *sites, int site_stride, const double *accept_u, const double *accept_pref,
                      double temperature, double recip_energy, double *old_energy, double *new_energy, int *first_accepted,
                      int *undecided) {
    if (!e) return set_error(MGPU_ERR_INVALID_ARG, "null engine");
    if (!t || !m || !kind || !link || !sites || !accept_u || !accept_pref || !old_energy || !new_energy || !first_accepted || !undecided)
        return set_error(MGPU_ERR_INVALID_ARG, "chain_window: null argument");
    const int n_max = chain_max_candidates(e);
    if (n_max == 0) return set_error(MGPU_ERR_STATE, "chain_window: not available for this engine (mgpu_chain_window_capacity)");
    if (n < 1 || n > n_max) return set_error(MGPU_ERR_INVALID_ARG, "chain_window: window size out of range");
    if (replica < 0 || replica >= e->n_replicas) return set_error(MGPU_ERR_INVALID_ARG, "chain_window: replica out of range");
    if (!(temperature > 0.0)) return set_error(MGPU_ERR_INVALID_ARG, "chain_window: temperature must be positive");
    int rc = use_device(e);
    if (rc) return rc;
    Lane &ln = e->lanes[0];
    if (ln.n_submitted != 0) return set_error(MGPU_ERR_STATE, "chain_window: lane 0 still holds an un-waited trial");
    mgpu_engine::Chain &ch = e->chain;
    if (!ch.h_tag) {
        HIP_TRY(hipHostMalloc((void **)&ch.h_out, sizeof(double) * (10 * kChainMaxCand + 2 + 3 * kChainStamps), hipHostMallocCoherent));
        std::memset(ch.h_out, 0, sizeof(double) * (10 * kChainMaxCand + 2 + 3 * kChainStamps));
        HIP_TRY(hipHostMalloc((void **)&ch.h_tag, 64, hipHostMallocCoherent));
        *ch.h_tag = 0;
        HIP_TRY(hipMalloc((void **)&ch.d_res, sizeof(ChainResult) * kChainMaxCand));
        HIP_TRY(hipMalloc((void **)&ch.d_part, sizeof(double2) * 2 * kChainMaxCand * (size_t)e->pair_nsplit));
        HIP_TRY(hipMalloc((void **)&ch.d_alt, sizeof(double2) * kChainMaxCand * (size_t)e->n_slots));
        HIP_TRY(hipMalloc((void **)&ch.d_ticket, sizeof(int)));
        HIP_TRY(hipMemset(ch.d_ticket, 0, sizeof(int)));
        HIP_TRY(hipDeviceSynchronize());
    }
    const Topo *d_topo = nullptr;
    if ((rc = chain_topo(e, &d_topo))) return rc;
    if (e->farm.dirty && (rc = farm_window_normalize(e))) return rc;
    // ---- the window travels in the kernel arguments
    ChainArgs g{};
    bool fast = replica_in_range(e, replica);
    char cand_ok[kChainMaxCand];
    int n1_max = 1, n_ent = 0;
    for (int c = 0; c < n; ++c) {
        const int k = kind[c];
        if (k < MGPU_MOVE || k > MGPU_DELETION) return set_error(MGPU_ERR_INVALID_ARG, "chain_window: unknown candidate kind");
        if (t[c] < 0 || t[c] >= e->tp.n_res) return set_error(MGPU_ERR_INVALID_ARG, "chain_window: residue type out of range");
        const int n1 = e->tp.n1[t[c]];
        if (n1 > site_stride || n1 > kMaxFusedSitesWide || e->tp.site_major[t[c]])
            return set_error(MGPU_ERR_INVALID_ARG, "chain_window: molecule too large for the one-launch path");
        const size_t idx = (size_t)replica * e->tp.n_res + t[c];
        if (e->d_com && e->frames_ok[idx])
            return set_error(MGPU_ERR_STATE, "chain_window: this replica holds molecule frames (mgpu_replica_set_frames)");
        const int lk = link[c];
        if (lk < -2 || lk >= n) return set_error(MGPU_ERR_INVALID_ARG, "chain_window: bad link");
        if (lk >= 0 && (k != MGPU_DELETION || link[lk] != -2 || kind[lk] != MGPU_CREATION || t[lk] != t[c]))
            return set_error(MGPU_ERR_INVALID_ARG, "chain_window: an as-written deletion links to an energy-only creation row of its type");
        if (lk == -2 && k != MGPU_CREATION) return set_error(MGPU_ERR_INVALID_ARG, "chain_window: energy-only rows are creation-kind");
        const int mc = (k == MGPU_CREATION) ? -1 : m[c];
        if ((rc = check_candidate(e, c, replica, t[c], mc, k != MGPU_CREATION))) return rc;
        if (k == MGPU_CREATION && lk != -2 && e->h_nmol[idx] >= e->tp.cap[t[c]])
            return set_error(MGPU_ERR_CAPACITY, "chain_window: residue type is at mol_capacity");
        n1_max = std::max(n1_max, n1);
        g.t[c] = t[c]; g.m[c] = mc; g.kind[c] = (signed char)k; g.link[c] = (signed char)lk;
        g.u[c] = accept_u[c]; g.pref[c] = accept_pref[c];
        const double *row = sites + (size_t)c * site_stride * 3;
        cand_ok[c] = 1;
        if (k != MGPU_DELETION) {
            // the engine's site order for a frozen type is not the caller's: such types are inactive and never move
            if (e->frozen[t[c]]) return set_error(MGPU_ERR_INVALID_ARG, "chain_window: frozen residue types do not move");
            std::memcpy(&g.sites[c][0][0], row, (size_t)n1 * 3 * sizeof(double));
            cand_ok[c] = sites_in_range(e, row, n1) ? 1 : 0;
            if (lk != -2) fast = fast && cand_ok[c];
        }
        g.ent_old_of[c] = g.ent_new_of[c] = -1;
        if (lk == -2) continue;
        if (k != MGPU_CREATION) { g.ent_old_of[c] = (signed char)n_ent; g.ent_c[n_ent] = (unsigned char)c; g.ent_new[n_ent] = 0; ++n_ent; }
        if (k != MGPU_DELETION) { g.ent_new_of[c] = (signed char)n_ent; g.ent_c[n_ent] = (unsigned char)c; g.ent_new[n_ent] = 1; ++n_ent; }
    }
    const int nsplit = e->pair_nsplit;
    const size_t lds = std::max(std::max(e->coul_bytes, recip_rows_lds_bytes(e, n1_max)), (size_t)n_ent * nsplit * sizeof(double2));
    if (lds > 64 * 1024) return set_error(MGPU_ERR_CAPACITY, "chain_window: the window does not fit the LDS budget");
    ch.seq += 1;
    for (int tt = 0; tt < e->tp.n_res; ++tt) g.self_of_type[tt] = e->self_of_type[tt];
    g.stamps = ch.timing ? 1 : 0;
    g.res = ch.d_res; g.partials = ch.d_part; g.ticket = ch.d_ticket; g.alt = ch.d_alt;
    g.host_out = ch.h_out; g.host_tag = ch.h_tag; g.seq = ch.seq;
    g.n = n; g.n_ent = n_ent; g.nsplit = nsplit; g.replica = replica;
    g.temperature = temperature; g.e_recip = recip_energy; g.margin = ch.margin;
    const int grid = n + (n_ent * nsplit + kPairWaves - 1) / kPairWaves;
    const bool ff = fast && e->pair_fast_fold;
    ln.dirty = true;
    ln.last_trial_n = 0;
    ln.d_trial_items = nullptr;
    ln.h_trial_items = nullptr;
#define MGPU_LAUNCH_CHAIN(FL, FW)                                                                                          \
    hipLaunchKernelGGL((chain_window_kernel<FL, FW>), dim3(grid), dim3(kChainBlock), lds, ln.stream, d_topo, e->bx, e->d_pos, e->d_nmol, \
                       e->d_res_q, e->d_res_atype, e->d_pair_tab, e->d_coul_tab, e->d_trj, e->d_tw, e->n_rtasks, e->d_rrows, e->n_rrows, \
                       e->d_A, g)
    if (e->bx.triclinic)
        hipLaunchKernelGGL((chain_window_kernel<false, false, true>), dim3(grid), dim3(kChainBlock), lds, ln.stream, d_topo, e->bx, e->d_pos,
                           e->d_nmol, e->d_res_q, e->d_res_atype, e->d_pair_tab, e->d_coul_tab, e->d_trj, e->d_tw, e->n_rtasks, e->d_rrows,
                           e->n_rrows, e->d_A, g);
    else if (e->pair_flat) { if (ff) MGPU_LAUNCH_CHAIN(true, true); else MGPU_LAUNCH_CHAIN(true, false); }
    else { if (ff) MGPU_LAUNCH_CHAIN(false, true); else MGPU_LAUNCH_CHAIN(false, false); }
#undef MGPU_LAUNCH_CHAIN
    HIP_TRY(hipGetLastError());
    // ---- wait for the tag: the results are in host memory when it shows this window's number
    {
        volatile unsigned long long *tag = ch.h_tag;
        long long spins = 0;
        while (*tag != ch.seq) {
            __builtin_ia32_pause();
            if (++spins >= 20000 && (spins % 4096) == 0) {
                // long past any window's run time: make sure the stream is still alive
                const hipError_t q = hipStreamQuery(ln.stream);
                if (q == hipSuccess && *tag != ch.seq) return set_error(MGPU_ERR_HIP, "chain_window: the kernel finished without publishing its results");
                if (q != hipSuccess && q != hipErrorNotReady) return set_error(MGPU_ERR_HIP, std::string("chain_window: ") + hipGetErrorString(q));
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    for (int c = 0; c < n; ++c) {
        std::memcpy(old_energy + 5 * (size_t)c, ch.h_out + 10 * (size_t)c, 5 * sizeof(double));
        std::memcpy(new_energy + 5 * (size_t)c, ch.h_out + 10 * (size_t)c + 5, 5 * sizeof(double));
    }
    const int *hi = (const int *)(ch.h_out + 10 * (size_t)kChainMaxCand);
    const int first = hi[0], und = hi[1];
    *first_accepted = first;
    *undecided = und;
    ch.windows += 1;
    if (und >= 0) ch.undecided += 1;
    if (first >= 0) {
        // the device is committing candidate `first` behind the tag: the host mirrors follow
        const size_t idx = (size_t)replica * e->tp.n_res + t[first];
        if (kind[first] == MGPU_CREATION) e->h_nmol[idx] += 1;
        if (kind[first] == MGPU_DELETION) e->h_nmol[idx] -= 1;
        if (kind[first] != MGPU_DELETION && !cand_ok[first]) e->in_range[idx] = 0;
        // (an as-written deletion moves resident atoms only: the range flag stands)
        frozen_changed(e, replica, t[first]);
    }
    return MGPU_OK;
}


// ---- farm windows ------------------------------------------------------------------------------
// One launch per lane step of a farm of chains (farm_window_kernel, mgpu_kernels_windows.h): the caller hands over, per chain, the
// move it selected and the uniform numbers of its construction and of its acceptance test; the launch evaluates,
// decides and commits; the host collects energies and verdicts from pinned memory by polling per-chain tags.  Up to
// kFarmDepth windows per lane may be in flight (a farm whose move selection does not depend on earlier outcomes -- NVT --
// queues the next step before it has seen the last).

// chains per launch the engine accepts, 0 where the path does not apply (triclinic box, an active molecule of more than
// kMaxFusedSitesWide sites, per-k reciprocal form, a Coulomb table beyond 64 KiB)
static int farm_max_chains(const mgpu_engine *e) {
    if (e->bx.triclinic) return 0;
    int n1_max = 1;
    for (int t = 0; t < e->tp.n_res; ++t) {
        if (!e->is_active[t]) continue;
        if (e->tp.n1[t] > kMaxFusedSitesWide || e->tp.site_major[t]) return 0;
        n1_max = std::max(n1_max, e->tp.n1[t]);
    }
    if (!recip_by_rows(e, n1_max)) return 0;
    if (e->coul_bytes > 64 * 1024) return 0;
    if ((size_t)kPairWaves * (4 * e->pair_nsplit + 4) * sizeof(double) > 64 * 1024) return 0;      // the resolving waves' scratch
    return std::min(kFarmMaxChains, e->n_replicas);
}

int mgpu_farm_window_capacity(const mgpu_engine *e, int *max_chains, int *max_in_flight) {
    if (!e || !max_chains) return set_error(MGPU_ERR_INVALID_ARG, "farm_window_capacity: null argument");
    *max_chains = farm_max_chains(e);
    if (max_in_flight) *max_in_flight = kFarmDepth;
    return MGPU_OK;
}

int mgpu_farm_window_get_stats(const mgpu_engine *e, long long *windows, long long *undecided) {
    if (!e) return set_error(MGPU_ERR_INVALID_ARG, "null engine");
    if (windows) *windows = e->farm.windows;
    if (undecided) *undecided = e->farm.undecided;
    return MGPU_OK;
}

static int farm_lane(mgpu_engine *e, int lane, Lane **ln) {
    if (!e) return set_error(MGPU_ERR_INVALID_ARG, "null engine");
    if (lane < 0 || lane >= kLanes) return set_error(MGPU_ERR_INVALID_ARG, "lane out of range");
    *ln = &e->lanes[lane];
    return MGPU_OK;
}

// the lane's blocks for windows of up to `cap` chains; the engine's second A(k) buffer and flags
static int farm_reserve(mgpu_engine *e, Lane &ln, int cap) {
    Lane::FarmWindow &fw = ln.farm;
    if (!e->farm.d_A_alt) {
        const size_t bytes = (size_t)e->n_replicas * e->n_slots * sizeof(double2);
        HIP_TRY(hipMalloc((void **)&e->farm.d_A_alt, bytes));
        HIP_TRY(hipMalloc((void **)&e->farm.d_acur, (size_t)e->n_replicas * sizeof(int)));
        HIP_TRY(hipMalloc((void **)&e->farm.d_stalled, (size_t)e->n_replicas * sizeof(int)));
        HIP_TRY(hipMemset(e->farm.d_acur, 0, (size_t)e->n_replicas * sizeof(int)));
        HIP_TRY(hipMemset(e->farm.d_stalled, 0, (size_t)e->n_replicas * sizeof(int)));
        HIP_TRY(hipDeviceSynchronize());
    }
    if (cap <= fw.cap) return MGPU_OK;
    if (!fw.pending.empty()) return set_error(MGPU_ERR_STATE, "farm_window_submit: a larger window than the lane's blocks while windows are in flight");
    HIP_TRY(hipStreamSynchronize(ln.stream));
    const unsigned long long seq = fw.seq;
    fw.release();
    fw.seq = seq;
    cap = std::max(cap, 64);
    HIP_TRY(hipHostMalloc((void **)&fw.h_recs, (size_t)kFarmDepth * cap * sizeof(FarmRec), hipHostMallocCoherent));
    HIP_TRY(hipHostMalloc((void **)&fw.h_out, (size_t)kFarmDepth * cap * kFarmOut * sizeof(double), hipHostMallocCoherent));
    HIP_TRY(hipHostMalloc((void **)&fw.h_tag, (size_t)kFarmDepth * cap * sizeof(unsigned long long), hipHostMallocCoherent));
    std::memset(fw.h_tag, 0xff, (size_t)kFarmDepth * cap * sizeof(unsigned long long));      // no window carries this number
    HIP_TRY(hipMalloc((void **)&fw.d_part, (size_t)cap * 2 * e->pair_nsplit * sizeof(double2)));
    HIP_TRY(hipMalloc((void **)&fw.d_res, (size_t)cap * sizeof(ChainResult)));
    HIP_TRY(hipMalloc((void **)&fw.d_tickets, (size_t)cap * sizeof(int)));
    HIP_TRY(hipMemset(fw.d_tickets, 0, (size_t)cap * sizeof(int)));
    HIP_TRY(hipMemset(fw.d_res, 0, (size_t)cap * sizeof(ChainResult)));
    HIP_TRY(hipDeviceSynchronize());
    fw.cap = cap;
    return MGPU_OK;
}

int mgpu_farm_window_submit(mgpu_engine *e, int lane, int n, const int *replica, const int *t, const int *m, const int *move,
                            const int *forced, const double *u5, const double *accept_u, const double *accept_pref,
                            const double *slot_u, double t_step, double r_step, double temperature) {
    Lane *lp = nullptr;
    int rc = farm_lane(e, lane, &lp);
    if (rc) return rc;
    Lane &ln = *lp;
    if (!replica || !t || !m || !move || !u5 || !accept_u || !accept_pref)
        return set_error(MGPU_ERR_INVALID_ARG, "farm_window_submit: null argument");
    const int n_max = farm_max_chains(e);
    if (n_max == 0) return set_error(MGPU_ERR_STATE, "farm_window_submit: not available for this engine (mgpu_farm_window_capacity)");
    if (n < 1 || n > n_max) return set_error(MGPU_ERR_INVALID_ARG, "farm_window_submit: number of chains out of range");
    if (!(temperature > 0.0)) return set_error(MGPU_ERR_INVALID_ARG, "farm_window_submit: temperature must be positive");
    if ((rc = use_device(e))) return rc;
    if (ln.n_submitted != 0) return set_error(MGPU_ERR_STATE, "farm_window_submit: the lane still holds an un-waited trial");
    Lane::FarmWindow &fw = ln.farm;
    if ((int)fw.pending.size() >= kFarmDepth) return set_error(MGPU_ERR_STATE, "farm_window_submit: too many windows of this lane in flight");
    if (!fw.pending.empty() && fw.pending.back().counts_change)
        return set_error(MGPU_ERR_STATE, "farm_window_submit: the lane's last window carries an insertion / deletion: collect it first");
    const Topo *d_topo = nullptr;
    {
        // (one lane at a time: the engine's second A(k) buffer, its flags and the device topology are made on first use)
        std::lock_guard<std::mutex> lock(e->farm.mu);
        if ((rc = farm_reserve(e, ln, n))) return rc;
        if ((rc = chain_topo(e, &d_topo))) return rc;
    }
    const int slot = (int)(fw.seq % kFarmDepth);
    Lane::FarmWindow::Pending pd;
    pd.seq = fw.seq + 1;
    pd.n = n; pd.slot = slot; pd.counts_change = false;
    pd.rep.assign(replica, replica + n);
    pd.t.assign(t, t + n);
    pd.kind.assign(n, -1);
    pd.ok.assign(n, 1);
    FarmArgs g{};
    FarmRec *recs = n <= kFarmInline ? g.inline_recs : fw.h_recs + (size_t)slot * fw.cap;
    bool fast = true;
    int n1_max = 1;
    if ((int)ln.mark.size() != e->n_replicas) ln.mark.assign(e->n_replicas, -1);
    bool twice = false;
    for (int c = 0; c < n; ++c) {
        FarmRec &r = recs[c];
        r = FarmRec{};
        const int mv = move[c];
        if (mv < 0 || mv > 4) return set_error(MGPU_ERR_INVALID_ARG, "farm_window_submit: unknown move code");
        if (replica[c] < 0 || replica[c] >= e->n_replicas) return set_error(MGPU_ERR_INVALID_ARG, "farm_window_submit: replica out of range");
        twice = twice || ln.mark[replica[c]] == -3;
        ln.mark[replica[c]] = -3;
        r.replica = replica[c];
        if (mv == 0) continue;                                   // the chain does nothing this step
        if (t[c] < 0 || t[c] >= e->tp.n_res) { rc = set_error(MGPU_ERR_INVALID_ARG, "farm_window_submit: residue type out of range"); break; }
        const int n1 = e->tp.n1[t[c]];
        if (n1 > kMaxFusedSitesWide || e->tp.site_major[t[c]] || e->frozen[t[c]]) { rc = set_error(MGPU_ERR_INVALID_ARG, "farm_window_submit: molecule too large for the one-launch path"); break; }
        const size_t idx = (size_t)replica[c] * e->tp.n_res + t[c];
        if (!e->d_com || !e->frames_ok[idx]) { rc = set_error(MGPU_ERR_STATE, "farm_window_submit: no molecule frames for chain " + std::to_string(c) + " (mgpu_replica_set_frames)"); break; }
        const int k = mv <= 2 ? MGPU_MOVE : (mv == 3 ? MGPU_CREATION : MGPU_DELETION);
        const int mc = k == MGPU_CREATION ? -1 : (slot_u ? 0 : m[c]);
        if (!slot_u) {
            // the caller picked the molecule: against the engine's counts, which must then be current
            if ((rc = check_candidate(e, c, replica[c], t[c], mc, k != MGPU_CREATION))) break;
            if (k == MGPU_CREATION && e->h_nmol[idx] >= e->tp.cap[t[c]]) { rc = set_error(MGPU_ERR_CAPACITY, "farm_window_submit: residue type is at mol_capacity"); break; }
            if (k == MGPU_CREATION && e->h_nmol[idx] < 1) { rc = set_error(MGPU_ERR_STATE, "farm_window_submit: an insertion copies the geometry of molecule 1 of its type"); break; }
            if (k != MGPU_MOVE) pd.counts_change = true;
        } else {
            // the device picks it from the replica's count when the launch runs (FarmRec::by_count)
            if (!(slot_u[c] >= 0.0 && slot_u[c] < 1.0)) { rc = set_error(MGPU_ERR_INVALID_ARG, "farm_window_submit: slot_u must lie in [0, 1)"); break; }
            r.by_count = 1;
            r.sel_u = slot_u[c];
        }
        n1_max = std::max(n1_max, n1);
        pd.kind[c] = k;
        // a built candidate's centre lies in the cell; with tight frames its sites are within the fast fold's range
        fast = fast && replica_in_range(e, replica[c]);
        if (k != MGPU_DELETION) { pd.ok[c] = e->frames_tight[idx]; fast = fast && pd.ok[c]; }
        r.t = t[c]; r.m = (k == MGPU_CREATION || slot_u) ? 0 : m[c]; r.move = mv;
        r.forced = forced ? forced[c] : 0;
        if (r.forced < 0 || r.forced > 2) { rc = set_error(MGPU_ERR_INVALID_ARG, "farm_window_submit: forced is 0, 1 (accept) or 2 (reject)"); break; }
        for (int d = 0; d < 5; ++d) r.u[d] = u5[5 * (size_t)c + d];
        r.acc_u = accept_u[c];
        r.pref = accept_pref[c];
    }
    for (int c = 0; c < n; ++c)
        if (replica[c] >= 0 && replica[c] < e->n_replicas) ln.mark[replica[c]] = -1;
    if (rc) return rc;
    if (twice) return set_error(MGPU_ERR_INVALID_ARG, "farm_window_submit: more than one chain record for a replica");
    if (pd.counts_change && !fw.pending.empty())
        return set_error(MGPU_ERR_STATE, "farm_window_submit: a window with an insertion / deletion needs the lane's earlier windows collected "
                                         "(its molecule counts must be current)");
    const int nsplit = e->pair_nsplit;
    const size_t lds = std::max(std::max(e->coul_bytes, recip_rows_lds_bytes(e, n1_max)), (size_t)kPairWaves * (4 * nsplit + 4) * sizeof(double));
    if (lds > 64 * 1024) return set_error(MGPU_ERR_CAPACITY, "farm_window_submit: the window does not fit the LDS budget");
    fw.seq += 1;
    for (int tt = 0; tt < e->tp.n_res; ++tt) g.self_of_type[tt] = e->self_of_type[tt];
    g.recs = fw.h_recs + (size_t)slot * fw.cap;
    g.partials = fw.d_part; g.res = fw.d_res; g.tickets = fw.d_tickets;
    g.stalled = e->farm.d_stalled; g.acur = e->farm.d_acur; g.A_alt = e->farm.d_A_alt;
    g.host_out = fw.h_out + (size_t)slot * fw.cap * kFarmOut;
    g.host_tag = fw.h_tag + (size_t)slot * fw.cap;
    g.seq = pd.seq;
    g.n = n; g.nsplit = nsplit;
    g.t_step = t_step; g.r_step = r_step; g.temperature = temperature; g.margin = e->chain.margin;
    const int wpc = 2 * nsplit;
    const int grid = (n * wpc + kPairWaves - 1) / kPairWaves + n;
    const bool ff = fast && e->pair_fast_fold;
    ln.dirty = true;
    ln.last_trial_n = 0;
    ln.d_trial_items = nullptr;
    ln.h_trial_items = nullptr;
    e->farm.dirty = true;
#define MGPU_LAUNCH_FARM(FL, FW)                                                                                           \
    hipLaunchKernelGGL((farm_window_kernel<FL, FW>), dim3(grid), dim3(kChainBlock), lds, ln.stream, d_topo, e->bx, e->d_pos, e->d_nmol, \
                       e->d_res_q, e->d_res_atype, e->d_pair_tab, e->d_coul_tab, e->d_trj, e->d_tw, e->n_rtasks, e->d_rrows, e->n_rrows, \
                       e->d_A, g)
    if (e->pair_flat) { if (ff) MGPU_LAUNCH_FARM(true, true); else MGPU_LAUNCH_FARM(true, false); }
    else { if (ff) MGPU_LAUNCH_FARM(false, true); else MGPU_LAUNCH_FARM(false, false); }
#undef MGPU_LAUNCH_FARM
    HIP_TRY(hipGetLastError());
    fw.pending.push_back(std::move(pd));
    e->farm.windows += 1;
    return MGPU_OK;
}

// The lane's OLDEST window in flight: energies (rows of five: non_coulomb, coulomb, recip_coulomb, ewald_self, intra_coulomb,
// as mgpu_gcmc_trial_wait fills them) and one verdict per chain -- 0 rejected, 1 accepted (and committed), 2 undecided (the
// host decides and sends the step again with `forced`), 4 nothing done: the replica waits for such a decision, 5 idle record.
int mgpu_farm_window_wait(mgpu_engine *e, int lane, double *old_energy, double *new_energy, int *verdict) {
    Lane *lp = nullptr;
    int rc = farm_lane(e, lane, &lp);
    if (rc) return rc;
    Lane &ln = *lp;
    if (!old_energy || !new_energy || !verdict) return set_error(MGPU_ERR_INVALID_ARG, "farm_window_wait: null argument");
    Lane::FarmWindow &fw = ln.farm;
    if (fw.pending.empty()) return set_error(MGPU_ERR_STATE, "farm_window_wait: no window of this lane is in flight");
    const Lane::FarmWindow::Pending &pd = fw.pending.front();
    const int n = pd.n;
    const double *out = fw.h_out + (size_t)pd.slot * fw.cap * kFarmOut;
    volatile unsigned long long *tag = fw.h_tag + (size_t)pd.slot * fw.cap;
    long long spins = 0;
    for (int c = 0; c < n; ++c) {
        while (tag[c] != pd.seq) {
            __builtin_ia32_pause();
            if (++spins >= 200000 && (spins % 65536) == 0) {
                // long past any window's run time: make sure the stream is still alive
                const hipError_t q = hipStreamQuery(ln.stream);
                if (q == hipSuccess && tag[c] != pd.seq) return set_error(MGPU_ERR_HIP, "farm_window_wait: the kernel finished without publishing its results");
                if (q != hipSuccess && q != hipErrorNotReady) return set_error(MGPU_ERR_HIP, std::string("farm_window_wait: ") + hipGetErrorString(q));
            }
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    for (int c = 0; c < n; ++c) {
        const double *o = out + (size_t)kFarmOut * c;
        std::memcpy(old_energy + 5 * (size_t)c, o, 5 * sizeof(double));
        std::memcpy(new_energy + 5 * (size_t)c, o + 5, 5 * sizeof(double));
        const int v = (int)o[10];
        verdict[c] = v;
        if (v == kFarmVerdictUndecided) e->farm.undecided += 1;
        if (v != kFarmVerdictAccepted) continue;
        // the device has committed the step: the host mirrors follow
        const size_t idx = (size_t)pd.rep[c] * e->tp.n_res + pd.t[c];
        if (pd.kind[c] == MGPU_CREATION) e->h_nmol[idx] += 1;
        if (pd.kind[c] == MGPU_DELETION) e->h_nmol[idx] -= 1;
        if (pd.kind[c] != MGPU_DELETION && !pd.ok[c]) e->in_range[idx] = 0;
    }
    fw.pending.pop_front();
    return MGPU_OK;
}

#ifdef MGPU_FARM_STAMPS
// diagnostic builds only (tools/farm_stages.py): the last window's stamps, 3 x 8 ticks of the 100 MHz wall clock
int mgpu_farm_window_get_stamps(mgpu_engine *e, long long *out24) {
    if (!e || !out24) return set_error(MGPU_ERR_INVALID_ARG, "farm_window_get_stamps: null argument");
    int rc = use_device(e);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out24, HIP_SYMBOL(g_farm_stamps), 24 * sizeof(long long)));
    return MGPU_OK;
}
#endif

int mgpu_farm_window_flush(mgpu_engine *e) {
    if (!e) return set_error(MGPU_ERR_INVALID_ARG, "null engine");
    int rc = use_device(e);
    if (rc) return rc;
    return sync_all_lanes(e);
}

}  // extern "C"
