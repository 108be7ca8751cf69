// Batched candidates and the asynchronous submission lanes (include/maniac_gpu.h): trial submit / wait / commit, device-built
// and device-decided trials, the host team that runs their per-candidate loops.
#include "mgpu_engine.h"

extern "C" {

// ---- batched candidates ----------------------------------------------------------------------

int mgpu_pair_energy_candidates(mgpu_engine *e, int n, const int *replica, const int *t, const int *m,
                                const int *use_resident, const double *sites, int site_stride, double *e_nc,
                                double *e_c) {
    if (!e) return set_error(MGPU_ERR_INVALID_ARG, "null engine");
    if (n == 0) return MGPU_OK;
    if (n < 0 || !replica || !t || !m || !e_nc || !e_c) return set_error(MGPU_ERR_INVALID_ARG, "pair_energy_candidates: bad argument");
    int rc = use_device(e);
    if (rc) return rc;
    if ((rc = sync_all_lanes(e))) return rc;
    std::vector<PairItem> items(n);
    bool any_sites = false;
    for (int c = 0; c < n; ++c) {
        const bool res = use_resident && use_resident[c];
        if ((rc = check_candidate(e, c, replica[c], t[c], m[c], res))) return rc;
        if (!res) {
            any_sites = true;
            if (e->tp.n1[t[c]] > site_stride) return set_error(MGPU_ERR_INVALID_ARG, "site_stride smaller than atoms_in_res");
        }
        items[c] = PairItem{replica[c], t[c], m[c], res ? -1 : c, 0};
    }
    if (any_sites && !sites) return set_error(MGPU_ERR_INVALID_ARG, "pair_energy_candidates: sites is null");
    if ((rc = e->d_items.reserve(n * sizeof(PairItem)))) return rc;
    if ((rc = e->d_out.reserve((size_t)2 * n * sizeof(double)))) return rc;
    if ((rc = e->h_out.reserve((size_t)2 * n * sizeof(double)))) return rc;
    HIP_TRY(hipMemcpyAsync(e->d_items.p, items.data(), n * sizeof(PairItem), hipMemcpyHostToDevice, e->stream));
    if (any_sites && (rc = upload_sites(e, sites, n, site_stride, t))) return rc;
    double *d_lj = (double *)e->d_out.p, *d_c = d_lj + n;
    const int nsplit = e->pair_nsplit;
    bool fast = true;
    for (int c = 0; c < n && fast; ++c) {
        fast = replica_in_range(e, replica[c]);
        if (fast && !(use_resident && use_resident[c]))
            fast = sites_in_range(e, sites + (size_t)c * site_stride * 3, e->tp.n1[t[c]]);
    }
    if ((rc = launch_pair(e, e->lanes[0], (const PairItem *)e->d_items.p, n, common_site_count(e, items), site_stride, nsplit, d_lj, d_c,
                          false, nullptr, false, fast))) return rc;
    HIP_TRY(hipMemcpyAsync(e->h_out.p, e->d_out.p, (size_t)2 * n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    if ((rc = sync_stream(e))) return rc;
    std::memcpy(e_nc, e->h_out.p, n * sizeof(double));
    std::memcpy(e_c, (double *)e->h_out.p + n, n * sizeof(double));
    return MGPU_OK;
}

int mgpu_recip_energy_candidates(mgpu_engine *e, int n, const int *replica, const int *t, const int *m, const int *kind,
                                 const double *sites, int site_stride, double *u) {
    if (!e) return set_error(MGPU_ERR_INVALID_ARG, "null engine");
    if (n == 0) return MGPU_OK;
    if (n < 0 || !replica || !t || !m || !kind || !u) return set_error(MGPU_ERR_INVALID_ARG, "recip_energy_candidates: bad argument");
    int rc = use_device(e);
    if (rc) return rc;
    if ((rc = sync_all_lanes(e))) return rc;
    std::vector<RecipItem> items(n);
    bool any_sites = false;
    int n1_max = 1;
    for (int c = 0; c < n; ++c) {
        if (kind[c] < MGPU_MOVE || kind[c] > MGPU_NONE) return set_error(MGPU_ERR_INVALID_ARG, "unknown candidate kind");
        const bool need_old = (kind[c] == MGPU_MOVE || kind[c] == MGPU_DELETION);
        const bool need_new = (kind[c] == MGPU_MOVE || kind[c] == MGPU_CREATION);
        if ((rc = check_candidate(e, c, replica[c], t[c], m[c], need_old))) return rc;
        if (need_new) {
            any_sites = true;
            if (e->tp.n1[t[c]] > site_stride) return set_error(MGPU_ERR_INVALID_ARG, "site_stride smaller than atoms_in_res");
        }
        n1_max = std::max(n1_max, e->tp.n1[t[c]]);
        items[c] = RecipItem{replica[c], t[c], m[c], kind[c], need_new ? c : -1, 0};
    }
    if (any_sites && !sites) return set_error(MGPU_ERR_INVALID_ARG, "recip_energy_candidates: sites is null");
    if ((rc = e->d_items2.reserve(n * sizeof(RecipItem)))) return rc;
    if ((rc = e->d_out.reserve((size_t)n * sizeof(double)))) return rc;
    if ((rc = e->h_out.reserve((size_t)n * sizeof(double)))) return rc;
    HIP_TRY(hipMemcpyAsync(e->d_items2.p, items.data(), n * sizeof(RecipItem), hipMemcpyHostToDevice, e->stream));
    if (any_sites && (rc = upload_sites(e, sites, n, site_stride, t))) return rc;
    if ((rc = launch_recip(e, e->lanes[0], (const RecipItem *)e->d_items2.p, n, n1_max, site_stride, false, e->d_A, (double *)e->d_out.p)))
        return rc;
    HIP_TRY(hipMemcpyAsync(e->h_out.p, e->d_out.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    if ((rc = sync_stream(e))) return rc;
    std::memcpy(u, e->h_out.p, n * sizeof(double));
    return MGPU_OK;
}

int mgpu_self_energy(const mgpu_engine *e, int t, double *e_self) {
    if (!e || !e_self) return set_error(MGPU_ERR_INVALID_ARG, "mgpu_self_energy: null argument");
    if (t < 0 || t >= e->tp.n_res) return set_error(MGPU_ERR_INVALID_ARG, "residue type out of range");
    *e_self = self_energy_host(e, t);
    return MGPU_OK;
}

int mgpu_intra_energy_candidates(mgpu_engine *e, int n, const int *replica, const int *t, const int *m,
                                 const int *use_resident, const double *sites, int site_stride, double *u) {
    if (!e) return set_error(MGPU_ERR_INVALID_ARG, "null engine");
    if (n == 0) return MGPU_OK;
    if (n < 0 || !replica || !t || !m || !u) return set_error(MGPU_ERR_INVALID_ARG, "intra_energy_candidates: bad argument");
    int rc = use_device(e);
    if (rc) return rc;
    if ((rc = sync_all_lanes(e))) return rc;
    std::vector<PairItem> items(n);
    bool any_sites = false;
    for (int c = 0; c < n; ++c) {
        const bool res = use_resident && use_resident[c];
        if ((rc = check_candidate(e, c, replica[c], t[c], m[c], res))) return rc;
        if (!res) {
            any_sites = true;
            if (e->tp.n1[t[c]] > site_stride) return set_error(MGPU_ERR_INVALID_ARG, "site_stride smaller than atoms_in_res");
        }
        items[c] = PairItem{replica[c], t[c], m[c], res ? -1 : c, 0};
    }
    if (any_sites && !sites) return set_error(MGPU_ERR_INVALID_ARG, "intra_energy_candidates: sites is null");
    if ((rc = e->d_items.reserve(n * sizeof(PairItem)))) return rc;
    if ((rc = e->d_out.reserve((size_t)n * sizeof(double)))) return rc;
    if ((rc = e->h_out.reserve((size_t)n * sizeof(double)))) return rc;
    HIP_TRY(hipMemcpyAsync(e->d_items.p, items.data(), n * sizeof(PairItem), hipMemcpyHostToDevice, e->stream));
    if (any_sites && (rc = upload_sites(e, sites, n, site_stride, t))) return rc;
    if ((rc = launch_intra(e, e->lanes[0], (const PairItem *)e->d_items.p, n, (const double *)e->d_sites.p, site_stride, (double *)e->d_out.p)))
        return rc;
    HIP_TRY(hipMemcpyAsync(e->h_out.p, e->d_out.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    if ((rc = sync_stream(e))) return rc;
    std::memcpy(u, e->h_out.p, n * sizeof(double));
    return MGPU_OK;
}

// Queue one trial per candidate on a lane: inputs are staged through pinned host memory, so the call
// returns as soon as the copies and the kernels are enqueued.  kind == nullptr: all MGPU_MOVE.
// Per candidate (ComputeOldEnergy / ComputeNewEnergy, monte_carlo_utils.f90:275-395):
//   MOVE      pair(resident) | pair(sites)          recip(A) | recip(A + new - old)
//   CREATION  --             | pair(sites), intra   recip(A) | recip(A + new)         (m ignored)
//   DELETION  pair(resident), intra | --            recip(A) | recip(A - old)
// One pass over k per candidate yields both reciprocal energies.  Device output rows (doubles):
//   lj[n_pair] c[n_pair] u_old[n] u_new[n] intra[n]; the lane remembers where each candidate's
//   pair items are.
// build != nullptr: the candidate rows are built on the device (trial_build_kernel) from the molecule frames, the move
// codes (1 translation, 2 rotation, 3 creation, 4 deletion) and five uniform numbers per candidate; `sites` is null and
// site_stride is ignored (a row is [sites (n1_max) | com | offsets (n1_max)])
struct TrialBuild {
    const int *move;
    const double *u;              // [n][5]
    double t_step, r_step;
};
// decide != nullptr: the acceptance test runs on the device behind the k sweep and accepted candidates are committed there
// (DecideItem, mgpu_kernels_recip.h); accept_u[n] = the test's uniform numbers, accept_pref[n] = its prefactors
struct TrialDecide {
    const double *u, *pref;
    double temperature;
};
static int trial_submit_impl(mgpu_engine *e, Lane &ln, int n, const int *replica, const int *t, const int *m,
                             const int *kind, const double *sites, int site_stride, const TrialBuild *build = nullptr,
                             const TrialDecide *decide = nullptr) {
    if (ln.n_submitted != 0) return set_error(MGPU_ERR_STATE, "trial_submit: the lane still holds an un-waited trial");
    int rc;
    if (e->farm.dirty && (rc = farm_window_normalize(e))) return rc;       // (farm windows ran before: A(k) back into its primary buffer)
    if (decide) {
        if (!(decide->temperature > 0.0)) return set_error(MGPU_ERR_INVALID_ARG, "trial_decide_submit: temperature must be positive");
        // one candidate per replica: the workgroups commit independently
        if ((int)ln.mark.size() != e->n_replicas) ln.mark.assign(e->n_replicas, -1);
        bool twice = false;
        for (int c = 0; c < n; ++c) {
            if (replica[c] < 0 || replica[c] >= e->n_replicas) return set_error(MGPU_ERR_INVALID_ARG, "trial_decide_submit: replica out of range");
            twice = twice || ln.mark[replica[c]] == -3;
            ln.mark[replica[c]] = -3;
        }
        for (int c = 0; c < n; ++c) ln.mark[replica[c]] = -1;
        if (twice) return set_error(MGPU_ERR_INVALID_ARG, "trial_decide_submit: more than one candidate for a replica");
    }
    ln.decided_wait_n = 0;
    ln.dirty = true;
    ln.last_trial_built = false;
    int frame_at = 0;
    if (build) {
        int n1_all = 1;
        for (int c = 0; c < n; ++c) {
            if (t[c] < 0 || t[c] >= e->tp.n_res) return set_error(MGPU_ERR_INVALID_ARG, "trial_submit: residue type out of range");
            n1_all = std::max(n1_all, e->tp.n1[t[c]]);
        }
        frame_at = n1_all;
        site_stride = 2 * n1_all + 1;
    }
    // from here on the rows of the lane's previous trial are gone (the staging block below may be regrown and is
    // overwritten): a failed submit must not leave them committable "from the lane's resident rows"
    ln.last_trial_n = 0;
    ln.d_trial_items = nullptr;
    ln.h_trial_items = nullptr;
    const size_t site_bytes = (size_t)n * site_stride * 3 * sizeof(double);
    const size_t pit_cap = 2 * (size_t)n * sizeof(PairItem), rit_bytes = (size_t)n * sizeof(RecipItem);
    const size_t iit_cap = (size_t)n * sizeof(PairItem);       // intra items
    // device-built trials append [move codes (n ints) | uniforms (5 n doubles)] behind everything else, 8-byte aligned
    const size_t build_at = (site_bytes + pit_cap + rit_bytes + iit_cap + 7) & ~(size_t)7;
    const size_t build_mv = ((size_t)n * sizeof(int) + 7) & ~(size_t)7;
    const size_t build_bytes = build ? build_mv + (size_t)5 * n * sizeof(double) : 0;
    // ... and the deciding form its DecideItems behind that
    const size_t dec_at = (build_at + build_bytes + 7) & ~(size_t)7;
    const size_t dec_bytes = decide ? (size_t)n * sizeof(DecideItem) : 0;
    if (sites && sites == ln.h_in.p && site_bytes + pit_cap + rit_bytes + iit_cap > ln.h_in.bytes)
        return set_error(MGPU_ERR_INVALID_ARG, "trial_submit: more candidates than the lane's site buffer was sized for");
    if (sites && sites == ln.h_in.p && dec_at + dec_bytes > ln.h_in.bytes)
        return set_error(MGPU_ERR_INVALID_ARG, "trial_decide_submit: the lane's site buffer is too small for the acceptance records "
                                               "(mgpu_lane_site_buffer sizes it for them)");
    // a block lent to the caller is never regrown behind their back (they keep the pointer for the farm's lifetime)
    if (ln.h_in_lent && dec_at + dec_bytes > ln.h_in.bytes)
        return set_error(MGPU_ERR_STATE, "trial_submit: this trial needs a larger staging block than the one lent out by "
                                         "mgpu_lane_site_buffer; call it again with the larger size first");
    if ((rc = ln.h_in.reserve(dec_at + dec_bytes))) return rc;
    double *h_sites = (double *)ln.h_in.p;
    PairItem *pit = (PairItem *)((char *)ln.h_in.p + site_bytes);
    RecipItem *rit = (RecipItem *)((char *)ln.h_in.p + site_bytes + pit_cap);
    PairItem *iit = (PairItem *)((char *)ln.h_in.p + site_bytes + pit_cap + rit_bytes);
    ln.pair_old.assign(n, -1);
    ln.pair_new.assign(n, -1);
    ln.intra_idx.assign(n, -1);
    ln.kinds.assign(n, MGPU_MOVE);
    ln.self_of.assign(n, 0.0);
    int n1_max = 1, n_intra = 0;
    // Candidates are grouped by residue type: every type gets its own pair-sweep launches with the register-site kernels
    // of its size (a mixture of a 3-site and a 2-site species used to fall to the generic NS = 0 sweep for the whole
    // launch), and which kernels a type's candidates take never depends on what else shares the launch.  Within a type,
    // trial moves of molecules with a few sites are swept old + new together (fused items, two entries each);
    // insertions, deletions and everything else are single-state items.
    struct Seg { int n1, fused, first_item, n_items, first_entry, type, nsplit, first_partial; };
    int cls_n1[kMaxRes], cls_moves[kMaxRes], cls_single[kMaxRes], cls_type[kMaxRes], n_cls = 0;
    for (int c = 0; c < n; ++c) {
        const int k = kind ? kind[c] : MGPU_MOVE;
        if (k < MGPU_MOVE || k > MGPU_DELETION) return set_error(MGPU_ERR_INVALID_ARG, "trial_submit: unknown candidate kind");
        if (t[c] < 0 || t[c] >= e->tp.n_res) return set_error(MGPU_ERR_INVALID_ARG, "trial_submit: residue type out of range");
        const int n1 = e->tp.n1[t[c]];
        int ci = 0;
        while (ci < n_cls && cls_type[ci] != t[c]) ++ci;
        if (ci == n_cls) { cls_n1[ci] = n1; cls_moves[ci] = 0; cls_single[ci] = 0; cls_type[ci] = t[c]; ++n_cls; }   // <= n_res classes
        // (triclinic boxes: two single-state items -- the image search's registers leave no room for 2 NS sites without spills;
        //  measured round 5, 10 125-atom box, 1024 moves: fused 267-391 us, two single-state sweeps 239 us)
        const bool fz = k == MGPU_MOVE && n1 <= kMaxFusedSites && !e->bx.triclinic;
        if (fz) cls_moves[ci] += 1;
        else cls_single[ci] += (k == MGPU_MOVE) ? 2 : 1;
    }
    // Framework boxes: one frozen residue type, identical in every replica, flat kernels in use, an active residue type of
    // <= 5 sites -> the type's items go to pair_frozen_kernel (candidates in the lanes; framework atoms as scalars, then the
    // replica's few other atoms per lane); their sums arrive as ONE extra record per entry behind the other results
    int t_frozen = -1;
    if (e->pair_flat && e->frozen_batch && !e->bx.triclinic) {
        int nf = 0;
        for (int tt = 0; tt < e->tp.n_res; ++tt)
            if (e->frozen[tt]) { ++nf; t_frozen = tt; }
        if (nf != 1 || e->frozen_diff[t_frozen] != 0 || e->h_nmol[t_frozen] < 1) t_frozen = -1;
    }
    auto type_batched = [&](int ty, int n1) { return t_frozen >= 0 && ty != t_frozen && n1 <= kMaxFusedSitesWide; };
    const int n_atoms_f = t_frozen >= 0 ? e->h_nmol[t_frozen] * e->tp.n1[t_frozen] : 0;
    const int n_chunks_f = t_frozen >= 0 ? (n_atoms_f + frozen_chunk_atoms(e, n_atoms_f) - 1) / frozen_chunk_atoms(e, n_atoms_f) : 0;
    const int nsplit_engine = e->pair_nsplit;
    std::vector<Seg> segs;
    int seg_fused[kMaxRes], seg_single[kMaxRes];        // per class: index of its fused / single segment (-1: none)
    int n_items_total = 0, n_pair = 0, n_fused = 0, n_partials = 0;
    for (int ci = 0; ci < n_cls; ++ci) {
        seg_fused[ci] = seg_single[ci] = -1;
        const int ns_seg = type_batched(cls_type[ci], cls_n1[ci]) ? 0 : nsplit_engine;      // batched: the extra record is all
        if (cls_moves[ci]) {
            seg_fused[ci] = (int)segs.size();
            segs.push_back(Seg{cls_n1[ci], 1, n_items_total, 0, n_pair, cls_type[ci], ns_seg, n_partials});
            n_items_total += cls_moves[ci];
            n_pair += 2 * cls_moves[ci];
            n_partials += 2 * cls_moves[ci] * ns_seg;
            n_fused += cls_moves[ci];
        }
        if (cls_single[ci]) {
            seg_single[ci] = (int)segs.size();
            segs.push_back(Seg{cls_n1[ci], 0, n_items_total, 0, n_pair, cls_type[ci], ns_seg, n_partials});
            n_items_total += cls_single[ci];
            n_pair += cls_single[ci];
            n_partials += cls_single[ci] * ns_seg;
        }
    }
    ln.ent_off.assign(n_pair, 0);
    ln.ent_stride.assign(n_pair, 2);
    ln.ent_ns.assign(n_pair, 1);
    auto put_item = [&](const Seg &sg, int i, const PairItem &it) {     // item i of the segment; returns its first entry
        pit[sg.first_item + i] = it;
        const int e0 = sg.first_entry + (sg.fused ? 2 * i : i);
        // partial records (double2) of the segment start at first_partial; [split][state] for fused items
        if (sg.fused) {
            ln.ent_off[e0] = 2 * (sg.first_partial + 2 * i * sg.nsplit);
            ln.ent_off[e0 + 1] = ln.ent_off[e0] + 2;
            ln.ent_stride[e0] = ln.ent_stride[e0 + 1] = 4;
            ln.ent_ns[e0] = ln.ent_ns[e0 + 1] = sg.nsplit;
        } else {
            ln.ent_off[e0] = 2 * (sg.first_partial + i * sg.nsplit);
            ln.ent_ns[e0] = sg.nsplit;
        }
        return e0;
    };
    bool fast = true;                 // all replicas of this trial within the fast fold's range
    ln.cand_ok.assign(n, 1);          // and per candidate: would committing it keep its replica there
    // Two passes over the candidates, each cut into ranges run side by side (for_parts): the first validates a candidate,
    // fills what belongs to it alone and counts the items it will add to its class's segments; the second, knowing every
    // range's first item in every segment, writes the items -- in candidate order within a segment, as one loop would.
    struct Part {
        int n1_max = 1, n_intra = 0;
        bool fast = true;
        int n_fused[kMaxRes], n_single[kMaxRes];
        int at_fused[kMaxRes], at_single[kMaxRes], at_intra = 0;
    };
    const int parts = host_parts(e, n);
    Part part_of[kMaxHostParts];
    PartError errs[kMaxHostParts];
    auto class_of = [&](int ty) { int ci = 0; while (cls_type[ci] != ty) ++ci; return ci; };
    // candidate c's pair and intra items at the places the running indices say
    auto place = [&](int c, int k, int mc, int ci, int *i_f, int *i_s, int &i_intra) {
        if (k == MGPU_MOVE && seg_fused[ci] >= 0) {
            const int e0 = put_item(segs[seg_fused[ci]], i_f[ci]++, PairItem{replica[c], t[c], mc, c, 0});
            ln.pair_old[c] = e0; ln.pair_new[c] = e0 + 1;
        } else {
            const Seg &sg = segs[seg_single[ci]];
            if (k != MGPU_CREATION) ln.pair_old[c] = put_item(sg, i_s[ci]++, PairItem{replica[c], t[c], mc, -1, 0});
            if (k != MGPU_DELETION) ln.pair_new[c] = put_item(sg, i_s[ci]++, PairItem{replica[c], t[c], mc, c, 0});
        }
        if (k == MGPU_CREATION) { ln.intra_idx[c] = i_intra; iit[i_intra++] = PairItem{replica[c], t[c], -1, c, 0}; }
        if (k == MGPU_DELETION) { ln.intra_idx[c] = i_intra; iit[i_intra++] = PairItem{replica[c], t[c], mc, -1, 0}; }
    };
    for_parts(parts, [&](int q) {
        Part &P = part_of[q];
        for (int ci = 0; ci < n_cls; ++ci) P.n_fused[ci] = P.n_single[ci] = 0;
        int c0, c1;
        part_range(n, parts, q, c0, c1);
        for (int c = c0; c < c1; ++c) {
            const int k = kind ? kind[c] : MGPU_MOVE;
            const int mc = (k == MGPU_CREATION) ? -1 : m[c];
            if (const int r = check_candidate(e, c, replica[c], t[c], mc, k != MGPU_CREATION)) { errs[q].set(c, r, mgpu_last_error()); return; }
            const int n1 = e->tp.n1[t[c]];
            if (n1 > site_stride) { errs[q].set(c, MGPU_ERR_INVALID_ARG, "site_stride smaller than atoms_in_res"); return; }
            P.n1_max = std::max(P.n1_max, n1);
            const int ci = class_of(t[c]);
            ln.kinds[c] = k;
            P.fast = P.fast && replica_in_range(e, replica[c]);
            if (build) {
                const size_t idx = (size_t)replica[c] * e->tp.n_res + t[c];
                if (!e->d_com || !e->frames_ok[idx]) {
                    errs[q].set(c, MGPU_ERR_STATE, "move_trial_submit: no molecule frames for candidate " + std::to_string(c) +
                                                       " (mgpu_replica_set_frames)");
                    return;
                }
                const int mv = build->move[c];
                if (mv < 1 || mv > 4 || (k == MGPU_MOVE) != (mv <= 2) || (k == MGPU_CREATION) != (mv == 3)) {
                    errs[q].set(c, MGPU_ERR_INVALID_ARG, "move_trial_submit: move code does not match the candidate kind");
                    return;
                }
                // a built candidate's centre lies in the cell (ApplyPBC / uniform insertion); with tight frames its sites are
                // within the fast fold's range
                if (k != MGPU_DELETION) { ln.cand_ok[c] = e->frames_tight[idx]; P.fast = P.fast && ln.cand_ok[c]; }
            } else if (k != MGPU_DELETION) {
                ln.cand_ok[c] = sites_in_range(e, sites + (size_t)c * site_stride * 3, n1) ? 1 : 0;
                P.fast = P.fast && ln.cand_ok[c];          // the candidate's own sites are swept in this launch
            }
            if (k != MGPU_MOVE) ln.self_of[c] = e->self_of_type[t[c]];
            if (decide) {
                const size_t idx = (size_t)replica[c] * e->tp.n_res + t[c];
                if (k == MGPU_CREATION && e->h_nmol[idx] >= e->tp.cap[t[c]]) {
                    errs[q].set(c, MGPU_ERR_CAPACITY, "trial_decide_submit: residue type is at mol_capacity");
                    return;
                }
                if (!build && k != MGPU_DELETION && e->d_com && e->frames_ok[idx]) {
                    errs[q].set(c, MGPU_ERR_STATE, "trial_decide_submit: this replica holds molecule frames: submit device-built trials");
                    return;
                }
            }
            rit[c] = RecipItem{replica[c], t[c], mc, k, k == MGPU_DELETION ? -1 : c, 0, frame_at};   // one k sweep: old and new
            if (parts == 1) {            // one range: its counters ARE the items' places, no second pass
                place(c, k, mc, ci, P.n_fused, P.n_single, P.n_intra);
                continue;
            }
            if (k == MGPU_MOVE && seg_fused[ci] >= 0) P.n_fused[ci] += 1;
            else P.n_single[ci] += (k != MGPU_CREATION) + (k != MGPU_DELETION);
            if (k != MGPU_MOVE) P.n_intra += 1;
        }
    });
    if ((rc = report_first(errs, parts))) return rc;
    {
        int run_f[kMaxRes] = {0}, run_s[kMaxRes] = {0};
        for (int q = 0; q < parts; ++q) {
            Part &P = part_of[q];
            n1_max = std::max(n1_max, P.n1_max);
            fast = fast && P.fast;
            P.at_intra = n_intra;
            n_intra += P.n_intra;
            for (int ci = 0; ci < n_cls; ++ci) {
                P.at_fused[ci] = run_f[ci]; run_f[ci] += P.n_fused[ci];
                P.at_single[ci] = run_s[ci]; run_s[ci] += P.n_single[ci];
            }
        }
        for (int ci = 0; ci < n_cls; ++ci) {
            if (seg_fused[ci] >= 0) segs[seg_fused[ci]].n_items = run_f[ci];
            if (seg_single[ci] >= 0) segs[seg_single[ci]].n_items = run_s[ci];
        }
    }
    if (parts > 1)
        for_parts(parts, [&](int q) {
            const Part &P = part_of[q];
            int i_f[kMaxRes], i_s[kMaxRes], i_intra = P.at_intra;
            for (int ci = 0; ci < n_cls; ++ci) { i_f[ci] = P.at_fused[ci]; i_s[ci] = P.at_single[ci]; }
            int c0, c1;
            part_range(n, parts, q, c0, c1);
            for (int c = c0; c < c1; ++c) {
                const int k = ln.kinds[c];
                place(c, k, (k == MGPU_CREATION) ? -1 : m[c], class_of(t[c]), i_f, i_s, i_intra);
            }
        });
    if (build) {
        std::memcpy((char *)ln.h_in.p + build_at, build->move, (size_t)n * sizeof(int));
        std::memcpy((char *)ln.h_in.p + build_at + build_mv, build->u, (size_t)5 * n * sizeof(double));
    } else {
        if (sites != h_sites) std::memcpy(h_sites, sites, site_bytes);    // rows built in place (mgpu_lane_site_buffer): no copy
        if (any_frozen(e, n, t)) permute_frozen_rows(e, h_sites, n, site_stride, t);
    }
    const size_t iit_bytes = (size_t)n_intra * sizeof(PairItem);
    // results in device memory, copied out once: [split partials of the pair sweep (n_pair * nsplit complex-sized
    // records, reduced on the host in trial_wait) | u_old | u_new | intra]
    auto seg_batched = [&](const Seg &sg) { return type_batched(sg.type, sg.n1); };
    ln.ent_extra.assign(n_pair, 0);
    size_t scratch_records = 0;
    for (const Seg &sg : segs)
        if (seg_batched(sg)) {
            const int ne = sg.n_items * (sg.fused ? 2 : 1);
            for (int i = 0; i < ne; ++i) ln.ent_extra[sg.first_entry + i] = 1;
            scratch_records += (size_t)ne * n_chunks_f;
        }
    const size_t extra_at = 2 * (size_t)n_partials + 3 * (size_t)n;          // doubles
    const size_t acc_at = extra_at + (scratch_records ? 2 * (size_t)n_pair : 0);     // the deciding form's flags (ints)
    const size_t out_doubles = acc_at + (decide ? ((size_t)n + 1) / 2 : 0);
    if (decide) {
        if (!recip_by_rows(e, n1_max)) return set_error(MGPU_ERR_STATE, "trial_decide_submit: needs the row-form k sweep");
        DecideItem *dit = (DecideItem *)((char *)ln.h_in.p + dec_at);
        for (int c = 0; c < n; ++c) {
            DecideItem d{0, 2, -1, -1, 0, 2, -1, -1, ln.intra_idx[c], ln.kinds[c], ln.self_of[c], decide->pref[c], decide->u[c]};
            if (const int i = ln.pair_old[c]; i >= 0) {
                d.old_off = ln.ent_off[i]; d.old_stride = ln.ent_stride[i]; d.old_ns = ln.ent_ns[i];
                d.old_extra = ln.ent_extra[i] ? (int)(extra_at + 2 * (size_t)i) : -1;
            }
            if (const int i = ln.pair_new[c]; i >= 0) {
                d.new_off = ln.ent_off[i]; d.new_stride = ln.ent_stride[i]; d.new_ns = ln.ent_ns[i];
                d.new_extra = ln.ent_extra[i] ? (int)(extra_at + 2 * (size_t)i) : -1;
            }
            dit[c] = d;
        }
    }
    if (scratch_records && (rc = ln.d_scratch.reserve(scratch_records * sizeof(double2)))) return rc;
    // one staging block [sites | pair items (2n slots) | recip items | intra items] -> one H2D copy
    const size_t in_bytes = site_bytes + pit_cap + rit_bytes + iit_bytes;
    if ((rc = ln.d_sites.reserve(dec_at + dec_bytes))) return rc;
    if ((rc = ln.d_out.reserve(out_doubles * sizeof(double)))) return rc;
    if ((rc = ln.h_out.reserve(out_doubles * sizeof(double)))) return rc;
    if (build) {
        // the rows are written by the device: only [items | move codes | uniforms] travel
        HIP_TRY(hipMemcpyAsync((char *)ln.d_sites.p + site_bytes, (char *)ln.h_in.p + site_bytes, dec_at + dec_bytes - site_bytes,
                               hipMemcpyHostToDevice, ln.stream));
        hipLaunchKernelGGL(trial_build_kernel, dim3((n + 127) / 128), dim3(128), 0, ln.stream, e->tp, e->bx,
                           (const RecipItem *)((char *)ln.d_sites.p + site_bytes + pit_cap), (const int *)((char *)ln.d_sites.p + build_at),
                           (const double *)((char *)ln.d_sites.p + build_at + build_mv), build->t_step, build->r_step,
                           (double *)ln.d_sites.p, site_stride, frame_at, n);
        HIP_TRY(hipGetLastError());
    } else {
        HIP_TRY(hipMemcpyAsync(ln.d_sites.p, ln.h_in.p, in_bytes, hipMemcpyHostToDevice, ln.stream));
        if (decide)
            HIP_TRY(hipMemcpyAsync((char *)ln.d_sites.p + dec_at, (char *)ln.h_in.p + dec_at, dec_bytes, hipMemcpyHostToDevice, ln.stream));
    }
    const PairItem *d_pit = (const PairItem *)((char *)ln.d_sites.p + site_bytes);
    const RecipItem *d_rit = (const RecipItem *)((char *)ln.d_sites.p + site_bytes + pit_cap);
    const PairItem *d_iit = (const PairItem *)((char *)ln.d_sites.p + site_bytes + pit_cap + rit_bytes);
    double2 *d_part = (double2 *)ln.d_out.p;
    double *d_uo = (double *)ln.d_out.p + 2 * (size_t)n_partials, *d_un = d_uo + n, *d_in = d_un + n;
    // Kernel order: pair sweep first, k sweep second (the order the stand-alone commit of the other lanes overlaps best
    // with; k sweep first was measured 10 % slower there).
    size_t scratch_at = 0;
    for (const Seg &sg : segs) {
        const bool fb = seg_batched(sg);
        if (!fb && (rc = launch_pair(e, ln, d_pit + sg.first_item, sg.n_items, sg.n1, site_stride, sg.nsplit, nullptr, nullptr, false,
                                     d_part + sg.first_partial, sg.fused != 0, fast)))
            return rc;
        if (fb) {
            if ((rc = launch_frozen(e, ln, d_pit + sg.first_item, sg.n_items, sg.n1, site_stride, sg.fused != 0, fast, t_frozen,
                                    (double2 *)ln.d_scratch.p + scratch_at, (double2 *)((double *)ln.d_out.p + extra_at) + sg.first_entry)))
                return rc;
            scratch_at += (size_t)sg.n_items * (sg.fused ? 2 : 1) * n_chunks_f;
        }
    }
    if (!decide && (rc = launch_recip(e, ln, d_rit, n, n1_max, site_stride, false, e->d_A, d_un, d_uo)))
        return rc;
    if (n_intra && (rc = launch_intra(e, ln, d_iit, n_intra, (const double *)ln.d_sites.p, site_stride, d_in))) return rc;
    if (decide) {
        // the k sweep comes last: its workgroups decide and commit (everything else of the trial has read the old state)
        const DecideArgs da{(const DecideItem *)((const char *)ln.d_sites.p + dec_at), (const double *)ln.d_out.p, d_in,
                            (int *)((double *)ln.d_out.p + acc_at), decide->temperature};
        if ((rc = launch_recip(e, ln, d_rit, n, n1_max, site_stride, false, e->d_A, d_un, d_uo, nullptr, nullptr, &da))) return rc;
        ln.decided_n = n;
        ln.decided_wait_n = n;
        ln.decided_at = acc_at * sizeof(double);
    }
    HIP_TRY(hipMemcpyAsync(ln.h_out.p, ln.d_out.p, out_doubles * sizeof(double), hipMemcpyDeviceToHost, ln.stream));
    ln.n_submitted = n;
    ln.n_pair_items = n_pair;
    ln.n_fused = n_fused;
    ln.n_partials = n_partials;
    ln.last_trial_n = n;
    ln.last_trial_stride = site_stride;
    ln.last_trial_built = build != nullptr;
    ln.last_trial_frame = frame_at;
    ln.d_trial_items = d_rit;
    ln.h_trial_items = rit;
    ln.trial_n1_max = n1_max;
    return MGPU_OK;
}

// ncomp = 3: non_coulomb, coulomb, recip_coulomb; ncomp = 5: + ewald_self, intra_coulomb
static int trial_wait_impl(mgpu_engine *e, Lane &ln, double *old_energy, double *new_energy, int ncomp, int *accepted = nullptr) {
    const int n = ln.n_submitted;
    if (n == 0) return set_error(MGPU_ERR_STATE, "trial_wait: nothing was submitted on this lane");
    // (a drain in between -- mgpu_synchronize or any synchronous entry point -- has already folded the outcomes into the
    // engine's mirrors; the flags are still in the result block)
    if (accepted && ln.decided_wait_n != n) return set_error(MGPU_ERR_STATE, "trial_decide_wait: the lane's trial was not submitted with an acceptance test");
    ln.decided_wait_n = 0;
    ln.n_submitted = 0;
    const size_t flags_at = ln.decided_at;
    int rc = sync_lane(e, ln);
    if (rc) return rc;
    if (accepted) std::memcpy(accepted, (const char *)ln.h_out.p + flags_at, (size_t)n * sizeof(int));
    const int np = ln.n_pair_items;
    const double *h = (const double *)ln.h_out.p;
    const double *uo = h + 2 * (size_t)ln.n_partials, *un = uo + n, *in = un + n, *ex = in + n;
    // the ordered sum of the split partials and the Coulomb rescale e_coulomb * EPS0_INV_eVA / KB_eVK
    // (energy_utils.f90:440), exactly as pair_finalize_kernel does them.  Partials of a fused item are laid out
    // [split][state], those of a single item [split].
    ln.h_lj.resize(np);
    ln.h_cc.resize(np);
    const int team = host_parts(e, n);       // (both loops are independent per entry / per candidate)
#pragma omp parallel for num_threads(team) schedule(static) if (team > 1)
    for (int i = 0; i < np; ++i) {
        double a = 0.0, b = 0.0;
        const double *p = h + ln.ent_off[i];
        const int stride = ln.ent_stride[i], ns = ln.ent_ns[i];
        for (int s2 = 0; s2 < ns; ++s2) { a += p[stride * s2]; b += p[stride * s2 + 1]; }
        if (ln.ent_extra[i]) { a += ex[2 * i]; b += ex[2 * i + 1]; }       // the framework part (pair_frozen_kernel), last
        ln.h_lj[i] = a;
        ln.h_cc[i] = b * kEps0InvEvA / kKbEvK;
    }
    const double *lj = ln.h_lj.data(), *cc = ln.h_cc.data();
#pragma omp parallel for num_threads(team) schedule(static) if (team > 1)
    for (int c = 0; c < n; ++c) {
        double *o = old_energy + (size_t)ncomp * c, *w = new_energy + (size_t)ncomp * c;
        for (int k = 0; k < ncomp; ++k) { o[k] = 0.0; w[k] = 0.0; }
        if (ln.pair_old[c] >= 0) { o[0] = lj[ln.pair_old[c]]; o[1] = cc[ln.pair_old[c]]; }
        if (ln.pair_new[c] >= 0) { w[0] = lj[ln.pair_new[c]]; w[1] = cc[ln.pair_new[c]]; }
        o[2] = uo[c];
        w[2] = un[c];
        if (ncomp == 5) {
            // ewald_self / intra_coulomb enter on the side where the molecule exists
            // (monte_carlo_utils.f90:298-299 creation new, :378-379 deletion old)
            if (ln.kinds[c] == MGPU_CREATION) { w[3] = ln.self_of[c]; w[4] = in[ln.intra_idx[c]]; }
            if (ln.kinds[c] == MGPU_DELETION) { o[3] = ln.self_of[c]; o[4] = in[ln.intra_idx[c]]; }
        }
    }
    return MGPU_OK;
}

// Queue the commit of the accepted candidates on a lane (no synchronisation).  The host-side
// molecule counts are updated immediately; the device applies them in stream order.
// reuse_sites: `sites` may be NULL, meaning "the rows the lane's last trial_submit uploaded" (same
// candidates, same order), which are still resident in the lane's device scratch.
static int commit_submit_impl(mgpu_engine *e, Lane &ln, int n, const int *replica, const int *t, const int *m,
                              const int *kind, const double *sites, int site_stride, const int *accept,
                              bool reuse_sites = false) {
    int rc;
    const size_t site_bytes = sites ? (size_t)n * site_stride * 3 * sizeof(double) : 0;
    if (ln.n_submitted != 0) return set_error(MGPU_ERR_STATE, "commit_submit: wait for the lane's trial first");
    if (e->farm.dirty && (rc = farm_window_normalize(e))) return rc;
    ln.dirty = true;
    // committing a device-built trial from its resident rows: the rows carry the candidates' frames
    const bool built = !sites && reuse_sites && ln.last_trial_built && n == ln.last_trial_n;
    if (built) site_stride = ln.last_trial_stride;
    // the pinned staging block may still feed the H2D copy of the lane's previous commit
    if (ln.commit_staged) {
        HIP_TRY(hipEventSynchronize(ln.commit_staged_ev));
        ln.commit_staged = false;
    }
    if ((rc = ln.h_commit.reserve(site_bytes + (size_t)n * sizeof(RecipItem)))) return rc;
    RecipItem *items = (RecipItem *)((char *)ln.h_commit.p + site_bytes);
    int n_items = 0;
    // one accepted candidate per replica: ln.commit_mark[replica] holds the stamp of the call that last committed there (a
    // fresh stamp per call instead of clearing n_replicas flags; exchanged atomically: the ranges below run side by side)
    if ((int)ln.commit_mark.size() != e->n_replicas) { ln.commit_mark.assign(e->n_replicas, -1); ln.commit_stamp = 0; }
    if (++ln.commit_stamp == 0x7fffffff) { std::fill(ln.commit_mark.begin(), ln.commit_mark.end(), -1); ln.commit_stamp = 1; }
    const int stamp = ln.commit_stamp;
    bool any_sites = false;
    int n1_max = 1;
    // two passes in ranges, as in trial_submit_impl: count the accepted candidates of every range, then validate them and
    // write their items at the range's place -- the items keep candidate order
    struct Part {
        int n_acc = 0, at = 0, n1_max = 1;
        bool any_sites = false;
        std::vector<int> new_counts;  // (index into h_nmol, value) pairs applied after validation
        std::vector<int> range_lost;  // (replica, type) entries whose atoms leave the fast fold's range with this commit
    };
    const int parts = host_parts(e, n);
    Part part_of[kMaxHostParts];
    PartError errs[kMaxHostParts];
    if (parts > 1) {
        for_parts(parts, [&](int q) {
            int c0, c1, k = 0;
            part_range(n, parts, q, c0, c1);
            for (int c = c0; c < c1; ++c) k += accept[c] != 0;
            part_of[q].n_acc = k;
        });
        for (int q = 0; q < parts; ++q) { part_of[q].at = n_items; n_items += part_of[q].n_acc; }
    }
    for_parts(parts, [&](int q) {
        Part &P = part_of[q];
        int c0, c1, at = P.at;
        part_range(n, parts, q, c0, c1);
        for (int c = c0; c < c1; ++c) {
            if (!accept[c]) continue;
            if (kind[c] < MGPU_MOVE || kind[c] > MGPU_DELETION) { errs[q].set(c, MGPU_ERR_INVALID_ARG, "commit: unknown candidate kind"); return; }
            if (replica[c] < 0 || replica[c] >= e->n_replicas) { errs[q].set(c, MGPU_ERR_INVALID_ARG, "commit: replica out of range"); return; }
            int &mark = ln.commit_mark[replica[c]];
            int before;
            if (parts > 1) before = __atomic_exchange_n(&mark, stamp, __ATOMIC_RELAXED);     // (the exchange IS the store)
            else { before = mark; mark = stamp; }
            if (before == stamp) {
                errs[q].set(c, MGPU_ERR_INVALID_ARG, "commit: more than one accepted candidate for a replica");
                return;
            }
            if (t[c] < 0 || t[c] >= e->tp.n_res) { errs[q].set(c, MGPU_ERR_INVALID_ARG, "commit: residue type out of range"); return; }
            const int idx = replica[c] * e->tp.n_res + t[c], nm = e->h_nmol[idx];
            RecipItem it{replica[c], t[c], m[c], kind[c], -1, nm};
            if (kind[c] == MGPU_CREATION) {
                if (nm >= e->tp.cap[t[c]]) { errs[q].set(c, MGPU_ERR_CAPACITY, "commit: residue type is at mol_capacity"); return; }
                it.m = nm;  // appended at the first free slot: num_residues + 1 (monte_carlo.f90:63, create_molecule.f90:64)
                it.aux = nm + 1;
            } else {
                if (const int r = check_candidate(e, c, replica[c], t[c], m[c], true)) { errs[q].set(c, r, mgpu_last_error()); return; }
                if (kind[c] == MGPU_DELETION) it.aux = nm - 1;
            }
            if (kind[c] != MGPU_DELETION) {
                P.any_sites = true;
                it.src = c;
                it.frame = built ? ln.last_trial_frame : 0;
                // where the engine keeps molecule frames they must stay the mirror of the sites: a move / insertion given as
                // bare sites cannot update them
                if (!built && e->d_com && e->frames_ok[idx]) {
                    errs[q].set(c, MGPU_ERR_STATE, "commit: this replica holds molecule frames (mgpu_replica_set_frames): commit "
                                                   "device-built trials from the lane's resident rows, or set the molecules again");
                    return;
                }
                // the accepted sites become resident atoms: keep the replica's range flag honest
                const bool ok = sites ? sites_in_range(e, sites + (size_t)c * site_stride * 3, e->tp.n1[t[c]])
                                      : (c < (int)ln.cand_ok.size() && ln.cand_ok[c]);
                if (!ok) P.range_lost.push_back(idx);
                if (e->tp.n1[t[c]] > site_stride) { errs[q].set(c, MGPU_ERR_INVALID_ARG, "site_stride smaller than atoms_in_res"); return; }
            }
            P.n1_max = std::max(P.n1_max, e->tp.n1[t[c]]);
            if (kind[c] != MGPU_MOVE) { P.new_counts.push_back(idx); P.new_counts.push_back(it.aux); }
            items[at++] = it;
        }
        if (parts == 1) n_items = at;        // (one range: counted as it went)
    });
    if ((rc = report_first(errs, parts))) {
        // (the stamps of this refused call must not make a repeat of it look like a duplicate)
        for (int c = 0; c < n; ++c)
            if (accept[c] && replica[c] >= 0 && replica[c] < e->n_replicas) ln.commit_mark[replica[c]] = -1;
        if (parts > 1) {
            // Two accepted candidates of one replica in DIFFERENT ranges are noticed by whichever range came second in time:
            // which candidate that is depends on the threads.  The serial loop reports the second of the pair in candidate
            // order, and it stops at the first refusal of any kind: scan for a duplicate below the refusal just found.
            int first_bad = n;
            for (int q = 0; q < parts; ++q)
                if (errs[q].c >= 0) first_bad = std::min(first_bad, errs[q].c);
            const int scan = ++ln.commit_stamp;
            int dup = -1;
            for (int c = 0; c < n && c <= first_bad && dup < 0; ++c) {
                if (!accept[c] || replica[c] < 0 || replica[c] >= e->n_replicas) continue;
                if (ln.commit_mark[replica[c]] == scan) dup = c;
                ln.commit_mark[replica[c]] = scan;
            }
            for (int c = 0; c < n; ++c)
                if (accept[c] && replica[c] >= 0 && replica[c] < e->n_replicas) ln.commit_mark[replica[c]] = -1;
            if (dup >= 0 && dup <= first_bad) return set_error(MGPU_ERR_INVALID_ARG, "commit: more than one accepted candidate for a replica");
        }
        return rc;
    }
    for (int q = 0; q < parts; ++q) { any_sites = any_sites || part_of[q].any_sites; n1_max = std::max(n1_max, part_of[q].n1_max); }
    if (n_items == 0) return MGPU_OK;
    if (any_sites && !sites && !reuse_sites) return set_error(MGPU_ERR_INVALID_ARG, "commit_candidates: sites is null");
    // Committing the lane's last trial from its resident rows: the trial's items are still on the device too,
    // so the accept flags travel as a kernel argument and nothing is uploaded.
    if (!sites && reuse_sites && n == ln.last_trial_n && ln.d_trial_items && n <= 32 * kAcceptWords &&
        recip_by_rows(e, ln.trial_n1_max)) {
        AcceptBits bits{};
        bool same_of[kMaxHostParts];   // the caller promises the trial's candidates in the trial's order: verify
        for_parts(parts, [&](int q) {  // (the ranges end on multiples of 32 candidates: a mask word belongs to one range)
            bool same = true;
            int c0, c1;
            part_range(n, parts, q, c0, c1);
            for (int c = c0; c < c1; ++c) {
                if (!accept[c]) continue;
                const RecipItem &ti = ln.h_trial_items[c];
                same = same && ti.replica == replica[c] && ti.t == t[c] && ti.kind == kind[c] &&
                       (kind[c] == MGPU_CREATION || ti.m == m[c]);
                bits.w[c >> 5] |= 1u << (c & 31);
            }
            same_of[q] = same;
        });
        bool same = true;
        for (int q = 0; q < parts; ++q) same = same && same_of[q];
        if (!same) return set_error(MGPU_ERR_INVALID_ARG, "commit_submit: candidates differ from the lane's last trial");
        if ((rc = launch_recip(e, ln, ln.d_trial_items, n, ln.trial_n1_max, site_stride, true, e->d_A, nullptr, nullptr, &bits)))
            return rc;
        // applied once: a second commit_submit(sites = NULL) must not find these rows "resident" again
        ln.last_trial_n = 0;
        ln.d_trial_items = nullptr;
        ln.h_trial_items = nullptr;
    } else {
        if ((rc = ln.d_items2.reserve((size_t)n_items * sizeof(RecipItem)))) return rc;
        HIP_TRY(hipMemcpyAsync(ln.d_items2.p, items, (size_t)n_items * sizeof(RecipItem), hipMemcpyHostToDevice, ln.stream));
        if (any_sites && sites) {
            ln.last_trial_n = 0;
            std::memcpy(ln.h_commit.p, sites, site_bytes);
            if (any_frozen(e, n, t)) permute_frozen_rows(e, (double *)ln.h_commit.p, n, site_stride, t);
            if ((rc = ln.d_sites.reserve(site_bytes))) return rc;
            HIP_TRY(hipMemcpyAsync(ln.d_sites.p, ln.h_commit.p, site_bytes, hipMemcpyHostToDevice, ln.stream));
        }
        if (!ln.commit_staged_ev) HIP_TRY(hipEventCreateWithFlags(&ln.commit_staged_ev, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(ln.commit_staged_ev, ln.stream));
        ln.commit_staged = true;
        if ((rc = launch_recip(e, ln, (const RecipItem *)ln.d_items2.p, n_items, n1_max, site_stride, true, e->d_A, nullptr)))
            return rc;
    }
    for (int q = 0; q < parts; ++q) {
        const std::vector<int> &new_counts = part_of[q].new_counts;
        for (size_t i = 0; i < new_counts.size(); i += 2) e->h_nmol[new_counts[i]] = new_counts[i + 1];
        for (int idx : part_of[q].range_lost) e->in_range[idx] = 0;
    }
    if (e->any_frozen)
        for (int c = 0; c < n; ++c)
            if (accept[c]) frozen_changed(e, replica[c], t[c]);
    return MGPU_OK;
}

static int check_lane(const mgpu_engine *e, int lane) {
    if (!e) return set_error(MGPU_ERR_INVALID_ARG, "null engine");
    if (lane < 0 || lane >= kLanes) return set_error(MGPU_ERR_INVALID_ARG, "lane out of range");
    return MGPU_OK;
}

static size_t trial_staging_bytes(int n, int site_stride) {
    return (size_t)n * site_stride * 3 * sizeof(double) + 2 * (size_t)n * sizeof(PairItem) + (size_t)n * sizeof(RecipItem) +
           (size_t)n * sizeof(PairItem) + 16 + (size_t)n * sizeof(DecideItem);   // + acceptance records
}

int mgpu_lane_site_buffer(mgpu_engine *e, int lane, int n_max, int site_stride, double **sites) {
    int rc = check_lane(e, lane);
    if (rc) return rc;
    if (n_max < 1 || site_stride < 1 || !sites) return set_error(MGPU_ERR_INVALID_ARG, "lane_site_buffer: bad argument");
    if ((rc = use_device(e))) return rc;
    Lane &ln = e->lanes[lane];
    if (ln.n_submitted != 0) return set_error(MGPU_ERR_STATE, "lane_site_buffer: the lane holds an un-waited trial");
    // a regrown block would leave the previous trial's item image dangling
    ln.last_trial_n = 0;
    ln.d_trial_items = nullptr;
    ln.h_trial_items = nullptr;
    // sized for the largest trial shape the lane accepts for n_max candidates: host rows of site_stride sites, or
    // device-built rows [sites | com | offsets] with their move codes and uniform numbers, acceptance records included
    const size_t built = trial_staging_bytes(n_max, 2 * site_stride + 1) + ((size_t)n_max * sizeof(int) + 8) + (size_t)5 * n_max * sizeof(double) + 16;
    if ((rc = ln.h_in.reserve(std::max(trial_staging_bytes(n_max, site_stride), built)))) return rc;
    ln.h_in_lent = true;
    *sites = (double *)ln.h_in.p;
    return MGPU_OK;
}

int mgpu_trial_submit(mgpu_engine *e, int lane, int n, const int *replica, const int *t, const int *m,
                      const double *sites, int site_stride) {
    int rc = check_lane(e, lane);
    if (rc) return rc;
    if (n <= 0 || !replica || !t || !m || !sites) return set_error(MGPU_ERR_INVALID_ARG, "trial_submit: bad argument");
    if ((rc = use_device(e))) return rc;
    return trial_submit_impl(e, e->lanes[lane], n, replica, t, m, nullptr, sites, site_stride);
}

int mgpu_gcmc_trial_submit(mgpu_engine *e, int lane, int n, const int *replica, const int *t, const int *m,
                           const int *kind, const double *sites, int site_stride) {
    int rc = check_lane(e, lane);
    if (rc) return rc;
    if (n <= 0 || !replica || !t || !m || !kind || !sites) return set_error(MGPU_ERR_INVALID_ARG, "gcmc_trial_submit: bad argument");
    if ((rc = use_device(e))) return rc;
    return trial_submit_impl(e, e->lanes[lane], n, replica, t, m, kind, sites, site_stride);
}

int mgpu_move_trial_submit(mgpu_engine *e, int lane, int n, const int *replica, const int *t, const int *m, const int *move,
                           const double *u, double translation_step, double rotation_step) {
    int rc = check_lane(e, lane);
    if (rc) return rc;
    if (n <= 0 || !replica || !t || !m || !move || !u) return set_error(MGPU_ERR_INVALID_ARG, "move_trial_submit: bad argument");
    if (e->bx.triclinic) return set_error(MGPU_ERR_STATE, "move_trial_submit: orthorhombic boxes only");
    if ((rc = use_device(e))) return rc;
    Lane &ln = e->lanes[lane];
    ln.build_kind.resize(n);
    for (int c = 0; c < n; ++c) {
        if (move[c] < 1 || move[c] > 4) return set_error(MGPU_ERR_INVALID_ARG, "move_trial_submit: unknown move code");
        ln.build_kind[c] = move[c] <= 2 ? MGPU_MOVE : (move[c] == 3 ? MGPU_CREATION : MGPU_DELETION);
    }
    const TrialBuild build{move, u, translation_step, rotation_step};
    return trial_submit_impl(e, ln, n, replica, t, m, ln.build_kind.data(), nullptr, 0, &build);
}

int mgpu_move_trial_decide_submit(mgpu_engine *e, int lane, int n, const int *replica, const int *t, const int *m, const int *move,
                                  const double *u, double translation_step, double rotation_step, const double *accept_u,
                                  const double *accept_pref, double temperature) {
    int rc = check_lane(e, lane);
    if (rc) return rc;
    if (n <= 0 || !replica || !t || !m || !move || !u || !accept_u || !accept_pref)
        return set_error(MGPU_ERR_INVALID_ARG, "move_trial_decide_submit: bad argument");
    if (e->bx.triclinic) return set_error(MGPU_ERR_STATE, "move_trial_decide_submit: orthorhombic boxes only");
    if ((rc = use_device(e))) return rc;
    Lane &ln = e->lanes[lane];
    ln.build_kind.resize(n);
    for (int c = 0; c < n; ++c) {
        if (move[c] < 1 || move[c] > 4) return set_error(MGPU_ERR_INVALID_ARG, "move_trial_decide_submit: unknown move code");
        ln.build_kind[c] = move[c] <= 2 ? MGPU_MOVE : (move[c] == 3 ? MGPU_CREATION : MGPU_DELETION);
    }
    const TrialBuild build{move, u, translation_step, rotation_step};
    const TrialDecide dec{accept_u, accept_pref, temperature};
    return trial_submit_impl(e, ln, n, replica, t, m, ln.build_kind.data(), nullptr, 0, &build, &dec);
}

int mgpu_gcmc_trial_decide_submit(mgpu_engine *e, int lane, int n, const int *replica, const int *t, const int *m,
                                  const int *kind, const double *sites, int site_stride, const double *accept_u,
                                  const double *accept_pref, double temperature) {
    int rc = check_lane(e, lane);
    if (rc) return rc;
    if (n <= 0 || !replica || !t || !m || !kind || !sites || !accept_u || !accept_pref)
        return set_error(MGPU_ERR_INVALID_ARG, "gcmc_trial_decide_submit: bad argument");
    if ((rc = use_device(e))) return rc;
    const TrialDecide dec{accept_u, accept_pref, temperature};
    return trial_submit_impl(e, e->lanes[lane], n, replica, t, m, kind, sites, site_stride, nullptr, &dec);
}

int mgpu_trial_decide_wait(mgpu_engine *e, int lane, double *old_energy, double *new_energy, int *accepted) {
    int rc = check_lane(e, lane);
    if (rc) return rc;
    if (!old_energy || !new_energy || !accepted) return set_error(MGPU_ERR_INVALID_ARG, "trial_decide_wait: null output");
    if ((rc = use_device(e))) return rc;
    return trial_wait_impl(e, e->lanes[lane], old_energy, new_energy, 5, accepted);
}

int mgpu_gcmc_trial_wait(mgpu_engine *e, int lane, double *old_energy, double *new_energy) {
    int rc = check_lane(e, lane);
    if (rc) return rc;
    if (!old_energy || !new_energy) return set_error(MGPU_ERR_INVALID_ARG, "gcmc_trial_wait: null output");
    if ((rc = use_device(e))) return rc;
    return trial_wait_impl(e, e->lanes[lane], old_energy, new_energy, 5);
}

int mgpu_trial_wait(mgpu_engine *e, int lane, double *old_energy, double *new_energy) {
    int rc = check_lane(e, lane);
    if (rc) return rc;
    if (!old_energy || !new_energy) return set_error(MGPU_ERR_INVALID_ARG, "trial_wait: null output");
    if ((rc = use_device(e))) return rc;
    return trial_wait_impl(e, e->lanes[lane], old_energy, new_energy, 3);
}

int mgpu_commit_submit(mgpu_engine *e, int lane, int n, const int *replica, const int *t, const int *m, const int *kind,
                       const double *sites, int site_stride, const int *accept) {
    int rc = check_lane(e, lane);
    if (rc) return rc;
    if (n == 0) return MGPU_OK;
    if (n < 0 || !replica || !t || !m || !kind || !accept) return set_error(MGPU_ERR_INVALID_ARG, "commit_submit: bad argument");
    if ((rc = use_device(e))) return rc;
    Lane &ln = e->lanes[lane];
    const bool reuse = (sites == nullptr) && ln.last_trial_n == n && (ln.last_trial_stride == site_stride || ln.last_trial_built);
    return commit_submit_impl(e, ln, n, replica, t, m, kind, sites, site_stride, accept, reuse);
}

int mgpu_trial_energy_candidates(mgpu_engine *e, int n, const int *replica, const int *t, const int *m,
                                 const double *sites, int site_stride, double *old_energy, double *new_energy) {
    if (!e) return set_error(MGPU_ERR_INVALID_ARG, "null engine");
    if (n == 0) return MGPU_OK;
    if (n < 0 || !replica || !t || !m || !sites || !old_energy || !new_energy)
        return set_error(MGPU_ERR_INVALID_ARG, "trial_energy_candidates: bad argument");
    int rc = use_device(e);
    if (rc) return rc;
    if ((rc = sync_all_lanes(e))) return rc;
    if ((rc = trial_submit_impl(e, e->lanes[0], n, replica, t, m, nullptr, sites, site_stride))) return rc;
    return trial_wait_impl(e, e->lanes[0], old_energy, new_energy, 3);
}

int mgpu_commit_candidates(mgpu_engine *e, int n, const int *replica, const int *t, const int *m, const int *kind,
                           const double *sites, int site_stride, const int *accept) {
    if (!e) return set_error(MGPU_ERR_INVALID_ARG, "null engine");
    if (n == 0) return MGPU_OK;
    if (n < 0 || !replica || !t || !m || !kind || !accept) return set_error(MGPU_ERR_INVALID_ARG, "commit_candidates: bad argument");
    int rc = use_device(e);
    if (rc) return rc;
    if ((rc = sync_all_lanes(e))) return rc;
    if ((rc = commit_submit_impl(e, e->lanes[0], n, replica, t, m, kind, sites, site_stride, accept))) return rc;
    return sync_stream(e);
}


int mgpu_set_host_team(mgpu_engine *e, int n_threads) {
    if (!e || n_threads < 1) return set_error(MGPU_ERR_INVALID_ARG, "set_host_team: bad argument");
    e->host_team = std::min(n_threads, kMaxHostParts);
    return MGPU_OK;
}


}  // extern "C"
