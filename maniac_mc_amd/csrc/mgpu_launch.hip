// The kernel launches of the engine: pair sweeps (plane by plane, flat, frozen), the reciprocal update (row form, per k),
// the intra-molecular sum and S(k).  Every launch goes to a lane's stream; nothing here synchronises.
#include "mgpu_engine.h"

namespace mgpu {

template <auto Kernel>
int resident_blocks(Lane &ln, size_t dyn_lds) {
    const void *key = (const void *)Kernel;
    for (const auto &o : ln.occ)
        if (o.kernel == key && o.lds == dyn_lds) return o.blocks;
    int v = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, Kernel, kPairBlock, dyn_lds) != hipSuccess || v < 1) v = 1;
    ln.occ.push_back({key, dyn_lds, std::min(v, 4)});
    return ln.occ.back().blocks;
}

// launch the pair sweep + finalize for items already on the device; results land in d_lj / d_c.
// common_n1 = number of sites when every item has the same count (register path for <= 4), else 0.
// host_partials != nullptr: the split partials are written there and NOT reduced on the device (the caller
// copies them out with its results and adds them up in the same order on the host: one launch and one
// inter-kernel gap less per batch; d_lj / d_c are unused).
int launch_pair(mgpu_engine *e, Lane &ln, const PairItem *d_items, int n_items, int common_n1, int site_stride,
                int nsplit, double *d_lj, double *d_c, bool ordered, double2 *host_partials, bool fused, bool fast_fold,
                bool skip_frozen) {
    const int n_work = n_items * nsplit;
    int rc = MGPU_OK;
    if (fused && (!host_partials || ordered || e->bx.triclinic || common_n1 < 1 || common_n1 > kMaxFusedSites))
        return set_error(MGPU_ERR_STATE, "launch_pair: fused sweep needs register sites, an orthorhombic box and a partials buffer");
    if (!host_partials && (rc = ln.d_partials.reserve((size_t)n_work * sizeof(double2)))) return rc;
    double2 *d_part = host_partials ? host_partials : (double2 *)ln.d_partials.p;
    // persistent waves: 2 workgroups of 8 waves per CU (VGPRs: 4 waves per SIMD at <= 128), never more
    // workgroups than there is work for
    const int per_cu = e->pair_blocks_per_cu;
    const int grid = std::max(1, std::min((n_work + kPairWaves - 1) / kPairWaves, e->n_cu * per_cu));
    hipEvent_t a = nullptr, b = nullptr;
    rc = prof_begin(e, ln, MGPU_KERNEL_PAIR, &a, &b);
    if (rc) return rc;
#define MGPU_LAUNCH_PAIR(NS, ORD, TRI, ...)                                                                             \
    hipExtLaunchKernelGGL((pair_sweep_kernel<NS, ORD, TRI, ##__VA_ARGS__>), dim3(grid), dim3(kPairBlock), e->coul_bytes, ln.stream, \
                          a, b, 0, e->tp, e->bx, e->d_pos, e->d_nmol, e->d_res_q, e->d_res_atype, e->d_pair_tab, e->d_coul_tab,     \
                       d_items, (const double *)ln.d_sites.p, site_stride, nsplit, n_work, d_part)
    // fast_fold: every atom of the replicas involved lies within one box length of the cell centre (tracked on the
    // host), so the register-site kernels may fold separations with two instructions per axis (image_r2_fast)
    const bool ff = fast_fold && !ordered && !e->bx.triclinic && e->pair_fast_fold;
#define MGPU_PAIR_FF(NS, FU)                                                                      \
    do {                                                                                          \
        if (ff) MGPU_LAUNCH_PAIR(NS, false, false, FU, true);                                     \
        else MGPU_LAUNCH_PAIR(NS, false, false, FU, false);                                       \
    } while (0)
    // flat kernels: as many workgroups per CU as their registers and the LDS tables allow
#define MGPU_LAUNCH_FLAT_1(NS, FU, FW)                                                                                  \
    do {                                                                                                               \
        const int nb = resident_blocks<&pair_flat_kernel<NS, FU, FW>>(ln, e->coul_bytes);                                  \
        const int grid_f = std::max(1, std::min((n_work + kPairWaves - 1) / kPairWaves, e->n_cu * nb));               \
        hipExtLaunchKernelGGL((pair_flat_kernel<NS, FU, FW>), dim3(grid_f), dim3(kPairBlock), e->coul_bytes, ln.stream, a, b, 0, \
                              e->tp, e->bx, e->d_pos, e->d_nmol, e->d_res_q, e->d_res_atype, e->d_pair_tab, e->d_coul_tab,   \
                              d_items, (const double *)ln.d_sites.p, site_stride, nsplit, n_work, d_part, skip_frozen ? 1 : 0); \
    } while (0)
#define MGPU_LAUNCH_FLAT(NS, FU)                                                                                        \
    do {                                                                                                               \
        if (ff) MGPU_LAUNCH_FLAT_1(NS, FU, true);                                                                      \
        else MGPU_LAUNCH_FLAT_1(NS, FU, false);                                                                        \
    } while (0)
    const bool flat = e->pair_flat && !ordered && !e->bx.triclinic && common_n1 >= 1 && common_n1 <= kMaxFusedSitesWide;
    if (flat && fused) {
        switch (common_n1) {
            case 1: MGPU_LAUNCH_FLAT(1, true); break;
            case 2: MGPU_LAUNCH_FLAT(2, true); break;
            default: MGPU_LAUNCH_FLAT(3, true); break;      // (fused items have at most kMaxFusedSites sites: checked above)
        }
    } else if (flat) {
        switch (common_n1) {
            case 1: MGPU_LAUNCH_FLAT(1, false); break;
            case 2: MGPU_LAUNCH_FLAT(2, false); break;
            case 3: MGPU_LAUNCH_FLAT(3, false); break;
            case 4: MGPU_LAUNCH_FLAT(4, false); break;
            default: MGPU_LAUNCH_FLAT(5, false); break;
        }
    } else if (fused) {
        switch (common_n1) {
            case 1: MGPU_PAIR_FF(1, true); break;
            case 2: MGPU_PAIR_FF(2, true); break;
            default: MGPU_PAIR_FF(3, true); break;
        }
    } else if (e->bx.triclinic) {
        // triclinic boxes, round 5: the register-site sweeps with ComputeDistance's image search (image_r2_tri_lower / the full
        // 27); a move is two single-state items (trial_submit_impl)
        if (ordered) MGPU_LAUNCH_PAIR(0, true, true);
        else switch (common_n1) {
            case 1: MGPU_LAUNCH_PAIR(1, false, true); break;
            case 2: MGPU_LAUNCH_PAIR(2, false, true); break;
            case 3: MGPU_LAUNCH_PAIR(3, false, true); break;
            case 4: MGPU_LAUNCH_PAIR(4, false, true); break;
            case 5: MGPU_LAUNCH_PAIR(5, false, true); break;
            default: MGPU_LAUNCH_PAIR(0, false, true); break;
        }
    } else if (ordered) {
        MGPU_LAUNCH_PAIR(0, true, false);
    } else {
        switch (common_n1) {
            case 1: MGPU_PAIR_FF(1, false); break;
            case 2: MGPU_PAIR_FF(2, false); break;
            case 3: MGPU_PAIR_FF(3, false); break;
            case 4: MGPU_PAIR_FF(4, false); break;
            case 5: MGPU_PAIR_FF(5, false); break;
            default: MGPU_LAUNCH_PAIR(0, false, false); break;
        }
    }
#undef MGPU_PAIR_FF
#undef MGPU_LAUNCH_FLAT
#undef MGPU_LAUNCH_FLAT_1
#undef MGPU_LAUNCH_PAIR
    rc = prof_end(e, ln, MGPU_KERNEL_PAIR, a, b);
    if (rc) return rc;
    // (The reduction stays a separate launch: letting the last wave of an item reduce the partials needs
    //  agent-scope fences, and on the 8-XCD part those write back / invalidate the XCD's L2 -- measured:
    //  pair sweep 110 -> 275 us.  Likewise results are copied out once rather than stored by the kernels
    //  into pinned host memory: thousands of 8-byte PCIe writes were 3-7x slower than the blit.)
    if (!host_partials)
        hipLaunchKernelGGL(pair_finalize_kernel, dim3((n_items + 255) / 256), dim3(256), 0, ln.stream,
                           (const double2 *)ln.d_partials.p, n_items, nsplit, d_lj, d_c);
    HIP_TRY(hipGetLastError());
    return MGPU_OK;
}

// Framework atoms per work unit of pair_frozen_kernel: the fewest chunks that are a multiple of the eight waves of a
// workgroup (a workgroup takes eight chunks of one candidate group: no idle wave in the last one) and hold at most 30
// atoms.  Measured at the 2208-atom framework, chunks of 24 / 28 / 32 / 36 / 40 atoms, us per launch with its finalize:
// 1531 evaluations 52.2 / 42.4 / 42.3 / 46.2 / 47.6, 3066: 66.6 / 60.4 / 62.1 / 68.7 / 72.0, 6156: 106.5 / 105.0 / 110.8 /
// 111.2 / 97.6 -> 28 atoms (80 chunk slots, 79 used).
int frozen_chunk_atoms(const mgpu_engine *e, int n_atoms) {
    const int n_slots = kPairWaves * std::max(1, (n_atoms + kPairWaves * 30 - 1) / (kPairWaves * 30));
    return std::max(1, std::min(64, (n_atoms + n_slots - 1) / n_slots));
}

// The framework part of a launch segment, candidates in the lanes (pair_frozen_kernel): items of
// ONE residue type with n1 register sites; one extra record {e_lj, e_coul} per entry lands in d_extra.
int launch_frozen(mgpu_engine *e, Lane &ln, const PairItem *d_items, int n_items, int n1, int site_stride, bool fused, bool fast_fold,
                  int t_frozen, double2 *d_scratch, double2 *d_extra) {
    const int n_atoms = e->h_nmol[t_frozen] * e->tp.n1[t_frozen];
    const int chunk_atoms = frozen_chunk_atoms(e, n_atoms);
    const int n_chunks = (n_atoms + chunk_atoms - 1) / chunk_atoms;
    if (n_chunks == 0 || n_items == 0) return MGPU_OK;
    // one workgroup per (group of 64 candidates, eight chunks): pair_frozen_kernel
    const int n_wg_units = ((n_items + 63) / 64) * ((n_chunks + kPairWaves - 1) / kPairWaves);
    const bool ff = fast_fold && e->pair_fast_fold;
    hipEvent_t a = nullptr, b = nullptr;
    int rc;
    {
        const size_t need = (size_t)((n_items + 63) / 64) * sizeof(int);
        const void *before = ln.d_tickets.p;
        if ((rc = ln.d_tickets.reserve(need))) return rc;
        if (ln.d_tickets.p != before) HIP_TRY(hipMemsetAsync(ln.d_tickets.p, 0, ln.d_tickets.bytes, ln.stream));
    }
    if ((rc = prof_begin(e, ln, MGPU_KERNEL_PAIR, &a, &b))) return rc;
#define MGPU_LAUNCH_FROZEN_1(NS, FU, FW)                                                                                \
    do {                                                                                                               \
        const int nb = resident_blocks<&pair_frozen_kernel<NS, FU, FW>>(ln, e->coul_bytes);                                \
        const int grid_f = std::max(1, std::min(n_wg_units, e->n_cu * nb));                                           \
        hipExtLaunchKernelGGL((pair_frozen_kernel<NS, FU, FW>), dim3(grid_f), dim3(kPairBlock), e->coul_bytes, ln.stream, a, b, 0, \
                              e->tp, e->bx, e->d_pos, e->d_nmol, e->d_res_q, e->d_res_atype, e->d_pair_tab, e->d_coul_tab,   \
                              d_items, (const double *)ln.d_sites.p, site_stride, n_items, t_frozen, n_chunks, chunk_atoms, d_scratch,    \
                              (int *)ln.d_tickets.p, d_extra, (const double *)e->d_atom_q_on, (const int *)e->tp.slot_ty);     \
    } while (0)
#define MGPU_LAUNCH_FROZEN(NS)                                                                                          \
    do {                                                                                                               \
        if (fused && ff) MGPU_LAUNCH_FROZEN_1(NS, true, true);                                                         \
        else if (fused) MGPU_LAUNCH_FROZEN_1(NS, true, false);                                                         \
        else if (ff) MGPU_LAUNCH_FROZEN_1(NS, false, true);                                                            \
        else MGPU_LAUNCH_FROZEN_1(NS, false, false);                                                                   \
    } while (0)
    // (fused items have at most kMaxFusedSites sites: trial_submit_impl sends larger molecules' moves as two single-state items)
#define MGPU_LAUNCH_FROZEN_SINGLE(NS)                                                                                   \
    do {                                                                                                               \
        if (ff) MGPU_LAUNCH_FROZEN_1(NS, false, true);                                                                 \
        else MGPU_LAUNCH_FROZEN_1(NS, false, false);                                                                   \
    } while (0)
    if (fused && n1 > kMaxFusedSites) return set_error(MGPU_ERR_STATE, "launch_frozen: fused items have at most three sites");
    switch (n1) {
        case 1: MGPU_LAUNCH_FROZEN(1); break;
        case 2: MGPU_LAUNCH_FROZEN(2); break;
        case 3: MGPU_LAUNCH_FROZEN(3); break;
        case 4: MGPU_LAUNCH_FROZEN_SINGLE(4); break;
        default: MGPU_LAUNCH_FROZEN_SINGLE(5); break;
    }
#undef MGPU_LAUNCH_FROZEN_SINGLE
#undef MGPU_LAUNCH_FROZEN
#undef MGPU_LAUNCH_FROZEN_1
    if ((rc = prof_end(e, ln, MGPU_KERNEL_PAIR, a, b))) return rc;
    HIP_TRY(hipGetLastError());
    return MGPU_OK;
}
size_t recip_lds_bytes(const mgpu_engine *e, int n1_max) {
    const int ktot = e->kmax[0] + e->kmax[1] + e->kmax[2] + 3;
    return (size_t)2 * n1_max * ktot * sizeof(double2) + (size_t)n1_max * sizeof(double);
}

// sites per LDS tile of the per-k form: as many as fit kRecipTileBytes with both table sets, at least one
constexpr size_t kRecipTileBytes = 48 * 1024;
int recip_tile_sites(const mgpu_engine *e, int n1_max) {
    const size_t per_site = recip_lds_bytes(e, 1);
    return std::max(1, std::min(n1_max, (int)(kRecipTileBytes / per_site)));
}

size_t recip_rows_lds_bytes(const mgpu_engine *e, int n1_max) {
    return recip_lds_bytes(e, n1_max) + (size_t)e->n_rrows * (2 * n1_max * sizeof(double2));
}

// d_u_old != nullptr: also return the energy of the unchanged A(k) from the same pass (trial moves)
// row form while its XY table fits the LDS budget (molecules of a few sites), else the per-k form
bool recip_by_rows(const mgpu_engine *e, int n1_max) {
    return !e->recip_force_per_k && e->n_rtasks > 0 && recip_rows_lds_bytes(e, n1_max) <= 40 * 1024;
}

// The wide row form (recip_rows_wide_kernel): the phase tables of every site-state of the largest molecule of the launch
// in LDS, the XY table a tile of rows at a time.  Rows per tile (0: does not apply -- no row structure, tables beyond
// kRecipWideTableBytes, fewer than eight rows per tile).
constexpr size_t kRecipWideTableBytes = 40 * 1024, kRecipWideLdsBytes = 60 * 1024;   // (dynamic LDS: 64 KiB with the static part)
int recip_wide_rows_per_tile(const mgpu_engine *e, int n1_max) {
    if (e->recip_force_per_k || e->n_rtasks <= 0 || !e->d_row_first) return 0;
    const int ktot = e->kmax[0] + e->kmax[1] + e->kmax[2] + 3;
    const size_t nss = (size_t)2 * n1_max;
    const size_t tables = nss * ktot * sizeof(double2) + nss * sizeof(double);
    if (tables > kRecipWideTableBytes) return 0;
    const int rpt = (int)((kRecipWideLdsBytes - tables) / (nss * sizeof(double2)));
    return rpt >= 8 ? std::min(rpt, e->n_rrows) : 0;
}

// The matrix-unit form of the wide row sweep (recip_rows_wide_kernel<..., MFMA>): only the 1-D phase tables of a TILE of
// site-states in LDS.  Site-states per tile (a multiple of four; 0: the form does not apply): the fewest tiles of at most
// kRecipWideLdsBytes each, balanced -- one tile for a molecule of a few dozen sites; a molecule of any size otherwise, the four
// sums of a task carried from tile to tile.  (Measured, 1024 candidates of 128 / 300 sites: tiles of 52-72 KB 147 / 456-466 us,
// of 100-144 KB -- one workgroup per CU, opted in with hipFuncAttributeMaxDynamicSharedMemorySize -- 148-184 / 613-623 us.)
int recip_wide_mfma_tile(const mgpu_engine *e, int n1_max) {
    if (e->recip_force_per_k || e->recip_no_mfma || e->n_rtasks <= 0 || !e->d_row_first || !e->rows_contiguous) return 0;
    const int ktot = e->kmax[0] + e->kmax[1] + e->kmax[2] + 3;
    const size_t nss = ((size_t)2 * n1_max + 3) & ~(size_t)3;
    const size_t per_ss = (size_t)ktot * sizeof(double2) + sizeof(double), fixed = (size_t)e->n_rrows * sizeof(int4);
    if (fixed + 4 * per_ss > kRecipWideLdsBytes) return 0;
    const size_t fit = ((kRecipWideLdsBytes - fixed) / per_ss) & ~(size_t)3;
    const size_t n_tiles = (nss + fit - 1) / fit;
    return (int)((((nss + n_tiles - 1) / n_tiles) + 3) & ~(size_t)3);
}
bool recip_wide_mfma(const mgpu_engine *e, int n1_max) { return recip_wide_mfma_tile(e, n1_max) > 0; }

// accept != nullptr (commit, row form only): d_items are the candidates of the lane's last trial and only
// those whose bit is set are applied
int launch_recip(mgpu_engine *e, Lane &ln, const RecipItem *d_items, int n_items, int n1_max, int site_stride,
                 bool commit, double2 *A_base, double *d_u, double *d_u_old, const AcceptBits *accept, const double *sites_override,
                 const DecideArgs *decide) {
    const bool by_rows = recip_by_rows(e, n1_max);
    const double *d_cand = sites_override ? sites_override : (const double *)ln.d_sites.p;
    static const AcceptBits no_bits{};
    const AcceptBits &bits = accept ? *accept : no_bits;
    const int use_accept = accept ? 1 : 0;
    if (accept && !by_rows) return set_error(MGPU_ERR_STATE, "commit by accept mask needs the row-form kernel");
    if (decide && (!by_rows || commit || !d_u_old)) return set_error(MGPU_ERR_STATE, "device-side acceptance needs the row-form old + new k sweep");
    const DecideArgs no_decide{};
    // per-k form: the molecule's sites pass through LDS a tile at a time (recip_kernel), so no molecule is too large;
    // the tile is the most sites whose two table sets fit kRecipTileBytes (a few-site molecule: one tile, as before)
    const int tile = by_rows ? n1_max : recip_tile_sites(e, n1_max);
    const size_t lds = by_rows ? recip_rows_lds_bytes(e, n1_max) : recip_lds_bytes(e, tile);
    if (lds > 64 * 1024)
        return set_error(MGPU_ERR_CAPACITY, "reciprocal update: kmax too large for the LDS phase tables (" +
                                                std::to_string(lds) + " B > 64 KiB for one site)");
    hipEvent_t a = nullptr, b = nullptr;
    const int slot = commit ? MGPU_KERNEL_COMMIT : MGPU_KERNEL_RECIP;
    int rc = prof_begin(e, ln, slot, &a, &b);
    if (rc) return rc;
#define MGPU_LAUNCH_RECIP(COMMIT, BOTH)                                                                              \
    do {                                                                                                             \
        if (by_rows)                                                                                                 \
            hipExtLaunchKernelGGL((recip_rows_kernel<COMMIT, BOTH>), dim3(n_items), dim3(kBlock), lds, ln.stream, a, b, \
                                  0, e->tp, e->bx, e->d_pos, e->d_nmol, e->d_res_q, e->d_trj, e->d_tw, e->n_rtasks, e->d_rrows, e->n_rrows, \
                               A_base, d_items, d_cand, site_stride, d_u, d_u_old,      \
                                  bits, use_accept, no_decide);                                                     \
        else                                                                                                         \
            hipExtLaunchKernelGGL((recip_kernel<COMMIT, BOTH>), dim3(n_items), dim3(kBlock), lds, ln.stream, a, b, 0,   \
                                  e->tp, e->bx, e->d_pos, e->d_nmol, e->d_res_q, e->d_kpack, e->d_kslot, e->d_kw, A_base, d_items,  \
                               d_cand, site_stride, tile, d_u, d_u_old);                         \
    } while (0)
    const bool wide_ok = !by_rows && !accept && !decide;
    const bool wide_mfma = wide_ok && recip_wide_mfma(e, n1_max);
    const int wide_rpt = wide_mfma ? 0 : (wide_ok ? recip_wide_rows_per_tile(e, n1_max) : 0);
    if (wide_mfma || wide_rpt > 0) {
        // (matrix-unit form: nss_max = the site-states of one LDS tile, a multiple of four)
        const int ktot = e->kmax[0] + e->kmax[1] + e->kmax[2] + 3, nss_max = wide_mfma ? recip_wide_mfma_tile(e, n1_max) : 2 * n1_max;
        const size_t lds_w = (size_t)nss_max * ktot * sizeof(double2) + (size_t)wide_rpt * nss_max * sizeof(double2) + (size_t)nss_max * sizeof(double) +
                             (wide_mfma ? (size_t)e->n_rrows * sizeof(int4) : 0);
        // more than one tile of site-states: the tasks' four sums travel through a per-lane block [item][task][4]
        double *tile_sums = nullptr;
        if (wide_mfma && nss_max < ((2 * n1_max + 3) & ~3)) {
            if ((rc = ln.d_recip_sums.reserve((size_t)n_items * e->n_rtasks * 4 * sizeof(double)))) return rc;
            tile_sums = (double *)ln.d_recip_sums.p;
        }
#define MGPU_LAUNCH_WIDE_1(COMMIT, BOTH, MF, TI)                                                                     \
        do {                                                                                                         \
            hipExtLaunchKernelGGL((recip_rows_wide_kernel<COMMIT, BOTH, MF, TI>), dim3(n_items), dim3(kBlock), lds_w, ln.stream, a, b, 0, e->tp, \
                                  e->bx, e->d_pos, e->d_nmol, e->d_res_q, e->d_trj, e->d_tw, e->d_rrows, e->d_row_first, e->n_rrows, wide_rpt, \
                                  nss_max, A_base, d_items, d_cand, site_stride, d_u, d_u_old, tile_sums, e->n_rtasks);     \
        } while (0)
#define MGPU_LAUNCH_WIDE(COMMIT, BOTH, MF)                                                                           \
        do { if (MF && tile_sums) MGPU_LAUNCH_WIDE_1(COMMIT, BOTH, MF, true); else MGPU_LAUNCH_WIDE_1(COMMIT, BOTH, MF, false); } while (0)
        if (wide_mfma) {
            if (commit) MGPU_LAUNCH_WIDE(true, false, true);
            else if (d_u_old) MGPU_LAUNCH_WIDE(false, true, true);
            else MGPU_LAUNCH_WIDE(false, false, true);
        } else {
            if (commit) MGPU_LAUNCH_WIDE(true, false, false);
            else if (d_u_old) MGPU_LAUNCH_WIDE(false, true, false);
            else MGPU_LAUNCH_WIDE(false, false, false);
        }
#undef MGPU_LAUNCH_WIDE
#undef MGPU_LAUNCH_WIDE_1
    } else if (decide)
        hipExtLaunchKernelGGL((recip_rows_kernel<false, true, true>), dim3(n_items), dim3(kBlock), lds, ln.stream, a, b, 0, e->tp,
                              e->bx, e->d_pos, e->d_nmol, e->d_res_q, e->d_trj, e->d_tw, e->n_rtasks, e->d_rrows, e->n_rrows, A_base,
                              d_items, d_cand, site_stride, d_u, d_u_old, bits, 0, *decide);
    else if (commit) MGPU_LAUNCH_RECIP(true, false);
    else if (d_u_old) MGPU_LAUNCH_RECIP(false, true);
    else MGPU_LAUNCH_RECIP(false, false);
#undef MGPU_LAUNCH_RECIP
    rc = prof_end(e, ln, slot, a, b);
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return MGPU_OK;
}

// S(k) of one replica into dst[Nk]
int launch_sfactor(mgpu_engine *e, int replica, double2 *dst) {
    const int ncap = e->tp.n_cap_atoms;
    hipEvent_t a = nullptr, b = nullptr;
    int rc = prof_begin(e, e->lanes[0], MGPU_KERNEL_SFACTOR, &a, &b);
    if (rc) return rc;
    hipLaunchKernelGGL(phase_table_kernel, dim3((ncap + 255) / 256), dim3(256), 0, e->stream, e->tp, e->bx, e->d_pos,
                       e->d_nmol, e->d_atom_res, e->d_atom_mol, replica, e->d_phase_tab);
    hipExtLaunchKernelGGL(sfactor_kernel, dim3(e->nk), dim3(kBlock), 0, e->stream, a, b, 0, e->tp, e->bx, e->d_nmol,
                          e->d_atom_res, e->d_atom_mol, e->d_atom_q, e->d_kpack, e->d_kslot, replica, e->d_phase_tab, dst);
    rc = prof_end(e, e->lanes[0], MGPU_KERNEL_SFACTOR, a, b);
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return MGPU_OK;
}

// ComputeIntraResidueRealCoulombEnergySingleMol for items already on the device: one thread per molecule of up to
// kIntraThreadMax sites, one wave per larger one (each kernel skips the other's items; a kernel none of whose items can be
// its own is not launched)
int launch_intra(mgpu_engine *e, Lane &ln, const PairItem *d_items, int n_items, const double *d_sites, int site_stride, double *d_out) {
    if (n_items <= 0) return MGPU_OK;
    bool small = false, large = false;
    for (int t = 0; t < e->tp.n_res; ++t) {
        if (e->tp.n1[t] > kIntraThreadMax) large = true;
        else small = true;
    }
    if (small)
        hipLaunchKernelGGL(intra_kernel, dim3((n_items + 63) / 64), dim3(64), 0, ln.stream, e->tp, e->bx, e->d_pos, e->d_res_q, d_items,
                           n_items, d_sites, site_stride, d_out);
    if (large)
        hipLaunchKernelGGL(intra_wave_kernel, dim3(n_items), dim3(64), 0, ln.stream, e->tp, e->bx, e->d_pos, e->d_res_q, d_items,
                           n_items, d_sites, site_stride, d_out);
    HIP_TRY(hipGetLastError());
    return MGPU_OK;
}

}  // namespace mgpu
