// Internal types of the engine's translation units: the engine object, its submission lanes, and the helpers they share.
//   mgpu_engine.hip    life cycle, replica state, structure factor, static energy, measurement
//   mgpu_launch.hip    the kernel launches (pair sweeps, reciprocal update, intra, S(k))
//   mgpu_lanes.hip     batched candidates, the asynchronous lanes (trial submit / wait / commit), the host team
//   mgpu_windows.hip   one-launch windows: a single chain (mgpu_chain_window) and a farm of chains (mgpu_farm_window_*)
#ifndef MGPU_ENGINE_H
#define MGPU_ENGINE_H

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <omp.h>
#include <string>
#include <atomic>
#include <mutex>
#include <vector>

#include "../../include/maniac_gpu.h"
#include "mgpu_internal.h"
#include "mgpu_kernels.h"

namespace mgpu {

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t err__ = (expr);                                                                      \
        if (err__ != hipSuccess)                                                                        \
            return set_error(MGPU_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(err__));       \
    } while (0)

// grow-only device / pinned-host scratch
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int reserve(size_t need) {
        if (need <= bytes) return MGPU_OK;
        if (p) HIP_TRY(hipFree(p));
        p = nullptr; bytes = 0;
        size_t cap = std::max<size_t>(need, 4096);
        cap += cap / 2;
        HIP_TRY(hipMalloc(&p, cap));
        bytes = cap;
        return MGPU_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};
struct HostBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int reserve(size_t need) {
        if (need <= bytes) return MGPU_OK;
        if (p) HIP_TRY(hipHostFree(p));
        p = nullptr; bytes = 0;
        size_t cap = std::max<size_t>(need, 4096);
        cap += cap / 2;
        HIP_TRY(hipHostMalloc(&p, cap, hipHostMallocDefault));
        bytes = cap;
        return MGPU_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; bytes = 0; }
};

struct ProfileSlot {
    long long launches = 0;
    double total_ms = 0.0;
};

// One submission lane: a HIP stream with its own scratch, so that work queued on one lane (for
// one group of replicas) overlaps the host's processing of the other lane's results.
struct Lane {
    hipStream_t stream = nullptr;
    DevBuf d_items, d_items2, d_sites, d_partials, d_out;
    HostBuf h_in, h_commit, h_out;   // pinned staging: trial inputs, commit inputs, results
    bool h_in_lent = false;          // h_in.p was handed to the caller (mgpu_lane_site_buffer): it must never be freed under them
    struct Pending { int kernel; hipEvent_t a, b; };
    std::vector<Pending> pending;
    struct Occupancy { const void *kernel; size_t lds; int blocks; };
    std::vector<Occupancy> occ;      // resident_blocks() cache
    // profiling state is per lane: lanes may be driven by different host threads (one thread per lane at a time)
    std::vector<hipEvent_t> ev_pool;
    ProfileSlot prof[MGPU_KERNEL_COUNT];
    int n_submitted = 0;          // candidates of the trial in flight (0 = none)
    bool dirty = false;           // something was queued on the stream since its last synchronise (asynchronous entry points)
    int last_trial_n = 0, last_trial_stride = 0;   // shape of the site rows still resident in d_sites
    bool last_trial_built = false;                 // ... built on the device (rows carry the candidates' frames)
    int last_trial_frame = 0;                      // site index of the frame inside such a row
    int n_pair_items = 0, n_partials = 0;   // reduced pair-energy entries of the trial in flight (2 per fused item + 1 per single) and its split partials
    int n_fused = 0;                      // fused (old + new) items of the trial in flight, all site-count classes together
    // reduced pair-energy entry i sums `n_split` partials starting at double ent_off[i] of the result block, ent_stride[i]
    // doubles apart (a fused item's partials are laid out [split][state], a single item's [split])
    std::vector<int> ent_off, ent_stride, ent_ns;     // ... ent_ns[i] of them
    std::vector<char> ent_extra;          // entry i has an extra record (the framework part, pair_frozen_kernel) behind the energies
    DevBuf d_scratch;                     // chunk partials of pair_frozen_kernel
    DevBuf d_recip_sums;                  // [item][task][4]: a task's sums between the site tiles of the matrix-unit row sweep
    DevBuf d_tickets;                     // its per-group tickets: zero between launches (the kernel leaves them so)
    const RecipItem *d_trial_items = nullptr;   // RecipItems of the last trial, resident while last_trial_n != 0
    const RecipItem *h_trial_items = nullptr;   // their host image in h_in (valid until the next trial_submit)
    int trial_n1_max = 1;
    std::vector<int> pair_old, pair_new, intra_idx, kinds;   // per-candidate rows of the trial in flight
    std::vector<double> self_of;                              // per-candidate Ewald self term (host constant)
    std::vector<char> cand_ok;                                // per candidate: its sites are within the fast fold's range
    std::vector<int> build_kind;                              // candidate kinds of a device-built trial
    std::vector<double> h_lj, h_cc;                           // pair energies of the trial being collected
    std::vector<int> mark;                    // [n_replicas] scratch of the one-candidate-per-replica check
    std::vector<int> commit_mark;             // [n_replicas]: the stamp of the commit_submit_impl call that last committed there
    int commit_stamp = 0;
    // A trial whose acceptance is decided (and whose accepted candidates are committed) on the device: the flags arrive
    // with the energies; the engine's host mirrors (counts, range flags) follow when the lane is next synchronised
    int decided_n = 0;                        // candidates of such a trial not yet folded into the mirrors (0 = none)
    int decided_wait_n = 0;                   // ... whose outcomes the caller has not collected yet (mgpu_trial_decide_wait)
    size_t decided_at = 0;                    // byte offset of the flags in h_out
    hipEvent_t commit_staged_ev = nullptr;                    // recorded behind the H2D copies that read h_commit
    bool commit_staged = false;
    // Farm windows (mgpu_farm_window_submit / _wait): up to kFarmDepth windows of this lane in flight.  The host-side blocks
    // are rings (a window's records are read, and its results written, while the next window is being prepared); the
    // device scratch is shared (a stream runs its kernels one after the other).
    struct FarmWindow {
        FarmRec *h_recs = nullptr;               // pinned [kFarmDepth][cap]
        double *h_out = nullptr;                 // pinned [kFarmDepth][cap][kFarmOut]
        unsigned long long *h_tag = nullptr;     // pinned [kFarmDepth][cap]
        double2 *d_part = nullptr;               // [cap][2 nsplit]
        ChainResult *d_res = nullptr;            // [cap]
        int *d_tickets = nullptr;                // [cap], zero between launches
        int cap = 0;
        unsigned long long seq = 0;              // windows submitted so far
        struct Pending {
            unsigned long long seq;
            int n, slot;
            bool counts_change;                  // carries an insertion / deletion: must be waited before the next submit
            std::vector<int> rep, t, kind;
            std::vector<char> ok;                // the candidate's sites are within the fast fold's range
        };
        std::deque<Pending> pending;
        void release() {
            if (h_recs) (void)hipHostFree(h_recs);
            if (h_out) (void)hipHostFree(h_out);
            if (h_tag) (void)hipHostFree(h_tag);
            if (d_part) (void)hipFree(d_part);
            if (d_res) (void)hipFree(d_res);
            if (d_tickets) (void)hipFree(d_tickets);
            h_recs = nullptr; h_out = nullptr; h_tag = nullptr; d_part = nullptr; d_res = nullptr; d_tickets = nullptr;
            cap = 0;
            pending.clear();
        }
    } farm;
    void release() {
        farm.release();
        if (commit_staged_ev) { (void)hipEventDestroy(commit_staged_ev); commit_staged_ev = nullptr; }
        d_items.release(); d_items2.release(); d_sites.release(); d_partials.release(); d_out.release();
        d_scratch.release(); d_recip_sums.release();
        d_tickets.release();
        h_in.release(); h_commit.release(); h_out.release();
    }
};
constexpr int kLanes = 4;
constexpr int kFarmDepth = 4;           // farm windows a lane may have in flight
constexpr int kFarmMaxChains = 4096;   // chains per farm window

}  // namespace mgpu

using namespace mgpu;

struct mgpu_engine {
    int device = 0;
    int n_replicas = 0;
    Lane lanes[kLanes];
    Topo tp{};
    BoxDev bx{};
    // host copies
    std::vector<int> atoms_in_res, mol_capacity, is_active, atom_types;  // atom_types 1-based [n_res][max_atom]
    std::vector<double> charges, epsilon, sigma;
    std::vector<int> kx, ky, kz;
    std::vector<double> k2mag, form_factor, weights;
    std::vector<int> h_nmol;  // [R][n_res]
    // [R][n_res]: 1 while every site ever written for (replica, type) lies within one box length of the cell centre
    // on every axis -- the condition under which the pair sweep may fold separations with two instructions per axis
    std::vector<char> in_range;
    double rc = 0, tol = 0, alpha = 0, volume = 0;
    int box_type = 0, kmax[3] = {0, 0, 0}, nk = 0;
    double box_matrix[9]{}, bounds_lo[3]{}, reciprocal[9]{}, metrics[9]{};
    // device state
    double *d_pos = nullptr;      // [R][3][Ncap]
    int *d_nmol = nullptr;        // [R][n_res]
    double2 *d_A = nullptr;       // [R][Nk]
    int *d_kpack = nullptr;
    double *d_kw = nullptr;
    int *d_trj = nullptr;            // row form of the k list (recip_rows_kernel): packed task words,
    double2 *d_tw = nullptr;         // task weights {ff W (+j), ff W (-j)}
    int *d_kslot = nullptr;          // k (reference order) -> slot of A(k)
    std::vector<int> kslot;
    int n_slots = 0;                 // complex entries of A(k) per replica
    RecipRow *d_rrows = nullptr;
    int *d_row_first = nullptr;      // [n_rrows + 1] first task of every row (a row's tasks are contiguous): recip_rows_wide_kernel
    int n_rtasks = 0, n_rrows = 0;
    double2 *d_pair_tab = nullptr;
    char *d_coul_tab = nullptr;      // Coulomb table rows (build_coulomb_table), staged into LDS by the pair sweep
    size_t coul_bytes = 0;
    int n_cu = 256;                  // compute units of the device
    int pair_blocks_per_cu = kPairBlock >= 1024 ? 1 : 2;      // resident pair-sweep workgroups per CU (VGPR / LDS bound)
    int pair_nsplit = 1;             // waves per pair-sweep item: an engine constant (see engine_nsplit)
    std::vector<double> self_of_type; // ComputeEwaldSelfInteractionSingleMol per residue type (host constant)
    bool rows_contiguous = false;    // every row's tasks are consecutive |kz| (the matrix-unit row sweep's addressing)
    bool recip_no_mfma = false;      // MGPU_RECIP_NO_MFMA=1: many-site molecules through the vector form of the wide row sweep
    bool pair_fast_fold = true;      // two-instruction minimum-image fold where the atoms' range allows it (MGPU_PAIR_EXACT_FOLD=1: off)
    bool recip_force_per_k = false;  // MGPU_RECIP_PER_K=1: per-k reciprocal kernel even where the row form fits (tests)
    double *d_res_q = nullptr;
    int *d_res_atype = nullptr;
    // frozen residues (inactive, n1 >= 64): site_perm[t][a] = position of the caller's site a in the engine's
    // atom-type-sorted order (identity for every other residue type)
    std::vector<std::vector<int>> site_perm;
    std::vector<char> frozen;        // [n_res]
    bool any_frozen = false;
    int *d_atom_ty = nullptr;        // [Ncap] 0-based atom type of every slot (pair_flat_kernel fetches it per lane)
    // A frozen framework is normally the SAME in every replica (a farm copies replica 0): frozen_ref[t] = the coordinates
    // replica 0 was given (engine site order), frozen_same[r * n_res + t] = replica r holds exactly those, frozen_diff[t] =
    // replicas that do not.  Where all agree, batched trials sweep the framework with pair_frozen_kernel (candidates in the
    // lanes, the atoms scalar) -- MGPU_NO_FROZEN_BATCH=1 keeps pair_flat_kernel for it.
    std::vector<std::vector<double>> frozen_ref;
    std::vector<char> frozen_same;
    std::vector<int> frozen_diff;
    bool frozen_batch = true;
    int host_team = 1;               // host threads the per-candidate loops of submit / wait / commit may use (mgpu_set_host_team)
    // molecule frames (mgpu_replica_set_frames): com [R][3][n_mol_slots], off [R][3][Ncap]; allocated on first use
    double *d_com = nullptr, *d_off = nullptr;
    std::vector<char> frames_ok;     // [R][n_res]: the frames of (replica, type) mirror its sites
    std::vector<char> frames_tight;  // [R][n_res]: every molecule's centre lies in the cell and its offsets within 0.24 L:
                                     // any device-built candidate then lies within the fast fold's range
    // Register-site sweeps of this engine go through pair_flat_kernel (one software-pipelined loop over all units of
    // a work unit) instead of the plane-by-plane pair_sweep_kernel: chosen at creation for topologies with short planes
    // (every plane-major residue type has at most kFlatMaxCap molecule slots) or a frozen residue; MGPU_PAIR_FLAT=0 / 1
    // overrides (tuning / A-B).  Orthorhombic boxes only; a site-major ACTIVE residue (n1 >= 64) keeps the other kernel.
    bool pair_flat = false;
    int *d_atom_res = nullptr, *d_atom_mol = nullptr;
    double *d_atom_q = nullptr;
    double *d_atom_q_on = nullptr;   // the same with charges below CoulombEnergy's threshold set to zero (pair_frozen_kernel's scalars)
    double2 *d_phase_tab = nullptr;  // [ktot][Ncap] scratch for S(k)
    double2 *d_S = nullptr;          // [Nk] scratch
    // lane 0 doubles as the synchronous path's stream and scratch
    hipStream_t &stream = lanes[0].stream;
    DevBuf &d_items = lanes[0].d_items, &d_items2 = lanes[0].d_items2, &d_sites = lanes[0].d_sites,
           &d_partials = lanes[0].d_partials, &d_out = lanes[0].d_out;
    HostBuf &h_out = lanes[0].h_out;
    HostBuf h_stage;
    // single-chain windows (mgpu_chain_window): pinned, host-coherent blocks the kernel reads its candidates from and
    // writes its results to (no copies, no stream synchronisation: the host polls the tag), and device scratch
    struct Chain {
        double *h_out = nullptr;                     // [kChainMaxCand][10] energies | first, undecided | stage stamps
        unsigned long long *h_tag = nullptr;
        Topo *d_topo = nullptr;                      // the engine's Topo in device memory (the kernel indexes it by loaded residue types)
        bool topo_stale = true;
        ChainResult *d_res = nullptr;
        double2 *d_part = nullptr;
        double2 *d_alt = nullptr;                    // [kChainMaxCand][n_slots]: A + delta of every candidate of the window
        int *d_ticket = nullptr;
        unsigned long long seq = 0;
        double margin = 16.0 * 2.220446049250313e-16;   // relative band around the acceptance probability left to the host's exp
        long long windows = 0, undecided = 0;
        bool timing = false;                         // stage stamps wanted (mgpu_chain_set_timing)
    } chain;
    // farm windows (mgpu_farm_window_*): the other A(k) buffer of every replica, which of the two is current, and the
    // per-replica stall flag, all allocated on first use; `dirty` = some replica's current A(k) may live in d_A_alt (every
    // entry point that reads or writes A(k) outside a farm window first copies it back: farm_window_normalize)
    struct Farm {
        double2 *d_A_alt = nullptr;
        int *d_acur = nullptr, *d_stalled = nullptr;
        std::atomic<bool> dirty{false};
        std::atomic<long long> windows{0}, undecided{0};   // (the lanes may be driven by different host threads)
        std::mutex mu;                                     // the engine-wide blocks' first allocation
    } farm;
    // profiling
    bool profiling = false;
};

namespace mgpu {

const char *last_error_text();
int use_device(const mgpu_engine *e);
int prof_begin(mgpu_engine *e, Lane &ln, int kernel, hipEvent_t *a, hipEvent_t *b);
int prof_end(mgpu_engine *e, Lane &ln, int kernel, hipEvent_t a, hipEvent_t b);
int prof_collect(mgpu_engine *e, Lane &ln);
void finish_decided(mgpu_engine *e, Lane &ln);
void frozen_changed(mgpu_engine *e, int replica, int t);
int sync_lane(mgpu_engine *e, Lane &ln);
int sync_stream(mgpu_engine *e);
int sync_all_lanes(mgpu_engine *e);
int check_candidate(const mgpu_engine *e, int c, int replica, int t, int m, bool need_resident);
bool sites_in_range(const mgpu_engine *e, const double *sites, int n_sites);
bool replica_in_range(const mgpu_engine *e, int replica);
int engine_nsplit(const mgpu_engine *e);
void permute_frozen_rows(const mgpu_engine *e, double *rows, int n_rows, int site_stride, const int *t);
bool any_frozen(const mgpu_engine *e, int n, const int *t);
int upload_sites(Lane &ln, const double *sites, int n_rows, int site_stride);
int upload_sites(mgpu_engine *e, const double *sites, int n_rows, int site_stride, const int *t);
double self_energy_host(const mgpu_engine *e, int t);
int farm_window_normalize(mgpu_engine *e);      // mgpu_windows.hip
int chain_topo(mgpu_engine *e, const Topo **d_topo);   // the engine's Topo in device memory (mgpu_windows.hip)
// mgpu_launch.hip
int launch_pair(mgpu_engine *e, Lane &ln, const PairItem *d_items, int n_items, int common_n1, int site_stride,
                int nsplit, double *d_lj, double *d_c, bool ordered = false, double2 *host_partials = nullptr,
                bool fused = false, bool fast_fold = false, bool skip_frozen = false);
int frozen_chunk_atoms(const mgpu_engine *e, int n_atoms);
int launch_frozen(mgpu_engine *e, Lane &ln, const PairItem *d_items, int n_items, int n1, int site_stride, bool fused, bool fast_fold,
                  int t_frozen, double2 *d_scratch, double2 *d_extra);
size_t recip_lds_bytes(const mgpu_engine *e, int n1_max);
int recip_tile_sites(const mgpu_engine *e, int n1_max);
size_t recip_rows_lds_bytes(const mgpu_engine *e, int n1_max);
bool recip_by_rows(const mgpu_engine *e, int n1_max);
int recip_wide_rows_per_tile(const mgpu_engine *e, int n1_max);   // 0: the wide row form does not apply
bool recip_wide_mfma(const mgpu_engine *e, int n1_max);
int recip_wide_mfma_tile(const mgpu_engine *e, int n1_max);       // site-states per LDS tile (0: the matrix-unit form does not apply)
int launch_recip(mgpu_engine *e, Lane &ln, const RecipItem *d_items, int n_items, int n1_max, int site_stride,
                 bool commit, double2 *A_base, double *d_u, double *d_u_old = nullptr, const AcceptBits *accept = nullptr,
                 const double *sites_override = nullptr, const DecideArgs *decide = nullptr);
int launch_sfactor(mgpu_engine *e, int replica, double2 *dst);
int launch_intra(mgpu_engine *e, Lane &ln, const PairItem *d_items, int n_items, const double *d_sites, int site_stride, double *d_out);

// The per-candidate loops of a submit / wait / commit are cut into `parts` contiguous ranges (boundaries on multiples of
// 32 candidates: the commit's accept mask is built a word per range) and run by an OpenMP team of the calling thread --
// the same runtime as the Fortran drivers', whose nested hot team is reused.  One part = the serial loop.
constexpr int kHostPartMin = 1024;          // candidates below which a team is not worth waking
constexpr int kMaxHostParts = 16;
static int host_parts(const mgpu_engine *e, int n) {
    return (e->host_team > 1 && n >= kHostPartMin) ? std::min(e->host_team, kMaxHostParts) : 1;
}
static void part_range(int n, int parts, int part, int &c0, int &c1) {
    const int words = (n + 31) / 32;
    c0 = std::min(n, (int)((long long)words * part / parts) * 32);
    c1 = std::min(n, (int)((long long)words * (part + 1) / parts) * 32);
}
template <class F>
static void for_parts(int parts, F &&f) {
    if (parts <= 1) { f(0); return; }
#pragma omp parallel for num_threads(parts) schedule(static, 1)
    for (int part = 0; part < parts; ++part) f(part);
}
// what a part has to say when a candidate is refused: the caller reports the lowest candidate's message (the serial loop's)
struct PartError {
    int c = -1, rc = MGPU_OK;
    std::string msg;
    void set(int cand, int code, const std::string &m) { if (c < 0) { c = cand; rc = code; msg = m; } }
};
static int report_first(const PartError *errs, int parts) {
    const PartError *first = nullptr;
    for (int q = 0; q < parts; ++q)
        if (errs[q].c >= 0 && (!first || errs[q].c < first->c)) first = &errs[q];
    return first ? set_error(first->rc, first->msg) : MGPU_OK;
}

// sites per item if all items agree, else 0
template <class Item>
int common_site_count(const mgpu_engine *e, const std::vector<Item> &items) {
    int n1 = 0;
    for (const auto &it : items) {
        const int v = e->tp.n1[it.t];
        if (n1 == 0) n1 = v;
        else if (n1 != v) return 0;
    }
    return n1;
}

}  // namespace mgpu

#endif
