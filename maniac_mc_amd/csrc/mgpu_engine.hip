// C-ABI entry points (include/maniac_gpu.h) and host-side orchestration of the HIP kernels.
// One mgpu_engine = one HIP device + one stream + R replicas sharing box / force field / k table.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <omp.h>
#include <string>
#include <vector>

#include "../../include/maniac_gpu.h"
#include "mgpu_internal.h"
#include "mgpu_kernels.h"

namespace mgpu {

static thread_local std::string g_last_error;

int set_error(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

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
    void release() {
        if (commit_staged_ev) { (void)hipEventDestroy(commit_staged_ev); commit_staged_ev = nullptr; }
        d_items.release(); d_items2.release(); d_sites.release(); d_partials.release(); d_out.release();
        d_scratch.release();
        d_tickets.release();
        h_in.release(); h_commit.release(); h_out.release();
    }
};
constexpr int kLanes = 4;

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
    int n_rtasks = 0, n_rrows = 0;
    double2 *d_pair_tab = nullptr;
    char *d_coul_tab = nullptr;      // Coulomb table rows (build_coulomb_table), staged into LDS by the pair sweep
    size_t coul_bytes = 0;
    int n_cu = 256;                  // compute units of the device
    int pair_blocks_per_cu = kPairBlock >= 1024 ? 1 : 2;      // resident pair-sweep workgroups per CU (VGPR / LDS bound)
    int pair_nsplit = 1;             // waves per pair-sweep item: an engine constant (see engine_nsplit)
    std::vector<double> self_of_type; // ComputeEwaldSelfInteractionSingleMol per residue type (host constant)
    bool pair_fuse = true;           // trial moves sweep old + new together (MGPU_PAIR_NO_FUSE=1: tuning / A-B only)
    int pair_fuse_max = kMaxFusedSites;   // largest molecule whose trial moves are fused (MGPU_PAIR_FUSE_MAX: up to kMaxFusedSitesWide)
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
    int frozen_chunk = 0;            // framework atoms per pair_frozen_kernel work unit; 0 = frozen_chunk_atoms' rule.  The chunk
                                     // partials are summed in order, so the chunking depends on the framework's size alone,
                                     // never on the batch (MGPU_FROZEN_CHUNK, <= 64, overrides)
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
        int *d_ticket = nullptr;
        unsigned long long seq = 0;
        double margin = 16.0 * 2.220446049250313e-16;   // relative band around the acceptance probability left to the host's exp
        long long windows = 0, undecided = 0;
        bool timing = false;                         // stage stamps wanted (mgpu_chain_set_timing)
    } chain;
    // profiling
    bool profiling = false;
};

namespace {

int use_device(const mgpu_engine *e) {
    HIP_TRY(hipSetDevice(e->device));
    return MGPU_OK;
}

int prof_begin(mgpu_engine *e, Lane &ln, int kernel, hipEvent_t *a, hipEvent_t *b) {
    if (!e->profiling) return MGPU_OK;
    for (hipEvent_t *ev : {a, b}) {
        if (!ln.ev_pool.empty()) { *ev = ln.ev_pool.back(); ln.ev_pool.pop_back(); }
        else HIP_TRY(hipEventCreate(ev));
    }
    (void)kernel;
    return MGPU_OK;
}
// The events are attached to the dispatch itself (hipExtLaunchKernelGGL start / stop events): they carry
// the kernel's own begin / end timestamps, as rocprofv3 reports them, and put no extra barrier packets
// into the stream.  With profiling off both are null and the launch is an ordinary one.
int prof_end(mgpu_engine *e, Lane &ln, int kernel, hipEvent_t a, hipEvent_t b) {
    if (!e->profiling) return MGPU_OK;
    ln.pending.push_back({kernel, a, b});
    return MGPU_OK;
}
// after a stream synchronise: fold the recorded event pairs into the per-kernel totals
int prof_collect(mgpu_engine *e, Lane &ln) {
    for (auto &p : ln.pending) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, p.a, p.b));
        ln.prof[p.kernel].launches += 1;
        ln.prof[p.kernel].total_ms += ms;
        ln.ev_pool.push_back(p.a);
        ln.ev_pool.push_back(p.b);
    }
    ln.pending.clear();
    return MGPU_OK;
}

void finish_decided(mgpu_engine *e, Lane &ln);
void frozen_changed(mgpu_engine *e, int replica, int t);
int sync_lane(mgpu_engine *e, Lane &ln) {
    HIP_TRY(hipStreamSynchronize(ln.stream));
    ln.dirty = false;
    if (ln.decided_n) finish_decided(e, ln);
    return prof_collect(e, ln);
}
int sync_stream(mgpu_engine *e) { return sync_lane(e, e->lanes[0]); }
// The synchronous entry points that read or rewrite replica state (coordinates, counts, A(k)) on lane 0's stream or
// the null stream first drain EVERY lane: the lanes' streams are non-blocking, so work still queued on lanes 1-3
// would otherwise race with them.
int sync_all_lanes(mgpu_engine *e) {
    // lane 0 doubles as the synchronous path's stream (those entry points synchronise it themselves before they
    // return); the other lanes only carry work queued by the asynchronous entry points, which mark them dirty
    for (auto &ln : e->lanes)
        if (&ln == &e->lanes[0] || ln.dirty || !ln.pending.empty())
            if (int rc = sync_lane(e, ln)) return rc;
    return MGPU_OK;
}

// The device accepted and committed some candidates of the lane's last trial (recip_rows_kernel<false, true, true>): the
// host mirrors of the molecule counts and of the fast-fold range flags catch up from the flags in the result block.
void finish_decided(mgpu_engine *e, Lane &ln) {
    const int n = ln.decided_n;
    ln.decided_n = 0;
    const int *flags = (const int *)((const char *)ln.h_out.p + ln.decided_at);
    const RecipItem *items = ln.h_trial_items;
    for (int c = 0; c < n; ++c) {
        if (!flags[c]) continue;
        const int idx = items[c].replica * e->tp.n_res + items[c].t;
        if (items[c].kind == MGPU_CREATION) e->h_nmol[idx] += 1;
        if (items[c].kind == MGPU_DELETION) e->h_nmol[idx] -= 1;
        if (items[c].kind != MGPU_DELETION && !(c < (int)ln.cand_ok.size() && ln.cand_ok[c])) e->in_range[idx] = 0;
        frozen_changed(e, items[c].replica, items[c].t);
    }
    // the rows were consumed by the device's commit: nothing is left to commit "from the lane's resident rows"
    ln.last_trial_n = 0;
    ln.d_trial_items = nullptr;
}

// The sites or the count of a frozen (framework) residue type changed on one replica: it no longer equals the reference
// copy pair_frozen_kernel sweeps (replica 0's); a change of replica 0 itself invalidates the reference for everyone.
void frozen_changed(mgpu_engine *e, int replica, int t) {
    if (!e->frozen[t]) return;
    auto clear = [&](int r) {
        char &f = e->frozen_same[(size_t)r * e->tp.n_res + t];
        if (f) { e->frozen_diff[t] += 1; f = 0; }
    };
    if (replica == 0) for (int r = 0; r < e->n_replicas; ++r) clear(r);
    else clear(replica);
}

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

int check_candidate(const mgpu_engine *e, int c, int replica, int t, int m, bool need_resident) {
    if (replica < 0 || replica >= e->n_replicas)
        return set_error(MGPU_ERR_INVALID_ARG, "candidate " + std::to_string(c) + ": replica out of range");
    if (t < 0 || t >= e->tp.n_res)
        return set_error(MGPU_ERR_INVALID_ARG, "candidate " + std::to_string(c) + ": residue type out of range");
    const int nm = e->h_nmol[replica * e->tp.n_res + t];
    if (m >= nm || m < -1)
        return set_error(MGPU_ERR_INVALID_ARG, "candidate " + std::to_string(c) + ": molecule slot " +
                                                   std::to_string(m) + " not live (count " + std::to_string(nm) + ")");
    if (need_resident && m < 0)
        return set_error(MGPU_ERR_INVALID_ARG, "candidate " + std::to_string(c) + ": needs a resident molecule");
    return MGPU_OK;
}

// every coordinate of `n_sites` sites within kFastFoldRange box lengths of the cell centre (orthorhombic axes): two
// such sites are less than 1.5 L apart on every axis, the fast fold's precondition.  Molecules whose centre of mass is
// wrapped into the cell (ApplyPBC) qualify as long as their radius stays below a quarter of the box.
constexpr double kFastFoldRange = 0.745;
bool sites_in_range(const mgpu_engine *e, const double *sites, int n_sites) {
    for (int i = 0; i < n_sites; ++i)
        for (int d = 0; d < 3; ++d)
            if (!(std::fabs(sites[3 * (size_t)i + d] - e->bx.ctr[d]) <= kFastFoldRange * e->bx.L[d])) return false;
    return true;
}
bool replica_in_range(const mgpu_engine *e, int replica) {
    for (int t = 0; t < e->tp.n_res; ++t)
        if (!e->in_range[(size_t)replica * e->tp.n_res + t]) return false;
    return true;
}

// Waves per pair-sweep item.  One wave sweeps every nsplit-th 64-atom unit of the item's replica and the split
// partials are added in split order, so the last bits of a pair energy depend on nsplit: it is therefore an
// ENGINE CONSTANT (a function of the topology's capacity only) -- never of how many candidates share a launch
// or of another replica's state -- and a chain's trajectory does not depend on what runs beside it.  Waves are
// persistent and stride over the n_items * nsplit work units, so a constant costs nothing when a launch has
// more work units than resident waves.  Policy: never fewer than ~8 sweep units per wave.  A farm engine (>= 256
// replicas) aims at one or two work units per resident wave for the launches a farm makes (a quarter to a half of
// the replicas per launch): n_cu * 64 / n_replicas rounded down to a power of two, between 1 and 4 (up to 16 below 1024
// chains, see below) -- 4 at 2048
// replicas, 2 at 8192 (measured at the 10 125-atom box, 1024 fused items per launch: 4 -> 98.9 us, 8 -> 104.6 us,
// 16 -> 116 us; 2048 items per launch on four lanes: 2 -> 6.94 M, 4 -> 6.79 M, 8 -> 6.52 M accepted moves/s).
// A small engine (< 256 replicas: single chains, a handful of chains) is latency-bound: up to 32 waves per item, two sweep
// units per wave (round 4, one chain, stage stamps of mgpu_chain_window: 10 125-atom box 59 -> 32 waves 16.7 -> 15.2 us to
// the window's results, 128 waves 15.1 us with a longer reduction; framework box 5 -> 32 waves 21.1 -> 12.1 us).
// MGPU_PAIR_NSPLIT overrides it (tuning only; read once at engine creation).
int engine_nsplit(const mgpu_engine *e) {
    int units = 0;
    for (int t = 0; t < e->tp.n_res; ++t) {
        const int cap = e->tp.cap[t], n1 = e->tp.n1[t];
        units += e->tp.site_major[t] ? cap * ((n1 + 63) / 64) : n1 * ((cap + 63) / 64);
    }
    int cap_split = 32, per_wave = 2;
    if (e->n_replicas >= 256) {
        per_wave = 8;
        int want = std::max(1, e->n_cu * 64 / e->n_replicas);
        // Short work units (the grand-canonical boxes: a few dozen units per item, and a launch carries 1.5 items per
        // candidate) are dominated by their tail: twice the waves per item fill the last round (round 3, framework box,
        // 3064 items per launch on 4096 resident waves: nsplit 2 = 1.5 rounds of 19 units, nsplit 4 = 3 rounds of 10)
        if (units <= 128) want *= 2;
        // Below 1024 chains a launch (half the chains on two lanes) leaves most of the GPU idle at 4 waves per item and is
        // a latency chain per step: a quarter of `want`, up to 16 (round 4, 10 125-atom box, two lanes, accepted moves/s at
        // nsplit 4 / 8 / 16 / 32: 256 chains 0.99 / 1.24 / 1.42 / 1.37 M, 512: 1.96 / 2.40 / 2.37 / 2.12 M,
        // 1024: 4.84 / 4.66 / 3.35 / 3.07 M, 2048: 6.00 / 5.77 / 4.29 / 3.71 M)
        const int cap_want = e->n_replicas >= 1024 ? std::min(want, 4) : std::min(want / 4, 16);
        cap_split = 1;
        while (cap_split * 2 <= cap_want) cap_split *= 2;
    }
    int ns = std::max(1, std::min(units / per_wave, cap_split));
    if (const char *ov = std::getenv("MGPU_PAIR_NSPLIT")) ns = std::max(1, std::min(std::atoi(ov), std::max(1, units)));
    return ns;
}

// Candidate rows of a frozen residue type (an inactive framework given as an explicit candidate: rare) are handed over
// in the caller's site order and live on the device in the engine's atom-type-sorted order: permute such rows in place.
void permute_frozen_rows(const mgpu_engine *e, double *rows, int n_rows, int site_stride, const int *t) {
    std::vector<double> tmp;
    for (int c = 0; c < n_rows; ++c) {
        if (t[c] < 0 || t[c] >= e->tp.n_res || !e->frozen[t[c]]) continue;
        const int n1 = e->tp.n1[t[c]];
        if (n1 > site_stride) continue;                       // the caller reports the error
        double *r = rows + (size_t)c * site_stride * 3;
        tmp.assign(r, r + (size_t)n1 * 3);
        const int *perm = e->site_perm[t[c]].data();
        for (int a = 0; a < n1; ++a)
            for (int d = 0; d < 3; ++d) r[(size_t)perm[a] * 3 + d] = tmp[(size_t)a * 3 + d];
    }
}
bool any_frozen(const mgpu_engine *e, int n, const int *t) {
    for (int c = 0; c < n; ++c)
        if (t[c] >= 0 && t[c] < e->tp.n_res && e->frozen[t[c]]) return true;
    return false;
}

int upload_sites(Lane &ln, const double *sites, int n_rows, int site_stride) {
    if (!sites || n_rows == 0) return MGPU_OK;
    ln.last_trial_n = 0;
    const size_t bytes = (size_t)n_rows * site_stride * 3 * sizeof(double);
    int rc = ln.d_sites.reserve(bytes);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(ln.d_sites.p, sites, bytes, hipMemcpyHostToDevice, ln.stream));
    return MGPU_OK;
}
// synchronous entry points: rows go straight from the caller's memory unless a frozen residue type is among them
int upload_sites(mgpu_engine *e, const double *sites, int n_rows, int site_stride, const int *t) {
    if (sites && n_rows > 0 && any_frozen(e, n_rows, t)) {
        const size_t bytes = (size_t)n_rows * site_stride * 3 * sizeof(double);
        int rc = e->h_stage.reserve(bytes);
        if (rc) return rc;
        std::memcpy(e->h_stage.p, sites, bytes);
        permute_frozen_rows(e, (double *)e->h_stage.p, n_rows, site_stride, t);
        if ((rc = upload_sites(e->lanes[0], (const double *)e->h_stage.p, n_rows, site_stride))) return rc;
        HIP_TRY(hipStreamSynchronize(e->lanes[0].stream));   // h_stage is reused by other entry points
        return MGPU_OK;
    }
    return upload_sites(e->lanes[0], sites, n_rows, site_stride);
}

// resident workgroups per CU of a pair-sweep instantiation: asked of the runtime once per (lane, instantiation, LDS
// size) -- a lane is driven by one host thread at a time and belongs to one engine (one device, one Coulomb table), so
// the cache needs no lock and never serves another engine's value
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
                int nsplit, double *d_lj, double *d_c, bool ordered = false, double2 *host_partials = nullptr,
                bool fused = false, bool fast_fold = false, bool skip_frozen = false) {
    const int n_work = n_items * nsplit;
    int rc = MGPU_OK;
    if (fused && (!host_partials || ordered || e->bx.triclinic || common_n1 < 1 || common_n1 > e->pair_fuse_max))
        return set_error(MGPU_ERR_STATE, "launch_pair: fused sweep needs register sites, an orthorhombic box and a partials buffer");
    if (!host_partials && (rc = ln.d_partials.reserve((size_t)n_work * sizeof(double2)))) return rc;
    double2 *d_part = host_partials ? host_partials : (double2 *)ln.d_partials.p;
    // persistent waves: 2 workgroups of 8 waves per CU (VGPRs: 4 waves per SIMD at <= 128), never more
    // workgroups than there is work for
    const int per_cu = (fused && common_n1 > kMaxFusedSites) ? 1 : e->pair_blocks_per_cu;
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
            case 3: MGPU_LAUNCH_FLAT(3, true); break;
            case 4: MGPU_LAUNCH_FLAT(4, true); break;
            default: MGPU_LAUNCH_FLAT(5, true); break;
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
            case 3: MGPU_PAIR_FF(3, true); break;
            case 4: MGPU_PAIR_FF(4, true); break;   // wide instantiations: 2 waves per SIMD, one workgroup per CU
            default: MGPU_PAIR_FF(5, true); break;
        }
    } else if (e->bx.triclinic) {
        if (ordered) MGPU_LAUNCH_PAIR(0, true, true);
        else MGPU_LAUNCH_PAIR(0, false, true);
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
    if (e->frozen_chunk > 0) return e->frozen_chunk;
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
    switch (n1) {
        case 1: MGPU_LAUNCH_FROZEN(1); break;
        case 2: MGPU_LAUNCH_FROZEN(2); break;
        case 3: MGPU_LAUNCH_FROZEN(3); break;
        case 4: MGPU_LAUNCH_FROZEN(4); break;
        default: MGPU_LAUNCH_FROZEN(5); break;
    }
#undef MGPU_LAUNCH_FROZEN
#undef MGPU_LAUNCH_FROZEN_1
    if ((rc = prof_end(e, ln, MGPU_KERNEL_PAIR, a, b))) return rc;
    HIP_TRY(hipGetLastError());
    return MGPU_OK;
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

size_t recip_lds_bytes(const mgpu_engine *e, int n1_max) {
    const int ktot = e->kmax[0] + e->kmax[1] + e->kmax[2] + 3;
    return (size_t)2 * n1_max * ktot * sizeof(double2) + (size_t)n1_max * sizeof(double);
}

size_t recip_rows_lds_bytes(const mgpu_engine *e, int n1_max) {
    return recip_lds_bytes(e, n1_max) + (size_t)e->n_rrows * (2 * n1_max * sizeof(double2));
}

// d_u_old != nullptr: also return the energy of the unchanged A(k) from the same pass (trial moves)
// row form while its XY table fits the LDS budget (molecules of a few sites), else the per-k form
bool recip_by_rows(const mgpu_engine *e, int n1_max) {
    return !e->recip_force_per_k && e->n_rtasks > 0 && recip_rows_lds_bytes(e, n1_max) <= 40 * 1024;
}

// accept != nullptr (commit, row form only): d_items are the candidates of the lane's last trial and only
// those whose bit is set are applied
int launch_recip(mgpu_engine *e, Lane &ln, const RecipItem *d_items, int n_items, int n1_max, int site_stride,
                 bool commit, double2 *A_base, double *d_u, double *d_u_old = nullptr, const AcceptBits *accept = nullptr,
                 const double *sites_override = nullptr, const DecideArgs *decide = nullptr) {
    const bool by_rows = recip_by_rows(e, n1_max);
    const double *d_cand = sites_override ? sites_override : (const double *)ln.d_sites.p;
    static const AcceptBits no_bits{};
    const AcceptBits &bits = accept ? *accept : no_bits;
    const int use_accept = accept ? 1 : 0;
    if (accept && !by_rows) return set_error(MGPU_ERR_STATE, "commit by accept mask needs the row-form kernel");
    if (decide && (!by_rows || commit || !d_u_old)) return set_error(MGPU_ERR_STATE, "device-side acceptance needs the row-form old + new k sweep");
    const DecideArgs no_decide{};
    const size_t lds = by_rows ? recip_rows_lds_bytes(e, n1_max) : recip_lds_bytes(e, n1_max);
    if (lds > 64 * 1024)
        return set_error(MGPU_ERR_CAPACITY, "reciprocal update: molecule too large for the LDS phase tables (" +
                                                std::to_string(lds) + " B > 64 KiB)");
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
                               d_cand, site_stride, d_u, d_u_old);                               \
    } while (0)
    if (decide)
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

double self_energy_host(const mgpu_engine *e, int t) {
    // ComputeEwaldSelfInteractionSingleMol, ewald_energy.f90:308-336
    double s = 0.0;
    const double sqrtpi = std::sqrt(kPi);
    for (int a = 0; a < e->tp.n1[t]; ++a) {
        const double q = e->charges[(size_t)t * e->tp.max_atom + a];
        if (std::fabs(q) < kErrorTol) continue;
        s = s - e->alpha / sqrtpi * (q * q);
    }
    return s * kEps0InvEvA / kKbEvK;
}

}  // namespace

extern "C" {

const char *mgpu_last_error(void) { return g_last_error.c_str(); }

int mgpu_abi_version(void) { return 1; }

int mgpu_device_count(int *count) {
    if (!count) return set_error(MGPU_ERR_INVALID_ARG, "mgpu_device_count: null argument");
    int n = 0;
    hipError_t err = hipGetDeviceCount(&n);
    if (err != hipSuccess || n <= 0) {
        *count = 0;
        return set_error(MGPU_ERR_NO_DEVICE, std::string("no HIP device: ") + hipGetErrorString(err));
    }
    *count = n;
    return MGPU_OK;
}

int mgpu_engine_create(mgpu_engine **out, int device, int n_replicas, int n_res, const int *atoms_in_res,
                       const int *mol_capacity, int max_atom, const int *atom_types, const double *charges,
                       const int *is_active, int n_types, const double *epsilon, const double *sigma,
                       const double box_matrix[9], const double bounds_lo[3], double real_space_cutoff,
                       double ewald_tolerance) {
    if (!out || !atoms_in_res || !mol_capacity || !atom_types || !charges || !is_active || !epsilon || !sigma ||
        !box_matrix || !bounds_lo)
        return set_error(MGPU_ERR_INVALID_ARG, "mgpu_engine_create: null argument");
    *out = nullptr;
    if (n_replicas < 1) return set_error(MGPU_ERR_INVALID_ARG, "mgpu_engine_create: n_replicas < 1");
    if (n_res < 1 || n_res > kMaxRes)
        return set_error(MGPU_ERR_INVALID_ARG, "mgpu_engine_create: n_res must be in [1, " + std::to_string(kMaxRes) + "]");
    if (n_types < 1 || n_types > kMaxTypes)
        return set_error(MGPU_ERR_INVALID_ARG, "mgpu_engine_create: n_types must be in [1, " + std::to_string(kMaxTypes) + "]");
    if (max_atom < 1) return set_error(MGPU_ERR_INVALID_ARG, "mgpu_engine_create: max_atom < 1");
    for (int t = 0; t < n_res; ++t) {
        if (atoms_in_res[t] < 1 || atoms_in_res[t] > max_atom || mol_capacity[t] < 1)
            return set_error(MGPU_ERR_INVALID_ARG, "mgpu_engine_create: bad atoms_in_res / mol_capacity");
        for (int a = 0; a < atoms_in_res[t]; ++a) {
            const int ty = atom_types[t * max_atom + a];
            if (ty < 1 || ty > n_types) return set_error(MGPU_ERR_INVALID_ARG, "mgpu_engine_create: atom type out of range");
        }
    }
    int ndev = 0;
    int rc = mgpu_device_count(&ndev);
    if (rc) return rc;
    if (device < 0 || device >= ndev) return set_error(MGPU_ERR_NO_DEVICE, "mgpu_engine_create: device ordinal out of range");

    auto *e = new mgpu_engine();
    e->device = device;
    e->n_replicas = n_replicas;
    std::memcpy(e->box_matrix, box_matrix, sizeof(double) * 9);
    std::memcpy(e->bounds_lo, bounds_lo, sizeof(double) * 3);
    rc = box_prepare(box_matrix, &e->box_type, &e->volume, e->reciprocal, e->metrics);
    if (rc) { delete e; return rc; }
    e->rc = real_space_cutoff;
    e->tol = ewald_tolerance;
    double screening, fprec;
    rc = ewald_setup(e->metrics, &e->rc, &e->tol, &e->alpha, &screening, &fprec, e->kmax, &e->nk);
    if (rc) { delete e; return rc; }
    if (e->kmax[0] > 127 || e->kmax[1] > 127 || e->kmax[2] > 127) {
        delete e;
        return set_error(MGPU_ERR_CAPACITY, "mgpu_engine_create: kmax > 127 does not fit the packed k table");
    }
    e->kx.resize(e->nk); e->ky.resize(e->nk); e->kz.resize(e->nk);
    e->k2mag.resize(e->nk); e->form_factor.resize(e->nk); e->weights.resize(e->nk);
    rc = ewald_kvectors(e->reciprocal, e->alpha, e->kmax, e->nk, e->kx.data(), e->ky.data(), e->kz.data(),
                        e->k2mag.data(), e->form_factor.data(), e->weights.data());
    if (rc) { delete e; return rc; }

    e->atoms_in_res.assign(atoms_in_res, atoms_in_res + n_res);
    e->mol_capacity.assign(mol_capacity, mol_capacity + n_res);
    e->is_active.assign(is_active, is_active + n_res);
    e->atom_types.assign(atom_types, atom_types + (size_t)n_res * max_atom);
    e->charges.assign(charges, charges + (size_t)n_res * max_atom);
    e->epsilon.assign(epsilon, epsilon + (size_t)n_types * n_types);
    e->sigma.assign(sigma, sigma + (size_t)n_types * n_types);
    e->h_nmol.assign((size_t)n_replicas * n_res, 0);
    e->in_range.assign((size_t)n_replicas * n_res, 1);

    Topo &tp = e->tp;
    tp.n_res = n_res; tp.n_types = n_types; tp.max_atom = max_atom;
    int off = 0;
    for (int t = 0; t < n_res; ++t) {
        tp.n1[t] = atoms_in_res[t];
        tp.cap[t] = mol_capacity[t];
        tp.seg_off[t] = off;
        tp.site_major[t] = atoms_in_res[t] >= 64 ? 1 : 0;
        off += atoms_in_res[t] * mol_capacity[t];
    }
    tp.n_cap_atoms = off;
    tp.com = nullptr;
    tp.off = nullptr;
    tp.n_mol_slots = 0;
    for (int t = 0; t < n_res; ++t) { tp.mol_off[t] = tp.n_mol_slots; tp.n_mol_slots += mol_capacity[t]; }
    e->frames_ok.assign((size_t)n_replicas * n_res, 0);
    e->frozen_ref.assign(n_res, std::vector<double>());
    e->frozen_same.assign((size_t)n_replicas * n_res, 0);
    e->frozen_diff.assign(n_res, n_replicas);
    e->frozen_batch = std::getenv("MGPU_NO_FROZEN_BATCH") == nullptr;
    if (const char *ov = std::getenv("MGPU_FROZEN_CHUNK")) e->frozen_chunk = std::max(1, std::min(std::atoi(ov), 64));
    e->frames_tight.assign((size_t)n_replicas * n_res, 0);
    // Layout of the big inactive residues and the register-site pair kernel go together: "frozen" (sites sorted by
    // atom type, one group per type present) + pair_flat_kernel, or site-major + pair_sweep_kernel.  Default: flat
    // wherever a framework (inactive, >= 64 atoms) is present -- measured round 3, 2208-atom framework + 4-site water,
    // 1532 evaluations per launch: flat 44.7 us, site-major plane-by-plane 46.8 us, type-sorted plane-by-plane 53.8 us;
    // CO2 box without a framework: flat 12.3 us, plane-by-plane 11.3 us, so plain boxes keep pair_sweep_kernel.
    // MGPU_PAIR_FLAT=1 forces the flat kernel wherever it is eligible, =0 never; MGPU_NO_FROZEN=1 keeps site-major.
    std::vector<int4> grp_tab;
    auto build_layout = [&](bool use_frozen) {
        grp_tab.clear();
        e->site_perm.assign(n_res, std::vector<int>());
        e->frozen.assign(n_res, 0);
        e->any_frozen = false;
        for (int t = 0; t < n_res; ++t) {
            std::vector<int> &perm = e->site_perm[t];
            perm.resize(tp.n1[t]);
            tp.n_grp[t] = 0;
            tp.grp_off[t] = (int)grp_tab.size();
            tp.site_major[t] = tp.n1[t] >= 64 ? ((use_frozen && is_active[t] == 0) ? 2 : 1) : 0;
            if (tp.site_major[t] != 2) {
                for (int a = 0; a < tp.n1[t]; ++a) perm[a] = a;
                continue;
            }
            e->frozen[t] = 1;
            e->any_frozen = true;
            std::vector<int> order(tp.n1[t]);
            for (int a = 0; a < tp.n1[t]; ++a) order[a] = a;
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
                return atom_types[(size_t)t * max_atom + a] < atom_types[(size_t)t * max_atom + b];
            });
            int cur_ty = -1;
            for (int pos = 0; pos < tp.n1[t]; ++pos) {
                const int a = order[pos], ty = atom_types[(size_t)t * max_atom + a] - 1;
                perm[a] = pos;
                if (ty != cur_ty) {
                    grp_tab.push_back(make_int4(pos, 0, ty, 0));
                    tp.n_grp[t] += 1;
                    cur_ty = ty;
                }
                grp_tab.back().y += 1;
            }
        }
    };
    {
        bool framework = false;
        for (int t = 0; t < n_res; ++t) framework = framework || (tp.n1[t] >= 64 && is_active[t] == 0);
        bool want_flat = framework;
        if (const char *ov = std::getenv("MGPU_PAIR_FLAT")) want_flat = std::atoi(ov) != 0;
        build_layout(want_flat && std::getenv("MGPU_NO_FROZEN") == nullptr);
        bool ok = e->box_type != 3 && (int)grp_tab.size() <= kMaxGrp;
        int planes = 0;                        // one lane of a wave builds one plane's record
        for (int t = 0; t < n_res; ++t) {
            if (tp.site_major[t] == 1) ok = false;
            planes += tp.site_major[t] == 2 ? tp.cap[t] * tp.n_grp[t] : tp.n1[t];
        }
        if (planes > kFlatMaxPlanes) ok = false;
        e->pair_flat = want_flat && ok;
        if (!e->pair_flat && e->any_frozen) build_layout(false);
        for (size_t g = 0; g < (size_t)kMaxGrp; ++g) {
            tp.grp_start[g] = g < grp_tab.size() ? grp_tab[g].x : 0;
            tp.grp_cnt[g] = g < grp_tab.size() ? grp_tab[g].y : 0;
            tp.grp_ty[g] = g < grp_tab.size() ? grp_tab[g].z : 0;
        }
    }
    e->pair_fuse = std::getenv("MGPU_PAIR_NO_FUSE") == nullptr;
    e->pair_fast_fold = std::getenv("MGPU_PAIR_EXACT_FOLD") == nullptr;
    if (const char *ov = std::getenv("MGPU_PAIR_FUSE_MAX")) e->pair_fuse_max = std::max(1, std::min(std::atoi(ov), kMaxFusedSitesWide));
    e->recip_force_per_k = std::getenv("MGPU_RECIP_PER_K") != nullptr;

    BoxDev &bx = e->bx;
    for (int d = 0; d < 3; ++d) {
        bx.L[d] = box_matrix[d * 3 + d]; bx.invL[d] = 1.0 / bx.L[d]; bx.kmax[d] = e->kmax[d];
        bx.ctr[d] = bounds_lo[d] + 0.5 * bx.L[d];
        bx.lo[d] = bounds_lo[d];
    }
    std::memcpy(bx.rcp, e->reciprocal, sizeof(double) * 9);
    std::memcpy(bx.m, box_matrix, sizeof(double) * 9);
    bx.triclinic = e->box_type == 3 ? 1 : 0;
    bx.rc2 = e->rc * e->rc;
    bx.alpha = e->alpha;
    bx.volume = e->volume;
    bx.nk = e->nk;

    // host images of the device tables
    std::vector<int> kpack(e->nk);
    std::vector<double> kw(e->nk);
    for (int i = 0; i < e->nk; ++i) {
        kpack[i] = e->kx[i] | ((e->ky[i] + 128) << 8) | ((e->kz[i] + 128) << 16);
        kw[i] = e->form_factor[i] * e->weights[i];  // ewald_energy.f90:266, (ff * W) * |A|^2
    }
    // row form of the k list: rows (kx, ky) in order of first appearance, one task per (row, |kz|)
    std::vector<RecipRow> rrows;
    std::vector<RecipTask> rtasks;
    {
        std::vector<std::vector<int>> plus, minus;          // per row: k index of +j / -j, -1 if absent
        for (int i = 0; i < e->nk; ++i) {
            if (rrows.empty() || rrows.back().kx != e->kx[i] || rrows.back().ky != e->ky[i]) {
                rrows.push_back(RecipRow{e->kx[i], e->ky[i]});
                plus.emplace_back(e->kmax[2] + 1, -1);
                minus.emplace_back(e->kmax[2] + 1, -1);
            }
            const int kz = e->kz[i];
            (kz >= 0 ? plus : minus).back()[kz >= 0 ? kz : -kz] = i;
        }
        for (size_t r = 0; r < rrows.size(); ++r)
            for (int j = 0; j <= e->kmax[2]; ++j)
                if (plus[r][j] >= 0 || minus[r][j] >= 0) rtasks.push_back(RecipTask{plus[r][j], minus[r][j], (int)r, j});
        // every k exactly once (the rows of the list are contiguous by construction; verify)
        size_t covered = 0;
        for (const auto &tk : rtasks) covered += (tk.kp >= 0) + (tk.km >= 0);
        if (covered != (size_t)e->nk) { rrows.clear(); rtasks.clear(); }   // fall back to the per-k kernel
    }
    e->n_rrows = (int)rrows.size();
    e->n_rtasks = (int)rtasks.size();
    // A(k) in task order: slots 2t / 2t + 1 hold the +j / -j member of task t (zero where the list has none)
    e->kslot.assign(e->nk, 0);
    std::vector<int> trj(rtasks.size());
    std::vector<double2> tw(rtasks.size());
    if (e->n_rtasks > 0) {
        e->n_slots = 2 * e->n_rtasks;
        for (size_t ti = 0; ti < rtasks.size(); ++ti) {
            const RecipTask &tk = rtasks[ti];
            trj[ti] = (tk.row << 8) | tk.j | (tk.kp >= 0 ? kTaskHasP : 0) | (tk.km >= 0 ? kTaskHasM : 0);
            tw[ti] = make_double2(tk.kp >= 0 ? kw[tk.kp] : 0.0, tk.km >= 0 ? kw[tk.km] : 0.0);
            if (tk.kp >= 0) e->kslot[tk.kp] = 2 * (int)ti;
            if (tk.km >= 0) e->kslot[tk.km] = 2 * (int)ti + 1;
        }
    } else {
        e->n_slots = e->nk;
        for (int i = 0; i < e->nk; ++i) e->kslot[i] = i;
    }
    bx.n_slots = e->n_slots;
    std::vector<double2> ptab((size_t)n_types * n_types);
    for (int i = 0; i < n_types * n_types; ++i) ptab[i] = make_double2(4.0 * epsilon[i], sigma[i] * sigma[i]);
    // device site templates in the ENGINE's site order (frozen residues: sorted by atom type); the host copies
    // e->charges / e->atom_types keep the caller's order (self energies are summed in the reference's order)
    std::vector<int> atype0((size_t)n_res * max_atom, 0);
    std::vector<double> q_dev((size_t)n_res * max_atom, 0.0);
    for (int t = 0; t < n_res; ++t)
        for (int a = 0; a < atoms_in_res[t]; ++a) {
            const int ap = e->site_perm[t][a];
            atype0[(size_t)t * max_atom + ap] = atom_types[(size_t)t * max_atom + a] - 1;
            q_dev[(size_t)t * max_atom + ap] = charges[(size_t)t * max_atom + a];
        }
    std::vector<int> a_res(tp.n_cap_atoms), a_mol(tp.n_cap_atoms), a_ty(tp.n_cap_atoms);
    std::vector<double> a_q(tp.n_cap_atoms);
    for (int t = 0; t < n_res; ++t)
        for (int m = 0; m < tp.cap[t]; ++m)
            for (int a = 0; a < tp.n1[t]; ++a) {
                const int ap = e->site_perm[t][a];
                const int j = tp.site_major[t] ? tp.seg_off[t] + m * tp.n1[t] + ap : tp.seg_off[t] + ap * tp.cap[t] + m;
                a_res[j] = t; a_mol[j] = m; a_q[j] = charges[(size_t)t * max_atom + a];
                a_ty[j] = atom_types[(size_t)t * max_atom + a] - 1;
            }

    auto fail = [&](int code) { mgpu_engine_destroy(e); return code; };
#define HIP_TRY_E(expr)                                                                                          \
    do {                                                                                                         \
        hipError_t err__ = (expr);                                                                               \
        if (err__ != hipSuccess)                                                                                 \
            return fail(set_error(MGPU_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(err__)));          \
    } while (0)
    HIP_TRY_E(hipSetDevice(device));
    for (auto &ln : e->lanes) HIP_TRY_E(hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking));
    const size_t ncap = tp.n_cap_atoms, R = n_replicas;
    const int ktot = e->kmax[0] + e->kmax[1] + e->kmax[2] + 3;
    HIP_TRY_E(hipMalloc(&e->d_pos, R * 3 * ncap * sizeof(double)));
    HIP_TRY_E(hipMemset(e->d_pos, 0, R * 3 * ncap * sizeof(double)));
    HIP_TRY_E(hipMalloc(&e->d_nmol, R * n_res * sizeof(int)));
    HIP_TRY_E(hipMemset(e->d_nmol, 0, R * n_res * sizeof(int)));
    HIP_TRY_E(hipMalloc(&e->d_A, R * e->n_slots * sizeof(double2)));
    HIP_TRY_E(hipMemset(e->d_A, 0, R * e->n_slots * sizeof(double2)));
    HIP_TRY_E(hipMalloc(&e->d_kpack, e->nk * sizeof(int)));
    HIP_TRY_E(hipMalloc(&e->d_kw, e->nk * sizeof(double)));
    HIP_TRY_E(hipMalloc(&e->d_pair_tab, ptab.size() * sizeof(double2)));
    {
        // Coulomb table for this alpha, covering every minimum-image distance of this box
        std::vector<CoulRow> rows;
        // orthorhombic: (half diagonal)^2; triclinic: the 27-image minimum of sites that may sit up to a
        // cell away is bounded by (|a| + |b| + |c|)^2
        const double sum_len = e->metrics[0] + e->metrics[1] + e->metrics[2];
        const double s_max = bx.triclinic ? sum_len * sum_len
                                          : 0.25 * (bx.L[0] * bx.L[0] + bx.L[1] * bx.L[1] + bx.L[2] * bx.L[2]) * 1.0001;
        int idx_base = 0;
        if (int rc2 = build_coulomb_table(e->alpha, std::max(s_max, 1.0), rows, &idx_base)) return fail(rc2);
        e->bx.coul_idx_base = idx_base;
        e->bx.coul_last_row = (int)rows.size() - 1;
        e->coul_bytes = rows.size() * sizeof(CoulRow);
        if (e->coul_bytes + 16 * 1024 > 150 * 1024) return fail(set_error(MGPU_ERR_CAPACITY, "Coulomb table does not fit LDS"));
        HIP_TRY_E(hipMalloc(&e->d_coul_tab, e->coul_bytes));
        HIP_TRY_E(hipMemcpy(e->d_coul_tab, rows.data(), e->coul_bytes, hipMemcpyHostToDevice));
    }
    {
        hipDeviceProp_t prop;
        HIP_TRY_E(hipGetDeviceProperties(&prop, device));
        e->n_cu = std::max(1, prop.multiProcessorCount);
        if (const char *ov = std::getenv("MGPU_PAIR_BLOCKS_PER_CU")) e->pair_blocks_per_cu = std::max(1, std::atoi(ov));   // tuning only
        e->pair_nsplit = engine_nsplit(e);
    }
    HIP_TRY_E(hipMalloc(&e->d_res_q, e->charges.size() * sizeof(double)));
    HIP_TRY_E(hipMalloc(&e->d_res_atype, atype0.size() * sizeof(int)));
    HIP_TRY_E(hipMalloc(&e->d_atom_res, ncap * sizeof(int)));
    HIP_TRY_E(hipMalloc(&e->d_atom_mol, ncap * sizeof(int)));
    HIP_TRY_E(hipMalloc(&e->d_atom_q, ncap * sizeof(double)));
    HIP_TRY_E(hipMalloc(&e->d_atom_q_on, ncap * sizeof(double)));
    HIP_TRY_E(hipMalloc(&e->d_phase_tab, (size_t)ktot * ncap * sizeof(double2)));
    HIP_TRY_E(hipMalloc(&e->d_S, e->n_slots * sizeof(double2)));
    HIP_TRY_E(hipMemset(e->d_S, 0, e->n_slots * sizeof(double2)));
    HIP_TRY_E(hipMalloc(&e->d_kslot, e->nk * sizeof(int)));
    HIP_TRY_E(hipMemcpy(e->d_kslot, e->kslot.data(), e->nk * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY_E(hipMemcpy(e->d_kpack, kpack.data(), e->nk * sizeof(int), hipMemcpyHostToDevice));
    if (e->n_rtasks > 0) {
        HIP_TRY_E(hipMalloc(&e->d_trj, trj.size() * sizeof(int)));
        HIP_TRY_E(hipMalloc(&e->d_tw, tw.size() * sizeof(double2)));
        HIP_TRY_E(hipMalloc(&e->d_rrows, rrows.size() * sizeof(RecipRow)));
        HIP_TRY_E(hipMemcpy(e->d_trj, trj.data(), trj.size() * sizeof(int), hipMemcpyHostToDevice));
        HIP_TRY_E(hipMemcpy(e->d_tw, tw.data(), tw.size() * sizeof(double2), hipMemcpyHostToDevice));
        HIP_TRY_E(hipMemcpy(e->d_rrows, rrows.data(), rrows.size() * sizeof(RecipRow), hipMemcpyHostToDevice));
    }
    HIP_TRY_E(hipMemcpy(e->d_kw, kw.data(), e->nk * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY_E(hipMemcpy(e->d_pair_tab, ptab.data(), ptab.size() * sizeof(double2), hipMemcpyHostToDevice));
    HIP_TRY_E(hipMemcpy(e->d_res_q, q_dev.data(), q_dev.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY_E(hipMalloc(&e->d_atom_ty, ncap * sizeof(int)));
    HIP_TRY_E(hipMemcpy(e->d_atom_ty, a_ty.data(), ncap * sizeof(int), hipMemcpyHostToDevice));
    tp.slot_q = e->d_atom_q;
    tp.slot_ty = e->d_atom_ty;
    HIP_TRY_E(hipMemcpy(e->d_res_atype, atype0.data(), atype0.size() * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY_E(hipMemcpy(e->d_atom_res, a_res.data(), ncap * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY_E(hipMemcpy(e->d_atom_mol, a_mol.data(), ncap * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY_E(hipMemcpy(e->d_atom_q, a_q.data(), ncap * sizeof(double), hipMemcpyHostToDevice));
    {
        std::vector<double> a_q_on(a_q);
        for (double &q : a_q_on)
            if (!(std::fabs(q) >= kErrorTol)) q = 0.0;                   // energy_utils.f90:430
        HIP_TRY_E(hipMemcpy(e->d_atom_q_on, a_q_on.data(), ncap * sizeof(double), hipMemcpyHostToDevice));
    }
#undef HIP_TRY_E
    e->self_of_type.resize(n_res);
    for (int t = 0; t < n_res; ++t) e->self_of_type[t] = self_energy_host(e, t);
    *out = e;
    return MGPU_OK;
}

int mgpu_engine_destroy(mgpu_engine *e) {
    if (!e) return MGPU_OK;
    (void)hipSetDevice(e->device);
    for (auto &ln : e->lanes) if (ln.stream) (void)hipStreamSynchronize(ln.stream);
    for (void *p : {(void *)e->d_pos, (void *)e->d_nmol, (void *)e->d_A, (void *)e->d_kpack, (void *)e->d_kw,
                    (void *)e->d_pair_tab, (void *)e->d_coul_tab, (void *)e->d_res_q, (void *)e->d_res_atype, (void *)e->d_atom_res,
                    (void *)e->d_atom_mol, (void *)e->d_atom_q, (void *)e->d_atom_q_on, (void *)e->d_phase_tab, (void *)e->d_S, (void *)e->d_trj,
                    (void *)e->d_tw, (void *)e->d_kslot, (void *)e->d_rrows, (void *)e->d_atom_ty, (void *)e->d_com, (void *)e->d_off})
        if (p) (void)hipFree(p);
    e->h_stage.release();
    for (void *p : {(void *)e->chain.h_out, (void *)e->chain.h_tag})
        if (p) (void)hipHostFree(p);
    for (void *p : {(void *)e->chain.d_res, (void *)e->chain.d_part, (void *)e->chain.d_ticket, (void *)e->chain.d_topo})
        if (p) (void)hipFree(p);
    for (auto &ln : e->lanes) {
        ln.release();
        for (auto &p : ln.pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
        if (ln.stream) (void)hipStreamDestroy(ln.stream);
    }
    for (auto &ln : e->lanes)
        for (auto ev : ln.ev_pool) (void)hipEventDestroy(ev);
    delete e;
    return MGPU_OK;
}

int mgpu_engine_get_ewald(const mgpu_engine *e, double *alpha, double *rc, double *tol, int kmax[3], int *nk,
                          double *volume, int *box_type) {
    if (!e) return set_error(MGPU_ERR_INVALID_ARG, "null engine");
    if (alpha) *alpha = e->alpha;
    if (rc) *rc = e->rc;
    if (tol) *tol = e->tol;
    if (kmax) { kmax[0] = e->kmax[0]; kmax[1] = e->kmax[1]; kmax[2] = e->kmax[2]; }
    if (nk) *nk = e->nk;
    if (volume) *volume = e->volume;
    if (box_type) *box_type = e->box_type;
    return MGPU_OK;
}

int mgpu_engine_get_kvectors(const mgpu_engine *e, int *kx, int *ky, int *kz, double *k2mag, double *ff, double *w) {
    if (!e) return set_error(MGPU_ERR_INVALID_ARG, "null engine");
    if (kx) std::memcpy(kx, e->kx.data(), e->nk * sizeof(int));
    if (ky) std::memcpy(ky, e->ky.data(), e->nk * sizeof(int));
    if (kz) std::memcpy(kz, e->kz.data(), e->nk * sizeof(int));
    if (k2mag) std::memcpy(k2mag, e->k2mag.data(), e->nk * sizeof(double));
    if (ff) std::memcpy(ff, e->form_factor.data(), e->nk * sizeof(double));
    if (w) std::memcpy(w, e->weights.data(), e->nk * sizeof(double));
    return MGPU_OK;
}

// ---- replica state ---------------------------------------------------------------------------

static int check_replica_t(const mgpu_engine *e, int replica, int t) {
    if (!e) return set_error(MGPU_ERR_INVALID_ARG, "null engine");
    if (replica < 0 || replica >= e->n_replicas) return set_error(MGPU_ERR_INVALID_ARG, "replica out of range");
    if (t < 0 || t >= e->tp.n_res) return set_error(MGPU_ERR_INVALID_ARG, "residue type out of range");
    return MGPU_OK;
}

int mgpu_replica_set_molecules(mgpu_engine *e, int replica, int t, int n_mol, const double *sites) {
    int rc = check_replica_t(e, replica, t);
    if (rc) return rc;
    if (n_mol < 0 || n_mol > e->tp.cap[t]) return set_error(MGPU_ERR_CAPACITY, "n_mol exceeds the residue type's mol_capacity");
    if (n_mol > 0 && !sites) return set_error(MGPU_ERR_INVALID_ARG, "null sites");
    if ((rc = use_device(e))) return rc;
    const Topo &tp = e->tp;
    const int n1 = tp.n1[t], cap = tp.cap[t];
    const size_t seg = (size_t)n1 * cap;
    rc = e->h_stage.reserve(3 * seg * sizeof(double));
    if (rc) return rc;
    double *st = (double *)e->h_stage.p;
    std::memset(st, 0, 3 * seg * sizeof(double));
    const int *perm = e->site_perm[t].data();
    for (int m = 0; m < n_mol; ++m)
        for (int a = 0; a < n1; ++a) {
            const size_t j = tp.site_major[t] ? (size_t)m * n1 + perm[a] : (size_t)a * cap + m;
            for (int d = 0; d < 3; ++d) st[d * seg + j] = sites[((size_t)m * n1 + a) * 3 + d];
        }
    if ((rc = sync_all_lanes(e))) return rc;
    for (int d = 0; d < 3; ++d)
        HIP_TRY(hipMemcpy(e->d_pos + ((size_t)replica * 3 + d) * tp.n_cap_atoms + tp.seg_off[t], st + d * seg,
                          seg * sizeof(double), hipMemcpyHostToDevice));
    e->h_nmol[replica * tp.n_res + t] = n_mol;
    if (e->frozen[t]) {
        auto set_same = [&](int r, char v) {
            char &f = e->frozen_same[(size_t)r * tp.n_res + t];
            e->frozen_diff[t] += (f ? 1 : 0) - (v ? 1 : 0);
            f = v;
        };
        if (replica == 0) {
            e->frozen_ref[t].assign(st, st + 3 * seg);
            for (int r = 1; r < e->n_replicas; ++r) set_same(r, 0);
            set_same(0, 1);
        } else {
            const bool same = e->frozen_ref[t].size() == 3 * seg && n_mol == e->h_nmol[t] &&
                              std::memcmp(e->frozen_ref[t].data(), st, 3 * seg * sizeof(double)) == 0;
            set_same(replica, same ? 1 : 0);
        }
    }
    e->frames_ok[(size_t)replica * tp.n_res + t] = 0;       // sites given without com / offsets (set_frames sets it again)
    e->in_range[(size_t)replica * tp.n_res + t] = sites_in_range(e, sites, n_mol * n1) ? 1 : 0;
    HIP_TRY(hipMemcpy(e->d_nmol + replica * tp.n_res + t, &n_mol, sizeof(int), hipMemcpyHostToDevice));
    return MGPU_OK;
}

int mgpu_replica_get_molecules(mgpu_engine *e, int replica, int t, int *n_mol, double *sites) {
    int rc = check_replica_t(e, replica, t);
    if (rc) return rc;
    if ((rc = use_device(e))) return rc;
    const Topo &tp = e->tp;
    const int n1 = tp.n1[t], cap = tp.cap[t], nm = e->h_nmol[replica * tp.n_res + t];
    if (n_mol) *n_mol = nm;
    if (!sites || nm == 0) return MGPU_OK;
    const size_t seg = (size_t)n1 * cap;
    rc = e->h_stage.reserve(3 * seg * sizeof(double));
    if (rc) return rc;
    double *st = (double *)e->h_stage.p;
    if ((rc = sync_all_lanes(e))) return rc;
    for (int d = 0; d < 3; ++d)
        HIP_TRY(hipMemcpy(st + d * seg, e->d_pos + ((size_t)replica * 3 + d) * tp.n_cap_atoms + tp.seg_off[t],
                          seg * sizeof(double), hipMemcpyDeviceToHost));
    const int *perm = e->site_perm[t].data();
    for (int m = 0; m < nm; ++m)
        for (int a = 0; a < n1; ++a) {
            const size_t j = tp.site_major[t] ? (size_t)m * n1 + perm[a] : (size_t)a * cap + m;
            for (int d = 0; d < 3; ++d) sites[((size_t)m * n1 + a) * 3 + d] = st[d * seg + j];
        }
    return MGPU_OK;
}

// Molecule frames of one residue type as the reference keeps them: com[n_mol][3] = primary%mol_com, off[n_mol][n1][3] =
// primary%site_offset (simulation_state.f90:115-116).  The sites com + off are formed here exactly as the reference forms
// them before every use (geometry_utils.f90:379-382) and uploaded as mgpu_replica_set_molecules does.
int mgpu_replica_set_frames(mgpu_engine *e, int replica, int t, int n_mol, const double *com, const double *off) {
    int rc = check_replica_t(e, replica, t);
    if (rc) return rc;
    if (n_mol < 0 || n_mol > e->tp.cap[t]) return set_error(MGPU_ERR_CAPACITY, "n_mol exceeds the residue type's mol_capacity");
    if (n_mol > 0 && (!com || !off)) return set_error(MGPU_ERR_INVALID_ARG, "set_frames: null argument");
    if (e->frozen[t]) return set_error(MGPU_ERR_INVALID_ARG, "set_frames: frozen (inactive framework) residue types carry no frames");
    if ((rc = use_device(e))) return rc;
    const Topo &tp = e->tp;
    const int n1 = tp.n1[t], cap = tp.cap[t];
    std::vector<double> sites((size_t)n_mol * n1 * 3);
    // "tight": every centre in the cell and every offset's Euclidean norm within 0.24 of the SHORTEST edge -- a bound no
    // rotation about a Cartesian axis can break (trial_build_kernel rotates offsets; a per-component bound would let a
    // rotated component grow by sqrt(2) and mix axes of different lengths)
    bool tight = true;
    const double r_max = 0.24 * std::min(e->bx.L[0], std::min(e->bx.L[1], e->bx.L[2]));
    for (int m = 0; m < n_mol; ++m) {
        for (int d = 0; d < 3; ++d)
            tight = tight && std::fabs(com[(size_t)m * 3 + d] - e->bx.ctr[d]) <= 0.5 * e->bx.L[d];
        for (int a = 0; a < n1; ++a) {
            double o2 = 0.0;
            for (int d = 0; d < 3; ++d) {
                const double o = off[((size_t)m * n1 + a) * 3 + d];
                sites[((size_t)m * n1 + a) * 3 + d] = com[(size_t)m * 3 + d] + o;
                o2 += o * o;
            }
            tight = tight && o2 <= r_max * r_max;
        }
    }
    if ((rc = mgpu_replica_set_molecules(e, replica, t, n_mol, sites.data()))) return rc;
    if (!e->d_com) {
        const size_t R = e->n_replicas;
        HIP_TRY(hipMalloc(&e->d_com, R * 3 * tp.n_mol_slots * sizeof(double)));
        HIP_TRY(hipMemset(e->d_com, 0, R * 3 * tp.n_mol_slots * sizeof(double)));
        HIP_TRY(hipMalloc(&e->d_off, R * 3 * tp.n_cap_atoms * sizeof(double)));
        HIP_TRY(hipMemset(e->d_off, 0, R * 3 * tp.n_cap_atoms * sizeof(double)));
        e->tp.com = e->d_com;
        e->tp.off = e->d_off;
        e->chain.topo_stale = true;
    }
    const size_t seg = (size_t)n1 * cap;
    if ((rc = e->h_stage.reserve(3 * (seg + cap) * sizeof(double)))) return rc;
    double *st = (double *)e->h_stage.p, *sc = st + 3 * seg;
    std::memset(st, 0, 3 * (seg + cap) * sizeof(double));
    for (int m = 0; m < n_mol; ++m) {
        for (int d = 0; d < 3; ++d) sc[(size_t)d * cap + m] = com[(size_t)m * 3 + d];
        for (int a = 0; a < n1; ++a) {
            const size_t j = tp.site_major[t] ? (size_t)m * n1 + a : (size_t)a * cap + m;
            for (int d = 0; d < 3; ++d) st[d * seg + j] = off[((size_t)m * n1 + a) * 3 + d];
        }
    }
    for (int d = 0; d < 3; ++d) {
        HIP_TRY(hipMemcpy(e->d_off + ((size_t)replica * 3 + d) * tp.n_cap_atoms + tp.seg_off[t], st + d * seg, seg * sizeof(double),
                          hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(e->d_com + ((size_t)replica * 3 + d) * tp.n_mol_slots + tp.mol_off[t], sc + (size_t)d * cap, cap * sizeof(double),
                          hipMemcpyHostToDevice));
    }
    e->frames_ok[(size_t)replica * tp.n_res + t] = 1;
    e->frames_tight[(size_t)replica * tp.n_res + t] = tight ? 1 : 0;
    return MGPU_OK;
}

int mgpu_replica_get_frames(mgpu_engine *e, int replica, int t, int *n_mol, double *com, double *off) {
    int rc = check_replica_t(e, replica, t);
    if (rc) return rc;
    if ((rc = use_device(e))) return rc;
    const Topo &tp = e->tp;
    const int n1 = tp.n1[t], cap = tp.cap[t], nm = e->h_nmol[replica * tp.n_res + t];
    if (n_mol) *n_mol = nm;
    if ((!com && !off) || nm == 0) return MGPU_OK;
    if (!e->d_com || !e->frames_ok[(size_t)replica * tp.n_res + t])
        return set_error(MGPU_ERR_STATE, "get_frames: the engine holds no frames for this replica / residue type");
    const size_t seg = (size_t)n1 * cap;
    if ((rc = e->h_stage.reserve(3 * (seg + cap) * sizeof(double)))) return rc;
    double *st = (double *)e->h_stage.p, *sc = st + 3 * seg;
    if ((rc = sync_all_lanes(e))) return rc;
    for (int d = 0; d < 3; ++d) {
        HIP_TRY(hipMemcpy(st + d * seg, e->d_off + ((size_t)replica * 3 + d) * tp.n_cap_atoms + tp.seg_off[t], seg * sizeof(double),
                          hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(sc + (size_t)d * cap, e->d_com + ((size_t)replica * 3 + d) * tp.n_mol_slots + tp.mol_off[t], cap * sizeof(double),
                          hipMemcpyDeviceToHost));
    }
    for (int m = 0; m < nm; ++m) {
        if (com) for (int d = 0; d < 3; ++d) com[(size_t)m * 3 + d] = sc[(size_t)d * cap + m];
        if (off)
            for (int a = 0; a < n1; ++a) {
                const size_t j = tp.site_major[t] ? (size_t)m * n1 + a : (size_t)a * cap + m;
                for (int d = 0; d < 3; ++d) off[((size_t)m * n1 + a) * 3 + d] = st[d * seg + j];
            }
    }
    return MGPU_OK;
}

int mgpu_replica_num_molecules(const mgpu_engine *e, int replica, int t, int *n_mol) {
    int rc = check_replica_t(e, replica, t);
    if (rc) return rc;
    if (!n_mol) return set_error(MGPU_ERR_INVALID_ARG, "null n_mol");
    *n_mol = e->h_nmol[replica * e->tp.n_res + t];
    return MGPU_OK;
}

int mgpu_replica_set_num_molecules(mgpu_engine *e, int replica, int t, int n_mol) {
    int rc = check_replica_t(e, replica, t);
    if (rc) return rc;
    if (n_mol < 0 || n_mol > e->tp.cap[t]) return set_error(MGPU_ERR_CAPACITY, "n_mol exceeds mol_capacity");
    if ((rc = use_device(e))) return rc;
    if ((rc = sync_all_lanes(e))) return rc;
    if (e->frozen[t]) {
        char &f = e->frozen_same[(size_t)replica * e->tp.n_res + t];
        e->frozen_diff[t] += f ? 1 : 0;
        f = 0;
    }
    // a larger count exposes slots the range flag was never computed for (zero-filled, or stale coordinates)
    if (n_mol > e->h_nmol[replica * e->tp.n_res + t]) e->in_range[(size_t)replica * e->tp.n_res + t] = 0;
    e->h_nmol[replica * e->tp.n_res + t] = n_mol;
    HIP_TRY(hipMemcpy(e->d_nmol + replica * e->tp.n_res + t, &n_mol, sizeof(int), hipMemcpyHostToDevice));
    return MGPU_OK;
}

int mgpu_replica_copy(mgpu_engine *e, int dst, int src) {
    int rc = check_replica_t(e, dst, 0);
    if (rc) return rc;
    if ((rc = check_replica_t(e, src, 0))) return rc;
    if (dst == src) return MGPU_OK;
    if ((rc = use_device(e))) return rc;
    if ((rc = sync_all_lanes(e))) return rc;
    const Topo &tp = e->tp;
    HIP_TRY(hipMemcpyAsync(e->d_pos + (size_t)dst * 3 * tp.n_cap_atoms, e->d_pos + (size_t)src * 3 * tp.n_cap_atoms,
                           (size_t)3 * tp.n_cap_atoms * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(e->d_nmol + dst * tp.n_res, e->d_nmol + src * tp.n_res, tp.n_res * sizeof(int),
                           hipMemcpyDeviceToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(e->d_A + (size_t)dst * e->n_slots, e->d_A + (size_t)src * e->n_slots, e->n_slots * sizeof(double2),
                           hipMemcpyDeviceToDevice, e->stream));
    if (e->d_com) {
        HIP_TRY(hipMemcpyAsync(e->d_com + (size_t)dst * 3 * tp.n_mol_slots, e->d_com + (size_t)src * 3 * tp.n_mol_slots,
                               (size_t)3 * tp.n_mol_slots * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
        HIP_TRY(hipMemcpyAsync(e->d_off + (size_t)dst * 3 * tp.n_cap_atoms, e->d_off + (size_t)src * 3 * tp.n_cap_atoms,
                               (size_t)3 * tp.n_cap_atoms * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
    }
    for (int t = 0; t < tp.n_res; ++t) {
        e->h_nmol[dst * tp.n_res + t] = e->h_nmol[src * tp.n_res + t];
        e->in_range[(size_t)dst * tp.n_res + t] = e->in_range[(size_t)src * tp.n_res + t];
        e->frames_ok[(size_t)dst * tp.n_res + t] = e->frames_ok[(size_t)src * tp.n_res + t];
        if (e->frozen[t]) {
            // a copy of replica 0 (or of a replica equal to it) equals replica 0; overwriting replica 0 itself loses the reference
            const char v = dst == 0 ? 0 : e->frozen_same[(size_t)src * tp.n_res + t];
            char &f = e->frozen_same[(size_t)dst * tp.n_res + t];
            e->frozen_diff[t] += (f ? 1 : 0) - (v ? 1 : 0);
            f = v;
            if (dst == 0)
                for (int r = 1; r < e->n_replicas; ++r) {
                    char &g = e->frozen_same[(size_t)r * tp.n_res + t];
                    e->frozen_diff[t] += g ? 1 : 0;
                    g = 0;
                }
        }
        e->frames_tight[(size_t)dst * tp.n_res + t] = e->frames_tight[(size_t)src * tp.n_res + t];
    }
    return sync_stream(e);
}

int mgpu_replica_replace_molecule(mgpu_engine *e, int replica, int t, int m_dst, int m_src) {
    int rc = check_replica_t(e, replica, t);
    if (rc) return rc;
    const Topo &tp = e->tp;
    if (m_dst < 0 || m_src < 0 || m_dst >= tp.cap[t] || m_src >= tp.cap[t])
        return set_error(MGPU_ERR_INVALID_ARG, "molecule slot out of range");
    if (m_dst == m_src) return MGPU_OK;
    if ((rc = use_device(e))) return rc;
    if ((rc = sync_all_lanes(e))) return rc;
    e->frames_ok[(size_t)replica * tp.n_res + t] = 0;       // a slot copy of the sites only
    frozen_changed(e, replica, t);
    const int n1 = tp.n1[t];
    for (int d = 0; d < 3; ++d) {
        double *base = e->d_pos + ((size_t)replica * 3 + d) * tp.n_cap_atoms + tp.seg_off[t];
        if (tp.site_major[t]) {
            HIP_TRY(hipMemcpyAsync(base + (size_t)m_dst * n1, base + (size_t)m_src * n1, n1 * sizeof(double),
                                   hipMemcpyDeviceToDevice, e->stream));
        } else {
            HIP_TRY(hipMemcpy2DAsync(base + m_dst, tp.cap[t] * sizeof(double), base + m_src, tp.cap[t] * sizeof(double),
                                     sizeof(double), n1, hipMemcpyDeviceToDevice, e->stream));
        }
    }
    return sync_stream(e);
}

// ---- structure factor ------------------------------------------------------------------------

int mgpu_init_structure_factor(mgpu_engine *e, int replica, int mode) {
    int rc = check_replica_t(e, replica, 0);
    if (rc) return rc;
    if ((rc = use_device(e))) return rc;
    if ((rc = sync_all_lanes(e))) return rc;
    double2 *A = e->d_A + (size_t)replica * e->n_slots;
    if (mode == 0) {
        HIP_TRY(hipMemsetAsync(A, 0, e->n_slots * sizeof(double2), e->stream));
    } else {
        if ((rc = launch_sfactor(e, replica, A))) return rc;
    }
    return sync_stream(e);
}

int mgpu_get_structure_factor(mgpu_engine *e, int replica, double *a) {
    int rc = check_replica_t(e, replica, 0);
    if (rc) return rc;
    if (!a) return set_error(MGPU_ERR_INVALID_ARG, "null buffer");
    if ((rc = use_device(e))) return rc;
    if ((rc = sync_all_lanes(e))) return rc;
    // the device keeps A(k) in task order; hand it out in the reference's k order
    std::vector<double2> slots(e->n_slots);
    HIP_TRY(hipMemcpy(slots.data(), e->d_A + (size_t)replica * e->n_slots, e->n_slots * sizeof(double2), hipMemcpyDeviceToHost));
    for (int k = 0; k < e->nk; ++k) { a[2 * k] = slots[e->kslot[k]].x; a[2 * k + 1] = slots[e->kslot[k]].y; }
    return MGPU_OK;
}

int mgpu_set_structure_factor(mgpu_engine *e, int replica, const double *a) {
    int rc = check_replica_t(e, replica, 0);
    if (rc) return rc;
    if (!a) return set_error(MGPU_ERR_INVALID_ARG, "null buffer");
    if ((rc = use_device(e))) return rc;
    if ((rc = sync_all_lanes(e))) return rc;
    std::vector<double2> slots(e->n_slots, make_double2(0.0, 0.0));
    for (int k = 0; k < e->nk; ++k) slots[e->kslot[k]] = make_double2(a[2 * k], a[2 * k + 1]);
    HIP_TRY(hipMemcpy(e->d_A + (size_t)replica * e->n_slots, slots.data(), e->n_slots * sizeof(double2), hipMemcpyHostToDevice));
    return MGPU_OK;
}

int mgpu_structure_factor_add(mgpu_engine *e, int replica, int t, const double *sites) {
    int rc = check_replica_t(e, replica, t);
    if (rc) return rc;
    if (!sites) return set_error(MGPU_ERR_INVALID_ARG, "structure_factor_add: null sites");
    if ((rc = use_device(e))) return rc;
    if ((rc = mgpu_synchronize(e))) return rc;
    Lane &ln = e->lanes[0];
    const int n1 = e->tp.n1[t];
    RecipItem it{replica, t, -1, MGPU_FOURIER_ADD, 0, 0};
    if ((rc = ln.d_items2.reserve(sizeof(RecipItem)))) return rc;
    HIP_TRY(hipMemcpyAsync(ln.d_items2.p, &it, sizeof(RecipItem), hipMemcpyHostToDevice, ln.stream));
    if ((rc = upload_sites(e, sites, 1, n1, &t))) return rc;
    if ((rc = launch_recip(e, ln, (const RecipItem *)ln.d_items2.p, 1, n1, n1, true, e->d_A, nullptr))) return rc;
    return sync_stream(e);
}

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
    hipLaunchKernelGGL(intra_kernel, dim3((n + 63) / 64), dim3(64), 0, e->stream, e->tp, e->bx, e->d_pos, e->d_res_q,
                       (const PairItem *)e->d_items.p, n, (const double *)e->d_sites.p, site_stride, (double *)e->d_out.p);
    HIP_TRY(hipGetLastError());
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
// (DecideItem, mgpu_kernels.h); accept_u[n] = the test's uniform numbers, accept_pref[n] = its prefactors
struct TrialDecide {
    const double *u, *pref;
    double temperature;
};
static int trial_submit_impl(mgpu_engine *e, Lane &ln, int n, const int *replica, const int *t, const int *m,
                             const int *kind, const double *sites, int site_stride, const TrialBuild *build = nullptr,
                             const TrialDecide *decide = nullptr) {
    if (ln.n_submitted != 0) return set_error(MGPU_ERR_STATE, "trial_submit: the lane still holds an un-waited trial");
    int rc;
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
        const bool fz = k == MGPU_MOVE && e->pair_fuse && !e->bx.triclinic && n1 <= e->pair_fuse_max;
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
    if (n_intra) {
        hipLaunchKernelGGL(intra_kernel, dim3((n_intra + 63) / 64), dim3(64), 0, ln.stream, e->tp, e->bx, e->d_pos, e->d_res_q,
                           d_iit, n_intra, (const double *)ln.d_sites.p, site_stride, d_in);
        HIP_TRY(hipGetLastError());
    }
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
            const int before = parts > 1 ? __atomic_exchange_n(&mark, stamp, __ATOMIC_RELAXED) : mark;
            mark = stamp;
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


// ---- single-chain windows --------------------------------------------------------------------

// largest window the engine accepts, 0 where the one-launch path does not apply (triclinic box, molecules of more than
// kMaxFusedSitesWide sites among the active types, per-k reciprocal form)
static int chain_max_candidates(const mgpu_engine *e) {
    if (e->bx.triclinic) return 0;
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

int mgpu_set_host_team(mgpu_engine *e, int n_threads) {
    if (!e || n_threads < 1) return set_error(MGPU_ERR_INVALID_ARG, "set_host_team: bad argument");
    e->host_team = std::min(n_threads, kMaxHostParts);
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
        HIP_TRY(hipMalloc((void **)&ch.d_ticket, sizeof(int)));
        HIP_TRY(hipMemset(ch.d_ticket, 0, sizeof(int)));
        HIP_TRY(hipMalloc((void **)&ch.d_topo, sizeof(Topo)));
        HIP_TRY(hipDeviceSynchronize());
    }
    if (ch.topo_stale) {
        if ((rc = sync_lane(e, ln))) return rc;
        HIP_TRY(hipMemcpy(ch.d_topo, &e->tp, sizeof(Topo), hipMemcpyHostToDevice));
        ch.topo_stale = false;
    }
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
    g.res = ch.d_res; g.partials = ch.d_part; g.ticket = ch.d_ticket;
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
    hipLaunchKernelGGL((chain_window_kernel<FL, FW>), dim3(grid), dim3(kChainBlock), lds, ln.stream, ch.d_topo, e->bx, e->d_pos, e->d_nmol, \
                       e->d_res_q, e->d_res_atype, e->d_pair_tab, e->d_coul_tab, e->d_trj, e->d_tw, e->n_rtasks, e->d_rrows, e->n_rrows, \
                       e->d_A, g)
    if (e->pair_flat) { if (ff) MGPU_LAUNCH_CHAIN(true, true); else MGPU_LAUNCH_CHAIN(true, false); }
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

// ---- static energy ---------------------------------------------------------------------------

int mgpu_system_energy(mgpu_engine *e, int replica, double out[6]) {
    int rc = check_replica_t(e, replica, 0);
    if (rc) return rc;
    if (!out) return set_error(MGPU_ERR_INVALID_ARG, "null out");
    if ((rc = use_device(e))) return rc;
    if ((rc = sync_all_lanes(e))) return rc;
    const Topo &tp = e->tp;
    // ComputePairwiseEnergy (energy_utils.f90:83-115): one ordered item per molecule, results
    // accumulated on the host in the reference's (type, molecule) order.
    std::vector<PairItem> items;
    for (int t = 0; t < tp.n_res; ++t)
        for (int m = 0; m < e->h_nmol[replica * tp.n_res + t]; ++m) items.push_back(PairItem{replica, t, m, -1, 1});
    const int n = (int)items.size();
    double e_nc = 0.0, e_c = 0.0, e_intra = 0.0, e_recip = 0.0;
    std::vector<double> h_lj(n), h_c(n), h_intra(n);
    if ((rc = e->d_out.reserve(((size_t)3 * n + 1) * sizeof(double)))) return rc;
    if (n > 0) {
        if ((rc = e->d_items.reserve(n * sizeof(PairItem)))) return rc;
        HIP_TRY(hipMemcpyAsync(e->d_items.p, items.data(), n * sizeof(PairItem), hipMemcpyHostToDevice, e->stream));
        double *d_lj = (double *)e->d_out.p, *d_c = d_lj + n, *d_in = d_c + n;
        const int nsplit = e->pair_nsplit;
        if ((rc = launch_pair(e, e->lanes[0], (const PairItem *)e->d_items.p, n, 0, 1, nsplit, d_lj, d_c, true))) return rc;
        hipLaunchKernelGGL(intra_kernel, dim3((n + 63) / 64), dim3(64), 0, e->stream, e->tp, e->bx, e->d_pos, e->d_res_q,
                           (const PairItem *)e->d_items.p, n, (const double *)nullptr, 1, d_in);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h_lj.data(), d_lj, n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipMemcpyAsync(h_c.data(), d_c, n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipMemcpyAsync(h_intra.data(), d_in, n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    }
    // ComputeEwaldRecip (energy_utils.f90:270-286): S(k) into scratch, then sum ff W |S|^2
    if ((rc = launch_sfactor(e, replica, e->d_S))) return rc;
    RecipItem rit{0, 0, -1, MGPU_NONE, -1, 0};
    if ((rc = e->d_items2.reserve(sizeof(RecipItem)))) return rc;
    HIP_TRY(hipMemcpyAsync(e->d_items2.p, &rit, sizeof(RecipItem), hipMemcpyHostToDevice, e->stream));
    double *d_u = (double *)e->d_out.p + (size_t)3 * n;
    if ((rc = launch_recip(e, e->lanes[0], (const RecipItem *)e->d_items2.p, 1, 1, 1, false, e->d_S, d_u))) return rc;
    HIP_TRY(hipMemcpyAsync(&e_recip, d_u, sizeof(double), hipMemcpyDeviceToHost, e->stream));
    if ((rc = sync_stream(e))) return rc;
    int i = 0;
    for (int t = 0; t < tp.n_res; ++t)
        for (int m = 0; m < e->h_nmol[replica * tp.n_res + t]; ++m, ++i) {
            e_nc = e_nc + h_lj[i];
            e_c = e_c + h_c[i];
            // ComputeTotalIntraResidueCoulombEnergy (energy_utils.f90:55-81): active types only
            if (e->is_active[t] == 1) e_intra = e_intra + h_intra[i];
        }
    // ComputeEwaldSelf (energy_utils.f90:307-330)
    double e_self = 0.0;
    for (int t = 0; t < tp.n_res; ++t) {
        double s = self_energy_host(e, t);
        s = s * (double)e->h_nmol[replica * tp.n_res + t];
        e_self = e_self + s;
    }
    out[0] = e_nc; out[1] = e_c; out[2] = e_recip; out[3] = e_self; out[4] = e_intra;
    out[5] = e_recip + e_nc + e_c + e_self + e_intra;  // energy_utils.f90:32-33
    return MGPU_OK;
}

// ---- test hook --------------------------------------------------------------------------------

int mgpu_phase_factors(mgpu_engine *e, int n, const double *theta, const int *k, double *cos_out, double *sin_out) {
    if (!e) return set_error(MGPU_ERR_INVALID_ARG, "null engine");
    if (n < 0 || (n > 0 && (!theta || !k || !cos_out || !sin_out))) return set_error(MGPU_ERR_INVALID_ARG, "mgpu_phase_factors: bad arguments");
    if (n == 0) return MGPU_OK;
    int rc = use_device(e);
    if (rc) return rc;
    double *d_theta = nullptr;
    int *d_k = nullptr;
    double2 *d_out = nullptr;
    std::vector<double2> h((size_t)n);
    hipError_t err = hipMalloc(&d_theta, (size_t)n * sizeof(double));
    if (err == hipSuccess) err = hipMalloc(&d_k, (size_t)n * sizeof(int));
    if (err == hipSuccess) err = hipMalloc(&d_out, (size_t)n * sizeof(double2));
    if (err == hipSuccess) err = hipMemcpy(d_theta, theta, (size_t)n * sizeof(double), hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMemcpy(d_k, k, (size_t)n * sizeof(int), hipMemcpyHostToDevice);
    if (err == hipSuccess) {
        phase_factors_kernel<<<(n + 255) / 256, 256>>>(n, d_theta, d_k, d_out);
        err = hipGetLastError();
    }
    if (err == hipSuccess) err = hipMemcpy(h.data(), d_out, (size_t)n * sizeof(double2), hipMemcpyDeviceToHost);
    (void)hipFree(d_theta); (void)hipFree(d_k); (void)hipFree(d_out);
    if (err != hipSuccess) return set_error(MGPU_ERR_HIP, hipGetErrorString(err));
    for (int i = 0; i < n; ++i) { cos_out[i] = h[i].x; sin_out[i] = h[i].y; }
    return MGPU_OK;
}

// ---- measurement -----------------------------------------------------------------------------

int mgpu_synchronize(mgpu_engine *e) {
    if (!e) return set_error(MGPU_ERR_INVALID_ARG, "null engine");
    int rc = use_device(e);
    if (rc) return rc;
    return sync_all_lanes(e);
}

int mgpu_profile_enable(mgpu_engine *e, int on) {
    if (!e) return set_error(MGPU_ERR_INVALID_ARG, "null engine");
    int rc = mgpu_synchronize(e);
    if (rc) return rc;
    e->profiling = on != 0;
    if (!e->profiling) return MGPU_OK;
    // Pay the one-time costs here, not inside the caller's timed region: the first dispatch that carries
    // start / stop events switches the stream's queue into profiling mode (measured: ~7 ms on the first
    // such launch), and the event pool is filled for every launch the lanes can have in flight.
    for (auto &ln : e->lanes) {
        while (ln.ev_pool.size() < 16) {
            hipEvent_t ev;
            HIP_TRY(hipEventCreate(&ev));
            ln.ev_pool.push_back(ev);
        }
        hipEvent_t a = ln.ev_pool.back(); ln.ev_pool.pop_back();
        hipEvent_t b = ln.ev_pool.back(); ln.ev_pool.pop_back();
        hipExtLaunchKernelGGL(prime_kernel, dim3(1), dim3(64), 0, ln.stream, a, b, 0, (const int *)e->d_nmol);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(ln.stream));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, a, b));
        ln.ev_pool.push_back(a);
        ln.ev_pool.push_back(b);
    }
    return MGPU_OK;
}

int mgpu_profile_reset(mgpu_engine *e) {
    if (!e) return set_error(MGPU_ERR_INVALID_ARG, "null engine");
    int rc = mgpu_synchronize(e);
    if (rc) return rc;
    for (auto &ln : e->lanes)
        for (auto &p : ln.prof) p = ProfileSlot{};
    return MGPU_OK;
}

int mgpu_profile_get(mgpu_engine *e, int kernel, long long *launches, double *total_ms) {
    if (!e || kernel < 0 || kernel >= MGPU_KERNEL_COUNT) return set_error(MGPU_ERR_INVALID_ARG, "profile_get: bad argument");
    int rc = mgpu_synchronize(e);
    if (rc) return rc;
    long long n = 0;
    double ms = 0.0;
    for (auto &ln : e->lanes) { n += ln.prof[kernel].launches; ms += ln.prof[kernel].total_ms; }
    if (launches) *launches = n;
    if (total_ms) *total_ms = ms;
    return MGPU_OK;
}

}  // extern "C"
