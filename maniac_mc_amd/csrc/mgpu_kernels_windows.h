// One launch per window: chain_window_kernel (one chain, K steps) and farm_window_kernel (a lock step of a farm of chains).
// (one of the four parts of mgpu_kernels.h: include that header, not this file)
#ifndef MGPU_KERNELS_WINDOWS_H
#define MGPU_KERNELS_WINDOWS_H

#include "mgpu_kernels_pair.h"
#include "mgpu_kernels_recip.h"

namespace mgpu {

// ------------------------------------------------------------------------------------------
// Single-chain windows: ONE launch evaluates, decides and commits a window of trial steps of one chain.
//
// A single Markov chain is latency-bound: with one engine call per step the host pays an upload, two or three kernel
// launches, a download, a stream synchronisation and a commit launch for ~10 us of arithmetic (profiles/r04/chain_latency.md).
// mc_chain.f90 therefore hands over a WINDOW of up to kChainMaxCand consecutive steps drawn under the assumption that
// every one of them is rejected (all of them are then trials from the same state) together with each step's acceptance
// draw, and this kernel does everything the steps need in one launch:
//   * workgroups [0, n)  ("k role"): the reciprocal-space sweep of candidate c (recip_rows_* pieces, old and new energy
//     from one pass over A(k)); one spare thread computes the intra-molecular term of an insertion / deletion meanwhile;
//   * the other workgroups ("pair role"): one wave per (candidate state, split) work unit of the pair sweep
//     (pair_sweep_item / pair_flat_item, single-state items; the engine's nsplit);
//   * every workgroup publishes its results to device memory, fences and takes a ticket; the workgroup that draws the
//     LAST ticket sums the split partials in split order (exactly the host's order in trial_wait), forms each step's
//     old / new totals as ComputeOldEnergy / ComputeNewEnergy and the move drivers form them
//     (monte_carlo_utils.f90:275-395, create_molecule.f90:100-112, delete_molecule.f90:100-142), walks the window in
//     order applying mc_acceptance_probability (monte_carlo_utils.f90:184-226) with the host's draws, writes energies +
//     outcome straight into pinned host memory (the host polls a tag: no download, no stream synchronisation) and then
//     commits the first accepted step itself (tables rebuilt, A <- A + delta, coordinates / count) while the host
//     already resolves the window.
// The rule on the device uses OCML's exp, the host (and the reference) glibc's: a step whose draw lies within
// `margin` (relative) of its acceptance probability -- or whose probability is not a number -- is left UNDECIDED: the
// device stops there, commits nothing from that step on, and the host decides it with its own exp.  Every decision the
// device does take is therefore the host's decision, bit for bit.
// kind 2 with link >= 0: the reference's deletion exactly as written (monte_carlo_utils.f90:301-309): the new
// reciprocal energy is the creation-kind energy of row `link` (the molecule RemoveMolecule swaps into the slot) and an
// accepted step adds THAT molecule's terms to A(k) while the coordinates lose slot m; row `link` itself is energy-only
// (link = -2).  Orthorhombic boxes, row-form k sweep, molecules of <= kMaxFusedSitesWide sites.
// ------------------------------------------------------------------------------------------
constexpr int kChainMaxCand = 16;
constexpr int kChainBlock = kPairBlock;          // 512 threads: 8 pair waves; the k role uses the first kBlock of them
constexpr int kChainStamps = 8;                  // stage time stamps per role (k role of candidate 0, first pair workgroup, resolver)
struct ChainResult {                             // what the k role of candidate c leaves for the resolving workgroup
    double u_old, u_new, intra;
};
// The whole window travels in the KERNEL ARGUMENTS (3.6 KB with Topo and BoxDev, under the 4 KB limit): no upload, no
// staging block, and no read of host memory on the kernel's critical path.
struct ChainArgs {
    ChainResult *res;                            // [n] device scratch
    double2 *partials;                           // [n_ent * nsplit] device scratch
    int *ticket;                                 // device counter, 0 between launches
    double2 *alt;                                // [n][n_slots] device scratch: candidate c's k role leaves A + delta_c here
    double *host_out;                            // pinned host: [n][10] energies | first | undecided (ints) | stage stamps
    unsigned long long *host_tag;                // pinned host: window sequence number, written last
    unsigned long long seq;
    int n, n_ent, nsplit, replica;
    int stamps;                                  // 1: record wall_clock64() at the stages (mgpu_chain_set_timing)
    double temperature, e_recip, margin;
    double self_of_type[kMaxRes];                // ewald_self per residue type
    int t[kChainMaxCand], m[kChainMaxCand];
    signed char kind[kChainMaxCand];
    signed char link[kChainMaxCand];             // -1 none, >= 0 companion row of an as-written deletion, -2 energy-only row
    unsigned char ent_c[2 * kChainMaxCand], ent_new[2 * kChainMaxCand];   // pair entries: candidate, 0 = resident (old) / 1 = candidate row (new)
    signed char ent_old_of[kChainMaxCand], ent_new_of[kChainMaxCand];     // per candidate: its old / new pair entry, -1 none
    double u[kChainMaxCand], pref[kChainMaxCand];                          // acceptance draw, prefactor (1; phi V / N; (N + 1) / (phi V))
    double sites[kChainMaxCand][kMaxFusedSitesWide][3];                    // candidate rows, site stride kMaxFusedSitesWide
};
static_assert(sizeof(BoxDev) + sizeof(ChainArgs) + 160 <= 4096, "a window must fit the kernel-argument segment");

// (The topology comes through a pointer: a by-value Topo indexed by a residue type that is itself loaded -- g.t[c] -- makes
//  the compiler copy all 664 bytes of it into every lane's scratch at kernel start: measured 4 us per window.)
// TRI: triclinic box -- the pair role runs the register-site sweeps with ComputeDistance's image search (the batched path's
// kernels for such boxes: the same sums); everything else of a window is the same (the phase tables take the box's reciprocal
// matrix either way).
template <bool FLAT, bool FASTW, bool TRI = false>
__global__ __launch_bounds__(kChainBlock, 1) void chain_window_kernel(
    const Topo *__restrict__ tpp, BoxDev bx, double *__restrict__ pos, int *__restrict__ nmol, const double *__restrict__ res_q,
    const int *__restrict__ res_atype, const double2 *__restrict__ pair_tab, const char *__restrict__ coul_tab_g,
    const int *__restrict__ trj, const double2 *__restrict__ tw, int n_tasks, const RecipRow *__restrict__ rows, int n_rows,
    double2 *__restrict__ A_base, const ChainArgs g) {
    extern __shared__ __attribute__((aligned(16))) char s_dyn[];      // Coulomb table | phase tables | partials staging
    __shared__ double2 s_pair[kMaxTypes * kMaxTypes];
    __shared__ int4 s_grp[kMaxGrp];
    __shared__ int4 s_plane[FLAT ? kPairWaves * kFlatMaxPlanes : 1];
    __shared__ double s_red[2 * kWavesPerBlock];
    __shared__ double s_ent[2 * 2 * kChainMaxCand];                    // reduced pair entries {lj, cc}
    __shared__ ChainResult s_res[kChainMaxCand];
    __shared__ int s_verdict[kChainMaxCand];
    __shared__ int s_flag;

    const Topo &tp = *tpp;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = g.n;
    double2 *A = A_base + (size_t)g.replica * bx.n_slots;
    const double *cand_sites = &g.sites[0][0][0];
    // stage stamps (100 MHz wall clock): role 0 = the k role of candidate 0, 1 = the first pair workgroup, 2 = the resolver
    long long *stamp = reinterpret_cast<long long *>(g.host_out + 10 * kChainMaxCand + 2);
    const int my_role = g.stamps ? ((int)blockIdx.x == 0 ? 0 : ((int)blockIdx.x == n ? 1 : -1)) : -1;
    auto mark = [&](int role, int i) {
        if (tid == 0 && role >= 0) stamp[role * kChainStamps + i] = wall_clock64();
    };
    mark(my_role, 0);

    if ((int)blockIdx.x < n) {
        // ---------------- k role: candidate c
        const int c = blockIdx.x;
        const int kind = g.kind[c], link = g.link[c];
        RecipItem it{g.replica, g.t[c], g.m[c], kind, c, 0, 0};
        const RecipLds v = recip_lds_view(tp, bx, it, n_rows, reinterpret_cast<double2 *>(s_dyn));
        const bool active = tid < kBlock;
        RecipInFlight<kRecipTaskChunk> inflight;
        recip_rows_tables(tp, bx, pos, res_q, rows, n_rows, it, &g.sites[c][0][0], v, tid, active,
                          [&] { recip_rows_prefetch<false>(inflight, trj, tw, n_tasks, A, tid); });
        mark(my_role, 1);
        double acc = 0.0, acc0 = 0.0;
        // (A + delta of this candidate goes to its own buffer with write-through stores: the resolving workgroup commits an
        //  accepted step by COPYING that buffer -- round 5; until then it rebuilt the phase tables and made a second pass:
        //  6.4 us of GPU time per window, hidden behind the host's turn-round for one chain, not for several sharing a GPU)
        if (active) recip_rows_pass<false, true, kRecipTaskChunk, 2>(v, trj, tw, n_tasks, A, tid, inflight, acc, acc0, g.alt + (size_t)c * bx.n_slots);
        if (tid == kBlock && link != -2 && (kind == 1 || kind == 2)) {
            // ComputeIntraResidueRealCoulombEnergySingleMol of the inserted (candidate row) / deleted (resident) molecule
            const PairItem pit{g.replica, it.t, it.m, kind == 1 ? c : -1, 0};
            __hip_atomic_store(&g.res[c].intra, intra_energy(tp, bx, pos, res_q, pit, cand_sites, kMaxFusedSitesWide), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        if (active) {
            acc = wave_sum(acc);
            acc0 = wave_sum(acc0);
            if (lane == 0) { s_red[2 * wave] = acc; s_red[2 * wave + 1] = acc0; }
        }
        __syncthreads();
        if (tid == 0) {
            double u = 0.0, u0 = 0.0;
            for (int wv = 0; wv < kWavesPerBlock; ++wv) { u += s_red[2 * wv]; u0 += s_red[2 * wv + 1]; }
            __hip_atomic_store(&g.res[c].u_new, u * kEps0InvEvA / kKbEvK * kTwoPi / bx.volume, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ewald_energy.f90:272
            __hip_atomic_store(&g.res[c].u_old, u0 * kEps0InvEvA / kKbEvK * kTwoPi / bx.volume, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        mark(my_role, 2);
    } else {
        // ---------------- pair role: one wave per (entry, split)
        for (int i = tid; i < (bx.coul_last_row + 1) * kCoulRowVec; i += kChainBlock)
            reinterpret_cast<double2 *>(s_dyn)[i] = reinterpret_cast<const double2 *>(coul_tab_g)[i];
        const int nt = tp.n_types;
        for (int i = tid; i < nt * nt; i += kChainBlock) s_pair[i] = pair_tab[i];
        if (FLAT && tid < kMaxGrp) s_grp[tid] = make_int4(tp.grp_start[tid], tp.grp_cnt[tid], tp.grp_ty[tid], 0);
        __syncthreads();
        mark(my_role, 1);
        const int w = ((int)blockIdx.x - n) * kPairWaves + wave;
        if (w < g.n_ent * g.nsplit) {
            const int ent = w / g.nsplit, split = w - ent * g.nsplit;
            const int c = g.ent_c[ent];
            const int t = g.t[c], kind = g.kind[c];
            // old state: the resident molecule; new state: the candidate row; an insertion excludes nothing
            const PairItem it{g.replica, t, kind == 1 ? -1 : g.m[c], g.ent_new[ent] ? c : -1, 0};
            const int n1 = tp.n1[t];
#define MGPU_CHAIN_PAIR(NS)                                                                                              \
            do {                                                                                                         \
                if constexpr (FLAT)                                                                                      \
                    pair_flat_item<NS, false, FASTW, true>(tp, bx, pos, nmol, res_q, res_atype, s_dyn, s_pair, s_grp,         \
                                                     s_plane + wave * kFlatMaxPlanes, it, cand_sites, kMaxFusedSitesWide, split, g.nsplit, lane, 0, g.partials, w); \
                else                                                                                                     \
                    pair_sweep_item<NS, false, TRI, false, FASTW, true>(tp, bx, pos, nmol, res_q, res_atype, pair_tab, s_dyn, s_pair, nullptr, \
                                                                    nullptr, it, cand_sites, kMaxFusedSitesWide, split, g.nsplit, lane, g.partials, w);   \
            } while (0)
            switch (n1) {
                case 1: MGPU_CHAIN_PAIR(1); break;
                case 2: MGPU_CHAIN_PAIR(2); break;
                case 3: MGPU_CHAIN_PAIR(3); break;
                case 4: MGPU_CHAIN_PAIR(4); break;
                default: MGPU_CHAIN_PAIR(5); break;
            }
#undef MGPU_CHAIN_PAIR
        }
        mark(my_role, 2);
    }

    // ---------------- ticket: the last workgroup to arrive resolves the window.  Hand-off without cache-wide fences
    // (/opt/skills/guides/MI355X_MICROARCH.md, "Valid forms": everything handed over is stored `sc1` (agent-scope,
    // write-through), every storing wave waits for its stores, ONE lane per workgroup adds to the counter behind a
    // workgroup barrier, and the workgroup whose add came last reads everything with `sc1` loads, its other waves behind a
    // barrier that the adding wave joins.  An agent release + acquire pair here cost 3-4 us of an 18 us window.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    mark(my_role, 3);
    if (tid == 0) s_flag = (atomicAdd(g.ticket, 1) == (int)gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (!s_flag) return;
    const int rs = g.stamps ? 2 : -1;
    mark(rs, 0);
    mark(rs, 1);
    // split partials of every pair entry into LDS in one round trip, then one thread per (entry, component) adds them in
    // split order -- the order trial_wait uses on the host
    {
        double *st = reinterpret_cast<double *>(s_dyn);
        const int np = g.n_ent * g.nsplit;
        for (int i = tid; i < 2 * np; i += kChainBlock) st[i] = load_sc1(reinterpret_cast<const double *>(g.partials) + i);
        if (tid < 3 * n) reinterpret_cast<double *>(s_res)[tid] = load_sc1(reinterpret_cast<const double *>(g.res) + tid);
        __syncthreads();
        if (tid < 2 * g.n_ent) {
            const int ent = tid >> 1, comp = tid & 1;
            double a = 0.0;
            for (int s2 = 0; s2 < g.nsplit; ++s2) a += st[2 * (ent * g.nsplit + s2) + comp];
            s_ent[tid] = comp ? a * kEps0InvEvA / kKbEvK : a;                  // energy_utils.f90:440
        }
        __syncthreads();
    }
    mark(rs, 2);
    // every step's totals and verdict by its own thread (the rule needs one exp per step); thread 0 then walks the verdicts
    // in order: the window ends at the first accepted or undecided step
    if (tid < n) {
        const int c = tid;
        const ChainResult &r = s_res[c];
        const int kind = g.kind[c], link = g.link[c];
        double o[5] = {0.0, 0.0, r.u_old, 0.0, 0.0}, w[5] = {0.0, 0.0, r.u_new, 0.0, 0.0};
        if (g.ent_old_of[c] >= 0) { o[0] = s_ent[2 * g.ent_old_of[c]]; o[1] = s_ent[2 * g.ent_old_of[c] + 1]; }
        if (g.ent_new_of[c] >= 0) { w[0] = s_ent[2 * g.ent_new_of[c]]; w[1] = s_ent[2 * g.ent_new_of[c] + 1]; }
        if (link != -2) {
            if (kind == 1) { w[3] = g.self_of_type[g.t[c]]; w[4] = r.intra; }
            if (kind == 2) { o[3] = g.self_of_type[g.t[c]]; o[4] = r.intra; }
        }
        double *ho = g.host_out + 10 * (size_t)c;
        for (int k = 0; k < 5; ++k) { ho[k] = o[k]; ho[5 + k] = w[k]; }
        int verdict = 3;                                           // 0 rejected, 1 accepted, 2 undecided, 3 energy-only row
        if (link != -2) {
            // totals as the move drivers form them (mc_chain.f90 resolve_step)
            double e_old, e_new;
            if (kind == 0) {
                e_old = (o[0] + o[1]) + o[2];
                e_new = (w[0] + w[1]) + w[2];
            } else if (kind == 1) {
                e_old = g.e_recip;
                e_new = (((w[0] + w[1]) + w[2]) + w[3]) + w[4];
            } else {
                e_old = (((o[0] + o[1]) + g.e_recip) + o[3]) + o[4];
                e_new = link >= 0 ? s_res[link].u_new : w[2];
            }
            const double x = g.pref[c] * exp(-(e_new - e_old) / g.temperature);
            const double p = x < 1.0 ? x : 1.0;                        // min(1, x)
            // too close to call with another exp (or not a number): the host decides this step
            if (!(x == x) || (x < 1.0 + g.margin && fabs(g.u[c] - x) <= g.margin * x)) verdict = 2;
            else verdict = g.u[c] <= p ? 1 : 0;
        }
        s_verdict[c] = verdict;
    }
    __syncthreads();
    if (tid == 0) {
        int first = -1, undecided = -1;
        for (int c = 0; c < n && first < 0 && undecided < 0; ++c) {
            if (s_verdict[c] == 1) first = c;
            if (s_verdict[c] == 2) undecided = c;
        }
        int *hi = reinterpret_cast<int *>(g.host_out + 10 * (size_t)kChainMaxCand);
        hi[0] = first;
        hi[1] = undecided;
        mark(rs, 3);
        __threadfence_system();
        __hip_atomic_store(g.host_tag, g.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(g.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_flag = first;
        mark(rs, 4);
    }
    __syncthreads();
    const int first = s_flag;
    if (first < 0) return;
    // ---------------- commit of the accepted step by this workgroup: A <- the buffer the step's k role filled with A + delta
    // (the stand-alone commit's arithmetic, done once, by the sweep), then coordinates / count.  An as-written deletion
    // takes the buffer of its companion row: A + the terms of the molecule RemoveMolecule moves into the slot.
    {
        const bool as_written = g.kind[first] == 2 && g.link[first] >= 0;
        const int src = as_written ? g.link[first] : first;
        RecipItem it{g.replica, g.t[first], g.m[first], as_written ? 5 : g.kind[first], src, 0, 0};
        recip_commit_target(tp, nmol, it);
        const double *from = reinterpret_cast<const double *>(g.alt + (size_t)src * bx.n_slots);
        double *to = reinterpret_cast<double *>(A);
        for (int i = tid; i < 2 * bx.n_slots; i += kChainBlock) to[i] = load_sc1(from + i);
        mark(rs, 5);
        if (tid < kBlock) recip_commit_tail(tp, pos, nmol, it, &g.sites[src][0][0], tid);
        mark(rs, 6);
    }
}

// ------------------------------------------------------------------------------------------
// Farm windows: ONE launch per lane step of a FARM of chains (mc_farm.f90, few chains per GPU).
//
// A farm advances its chains in lock step; with few chains a step through the batched path is a latency chain of five
// launches, two copies and two host round trips (45-110 us for ~10 us of arithmetic).  Here the host hands over one RECORD
// per chain -- the move it selected and the uniform numbers of its construction and of its acceptance test, nothing else
// -- and one launch does the step for every chain of the lane:
//   * workgroups [0, P)      ("pair role"): one WAVE per (chain, state, split) work unit of the pair sweep
//     (pair_sweep_item / pair_flat_item, the engine's nsplit: the batched path's partials, bit for bit);
//   * workgroups [P, P + n)  ("k role"): the reciprocal-space sweep of chain c (recip_rows_* pieces), which also stores
//     A + delta into the replica's OTHER A(k) buffer, and the intra-molecular term of an insertion / deletion;
//   * every role rebuilds the candidate it needs from the resident molecule frames and the record's numbers
//     (trial_frame / trial_offset: trial_build_kernel's functions);
//   * ONE TICKET COUNTER PER CHAIN: a workgroup publishes its results with `sc1` stores, waits for them, and one lane
//     adds the number of the chain's work units it carried; the workgroup whose add completes the chain's count resolves
//     the chain with one wave (MI355X_MICROARCH.md "valid forms": no cache-wide fences, no assumption about dispatch
//     order or placement, nobody waits for anybody): split partials summed in split order, the totals formed and the
//     rule applied exactly as mc_farm.f90's resolve_and_commit does, energies + verdict into pinned host memory behind
//     a per-chain tag the host polls -- and an accepted step committed at once: coordinates, frames, count, and the
//     replica's current A(k) buffer switched to the one the k role has just filled.
// The rule on the device uses OCML's exp, the host glibc's: a step whose draw lies within `margin` (relative) of its
// probability -- or whose probability is not a number -- is left UNDECIDED (verdict 2): nothing is committed, the
// replica is marked `stalled`, and every later window already queued for it does nothing (verdict 4) until the host,
// which decides with its own exp, sends the step again with `forced` set.  The host checks every other verdict against
// its own rule, so every decision taken is the host's.
// Records `by_count` (insertion / deletion farms): which molecule a step picks and its prefactor depend on the molecule count
// N, i.e. on the outcome of the chain's previous step -- the one thing that would keep such a farm from queueing a window
// before it has seen the last.  The driver therefore hands over what does NOT depend on N (the residue type and the kind of
// move it drew, the draw of PickRandomMoleculeIndex, phi V) and every workgroup completes its records from the replica's
// count as it is when the launch runs: slot = int(u N) of N (a move or deletion of an empty type, an insertion into a full
// one: nothing to do, as in the reference's drivers), prefactor phi V / (N + 1) or N / (phi V)
// (src/monte_carlo.f90:50-75, src/monte_carlo_utils.f90:184-226).  The driver replays the same integer arithmetic with
// its own counts when it collects the window.
// Orthorhombic boxes, row-form k sweep, molecules of <= kMaxFusedSitesWide sites, frames resident.
// ------------------------------------------------------------------------------------------
struct FarmRec {
    int replica, t, m, move;      // move 0: the chain does nothing this step; 1 translation, 2 rotation, 3 creation, 4 deletion
    int forced;                   // 0: apply the rule; 1 / 2: the host has decided this step -- accept / reject
    int by_count;                 // 1: the molecule slot and the prefactor follow from the replica's molecule COUNT on the device
    double u[5];                  // the construction's uniform numbers (trial_build_kernel)
    double acc_u, pref;           // the test's uniform number and prefactor (1; phi V / (N + 1); N / (phi V)); by_count: phi V
    double sel_u;                 // by_count: the uniform number of PickRandomMoleculeIndex
};
constexpr int kFarmRecWords = 11;
static_assert(sizeof(FarmRec) == 8 * kFarmRecWords, "FarmRec is read as 8-byte words");
constexpr int kFarmInline = 32;                  // records that travel in the kernel arguments (more: read from pinned host memory)
constexpr int kFarmOut = 11;                     // doubles per chain in the host block: old[5] | new[5] | verdict
constexpr int kFarmVerdictRejected = 0, kFarmVerdictAccepted = 1, kFarmVerdictUndecided = 2, kFarmVerdictStalled = 4, kFarmVerdictIdle = 5;
struct FarmArgs {
    const FarmRec *recs;                         // [n] pinned host memory; unused when n <= kFarmInline
    double2 *partials;                           // [n][2][nsplit] device scratch of the lane: entry 0 = old state, 1 = new state
    ChainResult *res;                            // [n] device scratch of the lane
    int *tickets;                                // [n] zero between launches
    int *stalled;                                // [R] per replica: a step waits for the host's decision
    int *acur;                                   // [R] per replica: 1 = its current A(k) lives in A_alt
    double2 *A_alt;                              // [R][n_slots]
    double *host_out;                            // pinned host [n][kFarmOut]
    unsigned long long *host_tag;                // pinned host [n]: the window's sequence number, written last
    unsigned long long seq;
    int n, nsplit;
    double t_step, r_step, temperature, margin;
    double self_of_type[kMaxRes];
    FarmRec inline_recs[kFarmInline];
};
static_assert(sizeof(BoxDev) + sizeof(FarmArgs) + 160 <= 4096, "a farm window must fit the kernel-argument segment");

// Diagnostic builds only (-DMGPU_FARM_STAMPS, tools/farm_stages.py; the shipped library has none of this): wall-clock stamps of
// chain 0's k role (row 0), the launch's first pair workgroup (row 1) and chain 0's resolver (row 2).
#ifdef MGPU_FARM_STAMPS
static __device__ long long g_farm_stamps[3][8];
#define MGPU_FSTAMP(cond, role, i) do { if (cond) g_farm_stamps[role][i] = wall_clock64(); } while (0)
#else
#define MGPU_FSTAMP(cond, role, i) do { } while (0)
#endif

// One chain resolved by ONE WAVE (all 64 lanes arrive): `scratch` = 4 nsplit + 4 doubles of LDS of its own.
__device__ __forceinline__ void farm_resolve(const Topo &tp, const BoxDev &bx, double *__restrict__ pos, int *__restrict__ nmol,
                                             const FarmArgs &g, const FarmRec &rec, int c, int lane, double *scratch) {
    double *ho = g.host_out + (size_t)kFarmOut * c;
    int verdict;
    MGPU_FSTAMP(c == 0 && lane == 0, 2, 0);
    // the roles sweep whatever the replica's stall flag says (they only read, and the k role's A + delta goes to the buffer
    // that is NOT current): the flag is looked at here, once, beside the partials -- not on every role's critical path
    const int waits = rec.forced == 0 && g.stalled[rec.replica] != 0;
    const int skip = rec.move == 0 || waits;
    double o[5] = {0.0, 0.0, 0.0, 0.0, 0.0}, w[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    const int kind = rec.move <= 2 ? 0 : (rec.move == 3 ? 1 : 2);
    if (skip) {
        // (a by-count record behind an undecided step waits even when THIS count makes it a no-op: the step it waits for
        //  may change the count, and the chain's steps are followed in order)
        verdict = waits && (rec.by_count || rec.move != 0) ? kFarmVerdictStalled : kFarmVerdictIdle;
    } else {
        // every split partial of the chain's two entries in one round trip, then one lane per (entry, component) adds them
        // in split order -- the order trial_wait uses on the host
        const int ns = g.nsplit, np = 4 * ns;
        const double *pd = reinterpret_cast<const double *>(g.partials + (size_t)c * 2 * ns);
        for (int i = lane; i < np; i += 64) {
            const int ent = i / (2 * ns);
            const bool have = ent == 0 ? kind != 1 : kind != 2;
            scratch[i] = have ? load_sc1(pd + i) : 0.0;
        }
        if (lane < 3) scratch[np + lane] = load_sc1(reinterpret_cast<const double *>(g.res + c) + lane);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        MGPU_FSTAMP(c == 0 && lane == 0, 2, 1);
        double sum = 0.0;
        if (lane < 4) {
            const int ent = lane >> 1, comp = lane & 1;
            for (int s2 = 0; s2 < ns; ++s2) sum += scratch[2 * (ent * ns + s2) + comp];
            if (comp) sum = sum * kEps0InvEvA / kKbEvK;                        // energy_utils.f90:440
        }
        const double lj_o = __shfl(sum, 0, 64), cc_o = __shfl(sum, 1, 64), lj_n = __shfl(sum, 2, 64), cc_n = __shfl(sum, 3, 64);
        const double u_old = scratch[np], u_new = scratch[np + 1], intra = scratch[np + 2];
        // old / new components as trial_wait_impl fills them (ncomp = 5)
        o[2] = u_old; w[2] = u_new;
        if (kind != 1) { o[0] = lj_o; o[1] = cc_o; }
        if (kind != 2) { w[0] = lj_n; w[1] = cc_n; }
        if (kind == 1) { w[3] = g.self_of_type[rec.t]; w[4] = intra; }
        if (kind == 2) { o[3] = g.self_of_type[rec.t]; o[4] = intra; }
        if (rec.forced) {
            verdict = rec.forced == 1 ? kFarmVerdictAccepted : kFarmVerdictRejected;
        } else {
            // old%total, new%total and the rule as mc_farm.f90 resolve_and_commit forms them (monte_carlo_utils.f90:184-226)
            double e_old = 0.0, e_new = 0.0;
            for (int k = 0; k < 5; ++k) { e_old = e_old + o[k]; e_new = e_new + w[k]; }
            const double x = rec.pref * exp(-(e_new - e_old) / g.temperature);
            const double pr = x < 1.0 ? x : 1.0;                               // min(1, x)
            if (!(x == x) || (x < 1.0 + g.margin && fabs(rec.acc_u - x) <= g.margin * x)) verdict = kFarmVerdictUndecided;
            else verdict = rec.acc_u <= pr ? kFarmVerdictAccepted : kFarmVerdictRejected;
        }
    }
    // ---- energies + verdict into pinned host memory, the tag behind them.  No system-scope FENCE: a fence writes back the
    // XCD's whole L2 -- every chain's freshly stored A + delta -- once per chain (measured: 256 chains per launch took as
    // long as 512, ~94 us, and two lanes' launches ran at half speed).  The block is fine-grained host memory: the stores
    // are system-scope write-through stores, the wave waits for them to be acknowledged, then stores the tag.
    MGPU_FSTAMP(c == 0 && lane == 0, 2, 2);
    {
        double val = 0.0;
        if (lane < 5) val = o[lane];
        else if (lane < 10) val = w[lane - 5];
        else if (lane == 10) val = (double)verdict;
        if (lane < kFarmOut) __hip_atomic_store(ho + lane, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // (the tag follows BEHIND the commit: the eleven words cross PCIe while the accepted step is committed -- stage stamps, 8
    //  chains: 0.9 us of acknowledgement and 1.4 us of commit were one after the other; the launch's end is what the next
    //  window waits for, and the driver, with windows in flight, is not waiting for this tag)
    auto publish_tag = [&] {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(g.host_tag + c, g.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        MGPU_FSTAMP(c == 0 && lane == 0, 2, 3);
    };
    // ---- the chain's device state: ticket, stall flag, and the accepted step itself
    if (lane == 0) {
        __hip_atomic_store(g.tickets + c, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (rec.move != 0 && (verdict == kFarmVerdictUndecided || rec.forced)) g.stalled[rec.replica] = verdict == kFarmVerdictUndecided ? 1 : 0;
    }
    if (verdict != kFarmVerdictAccepted) { publish_tag(); return; }
    const int n1 = tp.n1[rec.t];
    double *px = pos + (size_t)rec.replica * 3 * tp.n_cap_atoms;
    double *py = px + tp.n_cap_atoms, *pz = py + tp.n_cap_atoms;
    const size_t rep3 = (size_t)rec.replica * 3;
    double *fcom = tp.com + rep3 * tp.n_mol_slots + tp.mol_off[rec.t];
    double *foff = tp.off + rep3 * tp.n_cap_atoms;
    const int nm = nmol[rec.replica * tp.n_res + rec.t];
    if (kind != 2) {
        const int m = kind == 1 ? nm : rec.m;                 // appended at the first free slot (monte_carlo.f90:63, create_molecule.f90:64)
        const TrialFrame f = trial_frame(tp, bx, rec.replica, rec.t, rec.m, rec.move, rec.u, g.t_step, g.r_step);
        if (lane < n1) {
            double off[3];
            trial_offset(tp, f, rec.replica, rec.t, lane, off);
            const int j = atom_slot(tp, rec.t, m, lane);
            px[j] = f.com[0] + off[0]; py[j] = f.com[1] + off[1]; pz[j] = f.com[2] + off[2];
            for (int d = 0; d < 3; ++d) foff[(size_t)d * tp.n_cap_atoms + j] = off[d];
        }
        if (lane < 3) fcom[(size_t)lane * tp.n_mol_slots + m] = f.com[lane];
        if (lane == 0 && kind == 1) nmol[rec.replica * tp.n_res + rec.t] = nm + 1;
    } else {
        const int last = nm - 1;                                 // swap-with-last, delete_molecule.f90:107-114
        if (last != rec.m) {
            if (lane < n1) {
                const int j = atom_slot(tp, rec.t, rec.m, lane), jl = atom_slot(tp, rec.t, last, lane);
                px[j] = px[jl]; py[j] = py[jl]; pz[j] = pz[jl];
                for (int d = 0; d < 3; ++d) foff[(size_t)d * tp.n_cap_atoms + j] = foff[(size_t)d * tp.n_cap_atoms + jl];
            }
            if (lane < 3) fcom[(size_t)lane * tp.n_mol_slots + rec.m] = fcom[(size_t)lane * tp.n_mol_slots + last];
        }
        if (lane == 0) nmol[rec.replica * tp.n_res + rec.t] = last;
    }
    // A(k): the buffer the k role filled with A + delta becomes the replica's current one
    if (lane == 0) g.acur[rec.replica] ^= 1;
    MGPU_FSTAMP(c == 0 && lane == 0, 2, 4);
    publish_tag();
}

// (launch bounds: four waves per SIMD = two of these 8-wave workgroups per CU, i.e. at most 128 VGPRs.  Left to itself
//  the compiler took 130-132 -- ONE workgroup per CU -- and every farm of more than ~28 chains paid a second round of
//  workgroups: 64 chains 22 -> 32 us per step, 512 chains 57 -> 82 us.)
template <bool FLAT, bool FASTW>
__global__ __launch_bounds__(kChainBlock, 4) void farm_window_kernel(
    const Topo *__restrict__ tpp, BoxDev bx, double *__restrict__ pos, int *__restrict__ nmol, const double *__restrict__ res_q,
    const int *__restrict__ res_atype, const double2 *__restrict__ pair_tab, const char *__restrict__ coul_tab_g,
    const int *__restrict__ trj, const double2 *__restrict__ tw, int n_tasks, const RecipRow *__restrict__ rows, int n_rows,
    double2 *__restrict__ A_base, const FarmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char s_dyn[];      // Coulomb table | phase tables; then the resolving waves' scratch
    __shared__ double2 s_pair[kMaxTypes * kMaxTypes];
    __shared__ int4 s_grp[kMaxGrp];
    __shared__ int4 s_plane[FLAT ? kPairWaves * kFlatMaxPlanes : 1];
    __shared__ double s_red[2 * kWavesPerBlock];
    __shared__ FarmRec s_rec[kPairWaves];                              // the records of the chains this workgroup works for
    __shared__ int s_skip[kPairWaves], s_resolve[kPairWaves], s_acur;
    __shared__ double s_cand[kPairWaves][kMaxFusedSitesWide * 3];      // candidate rows: one per wave (pair role) / row 0 (k role)

    const Topo &tp = *tpp;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = g.n, ns = g.nsplit, wpc = 2 * ns, expected = wpc + 1;
    // Workgroup -> (chain, role): pair workgroups first, 8 consecutive work units each (a workgroup may serve several
    // chains), then one k workgroup per chain.  (Measured and dropped, round 5: a chain's workgroups congruent modulo 8 --
    // one XCD, one L2 per chain under the observed round-robin placement: no gain at 8 chains, 64 chains 2.05 -> 1.86 M,
    // 512: 5.5 -> 3.9 M on one lane: a chain's work units read DISJOINT atoms, so one L2 saves nothing and its channels
    // become the chain's bottleneck.)
    const int n_pair_wg = (n * wpc + kPairWaves - 1) / kPairWaves;
    const bool k_role = (int)blockIdx.x >= n_pair_wg;
    int c_lo, n_c, w0 = 0, w1 = 0;             // chains [c_lo, c_lo + n_c) and global pair work units [w0, w1) of this workgroup
    if (k_role) { c_lo = (int)blockIdx.x - n_pair_wg; n_c = 1; }
    else {
        w0 = (int)blockIdx.x * kPairWaves;
        w1 = min(w0 + kPairWaves, n * wpc);
        c_lo = w0 / wpc; n_c = (w1 - 1) / wpc - c_lo + 1;
    }
    MGPU_FSTAMP(tid == 0 && blockIdx.x == 0, 1, 0);
    MGPU_FSTAMP(tid == 0 && k_role && c_lo == 0, 0, 0);
    // ---- records: ten 8-byte words per chain from the kernel arguments (few chains) or from pinned host memory
    {
        const double *src = reinterpret_cast<const double *>((n <= kFarmInline ? g.inline_recs : g.recs) + c_lo);
        if (tid < kFarmRecWords * n_c) reinterpret_cast<double *>(s_rec)[tid] = src[tid];
    }
    if (!k_role) {
        // (the table staging runs under the records' load)
        for (int i = tid; i < (bx.coul_last_row + 1) * kCoulRowVec; i += kChainBlock)
            reinterpret_cast<double2 *>(s_dyn)[i] = reinterpret_cast<const double2 *>(coul_tab_g)[i];
        const int nt = tp.n_types;
        for (int i = tid; i < nt * nt; i += kChainBlock) s_pair[i] = pair_tab[i];
        if (FLAT && tid < kMaxGrp) s_grp[tid] = make_int4(tp.grp_start[tid], tp.grp_cnt[tid], tp.grp_ty[tid], 0);
    }
    __syncthreads();
    if (tid < n_c && s_rec[tid].by_count && s_rec[tid].move != 0) {
        // complete the record from the replica's molecule count (see above); every workgroup of the chain does the same
        FarmRec &r = s_rec[tid];
        const int nm = nmol[r.replica * tp.n_res + r.t];
        if (r.move == 3) {
            if (nm >= tp.cap[r.t]) r.move = 0;                          // full: nothing to do (monte_carlo.f90:63)
            else { r.m = 0; r.pref = r.pref / (double)(nm + 1); }       // phi V / (N + 1), N + 1 = the count after it
        } else if (nm <= 0) {
            r.move = 0;                                                 // PickRandomMoleculeIndex of an empty type: the drivers return
        } else {
            r.m = min((int)(r.sel_u * nm), nm - 1);
            if (r.move == 4) r.pref = ((double)(nm - 1) + 1.0) / r.pref;   // (N' + 1) / (phi V), N' = the count after it
        }
    }
    if (tid < n_c) s_skip[tid] = s_rec[tid].move == 0 ? 1 : 0;          // (a stalled replica is the resolver's business)
    if (k_role && tid == 0) s_acur = s_rec[0].move != 0 ? g.acur[s_rec[0].replica] : 0;
    __syncthreads();
    MGPU_FSTAMP(tid == 0 && blockIdx.x == 0, 1, 1);
    MGPU_FSTAMP(tid == 0 && k_role && c_lo == 0, 0, 1);

    if (k_role) {
        // ---------------- k role: chain c_lo
        const int c = c_lo;
        const FarmRec &rec = s_rec[0];
        if (!s_skip[0]) {
            const int kind = rec.move <= 2 ? 0 : (rec.move == 3 ? 1 : 2);
            const int n1 = tp.n1[rec.t];
            if (kind != 2 && tid < n1) {
                const TrialFrame f = trial_frame(tp, bx, rec.replica, rec.t, rec.m, rec.move, rec.u, g.t_step, g.r_step);
                double off[3];
                trial_offset(tp, f, rec.replica, rec.t, tid, off);
                for (int d = 0; d < 3; ++d) s_cand[0][tid * 3 + d] = f.com[d] + off[d];
            }
            __syncthreads();                                           // (uniform: s_skip is the workgroup's)
            MGPU_FSTAMP(tid == 0 && c == 0, 0, 2);
            double2 *A = (s_acur ? g.A_alt : A_base) + (size_t)rec.replica * bx.n_slots;
            double2 *A_other = (s_acur ? A_base : g.A_alt) + (size_t)rec.replica * bx.n_slots;
            RecipItem it{rec.replica, rec.t, kind == 1 ? -1 : rec.m, kind, 0, 0, 0};
            const RecipLds v = recip_lds_view(tp, bx, it, n_rows, reinterpret_cast<double2 *>(s_dyn));
            const bool active = tid < kBlock;
            RecipInFlight<kRecipTaskChunk> inflight;
            recip_rows_tables(tp, bx, pos, res_q, rows, n_rows, it, &s_cand[0][0], v, tid, active,
                              [&] { recip_rows_prefetch<false>(inflight, trj, tw, n_tasks, A, tid); });
            double acc = 0.0, acc0 = 0.0;
            MGPU_FSTAMP(tid == 0 && c == 0, 0, 3);
            if (active) recip_rows_pass<false, true, kRecipTaskChunk, 1>(v, trj, tw, n_tasks, A, tid, inflight, acc, acc0, A_other);
            MGPU_FSTAMP(tid == 0 && c == 0, 0, 4);
            if (tid == kBlock && kind != 0) {
                // ComputeIntraResidueRealCoulombEnergySingleMol of the inserted (candidate row) / deleted (resident) molecule
                const PairItem pit{rec.replica, rec.t, rec.m, kind == 1 ? 0 : -1, 0};
                __hip_atomic_store(&g.res[c].intra, intra_energy(tp, bx, pos, res_q, pit, &s_cand[0][0], kMaxFusedSitesWide), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
            if (active) {
                acc = wave_sum(acc);
                acc0 = wave_sum(acc0);
                if (lane == 0) { s_red[2 * wave] = acc; s_red[2 * wave + 1] = acc0; }
            }
            __syncthreads();
            if (tid == 0) {
                double u = 0.0, u0 = 0.0;
                for (int wv = 0; wv < kWavesPerBlock; ++wv) { u += s_red[2 * wv]; u0 += s_red[2 * wv + 1]; }
                __hip_atomic_store(&g.res[c].u_new, u * kEps0InvEvA / kKbEvK * kTwoPi / bx.volume, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ewald_energy.f90:272
                __hip_atomic_store(&g.res[c].u_old, u0 * kEps0InvEvA / kKbEvK * kTwoPi / bx.volume, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                MGPU_FSTAMP(c == 0, 0, 5);
            }
        }
    } else {
        // ---------------- pair role: one wave per (chain, entry, split); entry 0 = the resident molecule, 1 = the candidate
        const int wg = w0 + wave;
        const int c = wg / wpc, j = wg - c * wpc;
        if (wg < w1 && !s_skip[c - c_lo]) {
            const FarmRec &rec = s_rec[c - c_lo];
            const int kind = rec.move <= 2 ? 0 : (rec.move == 3 ? 1 : 2);
            const int ent = j / ns, split = j - ent * ns;
            const int n1 = tp.n1[rec.t];
            if (ent == 0 ? kind != 1 : kind != 2) {
                double *cand = &s_cand[wave][0];
                if (ent == 1) {
                    const TrialFrame f = trial_frame(tp, bx, rec.replica, rec.t, rec.m, rec.move, rec.u, g.t_step, g.r_step);
                    if (lane < n1) {
                        double off[3];
                        trial_offset(tp, f, rec.replica, rec.t, lane, off);
                        for (int d = 0; d < 3; ++d) cand[lane * 3 + d] = f.com[d] + off[d];
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
                // old state: the resident molecule; new state: the candidate row; an insertion excludes nothing
                const PairItem it{rec.replica, rec.t, kind == 1 ? -1 : rec.m, ent == 1 ? 0 : -1, 0};
#define MGPU_FARM_PAIR(NS)                                                                                               \
                do {                                                                                                     \
                    if constexpr (FLAT)                                                                                  \
                        pair_flat_item<NS, false, FASTW, true>(tp, bx, pos, nmol, res_q, res_atype, s_dyn, s_pair, s_grp,     \
                                                         s_plane + wave * kFlatMaxPlanes, it, cand, kMaxFusedSitesWide, split, ns, lane, 0, g.partials, wg); \
                    else                                                                                                 \
                        pair_sweep_item<NS, false, false, false, FASTW, true>(tp, bx, pos, nmol, res_q, res_atype, pair_tab, s_dyn, s_pair, nullptr, \
                                                                        nullptr, it, cand, kMaxFusedSitesWide, split, ns, lane, g.partials, wg);   \
                } while (0)
                switch (n1) {
                    case 1: MGPU_FARM_PAIR(1); break;
                    case 2: MGPU_FARM_PAIR(2); break;
                    case 3: MGPU_FARM_PAIR(3); break;
                    case 4: MGPU_FARM_PAIR(4); break;
                    default: MGPU_FARM_PAIR(5); break;
                }
#undef MGPU_FARM_PAIR
            }
        }
        MGPU_FSTAMP(tid == 0 && blockIdx.x == 0, 1, 2);
    }

    // ---------------- tickets: one counter per chain; the workgroup whose add completes a chain's count resolves it
    // (every storing wave waits for its `sc1` stores, ONE lane per workgroup and chain adds behind the barrier, and the
    // resolving wave loads with `sc1` behind a second barrier that the adding wave joins)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    MGPU_FSTAMP(tid == 0 && blockIdx.x == 0, 1, 3);
    MGPU_FSTAMP(tid == 0 && k_role && c_lo == 0, 0, 6);
    if (tid < n_c) {
        int count = 1;
        if (!k_role) {
            const int a = max(w0, (c_lo + tid) * wpc), b = min(w1, (c_lo + tid + 1) * wpc);
            count = b - a;
        }
        s_resolve[tid] = (atomicAdd(g.tickets + c_lo + tid, count) + count == expected) ? 1 : 0;
    }
    __syncthreads();
    MGPU_FSTAMP(tid == 0 && blockIdx.x == 0, 1, 4);
    MGPU_FSTAMP(tid == 0 && k_role && c_lo == 0, 0, 7);
    if (wave < n_c && s_resolve[wave]) {
        double *scratch = reinterpret_cast<double *>(s_dyn) + (size_t)wave * (4 * ns + 4);
        farm_resolve(tp, bx, pos, nmol, g, s_rec[wave], c_lo + wave, lane, scratch);
    }
}

// A(k) of every replica back into the engine's primary buffer (farm windows leave a replica's current A(k) in either):
// one workgroup per replica.
static __global__ __launch_bounds__(kBlock) void farm_normalize_kernel(int *__restrict__ acur, double2 *__restrict__ A_base,
                                                                const double2 *__restrict__ A_alt, int n_slots) {
    const int r = blockIdx.x;
    if (!acur[r]) return;                                              // uniform per workgroup
    double2 *dst = A_base + (size_t)r * n_slots;
    const double2 *src = A_alt + (size_t)r * n_slots;
    for (int i = threadIdx.x; i < n_slots; i += kBlock) dst[i] = src[i];
    __syncthreads();
    if (threadIdx.x == 0) acur[r] = 0;
}

// empty dispatch used by mgpu_profile_enable to switch a stream's queue into profiling mode ahead of time
static __global__ void prime_kernel(const int *p) { (void)p; }

}  // namespace mgpu

#endif
