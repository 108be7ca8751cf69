// Reciprocal-space update (per k, row form, wide row form on the matrix units), trial geometry, S(k), intra-molecular sum.
// (one of the four parts of mgpu_kernels.h: include that header, not this file)
#ifndef MGPU_KERNELS_RECIP_H
#define MGPU_KERNELS_RECIP_H

#include "mgpu_kernels_common.h"

namespace mgpu {

// ------------------------------------------------------------------------------------------
// Reciprocal-space update.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
    return make_double2(fma(a.x, b.x, -a.y * b.y), fma(a.x, b.y, a.y * b.x));
}

// Fractional phase theta = 2 pi (reciprocal^T r) along one axis: ComputeAtomPhase (ewald_phase.f90:41-64), the same sum
// in the same association order.
__device__ __forceinline__ double atom_phase(const BoxDev &bx, int axis, double x, double y, double z) {
    double acc = 0.0;
    acc = acc + bx.rcp[0 * 3 + axis] * x;
    acc = acc + bx.rcp[1 * 3 + axis] * y;
    acc = acc + bx.rcp[2 * 3 + axis] * z;
    return kTwoPi * acc;
}

// sin and cos of x for |x| < 2^30 (here |k theta| <= 255 * 2 pi * a fractional coordinate of order one): the argument
// is reduced by n = rint(x * 2/pi) against pi/2 held in three doubles (Cody-Waite with fused multiply-adds: the
// products are exact inside the fma, so the reduction holds next to the multiples of pi/2 as well), the two kernels are
// the classic minimax polynomials on [-pi/4, pi/4] (degree 13 / 14; the cosine's 1 - z/2 carries its rounding error
// along), the quadrant picks and signs them.  Within 1.5 ulp of the exact value over the whole range (measured on
// 2 x 10^7 arguments, tests/test_gpu_parity.py::test_phase_factors_are_within_two_ulp) -- the accuracy class of the
// library's sincos -- in ~48 vector instructions against the library routine's ~130 with its large-argument branch,
// and 20 fewer registers: phase 1 of the k sweep is one such evaluation per thread, and the registers buy the sweep
// its sixth workgroup per CU.
__device__ __forceinline__ void sincos_bounded(double x, double &sn, double &cs) {
    const double n = rint(x * 6.36619772367581382433e-01);
    double r = fma(-n, 1.5707963267948966, x);                    // pi/2 = 1.5707963267948966 + 6.123233995736766e-17 - 1.4973849048591698e-33
    r = fma(-n, 6.123233995736766e-17, r);
    r = fma(-n, -1.4973849048591698e-33, r);
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    ps = fma(z, ps, -1.66666666666666324348e-01);
    const double s0 = fma(r * z, ps, r);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    const double hz = 0.5 * z, w = 1.0 - hz;
    const double c0 = w + fma(z * z, pc, (1.0 - w) - hz);         // (1 - w) - hz: what rounding w lost, exactly
    const int q = (int)n;
    const bool swap = q & 1;
    const double sv = swap ? c0 : s0, cv = swap ? s0 : c0;
    sn = (q & 2) ? -sv : sv;
    cs = ((q + 1) & 2) ? -cv : cv;
}

// exp(i k theta): dcos / dsin of the rounded product k * theta, as ComputePhaseFactors1D (ewald_phase.f90:100-109).
__device__ __forceinline__ double2 phase_entry(double theta, int k) {
    double s, c;
    sincos_bounded((double)k * theta, s, c);
    return make_double2(c, s);
}

// test and diagnostic hook (mgpu_phase_factors): the table entries exactly as the sweeps form them
static __global__ void phase_factors_kernel(int n, const double *__restrict__ theta, const int *__restrict__ k, double2 *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = phase_entry(theta[i], k[i]);
}

// SingleMolFourierTerms + ComputeRecipEnergySingleMol (ewald_phase.f90:383-420,
// ewald_energy.f90:191-274) for one item per workgroup.
// COMMIT = false: u_new[item] = prefactor * sum_k ff W |A + delta|^2, A untouched; with BOTH also
//                 u_old[item] = prefactor * sum_k ff W |A|^2 from the same pass over k (the reference's
//                 ComputeOldEnergy call, where delta = 0, monte_carlo_utils.f90:388).
// COMMIT = true : A <- A + delta, then the replica's coordinates / molecule count are updated.
// Dynamic LDS: two table sets (new, old) of `tile` * ktot complex entries (entry (a, axis, k >= 0)), then `tile` charges:
// the molecule's sites pass through LDS `tile` at a time (the engine picks the tile from its LDS budget: a molecule of a
// few sites is one tile, a 300-site adsorbate or a framework seven), so a molecule of ANY size is updated -- the
// reference's tables are sized by max_atom_in_residue (prepare_utils.f90:233-235), not by a cache.
// Each thread owns k = tid + 256 j and takes kRecipChunk of them per pass over the tiles, their delta(k) held in
// registers across the tiles (sites added in the order a = 0, 1, ..., whatever the tiling: the same bits); A, ff*W and
// the packed indices of a chunk are loaded up front so that the L2 latencies overlap instead of serialising per k.
constexpr int kRecipChunk = 8;
template <bool COMMIT, bool BOTH>
__global__ __launch_bounds__(kBlock) void recip_kernel(
    Topo tp, BoxDev bx, double *__restrict__ pos, int *__restrict__ nmol, const double *__restrict__ res_q,
    const int *__restrict__ kpack, const int *__restrict__ kslot, const double *__restrict__ kw,
    double2 *__restrict__ A_base, const RecipItem *__restrict__ items, const double *__restrict__ cand_sites,
    int site_stride, int tile, double *__restrict__ u_new, double *__restrict__ u_old) {
    extern __shared__ double2 s_tab[];
    __shared__ double s_red[2 * kWavesPerBlock];

    const RecipItem it = items[blockIdx.x];
    const int n1 = tp.n1[it.t];
    const int kofs[3] = {0, bx.kmax[0] + 1, bx.kmax[0] + bx.kmax[1] + 2};
    const int ktot = bx.kmax[0] + bx.kmax[1] + bx.kmax[2] + 3;
    double2 *tab_new = s_tab, *tab_old = s_tab + tile * ktot;
    double *s_q = reinterpret_cast<double *>(s_tab + 2 * tile * ktot);
    double *px = pos + (size_t)it.replica * 3 * tp.n_cap_atoms;
    double *py = px + tp.n_cap_atoms, *pz = py + tp.n_cap_atoms;
    const bool use_new = (it.kind == 0 /*MOVE*/ || it.kind == 1 /*CREATION*/ || it.kind == 4 /*FOURIER_ADD*/);
    const bool use_old = (it.kind == 0 /*MOVE*/ || it.kind == 2 /*DELETION*/);

    // tables of the sites [a0, a0 + na)
    auto build_tile = [&](int a0, int na) {
        for (int e = threadIdx.x; e < 2 * na * ktot; e += kBlock) {
            const int set = e / (na * ktot), r = e - set * na * ktot;
            const int a = r / ktot, kk = r - a * ktot;
            const int axis = (kk >= kofs[2]) ? 2 : (kk >= kofs[1] ? 1 : 0);
            const int k = kk - kofs[axis];
            if ((set == 0 && !use_new) || (set == 1 && !use_old)) continue;
            double x, y, z;
            if (set == 0) {
                const double *c = cand_sites + ((size_t)it.src * site_stride + a0 + a) * 3;
                x = c[0]; y = c[1]; z = c[2];
            } else {
                const int j = atom_slot(tp, it.t, it.m, a0 + a);
                x = px[j]; y = py[j]; z = pz[j];
            }
            (set == 0 ? tab_new : tab_old)[a * ktot + kk] = phase_entry(atom_phase(bx, axis, x, y, z), k);
        }
        for (int a = threadIdx.x; a < na; a += kBlock) s_q[a] = res_q[it.t * tp.max_atom + a0 + a];
    };

    double2 *A = A_base + (size_t)it.replica * bx.n_slots;
    double acc = 0.0, acc0 = 0.0;
    // (the trip count is uniform over the workgroup: the tile barriers sit inside)
    for (int kb = 0; kb < bx.nk; kb += kBlock * kRecipChunk) {
        const int k0 = kb + threadIdx.x;
        double2 Ak[kRecipChunk];
        double w[kRecipChunk], dre[kRecipChunk], dim[kRecipChunk];
        int kp[kRecipChunk], ks[kRecipChunk];
#pragma unroll
        for (int j = 0; j < kRecipChunk; ++j) {
            const int k = k0 + j * kBlock;
            const bool in = k < bx.nk;
            ks[j] = in ? kslot[k] : 0;
            Ak[j] = in ? A[ks[j]] : make_double2(0.0, 0.0);
            w[j] = (in && !COMMIT) ? kw[k] : 0.0;
            kp[j] = in ? kpack[k] : ((128 << 8) | (128 << 16));   // (0, 0, 0): harmless filler
            dre[j] = 0.0; dim[j] = 0.0;
        }
        for (int a0 = 0; a0 < n1; a0 += tile) {
            const int na = min(tile, n1 - a0);
            __syncthreads();                                       // the previous tile's readers are done
            build_tile(a0, na);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < kRecipChunk; ++j) {
                const int kx = kp[j] & 0xff, ky = ((kp[j] >> 8) & 0xff) - 128, kz = ((kp[j] >> 16) & 0xff) - 128;
                const int aky = ky < 0 ? -ky : ky, akz = kz < 0 ? -kz : kz;
                for (int a = 0; a < na; ++a) {
                    const double q = s_q[a];
                    double2 pn = make_double2(0.0, 0.0), po = make_double2(0.0, 0.0);
                    if (use_new) {
                        const double2 *t = tab_new + a * ktot;
                        double2 Y = t[kofs[1] + aky], Z = t[kofs[2] + akz];
                        if (ky < 0) Y.y = -Y.y;
                        if (kz < 0) Z.y = -Z.y;
                        pn = cmul(cmul(t[kx], Y), Z);
                    }
                    if (use_old) {
                        const double2 *t = tab_old + a * ktot;
                        double2 Y = t[kofs[1] + aky], Z = t[kofs[2] + akz];
                        if (ky < 0) Y.y = -Y.y;
                        if (kz < 0) Z.y = -Z.y;
                        po = cmul(cmul(t[kx], Y), Z);
                    }
                    // ewald_energy.f90:241-256
                    dre[j] += q * (pn.x - po.x);
                    dim[j] += q * (pn.y - po.y);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < kRecipChunk; ++j) {
            if (BOTH) acc0 += w[j] * fma(Ak[j].x, Ak[j].x, Ak[j].y * Ak[j].y);
            const double nx = Ak[j].x + dre[j], ny = Ak[j].y + dim[j];
            if (COMMIT) {
                if (k0 + j * kBlock < bx.nk) A[ks[j]] = make_double2(nx, ny);
            } else {
                acc += w[j] * fma(nx, nx, ny * ny);                 // ewald_energy.f90:259-266
            }
        }
    }
    __syncthreads();            // every read of the old coordinates (all tiles, all passes) lies before the commit's writes

    if (!COMMIT) {
        acc = wave_sum(acc);
        if (BOTH) acc0 = wave_sum(acc0);
        if ((threadIdx.x & 63) == 0) { s_red[2 * (threadIdx.x >> 6)] = acc; s_red[2 * (threadIdx.x >> 6) + 1] = acc0; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double u = 0.0, u0 = 0.0;
            for (int wv = 0; wv < kWavesPerBlock; ++wv) { u += s_red[2 * wv]; u0 += s_red[2 * wv + 1]; }
            u_new[blockIdx.x] = u * kEps0InvEvA / kKbEvK * kTwoPi / bx.volume;   // ewald_energy.f90:272
            if (BOTH) u_old[blockIdx.x] = u0 * kEps0InvEvA / kKbEvK * kTwoPi / bx.volume;
        }
    } else {
        // every read of the old coordinates happened before the barrier above; a molecule may have more sites than the
        // workgroup has threads
        if (it.kind == 0 || it.kind == 1) {
            for (int a = threadIdx.x; a < n1; a += kBlock) {
                const double *c = cand_sites + ((size_t)it.src * site_stride + a) * 3;
                const int j = atom_slot(tp, it.t, it.m, a);
                px[j] = c[0]; py[j] = c[1]; pz[j] = c[2];
            }
        } else if (it.kind == 2) {
            // swap-with-last, delete_molecule.f90:107-114: slot m <- slot (new count)
            const int last = it.aux;
            if (last != it.m)
                for (int a = threadIdx.x; a < n1; a += kBlock) {
                    const int j = atom_slot(tp, it.t, it.m, a), jl = atom_slot(tp, it.t, last, a);
                    px[j] = px[jl]; py[j] = py[jl]; pz[j] = pz[jl];
                }
        }
        // molecule frames, where the engine keeps them: a device-built move / insertion writes its com and offsets back,
        // a deletion moves the last molecule's frame with its sites
        if (tp.com) {
            const size_t rep3 = (size_t)it.replica * 3;
            double *fcom = tp.com + rep3 * tp.n_mol_slots + tp.mol_off[it.t];
            double *foff = tp.off + rep3 * tp.n_cap_atoms;
            if ((it.kind == 0 || it.kind == 1) && it.frame > 0) {
                const double *fr = cand_sites + ((size_t)it.src * site_stride + it.frame) * 3;
                if (threadIdx.x < 3) fcom[(size_t)threadIdx.x * tp.n_mol_slots + it.m] = fr[threadIdx.x];
                for (int a = threadIdx.x; a < n1; a += kBlock) {
                    const int j = atom_slot(tp, it.t, it.m, a);
                    for (int d = 0; d < 3; ++d) foff[(size_t)d * tp.n_cap_atoms + j] = fr[(1 + a) * 3 + d];
                }
            } else if (it.kind == 2 && it.aux != it.m) {
                const int last = it.aux;
                if (threadIdx.x < 3) fcom[(size_t)threadIdx.x * tp.n_mol_slots + it.m] = fcom[(size_t)threadIdx.x * tp.n_mol_slots + last];
                for (int a = threadIdx.x; a < n1; a += kBlock) {
                    const int j = atom_slot(tp, it.t, it.m, a), jl = atom_slot(tp, it.t, last, a);
                    for (int d = 0; d < 3; ++d) foff[(size_t)d * tp.n_cap_atoms + j] = foff[(size_t)d * tp.n_cap_atoms + jl];
                }
            }
        }
        if (threadIdx.x == 0 && (it.kind == 1 || it.kind == 2)) nmol[it.replica * tp.n_res + it.t] = it.aux;
    }
}

// ------------------------------------------------------------------------------------------
// Reciprocal-space update, row form (the fast path for molecules of a few sites).
//
// The k list is generated with kz innermost (ewald_kvectors.f90:150-246), so all k of one (kx, ky)
// "row" are contiguous and come in +-kz pairs.  Work is organised around that:
//   phase 1   1-D tables e^{i k theta} per site and axis, sincos(k * theta) as ComputePhaseFactors1D;
//   phase 2   per row and site-state (new sites, old sites):  XY = +-q * X[kx] * Y[ky]  into LDS;
//   phase 3   one TASK per (row, |kz|): with XY = (a, b), Z[|kz|] = (c, d) the four sums
//             S_ac, S_bd, S_ad, S_bc over the site-states give both members of the pair,
//                 delta(+kz) = (S_ac - S_bd,  S_ad + S_bc),   delta(-kz) = (S_ac + S_bd,  S_bc - S_ad),
//             i.e. 4 FMAs and 2 LDS reads per site-state for TWO k-vectors (the per-k form above costs
//             two complex products and three LDS reads per site-state for ONE).
// Same semantics as recip_kernel: COMMIT = false returns u_new (and u_old with BOTH), COMMIT = true
// applies A <- A + delta and then the coordinate / count update.
// Dynamic LDS: 2 n1 ktot (1-D tables) + n_rows 2 n1 (XY) complex entries + n1 charges + the row table.
// ------------------------------------------------------------------------------------------
struct RecipTask {                // host-side description of a task (the device reads the packed arrays below)
    int kp, km;                   // k index of (kx, ky, +j) and of (kx, ky, -j); -1: absent (j = 0 has no partner)
    int row, j;                   // (kx, ky) row and |kz|
};
// device form of a task: trj = row << 8 | j | kTaskHasP | kTaskHasM, tw = {ff W (+j), ff W (-j)} (0 where absent);
// A of task t sits in slots 2t, 2t + 1 of the replica
constexpr int kTaskHasP = 1 << 30;
constexpr int kTaskHasM = 1 << 29;

struct RecipRow {
    int kx, ky;
};

// Commit by accept mask: the launch covers the candidates of the lane's last trial (their RecipItems are still
// on the device) and every workgroup whose bit is clear leaves at once -- no item list has to be uploaded.
constexpr int kAcceptWords = 128;                     // 4096 candidates per launch
struct AcceptBits {
    unsigned w[kAcceptWords];
};

#ifndef MGPU_RECIP_MINWAVES
#define MGPU_RECIP_MINWAVES 6   // six 4-wave workgroups per CU, 1536 items resident at once (round 4: one-task chunks and the short
                                // sincos leave the sweep at 74 VGPRs; at five 39.8 us, at six 37.9, at seven -- spills -- 39.8-40.3)
#endif
#ifndef MGPU_COMMIT_MINWAVES
#define MGPU_COMMIT_MINWAVES 5  // the commit needs 76 VGPRs: five workgroups per CU (measured 26.8 -> 24.7 us at the SPC/E box, 17.2 -> 16.0 us
                                // at the framework box; round 3's k sweep, chunks of two, at five: 25.8 -> 28.8 us, spills)
#endif
// Phase 3: a thread's tasks are taken in chunks (recip_chunk_tasks) with TWO chunks in flight (the next chunk's A(k),
// weights and task words are requested before the current chunk's arithmetic).  A thread visits its tasks in ascending
// order whatever the chunking, so the sums are the same bits.  (Round-3 measurements of the alternatives -- one chunk of
// 3 / 4 / 5, pipelined 3 + 3, A(k) requested before the tables, cache prefetch, staggered starts: LABNOTES.md.)
#ifndef MGPU_RECIP_TASK_CHUNK
#define MGPU_RECIP_TASK_CHUNK 1
#endif
#ifndef MGPU_RECIP_COMMIT_CHUNK
#define MGPU_RECIP_COMMIT_CHUNK 2
#endif
constexpr int kRecipTaskChunk = MGPU_RECIP_TASK_CHUNK, kRecipCommitChunk = MGPU_RECIP_COMMIT_CHUNK;

// Acceptance decided on the device (recip_rows_kernel<false, true, true>): the k sweep's workgroup is the last kernel of a
// candidate's trial, so once its two reciprocal energies are summed thread 0 has everything mc_acceptance_probability
// needs (monte_carlo_utils.f90:184-226) -- the pair entries' split partials (summed in split order, framework record last,
// exactly as trial_wait does on the host), ewald_self and intra_coulomb on the side where the molecule exists
// (monte_carlo_utils.f90:298-299, :378-379) -- and an accepted candidate is committed by the SAME workgroup from the phase
// tables it already holds: a second pass over its replica's A(k) (A <- A + delta, the stand-alone commit's arithmetic) and the
// coordinate / frame / count update.  Offsets are in doubles into the lane's result block.
struct DecideItem {
    int old_off, old_stride, old_ns, old_extra;     // old-state pair entry: ns = -1 none; extra = -1 none
    int new_off, new_stride, new_ns, new_extra;
    int intra;                                      // index of the candidate's intra_coulomb result, -1 none
    int kind;                                       // MGPU_MOVE / CREATION / DELETION
    double self;                                    // ewald_self of the candidate's residue type
    double pref;                                    // acceptance prefactor: 1 (moves), phi V / (N + 1), N / (phi V)
    double u;                                       // the uniform number of the test
};
struct DecideArgs {
    const DecideItem *items;
    const double *out;                              // the lane's result block (partials | u_old | u_new | intra | extra)
    const double *intra;
    int *accepted;                                  // [n] flags, copied out with the energies
    double temperature;
};
// old%total and new%total as the host driver forms them: components added in the order non_coulomb, coulomb, recip_coulomb,
// ewald_self, intra_coulomb (mc_farm.f90 resolve_and_commit)
__device__ inline bool decide_candidate(const DecideItem &d, const DecideArgs &g, double u_old, double u_new) {
    double o[5] = {0.0, 0.0, u_old, 0.0, 0.0}, w[5] = {0.0, 0.0, u_new, 0.0, 0.0};
    auto entry = [&](int off, int stride, int ns, int extra, double &lj, double &cc) {
        double a = 0.0, b = 0.0;
        const double *p = g.out + off;
        for (int s2 = 0; s2 < ns; ++s2) { a += p[stride * s2]; b += p[stride * s2 + 1]; }
        if (extra >= 0) { a += g.out[extra]; b += g.out[extra + 1]; }
        lj = a;
        cc = b * kEps0InvEvA / kKbEvK;                                        // energy_utils.f90:440
    };
    if (d.old_ns >= 0) entry(d.old_off, d.old_stride, d.old_ns, d.old_extra, o[0], o[1]);
    if (d.new_ns >= 0) entry(d.new_off, d.new_stride, d.new_ns, d.new_extra, w[0], w[1]);
    if (d.kind == 1) { w[3] = d.self; w[4] = g.intra[d.intra]; }
    if (d.kind == 2) { o[3] = d.self; o[4] = g.intra[d.intra]; }
    double e_old = 0.0, e_new = 0.0;
    for (int k = 0; k < 5; ++k) { e_old = e_old + o[k]; e_new = e_new + w[k]; }
    const double x = d.pref * exp(-(e_new - e_old) / g.temperature);          // min(1, x); a NaN (overlap) rejects
    return x >= 1.0 || d.u <= x;
}

// ---- the row-form update in pieces (shared by recip_rows_kernel and chain_window_kernel).  All of them are executed by
//      the first kBlock threads of a workgroup (`tid` < kBlock: `active`); every thread of the workgroup must reach the
//      barriers inside.
// LDS view of one item: 1-D tables [nss][ktot] | XY [n_rows][nss] | charges [n1]
struct RecipLds {
    double2 *tab, *xy;
    double *q;
    int n1, nss, ktot, kofs1, kofs2;
    bool use_new, use_old, two_sets;
};
// site-states: the new sites and the old sites of a move (2 n1); an insertion, a deletion or A += S(sites) carries ONE set
// (n1) -- half the table entries, XY products and inner-loop terms.  kind 5 (chain windows only): the reference's
// deletion as written: A gains the terms of the sites in the candidate row, the coordinates lose slot m.
__device__ __forceinline__ RecipLds recip_lds_view(const Topo &tp, const BoxDev &bx, const RecipItem &it, int n_rows, double2 *s_tab) {
    RecipLds v;
    v.use_new = (it.kind == 0 /*MOVE*/ || it.kind == 1 /*CREATION*/ || it.kind == 4 /*FOURIER_ADD*/ || it.kind == 5 /*DELETION as written*/);
    v.use_old = (it.kind == 0 /*MOVE*/ || it.kind == 2 /*DELETION*/);
    v.two_sets = v.use_new && v.use_old;
    v.n1 = tp.n1[it.t];
    v.nss = v.two_sets ? 2 * v.n1 : v.n1;
    v.kofs1 = bx.kmax[0] + 1;
    v.kofs2 = bx.kmax[0] + bx.kmax[1] + 2;
    v.ktot = bx.kmax[0] + bx.kmax[1] + bx.kmax[2] + 3;
    v.tab = s_tab;
    v.xy = s_tab + v.nss * v.ktot;
    v.q = reinterpret_cast<double *>(v.xy + n_rows * v.nss);
    return v;
}

// phases 1 and 2 in pieces WITHOUT barriers (recip_rows_tables puts them together; recip_rows2_kernel runs two items through
// each phase between one pair of barriers).  cand_row = the item's candidate row (new sites), unused without one.
// `after_loads()` runs once per active thread after the loads phases 1 and 2 wait for (the thread's first table entry's
// coordinates, the charge, its first row) have been requested and before the first wait: the place where the kernels
// request their first chunks of A(k) (recip_rows_prefetch), so that the wait for the small loads leaves the large ones
// in flight (the memory counter retires in order) and phases 1 and 2 run under them.
template <class Hook>
__device__ __forceinline__ RecipRow recip_rows_phase1(const Topo &tp, const BoxDev &bx, const double *__restrict__ pos,
                                                      const double *__restrict__ res_q, const RecipRow *__restrict__ rows, int n_rows,
                                                      const RecipItem &it, const double *__restrict__ cand_row, const RecipLds &v,
                                                      int tid, bool active, Hook &&after_loads) {
    const double *px = pos + (size_t)it.replica * 3 * tp.n_cap_atoms;
    const double *py = px + tp.n_cap_atoms, *pz = py + tp.n_cap_atoms;
    const int n1 = v.n1, nss = v.nss, ktot = v.ktot;
    // with no set at all (MGPU_NONE: the energy of A as it is) the entries are zero and phase 3 adds nothing
    const bool used = v.use_new || v.use_old;
    RecipRow r_first{0, 0};
    if (active) {
        // phase 1: entry (s, axis, k >= 0) at tab[s * ktot + kofs[axis] + k]; s = set * n1 + a with both sets, s = a with one
        // (set 0 = the new sites, set 1 = the old ones).  e / ktot by a multiplication: floor(e M / 2^32) with
        // M = ceil(2^32 / ktot) is exact for e < 2^32 / ktot
        const unsigned ktot_magic = 0xffffffffu / (unsigned)ktot + 1u;
        const int n_ent = nss * ktot;
        auto fetch = [&](int e, double &x, double &y, double &z) {
            const int s = (int)__umulhi((unsigned)e, ktot_magic);
            const int set = v.two_sets ? (s >= n1 ? 1 : 0) : (v.use_old ? 1 : 0), a = s - (s >= n1 ? n1 : 0);
            x = 0.0; y = 0.0; z = 0.0;
            if (used) {
                if (set == 0) {
                    const double *c = cand_row + (size_t)a * 3;
                    x = c[0]; y = c[1]; z = c[2];
                } else {
                    const int j = atom_slot(tp, it.t, it.m, a);
                    x = px[j]; y = py[j]; z = pz[j];
                }
            }
        };
        double x = 0.0, y = 0.0, z = 0.0, q = 0.0;
        int e = tid;
        if (e < n_ent) fetch(e, x, y, z);
        if (tid < n1) q = res_q[it.t * tp.max_atom + tid];
        if (tid < n_rows) r_first = rows[tid];
        after_loads();
        for (; e < n_ent;) {
            const int s = (int)__umulhi((unsigned)e, ktot_magic), kk = e - s * ktot;
            const int axis = (kk >= v.kofs2) ? 2 : (kk >= v.kofs1 ? 1 : 0);
            const int k0 = axis == 2 ? v.kofs2 : (axis == 1 ? v.kofs1 : 0);
            v.tab[e] = used ? phase_entry(atom_phase(bx, axis, x, y, z), kk - k0) : make_double2(0.0, 0.0);
            e += kBlock;
            if (e < n_ent) fetch(e, x, y, z);
        }
        if (tid < n1) v.q[tid] = q;
        for (int a = tid + kBlock; a < n1; a += kBlock) v.q[a] = res_q[it.t * tp.max_atom + a];
    }
    return r_first;
}
// phase 2: XY[row][s] = (+q for the new sites, -q for the old ones) * X[kx] * Y[ky]   (ewald_energy.f90:241-256)
// (one thread per row, the site-states in its inner loop: the row's indices are read once -- straight from the
//  launch's row list -- and nothing is divided); r_first = the thread's first row as phase 1 loaded it
__device__ __forceinline__ void recip_rows_phase2(const RecipRow *__restrict__ rows, int n_rows, const RecipLds &v, int tid, bool active,
                                                  const RecipRow r_first) {
    const int n1 = v.n1, nss = v.nss, ktot = v.ktot;
    const bool used = v.use_new || v.use_old;
    if (active) {
        for (int row = tid; row < n_rows; row += kBlock) {
            const RecipRow r = row == tid ? r_first : rows[row];
            const int aky = r.ky < 0 ? -r.ky : r.ky;
            const double2 *tx = v.tab + r.kx, *ty = v.tab + v.kofs1 + aky;
            double2 *out = v.xy + row * nss;
            for (int s = 0; s < nss; ++s) {
                const int set = v.two_sets ? (s >= n1 ? 1 : 0) : (v.use_old ? 1 : 0), a = s - (s >= n1 ? n1 : 0);
                double2 xy = make_double2(0.0, 0.0);
                if (used) {
                    double2 Y = ty[s * ktot];
                    if (r.ky < 0) Y.y = -Y.y;
                    xy = cmul(tx[s * ktot], Y);
                    const double q = set == 0 ? v.q[a] : -v.q[a];
                    xy.x *= q; xy.y *= q;
                }
                out[s] = xy;
            }
        }
    }
}
// phases 1 and 2 of ONE item (two workgroup barriers inside)
template <class Hook>
__device__ __forceinline__ void recip_rows_tables(const Topo &tp, const BoxDev &bx, const double *__restrict__ pos,
                                                  const double *__restrict__ res_q, const RecipRow *__restrict__ rows, int n_rows,
                                                  const RecipItem &it, const double *__restrict__ cand_row, const RecipLds &v,
                                                  int tid, bool active, Hook &&after_loads) {
    const RecipRow r_first = recip_rows_phase1(tp, bx, pos, res_q, rows, n_rows, it, cand_row, v, tid, active, after_loads);
    __syncthreads();
    recip_rows_phase2(rows, n_rows, v, tid, active, r_first);
    __syncthreads();
}

// phase 3: one pass over the replica's A(k) by the first kBlock threads.  STORE = false: acc += ff W |A + delta|^2 and, with
// BOTH, acc0 += ff W |A|^2 (the reference's ComputeOldEnergy call, delta = 0); STORE = true: A <- A + delta.
// A(k) (32 contiguous bytes per task, the bulk of the kernel's memory traffic), ff*W and the task words of a whole chunk
// are requested before any of them is used; none of the addresses depends on a load OR on the tables, so the first two
// chunks are requested (recip_rows_prefetch) BEFORE the tables are built: the workgroup's one long memory round trip
// runs under phases 1 and 2 instead of after them.
template <int CH>
struct RecipChunk {
    int rj[CH];
    double2 Ap[CH], Am[CH], w[CH];
};
template <int CH>
struct RecipInFlight {
    RecipChunk<CH> ch0, ch1;
};
// tasks per chunk: ONE for the energy sweeps (two tasks of a thread in flight: 89 VGPRs, five workgroups per CU), two for
// the commit (76 VGPRs with them, five workgroups as well).  Measured round 4, k sweep / commit in us at the SPC/E, CO2 and
// framework boxes: chunks of two at four workgroups 43.5 / 39.4 / 32.4, of one at five 40.5 / 35.9 / 29.6 (at six: spills,
// 57.8 / 62.9 / 34.7); the commit with chunks of one 41.4-43.0 against 40.0.
template <bool STORE>
constexpr int recip_chunk_tasks() { return STORE ? kRecipCommitChunk : kRecipTaskChunk; }

template <bool STORE, int CH>
__device__ __forceinline__ void recip_load_chunk(RecipChunk<CH> &ch, const int *__restrict__ trj, const double2 *__restrict__ tw, int n_tasks,
                                                 const double2 *__restrict__ A, int t0) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int t = t0 + c * kBlock;
        const bool in = t < n_tasks;
        ch.rj[c] = in ? trj[t] : 0;                                // filler: row 0, j 0, nothing present
        ch.Ap[c] = in ? A[2 * t] : make_double2(0.0, 0.0);
        ch.Am[c] = in ? A[2 * t + 1] : make_double2(0.0, 0.0);
        ch.w[c] = (in && !STORE) ? tw[t] : make_double2(0.0, 0.0);
    }
}
template <bool STORE, int CH>
__device__ __forceinline__ void recip_rows_prefetch(RecipInFlight<CH> &f, const int *__restrict__ trj, const double2 *__restrict__ tw, int n_tasks,
                                                    const double2 *__restrict__ A, int tid) {
    constexpr int kStride = kBlock * CH;
    if (tid < n_tasks) recip_load_chunk<STORE>(f.ch0, trj, tw, n_tasks, A, tid);
    if (tid + kStride < n_tasks) recip_load_chunk<STORE>(f.ch1, trj, tw, n_tasks, A, tid + kStride);
}

// `f` holds the thread's first two chunks (recip_rows_prefetch with the same arguments)
// ALT (energy sweeps only): A + delta is ALSO stored, into `A_alt` -- another buffer of the replica's layout -- with the
// commit's arithmetic, so that a later acceptance only has to make that buffer the replica's current one
// (farm_window_kernel: its k role cannot know the verdict, and the workgroup that learns it has no phase tables).
// ALT = 2: the same with agent-scope write-through (`sc1`) stores, for a reader in ANOTHER workgroup of the same launch
// that loads with `sc1` (chain_window_kernel's resolving workgroup copies the accepted step's buffer into A).
template <bool STORE, bool BOTH, int CH, int ALT = 0>
__device__ __forceinline__ void recip_rows_pass(const RecipLds &v, const int *__restrict__ trj, const double2 *__restrict__ tw, int n_tasks,
                                                double2 *__restrict__ A, int tid, RecipInFlight<CH> &f, double &acc, double &acc0,
                                                double2 *__restrict__ A_alt = nullptr) {
    constexpr int kRecipStride = kBlock * CH;
    const int nss = v.nss, ktot = v.ktot;
    const double2 *zt = v.tab + v.kofs2;
    // the tasks of one chunk: a thread's tasks are visited in ascending order whatever the chunk size, so the sums do
    // not depend on it
    auto compute_chunk = [&](const RecipChunk<CH> &ch, int t0) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            // past the end for the whole wave (its first lane holds the smallest task): a filler adds exact zeros
            if (__builtin_amdgcn_readfirstlane(t0 + c * kBlock) >= n_tasks) continue;
            const double2 *xy = v.xy + ((ch.rj[c] >> 8) & 0xfffff) * nss;
            const double2 *z = zt + (ch.rj[c] & 0xff);
            double sac = 0.0, sbd = 0.0, sad = 0.0, sbc = 0.0;
            auto term = [&](const double2 p, const double2 q) {
                sac = fma(p.x, q.x, sac);
                sbd = fma(p.y, q.y, sbd);
                sad = fma(p.x, q.y, sad);
                sbc = fma(p.y, q.x, sbc);
            };
            // site-states two at a time (the four LDS reads of a pair are requested together), then the odd one; the
            // order of the sums is s = 0, 1, 2, ... either way
            int s = 0;
            for (; s + 2 <= nss; s += 2) {
                const double2 p0 = xy[s], p1 = xy[s + 1];
                const double2 q0 = z[s * ktot], q1 = z[(s + 1) * ktot];
                term(p0, q0); term(p1, q1);
            }
            for (; s < nss; ++s) term(xy[s], z[s * ktot]);
            const double wp = ch.w[c].x, wm = ch.w[c].y;
            // explicit fma forms: every kernel that forms these sums must produce the same bits, and a contraction left to
            // the compiler may pick a different product to fuse in a different kernel
            if (BOTH && !STORE) acc0 += fma(wp, fma(ch.Ap[c].x, ch.Ap[c].x, ch.Ap[c].y * ch.Ap[c].y), wm * fma(ch.Am[c].x, ch.Am[c].x, ch.Am[c].y * ch.Am[c].y));
            const double npx = ch.Ap[c].x + (sac - sbd), npy = ch.Ap[c].y + (sad + sbc);
            const double nmx = ch.Am[c].x + (sac + sbd), nmy = ch.Am[c].y + (sbc - sad);
            if (STORE) {
                const int t = t0 + c * kBlock;
                if (t < n_tasks) {        // absent members stay zero
                    A[2 * t] = (ch.rj[c] & kTaskHasP) ? make_double2(npx, npy) : make_double2(0.0, 0.0);
                    A[2 * t + 1] = (ch.rj[c] & kTaskHasM) ? make_double2(nmx, nmy) : make_double2(0.0, 0.0);
                }
            } else {
                acc += fma(wp, fma(npx, npx, npy * npy), wm * fma(nmx, nmx, nmy * nmy));   // ewald_energy.f90:259-266
                if constexpr (ALT != 0) {
                    const int t = t0 + c * kBlock;
                    if (t < n_tasks) {    // the commit's stores (STORE above), to the other buffer
                        const double2 vp = (ch.rj[c] & kTaskHasP) ? make_double2(npx, npy) : make_double2(0.0, 0.0);
                        const double2 vm = (ch.rj[c] & kTaskHasM) ? make_double2(nmx, nmy) : make_double2(0.0, 0.0);
                        if constexpr (ALT == 2) {
                            double *d = reinterpret_cast<double *>(A_alt + 2 * t);
                            __hip_atomic_store(d + 0, vp.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(d + 1, vp.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(d + 2, vm.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(d + 3, vm.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        } else {
                            A_alt[2 * t] = vp;
                            A_alt[2 * t + 1] = vm;
                        }
                    }
                }
            }
        }
    };
    // two chunks in flight: the chunk after the next is requested as soon as its registers are free
    int t0 = tid;
    while (t0 < n_tasks) {
        compute_chunk(f.ch0, t0);
        const int t1 = t0 + kRecipStride;
        if (t1 >= n_tasks) break;
        const int t2 = t1 + kRecipStride;
        if (t2 < n_tasks) recip_load_chunk<STORE>(f.ch0, trj, tw, n_tasks, A, t2);
        compute_chunk(f.ch1, t1);
        if (t2 >= n_tasks) break;
        if (t2 + kRecipStride < n_tasks) recip_load_chunk<STORE>(f.ch1, trj, tw, n_tasks, A, t2 + kRecipStride);
        t0 = t2;
    }
}

// The coordinate / frame / count part of a commit (every read of the old coordinates happened before the first barrier
// of recip_rows_tables).  it.m / it.aux are final here (see recip_commit_target).
__device__ __forceinline__ void recip_commit_tail(const Topo &tp, double *__restrict__ pos, int *__restrict__ nmol, const RecipItem &it,
                                                  const double *__restrict__ cand_row, int tid) {
    double *px = pos + (size_t)it.replica * 3 * tp.n_cap_atoms;
    double *py = px + tp.n_cap_atoms, *pz = py + tp.n_cap_atoms;
    const int n1 = tp.n1[it.t];
    if (it.kind == 0 || it.kind == 1) {
        if (tid < n1) {
            const double *c = cand_row + (size_t)tid * 3;
            const int j = atom_slot(tp, it.t, it.m, tid);
            px[j] = c[0]; py[j] = c[1]; pz[j] = c[2];
        }
    } else if (it.kind == 2 || it.kind == 5) {
        const int last = it.aux;          // swap-with-last, delete_molecule.f90:107-114: slot m <- slot (new count)
        if (tid < n1 && last != it.m) {
            const int j = atom_slot(tp, it.t, it.m, tid), jl = atom_slot(tp, it.t, last, tid);
            px[j] = px[jl]; py[j] = py[jl]; pz[j] = pz[jl];
        }
    }
    // molecule frames, where the engine keeps them: a device-built move / insertion writes its com and offsets back,
    // a deletion moves the last molecule's frame with its sites
    if (tp.com) {
        const size_t rep3 = (size_t)it.replica * 3;
        double *fcom = tp.com + rep3 * tp.n_mol_slots + tp.mol_off[it.t];
        double *foff = tp.off + rep3 * tp.n_cap_atoms;
        if ((it.kind == 0 || it.kind == 1) && it.frame > 0) {
            const double *fr = cand_row + (size_t)it.frame * 3;
            if (tid < 3) fcom[(size_t)tid * tp.n_mol_slots + it.m] = fr[tid];
            if (tid < n1) {
                const int j = atom_slot(tp, it.t, it.m, tid);
                for (int d = 0; d < 3; ++d) foff[(size_t)d * tp.n_cap_atoms + j] = fr[(1 + tid) * 3 + d];
            }
        } else if ((it.kind == 2 || it.kind == 5) && it.aux != it.m) {
            const int last = it.aux;
            if (tid < 3) fcom[(size_t)tid * tp.n_mol_slots + it.m] = fcom[(size_t)tid * tp.n_mol_slots + last];
            if (tid < n1) {
                const int j = atom_slot(tp, it.t, it.m, tid), jl = atom_slot(tp, it.t, last, tid);
                for (int d = 0; d < 3; ++d) foff[(size_t)d * tp.n_cap_atoms + j] = foff[(size_t)d * tp.n_cap_atoms + jl];
            }
        }
    }
    if (tid == 0 && (it.kind == 1 || it.kind == 2 || it.kind == 5)) nmol[it.replica * tp.n_res + it.t] = it.aux;
}
// a trial's item carries no target slot / new count: take them from the replica's live count
__device__ __forceinline__ void recip_commit_target(const Topo &tp, const int *__restrict__ nmol, RecipItem &it) {
    if (it.kind != 1 && it.kind != 2 && it.kind != 5) return;
    const int nm = nmol[it.replica * tp.n_res + it.t];
    if (it.kind == 1) { it.m = nm; it.aux = nm + 1; }     // appended (monte_carlo.f90:63, create_molecule.f90:64)
    else it.aux = nm - 1;                                 // swap-with-last target
}

template <bool COMMIT, bool BOTH, bool DECIDE = false>
__global__ __launch_bounds__(kBlock, COMMIT ? MGPU_COMMIT_MINWAVES : MGPU_RECIP_MINWAVES) void recip_rows_kernel(
    Topo tp, BoxDev bx, double *__restrict__ pos, int *__restrict__ nmol, const double *__restrict__ res_q,
    const int *__restrict__ trj, const double2 *__restrict__ tw, int n_tasks, const RecipRow *__restrict__ rows, int n_rows,
    double2 *__restrict__ A_base, const RecipItem *__restrict__ items,
    const double *__restrict__ cand_sites, int site_stride, double *__restrict__ u_new, double *__restrict__ u_old,
    AcceptBits accept, int use_accept, DecideArgs dec) {
    static_assert(!DECIDE || (!COMMIT && BOTH), "the deciding form is the old + new k sweep");
    extern __shared__ double2 s_tab[];
    __shared__ double s_red[2 * kWavesPerBlock];
    __shared__ int s_flag;

    RecipItem it = items[blockIdx.x];
    if (COMMIT && use_accept) {
        if (!((accept.w[blockIdx.x >> 5] >> (blockIdx.x & 31)) & 1u)) return;        // uniform per workgroup
        recip_commit_target(tp, nmol, it);
    }
    const double *cand_row = cand_sites + (size_t)(it.src < 0 ? 0 : it.src) * site_stride * 3;
    const RecipLds v = recip_lds_view(tp, bx, it, n_rows, s_tab);
    double2 *A = A_base + (size_t)it.replica * bx.n_slots;
    const int tid = threadIdx.x;

    // the energy sweeps request their first chunks of A(k) under the table phases; the commit, whose registers buy it a
    // fifth workgroup per CU, after them (measured at the SPC/E box: 40.9 us against 44.3 at four and 70 with spills)
    RecipInFlight<recip_chunk_tasks<COMMIT>()> inflight;
    // An energy sweep's waves issue at raised priority while they build their tables (short, arithmetic, and what stands
    // between the workgroup and its streaming pass) and at the default during the pass, where they mostly wait for A(k):
    // the six workgroups of a CU fall out of step sooner.  SPC/E box 38.2 -> 36.7-37.6 us on three boxes (0.54 -> 0.55-0.56 of
    // HBM peak), CO2 box 34.6 -> 34.5-34.9, framework box unchanged; raised priority for the PASS instead: 37.6 / 34.2.
    if (!COMMIT) __builtin_amdgcn_s_setprio(3);
    recip_rows_tables(tp, bx, pos, res_q, rows, n_rows, it, cand_row, v, tid, true,
                      [&] { if (!COMMIT) recip_rows_prefetch<COMMIT>(inflight, trj, tw, n_tasks, A, tid); });
    if (COMMIT) recip_rows_prefetch<COMMIT>(inflight, trj, tw, n_tasks, A, tid);
    double acc = 0.0, acc0 = 0.0;
    if (!COMMIT) __builtin_amdgcn_s_setprio(0);
    recip_rows_pass<COMMIT, BOTH>(v, trj, tw, n_tasks, A, tid, inflight, acc, acc0);

    if (!COMMIT) {
        acc = wave_sum(acc);
        if (BOTH) acc0 = wave_sum(acc0);
        if ((tid & 63) == 0) { s_red[2 * (tid >> 6)] = acc; s_red[2 * (tid >> 6) + 1] = acc0; }
        __syncthreads();
        if (tid == 0) {
            double u = 0.0, u0 = 0.0;
            for (int wv = 0; wv < kWavesPerBlock; ++wv) { u += s_red[2 * wv]; u0 += s_red[2 * wv + 1]; }
            const double e_new = u * kEps0InvEvA / kKbEvK * kTwoPi / bx.volume;   // ewald_energy.f90:272
            const double e_old = u0 * kEps0InvEvA / kKbEvK * kTwoPi / bx.volume;
            u_new[blockIdx.x] = e_new;
            if (BOTH) u_old[blockIdx.x] = e_old;
            if constexpr (DECIDE) {
                const bool yes = decide_candidate(dec.items[blockIdx.x], dec, e_old, e_new);
                dec.accepted[blockIdx.x] = yes ? 1 : 0;
                s_flag = yes ? 1 : 0;
            }
        }
    }
    if constexpr (DECIDE) {
        __syncthreads();
        if (!s_flag) return;                                          // uniform per workgroup
        recip_commit_target(tp, nmol, it);
        // A <- A + delta from the tables still in LDS: the stand-alone commit's pass (same sums, same bits); this
        // workgroup has just read the replica's A(k), so the second read comes from L2 / the Infinity Cache
        RecipInFlight<kRecipTaskChunk> again;       // (chunks of one here too: the registers are the sweep's)
        recip_rows_prefetch<true>(again, trj, tw, n_tasks, A, tid);
        recip_rows_pass<true, false>(v, trj, tw, n_tasks, A, tid, again, acc, acc0);
    }
    if (COMMIT || DECIDE) recip_commit_tail(tp, pos, nmol, it, cand_row, tid);
}

// ------------------------------------------------------------------------------------------
// Row form for molecules of MANY sites ("wide": a few dozen sites -- the 24-site adsorbate of the tests, a typical rigid
// organic adsorbate), whose XY table [rows][site-states] does not fit LDS at once.  The 1-D phase tables of ALL the item's
// site-states stay in LDS; the rows pass through the XY table a TILE of rows at a time, and with them the tasks of those
// rows (a row's tasks are contiguous in the task list: row_first[r] .. row_first[r + 1]).  Per task and site-state this is
// the row form's arithmetic, expression for expression -- 2 LDS reads and 4 FMAs for TWO k-vectors, against two complex
// products and three reads for ONE in the per-k form (recip_kernel) that such molecules took before: measured round 5,
// 1024 items of the 24-site adsorbate at Nk = 8936, 635 us per launch there (profiles/r05/recip_many_sites.txt).
// Molecules whose phase tables alone exceed the budget (hundreds of sites) keep the per-k form with its site tiles.
// Dynamic LDS: tab [nss][ktot] | xy [rows_per_tile][nss] | signed charges [nss].
// ------------------------------------------------------------------------------------------
// MFMA: per item the row form's sums ARE four real matrix products [kz][site-state] x [site-state][row] (sac, sbd, sad, sbc
// of the pass above), so each wave takes tiles of 16 rows x 16 kz through v_mfma_f64_16x16x4_f64, twelve steps of four
// site-states for a 24-site move: operand traffic 16 B per lane per 2 x 2048 flop instead of 32 B per 8 -- the vector form is
// LDS-bound at a quarter of the fp64 peak before bank conflicts (measured 0.07).  The XY factor of a (row, site-state) is
// formed in registers from the 1-D tables (no XY table, no row tiles, no barrier after phase 1).  Needs every row's tasks
// to be a run of consecutive kz (the engine checks: rows_contiguous).  A sum over site-states in the matrix unit's order:
// the trial and the commit pass share it, so A + delta is the same in both.
template <bool COMMIT, bool BOTH, bool MFMA = false, bool TILED = false>
__global__ __launch_bounds__(kBlock, 2) void recip_rows_wide_kernel(
    Topo tp, BoxDev bx, double *__restrict__ pos, int *__restrict__ nmol, const double *__restrict__ res_q,
    const int *__restrict__ trj, const double2 *__restrict__ tw, const RecipRow *__restrict__ rows, const int *__restrict__ row_first,
    int n_rows, int rows_per_tile, int nss_max, double2 *__restrict__ A_base, const RecipItem *__restrict__ items,
    const double *__restrict__ cand_sites, int site_stride, double *__restrict__ u_new, double *__restrict__ u_old,
    double *__restrict__ site_tile_sums, int n_tasks) {
    extern __shared__ double2 s_tab[];
    __shared__ double s_red[2 * kWavesPerBlock];
    const RecipItem it = items[blockIdx.x];
    const int tid = threadIdx.x;
    const int n1 = tp.n1[it.t];
    const bool use_new = (it.kind == 0 /*MOVE*/ || it.kind == 1 /*CREATION*/ || it.kind == 4 /*FOURIER_ADD*/);
    const bool use_old = (it.kind == 0 /*MOVE*/ || it.kind == 2 /*DELETION*/);
    const bool two_sets = use_new && use_old, used = use_new || use_old;
    const int nss = two_sets ? 2 * n1 : n1;
    const int kofs1 = bx.kmax[0] + 1, kofs2 = bx.kmax[0] + bx.kmax[1] + 2, ktot = bx.kmax[0] + bx.kmax[1] + bx.kmax[2] + 3;
    double2 *tab = s_tab, *xyt = s_tab + (size_t)nss_max * ktot;
    double *sq = reinterpret_cast<double *>(xyt + (size_t)rows_per_tile * nss_max);
    double *px = pos + (size_t)it.replica * 3 * tp.n_cap_atoms;
    double *py = px + tp.n_cap_atoms, *pz = py + tp.n_cap_atoms;
    const double *cand_row = cand_sites + (size_t)(it.src < 0 ? 0 : it.src) * site_stride * 3;
    // ---- phase 1, once: entry (s, axis, k >= 0) at tab[s * ktot + kofs[axis] + k]; s = set * n1 + a with both sets (set 0 =
    //      the new sites, 1 = the old ones), s = a with one
    // (matrix-unit form: the site-states padded to a multiple of four with entries of 0 and charge 0 -- its steps of four
    //  site-states then need no mask)
    const int nss_fill = MFMA ? ((nss + 3) & ~3) : nss;
    // Matrix-unit form, molecules of ANY size: the site-states pass through LDS in tiles of nss_max (a multiple of four; one
    // tile where everything fits); the four sums of every task go from tile to tile through site_tile_sums[item][task][4]
    // (written by the first tile, added to by the middle ones, read by the last, which finishes the task with A(k)); a task
    // is always worked by the same lane, tile after tile.
    // (TILED is a template flag so that the one-tile kernel keeps its 118 registers: with the carried sums compiled in it took 156)
    const int tile_ss = MFMA ? nss_max : nss_fill, n_st = (MFMA && TILED) ? (nss_fill + tile_ss - 1) / tile_ss : 1;
    [[maybe_unused]] int4 *rowmeta = reinterpret_cast<int4 *>(sq + nss_max);
    double2 *A = A_base + (size_t)it.replica * bx.n_slots;
    const double2 *zt = tab + kofs2;
    double acc = 0.0, acc0 = 0.0;
    for (int st = 0; st < n_st; ++st) {
    const int ss0 = (MFMA && TILED) ? st * tile_ss : 0, ssn = (MFMA && TILED) ? min(tile_ss, nss_fill - ss0) : nss_fill;
    if (st > 0) __syncthreads();                        // every wave has left the tables of the tile before
    for (int e = tid; e < ssn * ktot; e += kBlock) {
        const int sl = e / ktot, kk = e - sl * ktot, s = ss0 + sl;
        if (MFMA && s >= nss) { tab[e] = make_double2(0.0, 0.0); continue; }
        const int set = two_sets ? (s >= n1 ? 1 : 0) : (use_old ? 1 : 0), a = s - (s >= n1 ? n1 : 0);
        double x = 0.0, y = 0.0, z = 0.0;
        if (used) {
            if (set == 0) { const double *c = cand_row + (size_t)a * 3; x = c[0]; y = c[1]; z = c[2]; }
            else { const int j = atom_slot(tp, it.t, it.m, a); x = px[j]; y = py[j]; z = pz[j]; }
        }
        const int axis = (kk >= kofs2) ? 2 : (kk >= kofs1 ? 1 : 0);
        const int k0 = axis == 2 ? kofs2 : (axis == 1 ? kofs1 : 0);
        tab[e] = used ? phase_entry(atom_phase(bx, axis, x, y, z), kk - k0) : make_double2(0.0, 0.0);
    }
    for (int sl = tid; sl < ssn; sl += kBlock) {
        const int s = ss0 + sl;
        if (MFMA && s >= nss) { sq[sl] = 0.0; continue; }
        const int set = two_sets ? (s >= n1 ? 1 : 0) : (use_old ? 1 : 0), a = s - (s >= n1 ? n1 : 0);
        const double q = res_q[it.t * tp.max_atom + a];
        sq[sl] = (set == 0 ? q : -q) * (used ? 1.0 : 0.0);   // + for the new sites, - for the old ones (ewald_energy.f90:241-256)
    }
    // matrix-unit form: every row's {kx, ky, first task, first kz | tasks << 8} beside the tables, so that a tile's
    // addresses cost one LDS read instead of a chain of three global loads per tile
    if constexpr (MFMA) {
        if (st == 0)
            for (int rr = tid; rr < n_rows; rr += kBlock) {
                const RecipRow r = rows[rr];
                const int t0 = row_first[rr], t1 = row_first[rr + 1];
                const int j0 = t1 > t0 ? (trj[t0] & 0xff) : 0;
                rowmeta[rr] = make_int4(r.kx, r.ky, t0, j0 | ((t1 - t0) << 8));
            }
    }
    __syncthreads();
    if constexpr (MFMA) {
        typedef double double4v __attribute__((ext_vector_type(4)));
        const int lane = tid & 63, wave = tid >> 6;
        const int li = lane & 15, lk = lane >> 4;          // operand A: (kz li, site-state lk); B: (site-state lk, row li); D: (kz lk + 4 i, row li)
        // kz columns in tiles of 16; a tile of rows takes as many as its longest row needs (kmax_z = 16: 17 columns, the 17th
        // in the few rows around kx = ky = 0 only -- a second column tile for every tile of rows would double the work)
        const int n_rt = (n_rows + 15) >> 4;
        for (int rt = wave; rt < n_rt; rt += kWavesPerBlock) {
            int zmax;
            {
                const int rowq = rt * 16 + li;
                const int4 rq = rowmeta[rowq < n_rows ? rowq : n_rows - 1];
                zmax = rowq < n_rows ? (rq.w & 0xff) + (rq.w >> 8) : 0;
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) zmax = max(zmax, __shfl_xor(zmax, off, 64));
                zmax = __builtin_amdgcn_readfirstlane(zmax);
            }
            const int n_zt = (zmax + 15) >> 4;
            for (int ztile = 0; ztile < n_zt; ++ztile) {
            const int row = rt * 16 + li;
            const bool rv = row < n_rows;
            const int4 rm = rowmeta[rv ? row : n_rows - 1];
            const RecipRow r{rm.x, rm.y};
            // this lane's four tasks: kz = 16 ztile + lk + 4 i of its row, where the row has them
            const int t0 = rm.z, t1 = rv ? rm.z + (rm.w >> 8) : rm.z;
            const int j0 = rm.w & 0xff;
            const bool last_st = !TILED || st == n_st - 1;
            [[maybe_unused]] double *sums = site_tile_sums + (size_t)blockIdx.x * n_tasks * 4;
            int tt[4], rjv[4];
            double2 Apv[4], Amv[4], wv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int kzz = ztile * 16 + lk + 4 * i;
                const int t = t0 + (kzz - j0);
                tt[i] = (kzz >= j0 && t < t1) ? t : -1;
                const int tc = tt[i] >= 0 ? tt[i] : 0;
                if (last_st) {                                          // (uniform: the last tile of sites finishes the tasks)
                    rjv[i] = trj[tc];
                    Apv[i] = A[2 * tc]; Amv[i] = A[2 * tc + 1];
                    wv[i] = COMMIT ? make_double2(0.0, 0.0) : tw[tc];
                }
            }
            const int aky = r.ky < 0 ? -r.ky : r.ky;
            const double ysign = r.ky < 0 ? -1.0 : 1.0;          // conjugate for -ky (times -1: exact)
            // Branch-free steps, every address inside the tables, nothing undefined into the matrix unit: the site-states are
            // padded with zeros, a row beyond the last reads the last row's (its outputs belong to no task), a kz beyond the
            // table reads the last kz's entry and is cleared by an AND mask (a select would become a branch around the load).
            const int kz_a = ztile * 16 + li, nkz = bx.kmax[2] + 1;
            const unsigned zmask = kz_a < nkz ? 0xffffffffu : 0u;
            auto keep = [&](double v) { return __hiloint2double(__double2hiint(v) & (int)zmask, __double2loint(v) & (int)zmask); };
            const double2 *xp = tab + lk * ktot + r.kx, *yp = tab + lk * ktot + kofs1 + aky, *zp = zt + lk * ktot + min(kz_a, nkz - 1);
            const double *qp = sq + lk;
            double4v d_ac = {0.0, 0.0, 0.0, 0.0}, d_bd = d_ac, d_ad = d_ac, d_bc = d_ac;
            double2 Xn = xp[0], Yn = yp[0], zn = zp[0];
            double qn = qp[0];
            for (int s0 = 0; s0 < ssn; s0 += 4) {
                const double2 X = Xn, z = make_double2(keep(zn.x), keep(zn.y));
                double2 Y = Yn;
                const double q = qn;
                const int sn = s0 + 4 < ssn ? s0 + 4 : s0;         // (the last step re-reads its own operands)
                Xn = xp[sn * ktot]; Yn = yp[sn * ktot]; zn = zp[sn * ktot]; qn = qp[sn];
                Y.y *= ysign;
                double2 xy = cmul(X, Y);
                xy.x *= q; xy.y *= q;
                d_ac = __builtin_amdgcn_mfma_f64_16x16x4f64(z.x, xy.x, d_ac, 0, 0, 0);
                d_bd = __builtin_amdgcn_mfma_f64_16x16x4f64(z.y, xy.y, d_bd, 0, 0, 0);
                d_ad = __builtin_amdgcn_mfma_f64_16x16x4f64(z.y, xy.x, d_ad, 0, 0, 0);
                d_bc = __builtin_amdgcn_mfma_f64_16x16x4f64(z.x, xy.y, d_bc, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (tt[i] < 0) continue;
                const int t = tt[i];
                double sac = d_ac[i], sbd = d_bd[i], sad = d_ad[i], sbc = d_bc[i];
                if constexpr (TILED) {
                    // the sums of the tiles before, in tile order (this lane's own earlier stores)
                    double2 *sm = reinterpret_cast<double2 *>(sums + (size_t)t * 4);
                    if (st > 0) {
                        const double2 p0 = sm[0], p1 = sm[1];
                        sac = p0.x + sac; sbd = p0.y + sbd; sad = p1.x + sad; sbc = p1.y + sbc;
                    }
                    if (!last_st) { sm[0] = make_double2(sac, sbd); sm[1] = make_double2(sad, sbc); continue; }
                }
                const double2 Ap = Apv[i], Am = Amv[i], w = wv[i];
                if (BOTH && !COMMIT) acc0 += fma(w.x, fma(Ap.x, Ap.x, Ap.y * Ap.y), w.y * fma(Am.x, Am.x, Am.y * Am.y));
                const double npx = Ap.x + (sac - sbd), npy = Ap.y + (sad + sbc);
                const double nmx = Am.x + (sac + sbd), nmy = Am.y + (sbc - sad);
                if (COMMIT) {
                    A[2 * t] = (rjv[i] & kTaskHasP) ? make_double2(npx, npy) : make_double2(0.0, 0.0);
                    A[2 * t + 1] = (rjv[i] & kTaskHasM) ? make_double2(nmx, nmy) : make_double2(0.0, 0.0);
                } else {
                    acc += fma(w.x, fma(npx, npx, npy * npy), w.y * fma(nmx, nmx, nmy * nmy));   // ewald_energy.f90:259-266
                }
            }
            }
        }
    }
    }                                                    // site tiles
    if constexpr (!MFMA)
    for (int r0 = 0; r0 < n_rows; r0 += rows_per_tile) {
        const int r1 = min(n_rows, r0 + rows_per_tile);
        // ---- phase 2 for the rows of this tile: XY[row][s] = +-q X[kx] Y[ky]  (recip_rows_phase2's expression)
        for (int idx = tid; idx < (r1 - r0) * nss; idx += kBlock) {
            const int rr = idx / nss, s = idx - rr * nss;
            const RecipRow r = rows[r0 + rr];
            const int aky = r.ky < 0 ? -r.ky : r.ky;
            double2 xy = make_double2(0.0, 0.0);
            if (used) {
                double2 Y = tab[s * ktot + kofs1 + aky];
                if (r.ky < 0) Y.y = -Y.y;
                xy = cmul(tab[s * ktot + r.kx], Y);
                const double q = sq[s];
                xy.x *= q; xy.y *= q;
            }
            xyt[rr * nss + s] = xy;
        }
        __syncthreads();
        // ---- phase 3 for the tasks of those rows (recip_rows_pass's arithmetic per task)
        for (int t = row_first[r0] + tid; t < row_first[r1]; t += kBlock) {
            const int rj = trj[t];
            const double2 Ap = A[2 * t], Am = A[2 * t + 1];
            const double2 w = COMMIT ? make_double2(0.0, 0.0) : tw[t];
            const double2 *xy = xyt + (((rj >> 8) & 0xfffff) - r0) * nss;
            const double2 *z = zt + (rj & 0xff);
            double sac = 0.0, sbd = 0.0, sad = 0.0, sbc = 0.0;
            auto term = [&](const double2 p, const double2 q) {
                sac = fma(p.x, q.x, sac);
                sbd = fma(p.y, q.y, sbd);
                sad = fma(p.x, q.y, sad);
                sbc = fma(p.y, q.x, sbc);
            };
            int s = 0;
            for (; s + 2 <= nss; s += 2) {
                const double2 p0 = xy[s], p1 = xy[s + 1];
                const double2 q0 = z[s * ktot], q1 = z[(s + 1) * ktot];
                term(p0, q0); term(p1, q1);
            }
            for (; s < nss; ++s) term(xy[s], z[s * ktot]);
            if (BOTH && !COMMIT) acc0 += fma(w.x, fma(Ap.x, Ap.x, Ap.y * Ap.y), w.y * fma(Am.x, Am.x, Am.y * Am.y));
            const double npx = Ap.x + (sac - sbd), npy = Ap.y + (sad + sbc);
            const double nmx = Am.x + (sac + sbd), nmy = Am.y + (sbc - sad);
            if (COMMIT) {
                A[2 * t] = (rj & kTaskHasP) ? make_double2(npx, npy) : make_double2(0.0, 0.0);
                A[2 * t + 1] = (rj & kTaskHasM) ? make_double2(nmx, nmy) : make_double2(0.0, 0.0);
            } else {
                acc += fma(w.x, fma(npx, npx, npy * npy), w.y * fma(nmx, nmx, nmy * nmy));   // ewald_energy.f90:259-266
            }
        }
        __syncthreads();                                 // the next tile overwrites XY
    }
    if (!COMMIT) {
        acc = wave_sum(acc);
        if (BOTH) acc0 = wave_sum(acc0);
        if ((tid & 63) == 0) { s_red[2 * (tid >> 6)] = acc; s_red[2 * (tid >> 6) + 1] = acc0; }
        __syncthreads();
        if (tid == 0) {
            double u = 0.0, u0 = 0.0;
            for (int wv = 0; wv < kWavesPerBlock; ++wv) { u += s_red[2 * wv]; u0 += s_red[2 * wv + 1]; }
            u_new[blockIdx.x] = u * kEps0InvEvA / kKbEvK * kTwoPi / bx.volume;   // ewald_energy.f90:272
            if (BOTH) u_old[blockIdx.x] = u0 * kEps0InvEvA / kKbEvK * kTwoPi / bx.volume;
        }
    } else {
        // (every read of the old coordinates happened in phase 1, before the first barrier; a molecule may have more sites
        //  than the workgroup has threads)
        if (it.kind == 0 || it.kind == 1) {
            for (int a = tid; a < n1; a += kBlock) {
                const double *c = cand_row + (size_t)a * 3;
                const int j = atom_slot(tp, it.t, it.m, a);
                px[j] = c[0]; py[j] = c[1]; pz[j] = c[2];
            }
        } else if (it.kind == 2) {
            const int last = it.aux;                      // swap-with-last, delete_molecule.f90:107-114
            if (last != it.m)
                for (int a = tid; a < n1; a += kBlock) {
                    const int j = atom_slot(tp, it.t, it.m, a), jl = atom_slot(tp, it.t, last, a);
                    px[j] = px[jl]; py[j] = py[jl]; pz[j] = pz[jl];
                }
        }
        if (tp.com) {
            const size_t rep3 = (size_t)it.replica * 3;
            double *fcom = tp.com + rep3 * tp.n_mol_slots + tp.mol_off[it.t];
            double *foff = tp.off + rep3 * tp.n_cap_atoms;
            if ((it.kind == 0 || it.kind == 1) && it.frame > 0) {
                const double *fr = cand_row + (size_t)it.frame * 3;
                if (tid < 3) fcom[(size_t)tid * tp.n_mol_slots + it.m] = fr[tid];
                for (int a = tid; a < n1; a += kBlock) {
                    const int j = atom_slot(tp, it.t, it.m, a);
                    for (int d = 0; d < 3; ++d) foff[(size_t)d * tp.n_cap_atoms + j] = fr[(1 + a) * 3 + d];
                }
            } else if (it.kind == 2 && it.aux != it.m) {
                const int last = it.aux;
                if (tid < 3) fcom[(size_t)tid * tp.n_mol_slots + it.m] = fcom[(size_t)tid * tp.n_mol_slots + last];
                for (int a = tid; a < n1; a += kBlock) {
                    const int j = atom_slot(tp, it.t, it.m, a), jl = atom_slot(tp, it.t, last, a);
                    for (int d = 0; d < 3; ++d) foff[(size_t)d * tp.n_cap_atoms + j] = foff[(size_t)d * tp.n_cap_atoms + jl];
                }
            }
        }
        if (tid == 0 && (it.kind == 1 || it.kind == 2)) nmol[it.replica * tp.n_res + it.t] = it.aux;
    }
}

// (Two items per workgroup for short k lists -- each phase's barriers shared, half the workgroups: built and measured in
//  round 5 at the framework box, Nk = 1152: 18.5 -> 19.2 us per 2048 candidates, 28.5 -> 33.3 per 4096; its 117 registers
//  leave four workgroups per CU and a workgroup's life simply doubles.  Not kept; LABNOTES.md.)

// ------------------------------------------------------------------------------------------
// Trial geometry built on the device (the farm's moves: Translation / Rotation / CreateMolecule of the reference,
// src/translation.f90:93-112, src/monte_carlo_utils.f90:30-92, src/create_molecule.f90:166-207) from the molecule
// frames the engine keeps resident (com = primary%mol_com, off = primary%site_offset) and the host's uniform numbers:
//   move 1  translation   com <- ApplyPBC(com + (u[0..2] - 1/2) * translation_step)          offsets unchanged
//   move 2  rotation      offsets rotated by (u[3] - 1/2) * rotation_step about Cartesian axis int(3 u[4]) + 1
//   move 3  creation      com <- lo + L * u[0..2]; offsets of molecule 1 of the type, rotated by 2 pi u[3] about that axis
//   move 4  deletion      nothing to build
// One thread per candidate.  Row c of `rows` (row_stride "sites" of three doubles) receives the candidate's sites
// com + off at [0, n1), its frame at [frame_at] (com) and [frame_at + 1, frame_at + 1 + n1) (offsets): the sweeps read
// the sites, the commit writes sites AND frame back.  Orthorhombic boxes.
// ------------------------------------------------------------------------------------------
// The construction in two pieces, shared with farm_window_kernel (every role of a window rebuilds the candidate it needs
// from the same frames and numbers: the same functions, so the same bits):
//   trial_frame   the candidate's centre and, for a rotation / insertion, the rotation (cos, sin, the two mixed axes);
//   trial_offset  the (rotated) offset of site a; the site itself is frame.com + offset.
struct TrialFrame {
    double com[3];
    double cs, sn;
    int p, q;
    int src_m;                    // the molecule whose frame the candidate starts from (creation: molecule 1 of the type)
    bool rot;
};
template <class TopoT>
__device__ __forceinline__ TrialFrame trial_frame(const TopoT &tp, const BoxDev &bx, int replica, int t, int m, int mv, const double *u,
                                                  double t_step, double r_step) {
    TrialFrame f;
    const int n1 = tp.n1[t];
    const size_t rep3 = (size_t)replica * 3;
    f.src_m = mv == 3 ? 0 : m;                     // creation: the geometry of molecule 1 (create_molecule.f90:197-199)
    for (int d = 0; d < 3; ++d) f.com[d] = tp.com[(rep3 + d) * tp.n_mol_slots + tp.mol_off[t] + f.src_m];
    f.p = 0; f.q = 0;
    f.cs = 1.0; f.sn = 0.0;
    f.rot = false;
    if (mv == 1) {
        for (int d = 0; d < 3; ++d) {
            // translation.f90:104-110, geometry_utils.f90:190: lo + modulo(pos - lo, L)
            double x = (f.com[d] + (u[d] - 0.5) * t_step) - bx.lo[d];
            if (x < 0.0 || x >= bx.L[d]) {
                x = fmod(x, bx.L[d]);
                if (x < 0.0) x += bx.L[d];
            }
            f.com[d] = bx.lo[d] + x;
        }
    } else if (mv == 2 || (mv == 3 && n1 > 1)) {
        const int axis = (int)(u[4] * 3.0) + 1;                              // monte_carlo_utils.f90:54-64
        const double theta = mv == 2 ? (u[3] - 0.5) * r_step : u[3] * kTwoPi;
        sincos(theta, &f.sn, &f.cs);
        f.p = axis % 3;                                                      // RotationMatrix: X -> (Y, Z), Y -> (Z, X), Z -> (X, Y)
        f.q = (axis + 1) % 3;
        f.rot = true;
    }
    if (mv == 3)
        for (int d = 0; d < 3; ++d) f.com[d] = bx.lo[d] + bx.L[d] * u[d];     // create_molecule.f90:180-184
    return f;
}
template <class TopoT>
__device__ __forceinline__ void trial_offset(const TopoT &tp, const TrialFrame &f, int replica, int t, int a, double o[3]) {
    const size_t rep3 = (size_t)replica * 3;
    const int j = atom_slot(tp, t, f.src_m, a);
    for (int d = 0; d < 3; ++d) o[d] = tp.off[(rep3 + d) * tp.n_cap_atoms + j];
    if (f.rot) {                                                             // (p, q) = (1, 2), (2, 0) or (0, 1)
        const int p = f.p, q = f.q;
        const double x = p == 0 ? o[0] : (p == 1 ? o[1] : o[2]);
        const double y = q == 0 ? o[0] : (q == 1 ? o[1] : o[2]);
        const double xn = f.cs * x - f.sn * y, yn = f.sn * x + f.cs * y;
        o[0] = p == 0 ? xn : (q == 0 ? yn : o[0]);
        o[1] = p == 1 ? xn : (q == 1 ? yn : o[1]);
        o[2] = p == 2 ? xn : (q == 2 ? yn : o[2]);
    }
}

static __global__ void trial_build_kernel(Topo tp, BoxDev bx, const RecipItem *__restrict__ items, const int *__restrict__ move,
                                   const double *__restrict__ uu, double t_step, double r_step, double *__restrict__ rows,
                                   int row_stride, int frame_at, int n) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    const RecipItem it = items[c];
    const int mv = move[c];
    if (mv == 4) return;
    const int n1 = tp.n1[it.t];
    const TrialFrame f = trial_frame(tp, bx, it.replica, it.t, it.m, mv, uu + 5 * (size_t)c, t_step, r_step);
    double *row = rows + (size_t)c * row_stride * 3;
    for (int d = 0; d < 3; ++d) row[(size_t)frame_at * 3 + d] = f.com[d];
    for (int a = 0; a < n1; ++a) {
        double o[3];
        trial_offset(tp, f, it.replica, it.t, a, o);
        for (int d = 0; d < 3; ++d) {
            row[(size_t)(frame_at + 1 + a) * 3 + d] = o[d];
            row[(size_t)a * 3 + d] = f.com[d] + o[d];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Full structure factor S(k) (ComputeAllFourierTerms + ComputeRecipAmplitude,
// ewald_phase.f90:340-360, ewald_energy.f90:40-77).
// Step 1: per-atom 1-D phase tables, tab[axis][k][slot]; dead slots are skipped.
// ------------------------------------------------------------------------------------------
static __global__ void phase_table_kernel(Topo tp, BoxDev bx, const double *__restrict__ pos, const int *__restrict__ nmol,
                                   const int *__restrict__ atom_res, const int *__restrict__ atom_mol, int replica,
                                   double2 *__restrict__ tab) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= tp.n_cap_atoms) return;
    const int t = atom_res[j];
    if (atom_mol[j] >= nmol[replica * tp.n_res + t]) return;
    const double *px = pos + (size_t)replica * 3 * tp.n_cap_atoms;
    const double x = px[j], y = px[tp.n_cap_atoms + j], z = px[2 * tp.n_cap_atoms + j];
    int row = 0;
    for (int axis = 0; axis < 3; ++axis) {
        const double th = atom_phase(bx, axis, x, y, z);
        for (int k = 0; k <= bx.kmax[axis]; ++k, ++row) tab[(size_t)row * tp.n_cap_atoms + j] = phase_entry(th, k);
    }
}

// Step 2: one workgroup per k-vector sums q_j X_j(kx) Y_j(ky) Z_j(kz) over the live atoms.
static __global__ __launch_bounds__(kBlock) void sfactor_kernel(Topo tp, BoxDev bx, const int *__restrict__ nmol,
                                                         const int *__restrict__ atom_res,
                                                         const int *__restrict__ atom_mol,
                                                         const double *__restrict__ atom_q,
                                                         const int *__restrict__ kpack, const int *__restrict__ kslot,
                                                         int replica, const double2 *__restrict__ tab,
                                                         double2 *__restrict__ S) {
    __shared__ double s_red[2 * kWavesPerBlock];
    const int k = blockIdx.x;
    const int kp = kpack[k];
    const int kx = kp & 0xff, ky = ((kp >> 8) & 0xff) - 128, kz = ((kp >> 16) & 0xff) - 128;
    const int aky = ky < 0 ? -ky : ky, akz = kz < 0 ? -kz : kz;
    const size_t nc = tp.n_cap_atoms;
    const double2 *tx = tab + (size_t)kx * nc;
    const double2 *ty = tab + (size_t)(bx.kmax[0] + 1 + aky) * nc;
    const double2 *tz = tab + (size_t)(bx.kmax[0] + bx.kmax[1] + 2 + akz) * nc;
    double re = 0.0, im = 0.0;
    for (int j = threadIdx.x; j < tp.n_cap_atoms; j += kBlock) {
        if (atom_mol[j] >= nmol[replica * tp.n_res + atom_res[j]]) continue;
        double2 Y = ty[j], Z = tz[j];
        if (ky < 0) Y.y = -Y.y;
        if (kz < 0) Z.y = -Z.y;
        const double2 p = cmul(cmul(tx[j], Y), Z);
        const double q = atom_q[j];
        re += q * p.x;
        im += q * p.y;
    }
    re = wave_sum(re);
    im = wave_sum(im);
    if ((threadIdx.x & 63) == 0) { s_red[2 * (threadIdx.x >> 6)] = re; s_red[2 * (threadIdx.x >> 6) + 1] = im; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < kWavesPerBlock; ++w) { a += s_red[2 * w]; b += s_red[2 * w + 1]; }
        S[kslot[k]] = make_double2(a, b);            // task-ordered slot of this k
    }
}

// ------------------------------------------------------------------------------------------
// ComputeIntraResidueRealCoulombEnergySingleMol (ewald_energy.f90:371-411): one thread per item,
// pairs visited in the reference's order.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double intra_energy(const Topo &tp, const BoxDev &bx, const double *__restrict__ pos, const double *__restrict__ res_q,
                                               const PairItem &it, const double *__restrict__ cand_sites, int site_stride) {
    const int n1 = tp.n1[it.t];
    const double *px = pos + (size_t)it.replica * 3 * tp.n_cap_atoms;
    const double *py = px + tp.n_cap_atoms, *pz = py + tp.n_cap_atoms;
    auto site = [&](int a, double &x, double &y, double &z) {
        if (it.src < 0) {
            const int j = atom_slot(tp, it.t, it.m, a);
            x = px[j]; y = py[j]; z = pz[j];
        } else {
            const double *c = cand_sites + ((size_t)it.src * site_stride + a) * 3;
            x = c[0]; y = c[1]; z = c[2];
        }
    };
    double u = 0.0;
    for (int a1 = 0; a1 < n1 - 1; ++a1) {
        const double q1 = res_q[it.t * tp.max_atom + a1];
        double x1, y1, z1;
        site(a1, x1, y1, z1);
        for (int a2 = a1 + 1; a2 < n1; ++a2) {
            const double q2 = res_q[it.t * tp.max_atom + a2];
            double x2, y2, z2;
            site(a2, x2, y2, z2);
            const double r = sqrt(bx.triclinic ? image_r2<true>(x2 - x1, y2 - y1, z2 - z1, bx)
                                               : image_r2<false>(x2 - x1, y2 - y1, z2 - z1, bx));
            if (r > kErrorTol) u = u + q1 * q2 * (erfc(bx.alpha * r) - 1.0) / r;
        }
    }
    return u * kEps0InvEvA / kKbEvK;
}

// Molecules of up to kIntraThreadMax sites take one thread each (the loop above: the reference's order); larger ones one
// WAVE each (intra_wave_kernel).  Which form an item takes depends on its own size only, so its bits never depend on the
// launch it is part of.
constexpr int kIntraThreadMax = 32;
static __global__ void intra_kernel(Topo tp, BoxDev bx, const double *__restrict__ pos, const double *__restrict__ res_q,
                             const PairItem *__restrict__ items, int n_items, const double *__restrict__ cand_sites,
                             int site_stride, double *__restrict__ u_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) return;
    if (tp.n1[items[i].t] > kIntraThreadMax) return;         // intra_wave_kernel's
    u_out[i] = intra_energy(tp, bx, pos, res_q, items[i], cand_sites, site_stride);
}

// The same sum for a large molecule (n1 > kIntraThreadMax: n1 (n1 - 1) / 2 erfc terms, 45 000 at 300 sites) by one WAVE per
// item: the sites (x, y, z, q) staged in LDS in tiles, site a1 wave-uniform, the lanes taking a2 = a1 + 1 + lane, + 64, ...;
// every lane adds its terms in that (a1, a2) order and the lanes are added by the wave butterfly: a fixed order, the same
// bits run to run (the one-thread loop's order it is not: the two differ by rounding, ~1e-13 relative).
constexpr int kIntraTile = 512;                  // sites per LDS tile (16 KB)
static __global__ __launch_bounds__(64) void intra_wave_kernel(Topo tp, BoxDev bx, const double *__restrict__ pos, const double *__restrict__ res_q,
                                                        const PairItem *__restrict__ items, int n_items, const double *__restrict__ cand_sites,
                                                        int site_stride, double *__restrict__ u_out) {
    __shared__ double4 s_a[kIntraTile], s_b[kIntraTile];
    const int i = blockIdx.x;
    if (i >= n_items) return;
    const PairItem it = items[i];
    const int n1 = tp.n1[it.t];
    if (n1 <= kIntraThreadMax) return;                        // intra_kernel's
    const int lane = threadIdx.x;
    const double *px = pos + (size_t)it.replica * 3 * tp.n_cap_atoms;
    const double *py = px + tp.n_cap_atoms, *pz = py + tp.n_cap_atoms;
    auto site = [&](int a) {
        double x, y, z;
        if (it.src < 0) {
            const int j = atom_slot(tp, it.t, it.m, a);
            x = px[j]; y = py[j]; z = pz[j];
        } else {
            const double *c = cand_sites + ((size_t)it.src * site_stride + a) * 3;
            x = c[0]; y = c[1]; z = c[2];
        }
        return make_double4(x, y, z, res_q[it.t * tp.max_atom + a]);
    };
    double u = 0.0;
    // tiles (A, B) with B >= A: a1 runs over tile A, a2 over tile B
    for (int a0 = 0; a0 < n1; a0 += kIntraTile) {
        const int na = min(kIntraTile, n1 - a0);
        __syncthreads();
        for (int a = lane; a < na; a += 64) s_a[a] = site(a0 + a);
        for (int b0 = a0; b0 < n1; b0 += kIntraTile) {
            const int nb = min(kIntraTile, n1 - b0);
            __syncthreads();
            for (int b = lane; b < nb; b += 64) s_b[b] = site(b0 + b);
            __syncthreads();
            for (int a = 0; a < na; ++a) {
                const double4 p1 = s_a[a];
                const int first = (b0 == a0) ? a + 1 : 0;
                for (int b = first + lane; b < nb; b += 64) {
                    const double4 p2 = s_b[b];
                    const double r = sqrt(bx.triclinic ? image_r2<true>(p2.x - p1.x, p2.y - p1.y, p2.z - p1.z, bx)
                                                       : image_r2<false>(p2.x - p1.x, p2.y - p1.y, p2.z - p1.z, bx));
                    if (r > kErrorTol) u = u + p1.w * p2.w * (erfc(bx.alpha * r) - 1.0) / r;
                }
            }
        }
    }
    u = wave_sum(u);
    if (lane == 0) u_out[i] = u * kEps0InvEvA / kKbEvK;
}

}  // namespace mgpu

#endif
