// HIP kernels of the GCMC energy engine, written for gfx950 (CDNA4, wave64) only.
//
// Data layout in HBM (per engine; R replicas):
//   pos      [R][3][Ncap]   fp64   x / y / z planes of every replica's atom slots
//   nmol     [R][n_res]     int32  live molecule count per residue type
//   A        [R][n_slots]   fp64x2 ewald%recip_amplitude of every replica, in TASK order: slot 2t holds
//                                  A(kx, ky, +j) and slot 2t + 1 holds A(kx, ky, -j) of row-form task t (zero where
//                                  the list has no such k), so a k sweep reads 32 contiguous bytes per task with no
//                                  index load in front of them; kslot[k] maps the reference's k order to slots
//   kpack    [Nk]           int32  kx | (ky+128)<<8 | (kz+128)<<16
//   kw       [Nk]           fp64   form_factor(k) * W(k)
//   pair_tab [nt][nt]       fp64x2 {4 epsilon, sigma^2} per atom-type pair
//   res_q / res_atype [n_res][max_atom]   site templates (charge, 0-based atom type)
// Atom slot index inside a replica, residue type t with n1 sites and `cap` molecule slots:
//   plane-major (many small molecules):  seg_off[t] + a * cap + m   -> a wave sweeps 64 molecules of
//                                        ONE site index, so charge / atom type / LJ pair are
//                                        wave-uniform and whole LJ or Coulomb halves are skipped
//                                        by a scalar branch (SPC/E: LJ for 1 of 9 site pairs);
//   site-major  (n1 >= 64):              seg_off[t] + m * n1 + a     -> a wave sweeps 64 sites of
//                                        one molecule, per-lane charge / type;
//   frozen      (inactive, n1 >= 64:     the same slots, with the residue's sites renumbered by atom type (the engine
//                frameworks)             translates at the API boundary): a wave sweeps 64 sites of ONE atom type, so
//                                        the LJ pair is wave-uniform again and only the charge is per lane.
// All arithmetic is IEEE fp64.  Reductions use a fixed tree (wave butterfly -> LDS -> ordered
// sum over waves -> ordered sum over splits): results are bitwise reproducible run to run.
#ifndef MGPU_KERNELS_H
#define MGPU_KERNELS_H

#include <hip/hip_runtime.h>

#include <type_traits>

#include "mgpu_internal.h"

namespace mgpu {

constexpr int kMaxRes = 8;        // residue types per engine
constexpr int kBlock = 256;       // threads per workgroup = 4 waves, one per SIMD
constexpr int kWavesPerBlock = kBlock / 64;
constexpr int kSiteChunk = 32;    // candidate sites staged in LDS per pass (generic path)
#ifndef MGPU_PAIR_BLOCK
#define MGPU_PAIR_BLOCK 512
#endif
#ifndef MGPU_PAIR_MINWAVES
#define MGPU_PAIR_MINWAVES 4   // <= 128 VGPRs: two 8-wave workgroups per CU (measured best, tools/bench_kernels.py)
#endif
constexpr int kPairBlock = MGPU_PAIR_BLOCK;   // pair sweep: persistent waves share one LDS Coulomb table
constexpr int kPairWaves = kPairBlock / 64;
constexpr int kMaxTypes = 16;     // atom types (LDS pair table 16 x 16 x 16 B = 4 KiB)
constexpr int kMaxGrp = 32;       // atom-type groups of all frozen residues of a topology together
constexpr int kFlatMaxPlanes = 64;  // planes of a replica pair_flat_kernel handles (one lane builds one plane's record)
constexpr int kMaxFusedSites = 3; // trial moves of molecules up to this size sweep old + new together (2 NS register sites) at 4 waves per SIMD
constexpr int kMaxFusedSitesWide = 5;   // largest molecule of the register-site sweeps (larger ones: the LDS-staged NS = 0 sweep)

struct Topo {
    int n_res;
    int n_types;
    int n_cap_atoms;              // atom slots per replica
    int max_atom;
    int n1[kMaxRes];              // nb%atom_in_residue
    int cap[kMaxRes];             // molecule slots
    int seg_off[kMaxRes];         // first atom slot of the residue type
    int site_major[kMaxRes];      // 0 plane-major, 1 site-major, 2 frozen: site-major with the sites sorted by atom type
    // frozen residues (inactive, n1 >= 64: frameworks; swept by pair_flat_kernel): the residue's sites are renumbered
    // so that sites of one atom type are contiguous; group g of residue t (g < n_grp[t]) is record grp_off[t] + g of
    // grp_start / grp_cnt / grp_ty = {first sorted site, count, 0-based atom type}.  A wave then sweeps 64 consecutive
    // sites of ONE atom type: the (4 epsilon, sigma^2) of every (candidate site, unit) pair is wave-uniform and the LJ
    // half is skipped by a scalar branch where epsilon = 0; only the charge is per lane (per-atom framework charges).
    int n_grp[kMaxRes];
    int grp_off[kMaxRes];
    int grp_start[kMaxGrp], grp_cnt[kMaxGrp], grp_ty[kMaxGrp];
    const double *slot_q;         // [n_cap_atoms] charge of every atom slot (same for all replicas)
    const int *slot_ty;           // [n_cap_atoms] 0-based atom type of every atom slot
    // Molecule frames (only once mgpu_replica_set_frames has been used, else null): what the reference keeps per
    // molecule -- com [R][3][n_mol_slots] = primary%mol_com (molecule slot mol_off[t] + m) and off [R][3][n_cap_atoms] =
    // primary%site_offset in the slot layout of pos -- so that trial moves can be built on the device
    // (trial_build_kernel); pos stays the rounded sum com + off, which is what the reference forms before every use.
    double *com;
    double *off;
    int mol_off[kMaxRes];
    int n_mol_slots;
};

struct BoxDev {
    double L[3], invL[3];         // orthorhombic edge lengths box%matrix(d,d)
    double ctr[3];                // centre of the primary cell (orthorhombic): bounds_lo + L / 2
    double lo[3];                 // bounds_lo
    double rcp[9];                // box%reciprocal, row-major
    double m[9];                  // box%matrix, row-major (cell vectors are its columns)
    int triclinic;                // box%type == 3: 27-image search instead of the per-axis fold
    int tri_lower;                // ... and box%matrix is lower triangular (m[1] = m[2] = m[5] = 0: every triclinic box the
                                  // reference's reader builds, readers_utils.f90:242-245): the search is done exactly in 4 + 4
                                  // evaluations instead of 27 (image_r2_tri_lower)
    double rc2;                   // real_space_cutoff^2
    double alpha;
    int coul_idx_base;            // Coulomb table: row = (hi32(r^2) >> 14) - coul_idx_base
    int coul_last_row;            // index of the all-zero clamp row (= number of real rows)
    double volume;
    int kmax[3];
    int nk;
    int n_slots;                  // complex entries of A(k) per replica (2 per row-form task; = nk without row form)
};

struct PairItem {
    int replica, t, m;            // m < 0: nothing excluded
    int src;                      // < 0: sites are the resident slot m; else row of cand_sites
    int ordered;                  // 1: SingleMolPairwiseEnergy semantics (energy_utils.f90:153-158)
};

struct RecipItem {
    int replica, t, m, kind;
    int src;                      // row of cand_sites holding the new sites (< 0: none)
    int aux;                      // commit: new molecule count of (replica, t) after the move
    int frame;                    // > 0: row `src` carries the candidate's frame at site index `frame` (com) and frame + 1 ...
                                  // (offsets): a device-built trial (trial_build_kernel); the commit writes it back
};

__device__ __forceinline__ int atom_slot(const Topo &tp, int t, int m, int a) {
    return tp.site_major[t] ? tp.seg_off[t] + m * tp.n1[t] + a : tp.seg_off[t] + a * tp.cap[t] + m;
}

// a pair-sweep partial {e_lj, e_coul}; SC1: agent-scope write-through stores (see pair_sweep_item)
template <bool SC1>
__device__ __forceinline__ void store_partial(double2 *p, double a, double b) {
    if constexpr (SC1) {
        __hip_atomic_store(&p->x, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&p->y, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        *p = make_double2(a, b);
    }
}
__device__ __forceinline__ double load_sc1(const double *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Minimum-image separation for cubic / orthorhombic boxes.  The reference folds with
// modulo(d + L/2, L) - L/2 (geometry_utils.f90:388-391); d - L*rint(d/L) returns the same
// image (|d'| <= L/2) and differs only in the last bits of d'.
__device__ __forceinline__ double min_image(double d, double L, double invL) {
    return fma(-L, rint(d * invL), d);
}

// Squared minimum-image distance for a raw separation (dx, dy, dz): ComputeDistance
// (geometry_utils.f90:359-415).  Cubic / orthorhombic: per-axis fold.  Triclinic: the minimum over
// the 27 neighbouring images delta + sx a + sy b + sz c, exactly the reference's search.
// The 27-image search of ComputeDistance (geometry_utils.f90:397-411) for a LOWER-TRIANGULAR box%matrix -- rows
// (lx, 0, 0), (xy, ly, 0), (xz, yz, lz): what the reference's reader stores for every triclinic cell -- where the image
// (sx, sy, sz) of a raw separation is
//     tx = dx + sx lx,   ty = (dy + sx xy) + sy ly,   tz = ((dz + sx xz) + sy yz) + sz lz
// (the reference's sums, term by term: the products with 0 and +-1 are exact).  The minimum of r2 = tz^2 + (ty^2 + tx^2)
// over the 27 is found EXACTLY in eight evaluations:
//   * sz: for fixed (sx, sy) r2 grows with |tz| (fma is monotone), so the best sz is the one of zb - lz, zb, zb + lz of
//     smallest magnitude: a fold, no evaluation;
//   * sy: zb is never the worst of the three, so the two best are sy = 0 and the nearer of sy = -1 / +1; the remaining
//     one cannot give less than tx^2 + ty_far^2;
//   * sx: likewise sx = 0 and the nearer of -1 / +1; the remaining one cannot give less than tx_far^2.
// If the best of the eight does not exceed the smallest of those lower bounds it IS the minimum over the 27 (the same
// double: the same expressions, and a minimum does not care how many larger values it is taken over); otherwise --
// cells much longer than wide, where a minimum-image distance can exceed a cell width -- the full search runs.
__device__ __forceinline__ bool image_r2_tri_lower(double dx, double dy, double dz, const BoxDev &bx, double &out) {
    const double lx = bx.m[0], xy = bx.m[3], ly = bx.m[4], xz = bx.m[6], yz = bx.m[7], lz = bx.m[8];
    double best = 1.7976931348623157e308;
    const double xm = dx - lx, xp = dx + lx;
    const bool xneg = fabs(xm) <= fabs(xp);
    const double x_out = xneg ? xm : xp, x_far = xneg ? xp : xm;
    double bound = x_far * x_far;
    auto with_sx = [&](double tx, double yb, double zb1) {
        const double x2 = tx * tx;
        const double ym = yb - ly, yp = yb + ly;
        const bool yneg = fabs(ym) <= fabs(yp);
        const double y_out = yneg ? ym : yp, y_far = yneg ? yp : ym;
        bound = fmin(bound, fma(y_far, y_far, x2));
        auto with_sy = [&](double ty, double zb) {
            const double zm = zb - lz, zp = zb + lz;
            const double z_out = fabs(zm) <= fabs(zp) ? zm : zp;
            const double tz = fabs(z_out) < fabs(zb) ? z_out : zb;
            best = fmin(best, fma(tz, tz, fma(ty, ty, x2)));
        };
        with_sy(yb, zb1);                                             // sy = 0: + 0 * yz is exact
        with_sy(y_out, yneg ? zb1 - yz : zb1 + yz);                   // sy = -1 / +1
    };
    with_sx(dx, dy, dz);                                              // sx = 0
    with_sx(x_out, xneg ? dy - xy : dy + xy, xneg ? dz - xz : dz + xz);
    out = best;
    return best <= bound;
}

// the full search
__device__ __forceinline__ double image_r2_search27(double dx, double dy, double dz, const double *__restrict__ m) {
    double best = 1.7976931348623157e308;     // huge(1.0_real64), geometry_utils.f90:399
    for (int sx = -1; sx <= 1; ++sx)
        for (int sy = -1; sy <= 1; ++sy)
            for (int sz = -1; sz <= 1; ++sz) {
                const double tx = dx + sx * m[0] + sy * m[1] + sz * m[2];
                const double ty = dy + sx * m[3] + sy * m[4] + sz * m[5];
                const double tz = dz + sx * m[6] + sy * m[7] + sz * m[8];
                const double t2 = fma(tz, tz, fma(ty, ty, tx * tx));
                best = t2 < best ? t2 : best;
            }
    return best;
}

template <bool TRI>
__device__ __forceinline__ double image_r2(double dx, double dy, double dz, const BoxDev &bx) {
    if constexpr (TRI) {
        if (bx.tri_lower) {                                           // uniform
            double r2;
            if (!image_r2_tri_lower(dx, dy, dz, bx, r2)) r2 = image_r2_search27(dx, dy, dz, bx.m);
            return r2;
        }
    }
    if constexpr (!TRI) {
        dx = min_image(dx, bx.L[0], bx.invL[0]);
        dy = min_image(dy, bx.L[1], bx.invL[1]);
        dz = min_image(dz, bx.L[2], bx.invL[2]);
        return fma(dz, dz, fma(dy, dy, dx * dx));
    } else {
        double best = 1.7976931348623157e308;     // huge(1.0_real64), geometry_utils.f90:399
        for (int sx = -1; sx <= 1; ++sx)
            for (int sy = -1; sy <= 1; ++sy)
                for (int sz = -1; sz <= 1; ++sz) {
                    const double tx = dx + sx * bx.m[0] + sy * bx.m[1] + sz * bx.m[2];
                    const double ty = dy + sx * bx.m[3] + sy * bx.m[4] + sz * bx.m[5];
                    const double tz = dz + sx * bx.m[6] + sy * bx.m[7] + sz * bx.m[8];
                    const double t2 = fma(tz, tz, fma(ty, ty, tx * tx));
                    best = t2 < best ? t2 : best;
                }
        return best;
    }
}

// Minimum-image r^2 for a raw separation known to satisfy |d| < 1.5 L on every axis (orthorhombic): the folded
// magnitude is min(|d|, L - |d|) -- two instructions per axis (the negation / absolute value are operand modifiers)
// instead of multiply, round, fused multiply-add.  Same value as min_image() applied to the SAME raw separation d:
// for |d| <= L/2 it is |d| itself, beyond it is L - |d|, one rounding of the same real number as fma(-L, +-1, d);
// L - |d| < 0 (L < |d| < 1.5 L) squares to the right thing.  (Only a separation within one rounding of exactly L/2 can
// come out as the other of two equally near images: |d'| differs by an ulp of L there.)  The engine launches the
// kernels built with it only when every resident atom of the replicas involved AND every candidate site of the launch
// lies within kFastFoldRange box lengths of the cell centre on every axis (it tracks that on the host), so that any
// two of them are less than 1.5 L apart; nothing is refolded, so both kernel families see the same d.
__device__ __forceinline__ double image_r2_fast(double dx, double dy, double dz, const BoxDev &bx) {
    const double ax = fabs(dx), ay = fabs(dy), az = fabs(dz);
    const double mx = fmin(ax, bx.L[0] - ax), my = fmin(ay, bx.L[1] - ay), mz = fmin(az, bx.L[2] - az);
    return fma(mz, mz, fma(my, my, mx * mx));
}

// 1/sqrt(x): v_rsq_f64 (2^-24 relative) + one Newton step with its second-order term; measured
// max relative error 1.4e-16 on gfx950 (tools/probe_math.hip), the same as ocml's rsqrt.
__device__ __forceinline__ double fast_rsqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-(x * y), y, 1.0);
    return fma(y * e, fma(0.375, e, 0.5), y);
}

// 1/x: v_rcp_f64 + one Newton step with its second-order term (used only for LJ pairs)
__device__ __forceinline__ double fast_rcp(double x) {
    const double y = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, y, 1.0);
    return fma(y, fma(e, e, e), y);
}

// G(s) = erfc(alpha sqrt(s)) / sqrt(s), s = r^2, from the LDS-resident Coulomb table
// (build_coulomb_table, mgpu_host_setup.cpp): the row is selected by the binary exponent and the top
// 6 mantissa bits of s, the local coordinate t = s - (s with the remaining mantissa bits cleared), the value a
// degree-6 polynomial (5 fp64 + 2 fp32 coefficients = 48 bytes = three ds_read_b128).  No sqrt, rsqrt,
// erfc, exp or division.
// Index path, two instructions (round 4; it was four): sh = the high word of s shifted down to (exponent | 6 mantissa bits),
// address = sh * 48 + tab_adj in ONE v_mad_u32_u24, where tab_adj = table - idx_base * 48 (coul_tab_adjusted) carries the
// subtraction of the table's first index.  There is NO clamp: a minimum-image r^2 cannot lie beyond the table (it is built
// up to the box's largest minimum-image distance), and an s BELOW the table (r < 0.5 A: sh < idx_base) makes the address
// fall outside the table -- just below it into the workgroup's other LDS arrays, or wrapped far outside the allocation.
// Such a read raises nothing (tools/probe_lds_oob.hip, measured on MI355X) but its VALUE IS UNDEFINED (stale LDS bytes
// near the allocation, zeros far from it): every caller MUST look at `sh_out` and replace those lanes by the slow path --
// it keeps the smallest sh of a unit (`sh_min`: v_min3_u32, one instruction per two or three terms) and compares once.
// (`sh_out` is a reference parameter, not a return value to ignore: a call site that drops it does not compile.)
__device__ __forceinline__ const char *coul_tab_adjusted(const char *tab, int idx_base) {
    return tab - (size_t)idx_base * sizeof(CoulRow);
}
__device__ __forceinline__ double coul_lds(double s, const char *__restrict__ tab_adj, unsigned &sh_out) {
    const int hi = __double2hiint(s);
    const unsigned sh = (unsigned)hi >> (20 - kCoulM);
    sh_out = sh;
    constexpr int kMant = (1 << (20 - kCoulM)) - 1;
    const double s0 = __hiloint2double(hi & ~kMant, 0);                  // the row's first s: low mantissa bits cleared
    const double t = s - s0;                                             // exact; rows are expanded in it
    const double2 *r = reinterpret_cast<const double2 *>(tab_adj + __umul24(sh, (unsigned)sizeof(CoulRow)));
    const double2 c01 = r[0], c23 = r[1], c4f = r[2];
    const double c5 = (double)__int_as_float(__double2loint(c4f.y));
    double p = fma((double)__int_as_float(__double2hiint(c4f.y)), t, c5);
    p = fma(p, t, c4f.x);
    p = fma(p, t, c23.y);
    p = fma(p, t, c23.x);
    p = fma(p, t, c01.y);
    p = fma(p, t, c01.x);
    return p;
}

// r < 0.5 A (never reached by a physical configuration): direct evaluation.  GUARD: CoulombEnergy's
// r < 1e-10 -> 0 (energy_utils.f90:244), which only the ordered static sweep applies.
__device__ __attribute__((noinline)) double coul_slow(double s, double alpha, bool guard) {
    const double r = sqrt(s);
    if (guard && r < kErrorTol) return 0.0;
    return erfc(alpha * r) / r;
}

// One site-atom pair: Lennard-Jones inside the cutoff (energy_utils.f90:417-424) and
// erfc(alpha r)/r for every distance (energy_utils.f90:427-432).  Generic (per-lane flags) form used
// by the site-major and NS = 0 sweeps; the register-site hot path inlines the same arithmetic.
template <bool GUARD_R0, bool TRI, bool FASTW = false>
__device__ __forceinline__ void pair_term(double dx, double dy, double dz, const BoxDev &bx, double qq,
                                          double eps4, double sig2, bool do_lj, bool do_c,
                                          const char *__restrict__ coul_tab, double &elj, double &ec) {
    const double r2 = FASTW ? image_r2_fast(dx, dy, dz, bx) : image_r2<TRI>(dx, dy, dz, bx);
    if (do_lj) {
        const double s2 = sig2 * fast_rcp(r2);
        const double s6 = s2 * s2 * s2;
        const double e = eps4 * fma(s6, s6, -s6);
        elj += (r2 < bx.rc2) ? e : 0.0;
    }
    if (do_c) {
        unsigned sh;
        double g = coul_lds(r2, coul_tab_adjusted(coul_tab, bx.coul_idx_base), sh);
        // (this generic path also serves triclinic boxes, where a site more than a cell outside the box can give an r^2
        //  beyond the table: such a lookup is discarded -- the all-zero last row's value -- as the clamped index gave it)
        if (sh >= (unsigned)(bx.coul_idx_base + bx.coul_last_row)) g = 0.0;
        if (sh < (unsigned)bx.coul_idx_base) g = coul_slow(r2, bx.alpha, GUARD_R0);
        ec += qq * g;
    }
}

// ------------------------------------------------------------------------------------------
// Pair sweep: ComputePairInteractionEnergy_singlemol (energy_utils.f90:374-442) and, with
// item.ordered, SingleMolPairwiseEnergy (energy_utils.f90:121-187) for a batch of items.
//
// Work unit = (item, split): ONE WAVE sweeps every nsplit-th unit of 64 atoms of the item's
// replica and writes one partial {e_lj, e_coul}.  Waves are persistent: the grid is sized to the
// chip and each wave strides over the n_items * nsplit work units, so the ~30 KiB Coulomb table is
// staged into LDS once per workgroup and no workgroup barrier sits inside the sweep.
// NS > 0: every item has exactly NS sites, held in registers; NS = 0: any count, staged through a
// per-wave LDS slab in chunks of kSiteChunk.  ORDERED selects SingleMolPairwiseEnergy semantics
// (only molecules after the item's, plus CoulombEnergy's r < 1e-10 guard) for the static total.
// TRI selects the triclinic 27-image distance (generic NS = 0 path only).
// FUSED (NS > 0): the item is a trial MOVE of the resident molecule (replica, t, m) to the candidate row
// `src`: the OLD state (resident sites) and the NEW state (candidate sites) are swept together, 2 NS
// register sites against every atom -- one set of coordinate loads, masks and loop bookkeeping for both
// ComputeOldEnergy and ComputeNewEnergy (monte_carlo_utils.f90:380-395 / :275-330); each state's sums are
// formed exactly as the unfused sweep forms them, and the work unit writes two partials {old, new}.
// ------------------------------------------------------------------------------------------
// One work unit (item, split) of the pair sweep, executed by ONE WAVE: every nsplit-th unit of the item's replica, one
// partial {e_lj, e_coul} per state into partials[w * NST ...].  cand_sites / site_stride: the candidate rows (row it.src);
// s_coul / s_pair: the workgroup's LDS copies of the Coulomb table and the pair table; w_site / w_sty: this wave's LDS slab
// (NS = 0 only).  Shared by pair_sweep_kernel and chain_window_kernel.
// SC1OUT: the partials are stored with agent-scope (`sc1`, write-through) stores, for a consumer in ANOTHER workgroup of the
// same launch that reads them with `sc1` loads (chain_window_kernel's ticket hand-off).
template <int NS, bool ORDERED, bool TRI, bool FUSED, bool FASTW, bool SC1OUT = false>
__device__ __forceinline__ void pair_sweep_item(
    const Topo &tp, const BoxDev &bx, const double *__restrict__ pos, const int *__restrict__ nmol,
    const double *__restrict__ res_q, const int *__restrict__ res_atype, const double2 *__restrict__ pair_tab,
    const char *__restrict__ s_coul, const double2 *__restrict__ s_pair, double *__restrict__ w_site, int *__restrict__ w_sty,
    const PairItem it, const double *__restrict__ cand_sites, int site_stride, int split, int nsplit, int lane,
    double2 *__restrict__ partials, int w) {
    static_assert(!FUSED || (NS > 0 && !ORDERED && !TRI), "the fused old + new sweep is a register-site, unordered, orthorhombic path");
    static_assert(!FASTW || (NS > 0 && !ORDERED && !TRI), "the two-instruction fold is a register-site, unordered, orthorhombic path");
    constexpr int NTY = NS > 0 ? NS : 1;                  // sites of the molecule (charge / type per site)
    constexpr int NST = FUSED ? 2 : 1;                    // states swept together (old, new)
    constexpr int NREG = NTY * NST;                       // register-resident sites: state-major, [state][site]
    const int nt = tp.n_types;
    [[maybe_unused]] const char *coul_adj = coul_tab_adjusted(s_coul, bx.coul_idx_base);
    {
        const double *px = pos + (size_t)it.replica * 3 * tp.n_cap_atoms;
        const double *py = px + tp.n_cap_atoms, *pz = py + tp.n_cap_atoms;
        const int *nm_r = nmol + it.replica * tp.n_res;
        const int n1 = NS > 0 ? NS : tp.n1[it.t];
        constexpr bool ordered = ORDERED;   // the host puts only one kind of item in a launch

        auto load_site = [&](int a, double &x, double &y, double &z) {
            if (it.src < 0) {
                const int j = atom_slot(tp, it.t, it.m, a);
                x = px[j]; y = py[j]; z = pz[j];
            } else {
                const double *c = cand_sites + ((size_t)it.src * site_stride + a) * 3;
                x = c[0]; y = c[1]; z = c[2];
            }
        };
        double rx[NREG], ry[NREG], rz[NREG], rq[NTY];
        int rty[NTY];
        if constexpr (NS > 0) {
#pragma unroll
            for (int a = 0; a < NTY; ++a) {
                if constexpr (FUSED) {
                    // old state: the resident slot; new state: the candidate row
                    const int j = atom_slot(tp, it.t, it.m, a);
                    rx[a] = px[j]; ry[a] = py[j]; rz[a] = pz[j];
                    const double *c = cand_sites + ((size_t)it.src * site_stride + a) * 3;
                    rx[NTY + a] = c[0]; ry[NTY + a] = c[1]; rz[NTY + a] = c[2];
                    asm volatile("" : "+v"(rx[NTY + a]), "+v"(ry[NTY + a]), "+v"(rz[NTY + a]));
                } else {
                    load_site(a, rx[a], ry[a], rz[a]);
                }
                rq[a] = res_q[it.t * tp.max_atom + a];
                rty[a] = res_atype[it.t * tp.max_atom + a];
                // wave-uniform values, but parked in VGPRs: the sweep already needs ~100 SGPRs for box,
                // pointers and per-plane parameters, and spilled SGPRs cost v_readlane in the hot loop
                asm volatile("" : "+v"(rx[a]), "+v"(ry[a]), "+v"(rz[a]));
            }
        }

        double elj[NST], ec[NST];
#pragma unroll
        for (int st = 0; st < NST; ++st) { elj[st] = 0.0; ec[st] = 0.0; }
        for (int sb = 0; sb < n1; sb += kSiteChunk) {
            const int ns = NS > 0 ? NS : min(kSiteChunk, n1 - sb);
            if constexpr (NS == 0) {
                // stage this chunk of sites in the wave's own LDS slab (no workgroup barrier needed)
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (lane < ns) {
                    double x, y, z;
                    load_site(sb + lane, x, y, z);
                    w_site[lane * 4 + 0] = x; w_site[lane * 4 + 1] = y; w_site[lane * 4 + 2] = z;
                    w_site[lane * 4 + 3] = res_q[it.t * tp.max_atom + sb + lane];
                    w_sty[lane] = res_atype[it.t * tp.max_atom + sb + lane];
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            for (int t2 = 0, plane_base = 0; t2 < tp.n_res; plane_base += tp.n1[t2], ++t2) {
                if (ordered && t2 < it.t) continue;
                const int nm = nm_r[t2], n2 = tp.n1[t2];
                if (nm == 0) continue;
                const bool same_t = (t2 == it.t) && (it.m >= 0);
                // ---- hot path of the register-site sweeps: one plane = site a2 of every molecule of a plane-major type
                //      (charge and atom type uniform), swept in units of 64 molecules, branch-free per unit, NS independent
                //      dependency chains, next unit's coordinates prefetched while this one computes ----
                [[maybe_unused]] auto plane_sweep = [&](const double *pxp, const double *pyp, const double *pzp,
                                                        int nm, bool excl, int dummy_m, double qj, int tyj, int key) {
                    const int cpp = (nm + 63) >> 6;
                    const bool qj_on = fabs(qj) >= kErrorTol;
                    double e4[NTY], sg2[NTY], qq[NTY];
                    bool lj[NTY], c_on[NTY];
                    bool any_c = false, all_c = true, any_lj = false;
#pragma unroll
                    for (int s = 0; s < NTY; ++s) {
                        const double2 pt = pair_tab[rty[s] * nt + tyj];     // scalar load
                        e4[s] = pt.x; sg2[s] = pt.y;
                        lj[s] = pt.x != 0.0;                               // epsilon = 0 contributes 0
                        c_on[s] = qj_on && (fabs(rq[s]) >= kErrorTol);      // energy_utils.f90:430
                        qq[s] = c_on[s] ? rq[s] * qj : 0.0;
                        any_c = any_c || c_on[s]; all_c = all_c && c_on[s]; any_lj = any_lj || lj[s];
                    }
                    if (!(any_c || any_lj)) return;
                    int c = split - (key * cpp) % nsplit;    // units are dealt round-robin
                    if (c < 0) c += nsplit;
                    if (c >= cpp) return;
                    // A unit is "special" when some lane must be masked off: the tail chunk of the
                    // plane, the chunk holding the excluded molecule, or any chunk of an ordered sweep.
                    // Ordinary units skip the masks entirely.
                    auto is_special = [&](int cc) {
                        return ORDERED || (cc == cpp - 1 && (nm & 63) != 0) || (excl && cc == (it.m >> 6));
                    };
                    auto fetch = [&](int cc, bool special, double &x, double &y, double &z, bool &ok) {
                        int m2 = cc * 64 + lane;
                        ok = true;
                        if (special) {
                            // (a real scalar branch: the empty asm keeps the compiler from turning the rare masked unit
                            //  into selects that every ordinary unit would pay for)
                            asm volatile("" ::: "memory");
                            ok = m2 < nm;
                            if (excl) ok = ok && (ORDERED ? (m2 > it.m) : (m2 != it.m));
                            m2 = ok ? m2 : dummy_m;
                        }
                        // one 32-bit byte offset serves the three loads (scalar plane bases + a VGPR offset: no 64-bit address
                        // arithmetic per lane; a plane is far shorter than 4 GB)
                        const unsigned ob = (unsigned)m2 * 8u;
                        x = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(pxp) + ob);
                        y = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(pyp) + ob);
                        z = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(pzp) + ob);
                    };
                    double acc[NREG];
#pragma unroll
                    for (int s = 0; s < NREG; ++s) acc[s] = 0.0;
                    // ALL_C: every site is charged -> the NS Coulomb chains form one basic block
                    auto sweep_plane = [&](auto all_tag) {
                        constexpr bool ALL_C = decltype(all_tag)::value;
                        auto unit = [&](double xj, double yj, double zj, double wq, bool valid) {
                            const double rc2l = valid ? bx.rc2 : -1.0;
                            double r2[NREG], g[NREG];
                            unsigned sh_min = ~0u;
                            if constexpr (TRI) {
                                // the exact eight-evaluation search where the cell allows it (image_r2_tri_lower); a lane whose
                                // certificate fails sends the WHOLE unit to the full search (a scalar branch, rarely taken: the
                                // hot path keeps its registers)
                                bool ok = bx.tri_lower != 0;
                                if (ok) {
#pragma unroll
                                    for (int s = 0; s < NREG; ++s) ok = image_r2_tri_lower(xj - rx[s], yj - ry[s], zj - rz[s], bx, r2[s]) && ok;
                                }
                                if (!__all(ok)) {
                                    asm volatile("" ::: "memory");
#pragma unroll
                                    for (int s = 0; s < NREG; ++s) r2[s] = image_r2_search27(xj - rx[s], yj - ry[s], zj - rz[s], bx.m);
                                }
                            } else {
#pragma unroll
                                for (int s = 0; s < NREG; ++s) {
                                    r2[s] = FASTW ? image_r2_fast(xj - rx[s], yj - ry[s], zj - rz[s], bx)
                                                  : image_r2<false>(xj - rx[s], yj - ry[s], zj - rz[s], bx);
                                }
                            }
#pragma unroll
                            for (int s = 0; s < NREG; ++s) {
                                if (!ALL_C && !c_on[s % NTY]) { g[s] = 0.0; continue; }
                                unsigned sh;
                                g[s] = coul_lds(r2[s], coul_adj, sh);
                                sh_min = min(sh_min, sh);
                            }
                            if (sh_min < (unsigned)bx.coul_idx_base) {   // r < 0.5 A for this lane: rare slow path
#pragma unroll
                                for (int s = 0; s < NREG; ++s)
                                    if ((ALL_C || c_on[s % NTY]) && r2[s] < kCoulSlowBelow) g[s] = coul_slow(r2[s], bx.alpha, ORDERED);
                            }
#pragma unroll
                            for (int s = 0; s < NREG; ++s)
                                if (ALL_C || c_on[s % NTY]) acc[s] = fma(wq, g[s], acc[s]);
                            if (any_lj) {
#pragma unroll
                                for (int s = 0; s < NREG; ++s) {
                                    if (!lj[s % NTY]) continue;
                                    const double s2 = sg2[s % NTY] * fast_rcp(r2[s]);
                                    const double s6 = s2 * s2 * s2;
                                    const double e = e4[s % NTY] * fma(s6, s6, -s6);  // energy_utils.f90:421-423
                                    elj[s / NTY] += (r2[s] < rc2l) ? e : 0.0;            // energy_utils.f90:417
                                }
                            }
                        };
                        double xj, yj, zj;
                        bool valid, special = is_special(c);
                        fetch(c, special, xj, yj, zj, valid);
                        for (; c < cpp; c += nsplit) {
                            double xn = xj, yn = yj, zn = zj;
                            bool vn = true;
                            const bool special_n = is_special(c + nsplit);
                            if (c + nsplit < cpp) fetch(c + nsplit, special_n, xn, yn, zn, vn);
                            // (a mask-free copy of the unit for ordinary chunks was measured slower:
                            //  the duplicated body costs more registers than the masks cost cycles)
                            unit(xj, yj, zj, valid ? 1.0 : 0.0, valid);          // masked-off lanes carry weight 0
                            xj = xn; yj = yn; zj = zn; valid = vn; special = special_n;
                        }
                    };
                    if (all_c) sweep_plane(std::true_type{});
                    else sweep_plane(std::false_type{});
#pragma unroll
                    for (int s = 0; s < NREG; ++s) ec[s / NTY] = fma(qq[s % NTY], acc[s], ec[s / NTY]);
                };
                if (!tp.site_major[t2]) {
                    // plane-major: unit = (site index a2, 64 consecutive molecules); q / type uniform
                    if constexpr (NS > 0) {
                        const int cap2 = tp.cap[t2], seg2 = tp.seg_off[t2];
                        int dummy_m = 0;   // a live, never-excluded molecule for masked-off lanes to read
                        if (same_t) {
                            if (ORDERED) { if (it.m + 1 >= nm) continue; dummy_m = nm - 1; }
                            else { if (nm == 1) continue; dummy_m = (it.m == 0) ? 1 : 0; }
                        }
                        for (int a2 = 0; a2 < n2; ++a2) {
                            const double qj = res_q[t2 * tp.max_atom + a2];
                            const int tyj = res_atype[t2 * tp.max_atom + a2];
                            plane_sweep(px + seg2 + a2 * cap2, py + seg2 + a2 * cap2, pz + seg2 + a2 * cap2, nm, same_t, dummy_m, qj, tyj,
                                        plane_base + a2);
                        }
                    } else {
                        const int cpp = (nm + 63) >> 6, units = n2 * cpp;
                        for (int u = split; u < units; u += nsplit) {
                            const int a2 = u / cpp, m2 = (u - a2 * cpp) * 64 + lane;
                            bool valid = m2 < nm;
                            if (same_t) valid = valid && (ordered ? (m2 > it.m) : (m2 != it.m));
                            const double qj = res_q[t2 * tp.max_atom + a2];
                            const int tyj = res_atype[t2 * tp.max_atom + a2];
                            const bool qj_on = fabs(qj) >= kErrorTol;
                            double xj = 0.0, yj = 0.0, zj = 0.0;
                            if (valid) {
                                const int j = tp.seg_off[t2] + a2 * tp.cap[t2] + m2;
                                xj = px[j]; yj = py[j]; zj = pz[j];
                            }
                            for (int s = 0; s < ns; ++s) {
                                const double qs = w_site[s * 4 + 3];
                                const double2 pt = s_pair[w_sty[s] * nt + tyj];              // (the LDS copy: a global load here sat in the serial chain of every site-term)
                                const bool do_c = qj_on && (fabs(qs) >= kErrorTol);   // energy_utils.f90:430
                                const bool do_lj = pt.x != 0.0;                        // epsilon = 0 contributes 0
                                if ((do_c || do_lj) && valid)
                                    pair_term<ORDERED, TRI>(xj - w_site[s * 4 + 0], yj - w_site[s * 4 + 1], zj - w_site[s * 4 + 2],
                                                       bx, qs * qj, pt.x, pt.y, do_lj, do_c, s_coul, elj[0], ec[0]);
                            }
                        }
                    }
                } else {
                    // site-major: unit = (molecule m2, 64 consecutive sites); per-lane q / type
                    const int cpp = (n2 + 63) >> 6, units = nm * cpp;
                    for (int u = split; u < units; u += nsplit) {
                        const int m2 = u / cpp, a2 = (u - m2 * cpp) * 64 + lane;
                        if (same_t && (ordered ? (m2 <= it.m) : (m2 == it.m))) continue;
                        const bool valid = a2 < n2;
                        double xj = 0.0, yj = 0.0, zj = 0.0, qj = 0.0;
                        int tyj = 0;
                        if (valid) {
                            const int j = tp.seg_off[t2] + m2 * n2 + a2;
                            xj = px[j]; yj = py[j]; zj = pz[j];
                            qj = res_q[t2 * tp.max_atom + a2];
                            tyj = res_atype[t2 * tp.max_atom + a2];
                        }
                        const bool qj_on = fabs(qj) >= kErrorTol;
                        auto one_site = [&](double sx, double sy, double sz, double qs, int tys, double &elj_s, double &ec_s) {
                            const double2 pt = s_pair[tys * nt + tyj];
                            const bool do_c = qj_on && (fabs(qs) >= kErrorTol);
                            const bool do_lj = pt.x != 0.0;                        // epsilon = 0 contributes 0
                            if (valid && (do_lj || do_c))
                                pair_term<ORDERED, TRI, FASTW>(xj - sx, yj - sy, zj - sz, bx, qs * qj, pt.x, pt.y, do_lj, do_c, s_coul, elj_s, ec_s);
                        };
                        if constexpr (NS > 0) {
#pragma unroll
                            for (int s = 0; s < NREG; ++s) one_site(rx[s], ry[s], rz[s], rq[s % NTY], rty[s % NTY], elj[s / NTY], ec[s / NTY]);
                        } else {
                            for (int s = 0; s < ns; ++s)
                                one_site(w_site[s * 4 + 0], w_site[s * 4 + 1], w_site[s * 4 + 2], w_site[s * 4 + 3], w_sty[s], elj[0], ec[0]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            const double a = wave_sum(elj[st]), b = wave_sum(ec[st]);
            if (lane == 0) store_partial<SC1OUT>(partials + (size_t)w * NST + st, a, b);     // fused: {old, new} per work unit
        }
    }
}

template <int NS, bool ORDERED, bool TRI, bool FUSED = false, bool FASTW = false>
__global__ __launch_bounds__(kPairBlock, MGPU_PAIR_MINWAVES) void pair_sweep_kernel(
    Topo tp, BoxDev bx, const double *__restrict__ pos, const int *__restrict__ nmol,
    const double *__restrict__ res_q, const int *__restrict__ res_atype, const double2 *__restrict__ pair_tab,
    const char *__restrict__ coul_tab_g, const PairItem *__restrict__ items, const double *__restrict__ cand_sites,
    int site_stride, int nsplit, int n_work, double2 *__restrict__ partials) {
    constexpr int NST = FUSED ? 2 : 1;
    constexpr int NSLAB = NS > 0 ? 1 : kPairWaves * kSiteChunk;
    extern __shared__ __attribute__((aligned(16))) char s_coul[];     // (coul_last_row + 1) x 48 B
    __shared__ double2 s_pair[kMaxTypes * kMaxTypes];
    __shared__ double s_site[NSLAB * 4];
    __shared__ int s_sty[NSLAB];

    for (int i = threadIdx.x; i < (bx.coul_last_row + 1) * kCoulRowVec; i += kPairBlock)
        reinterpret_cast<double2 *>(s_coul)[i] = reinterpret_cast<const double2 *>(coul_tab_g)[i];
    const int nt = tp.n_types;
    for (int i = threadIdx.x; i < nt * nt; i += kPairBlock) s_pair[i] = pair_tab[i];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_waves = gridDim.x * kPairWaves;
    double *w_site = s_site + (NS > 0 ? 0 : wave * kSiteChunk * 4);
    int *w_sty = s_sty + (NS > 0 ? 0 : wave * kSiteChunk);

    for (int w = blockIdx.x * kPairWaves + wave; w < n_work; w += n_waves) {
        const int item_id = w / nsplit, split = w - item_id * nsplit;
        const PairItem it = items[item_id];
        pair_sweep_item<NS, ORDERED, TRI, FUSED, FASTW>(tp, bx, pos, nmol, res_q, res_atype, pair_tab, s_coul, s_pair, w_site, w_sty, it,
                                                        cand_sites, site_stride, split, nsplit, lane, partials, w);
    }
}


// ------------------------------------------------------------------------------------------
// Flat pair sweep: the same sums as pair_sweep_kernel's register-site path (NS > 0, unordered, orthorhombic), organised
// for topologies whose planes are SHORT -- grand-canonical boxes (a plane of a few hundred molecules is a handful of
// units) and frozen frameworks (one plane per atom-type group).  There the plane-by-plane sweep spends its time on what
// surrounds the arithmetic: per plane a dependent chain scalar loads -> pointers -> first coordinate load -> wait
// (~1 us each, measured: a work unit with 9 units of arithmetic took ~18 us), so this kernel walks ALL units of a
// work unit in ONE software-pipelined loop:
//   * the lanes of the wave build the work unit's plane table in parallel (lane l = plane l: first slot, atom count,
//     exclusion, first unit by a wave scan) into a per-wave LDS slab; each of the item's nsplit waves then takes a
//     CONTIGUOUS share of the unit sequence, so the scalar unit generator is a counter that reads a new plane record
//     (one LDS broadcast) every few units -- dealing units round-robin made it change planes at every unit, and the
//     scalar bookkeeping of a plane change outweighed the unit's arithmetic (measured: no faster than plane by plane);
//   * everything a unit needs is fetched per lane -- x, y, z, the charge (slot_q) and the atom type (slot_ty): 36 bytes
//     per atom, SURVEY 8(d)'s algorithmic figure -- one unit ahead of the arithmetic, across plane and residue
//     boundaries; the atom type of a unit is wave-uniform by construction (readfirstlane), so the (4 epsilon, sigma^2)
//     of a (candidate site, unit) pair is one LDS broadcast read and the LJ half is skipped by a scalar branch where
//     epsilon = 0; the Coulomb half is skipped where no lane of the unit carries a charge (wave vote);
//   * the Coulomb sums run over all units of the work unit (acc[site] += q_lane G(r^2)); the candidate's charges are
//     applied once at the end.
// Semantics per pair term are those of pair_sweep_kernel: LJ inside the cutoff for epsilon != 0 (energy_utils.f90:417-424),
// erfc(alpha r)/r for every distance where both charges are at least 1e-10 in magnitude (energy_utils.f90:427-432).
// ------------------------------------------------------------------------------------------
// One work unit (item, split) of the flat sweep, executed by ONE WAVE; partials[w * NST ...] receives its partials.  s_grp: the
// workgroup's LDS copy of the frozen residues' group records; w_plane: this wave's LDS slab of kFlatMaxPlanes records.
// Shared by pair_flat_kernel and chain_window_kernel.
template <int NS, bool FUSED, bool FASTW, bool SC1OUT = false>
__device__ __forceinline__ void pair_flat_item(
    const Topo &tp, const BoxDev &bx, const double *__restrict__ pos, const int *__restrict__ nmol,
    const double *__restrict__ res_q, const int *__restrict__ res_atype, const char *__restrict__ s_coul,
    const double2 *__restrict__ s_pair, const int4 *__restrict__ s_grp, int4 *__restrict__ w_plane, const PairItem it,
    const double *__restrict__ cand_sites, int site_stride, int split, int nsplit, int lane, int skip_frozen,
    double2 *__restrict__ partials, int w) {
    static_assert(NS > 0, "register sites only");
    constexpr int NTY = NS;
    constexpr int NST = FUSED ? 2 : 1;
    constexpr int NREG = NTY * NST;
    const int nt = tp.n_types;
    const char *coul_adj = coul_tab_adjusted(s_coul, bx.coul_idx_base);
    {
        const double *px = pos + (size_t)it.replica * 3 * tp.n_cap_atoms;
        const double *py = px + tp.n_cap_atoms, *pz = py + tp.n_cap_atoms;
        const int *nm_r = nmol + it.replica * tp.n_res;
        // the replica's molecule counts, requested together (the generator selects among them without a load)
        int nmv[kMaxRes];
#pragma unroll
        for (int i = 0; i < kMaxRes; ++i) nmv[i] = i < tp.n_res ? nm_r[i] : 0;

        // Everything that depends only on the item is requested together -- both states' site coordinates, charges,
        // atom types, the molecule counts above -- so that the work unit pays ONE memory latency here, not one per site
        // (a work unit of a grand-canonical box is only a few units of arithmetic long).
        double rx[NREG], ry[NREG], rz[NREG], rq[NTY];
        int rty[NTY];
#pragma unroll
        for (int a = 0; a < NTY; ++a) {
            if constexpr (FUSED) {
                const int j = atom_slot(tp, it.t, it.m, a);            // old state: the resident slot
                rx[a] = px[j]; ry[a] = py[j]; rz[a] = pz[j];
                const double *c = cand_sites + ((size_t)it.src * site_stride + a) * 3;   // new state: the candidate row
                rx[NTY + a] = c[0]; ry[NTY + a] = c[1]; rz[NTY + a] = c[2];
            } else if (it.src < 0) {
                const int j = atom_slot(tp, it.t, it.m, a);
                rx[a] = px[j]; ry[a] = py[j]; rz[a] = pz[j];
            } else {
                const double *c = cand_sites + ((size_t)it.src * site_stride + a) * 3;
                rx[a] = c[0]; ry[a] = c[1]; rz[a] = c[2];
            }
            rq[a] = res_q[it.t * tp.max_atom + a];
            rty[a] = res_atype[it.t * tp.max_atom + a] * nt;          // row of the pair table
        }
        // wave-uniform values, parked in VGPRs (the sweep needs its SGPRs for the generator and the box)
#pragma unroll
        for (int k = 0; k < NREG; ++k) asm volatile("" : "+v"(rx[k]), "+v"(ry[k]), "+v"(rz[k]));
        // a site below CoulombEnergy's charge threshold contributes exactly 0 (energy_utils.f90:430): its chain is
        // still evaluated with the others (one basic block, NREG-way instruction-level parallelism -- a per-site
        // scalar branch serialises the chains, measured) and weighted 0 at the end
        bool any_c = false;
#pragma unroll
        for (int s = 0; s < NTY; ++s) {
            const bool on = fabs(rq[s]) >= kErrorTol;
            any_c = any_c || on;
            rq[s] = on ? rq[s] : 0.0;
        }

        // ---- plane table of this work unit, built by the lanes in parallel (lane l = plane l of the replica, residue
        //      types in order): {first slot, atoms, excluded-molecule flag | dummy molecule << 1, first unit}.  A plane =
        //      site a2 of every molecule of a plane-major type, or one atom-type group of one molecule of a frozen type.
        int e_off = 0, e_cnt = 0, e_flags = 0;
        {
            int first = 0;                                           // planes before residue type i
#pragma unroll
            for (int i = 0; i < kMaxRes; ++i) {
                const int nm2 = nmv[i];                                 // 0 beyond the topology's residue types
                const bool frozen = tp.site_major[i] == 2;
                const int npl = nm2 == 0 ? 0 : (frozen ? nm2 * tp.n_grp[i] : tp.n1[i]);
                const bool same_t = (i == it.t) && (it.m >= 0);
                const int pl = lane - first;
                if (pl >= 0 && pl < npl) {
                    if (frozen) {
                        const int ng = tp.n_grp[i];
                        const int m2 = nm2 == 1 ? 0 : pl / ng;
                        const int4 gr = s_grp[tp.grp_off[i] + (pl - m2 * ng)];
                        e_off = tp.seg_off[i] + m2 * tp.n1[i] + gr.x;
                        // an inactive molecule evaluated itself is skipped whole; skip_frozen: the frameworks are swept by
                        // pair_frozen_kernel (candidates in the lanes) in the same launch group
                        e_cnt = ((same_t && m2 == it.m) || skip_frozen) ? 0 : gr.y;
                        e_flags = 0;
                    } else {
                        e_off = tp.seg_off[i] + pl * tp.cap[i];
                        e_cnt = (same_t && nm2 == 1) ? 0 : nm2;         // the only molecule of the type is the excluded one
                        e_flags = same_t ? (1 | ((it.m == 0 ? 1 : 0) << 1)) : 0;   // dummy: a live, never-excluded molecule
                    }
                }
                first += npl;
            }
        }
        const int e_units = (e_cnt + 63) >> 6;
        int e_incl = e_units;                                        // inclusive scan over the lanes
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(e_incl, off, 64);
            e_incl += lane >= off ? v : 0;
        }
        const int n_units = __builtin_amdgcn_readlane(e_incl, 63);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        w_plane[lane] = make_int4(e_off, e_cnt, e_flags, e_incl - e_units);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // this wave's share: a contiguous range of the work unit's unit sequence (plane changes stay rare)
        const int u_begin = (int)(((long long)n_units * split) / nsplit), u_end = (int)(((long long)n_units * (split + 1)) / nsplit);
        // ---- unit generator (scalar state): plane p of the table, chunk c of it
        int u = u_begin, p = -1, c = 0, cpp = 0;
        int p_off = 0, p_cnt = 0, p_excl = -1, p_dummy = 0;
        bool done = u >= u_end;
        auto load_plane = [&]() {
            const int4 e = w_plane[p];                               // LDS broadcast read
            p_off = __builtin_amdgcn_readfirstlane(e.x);
            p_cnt = __builtin_amdgcn_readfirstlane(e.y);
            const int fl = __builtin_amdgcn_readfirstlane(e.z);
            const int us = __builtin_amdgcn_readfirstlane(e.w);
            p_excl = (fl & 1) ? it.m : -1;
            p_dummy = fl >> 1;
            cpp = (p_cnt + 63) >> 6;
            c = u - us;
        };
        if (!done) {
            // the plane that holds unit u_begin: the last one whose first unit is not beyond it (empty planes excluded)
            const unsigned long long mk = __ballot(e_units > 0 && (e_incl - e_units) <= u_begin);
            p = 63 - __builtin_clzll(mk);
            load_plane();
        }
        auto next_unit = [&]() {
            ++u;
            ++c;
            if (u >= u_end) { done = true; return; }
            while (c >= cpp) {                                       // next non-empty plane
                ++p;
                load_plane();
            }
        };
        // a unit's operands, fetched per lane: masked-off lanes (tail of the plane, the excluded molecule) read the
        // plane's dummy molecule and carry charge 0 and an LJ cutoff of -1
        auto fetch = [&](double &x, double &y, double &z, double &q, int &ty, bool &ok) {
            const int m2 = c * 64 + lane;
            ok = m2 < p_cnt && m2 != p_excl;
            const int j = p_off + (ok ? m2 : p_dummy);
            x = px[j]; y = py[j]; z = pz[j];
            q = tp.slot_q[j];
            ty = tp.slot_ty[j];
        };

        double acc[NREG], elj[NST];
#pragma unroll
        for (int s = 0; s < NREG; ++s) acc[s] = 0.0;
#pragma unroll
        for (int st = 0; st < NST; ++st) elj[st] = 0.0;

        double xn = 0.0, yn = 0.0, zn = 0.0, qn = 0.0;
        int tyn = 0;
        bool vn = false;
        if (!done) fetch(xn, yn, zn, qn, tyn, vn);
        while (!done) {
            const double xj = xn, yj = yn, zj = zn;
            const bool valid = vn;
            const double wq = (valid && fabs(qn) >= kErrorTol) ? qn : 0.0;       // energy_utils.f90:430
            const int tyj = __builtin_amdgcn_readfirstlane(tyn);                   // uniform over the unit
            next_unit();
            if (!done) fetch(xn, yn, zn, qn, tyn, vn);

            const double rc2l = valid ? bx.rc2 : -1.0;
            double r2[NREG];
#pragma unroll
            for (int s = 0; s < NREG; ++s)
                r2[s] = FASTW ? image_r2_fast(xj - rx[s], yj - ry[s], zj - rz[s], bx)
                              : image_r2<false>(xj - rx[s], yj - ry[s], zj - rz[s], bx);
            if (any_c && __ballot(wq != 0.0) != 0ull) {
                double g[NREG];
                unsigned sh_min = ~0u;
#pragma unroll
                for (int s = 0; s < NREG; ++s) {
                    unsigned sh;
                    g[s] = coul_lds(r2[s], coul_adj, sh);
                    sh_min = min(sh_min, sh);
                }
                if (sh_min < (unsigned)bx.coul_idx_base) {   // r < 0.5 A for this lane: rare slow path
#pragma unroll
                    for (int s = 0; s < NREG; ++s)
                        if (r2[s] < kCoulSlowBelow) g[s] = coul_slow(r2[s], bx.alpha, false);
                }
#pragma unroll
                for (int s = 0; s < NREG; ++s) acc[s] = fma(wq, g[s], acc[s]);
            }
            // (4 epsilon, sigma^2) of every site against this unit's atom type: LDS broadcast reads, requested together
            double2 pt[NTY];
            bool lj_on[NTY];
#pragma unroll
            for (int s = 0; s < NTY; ++s) pt[s] = s_pair[rty[s] + tyj];
#pragma unroll
            for (int s = 0; s < NTY; ++s)
                lj_on[s] = (__builtin_amdgcn_readfirstlane(__double2hiint(pt[s].x)) | __builtin_amdgcn_readfirstlane(__double2loint(pt[s].x))) != 0;
#pragma unroll
            for (int s = 0; s < NTY; ++s) {
                if (!lj_on[s]) continue;                                           // epsilon = 0 contributes 0
#pragma unroll
                for (int st = 0; st < NST; ++st) {
                    const double rr = r2[st * NTY + s];
                    const double s2 = pt[s].y * fast_rcp(rr);
                    const double s6 = s2 * s2 * s2;
                    const double e = pt[s].x * fma(s6, s6, -s6);                  // energy_utils.f90:421-423
                    elj[st] += (rr < rc2l) ? e : 0.0;                              // energy_utils.f90:417
                }
            }
        }
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            double ec = 0.0;
#pragma unroll
            for (int s = 0; s < NTY; ++s) ec = fma(rq[s], acc[st * NTY + s], ec);
            const double a = wave_sum(elj[st]), b = wave_sum(ec);
            if (lane == 0) store_partial<SC1OUT>(partials + (size_t)w * NST + st, a, b);
        }
    }
}

template <int NS, bool FUSED, bool FASTW>
__global__ __launch_bounds__(kPairBlock, MGPU_PAIR_MINWAVES) void pair_flat_kernel(
    Topo tp, BoxDev bx, const double *__restrict__ pos, const int *__restrict__ nmol,
    const double *__restrict__ res_q, const int *__restrict__ res_atype, const double2 *__restrict__ pair_tab,
    const char *__restrict__ coul_tab_g, const PairItem *__restrict__ items, const double *__restrict__ cand_sites,
    int site_stride, int nsplit, int n_work, double2 *__restrict__ partials, int skip_frozen) {
    constexpr int NST = FUSED ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) char s_coul[];     // (coul_last_row + 1) x 48 B
    __shared__ double2 s_pair[kMaxTypes * kMaxTypes];
    __shared__ int4 s_grp[kMaxGrp];                       // group records of the frozen residues
    __shared__ int4 s_plane[kPairWaves * kFlatMaxPlanes]; // per wave: the plane table of its current work unit

    if (threadIdx.x < kMaxGrp) s_grp[threadIdx.x] = make_int4(tp.grp_start[threadIdx.x], tp.grp_cnt[threadIdx.x], tp.grp_ty[threadIdx.x], 0);
    for (int i = threadIdx.x; i < (bx.coul_last_row + 1) * kCoulRowVec; i += kPairBlock)
        reinterpret_cast<double2 *>(s_coul)[i] = reinterpret_cast<const double2 *>(coul_tab_g)[i];
    const int nt = tp.n_types;
    for (int i = threadIdx.x; i < nt * nt; i += kPairBlock) s_pair[i] = pair_tab[i];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_waves = gridDim.x * kPairWaves;

    for (int w = blockIdx.x * kPairWaves + wave; w < n_work; w += n_waves) {
        const int item_id = w / nsplit, split = w - item_id * nsplit;
        const PairItem it = items[item_id];
        pair_flat_item<NS, FUSED, FASTW>(tp, bx, pos, nmol, res_q, res_atype, s_coul, s_pair, s_grp, s_plane + wave * kFlatMaxPlanes, it,
                                         cand_sites, site_stride, split, nsplit, lane, skip_frozen, partials, w);
    }
}


// ------------------------------------------------------------------------------------------
// Frameworks, transposed: the CANDIDATES sit in the lanes.  An inactive framework is the same in every replica of a
// farm (the engine verifies it on upload), so a wave takes 64 items of one residue type and ONE chunk of 64 framework
// atoms (chunk_atoms <= 64, an engine constant): each lane loads one atom of the chunk (coalesced), the wave then walks them with v_readlane -- atom
// coordinates, charge and type are SCALARS -- against the lane's own NREG candidate sites in registers.  No masks, no
// tails, no per-unit bookkeeping, no cross-lane reduction: per (site, atom) term just the distance, the table and the
// accumulate.  The (4 epsilon,
// sigma^2) of the candidate's sites against the atom's type are reloaded (LDS broadcast) only when the type changes: the
// frozen layout keeps atoms of one type together.  The few OTHER atoms of a framework box (the adsorbates of each lane's
// own replica) follow in the same wave with per-lane coordinates, so one kernel yields the item's whole pair energy.
// Work = (item group, chunk); a workgroup adds the partials {e_lj, e_coul} of its eight chunks in chunk order and writes one
// record per entry (scratch laid out [workgroup of the group][entry]); the group's last workgroup adds an entry's records in
// workgroup order -- one extra record per entry for the host's ordered sum.  Items: PairItem of ONE residue type (the engine checks), unordered, orthorhombic.
// ------------------------------------------------------------------------------------------
template <int NS, bool FUSED, bool FASTW>
__global__ __launch_bounds__(kPairBlock, MGPU_PAIR_MINWAVES) void pair_frozen_kernel(
    Topo tp, BoxDev bx, const double *__restrict__ pos, const int *__restrict__ nmol,
    const double *__restrict__ res_q, const int *__restrict__ res_atype, const double2 *__restrict__ pair_tab,
    const char *__restrict__ coul_tab_g, const PairItem *__restrict__ items, const double *__restrict__ cand_sites,
    int site_stride, int n_items, int t_frozen, int n_chunks, int chunk_atoms, double2 *__restrict__ scratch,
    int *__restrict__ tickets, double2 *__restrict__ extra, const double *__restrict__ slot_q_on, const int *__restrict__ slot_ty) {
    constexpr int NTY = NS;
    constexpr int NST = FUSED ? 2 : 1;
    constexpr int NREG = NTY * NST;
    extern __shared__ __attribute__((aligned(16))) char s_coul[];     // (coul_last_row + 1) x 48 B
    __shared__ double2 s_pair[kMaxTypes * kMaxTypes];
    __shared__ double s_cand[NREG * 3 * 64];                          // the group's candidate sites, [site-state][x, y, z][lane]
    __shared__ double2 s_part[kPairWaves * NST * 64];                 // the eight chunk partials of the group, [wave][state][lane]
    __shared__ int s_last;
    for (int i = threadIdx.x; i < (bx.coul_last_row + 1) * kCoulRowVec; i += kPairBlock)
        reinterpret_cast<double2 *>(s_coul)[i] = reinterpret_cast<const double2 *>(coul_tab_g)[i];
    const int nt = tp.n_types;
    for (int i = threadIdx.x; i < nt * nt; i += kPairBlock) s_pair[i] = pair_tab[i];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char *coul_adj = coul_tab_adjusted(s_coul, bx.coul_idx_base);
    const int n_groups = (n_items + 63) >> 6;
    // the framework's atoms: every molecule of the frozen type of replica 0 (identical in all replicas)
    const int n_atoms = nmol[t_frozen] * tp.n1[t_frozen];
    const double *fx = pos + tp.seg_off[t_frozen], *fy = fx + tp.n_cap_atoms, *fz = fy + tp.n_cap_atoms;
    // (the slots' charges -- those below CoulombEnergy's threshold already zero -- and types as read-only arguments of
    //  their own: only those become scalar loads; tp.slot_q / tp.slot_ty are members of a by-value struct)
    const double *__restrict__ fq = slot_q_on + tp.seg_off[t_frozen];
    const int *__restrict__ fty = slot_ty + tp.seg_off[t_frozen];

    // A workgroup takes ONE group of 64 candidates and eight of its chunks (one per wave).  The candidates' sites -- per
    // lane a different replica: 64 separate cache lines per load -- are gathered ONCE per workgroup into LDS, each thread
    // one or two of the NREG x 3 x 64 values, and every wave takes its lanes' values from there: an eighth of the gathers
    // of the one-unit-per-wave form, where each wave gathered all 12-30 values of its lanes itself (stage stamps inside
    // the kernel, round 4, bench's framework box: 8.0 -> 3.8 us from the unit's start to its first framework atom).
    // Which wave computes a unit does not enter its partial: same bits.
    const int wg_per_group = (n_chunks + kPairWaves - 1) / kPairWaves;
    const int n_wg_units = n_groups * wg_per_group;
    for (int b = blockIdx.x; b < n_wg_units; b += gridDim.x) {
        const int grp = b / wg_per_group, chunk = (b - grp * wg_per_group) * kPairWaves + wave;
        __syncthreads();                                                    // the previous group's readers are done
        for (int idx = threadIdx.x; idx < NREG * 3 * 64; idx += kPairBlock) {
            const int ln = idx & 63, comp = idx >> 6, sreg = comp / 3, d = comp - 3 * sreg;
            const int id = grp * 64 + ln;
            const PairItem il = items[id < n_items ? id : n_items - 1];
            const double *pd = pos + (size_t)il.replica * 3 * tp.n_cap_atoms + (size_t)d * tp.n_cap_atoms;
            const bool resident = FUSED ? (sreg < NTY) : (il.src < 0);
            const int a = (FUSED && sreg >= NTY) ? sreg - NTY : sreg;
            s_cand[idx] = resident ? pd[atom_slot(tp, il.t, il.m, a)]
                                   : cand_sites[((size_t)(il.src < 0 ? 0 : il.src) * site_stride + a) * 3 + d];
        }
        __syncthreads();
        do {
        if (chunk >= n_chunks) break;                                       // (uniform per wave; the barriers are outside)
        const int item_id = grp * 64 + lane;
        const bool live = item_id < n_items;
        const PairItem it = items[live ? item_id : n_items - 1];
        const double *px = pos + (size_t)it.replica * 3 * tp.n_cap_atoms;
        const double *py = px + tp.n_cap_atoms, *pz = py + tp.n_cap_atoms;
        // the lane's own candidate: both states' sites, charges / types of its residue type (uniform)
        double rx[NREG], ry[NREG], rz[NREG], rq[NTY];
        int rty[NTY];
#pragma unroll
        for (int sreg = 0; sreg < NREG; ++sreg) {
            rx[sreg] = s_cand[(sreg * 3 + 0) * 64 + lane];
            ry[sreg] = s_cand[(sreg * 3 + 1) * 64 + lane];
            rz[sreg] = s_cand[(sreg * 3 + 2) * 64 + lane];
        }
        const int t_item = __builtin_amdgcn_readfirstlane(it.t);          // one residue type per launch
        bool any_c = false, q_on[NTY];                                    // uniform: one residue type per launch
#pragma unroll
        for (int s = 0; s < NTY; ++s) {
            rq[s] = res_q[t_item * tp.max_atom + s];
            rty[s] = res_atype[t_item * tp.max_atom + s] * nt;
            q_on[s] = fabs(rq[s]) >= kErrorTol;                           // energy_utils.f90:430
            any_c = any_c || q_on[s];
            rq[s] = q_on[s] ? rq[s] : 0.0;
        }
        // this chunk's atoms
        const int a0 = chunk * chunk_atoms, na = min(chunk_atoms, n_atoms - a0);        // chunk_atoms <= 64

        double acc[NREG], elj[NST];
#pragma unroll
        for (int s = 0; s < NREG; ++s) acc[s] = 0.0;
#pragma unroll
        for (int st = 0; st < NST; ++st) elj[st] = 0.0;
        int cur_ty = -1;
        double e4[NTY], sg2[NTY];
        bool lj_on[NTY];
#pragma unroll
        for (int s = 0; s < NTY; ++s) { e4[s] = 0.0; sg2[s] = 0.0; lj_on[s] = false; }
        for (int k = 0; k < na; ++k) {
            // the framework atom as SCALARS: x, y, z, q and type through the scalar cache (the arrays are read-only kernel
            // arguments and the index is uniform), no vector instruction spent on broadcasting them (round 4: nine
            // v_readlane per atom before; 53.9 -> 51.5 us; with the next atom's five loads requested a step ahead the scalar
            // registers spill: 52.9)
            const int jk = __builtin_amdgcn_readfirstlane(a0 + k);
            const double xj = fx[jk], yj = fy[jk], zj = fz[jk];
            const double qj = fq[jk];                                     // (thresholded on the host: zero below 1e-10)
            const int tyj = fty[jk];
            if (tyj != cur_ty) {                                            // rare: the atoms are sorted by type
                cur_ty = tyj;
#pragma unroll
                for (int s = 0; s < NTY; ++s) {
                    const double2 pt = s_pair[rty[s] + tyj];                // LDS broadcast read
                    e4[s] = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(pt.x)), __builtin_amdgcn_readfirstlane(__double2loint(pt.x)));
                    sg2[s] = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(pt.y)), __builtin_amdgcn_readfirstlane(__double2loint(pt.y)));
                    lj_on[s] = e4[s] != 0.0;                               // epsilon = 0 contributes 0
                }
            }
            double r2[NREG];
#pragma unroll
            for (int s = 0; s < NREG; ++s)
                r2[s] = FASTW ? image_r2_fast(xj - rx[s], yj - ry[s], zj - rz[s], bx)
                              : image_r2<false>(xj - rx[s], yj - ry[s], zj - rz[s], bx);
            if (any_c && qj != 0.0) {
                double g[NREG];
                unsigned sh_min = ~0u;
#pragma unroll
                for (int s = 0; s < NREG; ++s) {
                    // a site without charge (the oxygen of a four-site water) takes no table row: the reference skips the
                    // pair (energy_utils.f90:430) and its sum is multiplied by q = 0 below
                    g[s] = 0.0;
                    if (q_on[s % NTY]) {
                        unsigned sh;
                        g[s] = coul_lds(r2[s], coul_adj, sh);
                        sh_min = min(sh_min, sh);
                    }
                }
                if (sh_min < (unsigned)bx.coul_idx_base) {   // r < 0.5 A for this lane: rare slow path
#pragma unroll
                    for (int s = 0; s < NREG; ++s)
                        if (q_on[s % NTY] && r2[s] < kCoulSlowBelow) g[s] = coul_slow(r2[s], bx.alpha, false);
                }
#pragma unroll
                for (int s = 0; s < NREG; ++s) acc[s] = fma(qj, g[s], acc[s]);
            }
#pragma unroll
            for (int s = 0; s < NTY; ++s) {
                if (!lj_on[s]) continue;
#pragma unroll
                for (int st = 0; st < NST; ++st) {
                    const double rr = r2[st * NTY + s];
                    const double s2 = sg2[s] * fast_rcp(rr);
                    const double s6 = s2 * s2 * s2;
                    const double e = e4[s] * fma(s6, s6, -s6);                     // energy_utils.f90:421-423
                    elj[st] += (rr < bx.rc2) ? e : 0.0;                            // energy_utils.f90:417
                }
            }
        }
        // ---- everything else in the lanes' replicas: the molecules of the plane-major (active) residue types, dealt over
        //      the group's chunk waves (molecule m2 = chunk, chunk + n_chunks, ...).  Coordinates are per lane here (every
        //      lane has its own replica: 64 separate 8-byte gathers per load, affordable for the few dozen adsorbate atoms of
        //      a framework box); charge, type and LJ pair are uniform per (type, site).  Masked lanes (no such molecule in
        //      their replica, or the candidate itself) are removed by selects, never by a zero weight: their slot may hold
        //      anything.
        for (int t2 = 0; t2 < tp.n_res; ++t2) {
            if (tp.site_major[t2] != 0) continue;
            const int nm_l = live ? nmol[it.replica * tp.n_res + t2] : 0;
            int nm_max = nm_l;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) nm_max = max(nm_max, __shfl_xor(nm_max, off, 64));
            nm_max = __builtin_amdgcn_readfirstlane(nm_max);
            const int n2 = tp.n1[t2], cap2 = tp.cap[t2], seg2 = tp.seg_off[t2];
            const bool same_t = t2 == t_item;
            for (int m2 = chunk; m2 < nm_max; m2 += n_chunks) {
                const bool ok = m2 < nm_l && !(same_t && m2 == it.m);
                for (int a2 = 0; a2 < n2; ++a2) {
                    double qj = res_q[t2 * tp.max_atom + a2];                       // scalar
                    qj = fabs(qj) >= kErrorTol ? qj : 0.0;
                    const int tyj = res_atype[t2 * tp.max_atom + a2];
                    const int j = seg2 + a2 * cap2 + (ok ? m2 : 0);
                    const double xj = px[j], yj = py[j], zj = pz[j];
                    double r2[NREG];
#pragma unroll
                    for (int s = 0; s < NREG; ++s)
                        r2[s] = FASTW ? image_r2_fast(xj - rx[s], yj - ry[s], zj - rz[s], bx)
                                      : image_r2<false>(xj - rx[s], yj - ry[s], zj - rz[s], bx);
                    if (any_c && qj != 0.0) {
                        double g[NREG];
                        unsigned sh_min = ~0u;
#pragma unroll
                        for (int s = 0; s < NREG; ++s) {
                            g[s] = 0.0;
                            if (q_on[s % NTY]) {
                                unsigned sh;
                                g[s] = coul_lds(r2[s], coul_adj, sh);
                                sh_min = min(sh_min, sh);
                            }
                        }
                        if (ok && sh_min < (unsigned)bx.coul_idx_base) {
#pragma unroll
                            for (int s = 0; s < NREG; ++s)
                                if (ok && q_on[s % NTY] && r2[s] < kCoulSlowBelow) g[s] = coul_slow(r2[s], bx.alpha, false);
                        }
#pragma unroll
                        for (int s = 0; s < NREG; ++s) acc[s] = ok ? fma(qj, g[s], acc[s]) : acc[s];
                    }
#pragma unroll
                    for (int s = 0; s < NTY; ++s) {
                        const double2 pt = pair_tab[rty[s] + tyj];                  // scalar load
                        if (pt.x == 0.0) continue;                                  // epsilon = 0 contributes 0
#pragma unroll
                        for (int st = 0; st < NST; ++st) {
                            const double rr = r2[st * NTY + s];
                            const double s2 = pt.y * fast_rcp(rr);
                            const double s6 = s2 * s2 * s2;
                            const double e = pt.x * fma(s6, s6, -s6);              // energy_utils.f90:421-423
                            elj[st] += (ok && rr < bx.rc2) ? e : 0.0;              // energy_utils.f90:417
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            double ec = 0.0;
#pragma unroll
            for (int s = 0; s < NTY; ++s) ec = fma(rq[s], acc[st * NTY + s], ec);
            s_part[(wave * NST + st) * 64 + lane] = make_double2(elj[st], ec);
        }
        } while (0);
        // the workgroup's chunks summed in chunk order by one thread per (state, candidate): one record per entry and
        // workgroup, an eighth of the chunk records (sweep + finalize 58.2 -> 54.7 us at the bench's framework box) ...
        __syncthreads();
        const int wgc = b - grp * wg_per_group;
        const int st_t = threadIdx.x >> 6, id_t = grp * 64 + lane;
        const bool summing = (int)threadIdx.x < NST * 64 && id_t < n_items;
        double2 *rec = scratch + ((size_t)id_t * NST + st_t);                        // + workgroup * n_items * NST: [workgroup][entry]
        if (summing) {
            const int n_valid = min(kPairWaves, n_chunks - wgc * kPairWaves);
            double ea = 0.0, eb = 0.0;
            for (int wv = 0; wv < n_valid; ++wv) {
                const double2 pp = s_part[(wv * NST + st_t) * 64 + lane];
                ea += pp.x; eb += pp.y;
            }
            store_partial<true>(rec + (size_t)wgc * ((size_t)n_items * NST), ea, eb);
        }
        // ... and the group's LAST workgroup to get here adds the group's records in workgroup order into the entries'
        // extra records (what a finalize kernel did in a launch of its own: 54.7 -> 53.7 us and one launch less).  Hand-off as in
        // chain_window_kernel: agent-scope write-through stores, every storing wave waits for them, one lane per workgroup
        // draws the group's ticket behind a barrier, the last one reads the records with agent-scope loads; it leaves the
        // ticket at zero for the lane's next launch.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) s_last = atomicAdd(&tickets[grp], 1) == wg_per_group - 1;
        __syncthreads();
        if (s_last) {
            if (summing) {
                double ea = 0.0, eb = 0.0;
                for (int q = 0; q < wg_per_group; ++q) {
                    const double *pr = reinterpret_cast<const double *>(rec + (size_t)q * ((size_t)n_items * NST));
                    ea += load_sc1(pr); eb += load_sc1(pr + 1);
                }
                extra[(size_t)id_t * NST + st_t] = make_double2(ea, eb);
            }
            if (threadIdx.x == 0) __hip_atomic_store(&tickets[grp], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Ordered sum of the split partials; Coulomb rescale e_coulomb * EPS0_INV_eVA / KB_eVK
// (energy_utils.f90:440).
static __global__ void pair_finalize_kernel(const double2 *__restrict__ partials, int n_items, int nsplit,
                                     double *__restrict__ e_lj, double *__restrict__ e_coul) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) return;
    double a = 0.0, b = 0.0;
    for (int s = 0; s < nsplit; ++s) { const double2 p = partials[(size_t)i * nsplit + s]; a += p.x; b += p.y; }
    e_lj[i] = a;
    e_coul[i] = b * kEps0InvEvA / kKbEvK;
}

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
template <bool COMMIT, bool BOTH, bool MFMA = false>
__global__ __launch_bounds__(kBlock, 2) void recip_rows_wide_kernel(
    Topo tp, BoxDev bx, double *__restrict__ pos, int *__restrict__ nmol, const double *__restrict__ res_q,
    const int *__restrict__ trj, const double2 *__restrict__ tw, const RecipRow *__restrict__ rows, const int *__restrict__ row_first,
    int n_rows, int rows_per_tile, int nss_max, double2 *__restrict__ A_base, const RecipItem *__restrict__ items,
    const double *__restrict__ cand_sites, int site_stride, double *__restrict__ u_new, double *__restrict__ u_old) {
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
    for (int e = tid; e < nss_fill * ktot; e += kBlock) {
        const int s = e / ktot, kk = e - s * ktot;
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
    for (int s = tid; s < nss_fill; s += kBlock) {
        if (MFMA && s >= nss) { sq[s] = 0.0; continue; }
        const int set = two_sets ? (s >= n1 ? 1 : 0) : (use_old ? 1 : 0), a = s - (s >= n1 ? n1 : 0);
        const double q = res_q[it.t * tp.max_atom + a];
        sq[s] = (set == 0 ? q : -q) * (used ? 1.0 : 0.0);   // + for the new sites, - for the old ones (ewald_energy.f90:241-256)
    }
    // matrix-unit form: every row's {kx, ky, first task, first kz | tasks << 8} beside the tables, so that a tile's
    // addresses cost one LDS read instead of a chain of three global loads per tile
    [[maybe_unused]] int4 *rowmeta = reinterpret_cast<int4 *>(sq + nss_max);
    if constexpr (MFMA) {
        for (int rr = tid; rr < n_rows; rr += kBlock) {
            const RecipRow r = rows[rr];
            const int t0 = row_first[rr], t1 = row_first[rr + 1];
            const int j0 = t1 > t0 ? (trj[t0] & 0xff) : 0;
            rowmeta[rr] = make_int4(r.kx, r.ky, t0, j0 | ((t1 - t0) << 8));
        }
    }
    __syncthreads();
    double2 *A = A_base + (size_t)it.replica * bx.n_slots;
    const double2 *zt = tab + kofs2;
    double acc = 0.0, acc0 = 0.0;
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
            int tt[4], rjv[4];
            double2 Apv[4], Amv[4], wv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int kzz = ztile * 16 + lk + 4 * i;
                const int t = t0 + (kzz - j0);
                tt[i] = (kzz >= j0 && t < t1) ? t : -1;
                const int tc = tt[i] >= 0 ? tt[i] : 0;
                rjv[i] = trj[tc];
                Apv[i] = A[2 * tc]; Amv[i] = A[2 * tc + 1];
                wv[i] = COMMIT ? make_double2(0.0, 0.0) : tw[tc];
            }
            const int aky = r.ky < 0 ? -r.ky : r.ky;
            const double ysign = r.ky < 0 ? -1.0 : 1.0;          // conjugate for -ky (times -1: exact)
            // Branch-free steps: every address is valid -- the site-states are padded with zeros, a row beyond the last reads
            // the last row's, a kz beyond the table's reads the neighbouring entries -- and what such lanes feed the matrix
            // unit only reaches outputs no task owns (an output depends on its own kz's and its own row's operands alone).
            const double2 *xp = tab + lk * ktot + r.kx, *yp = tab + lk * ktot + kofs1 + aky, *zp = zt + lk * ktot + ztile * 16 + li;
            const double *qp = sq + lk;
            double4v d_ac = {0.0, 0.0, 0.0, 0.0}, d_bd = d_ac, d_ad = d_ac, d_bc = d_ac;
            double2 Xn = xp[0], Yn = yp[0], zn = zp[0];
            double qn = qp[0];
            for (int s0 = 0; s0 < nss_fill; s0 += 4) {
                const double2 X = Xn, z = zn;
                double2 Y = Yn;
                const double q = qn;
                const int sn = s0 + 4 < nss_fill ? s0 + 4 : s0;    // (the last step re-reads its own operands)
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
                const double sac = d_ac[i], sbd = d_bd[i], sad = d_ad[i], sbc = d_bc[i];
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
    } else
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
template <bool FLAT, bool FASTW>
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
                    pair_sweep_item<NS, false, false, false, FASTW, true>(tp, bx, pos, nmol, res_q, res_atype, pair_tab, s_dyn, s_pair, nullptr, \
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

// One chain resolved by ONE WAVE (all 64 lanes arrive): `scratch` = 4 nsplit + 4 doubles of LDS of its own.
__device__ __forceinline__ void farm_resolve(const Topo &tp, const BoxDev &bx, double *__restrict__ pos, int *__restrict__ nmol,
                                             const FarmArgs &g, const FarmRec &rec, int c, int lane, double *scratch) {
    double *ho = g.host_out + (size_t)kFarmOut * c;
    int verdict;
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
    {
        double val = 0.0;
        if (lane < 5) val = o[lane];
        else if (lane < 10) val = w[lane - 5];
        else if (lane == 10) val = (double)verdict;
        if (lane < kFarmOut) __hip_atomic_store(ho + lane, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(g.host_tag + c, g.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // ---- the chain's device state: ticket, stall flag, and the accepted step itself
    if (lane == 0) {
        __hip_atomic_store(g.tickets + c, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (rec.move != 0 && (verdict == kFarmVerdictUndecided || rec.forced)) g.stalled[rec.replica] = verdict == kFarmVerdictUndecided ? 1 : 0;
    }
    if (verdict != kFarmVerdictAccepted) return;
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
            double2 *A = (s_acur ? g.A_alt : A_base) + (size_t)rec.replica * bx.n_slots;
            double2 *A_other = (s_acur ? A_base : g.A_alt) + (size_t)rec.replica * bx.n_slots;
            RecipItem it{rec.replica, rec.t, kind == 1 ? -1 : rec.m, kind, 0, 0, 0};
            const RecipLds v = recip_lds_view(tp, bx, it, n_rows, reinterpret_cast<double2 *>(s_dyn));
            const bool active = tid < kBlock;
            RecipInFlight<kRecipTaskChunk> inflight;
            recip_rows_tables(tp, bx, pos, res_q, rows, n_rows, it, &s_cand[0][0], v, tid, active,
                              [&] { recip_rows_prefetch<false>(inflight, trj, tw, n_tasks, A, tid); });
            double acc = 0.0, acc0 = 0.0;
            if (active) recip_rows_pass<false, true, kRecipTaskChunk, 1>(v, trj, tw, n_tasks, A, tid, inflight, acc, acc0, A_other);
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
    }

    // ---------------- tickets: one counter per chain; the workgroup whose add completes a chain's count resolves it
    // (every storing wave waits for its `sc1` stores, ONE lane per workgroup and chain adds behind the barrier, and the
    // resolving wave loads with `sc1` behind a second barrier that the adding wave joins)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid < n_c) {
        int count = 1;
        if (!k_role) {
            const int a = max(w0, (c_lo + tid) * wpc), b = min(w1, (c_lo + tid + 1) * wpc);
            count = b - a;
        }
        s_resolve[tid] = (atomicAdd(g.tickets + c_lo + tid, count) + count == expected) ? 1 : 0;
    }
    __syncthreads();
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
