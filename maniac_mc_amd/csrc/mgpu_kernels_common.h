// HIP kernels of the GCMC energy engine, written for gfx950 (CDNA4, wave64) only.  This part: constants, the device views of
// the topology and the box, minimum-image folds, the Coulomb table in LDS, the pair term.
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
#ifndef MGPU_KERNELS_COMMON_H
#define MGPU_KERNELS_COMMON_H

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

}  // namespace mgpu

#endif
