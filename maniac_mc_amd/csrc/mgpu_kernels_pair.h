// Pair sweeps: plane by plane (pair_sweep_kernel), flat (pair_flat_kernel), candidates in the lanes (pair_frozen_kernel).
// (one of the four parts of mgpu_kernels.h: include that header, not this file)
#ifndef MGPU_KERNELS_PAIR_H
#define MGPU_KERNELS_PAIR_H

#include "mgpu_kernels_common.h"

namespace mgpu {

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

}  // namespace mgpu

#endif
