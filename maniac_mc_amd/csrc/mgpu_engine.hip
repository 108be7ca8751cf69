// C-ABI entry points (include/maniac_gpu.h): the engine's life cycle, replica state, structure factor, static energy,
// measurement -- and the helpers every translation unit of the engine shares (mgpu_engine.h).
// One mgpu_engine = one HIP device + kLanes streams + R replicas sharing box / force field / k table.
#include "mgpu_engine.h"

namespace mgpu {

static thread_local std::string g_last_error;

int set_error(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}
const char *last_error_text() { return g_last_error.c_str(); }

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
    // farm windows leave a replica's current A(k) in either of its two buffers: back into the primary one
    if (e->farm.dirty) return farm_window_normalize(e);
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
// A small engine (< 48 replicas: single chains, a handful of chains) is latency-bound: up to 32 waves per item, two sweep
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
    if (e->n_replicas >= 48 && e->n_replicas <= 1024) {
        // The farm-window regime (round 5: one launch per lane step, mgpu_farm_window_submit; a launch carries 2 nsplit pair
        // waves per chain): the launch should fill the GPU's 4096 wave slots about once -- fewer waves leave it idle, more
        // run in rounds whose fixed costs (table staging, per-plane latency) add up.  Measured at the 10 125-atom box, one
        // lane, two windows in flight, accepted moves/s at nsplit 2 / 4 / 8 / 16 / 32: 64 chains 0.85 / 1.32 / 1.72 / 2.04 /
        // 1.45 M, 128: 1.65 / 2.53 / 3.29 / 2.55 / 2.00 M, 256: 3.17 / 4.73 / 4.10 / 3.39 / 2.44 M, 512: 5.87 / 5.49 (6.06 on
        // two lanes) / 4.51 / 3.79 / 2.63 M, 1024: 5.58-5.91 / 5.26-5.64 / 4.34-4.91 / 3.56 / 2.06 M
        // (profiles/r05/farm_window_nsplit.txt).  The batched path of such an engine is the fallback only (it preferred
        // 16 / 8 / 4 at 256 / 512 / 1024 chains: 1.42 / 2.40 / 4.84 M).
        const int want = (e->n_replicas < 512 ? 1024 : 2048) / e->n_replicas;
        cap_split = 2;
        while (cap_split * 2 <= std::min(want, 32)) cap_split *= 2;
    } else if (e->n_replicas >= 256) {
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


}  // namespace mgpu

extern "C" {

const char *mgpu_last_error(void) { return g_last_error.c_str(); }

int mgpu_abi_version(void) { return MGPU_ABI_VERSION; }

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
        build_layout(want_flat);
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
    e->pair_fast_fold = std::getenv("MGPU_PAIR_EXACT_FOLD") == nullptr;
    e->recip_no_mfma = std::getenv("MGPU_RECIP_NO_MFMA") != nullptr;
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
    // lower-triangular box%matrix (the reader's triclinic cells): the exact eight-evaluation image search applies
    bx.tri_lower = (bx.triclinic && box_matrix[1] == 0.0 && box_matrix[2] == 0.0 && box_matrix[5] == 0.0 &&
                    box_matrix[0] > 0.0 && box_matrix[4] > 0.0 && box_matrix[8] > 0.0 && std::getenv("MGPU_TRI_FULL_SEARCH") == nullptr) ? 1 : 0;
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
        // a row's tasks a run of consecutive |kz| (a k list cut by |k|^2: always; checked because the matrix-unit form of the
        // row sweep addresses task = first task of the row + (kz - first kz))
        e->rows_contiguous = !rtasks.empty() && e->kmax[2] < 255;
        for (size_t ti = 1; ti < rtasks.size(); ++ti)
            if (rtasks[ti].row == rtasks[ti - 1].row && rtasks[ti].j != rtasks[ti - 1].j + 1) e->rows_contiguous = false;
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
        std::vector<int> row_first(rrows.size() + 1, (int)rtasks.size());
        for (int ti = (int)rtasks.size() - 1; ti >= 0; --ti) row_first[rtasks[ti].row] = ti;
        HIP_TRY_E(hipMalloc(&e->d_row_first, row_first.size() * sizeof(int)));
        HIP_TRY_E(hipMemcpy(e->d_row_first, row_first.data(), row_first.size() * sizeof(int), hipMemcpyHostToDevice));
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
                    (void *)e->d_tw, (void *)e->d_kslot, (void *)e->d_rrows, (void *)e->d_row_first, (void *)e->d_atom_ty, (void *)e->d_com, (void *)e->d_off})
        if (p) (void)hipFree(p);
    e->h_stage.release();
    for (void *p : {(void *)e->chain.h_out, (void *)e->chain.h_tag})
        if (p) (void)hipHostFree(p);
    for (void *p : {(void *)e->chain.d_res, (void *)e->chain.d_part, (void *)e->chain.d_ticket, (void *)e->chain.d_topo, (void *)e->chain.d_alt,
                    (void *)e->farm.d_A_alt, (void *)e->farm.d_acur, (void *)e->farm.d_stalled})
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
    // ComputeTotalIntraResidueCoulombEnergy visits ACTIVE residue types only (energy_utils.f90:67-69): the intra items are
    // the active types' molecules, appended behind the pair items (an inactive 2208-atom framework is 2.4 M erfc terms)
    int n_in = 0;
    for (int t = 0; t < tp.n_res; ++t) {
        if (e->is_active[t] != 1) continue;
        for (int m = 0; m < e->h_nmol[replica * tp.n_res + t]; ++m, ++n_in) items.push_back(PairItem{replica, t, m, -1, 0});
    }
    double e_nc = 0.0, e_c = 0.0, e_intra = 0.0, e_recip = 0.0;
    std::vector<double> h_lj(n), h_c(n), h_intra(n_in);
    if ((rc = e->d_out.reserve(((size_t)2 * n + n_in + 1) * sizeof(double)))) return rc;
    if (n > 0) {
        if ((rc = e->d_items.reserve((size_t)(n + n_in) * sizeof(PairItem)))) return rc;
        HIP_TRY(hipMemcpyAsync(e->d_items.p, items.data(), (size_t)(n + n_in) * sizeof(PairItem), hipMemcpyHostToDevice, e->stream));
        double *d_lj = (double *)e->d_out.p, *d_c = d_lj + n, *d_in = d_c + n;
        const int nsplit = e->pair_nsplit;
        if ((rc = launch_pair(e, e->lanes[0], (const PairItem *)e->d_items.p, n, 0, 1, nsplit, d_lj, d_c, true))) return rc;
        if (n_in > 0) {
            if ((rc = launch_intra(e, e->lanes[0], (const PairItem *)e->d_items.p + n, n_in, nullptr, 1, d_in))) return rc;
            HIP_TRY(hipMemcpyAsync(h_intra.data(), d_in, n_in * sizeof(double), hipMemcpyDeviceToHost, e->stream));
        }
        HIP_TRY(hipMemcpyAsync(h_lj.data(), d_lj, n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipMemcpyAsync(h_c.data(), d_c, n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    }
    // ComputeEwaldRecip (energy_utils.f90:270-286): S(k) into scratch, then sum ff W |S|^2
    if ((rc = launch_sfactor(e, replica, e->d_S))) return rc;
    RecipItem rit{0, 0, -1, MGPU_NONE, -1, 0};
    if ((rc = e->d_items2.reserve(sizeof(RecipItem)))) return rc;
    HIP_TRY(hipMemcpyAsync(e->d_items2.p, &rit, sizeof(RecipItem), hipMemcpyHostToDevice, e->stream));
    double *d_u = (double *)e->d_out.p + (size_t)2 * n + n_in;
    if ((rc = launch_recip(e, e->lanes[0], (const RecipItem *)e->d_items2.p, 1, 1, 1, false, e->d_S, d_u))) return rc;
    HIP_TRY(hipMemcpyAsync(&e_recip, d_u, sizeof(double), hipMemcpyDeviceToHost, e->stream));
    if ((rc = sync_stream(e))) return rc;
    int i = 0, ii = 0;
    for (int t = 0; t < tp.n_res; ++t)
        for (int m = 0; m < e->h_nmol[replica * tp.n_res + t]; ++m, ++i) {
            e_nc = e_nc + h_lj[i];
            e_c = e_c + h_c[i];
            // ComputeTotalIntraResidueCoulombEnergy (energy_utils.f90:55-81): active types only
            if (e->is_active[t] == 1) e_intra = e_intra + h_intra[ii++];
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
