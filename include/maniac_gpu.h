/*
 * maniac_gpu.h -- C ABI of the MI355X (gfx950) GCMC energy engine.
 *
 * This is the drop-in boundary for MANIAC's per-move energy path.  The reference has no FFI:
 * the path sits behind Fortran module procedures called from ComputeOldEnergy / ComputeNewEnergy
 * (/root/reference/src/monte_carlo_utils.f90:275-395), the move drivers' save/restore calls and
 * main.f90:27.  Every entry point below names the reference procedure(s) it replaces; the
 * ISO_C_BINDING interface module a MANIAC maintainer would add is in INTEGRATION.md and
 * maniac_mc_amd/fortran/maniac_gpu.f90.
 *
 * Conventions
 *   - plain C types, caller-owned host buffers, engine-owned device buffers, no hidden globals:
 *     any number of engines (one per GPU x force field x box) may coexist in a process;
 *   - every function returns an int status (MGPU_OK = 0) instead of the reference's `stop`
 *     (output_utils.f90:535-562); mgpu_last_error() holds the message of the calling thread;
 *   - indices are 0-based (the reference is 1-based): residue type t, molecule slot m, site a;
 *   - energies are in Kelvin (E / k_B), lengths in Angstrom, charges in e, as in the reference;
 *   - an engine holds `n_replicas` independent configurations ("replicas": separate GCMC chains
 *     sharing box, force field and k-vector table), so that many trial moves -- one or more per
 *     replica -- are evaluated by one kernel launch;
 *   - site coordinates are ABSOLUTE positions com + offset, i.e. the sum the reference forms
 *     before every use (geometry_utils.f90:379-382, ewald_phase.f90:398-399);
 *   - the engine fails loudly (MGPU_ERR_NO_DEVICE / MGPU_ERR_HIP) when no gfx950 device or
 *     kernel image is available; there is no CPU fallback.
 */
#ifndef MANIAC_GPU_H
#define MANIAC_GPU_H

#ifdef __cplusplus
extern "C" {
#endif

#define MGPU_OK 0
#define MGPU_ERR_INVALID_ARG 1
#define MGPU_ERR_HIP 2
#define MGPU_ERR_CAPACITY 3
#define MGPU_ERR_NO_DEVICE 4
#define MGPU_ERR_STATE 5

/* Environment switches, all read ONCE in mgpu_engine_create, all for tests / A-B runs only, each one exercised by a test
 * (defaults are the measured best; none changes results beyond the last bits of a sum order):
 *   MGPU_PAIR_NSPLIT=<n>        waves per pair-sweep item (default: engine constant from the topology's capacity
 *                               and the replica count, see engine_nsplit in mgpu_engine.hip)
 *   MGPU_RECIP_PER_K=1          per-k reciprocal kernel even where a row form applies
 *   MGPU_PAIR_EXACT_FOLD=1      always the multiply / round / fma minimum-image fold in the pair sweep (default: the
 *                               two-instruction fold min(|d|, L - |d|) whenever every resident atom of the replicas in
 *                               a launch and every candidate site of it lies within 0.745 box lengths of the cell
 *                               centre, which the engine tracks on the host; both folds see the same raw separation
 *                               and return the same bits)
 *   MGPU_PAIR_FLAT=0 / 1        plane-by-plane (pair_sweep_kernel) / flat (pair_flat_kernel) register-site pair sweep
 *                               (default: flat for orthorhombic boxes with a frozen framework -- an inactive residue type
 *                               of 64 sites or more -- and at most 64 planes; plane by plane otherwise)
 *   MGPU_NO_FROZEN_BATCH=1      framework boxes keep one wave per candidate (pair_flat_kernel) instead of 64 candidates
 *                               per wave against chunks of the framework (pair_frozen_kernel; default where exactly one
 *                               frozen residue type exists and it is identical in every replica)
 *   MGPU_RECIP_NO_MFMA=1        molecules of a few dozen sites (wide row form of the k sweep): the vector form (rows a tile
 *                               at a time through an XY table in LDS) instead of the matrix-unit form (tiles of 16 rows x
 *                               16 kz through v_mfma_f64_16x16x4_f64; another order of the same sum over site-states)
 *   MGPU_TRI_FULL_SEARCH=1      triclinic boxes: always ComputeDistance's full 27-image search (default: for the reader's
 *                               lower-triangular cells the same minimum in eight evaluations, the full search only
 *                               where its certificate fails; bit-identical results)
 * Threading rule: the per-lane asynchronous entry points (mgpu_*_submit, mgpu_*_wait, mgpu_commit_submit,
 * mgpu_lane_site_buffer) may be called concurrently from different host threads on DIFFERENT lanes -- one thread per
 * lane at a time (the farm runs up to three driver threads that way); lanes must hold disjoint replicas while their
 * trials / commits are in flight.  Everything else is single-threaded: engine creation / destruction, the
 * mgpu_replica_* / structure-factor / system-energy calls, the synchronous candidate calls (mgpu_pair / recip / intra /
 * trial_energy_candidates, mgpu_commit_candidates), mgpu_synchronize and the profile calls must not run while another
 * thread is inside any engine call; those that read or rewrite replica state drain all lanes first. */

/* candidate kinds for the reciprocal-space update, ewald_energy.f90:241-256 */
#define MGPU_MOVE 0      /* A += sum q (phi_new - phi_old)   translation / rotation */
#define MGPU_CREATION 1  /* A += sum q phi_new                insertion              */
#define MGPU_DELETION 2  /* A -= sum q phi_old                deletion               */
#define MGPU_NONE 3      /* A unchanged: E = sum ff W |A|^2   (ComputeOldEnergy's recip call, where
                            the new tables equal the saved ones, monte_carlo_utils.f90:388)    */
#define MGPU_FOURIER_ADD 4 /* mgpu_structure_factor_add only: A += sum q phi(sites), nothing else changes */

/* which kernel mgpu_profile_get() reports */
#define MGPU_KERNEL_PAIR 0
#define MGPU_KERNEL_RECIP 1
#define MGPU_KERNEL_COMMIT 2
#define MGPU_KERNEL_SFACTOR 3
#define MGPU_KERNEL_COUNT 4

typedef struct mgpu_engine mgpu_engine;

/* Message of the last failing call made by this thread ("" if none). */
const char *mgpu_last_error(void);

/* Library ABI version (bumped on any signature change; a binding compares mgpu_abi_version() with the MGPU_ABI_VERSION
 * it was built against).  2: chain windows, farm windows, comm, host team, formatter and phase-factor exports; the
 * reciprocal update takes molecules of any size. */
#define MGPU_ABI_VERSION 2
int mgpu_abi_version(void);

/* Number of visible HIP devices; MGPU_ERR_NO_DEVICE if the runtime reports none. */
int mgpu_device_count(int *count);

/* ------------------------------------------------------------------------------------------
 * Host-side setup arithmetic (no GPU needed).  Replaces the setup half of
 * PrepareSimulationParameters (prepare_utils.f90:19-41) and PrepareSimulationBox
 * (geometry_utils.f90:20-57).
 * ---------------------------------------------------------------------------------------- */

/* DetermineBoxSymmetry + ComputeCellProperties + ComputeInverse (geometry_utils.f90:68-154,
 * :277-331).  box_matrix[i*3+j] = box%matrix(i+1,j+1); reciprocal likewise.
 * box_type: 1 cubic, 2 orthorhombic, 3 triclinic. */
int mgpu_box_prepare(const double box_matrix[9], int *box_type, double *volume,
                     double reciprocal[9], double metrics[9]);

/* SetupEwald (prepare_utils.f90:103-214): cutoff adjustment, tolerance clamp, alpha, kmax and the
 * k-vector count.  rc and tol are in/out exactly as the reference mutates input%real_space_cutoff
 * and input%ewald_tolerance. */
int mgpu_ewald_setup(const double metrics[9], double *rc, double *tol, double *alpha,
                     double *screening_factor, double *fourier_precision, int kmax[3],
                     int *n_kvectors);

/* PrecomputeValidReciprocalVectors (ewald_kvectors.f90:44-87) + ComputeReciprocalWeights
 * (ewald_kvectors.f90:225-246): the half-space k list in the reference's loop order, |k|^2,
 * form factor (1 if kx = 0 else 2) and W(k) = exp(-k^2 / 4 alpha^2) / k^2. */
int mgpu_ewald_kvectors(const double reciprocal[9], double alpha, const int kmax[3], int n_kvectors,
                        int *kx, int *ky, int *kz, double *k2mag, double *form_factor,
                        double *weights);

/* The pair sweep does not call libm's erfc (energy_utils.f90:432): it reads G(r^2) =
 * erfc(alpha r)/r from an LDS-resident table of degree-6 polynomials indexed by the binary exponent
 * and top 6 mantissa bits of r^2 (built in long double at engine creation for the engine's alpha and
 * box).  This evaluates the same table with the same arithmetic on the host, for accuracy checks:
 * out[i] = erfc(alpha sqrt(r2[i])) / sqrt(r2[i]) for r2[i] > 0; r2_max bounds the table range. */
int mgpu_coulomb_table_eval(double alpha, double r2_max, int n, const double *r2, double *out);

/* Seeds for a farm of independent chains: state[4 * n_streams] receives one xoshiro256+ state per chain, each
 * filled from its own splitmix64 stream (Blackman & Vigna's recommended seeding; all-zero states cannot occur),
 * started at mix(seed) + r * odd constant -- unsigned 64-bit arithmetic, which Fortran does not have.  The
 * reference seeds ONE intrinsic generator with seed + 37 (i - 1) (random_utils.f90:33-56); a farm needs R
 * statistically independent streams instead (host utility, no device involved). */
int mgpu_rng_seed_streams(long long seed, int n_streams, long long *state);
/* The next n_per uniform numbers in [0, 1) of each of n_streams CONSECUTIVE streams (state = the first stream's four
 * words; out[n_streams][n_per]): xoshiro256+, the top 53 bits of s0 + s3 times 2^-53 -- the numbers of mc_farm.f90's
 * chain_random, bit for bit, four streams abreast (AVX2 where the CPU has it).  A farm draws ten numbers per chain and
 * step (src/random_utils.f90:13-56 is the reference's one serial generator); generated one stream at a time that was a
 * quarter of a grand-canonical farm's host time. */
int mgpu_rng_fill(long long *state, int n_streams, int n_per, double *out);
/* Software prefetch of `bytes` bytes at p into the calling core's caches (host utility for Fortran callers, which
 * have no prefetch intrinsic: the farm driver announces the mirror records it is about to gather). */
void mgpu_host_prefetch(const void *p, int bytes);

/* ------------------------------------------------------------------------------------------
 * Engine life cycle
 * ---------------------------------------------------------------------------------------- */

/* Create an engine on HIP device `device` holding `n_replicas` configurations.
 *   atoms_in_res[n_res]            nb%atom_in_residue            (simulation_state.f90:122)
 *   mol_capacity[n_res]            molecule slots per replica    (reference: NB_MAX_MOLECULE)
 *   atom_types[n_res*max_atom]     primary%atom_types(t,a), 1-based atom type ids
 *   charges[n_res*max_atom]        primary%atom_charges(t,a)
 *   is_active[n_res]               input%is_active
 *   epsilon/sigma[n_types*n_types] per ATOM-TYPE pair, Kelvin / Angstrom -- the reference's 4-D
 *                                  coeff%epsilon/sigma are functions of the two atom types only
 *                                  (parameters_parser.f90:89-98, :141-176)
 *   box_matrix[9], bounds_lo[3]    box%matrix (row-major image), box%bounds(:,1)
 *   real_space_cutoff, ewald_tolerance  as read from the .maniac file; SetupEwald is applied.
 * Triclinic boxes (box_type 3) use the reference's 27-image distance search (geometry_utils.f90:397-411)
 * and take the generic sweep; cubic / orthorhombic boxes take the tuned one. */
int mgpu_engine_create(mgpu_engine **out, int device, int n_replicas, int n_res,
                       const int *atoms_in_res, const int *mol_capacity, int max_atom,
                       const int *atom_types, const double *charges, const int *is_active,
                       int n_types, const double *epsilon, const double *sigma,
                       const double box_matrix[9], const double bounds_lo[3],
                       double real_space_cutoff, double ewald_tolerance);
int mgpu_engine_destroy(mgpu_engine *e);

/* Scalars fixed at creation: what LogEwaldParameters prints (prepare_utils.f90:75-97). */
int mgpu_engine_get_ewald(const mgpu_engine *e, double *alpha, double *real_space_cutoff,
                          double *ewald_tolerance, int kmax[3], int *n_kvectors, double *volume,
                          int *box_type);
int mgpu_engine_get_kvectors(const mgpu_engine *e, int *kx, int *ky, int *kz, double *k2mag,
                             double *form_factor, double *weights);

/* ------------------------------------------------------------------------------------------
 * Replica state (replaces the module-global primary%mol_com / site_offset / num_residues)
 * ---------------------------------------------------------------------------------------- */

/* Load all molecules of residue type t of one replica.  sites[n_mol][atoms_in_res[t]][3].
 * Coordinates need not lie inside the primary cell (energies use minimum images, as ComputeDistance does).  The engine
 * notes whether every site it is ever given for a (replica, type) -- here or as a committed candidate -- lies within
 * 0.99 box lengths of the cell centre on every axis; while that holds for all replicas of a launch (it always does for
 * molecules whose centres of mass are kept in the cell, as ApplyPBC keeps them) the pair sweep uses a cheaper, bit-identical
 * form of the minimum-image fold; otherwise it takes the general one.  Results do not depend on which. */
int mgpu_replica_set_molecules(mgpu_engine *e, int replica, int t, int n_mol, const double *sites);
int mgpu_replica_get_molecules(mgpu_engine *e, int replica, int t, int *n_mol, double *sites);
int mgpu_replica_num_molecules(const mgpu_engine *e, int replica, int t, int *n_mol);
/* Copy a whole configuration (coordinates, counts, A(k)) from replica src to replica dst. */
int mgpu_replica_copy(mgpu_engine *e, int dst, int src);

/* ------------------------------------------------------------------------------------------
 * Static energies
 * ---------------------------------------------------------------------------------------- */

/* ComputeSystemEnergy (energy_utils.f90:18-35).  out[6] = non_coulomb, coulomb, recip_coulomb,
 * ewald_self, intra_coulomb, total.  Like the reference it does NOT store S(k) into A(k). */
int mgpu_system_energy(mgpu_engine *e, int replica, double out[6]);

/* The initialisation the reference omits: mode 1 sets A(k) to the full structure
 * factor S(k) = sum_j q_j exp(i k.r_j) (ComputeRecipAmplitude, ewald_energy.f90:40-77);
 * mode 0 zeroes it (for as-written comparisons). */
int mgpu_init_structure_factor(mgpu_engine *e, int replica, int mode);
/* a[n_kvectors][2] = (re, im) of ewald%recip_amplitude */
int mgpu_get_structure_factor(mgpu_engine *e, int replica, double *a);
int mgpu_set_structure_factor(mgpu_engine *e, int replica, const double *a);
/* A(k) <- A(k) + sum_a q_a exp(i k . sites_a) for ONE molecule of residue type t given by its site coordinates
 * (the "creation" line of ComputeRecipEnergySingleMol, ewald_energy.f90:241-246) -- coordinates, molecule counts
 * and everything else stay as they are.  A neutral building block: a host loop that wants the reference's
 * deletion update exactly as written (monte_carlo_utils.f90:308 passes is_creation = deletion_flag, so A gains
 * the swapped-in molecule's terms) composes it from mgpu_replica_replace_molecule, mgpu_replica_set_num_molecules
 * and this call; no kernel knows about that mode.  sites: [atoms_in_res[t]][3]. */
int mgpu_structure_factor_add(mgpu_engine *e, int replica, int t, const double *sites);

/* ------------------------------------------------------------------------------------------
 * Batched per-move energies.  A candidate c is (replica[c], t[c], m[c], sites[c]):
 *   m[c] >= 0  the molecule slot being moved / deleted; it is excluded from the pair sweep
 *              (energy_utils.f90:408-409) and its device-resident sites are the "old" state;
 *   m[c] = -1  no such molecule (insertion into an empty slot): nothing is excluded;
 *   sites      candidate site coordinates, [n_candidates][site_stride][3]; row c uses the first
 *              atoms_in_res[t[c]] entries; may be NULL when no candidate needs it.
 * None of the evaluate calls mutates engine state.
 * ---------------------------------------------------------------------------------------- */

/* ComputePairInteractionEnergy_singlemol (energy_utils.f90:374-442) for B candidates.
 * use_resident[c] != 0: evaluate the molecule where it currently is (sites ignored). */
int mgpu_pair_energy_candidates(mgpu_engine *e, int n_candidates, const int *replica, const int *t,
                                const int *m, const int *use_resident, const double *sites,
                                int site_stride, double *e_non_coulomb, double *e_coulomb);

/* SingleMolFourierTerms + ComputeRecipEnergySingleMol (ewald_phase.f90:383-420,
 * ewald_energy.f90:191-274) for B candidates, WITHOUT mutating A(k):
 * u[c] = prefactor * sum_k ff W |A + delta_c|^2 with delta by kind[c] (MGPU_MOVE/...). */
int mgpu_recip_energy_candidates(mgpu_engine *e, int n_candidates, const int *replica, const int *t,
                                 const int *m, const int *kind, const double *sites,
                                 int site_stride, double *u_recip);

/* ComputeEwaldSelfInteractionSingleMol (ewald_energy.f90:308-336); configuration independent. */
int mgpu_self_energy(const mgpu_engine *e, int t, double *e_self);

/* ComputeIntraResidueRealCoulombEnergySingleMol (ewald_energy.f90:371-411) for B candidates
 * (use_resident as above). */
int mgpu_intra_energy_candidates(mgpu_engine *e, int n_candidates, const int *replica, const int *t,
                                 const int *m, const int *use_resident, const double *sites,
                                 int site_stride, double *u_intra);

/* One translation / rotation trial per candidate, both halves in one call: what
 * ComputeOldEnergy + ComputeNewEnergy compute for the default branch
 * (monte_carlo_utils.f90:384-393, :310-318).  old_/new_ arrays are [n_candidates][3] =
 * non_coulomb, coulomb, recip_coulomb.  Every candidate costs two pair sweeps and one pass over k
 * that yields both reciprocal energies (sum ff W |A|^2 and sum ff W |A + delta|^2). */
int mgpu_trial_energy_candidates(mgpu_engine *e, int n_candidates, const int *replica, const int *t,
                                 const int *m, const double *sites, int site_stride,
                                 double *old_energy, double *new_energy);

/* Apply accepted candidates (accept[c] != 0), at most one per replica per call:
 *   MGPU_MOVE      A += delta, slot m <- sites                     (AcceptMove keeps the mutated
 *                  state, monte_carlo_utils.f90:410-422)
 *   MGPU_CREATION  A += delta, new molecule appended at slot n_mol, n_mol += 1
 *                  (create_molecule.f90:64-74)
 *   MGPU_DELETION  A -= old, last molecule copied into slot m, n_mol -= 1
 *                  (delete_molecule.f90:67-74, :99-116; intended physics, not defect F3)
 * Rejected candidates need no call at all: evaluation never mutated anything, which replaces
 * Save/RestoreSingleMolFourier (ewald_phase.f90:134-255). */
int mgpu_commit_candidates(mgpu_engine *e, int n_candidates, const int *replica, const int *t,
                           const int *m, const int *kind, const double *sites, int site_stride,
                           const int *accept);

/* Asynchronous forms of the two calls above, so that the host's Metropolis work on one group of
 * replicas overlaps the GPU's evaluation of another.  The engine has MGPU_LANES submission lanes,
 * each a HIP stream with private scratch; work on different lanes may run concurrently and must
 * touch disjoint replicas.  submit returns once copies and kernels are queued; wait blocks for the
 * lane and returns the energies; a commit is ordered before any later submit on the same lane.
 * mgpu_commit_submit accepts sites = NULL when it commits the candidates of the lane's last
 * mgpu_trial_submit (same n_candidates, order and site_stride): their rows are still on the device. */
#define MGPU_LANES 4
/* Molecule frames -- what the reference keeps per molecule: com[n_mol][3] = primary%mol_com and off[n_mol][n1][3] =
 * primary%site_offset (src/simulation_state.f90:115-116).  mgpu_replica_set_frames uploads them together with the sites
 * com + off (formed as src/geometry_utils.f90:379-382 forms them before every use) and keeps them resident, so that trial
 * moves can be BUILT on the device: a host then needs neither a mirror of the coordinates nor to stage candidate rows.
 * mgpu_replica_set_molecules (sites only) drops a residue type's frames again. */
int mgpu_replica_set_frames(mgpu_engine *e, int replica, int t, int n_mol, const double *com, const double *off);
int mgpu_replica_get_frames(mgpu_engine *e, int replica, int t, int *n_mol, double *com, double *off);
/* One trial per candidate, its geometry built on the device from the resident frames (orthorhombic boxes):
 *   move[c] = 1  Translation      com <- ApplyPBC(com + (u[c][0..2] - 1/2) translation_step)      src/translation.f90:93-112
 *           = 2  Rotation         offsets rotated by (u[c][3] - 1/2) rotation_step about Cartesian axis int(3 u[c][4]) + 1
 *                                                                                                src/monte_carlo_utils.f90:30-92
 *           = 3  CreateMolecule   com <- lo + L u[c][0..2]; offsets of molecule 1 of the type rotated by 2 pi u[c][3]
 *                                 about that axis; m ignored                                       src/create_molecule.f90:166-207
 *           = 4  DeleteMolecule   the resident molecule m as it is
 * u = n x 5 uniform numbers in [0, 1) drawn by the host (the random stream stays on the host, as the acceptance does).
 * Energies come back through mgpu_gcmc_trial_wait; mgpu_commit_submit(sites = NULL) on the same lane applies the accepted
 * candidates from the rows the device built, sites and frames. */
int mgpu_move_trial_submit(mgpu_engine *e, int lane, int n_candidates, const int *replica, const int *t, const int *m,
                           const int *move, const double *u, double translation_step, double rotation_step);
/* The same trials with the ACCEPTANCE TEST ON THE DEVICE.  The k sweep is the last kernel of a trial, one workgroup per
 * candidate; once it has summed the candidate's reciprocal energies its first thread holds everything
 * mc_acceptance_probability (src/monte_carlo_utils.f90:184-226) needs, so it applies the rule itself,
 *     accept  <=>  accept_u[c] <= min(1, accept_pref[c] * exp(-(new%total - old%total) / temperature)),
 * with old%total / new%total the sums of the five components in the order the wait calls return them, and the workgroup of
 * an accepted candidate commits it from the phase tables it already holds (A <- A + delta in a second pass over the
 * replica's A(k), then coordinates / frames / count: AcceptMove, src/monte_carlo_utils.f90:410-422; create_molecule.f90:
 * 100-112; delete_molecule.f90:100-142).  The caller supplies the test's uniform number and prefactor per candidate
 * (1 for translations / rotations; phi V / (N + 1) for an insertion, N / (phi V) for a deletion -- create_molecule.f90:64,
 * delete_molecule.f90:73) and learns the outcome from mgpu_trial_decide_wait: accepted[c] = 0 / 1 beside the energies.
 * There is no commit call, no second upload and no second kernel; the engine's host-side counts follow when the lane is
 * waited for (or drained by any synchronous entry point).  One candidate per replica per launch; row-form k sweep
 * (molecules of a few sites).  The committed state is bitwise that of mgpu_commit_submit with the same flags.
 * mgpu_gcmc_trial_decide_submit takes candidate rows from the host (replicas without resident frames). */
int mgpu_move_trial_decide_submit(mgpu_engine *e, int lane, int n_candidates, const int *replica, const int *t, const int *m,
                                  const int *move, const double *u, double translation_step, double rotation_step,
                                  const double *accept_u, const double *accept_pref, double temperature);
int mgpu_gcmc_trial_decide_submit(mgpu_engine *e, int lane, int n_candidates, const int *replica, const int *t, const int *m,
                                  const int *kind, const double *sites, int site_stride, const double *accept_u,
                                  const double *accept_pref, double temperature);
int mgpu_trial_decide_wait(mgpu_engine *e, int lane, double *old_energy, double *new_energy, int *accepted);
/* Pinned staging of a lane's NEXT trial, sized for n_max candidates of site_stride sites: a host that builds its
 * candidate rows directly in *sites and then passes that same pointer as `sites` to mgpu_trial_submit /
 * mgpu_gcmc_trial_submit on this lane (with at most n_max candidates and the same site_stride) saves the engine's copy
 * of the rows into its staging block (measured: 590 KB per lane step at 8192 CO2 candidates).  The pointer stays
 * valid until the engine is destroyed or the function is called again for the lane with a larger size -- the engine
 * never regrows a lent block on its own: a later trial on the lane that would not fit it (more candidates, a larger
 * stride) fails with MGPU_ERR_STATE instead.  The block is sized for n_max candidates in every submit form (host rows,
 * device-built rows, acceptance records).  The rows may be rewritten once the lane's trial has been waited for.  Replaces nothing in the reference (its candidates live in
 * primary%mol_com / site_offset, src/simulation_state.f90:115-116). */
int mgpu_lane_site_buffer(mgpu_engine *e, int lane, int n_max, int site_stride, double **sites);
/* Host threads the per-candidate loops INSIDE mgpu_*_submit, mgpu_*_wait and mgpu_commit_submit may use (default 1: the
 * calling thread alone).  With n_threads > 1 a call with >= 1024 candidates cuts its loops into n_threads contiguous
 * ranges run by an OpenMP team of the calling thread (LLVM's libomp, the runtime of the Fortran drivers: a driver's team
 * is reused); items, results and the first error reported are those of the serial loop
 * (tests/test_gpu_farm.py::test_host_team_is_the_serial_loop).  Set it to the size of the team each driver thread owns
 * (mc_farm.f90 does).  Replaces nothing in the reference (its loop handles one candidate per step). */
int mgpu_set_host_team(mgpu_engine *e, int n_threads);
int mgpu_trial_submit(mgpu_engine *e, int lane, int n_candidates, const int *replica, const int *t,
                      const int *m, const double *sites, int site_stride);
int mgpu_trial_wait(mgpu_engine *e, int lane, double *old_energy, double *new_energy);
int mgpu_commit_submit(mgpu_engine *e, int lane, int n_candidates, const int *replica, const int *t,
                       const int *m, const int *kind, const double *sites, int site_stride,
                       const int *accept);

/* Mixed batches for grand-canonical chains: kind[c] in {MGPU_MOVE, MGPU_CREATION, MGPU_DELETION}.
 * old_/new_ arrays are [n_candidates][5] = non_coulomb, coulomb, recip_coulomb, ewald_self,
 * intra_coulomb, filled exactly as ComputeOldEnergy / ComputeNewEnergy fill their energy_state
 * (monte_carlo_utils.f90:292-318, :364-393), except that a deletion's new recip_coulomb is the
 * intended sum ff W |A - S_mol|^2.  m[c] is ignored for creations (the new molecule is
 * appended by the commit).  Commit with mgpu_commit_submit (sites = NULL reuses the rows). */
int mgpu_gcmc_trial_submit(mgpu_engine *e, int lane, int n_candidates, const int *replica, const int *t,
                           const int *m, const int *kind, const double *sites, int site_stride);
int mgpu_gcmc_trial_wait(mgpu_engine *e, int lane, double *old_energy, double *new_energy);

/* ReplaceFourierTermsSingleMol + the coordinate copy of RemoveMolecule (ewald_phase.f90:276-322,
 * delete_molecule.f90:99-116): slot m_dst <- slot m_src. */
int mgpu_replica_replace_molecule(mgpu_engine *e, int replica, int t, int m_dst, int m_src);
/* Overwrite the molecule count of a replica's residue type (no coordinates touched). */
int mgpu_replica_set_num_molecules(mgpu_engine *e, int replica, int t, int n_mol);

/* ------------------------------------------------------------------------------------------
 * Single-chain windows: ONE kernel launch per window of trial steps of one Markov chain
 * ---------------------------------------------------------------------------------------- */

/* MonteCarloLoop (src/monte_carlo.f90:40-86) advances ONE chain one step at a time, and one step at a time a GPU is a
 * latency chain (upload, launches, download, synchronise, commit launch: 50-65 us for ~10 us of arithmetic).  A host that
 * draws the next n steps in the reference's random-number order ASSUMING every one of them is rejected holds n trials of
 * the SAME state (mc_chain.f90, run_window); mgpu_chain_window evaluates all of them in one launch, applies
 * mc_acceptance_probability (src/monte_carlo_utils.f90:184-226) to them IN ORDER with the host's own acceptance draws
 * accept_u[c] and prefactors accept_pref[c] (1 for a translation / rotation; phi V / N for an insertion, N = the count
 * after it; (N + 1) / (phi V) for a deletion, N = the count after it -- create_molecule.f90:64, delete_molecule.f90:73),
 * and commits the FIRST accepted step on the device (A(k) += delta, coordinates, count -- AcceptMove and the Accept* of
 * create_molecule.f90:100-112 / delete_molecule.f90:100-142).  The window travels in the kernel's arguments and the
 * kernel writes its results straight into pinned host memory; the call returns as soon as the results are there, while
 * the commit still runs (every later call is ordered behind it).
 *   kind[c]     MGPU_MOVE / MGPU_CREATION / MGPU_DELETION; m[c] ignored for creations (appended)
 *   link[c]     -1: an ordinary step.  >= 0 (deletions only): the reference's deletion AS WRITTEN (SURVEY F3,
 *               monte_carlo_utils.f90:301-309): the step's new reciprocal energy is the creation-kind energy of row
 *               link[c] -- the molecule RemoveMolecule moves into the slot -- and accepting it adds that molecule's terms
 *               to A(k).  -2: such an energy-only companion row (creation kind; never decided, never committed).
 *   sites       [n][site_stride][3] candidate sites (unused for deletions)
 *   recip_energy  the chain's running energy%recip_coulomb: ComputeOldEnergy's value for insertions / deletions
 *               (monte_carlo_utils.f90:366-372)
 *   old_energy / new_energy   [n][5] as mgpu_gcmc_trial_wait returns them (every row, decided or not)
 *   first_accepted   index of the step the device accepted and committed, -1: none
 *   undecided   -1, or the index of a step the device left to the host: its draw lies within the engine's relative
 *               margin (16 ulp; mgpu_chain_set_margin) of its acceptance probability, or the probability is not a number.
 *               The device's exp (OCML) and the host's (glibc) may differ in the last bits there; the device stops at
 *               such a step -- nothing at or after it is committed -- and the host decides it with its own exp from the
 *               energies returned.  Every decision the device takes is therefore the host's decision.
 * mgpu_chain_window_capacity: the largest n (0: the one-launch path does not apply -- per-k reciprocal
 * form, active molecules of more than 5 sites -- use mgpu_gcmc_trial_submit / wait).  Lane 0's stream carries the launch:
 * no trial may be in flight on lane 0. */
int mgpu_chain_window_capacity(const mgpu_engine *e, int *max_candidates);
int mgpu_chain_window(mgpu_engine *e, int replica, int n, const int *t, const int *m, const int *kind, const int *link,
                      const double *sites, int site_stride, const double *accept_u, const double *accept_pref,
                      double temperature, double recip_energy, double *old_energy, double *new_energy,
                      int *first_accepted, int *undecided);
int mgpu_chain_set_margin(mgpu_engine *e, double relative_margin);
int mgpu_chain_get_stats(const mgpu_engine *e, long long *windows, long long *undecided);
/* Stage times of a window, measured inside the kernel (the launch IS the step: there is nothing between its stages for a
 * host-side tracer to see).  set_timing(1): later windows record the device's 100 MHz wall clock at their stages;
 * get_timing: the last window's stages in microseconds since its first workgroup started --
 *   us[0..3]   k role of candidate 0: start, phase tables built, k sweep summed, at the ticket
 *   us[4..7]   first pair workgroup: start, Coulomb table staged, its work units swept, at the ticket
 *   us[8..14]  resolving workgroup: last ticket drawn, acquire fence passed, partials reduced, decided, tag published,
 *              commit tables built, commit done (the last two 0 when the window accepted nothing). */
int mgpu_chain_set_timing(mgpu_engine *e, int on);
int mgpu_chain_get_timing(mgpu_engine *e, double us[15]);

/* ------------------------------------------------------------------------------------------
 * Farm windows: ONE launch per lane step of a FARM of chains (few chains per GPU).
 * Replaces, for every chain of a lane at once, one pass of MonteCarloLoop's body (src/monte_carlo.f90:40-86): the move
 * drivers' geometry (src/translation.f90:93-112, src/monte_carlo_utils.f90:30-92, src/create_molecule.f90:166-207),
 * ComputeOldEnergy / ComputeNewEnergy (src/monte_carlo_utils.f90:275-395), mc_acceptance_probability (:184-226) and
 * AcceptMove / the Accept* of create_molecule.f90:100-112, delete_molecule.f90:100-142.
 * Through the batched path (mgpu_move_trial_submit / wait, the rule in the driver, mgpu_commit_submit) a lock step of a
 * farm is five launches, two copies and two host round trips: 45-110 us for ~10 us of arithmetic when the farm holds
 * 8-512 chains.  Here the driver hands over, per chain, only the move it selected and the uniform numbers of the move's
 * construction and of its acceptance test; one launch builds the candidates from the resident molecule frames
 * (mgpu_replica_set_frames), evaluates them (the batched path's sums, bit for bit: same kernels' bodies, the engine's
 * nsplit), applies the rule and commits accepted steps; the driver collects energies and verdicts from pinned host
 * memory (per-chain tags: no copy, no stream synchronisation).  A lane may hold up to max_in_flight windows: a driver whose
 * move selection does not depend on earlier outcomes (NVT) queues step i + 1 before it has seen step i.
 *   move[c]     0: chain c does nothing this step; 1 translation, 2 rotation, 3 creation, 4 deletion (m ignored for 3)
 *   forced      NULL, or per chain 0: the device applies the rule; 1 / 2: the driver has decided this step (accept /
 *               reject) and the device obeys
 *   u5          [n][5] the construction's uniform numbers (mgpu_move_trial_submit's)
 *   accept_u / accept_pref   the test's uniform number and prefactor (mgpu_chain_window's)
 *   slot_u      NULL: m[c] is the molecule the driver picked (against counts that must then be current: a window with an
 *               insertion / deletion is collected before the lane's next submit).  Otherwise BY COUNT: m is ignored,
 *               slot_u[c] is the uniform number of PickRandomMoleculeIndex and accept_pref[c] = phi V for insertions and
 *               deletions (1 for moves); the launch completes every record from the replica's molecule count as it is when
 *               it runs -- slot int(u N) of N (nothing to do for a move or deletion of an empty type or an insertion into a
 *               full one: verdict 5), prefactor phi V / (N + 1) or N / (phi V) -- so that insertion / deletion farms, too,
 *               queue a window before they have seen the last (the type and the kind of move do not depend on N); the
 *               driver replays the same arithmetic with its own counts when it collects the window (a by-count record of
 *               a replica that waits for the driver's decision comes back 4 whatever its count says: the step it waits
 *               for may change the count)
 * mgpu_farm_window_wait collects the lane's OLDEST window: old_energy / new_energy [n][5] as mgpu_gcmc_trial_wait fills
 * them, verdict[c] = 0 rejected, 1 accepted and committed, 2 UNDECIDED, 4 nothing done (the replica waits for the driver's
 * decision of an earlier step), 5 idle record.  Undecided: the draw lies within the engine's relative margin (16 ulp;
 * mgpu_chain_set_margin) of the acceptance probability -- the device's exp (OCML) and the driver's (glibc) may differ in
 * the last bits there -- or the probability is not a number: nothing is committed, the replica is marked, every window
 * already queued for it does nothing (verdict 4), and the driver, having decided with its own exp, sends the step again
 * with `forced` set.  The driver checks every other verdict against its own rule: every decision taken is its own.
 * One record per replica and launch; a window with an insertion / deletion must be collected before the lane's next
 * submit (the engine validates slots against its molecule counts).  A(k): a farm window leaves a replica's current A(k) in
 * one of two buffers; every other entry point that touches A(k) copies it back first (mgpu_farm_window_flush does only
 * that).  mgpu_farm_window_capacity: chains per launch (0: the path does not apply -- triclinic box, per-k reciprocal
 * form, an active molecule of more than 5 sites; otherwise min(4096, replicas)) and windows per lane in flight.  Lanes may be
 * driven by different host threads (one thread per lane at a time), as the other per-lane entry points. */
int mgpu_farm_window_capacity(const mgpu_engine *e, int *max_chains, int *max_in_flight);
int mgpu_farm_window_submit(mgpu_engine *e, int lane, int n, const int *replica, const int *t, const int *m, const int *move,
                            const int *forced, const double *u5, const double *accept_u, const double *accept_pref,
                            const double *slot_u, double t_step, double r_step, double temperature);
int mgpu_farm_window_wait(mgpu_engine *e, int lane, double *old_energy, double *new_energy, int *verdict);
int mgpu_farm_window_flush(mgpu_engine *e);
int mgpu_farm_window_get_stats(const mgpu_engine *e, long long *windows, long long *undecided);

/* ------------------------------------------------------------------------------------------
 * The path's one exchange step: replicas are farmed over the GPUs of a node, one process per GPU,
 * and never communicate while they run; once per block every rank contributes its chains' molecule-count histogram
 * (what the reference records per chain in number_<res>.dat, src/write_utils.f90:144-150) and a few running sums, and
 * receives the rank-ordered table: one RCCL all-gather over xGMI (<= 40 KB per rank: latency-bound).
 * ---------------------------------------------------------------------------------------- */
typedef struct mgpu_comm mgpu_comm;
#define MGPU_COMM_ID_BYTES 128
/* Rank 0 makes the id (ncclGetUniqueId) and hands it to the other ranks by whatever means the host has (a file, MPI,
 * a socket; bench.py broadcasts it over its torch.distributed group); every rank then calls mgpu_comm_create with it.
 * world == 1 needs no id (NULL) and never initialises RCCL: its gather is the identity. */
int mgpu_comm_unique_id(void *id128);
int mgpu_comm_create(mgpu_comm **out, int device, int rank, int world, const void *id128);
int mgpu_comm_destroy(mgpu_comm *c);
int mgpu_comm_rank(const mgpu_comm *c, int *rank, int *world);
/* sums_by_rank[world][n_sums] <- every rank's sums[n_sums]; hist_by_rank[world][n_bins] <- every rank's hist[n_bins]
 * (either may be empty).  Blocking; the same call, with the same sizes, on every rank. */
int mgpu_allgather_block_stats(mgpu_comm *c, int n_sums, const double *sums, int n_bins, const long long *hist,
                               double *sums_by_rank, long long *hist_by_rank);

/* ------------------------------------------------------------------------------------------
 * Output surface helper (host only)
 * ---------------------------------------------------------------------------------------- */

/* The atom records of trajectory.lammpstrj (mol == NULL: '(I6,1X,I4,3(1X,F12.7))', WriteLAMMPSTRJ, src/write_utils.f90:86)
 * or of topology.data's Atoms section ('(I6,1X,I6,1X,I4,1X,F12.8,3(1X,F12.7))', WriteLAMMPSData, :297-300) appended to
 * `path`, byte for byte what the Fortran runtime's edit descriptors produce (exact decimal conversion, round half to even,
 * asterisks on overflow) at a tenth of its cost: the single-chain driver writes 2 x 10 125 of them per block.  On any
 * error nothing has been written and the caller writes the records itself.  mgpu_format_fixed: the Fw.d conversion alone
 * (tests). */
int mgpu_append_atom_records(const char *path, int n, int first_serial, const int *mol, const int *type, const double *charge,
                             const double *xyz);
int mgpu_format_fixed(int n, const double *x, int w, int d, char *out);

/* Test and diagnostic hook: exp(i k theta) = (cos, sin)(k * theta) exactly as the device's phase tables hold it -- the
 * reference's ComputePhaseFactors1D, src/ewald_phase.f90:100-109, dcos / dsin of the rounded product.  The device
 * evaluates it with its own bounded-argument routine (|k theta| < 2^30, within 1.5 ulp of the exact value:
 * tests/test_gpu_parity.py::test_phase_factors_are_within_two_ulp).  Host arrays of n entries; synchronous. */
int mgpu_phase_factors(mgpu_engine *e, int n, const double *theta, const int *k, double *cos_out, double *sin_out);

/* ------------------------------------------------------------------------------------------
 * Measurement
 * ---------------------------------------------------------------------------------------- */

/* The engine launches on its own HIP stream.  Block until everything queued so far is done. */
int mgpu_synchronize(mgpu_engine *e);
/* When enabled, every launch of the four main kernels carries a start and a stop HIP event attached to
 * the dispatch on the lane's own stream (the kernel's begin / end timestamps, no extra packets in the
 * stream); mgpu_profile_get returns launches and total device milliseconds since the last reset.
 * Enabling pays the one-time costs up front (event pool, one empty profiled dispatch per lane: the first
 * such dispatch switches the stream's queue into profiling mode, ~7 ms), so call it outside a timed region. */
int mgpu_profile_enable(mgpu_engine *e, int on);
int mgpu_profile_reset(mgpu_engine *e);
int mgpu_profile_get(mgpu_engine *e, int kernel, long long *launches, double *total_ms);

#ifdef __cplusplus
}
#endif
#endif /* MANIAC_GPU_H */
