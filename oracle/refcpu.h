/*
 * TEST INFRASTRUCTURE ONLY (oracle) -- plain-C, fp64, single-thread restatement of the
 * reference's per-move energy hot path.  Never linked into, imported by or called from the
 * product path (maniac_mc_amd/); only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it, and only as the checker / the reported CPU baseline.
 *
 * Parity status: PINNED -- tests/test_oracle_pin.py checks every function below against the
 * reference itself (oracle/_ref/libmaniac_ref.so, the unmodified Fortran built with amdflang)
 * and against the committed golden vectors in tests/golden/ that were generated from it.
 * The reference's own test-suite holds no per-move golden vector for this path.
 *
 * All indices are 0-based here (the reference is 1-based).  Energies in Kelvin.
 */
#ifndef REFCPU_H
#define REFCPU_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct refcpu refcpu;

/* Build a system.  Layout of the inputs mirrors simulation_state.f90:85-117 flattened to C
 * row-major: atom_types/charges are [n_res][max_atom]; eps/sig are [n_types][n_types];
 * box_matrix[i*3+j] = box%matrix(i+1,j+1).  Runs the restated SetupEwald /
 * PrecomputeValidReciprocalVectors / ComputeReciprocalWeights. */
refcpu *refcpu_create(int n_res, const int *atoms_in_res, int max_atom, const int *is_active,
                      const double *box_matrix, const double *bounds_lo, int is_triclinic,
                      double rc, double tol, const double *charges, const int *atom_types,
                      int n_types, const double *eps, const double *sig, int mol_capacity);
void refcpu_destroy(refcpu *s);

/* com [n_mol][3], off [n_mol][atoms_in_res[t]][3] */
void refcpu_set_molecules(refcpu *s, int t, int n_mol, const double *com, const double *off);
void refcpu_set_molecule(refcpu *s, int t, int m, const double *com, const double *off);
void refcpu_get_molecule(const refcpu *s, int t, int m, double *com, double *off);
void refcpu_set_num_residues(refcpu *s, int t, int n);
int refcpu_get_num_residues(const refcpu *s, int t);

void refcpu_get_box(const refcpu *s, int *box_type, double *volume, double *reciprocal9, double *metrics9);
void refcpu_get_ewald(const refcpu *s, double *alpha, double *rc, double *tol, double *screening,
                      double *fourier_precision, int *kmax3, int *nk);
void refcpu_get_kvectors(const refcpu *s, int *kx, int *ky, int *kz, double *k2norm, double *k2mag,
                         double *form_factor, double *weights);

double refcpu_distance(const refcpu *s, int t1, int m1, int a1, int t2, int m2, int a2);
void refcpu_apply_pbc(const refcpu *s, double *pos3);
double refcpu_lj(const refcpu *s, double r, double sigma, double eps);
double refcpu_coulomb(const refcpu *s, double r, double q1, double q2);

void refcpu_pair_singlemol(const refcpu *s, int t, int m, double *e_non_coulomb, double *e_coulomb);
void refcpu_pair_ordered_singlemol(const refcpu *s, int t, int m, double *e_non_coulomb, double *e_coulomb);
void refcpu_system_energy(refcpu *s, double *out6);

void refcpu_fourier_singlemol(refcpu *s, int t, int m);
void refcpu_all_fourier_terms(refcpu *s);
void refcpu_save_fourier(refcpu *s, int t, int m);
void refcpu_restore_fourier(refcpu *s, int t, int m);
void refcpu_replace_fourier(refcpu *s, int t, int i1, int i2);
void refcpu_get_phase_tables(const refcpu *s, int t, int m, int a, double *px, double *py, double *pz);

void refcpu_init_amplitude(refcpu *s, int full);
void refcpu_get_amplitude(const refcpu *s, double *a_re_im);
void refcpu_set_amplitude(refcpu *s, const double *a_re_im);
double refcpu_recip_singlemol(refcpu *s, int t, int m, int mode);
double refcpu_recip_total(const refcpu *s);
double refcpu_self_singlemol(const refcpu *s, int t);
double refcpu_intra_singlemol(const refcpu *s, int t, int m);
void refcpu_set_energy_recip(refcpu *s, double u);

void refcpu_old_energy(refcpu *s, int t, int m, int kind, double *out6);
void refcpu_new_energy(refcpu *s, int t, int m, int kind, double *out6);
double refcpu_acceptance(double old_total, double new_total, double n_mol, double volume,
                         double fugacity, double temperature, int move_type);
double refcpu_table_lookup(const refcpu *s, int which, double r);
double refcpu_acceptance_swap(double old_total, double new_total, int n_old, int n_new, double phi_old, double phi_new,
                              double T);
void refcpu_rotation_matrix(int axis, double theta, double *r9_rowmajor);
double refcpu_convert_fugacity(double f_atm, double temp_K);

#ifdef __cplusplus
}
#endif
#endif
