/*
 * TEST INFRASTRUCTURE ONLY (oracle).  See refcpu.h for the rules of use.
 *
 * Plain-C restatement of the reference's per-move energy path: same loop order and the same
 * expression shapes (association order of every floating-point expression) as the Fortran it
 * follows, which is cited function by function as /root/reference/src/<file>:<lines>.
 * Compiled with -ffp-contract=off so no FMA contraction changes the rounding.
 */
#include "refcpu.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* constants.f90:7-20 (same decimal literals, so the doubles are bit-identical) */
static const double PI = 3.14159265358979323846;
#define TWOPI (2.0 * PI)
static const double KB_JK = 1.380658e-23;
static const double EPS0_INV_eVA = 14.40198;
static const double KB_eVK = 8.6173852e-5;
static const double ERR = 1.0e-10;
/* parameters.f90:28-29 */
static const double A3_TO_M3 = 1.0e-30;
static const double ATM_TO_PA = 1.01325e5;

typedef struct { double re, im; } cplx;

static inline cplx cmul(cplx a, cplx b) { cplx r = { a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re }; return r; }

struct refcpu {
    int n_res, max_atom, n_types, cap;
    int *atoms_in_res, *is_active, *atom_types; /* [n_res][max_atom] */
    double *charges;                            /* [n_res][max_atom] */
    double *eps, *sig;                          /* [n_types][n_types] */
    double m[3][3];                             /* box%matrix(i,j)     */
    double rcp[3][3];                           /* box%reciprocal(i,j) */
    double lo[3], metrics[9], volume, determinant;
    int box_type, is_triclinic;
    double rc, tol, alpha, screening, fourier_precision;
    int kmax[3], nk;
    int *kx, *ky, *kz;
    double *k2norm, *k2mag, *ff, *w;
    int *n_mol;
    double **com;                               /* [t] -> [cap][3]           */
    double **off;                               /* [t] -> [cap][max_atom][3] */
    cplx **px, **py, **pz;                      /* [t] -> [cap][max_atom][2kmax+1] */
    cplx *px_old, *py_old, *pz_old;             /* [max_atom][2kmax+1] */
    cplx *A, *A_old;
    double e_non_coulomb, e_coulomb, e_recip, e_self, e_intra, e_total;
};

/* Fortran MODULO for reals as lowered by flang (_FortranAModuloReal8): fmod, then shifted
 * into the sign of p.  Used by geometry_utils.f90:190, :210, :389. */
static double f_modulo(double a, double p)
{
    double r = fmod(a, p);
    if (r != 0.0 && ((r < 0.0) != (p < 0.0))) r += p;
    return r;
}

/* helper_utils.f90:150-161 CrossProduct */
static void cross(const double a[3], const double b[3], double c[3])
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

static double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

static void col(const refcpu *s, int j, double v[3]) { v[0] = s->m[0][j]; v[1] = s->m[1][j]; v[2] = s->m[2][j]; }

/* geometry_utils.f90:68-94 DetermineBoxSymmetry */
static void determine_box_symmetry(refcpu *s)
{
    double od[6] = { s->m[0][1], s->m[0][2], s->m[1][0], s->m[1][2], s->m[2][0], s->m[2][1] };
    double mx = 0.0;
    for (int i = 0; i < 6; ++i) if (fabs(od[i]) > mx) mx = fabs(od[i]);
    if (mx > ERR) s->box_type = 3;
    else if (fabs(s->m[0][0] - s->m[1][1]) > ERR || fabs(s->m[0][0] - s->m[2][2]) > ERR) s->box_type = 2;
    else s->box_type = 1;
}

/* geometry_utils.f90:110-154 ComputeCellProperties */
static void compute_cell_properties(refcpu *s)
{
    double a[3], b[3], c[3], axb[3], bxc[3], cxa[3], len[3];
    col(s, 0, a); col(s, 1, b); col(s, 2, c);
    for (int j = 0; j < 3; ++j)
        s->metrics[j] = sqrt(s->m[0][j] * s->m[0][j] + s->m[1][j] * s->m[1][j] + s->m[2][j] * s->m[2][j]);
    len[0] = sqrt(dot3(a, a)); len[1] = sqrt(dot3(b, b)); len[2] = sqrt(dot3(c, c));
    s->metrics[3] = dot3(a, b) / (len[0] * len[1]);
    s->metrics[4] = dot3(a, c) / (len[0] * len[2]);
    s->metrics[5] = dot3(b, c) / (len[1] * len[2]);
    cross(a, b, axb); cross(b, c, bxc); cross(c, a, cxa);
    s->volume = fabs(dot3(a, bxc));
    s->metrics[6] = s->volume / sqrt(dot3(bxc, bxc));
    s->metrics[7] = s->volume / sqrt(dot3(cxa, cxa));
    s->metrics[8] = s->volume / sqrt(dot3(axb, axb));
}

/* geometry_utils.f90:277-331 ComputeInverse: reciprocal(i,j) = adjugate(i,j)/det where
 * adjugate(:,1) = a2 x a3, adjugate(:,2) = a3 x a1, adjugate(:,3) = a1 x a2. */
static void compute_inverse(refcpu *s)
{
    double a[3], b[3], adj[3][3], c[3];
    col(s, 1, a); col(s, 2, b); cross(a, b, c); for (int i = 0; i < 3; ++i) adj[i][0] = c[i];
    col(s, 2, a); col(s, 0, b); cross(a, b, c); for (int i = 0; i < 3; ++i) adj[i][1] = c[i];
    col(s, 0, a); col(s, 1, b); cross(a, b, c); for (int i = 0; i < 3; ++i) adj[i][2] = c[i];
    col(s, 0, a);
    s->determinant = a[0] * adj[0][0] + a[1] * adj[1][0] + a[2] * adj[2][0];
    double r = 1.0 / s->determinant;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) s->rcp[i][j] = r * adj[i][j];
}

/* ewald_kvectors.f90:105-118 NormalizedKSquared */
static double normalized_k2(int kx, int ky, int kz, const int kmax[3])
{
    double x = (double)kx / (double)kmax[0], y = (double)ky / (double)kmax[1], z = (double)kz / (double)kmax[2];
    return x * x + y * y + z * z;
}

/* ewald_kvectors.f90:199-210 CheckValidReciprocalVector */
static int valid_k(double k2) { return (fabs(k2) >= ERR) && (k2 <= 1.0); }

/* prepare_utils.f90:103-214 SetupEwald = AdjustRealSpaceCutoff, ClampTolerance,
 * ComputeEwaldParameters, ComputeFourierIndices */
static void setup_ewald(refcpu *s)
{
    if (s->rc > s->metrics[0] || s->rc > s->metrics[1] || s->rc > s->metrics[2]) {
        double mn = s->metrics[0];
        if (s->metrics[1] < mn) mn = s->metrics[1];
        if (s->metrics[2] < mn) mn = s->metrics[2];
        s->rc = mn / 2.0;
    }
    s->tol = fmin(fabs(s->tol), 0.5);
    s->screening = sqrt(fabs(log(s->tol * s->rc)));
    s->alpha = sqrt(fabs(log(s->tol * s->rc * s->screening))) / s->rc;
    {
        double t = 2.0 * s->screening * s->alpha;
        s->fourier_precision = sqrt(-log(s->tol * s->rc * (t * t)));
    }
    for (int d = 0; d < 3; ++d)
        s->kmax[d] = (int)lround(0.25 + s->metrics[d] * s->alpha * s->fourier_precision / PI); /* nint */
    int count = 0;
    for (int kx = 0; kx <= s->kmax[0]; ++kx)
        for (int ky = -s->kmax[1]; ky <= s->kmax[1]; ++ky)
            for (int kz = -s->kmax[2]; kz <= s->kmax[2]; ++kz) {
                if (kx == 0 && ky == 0 && kz == 0) continue;
                if (valid_k(normalized_k2(kx, ky, kz, s->kmax))) ++count;
            }
    s->nk = count;
}

/* ewald_kvectors.f90:44-87 PrecomputeValidReciprocalVectors (+ :134-152, :167-180) and
 * ewald_kvectors.f90:225-246 ComputeReciprocalWeights */
static void precompute_kvectors(refcpu *s)
{
    double km[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) km[i][j] = TWOPI * s->rcp[i][j];
    int c = 0;
    for (int kx = 0; kx <= s->kmax[0]; ++kx)
        for (int ky = -s->kmax[1]; ky <= s->kmax[1]; ++ky)
            for (int kz = -s->kmax[2]; kz <= s->kmax[2]; ++kz) {
                if (kx == 0 && ky == 0 && kz == 0) continue;
                double k2 = normalized_k2(kx, ky, kz, s->kmax);
                if (!valid_k(k2)) continue;
                double kv[3];
                for (int i = 0; i < 3; ++i)
                    kv[i] = (double)kx * km[i][0] + (double)ky * km[i][1] + (double)kz * km[i][2];
                s->kx[c] = kx; s->ky[c] = ky; s->kz[c] = kz;
                s->k2norm[c] = k2;
                s->k2mag[c] = dot3(kv, kv);
                s->ff[c] = (kx == 0) ? 1.0 : 2.0;
                ++c;
            }
    double a2 = s->alpha * s->alpha;
    for (int i = 0; i < s->nk; ++i) s->w[i] = exp(-s->k2mag[i] / (4.0 * a2)) / s->k2mag[i];
}

#define AT(s, t, a) ((t) * (s)->max_atom + (a))
static inline int nkx(const refcpu *s) { return 2 * s->kmax[0] + 1; }
static inline int nky(const refcpu *s) { return 2 * s->kmax[1] + 1; }
static inline int nkz(const refcpu *s) { return 2 * s->kmax[2] + 1; }
static inline cplx *PX(const refcpu *s, int t, int m, int a) { return s->px[t] + ((size_t)m * s->max_atom + a) * nkx(s) + s->kmax[0]; }
static inline cplx *PY(const refcpu *s, int t, int m, int a) { return s->py[t] + ((size_t)m * s->max_atom + a) * nky(s) + s->kmax[1]; }
static inline cplx *PZ(const refcpu *s, int t, int m, int a) { return s->pz[t] + ((size_t)m * s->max_atom + a) * nkz(s) + s->kmax[2]; }

refcpu *refcpu_create(int n_res, const int *atoms_in_res, int max_atom, const int *is_active,
                      const double *box_matrix, const double *bounds_lo, int is_triclinic,
                      double rc, double tol, const double *charges, const int *atom_types,
                      int n_types, const double *eps, const double *sig, int mol_capacity)
{
    refcpu *s = (refcpu *)calloc(1, sizeof(refcpu));
    s->n_res = n_res; s->max_atom = max_atom; s->n_types = n_types; s->cap = mol_capacity;
    s->atoms_in_res = (int *)malloc(sizeof(int) * n_res);
    s->is_active = (int *)malloc(sizeof(int) * n_res);
    s->atom_types = (int *)malloc(sizeof(int) * n_res * max_atom);
    s->charges = (double *)malloc(sizeof(double) * n_res * max_atom);
    s->eps = (double *)malloc(sizeof(double) * n_types * n_types);
    s->sig = (double *)malloc(sizeof(double) * n_types * n_types);
    memcpy(s->atoms_in_res, atoms_in_res, sizeof(int) * n_res);
    memcpy(s->is_active, is_active, sizeof(int) * n_res);
    memcpy(s->atom_types, atom_types, sizeof(int) * n_res * max_atom);
    memcpy(s->charges, charges, sizeof(double) * n_res * max_atom);
    memcpy(s->eps, eps, sizeof(double) * n_types * n_types);
    memcpy(s->sig, sig, sizeof(double) * n_types * n_types);
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) s->m[i][j] = box_matrix[i * 3 + j]; s->lo[i] = bounds_lo[i]; }
    s->is_triclinic = is_triclinic;
    s->rc = rc; s->tol = tol;
    determine_box_symmetry(s);
    compute_cell_properties(s);
    compute_inverse(s);
    setup_ewald(s);
    s->kx = (int *)malloc(sizeof(int) * s->nk); s->ky = (int *)malloc(sizeof(int) * s->nk); s->kz = (int *)malloc(sizeof(int) * s->nk);
    s->k2norm = (double *)malloc(sizeof(double) * s->nk); s->k2mag = (double *)malloc(sizeof(double) * s->nk);
    s->ff = (double *)malloc(sizeof(double) * s->nk); s->w = (double *)malloc(sizeof(double) * s->nk);
    precompute_kvectors(s);
    s->n_mol = (int *)calloc(n_res, sizeof(int));
    s->com = (double **)calloc(n_res, sizeof(double *));
    s->off = (double **)calloc(n_res, sizeof(double *));
    s->px = (cplx **)calloc(n_res, sizeof(cplx *)); s->py = (cplx **)calloc(n_res, sizeof(cplx *)); s->pz = (cplx **)calloc(n_res, sizeof(cplx *));
    for (int t = 0; t < n_res; ++t) {
        s->com[t] = (double *)calloc((size_t)s->cap * 3, sizeof(double));
        s->off[t] = (double *)calloc((size_t)s->cap * max_atom * 3, sizeof(double));
        s->px[t] = (cplx *)calloc((size_t)s->cap * max_atom * nkx(s), sizeof(cplx));
        s->py[t] = (cplx *)calloc((size_t)s->cap * max_atom * nky(s), sizeof(cplx));
        s->pz[t] = (cplx *)calloc((size_t)s->cap * max_atom * nkz(s), sizeof(cplx));
    }
    s->px_old = (cplx *)calloc((size_t)max_atom * nkx(s), sizeof(cplx));
    s->py_old = (cplx *)calloc((size_t)max_atom * nky(s), sizeof(cplx));
    s->pz_old = (cplx *)calloc((size_t)max_atom * nkz(s), sizeof(cplx));
    s->A = (cplx *)calloc(s->nk, sizeof(cplx));
    s->A_old = (cplx *)calloc(s->nk, sizeof(cplx));
    return s;
}

void refcpu_destroy(refcpu *s)
{
    if (!s) return;
    for (int t = 0; t < s->n_res; ++t) { free(s->com[t]); free(s->off[t]); free(s->px[t]); free(s->py[t]); free(s->pz[t]); }
    free(s->com); free(s->off); free(s->px); free(s->py); free(s->pz);
    free(s->px_old); free(s->py_old); free(s->pz_old); free(s->A); free(s->A_old);
    free(s->kx); free(s->ky); free(s->kz); free(s->k2norm); free(s->k2mag); free(s->ff); free(s->w);
    free(s->n_mol); free(s->atoms_in_res); free(s->is_active); free(s->atom_types); free(s->charges); free(s->eps); free(s->sig);
    free(s);
}

void refcpu_set_molecule(refcpu *s, int t, int m, const double *com, const double *off)
{
    int n1 = s->atoms_in_res[t];
    memcpy(s->com[t] + (size_t)m * 3, com, sizeof(double) * 3);
    memcpy(s->off[t] + (size_t)m * s->max_atom * 3, off, sizeof(double) * 3 * n1);
}

void refcpu_set_molecules(refcpu *s, int t, int n_mol, const double *com, const double *off)
{
    int n1 = s->atoms_in_res[t];
    s->n_mol[t] = n_mol;
    for (int m = 0; m < n_mol; ++m) refcpu_set_molecule(s, t, m, com + (size_t)m * 3, off + (size_t)m * n1 * 3);
}

void refcpu_get_molecule(const refcpu *s, int t, int m, double *com, double *off)
{
    int n1 = s->atoms_in_res[t];
    memcpy(com, s->com[t] + (size_t)m * 3, sizeof(double) * 3);
    memcpy(off, s->off[t] + (size_t)m * s->max_atom * 3, sizeof(double) * 3 * n1);
}

void refcpu_set_num_residues(refcpu *s, int t, int n) { s->n_mol[t] = n; }
int refcpu_get_num_residues(const refcpu *s, int t) { return s->n_mol[t]; }

void refcpu_get_box(const refcpu *s, int *box_type, double *volume, double *reciprocal9, double *metrics9)
{
    *box_type = s->box_type; *volume = s->volume;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) reciprocal9[i * 3 + j] = s->rcp[i][j];
    memcpy(metrics9, s->metrics, sizeof(double) * 9);
}

void refcpu_get_ewald(const refcpu *s, double *alpha, double *rc, double *tol, double *screening,
                      double *fourier_precision, int *kmax3, int *nk)
{
    *alpha = s->alpha; *rc = s->rc; *tol = s->tol; *screening = s->screening; *fourier_precision = s->fourier_precision;
    kmax3[0] = s->kmax[0]; kmax3[1] = s->kmax[1]; kmax3[2] = s->kmax[2]; *nk = s->nk;
}

void refcpu_get_kvectors(const refcpu *s, int *kx, int *ky, int *kz, double *k2norm, double *k2mag,
                         double *form_factor, double *weights)
{
    memcpy(kx, s->kx, sizeof(int) * s->nk); memcpy(ky, s->ky, sizeof(int) * s->nk); memcpy(kz, s->kz, sizeof(int) * s->nk);
    memcpy(k2norm, s->k2norm, sizeof(double) * s->nk); memcpy(k2mag, s->k2mag, sizeof(double) * s->nk);
    memcpy(form_factor, s->ff, sizeof(double) * s->nk); memcpy(weights, s->w, sizeof(double) * s->nk);
}

static void site_pos(const refcpu *s, int t, int m, int a, double p[3])
{
    const double *c = s->com[t] + (size_t)m * 3;
    const double *o = s->off[t] + ((size_t)m * s->max_atom + a) * 3;
    p[0] = c[0] + o[0]; p[1] = c[1] + o[1]; p[2] = c[2] + o[2];
}

/* geometry_utils.f90:359-415 ComputeDistance */
double refcpu_distance(const refcpu *s, int t1, int m1, int a1, int t2, int m2, int a2)
{
    double p1[3], p2[3], d[3];
    site_pos(s, t1, m1, a1, p1);
    site_pos(s, t2, m2, a2, p2);
    for (int i = 0; i < 3; ++i) d[i] = p2[i] - p1[i];
    if (s->box_type == 1 || s->box_type == 2) {
        for (int i = 0; i < 3; ++i)
            d[i] = f_modulo(d[i] + 0.5 * s->m[i][i], s->m[i][i]) - 0.5 * s->m[i][i];
        return sqrt(dot3(d, d));
    }
    double best = HUGE_VAL; /* huge(1.0_real64) is DBL_MAX; any finite trial beats either */
    for (int sx = -1; sx <= 1; ++sx)
        for (int sy = -1; sy <= 1; ++sy)
            for (int sz = -1; sz <= 1; ++sz) {
                double tr[3];
                for (int i = 0; i < 3; ++i)
                    tr[i] = d[i] + (double)sx * s->m[i][0] + (double)sy * s->m[i][1] + (double)sz * s->m[i][2];
                double t2v = dot3(tr, tr);
                if (t2v < best) best = t2v;
            }
    return sqrt(best);
}

/* geometry_utils.f90:167-220 ApplyPBC */
void refcpu_apply_pbc(const refcpu *s, double *pos)
{
    if (!s->is_triclinic) {
        for (int d = 0; d < 3; ++d)
            pos[d] = s->lo[d] + f_modulo(pos[d] - s->lo[d], s->m[d][d]);
    } else {
        double v[3], f[3];
        for (int d = 0; d < 3; ++d) v[d] = pos[d] - s->lo[d];
        for (int i = 0; i < 3; ++i) f[i] = s->rcp[i][0] * v[0] + s->rcp[i][1] * v[1] + s->rcp[i][2] * v[2];
        for (int i = 0; i < 3; ++i) f[i] = f_modulo(f[i], 1.0);
        for (int i = 0; i < 3; ++i) pos[i] = s->lo[i] + (s->m[i][0] * f[0] + s->m[i][1] * f[1] + s->m[i][2] * f[2]);
    }
}

/* energy_utils.f90:192-220 LennardJonesEnergy (use_table = .false., parameters.f90:42) */
double refcpu_lj(const refcpu *s, double r, double sigma, double eps)
{
    if (r >= s->rc) return 0.0;
    double x = sigma / r;
    double r6 = x * x * x * x * x * x;
    double r12 = r6 * r6;
    return 4.0 * eps * (r12 - r6);
}

/* energy_utils.f90:225-255 CoulombEnergy */
double refcpu_coulomb(const refcpu *s, double r, double q1, double q2)
{
    if (fabs(q1) < ERR || fabs(q2) < ERR) return 0.0;
    if (r < ERR) return 0.0;
    return q1 * q2 * erfc(s->alpha * r) / r;
}

/* energy_utils.f90:374-442 ComputePairInteractionEnergy_singlemol */
void refcpu_pair_singlemol(const refcpu *s, int t1, int m1, double *e_nc, double *e_c)
{
    double enc = 0.0, ec = 0.0;
    for (int a1 = 0; a1 < s->atoms_in_res[t1]; ++a1)
        for (int t2 = 0; t2 < s->n_res; ++t2)
            for (int m2 = 0; m2 < s->n_mol[t2]; ++m2) {
                if (m1 == m2 && t1 == t2) continue;
                for (int a2 = 0; a2 < s->atoms_in_res[t2]; ++a2) {
                    double r = refcpu_distance(s, t1, m1, a1, t2, m2, a2);
                    if (r < s->rc) {
                        int ti = s->atom_types[AT(s, t1, a1)] - 1, tj = s->atom_types[AT(s, t2, a2)] - 1;
                        double sg = s->sig[ti * s->n_types + tj], ep = s->eps[ti * s->n_types + tj];
                        double x = sg / r;
                        double r6 = x * x * x * x * x * x;
                        double r12 = r6 * r6;
                        enc = enc + 4.0 * ep * (r12 - r6);
                    }
                    double q1 = s->charges[AT(s, t1, a1)], q2 = s->charges[AT(s, t2, a2)];
                    if (fabs(q1) < ERR || fabs(q2) < ERR) continue;
                    ec = ec + q1 * q2 * erfc(s->alpha * r) / r;
                }
            }
    *e_nc = enc;
    *e_c = ec * EPS0_INV_eVA / KB_eVK;
}

/* energy_utils.f90:121-187 SingleMolPairwiseEnergy (ordered pairs) */
void refcpu_pair_ordered_singlemol(const refcpu *s, int t1, int m1, double *e_nc, double *e_c)
{
    double enc = 0.0, ec = 0.0;
    for (int a1 = 0; a1 < s->atoms_in_res[t1]; ++a1)
        for (int t2 = 0; t2 < s->n_res; ++t2)
            for (int m2 = 0; m2 < s->n_mol[t2]; ++m2) {
                if (m1 == m2 && t1 == t2) continue;
                if (t2 < t1 || (t2 == t1 && m2 <= m1)) continue;
                for (int a2 = 0; a2 < s->atoms_in_res[t2]; ++a2) {
                    int ti = s->atom_types[AT(s, t1, a1)] - 1, tj = s->atom_types[AT(s, t2, a2)] - 1;
                    double sg = s->sig[ti * s->n_types + tj], ep = s->eps[ti * s->n_types + tj];
                    double q1 = s->charges[AT(s, t1, a1)], q2 = s->charges[AT(s, t2, a2)];
                    double r = refcpu_distance(s, t1, m1, a1, t2, m2, a2);
                    enc = enc + refcpu_lj(s, r, sg, ep);
                    ec = ec + refcpu_coulomb(s, r, q1, q2);
                }
            }
    *e_nc = enc;
    *e_c = ec * EPS0_INV_eVA / KB_eVK;
}

/* ewald_phase.f90:41-64 ComputeAtomPhase + :90-111 ComputePhaseFactors1D + :383-420 */
void refcpu_fourier_singlemol(refcpu *s, int t, int m)
{
    for (int a = 0; a < s->atoms_in_res[t]; ++a) {
        double p[3], ph[3];
        site_pos(s, t, m, a, p);
        for (int i = 0; i < 3; ++i) {
            double acc = 0.0;
            for (int j = 0; j < 3; ++j) acc = acc + s->rcp[j][i] * p[j];
            ph[i] = TWOPI * acc;
        }
        cplx *tab[3] = { PX(s, t, m, a), PY(s, t, m, a), PZ(s, t, m, a) };
        for (int d = 0; d < 3; ++d)
            for (int k = 0; k <= s->kmax[d]; ++k) {
                cplx e = { cos((double)k * ph[d]), sin((double)k * ph[d]) };
                tab[d][k] = e;
                if (k != 0) { cplx c = { e.re, -e.im }; tab[d][-k] = c; }
            }
    }
}

/* ewald_phase.f90:340-360 ComputeAllFourierTerms */
void refcpu_all_fourier_terms(refcpu *s)
{
    for (int t = 0; t < s->n_res; ++t)
        for (int m = 0; m < s->n_mol[t]; ++m) refcpu_fourier_singlemol(s, t, m);
}

/* ewald_phase.f90:134-183 SaveSingleMolFourierTerms (x: k >= 0 only, y/z: both signs; all A) */
void refcpu_save_fourier(refcpu *s, int t, int m)
{
    for (int a = 0; a < s->atoms_in_res[t]; ++a) {
        cplx *ox = s->px_old + (size_t)a * nkx(s) + s->kmax[0];
        cplx *oy = s->py_old + (size_t)a * nky(s) + s->kmax[1];
        cplx *oz = s->pz_old + (size_t)a * nkz(s) + s->kmax[2];
        for (int k = 0; k <= s->kmax[0]; ++k) ox[k] = PX(s, t, m, a)[k];
        for (int k = 0; k <= s->kmax[1]; ++k) { oy[k] = PY(s, t, m, a)[k]; if (k) oy[-k] = PY(s, t, m, a)[-k]; }
        for (int k = 0; k <= s->kmax[2]; ++k) { oz[k] = PZ(s, t, m, a)[k]; if (k) oz[-k] = PZ(s, t, m, a)[-k]; }
    }
    memcpy(s->A_old, s->A, sizeof(cplx) * s->nk);
}

/* ewald_phase.f90:205-255 RestoreSingleMolFourier */
void refcpu_restore_fourier(refcpu *s, int t, int m)
{
    for (int a = 0; a < s->atoms_in_res[t]; ++a) {
        const cplx *ox = s->px_old + (size_t)a * nkx(s) + s->kmax[0];
        const cplx *oy = s->py_old + (size_t)a * nky(s) + s->kmax[1];
        const cplx *oz = s->pz_old + (size_t)a * nkz(s) + s->kmax[2];
        for (int k = 0; k <= s->kmax[0]; ++k) PX(s, t, m, a)[k] = ox[k];
        for (int k = 0; k <= s->kmax[1]; ++k) { PY(s, t, m, a)[k] = oy[k]; if (k) PY(s, t, m, a)[-k] = oy[-k]; }
        for (int k = 0; k <= s->kmax[2]; ++k) { PZ(s, t, m, a)[k] = oz[k]; if (k) PZ(s, t, m, a)[-k] = oz[-k]; }
    }
    memcpy(s->A, s->A_old, sizeof(cplx) * s->nk);
}

/* ewald_phase.f90:276-322 ReplaceFourierTermsSingleMol (slot i1 <- slot i2) */
void refcpu_replace_fourier(refcpu *s, int t, int i1, int i2)
{
    for (int a = 0; a < s->atoms_in_res[t]; ++a) {
        for (int k = 0; k <= s->kmax[0]; ++k) PX(s, t, i1, a)[k] = PX(s, t, i2, a)[k];
        for (int k = 0; k <= s->kmax[1]; ++k) { PY(s, t, i1, a)[k] = PY(s, t, i2, a)[k]; if (k) PY(s, t, i1, a)[-k] = PY(s, t, i2, a)[-k]; }
        for (int k = 0; k <= s->kmax[2]; ++k) { PZ(s, t, i1, a)[k] = PZ(s, t, i2, a)[k]; if (k) PZ(s, t, i1, a)[-k] = PZ(s, t, i2, a)[-k]; }
    }
}

void refcpu_get_phase_tables(const refcpu *s, int t, int m, int a, double *px, double *py, double *pz)
{
    memcpy(px, PX(s, t, m, a) - s->kmax[0], sizeof(cplx) * nkx(s));
    memcpy(py, PY(s, t, m, a) - s->kmax[1], sizeof(cplx) * nky(s));
    memcpy(pz, PZ(s, t, m, a) - s->kmax[2], sizeof(cplx) * nkz(s));
}

/* ewald_energy.f90:40-77 ComputeRecipAmplitude */
static cplx recip_amplitude(const refcpu *s, int kx, int ky, int kz)
{
    cplx amp = { 0.0, 0.0 };
    for (int t = 0; t < s->n_res; ++t)
        for (int m = 0; m < s->n_mol[t]; ++m)
            for (int a = 0; a < s->atoms_in_res[t]; ++a) {
                double q = s->charges[AT(s, t, a)];
                cplx ph = cmul(cmul(PX(s, t, m, a)[kx], PY(s, t, m, a)[ky]), PZ(s, t, m, a)[kz]);
                amp.re = amp.re + q * ph.re;
                amp.im = amp.im + q * ph.im;
            }
    return amp;
}

/* helper_utils.f90:125-134 amplitude_squared: real(z * conjg(z)) */
static double amp2(cplx z) { return z.re * z.re - z.im * (-z.im); }

/* ewald_energy.f90:105-147 ComputeReciprocalEnergy */
double refcpu_recip_total(const refcpu *s)
{
    double u = 0.0;
    for (int i = 0; i < s->nk; ++i) {
        cplx amp = recip_amplitude(s, s->kx[i], s->ky[i], s->kz[i]);
        u = u + s->ff[i] * s->w[i] * amp2(amp);
    }
    return u * EPS0_INV_eVA / KB_eVK * TWOPI / s->volume;
}

/* The initialisation the reference omits: A(k) <- 0 or A(k) <- S(k) */
void refcpu_init_amplitude(refcpu *s, int full)
{
    for (int i = 0; i < s->nk; ++i) {
        if (full) s->A[i] = recip_amplitude(s, s->kx[i], s->ky[i], s->kz[i]);
        else { s->A[i].re = 0.0; s->A[i].im = 0.0; }
    }
}

void refcpu_get_amplitude(const refcpu *s, double *a) { memcpy(a, s->A, sizeof(cplx) * s->nk); }
void refcpu_set_amplitude(refcpu *s, const double *a) { memcpy(s->A, a, sizeof(cplx) * s->nk); }
void refcpu_set_energy_recip(refcpu *s, double u) { s->e_recip = u; }

/* ewald_energy.f90:191-274 ComputeRecipEnergySingleMol.  mode 0 move, 1 creation, 2 deletion.
 * Mutates A(k) in place exactly like the reference. */
double refcpu_recip_singlemol(refcpu *s, int t, int m, int mode)
{
    double u = 0.0;
    int n1 = s->atoms_in_res[t];
    for (int i = 0; i < s->nk; ++i) {
        int kx = s->kx[i], ky = s->ky[i], kz = s->kz[i];
        cplx sum = { 0.0, 0.0 };
        for (int a = 0; a < n1; ++a) {
            double q = s->charges[AT(s, t, a)];
            cplx pn = cmul(cmul(PX(s, t, m, a)[kx], PY(s, t, m, a)[ky]), PZ(s, t, m, a)[kz]);
            cplx po = cmul(cmul((s->px_old + (size_t)a * nkx(s) + s->kmax[0])[kx],
                                (s->py_old + (size_t)a * nky(s) + s->kmax[1])[ky]),
                           (s->pz_old + (size_t)a * nkz(s) + s->kmax[2])[kz]);
            if (mode == 1) { sum.re = sum.re + q * pn.re; sum.im = sum.im + q * pn.im; }
            else if (mode == 2) { sum.re = sum.re + q * po.re; sum.im = sum.im + q * po.im; }
            else { sum.re = sum.re + q * (pn.re - po.re); sum.im = sum.im + q * (pn.im - po.im); }
        }
        if (mode == 2) { s->A[i].re = s->A[i].re - sum.re; s->A[i].im = s->A[i].im - sum.im; }
        else { s->A[i].re = s->A[i].re + sum.re; s->A[i].im = s->A[i].im + sum.im; }
        u = u + s->ff[i] * s->w[i] * amp2(s->A[i]);
    }
    return u * EPS0_INV_eVA / KB_eVK * TWOPI / s->volume;
}

/* ewald_energy.f90:308-336 ComputeEwaldSelfInteractionSingleMol (== energy_utils.f90:340-368) */
double refcpu_self_singlemol(const refcpu *s, int t)
{
    double e = 0.0;
    const double sqrtpi = sqrt(PI);
    for (int a = 0; a < s->atoms_in_res[t]; ++a) {
        double q = s->charges[AT(s, t, a)];
        if (fabs(q) < ERR) continue;
        e = e - s->alpha / sqrtpi * (q * q);
    }
    return e * EPS0_INV_eVA / KB_eVK;
}

/* ewald_energy.f90:371-411 ComputeIntraResidueRealCoulombEnergySingleMol */
double refcpu_intra_singlemol(const refcpu *s, int t, int m)
{
    double u = 0.0;
    int n1 = s->atoms_in_res[t];
    for (int a1 = 0; a1 < n1 - 1; ++a1) {
        double q1 = s->charges[AT(s, t, a1)];
        for (int a2 = a1 + 1; a2 < n1; ++a2) {
            double q2 = s->charges[AT(s, t, a2)];
            double r = refcpu_distance(s, t, m, a1, t, m, a2);
            if (r > ERR) u = u + q1 * q2 * (erfc(s->alpha * r) - 1.0) / r;
        }
    }
    return u * EPS0_INV_eVA / KB_eVK;
}

/* energy_utils.f90:18-35 ComputeSystemEnergy = ComputePairwiseEnergy (:83-115), ComputeEwaldSelf
 * (:307-330, zeroed first here -- the reference relies on static zero-init), ComputeEwaldRecip
 * (:270-286), ComputeTotalIntraResidueCoulombEnergy (:55-81, active residue types only). */
void refcpu_system_energy(refcpu *s, double *out)
{
    s->e_non_coulomb = 0.0; s->e_coulomb = 0.0;
    for (int t = 0; t < s->n_res; ++t)
        for (int m = 0; m < s->n_mol[t]; ++m) {
            double a, b;
            refcpu_pair_ordered_singlemol(s, t, m, &a, &b);
            s->e_non_coulomb = s->e_non_coulomb + a;
            s->e_coulomb = s->e_coulomb + b;
        }
    s->e_self = 0.0;
    for (int t = 0; t < s->n_res; ++t) {
        double e = refcpu_self_singlemol(s, t);
        e = e * (double)s->n_mol[t];
        s->e_self = s->e_self + e;
    }
    refcpu_all_fourier_terms(s);
    s->e_recip = refcpu_recip_total(s);
    s->e_intra = 0.0;
    for (int t = 0; t < s->n_res; ++t)
        if (s->is_active[t] == 1)
            for (int m = 0; m < s->n_mol[t]; ++m) s->e_intra = s->e_intra + refcpu_intra_singlemol(s, t, m);
    s->e_total = s->e_recip + s->e_non_coulomb + s->e_coulomb + s->e_self + s->e_intra;
    out[0] = s->e_non_coulomb; out[1] = s->e_coulomb; out[2] = s->e_recip;
    out[3] = s->e_self; out[4] = s->e_intra; out[5] = s->e_total;
}

/* monte_carlo_utils.f90:347-395 ComputeOldEnergy. kind 0 move, 1 creation, 2 deletion.
 * out = non_coulomb, coulomb, recip_coulomb, ewald_self, intra_coulomb, total */
void refcpu_old_energy(refcpu *s, int t, int m, int kind, double *o)
{
    o[0] = o[1] = o[2] = o[3] = o[4] = o[5] = 0.0;
    if (kind == 1) {
        o[2] = s->e_recip;
        o[5] = o[0] + o[1] + o[2] + o[3] + o[4];
    } else if (kind == 2) {
        o[3] = refcpu_self_singlemol(s, t);
        o[4] = refcpu_intra_singlemol(s, t, m);
        refcpu_pair_singlemol(s, t, m, &o[0], &o[1]);
        o[2] = s->e_recip;
        o[5] = o[0] + o[1] + o[2] + o[3] + o[4];
    } else {
        o[2] = refcpu_recip_singlemol(s, t, m, 0);
        refcpu_pair_singlemol(s, t, m, &o[0], &o[1]);
        o[5] = o[0] + o[1] + o[2];
    }
}

/* monte_carlo_utils.f90:275-320 ComputeNewEnergy.  The deletion branch passes
 * is_creation = deletion_flag (:308) -- restated AS WRITTEN. */
void refcpu_new_energy(refcpu *s, int t, int m, int kind, double *o)
{
    o[0] = o[1] = o[2] = o[3] = o[4] = o[5] = 0.0;
    if (kind == 1) {
        refcpu_fourier_singlemol(s, t, m);
        o[2] = refcpu_recip_singlemol(s, t, m, 1);
        refcpu_pair_singlemol(s, t, m, &o[0], &o[1]);
        o[3] = refcpu_self_singlemol(s, t);
        o[4] = refcpu_intra_singlemol(s, t, m);
        o[5] = o[0] + o[1] + o[2] + o[3] + o[4];
    } else if (kind == 2) {
        o[2] = refcpu_recip_singlemol(s, t, m, 1);
        o[5] = o[0] + o[1] + o[2] + o[3] + o[4];
    } else {
        refcpu_fourier_singlemol(s, t, m);
        o[2] = refcpu_recip_singlemol(s, t, m, 0);
        refcpu_pair_singlemol(s, t, m, &o[0], &o[1]);
        o[5] = o[0] + o[1] + o[2];
    }
}

/* monte_carlo_utils.f90:184-226 mc_acceptance_probability.
 * move_type ids (parameters.f90:36-39): 1 creation, 2 deletion, 3 translation, 4 rotation */
double refcpu_acceptance(double old_total, double new_total, double N, double V, double phi, double T, int move_type)
{
    double de = new_total - old_total;
    if (move_type == 1) return fmin(1.0, (phi * V / N) * exp(-de / T));
    if (move_type == 2) return fmin(1.0, ((N + 1.0) / (phi * V)) * exp(-de / T));
    return fmin(1.0, exp(-de / T));
}

/* monte_carlo_utils.f90:228-268 mc_acceptance_probability_swap: a molecule of type `old` (N_old of them) becomes one of
 * type `new` (N_new of them); the reference has this rule but no swap move (monte_carlo.f90:50-75 has no branch).
 * The reference forms N_old / (N_new + 1) with `one` a real64: real(N_new + one) is N_new + 1.0 */
double refcpu_acceptance_swap(double old_total, double new_total, int n_old, int n_new, double phi_old, double phi_new, double T)
{
    double de = new_total - old_total;
    double combinatorial = (double)n_old / ((double)n_new + 1.0);
    return fmin(1.0, (phi_new / phi_old) * combinatorial * exp(-de / T));
}

/* tabulated_utils.f90:10-113: the reference's tabulated potentials -- InitializeTabulatedErfcR / InitializeTabulatedRPower
 * fill TABULATED_POINTS + 1 = 5001 points on [0, rc] (parameters.f90:41), LookupTabulated interpolates linearly and returns
 * 0 from rc on.  Dead code in the reference (use_table is a compile-time .false., parameters.f90:42): restated and pinned
 * for completeness; neither the reference nor the engine ever evaluates an energy with them.
 * which: 1 erfc(alpha r)/r, 2 r**6, 3 r**12. */
#define REFCPU_TABULATED_POINTS 5000
static double int_power(double r, int p)
{
    /* flang lowers r**6 / r**12 with an integer exponent to repeated squaring: r2 = r*r; r4 = r2*r2; ... */
    double acc = 1.0, b = r;
    int e = p;
    while (e > 0) {
        if (e & 1) acc *= b;
        e >>= 1;
        if (e) b *= b;
    }
    return acc;
}
static double table_point(const refcpu *s, int which, int i)
{
    const double dx = s->rc / (double)REFCPU_TABULATED_POINTS;
    const double r = i * dx;
    if (which == 1) return (r < ERR) ? 2.0 * s->alpha / sqrt(PI) : erfc(s->alpha * r) / r;
    if (r < ERR) return 0.0;
    return int_power(r, which == 2 ? 6 : 12);
}
double refcpu_table_lookup(const refcpu *s, int which, double r)
{
    const double dx = s->rc / (double)REFCPU_TABULATED_POINTS;
    if (r <= 0.0) return table_point(s, which, 0);
    if (r >= REFCPU_TABULATED_POINTS * dx) return 0.0;
    const int i = (int)(r / dx);
    const double f1 = table_point(s, which, i), f2 = table_point(s, which, i + 1);
    const double t = (r - i * dx) / dx;
    return (1.0 - t) * f1 + t * f2;
}

/* helper_utils.f90:39-77 RotationMatrix; r[i*3+j] = rotation_matrix(i+1, j+1) */
void refcpu_rotation_matrix(int axis, double theta, double *r)
{
    double c = cos(theta), sn = sin(theta);
    for (int i = 0; i < 9; ++i) r[i] = 0.0;
    r[0] = r[4] = r[8] = 1.0;
    if (axis == 1) { r[4] = c; r[5] = -sn; r[7] = sn; r[8] = c; }
    else if (axis == 2) { r[0] = c; r[2] = sn; r[6] = -sn; r[8] = c; }
    else if (axis == 3) { r[0] = c; r[1] = -sn; r[3] = sn; r[4] = c; }
}

/* prepare_utils.f90:48-73 ConvertFugacity: atm -> molecules per cubic Angstrom */
double refcpu_convert_fugacity(double f_atm, double temp_K)
{
    double thermal_energy = KB_JK * temp_K;
    return f_atm * ATM_TO_PA * A3_TO_M3 / thermal_energy;
}


/* ------------------------------------------------------------------------------------------------
 * CPU baseline helper (bench.py cpu_baseline, "all cores" leg): n_chains independent copies of the SAME
 * sequential Metropolis trial the reference runs (Translation / Rotation, translation.f90:36-112,
 * rotation.f90:34-75: save, old energy, move, new energy, accept or restore), one chain per OpenMP thread
 * at a time -- the replica-level parallelism the GPU farm uses, on the host's cores.  Runs until
 * `budget_s` seconds have passed (checked after every trial); counts[2 * c] = trials, counts[2 * c + 1] =
 * accepted moves of chain c.  Returns the elapsed wall time.  TEST INFRASTRUCTURE, like the rest of this file.
 * ---------------------------------------------------------------------------------------------- */
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline unsigned long long splitmix64(unsigned long long *x)
{
    unsigned long long z = (*x += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
static inline double u01(unsigned long long *x) { return (double)(splitmix64(x) >> 11) * (1.0 / 9007199254740992.0); }

static double wall_now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double refcpu_trial_farm(refcpu **chains, int n_chains, int n_threads, double budget_s, unsigned long long seed,
                         double translation_step, double rotation_step, double temperature, long long *counts)
{
    const double t0 = wall_now();
#ifdef _OPENMP
#pragma omp parallel for num_threads(n_threads) schedule(static, 1)
#endif
    for (int c = 0; c < n_chains; ++c) {
        refcpu *s = chains[c];
        unsigned long long rng = seed + 0x1234567ULL * (unsigned long long)(c + 1);
        const int t = 0, n = s->n_mol[0], n1 = s->atoms_in_res[0];
        double com[3], off[3 * 64], ncom[3], noff[3 * 64], rot[9], eo[6], en[6];
        long long trials = 0, accepted = 0;
        while (wall_now() - t0 < budget_s) {
            const int m = (int)(u01(&rng) * n) % n;
            refcpu_get_molecule(s, t, m, com, off);
            refcpu_save_fourier(s, t, m);
            refcpu_old_energy(s, t, m, 0, eo);
            if (u01(&rng) <= 0.5 || n1 == 1) {
                for (int d = 0; d < 3; ++d) ncom[d] = com[d] + (u01(&rng) - 0.5) * translation_step;
                refcpu_apply_pbc(s, ncom);
                refcpu_set_molecule(s, t, m, ncom, off);
            } else {
                refcpu_rotation_matrix(1 + (int)(u01(&rng) * 3.0) % 3, (u01(&rng) - 0.5) * rotation_step, rot);
                for (int a = 0; a < n1; ++a)
                    for (int i = 0; i < 3; ++i)
                        noff[3 * a + i] = rot[3 * i] * off[3 * a] + rot[3 * i + 1] * off[3 * a + 1] + rot[3 * i + 2] * off[3 * a + 2];
                refcpu_set_molecule(s, t, m, com, noff);
            }
            refcpu_new_energy(s, t, m, 0, en);
            ++trials;
            if (u01(&rng) <= fmin(1.0, exp(-(en[5] - eo[5]) / temperature))) {
                ++accepted;
            } else {
                refcpu_set_molecule(s, t, m, com, off);
                refcpu_restore_fourier(s, t, m);
            }
        }
        counts[2 * c] = trials;
        counts[2 * c + 1] = accepted;
    }
    return wall_now() - t0;
}
