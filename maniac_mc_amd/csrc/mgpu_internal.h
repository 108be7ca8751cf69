// Internal declarations shared by the host-setup and engine translation units.
#ifndef MGPU_INTERNAL_H
#define MGPU_INTERNAL_H

#include <string>
#include <vector>

namespace mgpu {

// /root/reference/src/constants.f90:7-20 -- same decimal literals, so bit-identical doubles
constexpr double kPi = 3.14159265358979323846;
constexpr double kTwoPi = 2.0 * kPi;
constexpr double kEps0InvEvA = 14.40198;
constexpr double kKbEvK = 8.6173852e-5;
constexpr double kErrorTol = 1.0e-10;

// records `msg` as the calling thread's last error and returns `code`
int set_error(int code, const std::string &msg);

int box_prepare(const double m[9], int *box_type, double *volume, double rcp[9], double metrics[9]);
int ewald_setup(const double metrics[9], double *rc, double *tol, double *alpha, double *screening,
                double *fourier_precision, int kmax[3], int *nk);
int ewald_kvectors(const double rcp[9], double alpha, const int kmax[3], int nk, int *kx, int *ky, int *kz,
                   double *k2mag, double *ff, double *w);

// Coulomb table rows (see build_coulomb_table in mgpu_host_setup.cpp).  MGPU_COUL_M = top mantissa bits of r^2 used for
// the row index: 6 -> 64 rows per octave, degree-6 rows (5 fp64 + 2 fp32 coefficients); 7 -> 128 rows per octave, degree-5
// rows (5 fp64 + 1 fp32): the same truncation error ((2^-6)^7 = (2^-7)^6), one FMA and one conversion fewer per pair
// term, twice the LDS -- which no longer fits two workgroups per CU, so the pair sweep then runs as one 1024-thread
// workgroup per CU sharing one table.  Measured on MI355X (4096 evaluations per launch): 7 -> 166-167 us, 6 -> 169-171 us,
// with a relative error of 2.9e-15 instead of 1.5e-15 for alpha r in [1, 2] (tests/test_coulomb_table.py) and an LDS
// footprint that grows twice as fast with the box: 2 % is not worth that, 6 stays the default.
#ifndef MGPU_COUL_M
#define MGPU_COUL_M 6
#endif
constexpr int kCoulM = MGPU_COUL_M;
// 8 -> 256 rows per octave, degree-4 rows of 32 bytes (3 fp64 + 2 fp32 coefficients: TWO 16-byte LDS reads instead of three;
// the same absolute error, 1.1e-15, as the degree-6 rows), table from r = 1 A
constexpr int kCoulDeg = kCoulM == 8 ? 4 : (kCoulM == 7 ? 5 : 6);     // polynomial degree of a row
constexpr int kCoulEmin = kCoulM == 8 ? 0 : -2;    // table starts at r^2 = 2^kCoulEmin (r = 0.5 A; 1 A); below it the slow path runs
constexpr double kCoulSlowBelow = kCoulM == 8 ? 1.0 : 0.25;
#if MGPU_COUL_M == 8
struct CoulRow {
    double c[3];
    float c3, c4;
};
static_assert(sizeof(CoulRow) == 32, "CoulRow must be two 16-byte LDS reads");
#else
struct CoulRow {
    double c[5];
    float c5, c6;                // c6 unused (zero) for degree-5 rows
};
static_assert(sizeof(CoulRow) == 48, "CoulRow must be three 16-byte LDS reads");
#endif
constexpr int kCoulRowVec = sizeof(CoulRow) / 16;  // 16-byte LDS reads per row
int build_coulomb_table(double alpha, double s_max, std::vector<CoulRow> &rows, int *idx_base);
double coulomb_table_eval_host(const std::vector<CoulRow> &rows, int idx_base, double alpha, double s);

}  // namespace mgpu

#endif
