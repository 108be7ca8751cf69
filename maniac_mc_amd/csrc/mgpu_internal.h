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

// Coulomb table rows (see build_coulomb_table in mgpu_host_setup.cpp): the row index is the binary exponent and the top
// kCoulM = 6 mantissa bits of r^2 (64 rows per octave); a row is a degree-6 polynomial, 5 fp64 + 2 fp32 coefficients =
// three 16-byte LDS reads.  (128 rows per octave with degree-5 rows and 256 with 32-byte degree-4 rows were built and
// measured in rounds 2-3: 2-3 % at twice / four times the LDS; LABNOTES.md.)
constexpr int kCoulM = 6;
constexpr int kCoulDeg = 6;                        // polynomial degree of a row
constexpr int kCoulEmin = -2;                      // table starts at r^2 = 2^kCoulEmin (r = 0.5 A); below it the slow path runs
constexpr double kCoulSlowBelow = 0.25;
struct CoulRow {
    double c[5];
    float c5, c6;
};
static_assert(sizeof(CoulRow) == 48, "CoulRow must be three 16-byte LDS reads");
constexpr int kCoulRowVec = sizeof(CoulRow) / 16;  // 16-byte LDS reads per row
int build_coulomb_table(double alpha, double s_max, std::vector<CoulRow> &rows, int *idx_base);
double coulomb_table_eval_host(const std::vector<CoulRow> &rows, int idx_base, double alpha, double s);

}  // namespace mgpu

#endif
