// Host-side setup arithmetic of the energy engine: box products, Ewald parameters and the
// half-space k-vector table.  Pure C++ (no HIP), so it also builds with g++ for CPU tests.
//
// Replaces, for the hot path only, the setup half of the reference:
//   PrepareSimulationBox      /root/reference/src/geometry_utils.f90:20-57
//   SetupEwald                /root/reference/src/prepare_utils.f90:103-214
//   PrecomputeValidReciprocalVectors / ComputeReciprocalWeights
//                             /root/reference/src/ewald_kvectors.f90:44-87, :225-246
// The arithmetic keeps the reference's association order so that alpha, kmax, Nk, |k|^2 and
// W(k) come out identical to the Fortran on the same inputs (checked in tests/).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/maniac_gpu.h"
#include "mgpu_internal.h"

namespace mgpu {

static inline void cross3(const double a[3], const double b[3], double c[3]) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
static inline double dot3(const double a[3], const double b[3]) {
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

int box_prepare(const double m[9], int *box_type, double *volume, double rcp[9], double metrics[9]) {
    if (!m || !box_type || !volume || !rcp || !metrics) return set_error(MGPU_ERR_INVALID_ARG, "box_prepare: null argument");
    // box symmetry, geometry_utils.f90:68-94
    const double od[6] = {m[1], m[2], m[3], m[5], m[6], m[7]};
    double mx = 0.0;
    for (double v : od) mx = std::fmax(mx, std::fabs(v));
    if (mx > kErrorTol) *box_type = 3;
    else if (std::fabs(m[0] - m[4]) > kErrorTol || std::fabs(m[0] - m[8]) > kErrorTol) *box_type = 2;
    else *box_type = 1;
    // cell properties, geometry_utils.f90:110-154 (cell vectors are the COLUMNS of the matrix)
    double a[3] = {m[0], m[3], m[6]}, b[3] = {m[1], m[4], m[7]}, c[3] = {m[2], m[5], m[8]};
    double axb[3], bxc[3], cxa[3];
    metrics[0] = std::sqrt(dot3(a, a));
    metrics[1] = std::sqrt(dot3(b, b));
    metrics[2] = std::sqrt(dot3(c, c));
    metrics[3] = dot3(a, b) / (metrics[0] * metrics[1]);
    metrics[4] = dot3(a, c) / (metrics[0] * metrics[2]);
    metrics[5] = dot3(b, c) / (metrics[1] * metrics[2]);
    cross3(a, b, axb);
    cross3(b, c, bxc);
    cross3(c, a, cxa);
    *volume = std::fabs(dot3(a, bxc));
    metrics[6] = *volume / std::sqrt(dot3(bxc, bxc));
    metrics[7] = *volume / std::sqrt(dot3(cxa, cxa));
    metrics[8] = *volume / std::sqrt(dot3(axb, axb));
    // "reciprocal", geometry_utils.f90:277-331: adjugate columns b x c, c x a, a x b over det
    double adj[3][3];
    for (int i = 0; i < 3; ++i) { adj[i][0] = bxc[i]; adj[i][1] = cxa[i]; adj[i][2] = axb[i]; }
    const double det = a[0] * adj[0][0] + a[1] * adj[1][0] + a[2] * adj[2][0];
    if (std::fabs(det) < 1.0) return set_error(MGPU_ERR_INVALID_ARG, "box_prepare: |det(box)| < 1 (reference aborts, geometry_utils.f90:310)");
    const double r = 1.0 / det;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) rcp[i * 3 + j] = r * adj[i][j];
    return MGPU_OK;
}

static inline double normalized_k2(int kx, int ky, int kz, const int kmax[3]) {
    const double x = double(kx) / double(kmax[0]), y = double(ky) / double(kmax[1]), z = double(kz) / double(kmax[2]);
    return x * x + y * y + z * z;
}
static inline bool valid_k(double k2) { return std::fabs(k2) >= kErrorTol && k2 <= 1.0; }

int ewald_setup(const double metrics[9], double *rc, double *tol, double *alpha, double *screening,
                double *fourier_precision, int kmax[3], int *nk) {
    if (!metrics || !rc || !tol || !alpha || !screening || !fourier_precision || !kmax || !nk)
        return set_error(MGPU_ERR_INVALID_ARG, "ewald_setup: null argument");
    if (!(*rc > 0.0)) return set_error(MGPU_ERR_INVALID_ARG, "ewald_setup: real_space_cutoff must be positive");
    // AdjustRealSpaceCutoff, prepare_utils.f90:134-151
    if (*rc > metrics[0] || *rc > metrics[1] || *rc > metrics[2])
        *rc = std::fmin(metrics[0], std::fmin(metrics[1], metrics[2])) / 2.0;
    // ClampTolerance, prepare_utils.f90:157-160
    *tol = std::fmin(std::fabs(*tol), 0.5);
    // ComputeEwaldParameters, prepare_utils.f90:169-180
    *screening = std::sqrt(std::fabs(std::log(*tol * *rc)));
    *alpha = std::sqrt(std::fabs(std::log(*tol * *rc * *screening))) / *rc;
    const double t = 2.0 * *screening * *alpha;
    *fourier_precision = std::sqrt(-std::log(*tol * *rc * (t * t)));
    // ComputeFourierIndices, prepare_utils.f90:187-214 (nint = round half away from zero)
    for (int d = 0; d < 3; ++d) kmax[d] = int(std::lround(0.25 + metrics[d] * *alpha * *fourier_precision / kPi));
    if (kmax[0] < 1 || kmax[1] < 1 || kmax[2] < 1) return set_error(MGPU_ERR_INVALID_ARG, "ewald_setup: kmax < 1");
    int count = 0;
    for (int kx = 0; kx <= kmax[0]; ++kx)
        for (int ky = -kmax[1]; ky <= kmax[1]; ++ky)
            for (int kz = -kmax[2]; kz <= kmax[2]; ++kz) {
                if (kx == 0 && ky == 0 && kz == 0) continue;
                if (valid_k(normalized_k2(kx, ky, kz, kmax))) ++count;
            }
    *nk = count;
    return MGPU_OK;
}

int ewald_kvectors(const double rcp[9], double alpha, const int kmax[3], int nk, int *kx_o, int *ky_o,
                   int *kz_o, double *k2mag, double *ff, double *w) {
    if (!rcp || !kmax || !kx_o || !ky_o || !kz_o || !k2mag || !ff || !w)
        return set_error(MGPU_ERR_INVALID_ARG, "ewald_kvectors: null argument");
    double km[9];
    for (int i = 0; i < 9; ++i) km[i] = kTwoPi * rcp[i];
    int c = 0;
    for (int kx = 0; kx <= kmax[0]; ++kx)
        for (int ky = -kmax[1]; ky <= kmax[1]; ++ky)
            for (int kz = -kmax[2]; kz <= kmax[2]; ++kz) {
                if (kx == 0 && ky == 0 && kz == 0) continue;
                if (!valid_k(normalized_k2(kx, ky, kz, kmax))) continue;
                if (c >= nk) return set_error(MGPU_ERR_INVALID_ARG, "ewald_kvectors: n_kvectors too small");
                double kv[3];
                for (int i = 0; i < 3; ++i)
                    kv[i] = double(kx) * km[i * 3 + 0] + double(ky) * km[i * 3 + 1] + double(kz) * km[i * 3 + 2];
                kx_o[c] = kx; ky_o[c] = ky; kz_o[c] = kz;
                k2mag[c] = dot3(kv, kv);
                ff[c] = (kx == 0) ? 1.0 : 2.0;  // ComputeSymmetryFormFactor, ewald_kvectors.f90:167-180
                ++c;
            }
    if (c != nk) return set_error(MGPU_ERR_INVALID_ARG, "ewald_kvectors: n_kvectors does not match the k list");
    const double a2 = alpha * alpha;
    for (int i = 0; i < nk; ++i) w[i] = std::exp(-k2mag[i] / (4.0 * a2)) / k2mag[i];
    return MGPU_OK;
}

// ------------------------------------------------------------------------------------------
// Coulomb table: G(s) = erfc(alpha sqrt(s)) / sqrt(s) against s = r^2, one row per
// (binary exponent, top kCoulM mantissa bits) of s, i.e. intervals of relative width 2^-6.  Each row
// is a degree-6 polynomial in the mantissa remainder t' = (s - s_lo) / 2^e in [0, 2^-6): five fp64 coefficients and
// two fp32 ones (c5, c6 are < 1e-9 of c0, so 24 bits suffice) = 48 bytes = three ds_read_b128.
// Coefficients: Chebyshev interpolation at 7 nodes in long double (64-bit mantissa, erfcl),
// re-expanded in t, rounded once.  Indexing by the bits of s removes sqrt / rsqrt / erfc / exp and
// every division from the per-pair arithmetic of energy_utils.f90:432.
// ------------------------------------------------------------------------------------------
int build_coulomb_table(double alpha, double s_max, std::vector<CoulRow> &rows, int *idx_base) {
    if (!(alpha > 0.0) || !(s_max > 0.0) || !std::isfinite(s_max))
        return set_error(MGPU_ERR_INVALID_ARG, "build_coulomb_table: bad alpha / range");
    int emax = kCoulEmin;
    while (std::ldexp(1.0, emax + 1) <= s_max) ++emax;          // 2^(emax+1) > s_max
    if (emax - kCoulEmin + 1 > 40) return set_error(MGPU_ERR_CAPACITY, "build_coulomb_table: range too large");
    const int per_oct = 1 << kCoulM, n_oct = emax - kCoulEmin + 1;
    rows.assign((size_t)n_oct * per_oct + 1, CoulRow{});          // last row stays all zero (clamp target)
    *idx_base = (1023 + kCoulEmin) * per_oct;
    constexpr int N = kCoulDeg + 1;           // Chebyshev nodes = coefficients of a row
    const long double pi = 3.14159265358979323846264338327950288L;
    long double node[N], cosjk[N][N];
    for (int k = 0; k < N; ++k) {
        node[k] = cosl(pi * (2 * k + 1) / (2.0L * N));
        for (int j = 0; j < N; ++j) cosjk[j][k] = cosl(pi * j * (2 * k + 1) / (2.0L * N));
    }
    // Chebyshev T_j as monomials in u
    long double T[N][N] = {};
    T[0][0] = 1.0L;
    T[1][1] = 1.0L;
    for (int j = 2; j < N; ++j)
        for (int i = 0; i < N; ++i) T[j][i] = (i > 0 ? 2.0L * T[j - 1][i - 1] : 0.0L) - T[j - 2][i];
    // binomials for u = 2t - 1
    long double binom[N][N] = {};
    for (int j = 0; j < N; ++j) {
        binom[j][0] = 1.0L;
        for (int i = 1; i <= j; ++i) binom[j][i] = binom[j - 1][i - 1] + (i <= j - 1 ? binom[j - 1][i] : 0.0L);
    }
    const long double al = alpha;
    for (int o = 0; o < n_oct; ++o)
        for (int q = 0; q < per_oct; ++q) {
            const long double base = ldexpl(1.0L, kCoulEmin + o);
            const long double a = base * (1.0L + (long double)q / per_oct), b = base * (1.0L + (long double)(q + 1) / per_oct);
            long double f[N], c[N], mono[N] = {}, tc[N] = {};
            for (int k = 0; k < N; ++k) {
                const long double sv = 0.5L * (a + b) + 0.5L * (b - a) * node[k], r = sqrtl(sv);
                f[k] = erfcl(al * r) / r;
            }
            for (int j = 0; j < N; ++j) {
                long double acc = 0.0L;
                for (int k = 0; k < N; ++k) acc += f[k] * cosjk[j][k];
                c[j] = acc * 2.0L / N;
            }
            c[0] *= 0.5L;
            for (int j = 0; j < N; ++j)
                for (int i = 0; i <= j; ++i) mono[i] += c[j] * T[j][i];
            for (int j = 0; j < N; ++j)
                for (int i = 0; i <= j; ++i) {
                    const long double sign = ((j - i) & 1) ? -1.0L : 1.0L;
                    tc[i] += mono[j] * binom[j][i] * ldexpl(1.0L, i) * sign;
                }
            // the device evaluates in ds = s - a (a = the row's first s: s with its low mantissa bits cleared; the
            // subtraction is exact), and t = ds * 64 / 2^e on a row of octave e: scale c_i by (64 / 2^e)^i, a power of two
            CoulRow &row = rows[(size_t)o * per_oct + q];
            const int sh = kCoulM - (kCoulEmin + o);
            for (int i = 0; i < 5; ++i) row.c[i] = (double)ldexpl(tc[i], sh * i);
            row.c5 = (float)ldexpl(tc[5], sh * 5);
            row.c6 = (float)ldexpl(tc[6], sh * 6);
        }
    return MGPU_OK;
}

double coulomb_table_eval_host(const std::vector<CoulRow> &rows, int idx_base, double alpha, double s) {
    unsigned long long bits;
    std::memcpy(&bits, &s, 8);
    const int hi = (int)(bits >> 32);
    int row = (hi >> (20 - kCoulM)) - idx_base;
    if (row < 0) {                                     // below the table (r < 0.5 A): direct evaluation
        const double r = std::sqrt(s);
        return std::erfc(alpha * r) / r;
    }
    const int last = (int)rows.size() - 1;
    if (row > last) row = last;
    const unsigned long long ab = bits & ~((1ull << (52 - kCoulM)) - 1);      // the row's first s
    double a;
    std::memcpy(&a, &ab, 8);
    const double t = s - a;
    const CoulRow &r = rows[row];
    double p = std::fma((double)r.c6, t, (double)r.c5);
    for (int i = 4; i >= 0; --i) p = std::fma(p, t, r.c[i]);
    return p;
}

}  // namespace mgpu

// ------------------------------------------------------------------------------------------
// Atom records of the reference's output files (trajectory.lammpstrj, topology.data: src/write_utils.f90:86, :297-300),
// formatted here instead of by the Fortran runtime: flang's Fw.d conversion costs ~0.2 us per number -- 16 ms per
// UpdateFiles at 10 125 atoms, a third of a single chain's run time once a window is one kernel launch.  The conversion
// below is EXACT (the double's integer mantissa times 10^d in 128-bit arithmetic, shifted with round-half-to-even), so
// the bytes are the runtime's own: tests/test_host_setup.py holds it to flang's output on random and boundary values.
// ------------------------------------------------------------------------------------------
namespace {

// x in Fortran Fw.d into out[0..w) (no terminator); false: not finite (the caller falls back to the Fortran write)
bool format_fixed(char *out, int w, int d, double x) {
    if (!std::isfinite(x)) return false;
    const bool neg = std::signbit(x);
    const double ax = std::fabs(x);
    auto stars = [&]() { std::memset(out, '*', (size_t)w); return true; };
    int e = 0;
    unsigned long long m = 0;
    if (ax != 0.0) {
        const double fr = std::frexp(ax, &e);                 // ax = fr * 2^e, fr in [0.5, 1)
        m = (unsigned long long)std::ldexp(fr, 53);           // exact 53-bit integer
        e -= 53;                                              // ax = m * 2^e
    }
    if (e >= 0 && m != 0) return stars();                     // |x| >= 2^52: never fits these widths
    static const unsigned long long p10[] = {1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull, 10000000ull, 100000000ull,
                                             1000000000ull};
    if (d < 0 || d > 9) return false;
    unsigned __int128 P = (unsigned __int128)m * p10[d];
    unsigned __int128 N = 0;
    const int sh = -e;
    if (m != 0) {
        if (sh >= 127) {
            N = 0;                                            // below half a unit of the last place by far
        } else {
            N = P >> sh;
            const unsigned __int128 rem = P & ((((unsigned __int128)1) << sh) - 1), half = ((unsigned __int128)1) << (sh - 1);
            if (rem > half || (rem == half && (N & 1))) N += 1;
        }
    }
    if (N >> 64) return stars();
    const unsigned long long n64 = (unsigned long long)N, ip = n64 / p10[d], fp = n64 % p10[d];
    char tmp[64];
    int len = 0;
    char digits[24];
    int nd = 0;
    unsigned long long v = ip;
    do { digits[nd++] = (char)('0' + v % 10); v /= 10; } while (v);
    if (neg) tmp[len++] = '-';
    for (int k = nd - 1; k >= 0; --k) tmp[len++] = digits[k];
    tmp[len++] = '.';
    for (int k = d - 1; k >= 0; --k) tmp[len++] = (char)('0' + (fp / p10[k]) % 10);
    if (len > w) {
        // the optional leading zero goes first (F editing, Fortran 2018 13.7.2.3.2)
        if (ip == 0 && len - 1 == w) {
            std::memmove(tmp + (neg ? 1 : 0), tmp + (neg ? 2 : 1), (size_t)(len - (neg ? 2 : 1)));
            len -= 1;
        } else {
            return stars();
        }
    }
    std::memset(out, ' ', (size_t)(w - len));
    std::memcpy(out + (w - len), tmp, (size_t)len);
    return true;
}

void format_int(char *out, int w, long long v) {
    char tmp[32];
    const int len = std::snprintf(tmp, sizeof tmp, "%lld", v);
    if (len > w) { std::memset(out, '*', (size_t)w); return; }
    std::memset(out, ' ', (size_t)(w - len));
    std::memcpy(out + (w - len), tmp, (size_t)len);
}

}  // namespace

using mgpu::set_error;

extern "C" {

// n atom records appended to `path`.  mol == NULL: '(I6,1X,I4,3(1X,F12.7))' = serial, type, x, y, z (WriteLAMMPSTRJ);
// else '(I6,1X,I6,1X,I4,1X,F12.8,3(1X,F12.7))' = serial, molecule, type, charge, x, y, z (WriteLAMMPSData).  xyz is
// [n][3]; serials count from first_serial.  Returns MGPU_OK, or an error WITHOUT having written anything (a value that is
// not finite, a file that cannot be opened): the caller then writes the records itself.
int mgpu_append_atom_records(const char *path, int n, int first_serial, const int *mol, const int *type, const double *charge,
                             const double *xyz) {
    if (!path || n < 0 || !type || !xyz || (mol && !charge)) return set_error(MGPU_ERR_INVALID_ARG, "append_atom_records: bad argument");
    const int reclen = mol ? (6 + 1 + 6 + 1 + 4 + 1 + 12 + 3 * 13 + 1) : (6 + 1 + 4 + 3 * 13 + 1);
    std::vector<char> buf((size_t)n * reclen);
    for (int i = 0; i < n; ++i) {
        char *r = buf.data() + (size_t)i * reclen;
        int at = 0;
        format_int(r + at, 6, (long long)first_serial + i); at += 6;
        r[at++] = ' ';
        if (mol) {
            format_int(r + at, 6, mol[i]); at += 6;
            r[at++] = ' ';
        }
        format_int(r + at, 4, type[i]); at += 4;
        if (mol) {
            r[at++] = ' ';
            if (!format_fixed(r + at, 12, 8, charge[i])) return set_error(MGPU_ERR_INVALID_ARG, "append_atom_records: charge not finite");
            at += 12;
        }
        for (int dd = 0; dd < 3; ++dd) {
            r[at++] = ' ';
            if (!format_fixed(r + at, 12, 7, xyz[(size_t)i * 3 + dd])) return set_error(MGPU_ERR_INVALID_ARG, "append_atom_records: coordinate not finite");
            at += 12;
        }
        r[at++] = '\n';
    }
    FILE *f = std::fopen(path, "ab");
    if (!f) return set_error(MGPU_ERR_STATE, std::string("append_atom_records: cannot open ") + path);
    const size_t wrote = buf.empty() ? 0 : std::fwrite(buf.data(), 1, buf.size(), f);
    const int rc = std::fclose(f);
    if (wrote != buf.size() || rc != 0) return set_error(MGPU_ERR_STATE, std::string("append_atom_records: short write to ") + path);
    return MGPU_OK;
}

// test hook: x[i] in Fortran Fw.d, w characters each, into out[n * w]; out[i * w] = '?' where the value is not finite
int mgpu_format_fixed(int n, const double *x, int w, int d, char *out) {
    if (n < 0 || !x || !out || w < 1 || w > 40) return set_error(MGPU_ERR_INVALID_ARG, "format_fixed: bad argument");
    for (int i = 0; i < n; ++i)
        if (!format_fixed(out + (size_t)i * w, w, d, x[i])) std::memset(out + (size_t)i * w, '?', (size_t)w);
    return MGPU_OK;
}

int mgpu_box_prepare(const double box_matrix[9], int *box_type, double *volume, double reciprocal[9], double metrics[9]) {
    return mgpu::box_prepare(box_matrix, box_type, volume, reciprocal, metrics);
}

int mgpu_ewald_setup(const double metrics[9], double *rc, double *tol, double *alpha, double *screening_factor,
                     double *fourier_precision, int kmax[3], int *n_kvectors) {
    return mgpu::ewald_setup(metrics, rc, tol, alpha, screening_factor, fourier_precision, kmax, n_kvectors);
}

// Host evaluation of the Coulomb table the pair sweep reads from LDS (same rows and the same
// index / Horner / FMA arithmetic as coul_lds() in mgpu_kernels_common.h), so its accuracy can be checked
// without a GPU.  out[i] = erfc(alpha sqrt(r2[i])) / sqrt(r2[i]).
int mgpu_coulomb_table_eval(double alpha, double r2_max, int n, const double *r2, double *out) {
    if (n < 0 || (n > 0 && (!r2 || !out))) return mgpu::set_error(MGPU_ERR_INVALID_ARG, "mgpu_coulomb_table_eval: bad argument");
    std::vector<mgpu::CoulRow> rows;
    int idx_base = 0;
    int rc = mgpu::build_coulomb_table(alpha, r2_max, rows, &idx_base);
    if (rc) return rc;
    for (int k = 0; k < n; ++k) out[k] = mgpu::coulomb_table_eval_host(rows, idx_base, alpha, r2[k]);
    return MGPU_OK;
}

void mgpu_host_prefetch(const void *p, int bytes) {
    for (int o = 0; o < bytes; o += 64) __builtin_prefetch((const char *)p + o, 0, 3);
}

int mgpu_rng_seed_streams(long long seed, int n_streams, long long *state) {
    if (n_streams < 0 || (n_streams > 0 && !state)) return mgpu::set_error(MGPU_ERR_INVALID_ARG, "mgpu_rng_seed_streams: bad argument");
    auto splitmix = [](unsigned long long &x) {
        unsigned long long z = (x += 0x9e3779b97f4a7c15ULL);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        return z ^ (z >> 31);
    };
    unsigned long long base = (unsigned long long)seed;
    base = splitmix(base);                                   // decorrelate neighbouring user seeds
    for (int r = 0; r < n_streams; ++r) {
        unsigned long long x = base + 0xd1342543de82ef95ULL * (unsigned long long)(r + 1);
        unsigned long long w[4];
        do {
            for (auto &v : w) v = splitmix(x);
        } while ((w[0] | w[1] | w[2] | w[3]) == 0ULL);
        for (int i = 0; i < 4; ++i) state[4 * (size_t)r + i] = (long long)w[i];
    }
    return MGPU_OK;
}

// n_per numbers of each of n consecutive streams: the scalar recurrence (mc_farm.f90 chain_random: xoshiro256+, the top 53
// bits of s0 + s3 scaled by 2^-53) on four streams at a time.  One stream is a dependent chain of ~5 cycles per number;
// four abreast, 256 bits wide where the CPU has AVX2, the loop runs at the adders' throughput.  The integer -> double
// conversion is exact in both forms (a 53-bit integer), so the numbers are the scalar loop's, bit for bit.
namespace {
inline void rng_fill_scalar(unsigned long long *st, int n, int n_per, double *out) {
    for (int r = 0; r < n; ++r) {
        unsigned long long s0 = st[4 * (size_t)r], s1 = st[4 * (size_t)r + 1], s2 = st[4 * (size_t)r + 2], s3 = st[4 * (size_t)r + 3];
        double *o = out + (size_t)r * n_per;
        for (int i = 0; i < n_per; ++i) {
            o[i] = (double)((s0 + s3) >> 11) * (1.0 / 9007199254740992.0);
            const unsigned long long t = s1 << 17;
            s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3;
            s2 ^= t;
            s3 = (s3 << 45) | (s3 >> 19);
        }
        st[4 * (size_t)r] = s0; st[4 * (size_t)r + 1] = s1; st[4 * (size_t)r + 2] = s2; st[4 * (size_t)r + 3] = s3;
    }
}
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("avx2"))) void rng_fill_avx2(unsigned long long *st, int n, int n_per, double *out) {
    const __m256i magic = _mm256_set1_epi64x(0x4330000000000000LL);           // 2^52 as a bit pattern
    const __m256d two52 = _mm256_set1_pd(4503599627370496.0), two32 = _mm256_set1_pd(4294967296.0);
    const __m256d scale = _mm256_set1_pd(1.0 / 9007199254740992.0);
    const __m256i lo_mask = _mm256_set1_epi64x(0xffffffffLL);
    int r = 0;
    for (; r + 4 <= n; r += 4) {
        // four states, transposed into one register per state word
        const __m256i a = _mm256_loadu_si256((const __m256i *)(st + 4 * (size_t)r)), b = _mm256_loadu_si256((const __m256i *)(st + 4 * (size_t)r + 4));
        const __m256i c = _mm256_loadu_si256((const __m256i *)(st + 4 * (size_t)r + 8)), d = _mm256_loadu_si256((const __m256i *)(st + 4 * (size_t)r + 12));
        const __m256i ab_lo = _mm256_unpacklo_epi64(a, b), ab_hi = _mm256_unpackhi_epi64(a, b);      // a0 b0 a2 b2 | a1 b1 a3 b3
        const __m256i cd_lo = _mm256_unpacklo_epi64(c, d), cd_hi = _mm256_unpackhi_epi64(c, d);
        __m256i s0 = _mm256_permute2x128_si256(ab_lo, cd_lo, 0x20), s1 = _mm256_permute2x128_si256(ab_hi, cd_hi, 0x20);
        __m256i s2 = _mm256_permute2x128_si256(ab_lo, cd_lo, 0x31), s3 = _mm256_permute2x128_si256(ab_hi, cd_hi, 0x31);
        for (int i = 0; i < n_per; ++i) {
            const __m256i x = _mm256_srli_epi64(_mm256_add_epi64(s0, s3), 11);                       // 53 bits
            const __m256d hi = _mm256_sub_pd(_mm256_castsi256_pd(_mm256_or_si256(_mm256_srli_epi64(x, 32), magic)), two52);
            const __m256d lo = _mm256_sub_pd(_mm256_castsi256_pd(_mm256_or_si256(_mm256_and_si256(x, lo_mask), magic)), two52);
            const __m256d v = _mm256_mul_pd(_mm256_add_pd(_mm256_mul_pd(hi, two32), lo), scale);       // exact: hi 2^32 + lo < 2^53
            alignas(32) double tmp[4];
            _mm256_store_pd(tmp, v);
            for (int q = 0; q < 4; ++q) out[(size_t)(r + q) * n_per + i] = tmp[q];
            const __m256i t = _mm256_slli_epi64(s1, 17);
            s2 = _mm256_xor_si256(s2, s0); s3 = _mm256_xor_si256(s3, s1); s1 = _mm256_xor_si256(s1, s2); s0 = _mm256_xor_si256(s0, s3);
            s2 = _mm256_xor_si256(s2, t);
            s3 = _mm256_or_si256(_mm256_slli_epi64(s3, 45), _mm256_srli_epi64(s3, 19));
        }
        // back to one state per stream
        const __m256i t0 = _mm256_unpacklo_epi64(s0, s1), t1 = _mm256_unpackhi_epi64(s0, s1);       // r0: s0 s1 | r2: s0 s1 ; r1 ... r3
        const __m256i t2 = _mm256_unpacklo_epi64(s2, s3), t3 = _mm256_unpackhi_epi64(s2, s3);
        _mm256_storeu_si256((__m256i *)(st + 4 * (size_t)r), _mm256_permute2x128_si256(t0, t2, 0x20));
        _mm256_storeu_si256((__m256i *)(st + 4 * (size_t)r + 4), _mm256_permute2x128_si256(t1, t3, 0x20));
        _mm256_storeu_si256((__m256i *)(st + 4 * (size_t)r + 8), _mm256_permute2x128_si256(t0, t2, 0x31));
        _mm256_storeu_si256((__m256i *)(st + 4 * (size_t)r + 12), _mm256_permute2x128_si256(t1, t3, 0x31));
    }
    if (r < n) rng_fill_scalar(st + 4 * (size_t)r, n - r, n_per, out + (size_t)r * n_per);
}
#endif
}  // namespace

int mgpu_rng_fill(long long *state, int n_streams, int n_per, double *out) {
    if (n_streams < 0 || n_per < 0 || (n_streams > 0 && n_per > 0 && (!state || !out)))
        return mgpu::set_error(MGPU_ERR_INVALID_ARG, "mgpu_rng_fill: bad argument");
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2 && n_streams >= 4) {
        rng_fill_avx2((unsigned long long *)state, n_streams, n_per, out);
        return MGPU_OK;
    }
#endif
    rng_fill_scalar((unsigned long long *)state, n_streams, n_per, out);
    return MGPU_OK;
}

int mgpu_ewald_kvectors(const double reciprocal[9], double alpha, const int kmax[3], int n_kvectors, int *kx, int *ky,
                        int *kz, double *k2mag, double *form_factor, double *weights) {
    return mgpu::ewald_kvectors(reciprocal, alpha, kmax, n_kvectors, kx, ky, kz, k2mag, form_factor, weights);
}

}  // extern "C"
