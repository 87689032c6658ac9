/*
 * thunder_oracle.c -- CPU restatement of thuem/THUNDER v1.4.14's per-iteration E/M hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under thunder_amd/ may include, link or call this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as the
 * checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the reference cannot be compiled in this image under the build rules
 * (every translation unit reaches include/Precision.h:26-46, which needs GSL, FFTW3 and the
 * cmake-generated THUNDERConfig.h; boost 1.60 is a missing blob), and the reference's own tests
 * hold no golden vector for this path.  Every function below therefore
 * restates the reference source line by line (file:line cited, paths relative to
 * /root/reference) including its float/double cast points; the GSL 2.4 routines it relies on
 * (gsl_hypot, gsl_hypot3, gsl_sf_bessel_j0) are restated from the vendored source
 * external/packages/gsl-2.4/{sys/hypot.c:24-76, specfunc/bessel_j.c:35-58}.
 *
 * Build flags that matter: -ffp-contract=off (the reference is built -O2 -mavx, CMakeLists.txt:95,
 * 132-133: no FMA contraction on x86-64), RFLOAT == float (SINGLE_PRECISION, CMakeLists.txt:48).
 *
 * Layouts (include/Image/Volume.h:567-575, include/Image/Image.h iFTHalf):
 *   volume FT : complex64 [P][P][P/2+1], index (k<0?k+P:k)*(P/2+1)*P + (j<0?j+P:j)*(P/2+1) + i
 *   image  FT : complex64 [N][N/2+1]
 *   dmat33    : 9 doubles, column-major (Eigen default, include/Typedef.h:149)
 *   T volume  : kept as a REAL float volume here; the reference stores it complex with the
 *               imaginary part never written (src/Image/Volume.cpp:676-677 adds to dat[0] only).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

typedef float RFLOAT;

/* ------------------------------------------------------------------------------------------ */
/* GSL 2.4 restatements                                                                       */
/* ------------------------------------------------------------------------------------------ */

/* external/packages/gsl-2.4/sys/hypot.c:24-55 (NORM, include/Functions/Functions.h:70) */
static double gsl_hypot_(double x, double y)
{
    double xabs = fabs(x), yabs = fabs(y), min, max;
    if (isinf(x) || isinf(y)) return INFINITY;
    if (xabs < yabs) { min = xabs; max = yabs; } else { min = yabs; max = xabs; }
    if (min == 0) return max;
    {
        double u = min / max;
        return max * sqrt(1 + u * u);
    }
}

/* external/packages/gsl-2.4/sys/hypot.c:57-76 (NORM_3, include/Functions/Functions.h:80) */
static double gsl_hypot3_(double x, double y, double z)
{
    double xabs = fabs(x), yabs = fabs(y), zabs = fabs(z);
    double w = xabs > (yabs > zabs ? yabs : zabs) ? xabs : (yabs > zabs ? yabs : zabs);
    if (w == 0.0) return 0.0;
    return w * sqrt((xabs / w) * (xabs / w) + (yabs / w) * (yabs / w) + (zabs / w) * (zabs / w));
}

/* external/packages/gsl-2.4/specfunc/bessel_j.c:35-58 */
static double gsl_sf_bessel_j0_(double x)
{
    double ax = fabs(x);
    if (ax < 0.5) {
        const double y = x * x;
        const double c1 = -1.0 / 6.0, c2 = 1.0 / 120.0, c3 = -1.0 / 5040.0, c4 = 1.0 / 362880.0,
                     c5 = -1.0 / 39916800.0, c6 = 1.0 / 6227020800.0;
        return 1.0 + y * (c1 + y * (c2 + y * (c3 + y * (c4 + y * (c5 + y * c6)))));
    }
    return sin(x) / x;
}

/* AROUND(a) = (int)rint(a), include/Functions/Functions.h:30 */
static inline int AROUND_(double a) { return (int)rint(a); }

/* TSGSL_pow_2/3/4 : RFLOAT in, gsl_pow_N in double, RFLOAT out (src/Precision.cpp:263-276) */
static inline RFLOAT pow2f_(RFLOAT x) { double d = x; return (RFLOAT)(d * d); }
static inline RFLOAT pow3f_(RFLOAT x) { double d = x; return (RFLOAT)(d * d * d); }
static inline RFLOAT pow4f_(RFLOAT x) { double d = x; double d2 = d * d; return (RFLOAT)(d2 * d2); }

/* TIK_RL, src/Functions/Functions.cpp:236-239; TSGSL_sf_bessel_j0 takes RFLOAT (src/Precision.cpp:362) */
RFLOAT orc_TIK_RL(RFLOAT r)
{
    RFLOAT x = (RFLOAT)(M_PI * r);
    RFLOAT j = (RFLOAT)gsl_sf_bessel_j0_((double)x);
    return pow2f_(j);
}

/* Bessel pieces of MKB_RL (order 0): I0 by its power series, I_{3/2} in closed form.
 * The reference calls gsl_sf_bessel_I0 / gsl_sf_bessel_Inu(1.5, v) (src/Functions/Functions.cpp:
 * 160-176); both are accurate to double rounding, as are these, and the result is narrowed to
 * float, so the table agrees to <= 1 float ulp. */
static double bessel_I0_(double x)
{
    double q = x * x / 4.0, term = 1.0, sum = 1.0;
    for (int k = 1; k < 500; k++) {
        term *= q / ((double)k * (double)k);
        sum += term;
        if (term < sum * 1e-17) break;
    }
    return sum;
}
static double bessel_I15_(double v) { return sqrt(2.0 / (M_PI * v)) * (cosh(v) - sinh(v) / v); }
static double bessel_J15_(double v) { return sqrt(2.0 / (M_PI * v)) * (sin(v) / v - cos(v)); }

/* MKB_RL, src/Functions/Functions.cpp:143-176 (FUNCTIONS_MKB_ORDER_0, include/Config.h) */
RFLOAT orc_MKB_RL(RFLOAT r, RFLOAT a, RFLOAT alpha)
{
    RFLOAT u = (RFLOAT)(2 * M_PI * a * r);
    RFLOAT v = (u <= alpha) ? (RFLOAT)sqrt(pow2f_(alpha) - pow2f_(u)) : (RFLOAT)sqrt(pow2f_(u) - pow2f_(alpha));
    RFLOAT I0a = (RFLOAT)bessel_I0_((double)alpha);
    RFLOAT w = (RFLOAT)(pow(2 * M_PI, 1.5) * pow3f_(a) / I0a / pow((double)v, 1.5));
    if (u <= alpha) return w * (RFLOAT)bessel_I15_((double)v);
    return w * (RFLOAT)bessel_J15_((double)v);
}

/* MKB_RL_R2, src/Functions/Functions.cpp:178-214 */
RFLOAT orc_MKB_RL_R2(RFLOAT r2, RFLOAT a, RFLOAT alpha)
{
    RFLOAT u2 = pow2f_((RFLOAT)(2 * M_PI * a)) * r2;
    RFLOAT v = (u2 <= pow2f_(alpha)) ? (RFLOAT)sqrt(pow2f_(alpha) - u2) : (RFLOAT)sqrt(u2 - pow2f_(alpha));
    RFLOAT I0a = (RFLOAT)bessel_I0_((double)alpha);
    RFLOAT w = (RFLOAT)(pow(2 * M_PI, 1.5) * pow3f_(a) / I0a / pow((double)v, 1.5));
    if (u2 <= pow2f_(alpha)) return w * (RFLOAT)bessel_I15_((double)v);
    return w * (RFLOAT)bessel_J15_((double)v);
}

/* TabFunction::init over [a,b] with n steps, src/TabFunction.cpp:26-39; Reconstructor::init builds
 * _kernelRL = MKB_RL_R2(., a, alpha) on [0,1] with 1e5 steps (src/Reconstructor.cpp:77-86).
 * tab must hold n+1 floats. */
void orc_kernelRL_table(RFLOAT* tab, int n, RFLOAT a, RFLOAT alpha)
{
    RFLOAT ta = 0, tb = 1;
    RFLOAT s = (tb - ta) / n;
    for (int i = 0; i <= n; i++) tab[i] = orc_MKB_RL_R2(ta + i * s, a, alpha);
}

/* TabFunction::operator(), src/TabFunction.cpp:42-45 (nearest sample) */
static inline RFLOAT tab_lookup_(const RFLOAT* tab, int n, RFLOAT x)
{
    RFLOAT ta = 0, tb = 1;
    RFLOAT s = (tb - ta) / n;
    return tab[AROUND_((x - ta) / s)];
}

/* ------------------------------------------------------------------------------------------ */
/* a1: pixel list  -- Optimiser::allocPreCalIdx, src/Optimiser.cpp:7991-8041                   */
/* ------------------------------------------------------------------------------------------ */
int orc_pixel_list(int N, RFLOAT rU, RFLOAT rL, int pf, int* iCol, int* iRow, int* iPxl, int* iSig,
                   int* iColPad, int* iRowPad)
{
    RFLOAT rU2 = pow2f_(rU), rL2 = pow2f_(rL);
    int n = 0;
    /* IMAGE_FOR_PIXEL_R_FT(rU + 1), include/Image/Image.h:68-70 */
    for (long j = (long)(-(rU + 1)); j < (rU + 1); j++)
        for (long i = 0; i <= (rU + 1); i++) {
            if ((i == 0) && (j < 0)) continue;
            RFLOAT u = (RFLOAT)((double)i * (double)i + (double)j * (double)j); /* QUAD */
            if ((u < rU2) && (u >= rL2)) {
                int v = AROUND_(gsl_hypot_((double)i, (double)j));
                if ((v < rU) && (v >= rL)) {
                    if (iPxl) iPxl[n] = (int)((j >= 0 ? j : j + N) * (N / 2 + 1) + i); /* Image::iFTHalf */
                    if (iCol) iCol[n] = (int)i;
                    if (iRow) iRow[n] = (int)j;
                    if (iSig) iSig[n] = v;
                    if (iColPad) iColPad[n] = (int)i * pf;
                    if (iRowPad) iRowPad[n] = (int)j * pf;
                    n++;
                }
            }
        }
    return n;
}

/* ------------------------------------------------------------------------------------------ */
/* a3: CTF on a pixel list -- src/CTF.cpp:113-151                                              */
/* ------------------------------------------------------------------------------------------ */
void orc_ctf(RFLOAT* dst, RFLOAT pixelSize, RFLOAT voltage, RFLOAT defocusU, RFLOAT defocusV, RFLOAT theta,
             RFLOAT Cs, RFLOAT amplitudeContrast, RFLOAT phaseShift, int nCol, int nRow, const int* iCol,
             const int* iRow, int nPxl)
{
    RFLOAT lambda = (RFLOAT)(12.2643247 / sqrt(voltage * (1 + voltage * 0.978466e-6)));
    RFLOAT w1 = sqrtf(1 - pow2f_(amplitudeContrast));
    RFLOAT w2 = amplitudeContrast;
    RFLOAT K1 = (RFLOAT)(M_PI * lambda);
    RFLOAT K2 = (RFLOAT)(M_PI_2 * Cs * pow3f_(lambda));
    for (int i = 0; i < nPxl; i++) {
        RFLOAT u = (RFLOAT)gsl_hypot_((double)(iCol[i] / (pixelSize * nCol)), (double)(iRow[i] / (pixelSize * nRow)));
        RFLOAT angle = (RFLOAT)(atan2((double)iRow[i], (double)iCol[i]) - theta);
        RFLOAT defocus = -(defocusU + defocusV + (defocusU - defocusV) * cosf(2 * angle)) / 2;
        RFLOAT ki = K1 * defocus * pow2f_(u) + K2 * pow4f_(u) - phaseShift;
        dst[i] = -w1 * sinf(ki) + w2 * cosf(ki);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* a4: phase ramps -- translate(), src/Image/ImageFunctions.cpp:233-252 and :471-492          */
/* M_2X_PI = 6.28318530717959 (include/Macro.h:14); COMPLEX_POLAR(-phase), include/Complex.h   */
/* ------------------------------------------------------------------------------------------ */
void orc_translate(RFLOAT* dst, RFLOAT nTransCol, RFLOAT nTransRow, int nCol, int nRow, const int* iCol,
                   const int* iRow, int nPxl)
{
    RFLOAT rCol = nTransCol / nCol, rRow = nTransRow / nRow;
    for (int i = 0; i < nPxl; i++) {
        RFLOAT phase = (RFLOAT)(6.28318530717959 * (iCol[i] * rCol + iRow[i] * rRow));
        dst[2 * i] = cosf(-phase);
        dst[2 * i + 1] = sinf(-phase);
    }
}

void orc_translate_src(RFLOAT* dst, const RFLOAT* src, RFLOAT nTransCol, RFLOAT nTransRow, int nCol, int nRow,
                       const int* iCol, const int* iRow, int nPxl)
{
    RFLOAT rCol = nTransCol / nCol, rRow = nTransRow / nRow;
    for (int i = 0; i < nPxl; i++) {
        RFLOAT phase = (RFLOAT)(6.28318530717959 * (iCol[i] * rCol + iRow[i] * rRow));
        RFLOAT c = cosf(-phase), s = sinf(-phase);
        RFLOAT a0 = src[2 * i], a1 = src[2 * i + 1];
        dst[2 * i] = a0 * c - a1 * s;     /* Complex * Complex, include/Complex.h operator* */
        dst[2 * i + 1] = a0 * s + a1 * c;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* a15: rotate3D(quaternion) -- src/Geometry/Euler.cpp:181-189; dst column-major              */
/* ------------------------------------------------------------------------------------------ */
void orc_rotate3D(double* dst, const double* q)
{
    double A[3][3] = {{0, -q[3], q[2]}, {q[3], 0, -q[1]}, {-q[2], q[1], 0}};
    double AA[3][3];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += A[r][k] * A[k][c];
            AA[r][c] = s;
        }
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) dst[c * 3 + r] = (r == c ? 1.0 : 0.0) + 2 * q[0] * A[r][c] + 2 * AA[r][c];
}

/* ------------------------------------------------------------------------------------------ */
/* a7: trilinear gather / scatter on the half-Hermitian volume                                 */
/* conjHalf include/Image/Volume.h:135-147; WG_TRI_INTERP_LINEAR include/Functions/           */
/* Interpolation.h:152-200; getFTHalf(w,x0) src/Image/Volume.cpp:491-563 -- the box fast path   */
/* (x0[1]!=-1 && x0[2]!=-1) and the per-neighbour-wrap slow path visit the same 8 voxels in     */
/* the same order (k outer, j, i inner) with the same arithmetic, so one wrapped loop restates  */
/* both.                                                                                        */
/* ------------------------------------------------------------------------------------------ */
static inline size_t iFTHalf3_(long i, long j, long k, long P)
{
    long nColFT = P / 2 + 1;
    return (size_t)((k >= 0 ? k : k + P) * nColFT * P + (j >= 0 ? j : j + P) * nColFT + i);
}

static inline void tri_weights_(RFLOAT w[2][2][2], long x0[3], RFLOAT x, RFLOAT y, RFLOAT z)
{
    RFLOAT xs[3] = {x, y, z}, xd[3], v[3][2];
    for (int a = 0; a < 3; a++) {
        x0[a] = (long)floor(xs[a]);
        xd[a] = xs[a] - x0[a];
        v[a][0] = 1 - xd[a];
        v[a][1] = xd[a];
    }
    for (int k = 0; k < 2; k++)
        for (int j = 0; j < 2; j++)
            for (int i = 0; i < 2; i++) w[k][j][i] = v[0][i] * v[1][j] * v[2][k];
}

/* Volume::getByInterpolationFT (LINEAR_INTERP), src/Image/Volume.cpp:314-338 */
void orc_interp_ft(const RFLOAT* vol, int P, RFLOAT x, RFLOAT y, RFLOAT z, RFLOAT* out)
{
    int conj = 0;
    if (!(x >= 0)) { x *= -1; y *= -1; z *= -1; conj = 1; }
    RFLOAT w[2][2][2];
    long x0[3];
    tri_weights_(w, x0, x, y, z);
    RFLOAT re = 0, im = 0;
    for (int k = 0; k < 2; k++)
        for (int j = 0; j < 2; j++)
            for (int i = 0; i < 2; i++) {
                size_t idx = iFTHalf3_(x0[0] + i, x0[1] + j, x0[2] + k, P);
                RFLOAT tr = vol[2 * idx] * w[k][j][i], ti = vol[2 * idx + 1] * w[k][j][i];
                re = re + tr;
                im = im + ti;
            }
    out[0] = re;
    out[1] = conj ? -im : im;
}

/* a6: Projector::project(Complex*, dmat33, iCol, iRow, nPxl), src/Projector.cpp:356-374.
 * dvec3 oldCor = mat * dvec3(iCol*pf, iRow*pf, 0) in double, narrowed to RFLOAT at the call
 * to getByInterpolationFT(RFLOAT, ...) (include/Image/Volume.h:442). */
void orc_project(RFLOAT* dst, const RFLOAT* vol, int P, int pf, const double* mat, const int* iCol,
                 const int* iRow, int nPxl)
{
#pragma omp parallel for
    for (int i = 0; i < nPxl; i++) {
        double nx = (double)(iCol[i] * pf), ny = (double)(iRow[i] * pf), nz = 0;
        double ox = mat[0] * nx + mat[3] * ny + mat[6] * nz;
        double oy = mat[1] * nx + mat[4] * ny + mat[7] * nz;
        double oz = mat[2] * nx + mat[5] * ny + mat[8] * nz;
        orc_interp_ft(vol, P, (RFLOAT)ox, (RFLOAT)oy, (RFLOAT)oz, dst + 2 * i);
    }
}

/* Volume::addFT(Complex value, x, y, z) / addFT(RFLOAT value, ...), src/Image/Volume.cpp:340-375,
 * 565-712.  Single-threaded here, so the reference's `omp atomic` adds become plain adds in
 * pixel order (the reference's multi-thread order is non-deterministic). */
static inline void add_ft_(RFLOAT* F, RFLOAT* T, int P, RFLOAT vre, RFLOAT vim, RFLOAT tval, RFLOAT x, RFLOAT y,
                           RFLOAT z)
{
    if (!(x >= 0)) { x *= -1; y *= -1; z *= -1; vim = -vim; }
    RFLOAT w[2][2][2];
    long x0[3];
    tri_weights_(w, x0, x, y, z);
    for (int k = 0; k < 2; k++)
        for (int j = 0; j < 2; j++)
            for (int i = 0; i < 2; i++) {
                size_t idx = iFTHalf3_(x0[0] + i, x0[1] + j, x0[2] + k, P);
                if (F) {
                    F[2 * idx] += vre * w[k][j][i];
                    F[2 * idx + 1] += vim * w[k][j][i];
                }
                if (T) T[idx] += tval * w[k][j][i];
            }
}

/* a12: Reconstructor::insertP(src, ctf, rot, w, sig=NULL), src/Reconstructor.cpp:782-863
 * (RECONSTRUCTOR_TRILINEAR_KERNEL + RECONSTRUCTOR_ADD_T_DURING_INSERT, include/Config.h).
 * iCol/iRow here are the PADDED indices handed over by setPreCal (src/Optimiser.cpp:6741). */
void orc_insertP(RFLOAT* F, RFLOAT* T, int P, const RFLOAT* src, const RFLOAT* ctf, const double* rot, RFLOAT w,
                 const int* iColPad, const int* iRowPad, int nPxl)
{
    for (int i = 0; i < nPxl; i++) {
        int iCol = iColPad[i], iRow = iRowPad[i];
        double ox = rot[0] * iCol + rot[3] * iRow;
        double oy = rot[1] * iCol + rot[4] * iRow;
        double oz = rot[2] * iCol + rot[5] * iRow;
        /* src[i] * ctf[i] * (sig == NULL ? 1 : ...) * w, left to right (Complex*RFLOAT) */
        RFLOAT vre = src[2 * i] * ctf[i], vim = src[2 * i + 1] * ctf[i];
        vre = vre * 1.0f; vim = vim * 1.0f;
        vre = vre * w; vim = vim * w;
        RFLOAT tv = pow2f_(ctf[i]) * 1.0f * w;
        add_ft_(F, T, P, vre, vim, tv, (RFLOAT)ox, (RFLOAT)oy, (RFLOAT)oz);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* a8/a9: likelihood -- src/Optimiser.cpp:9187-9213 and :9931-9973                             */
/* ------------------------------------------------------------------------------------------ */
RFLOAT orc_logDataVSPrior(const RFLOAT* dat, const RFLOAT* pri, const RFLOAT* ctf, const RFLOAT* sigRcp, int m)
{
    RFLOAT result2 = 0.0;
    for (int i = 0; i < m; i++) {
        RFLOAT tmpReal = ctf[i] * pri[2 * i];
        RFLOAT tmpImag = ctf[i] * pri[2 * i + 1];
        RFLOAT tmp1Real = dat[2 * i] - tmpReal;
        RFLOAT tmp1Imag = dat[2 * i + 1] - tmpImag;
        RFLOAT tmp2 = tmp1Real * tmp1Real + tmp1Imag * tmp1Imag;
        result2 += (tmp2 * sigRcp[i]);
    }
    return result2;
}

/* pixel-major batch form: dat/ctf/sigRcp indexed [i*n + j] (i pixel, j image) */
void orc_logDataVSPrior_mn(const RFLOAT* dat, const RFLOAT* pri, const RFLOAT* ctf, const RFLOAT* sigRcp, int n, int m,
                           RFLOAT* result)
{
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) {
            int idx = i * n + j;
            RFLOAT a = ctf[idx] * pri[2 * i], b = ctf[idx] * pri[2 * i + 1];
            RFLOAT c = dat[2 * idx] - a, d = dat[2 * idx + 1] - b;
            RFLOAT t = c * c + d * d;
            result[j] += (t * sigRcp[idx]);
        }
}

/* same sum carried in double: NOT a reference function; the error yard-stick the parity tests
 * use to bound both the reference's float summation order and the GPU's tree reduction. */
double orc_logDataVSPrior_f64(const RFLOAT* dat, const RFLOAT* pri, const RFLOAT* ctf, const RFLOAT* sigRcp, int m)
{
    double r = 0;
    for (int i = 0; i < m; i++) {
        double a = (double)ctf[i] * pri[2 * i], b = (double)ctf[i] * pri[2 * i + 1];
        double c = dat[2 * i] - a, d = dat[2 * i + 1] - b;
        r += (c * c + d * d) * sigRcp[i];
    }
    return r;
}

/* ------------------------------------------------------------------------------------------ */
/* a10 (local): one particle-filter phase of one image -- src/Optimiser.cpp:1225-1406          */
/* Support points are inputs (Particle stays host-side): nR rotation matrices (column-major),   */
/* nT translations, nD defocus factors; priors pC (scalar, one class), pR, pT, pD (double, as   */
/* Particle::wR etc).  ctfP: [nD][nPxl] when cSearch (computed by the caller as :1263-1287),     */
/* else one row.  Outputs wC (1), wR, wT, wD (RFLOAT, Eigen vec) and the final baseLine.         */
/* logW (optional, [nR][nT][nD]) receives every logDataVSPrior value for diagnostics.           */
/* ------------------------------------------------------------------------------------------ */
void orc_expect_local(const RFLOAT* vol, int P, int pf, int N, const int* iCol, const int* iRow, int nPxl,
                      const RFLOAT* dat, const RFLOAT* ctfP, int cSearch, const RFLOAT* sigRcp, const double* rot,
                      int nR, const double* tran, int nT, int nD, double pC, const double* pR, const double* pT,
                      const double* pD, RFLOAT* wC, RFLOAT* wR, RFLOAT* wT, RFLOAT* wD, RFLOAT* baseLineOut,
                      RFLOAT* logW)
{
    RFLOAT* traP = (RFLOAT*)malloc((size_t)nT * nPxl * 2 * sizeof(RFLOAT));
    RFLOAT* priRotP = (RFLOAT*)malloc((size_t)nPxl * 2 * sizeof(RFLOAT));
    RFLOAT* priAllP = (RFLOAT*)malloc((size_t)nPxl * 2 * sizeof(RFLOAT));
    RFLOAT baseLine = NAN;
    wC[0] = 0;
    for (int i = 0; i < nR; i++) wR[i] = 0;
    for (int i = 0; i < nT; i++) wT[i] = 0;
    for (int i = 0; i < nD; i++) wD[i] = 0;

    for (int iT = 0; iT < nT; iT++) /* :1236-1250; t(0),t(1) are double, narrowed at the call */
        orc_translate(traP + (size_t)iT * nPxl * 2, (RFLOAT)tran[2 * iT], (RFLOAT)tran[2 * iT + 1], N, N, iCol, iRow,
                      nPxl);

    for (int iR = 0; iR < nR; iR++) {
        orc_project(priRotP, vol, P, pf, rot + 9 * iR, iCol, iRow, nPxl); /* :1303-1310 */
        for (int iT = 0; iT < nT; iT++) {
            const RFLOAT* tp = traP + (size_t)iT * nPxl * 2;
            for (int i = 0; i < nPxl; i++) { /* priAllP = traP * priRotP, :1319-1320 */
                RFLOAT a0 = tp[2 * i], a1 = tp[2 * i + 1], b0 = priRotP[2 * i], b1 = priRotP[2 * i + 1];
                priAllP[2 * i] = a0 * b0 - a1 * b1;
                priAllP[2 * i + 1] = a0 * b1 + a1 * b0;
            }
            for (int iD = 0; iD < nD; iD++) {
                const RFLOAT* ctf = cSearch ? ctfP + (size_t)iD * nPxl : ctfP;
                RFLOAT w = orc_logDataVSPrior(dat, priAllP, ctf, sigRcp, nPxl); /* :1366-1380 scalar form */
                if (logW) logW[((size_t)iR * nT + iT) * nD + iD] = w;
                baseLine = isnan(baseLine) ? w : baseLine; /* :1383 */
                if (w > baseLine) {                         /* :1385-1395 */
                    RFLOAT nf = expf(baseLine - w); /* exp(float) resolves to the float overload under libstdc++'s <math.h> */
                    wC[0] *= nf;
                    for (int q = 0; q < nR; q++) wR[q] *= nf;
                    for (int q = 0; q < nT; q++) wT[q] *= nf;
                    for (int q = 0; q < nD; q++) wD[q] *= nf;
                    baseLine = w;
                }
                RFLOAT s = expf(w - baseLine); /* :1397 */
                /* :1399-1402 -- RFLOAT * (double products) evaluated in double, += into RFLOAT */
                wC[0] = (RFLOAT)(wC[0] + s * (pR[iR] * pT[iT] * pD[iD]));
                wR[iR] = (RFLOAT)(wR[iR] + s * (pC * pT[iT] * pD[iD]));
                wT[iT] = (RFLOAT)(wT[iT] + s * (pC * pR[iR] * pD[iD]));
                wD[iD] = (RFLOAT)(wD[iD] + s * (pC * pR[iR] * pT[iT]));
            }
        }
    }
    if (baseLineOut) *baseLineOut = baseLine;
    free(traP); free(priRotP); free(priAllP);
}

/* ------------------------------------------------------------------------------------------ */
/* a10 (global): the scanning phase -- src/Optimiser.cpp:756-894 for one class index t.        */
/* datP/ctfP/sigRcpP pixel-major [pix][img]; slices rotP [nR][nPxl] (already projected) and     */
/* ramps traP [nT][nPxl] as the reference precomputes them (:710-724, :781).  wC [nImg][nK],    */
/* wR [nImg][nR] and wT [nImg][nT] are the matrices for THIS class; baseLine [nImg] is carried  */
/* across classes (NaN = unset).  pR [nImg][nR], pT [nImg][nT] priors (Particle::wR/wT).        */
/* The reference's thread interleaving is arbitrary; this is the sequential (m, n) order.       */
/* ------------------------------------------------------------------------------------------ */
void orc_expect_global(const RFLOAT* rotP, const RFLOAT* traP, const RFLOAT* datP, const RFLOAT* ctfP,
                       const RFLOAT* sigRcpP, int nImg, int nPxl, int nK, int kIdx, int nR, int nT,
                       const double* pR, const double* pT, RFLOAT* wC, RFLOAT* wRall, RFLOAT* wTall,
                       RFLOAT* baseLine)
{
    /* wRall [nK][nImg][nR], wTall [nK][nImg][nT]: rescaling touches every class (:849-853) */
    RFLOAT* priAllP = (RFLOAT*)malloc((size_t)nPxl * 2 * sizeof(RFLOAT));
    RFLOAT* dvp = (RFLOAT*)malloc((size_t)nImg * sizeof(RFLOAT));
    for (int m = 0; m < nR; m++) {
        const RFLOAT* pr = rotP + (size_t)m * nPxl * 2;
        for (int n = 0; n < nT; n++) {
            const RFLOAT* tp = traP + (size_t)n * nPxl * 2;
            for (int i = 0; i < nPxl; i++) {
                RFLOAT a0 = tp[2 * i], a1 = tp[2 * i + 1], b0 = pr[2 * i], b1 = pr[2 * i + 1];
                priAllP[2 * i] = a0 * b0 - a1 * b1;
                priAllP[2 * i + 1] = a0 * b1 + a1 * b0;
            }
            memset(dvp, 0, (size_t)nImg * sizeof(RFLOAT));
            orc_logDataVSPrior_mn(datP, priAllP, ctfP, sigRcpP, nImg, nPxl, dvp);
            for (int l = 0; l < nImg; l++) {
                if (isnan(baseLine[l]))
                    baseLine[l] = dvp[l];
                else if (dvp[l] > baseLine[l]) {
                    RFLOAT offset = dvp[l] - baseLine[l];
                    RFLOAT nf = expf(-offset);
                    for (int q = 0; q < nK; q++) wC[(size_t)l * nK + q] *= nf;
                    for (int td = 0; td < nK; td++) {
                        RFLOAT* a = wRall + ((size_t)td * nImg + l) * nR;
                        RFLOAT* b = wTall + ((size_t)td * nImg + l) * nT;
                        for (int q = 0; q < nR; q++) a[q] *= nf;
                        for (int q = 0; q < nT; q++) b[q] *= nf;
                    }
                    baseLine[l] += offset;
                }
                RFLOAT w = expf(dvp[l] - baseLine[l]);
                double prm = pR[(size_t)l * nR + m], ptn = pT[(size_t)l * nT + n];
                wC[(size_t)l * nK + kIdx] = (RFLOAT)(wC[(size_t)l * nK + kIdx] + w * (prm * ptn));
                RFLOAT* a = wRall + ((size_t)kIdx * nImg + l) * nR;
                RFLOAT* b = wTall + ((size_t)kIdx * nImg + l) * nT;
                a[m] = (RFLOAT)(a[m] + w * ptn);
                b[n] = (RFLOAT)(b[n] + w * prm);
            }
        }
    }
    free(priAllP); free(dvp);
}

/* ------------------------------------------------------------------------------------------ */
/* a13: prepareTF pieces                                                                        */
/* ------------------------------------------------------------------------------------------ */
/* allReduceT tail, RECONSTRUCTOR_NORMALISE_T_F: sf = 1.0 / REAL(T[0]); SCALE_FT(T), SCALE_FT(F)
 * (src/Reconstructor.cpp:2455-2476; `RFLOAT sf = 1.0 / REAL(_T3D[0])` is a double division narrowed) */
void orc_normalise_TF(RFLOAT* F, RFLOAT* T, int P)
{
    size_t n = (size_t)P * P * (P / 2 + 1);
    RFLOAT sf = (RFLOAT)(1.0 / T[0]);
    for (size_t i = 0; i < n; i++) T[i] = T[i] * sf;
    for (size_t i = 0; i < 2 * n; i++) F[i] = F[i] * sf;
}

/* SYMMETRIZE_FT(dst = src + sum_s VOL_TRANSFORM_MAT_FT(src, R_s, r, LINEAR)),
 * include/Geometry/Transformation.h:105-131,170-194; called with r = maxRadius*pf + 1
 * (src/Reconstructor.cpp:2676-2690).  symMat: nSym column-major 3x3.  isComplex selects F / T. */
void orc_symmetrize(RFLOAT* dst, const RFLOAT* src, int P, int isComplex, const double* symMat, int nSym, double r)
{
    size_t n = (size_t)P * P * (P / 2 + 1);
    size_t nf = isComplex ? 2 * n : n;
    RFLOAT* se = (RFLOAT*)malloc(nf * sizeof(RFLOAT));
    RFLOAT* tmpc = NULL;
    if (!isComplex) { /* interpolate T through the complex path with a zero imaginary part */
        tmpc = (RFLOAT*)calloc(2 * n, sizeof(RFLOAT));
        for (size_t i = 0; i < n; i++) tmpc[2 * i] = src[i];
    }
    const RFLOAT* srcc = isComplex ? src : tmpc;
    RFLOAT* result = (RFLOAT*)malloc(nf * sizeof(RFLOAT));
    memcpy(result, src, nf * sizeof(RFLOAT));
    for (int s = 0; s < nSym; s++) {
        const double* mat = symMat + 9 * s;
        memset(se, 0, nf * sizeof(RFLOAT));
#pragma omp parallel for
        for (long k = -P / 2; k < P / 2; k++)
            for (long j = -P / 2; j < P / 2; j++)
                for (long i = 0; i <= P / 2; i++) {
                    double nx = (double)i, ny = (double)j, nz = (double)k;
                    double ox = mat[0] * nx + mat[3] * ny + mat[6] * nz;
                    double oy = mat[1] * nx + mat[4] * ny + mat[7] * nz;
                    double oz = mat[2] * nx + mat[5] * ny + mat[8] * nz;
                    if (ox * ox + oy * oy + oz * oz < r * r) {
                        RFLOAT o[2];
                        orc_interp_ft(srcc, P, (RFLOAT)ox, (RFLOAT)oy, (RFLOAT)oz, o);
                        size_t idx = iFTHalf3_(i, j, k, P);
                        if (isComplex) { se[2 * idx] = o[0]; se[2 * idx + 1] = o[1]; }
                        else se[idx] = o[0];
                    }
                }
        for (size_t i = 0; i < nf; i++) result[i] = result[i] + se[i]; /* ADD_FT */
    }
    memcpy(dst, result, nf * sizeof(RFLOAT));
    free(se); free(result); if (tmpc) free(tmpc);
}

/* ------------------------------------------------------------------------------------------ */
/* a14: reconstruct() element-wise stages, src/Reconstructor.cpp:1129-1831 (MODE_3D)           */
/* ------------------------------------------------------------------------------------------ */
/* [MAP] T /= FSC'(shell), :1242-1270.  FSC vector has nFSC entries (Eigen vec of RFLOAT). */
void orc_wiener_T(RFLOAT* T, int P, int pf, int maxRadius, const RFLOAT* FSC, int nFSC, int joinHalf)
{
#pragma omp parallel for
    for (long k = -P / 2; k < P / 2; k++)
        for (long j = -P / 2; j < P / 2; j++)
            for (long i = 0; i <= P / 2; i++) {
                double q = (double)i * i + (double)j * j + (double)k * k;
                if ((q >= pow2f_((RFLOAT)(5 * pf))) && (q < pow2f_((RFLOAT)(maxRadius * pf)))) {
                    int u = AROUND_(gsl_hypot3_((double)i, (double)j, (double)k));
                    RFLOAT f = (u / pf >= nFSC) ? 0 : FSC[u / pf];
                    RFLOAT lo = (RFLOAT)1e-3, hi = (RFLOAT)(1 - 1e-3);
                    RFLOAT mn = hi < f ? hi : f;       /* TSGSL_MIN_RFLOAT(FSC_BASE_H, FSC) */
                    f = lo > mn ? lo : mn;             /* TSGSL_MAX_RFLOAT(FSC_BASE_L, .) */
                    if (joinHalf) f = (RFLOAT)sqrt((double)(2 * f / (1 + f)));
                    size_t idx = iFTHalf3_(i, j, k, P);
                    T[idx] = T[idx] / f;
                }
            }
}

/* W = 1 inside |k| < maxRadius*pf else 0 (:1299-1304); T = max(T, 1e-25) (:1322-1324) */
void orc_init_W_floor_T(RFLOAT* W, RFLOAT* T, int P, int pf, int maxRadius)
{
    size_t n = (size_t)P * P * (P / 2 + 1);
    for (long k = -P / 2; k < P / 2; k++)
        for (long j = -P / 2; j < P / 2; j++)
            for (long i = 0; i <= P / 2; i++) {
                double q = (double)i * i + (double)j * j + (double)k * k;
                W[iFTHalf3_(i, j, k, P)] = (q < pow2f_((RFLOAT)(maxRadius * pf))) ? 1.0f : 0.0f;
            }
    for (size_t i = 0; i < n; i++) T[i] = T[i] > (RFLOAT)1e-25 ? T[i] : (RFLOAT)1e-25;
}

/* C = T * REAL(W) (:1389-1391); C is complex with imaginary part T.im*W = 0 */
void orc_calc_C(RFLOAT* C, const RFLOAT* T, const RFLOAT* W, int P)
{
    size_t n = (size_t)P * P * (P / 2 + 1);
    for (size_t i = 0; i < n; i++) { C[2 * i] = T[i] * W[i]; C[2 * i + 1] = 0.0f * W[i]; }
}

/* convoluteC real-space stage (:2635-2652): C_RL(i,j,k) *= kernelRL(QUAD_3/pow2(N*pf)) / nf.
 * crl is the real P^3 volume [k][j][i] (wrapped indices, Volume::iRL include/Image/Volume.h:520-528).
 * NP = _N * _pf. */
void orc_convolute_rl(RFLOAT* crl, int P, int NP, const RFLOAT* tab, int ntab, RFLOAT nf)
{
#pragma omp parallel for
    for (long k = -P / 2; k < P / 2; k++)
        for (long j = -P / 2; j < P / 2; j++)
            for (long i = -P / 2; i < P / 2; i++) {
                size_t idx = (size_t)((k >= 0 ? k : k + P) * (long)P * P + (j >= 0 ? j : j + P) * (long)P +
                                      (i >= 0 ? i : i + P));
                double q = (double)i * i + (double)j * j + (double)k * k;
                RFLOAT x = (RFLOAT)(q / pow2f_((RFLOAT)NP));
                crl[idx] = crl[idx] * tab_lookup_(tab, ntab, x) / nf;
            }
}

/* ts_hypot, include/Complex.h (ABS) */
static inline RFLOAT ts_hypot_(RFLOAT x, RFLOAT y)
{
    RFLOAT xabs = fabsf(x), yabs = fabsf(y), min, max;
    if (xabs < yabs) { min = xabs; max = yabs; } else { min = yabs; max = xabs; }
    if (min == 0) return max;
    RFLOAT u = min / max;
    return max * sqrtf(1 + u * u);
}

/* W /= max(ABS(C), 1e-6) inside the sphere (:1487-1496), then checkC (RECONSTRUCTOR_CHECK_C_MAX,
 * :2563-2592): max over the sphere of | ABS(C) - 1 |.  Returns diffC. */
RFLOAT orc_update_W_checkC(RFLOAT* W, const RFLOAT* C, int P, int pf, int maxRadius)
{
    RFLOAT diff = 0;
    for (long k = -P / 2; k < P / 2; k++)
        for (long j = -P / 2; j < P / 2; j++)
            for (long i = 0; i <= P / 2; i++) {
                double q = (double)i * i + (double)j * j + (double)k * k;
                if (q < pow2f_((RFLOAT)(maxRadius * pf))) {
                    size_t idx = iFTHalf3_(i, j, k, P);
                    RFLOAT a = ts_hypot_(C[2 * idx], C[2 * idx + 1]);
                    RFLOAT m = a > (RFLOAT)1e-6 ? a : (RFLOAT)1e-6;
                    W[idx] = W[idx] / m;
                    RFLOAT d = (RFLOAT)fabs((double)(a - 1));
                    if (d > diff) diff = d;
                }
            }
    return diff;
}

/* no-grid-correction branch: W = 1 / max(ABS(T), 1e-6) inside the sphere (:1566-1578) */
void orc_W_nogridcorr(RFLOAT* W, const RFLOAT* T, int P, int pf, int maxRadius)
{
    for (long k = -P / 2; k < P / 2; k++)
        for (long j = -P / 2; j < P / 2; j++)
            for (long i = 0; i <= P / 2; i++) {
                double q = (double)i * i + (double)j * j + (double)k * k;
                if (q < pow2f_((RFLOAT)(maxRadius * pf))) {
                    size_t idx = iFTHalf3_(i, j, k, P);
                    RFLOAT a = ts_hypot_(T[idx], 0.0f);
                    W[idx] = (RFLOAT)(1.0 / (a > (RFLOAT)1e-6 ? a : (RFLOAT)1e-6));
                }
            }
}

/* padDst = F * W inside the sphere, zero elsewhere (:1678-1701). padDst has the same PAD size
 * here (the reference allocates (_N*_pf)^3; equal to PAD_SIZE when _size == _N). */
void orc_FW(RFLOAT* padDst, const RFLOAT* F, const RFLOAT* W, int P, int pf, int maxRadius)
{
    size_t n = (size_t)P * P * (P / 2 + 1);
    memset(padDst, 0, 2 * n * sizeof(RFLOAT));
    for (long k = -P / 2; k < P / 2; k++)
        for (long j = -P / 2; j < P / 2; j++)
            for (long i = 0; i <= P / 2; i++) {
                double q = (double)i * i + (double)j * j + (double)k * k;
                if (q < pow2f_((RFLOAT)(maxRadius * pf))) {
                    size_t idx = iFTHalf3_(i, j, k, P);
                    /* Complex * Complex with W = (w, 0) */
                    RFLOAT a0 = F[2 * idx], a1 = F[2 * idx + 1], b0 = W[idx], b1 = 0.0f;
                    padDst[2 * idx] = a0 * b0 - a1 * b1;
                    padDst[2 * idx + 1] = a0 * b1 + a1 * b0;
                }
            }
}

/* the same with Reconstructor::resizeSpace in effect (_size < _N, src/Reconstructor.cpp:184-198): F and W live on the PF = _pf *
 * _size grid, padDst is the (_N * _pf)^3 half grid PN (:1677-1701: `Volume padDst(_N * _pf, ...)`, VOLUME_FOR_EACH_PIXEL_FT(_F3D)
 * walks the SMALL grid and padDst.setFTHalf(.., i, j, k) places the voxel at the same integer frequency of the large one). */
void orc_FW_pad(RFLOAT* padDst, int PN, const RFLOAT* F, const RFLOAT* W, int PF, int pf, int maxRadius)
{
    size_t n = (size_t)PN * PN * (PN / 2 + 1);
    memset(padDst, 0, 2 * n * sizeof(RFLOAT));
    for (long k = -PF / 2; k < PF / 2; k++)
        for (long j = -PF / 2; j < PF / 2; j++)
            for (long i = 0; i <= PF / 2; i++) {
                double q = (double)i * i + (double)j * j + (double)k * k;
                if (q < pow2f_((RFLOAT)(maxRadius * pf))) {
                    size_t idx = iFTHalf3_(i, j, k, PF), odx = iFTHalf3_(i, j, k, PN);
                    RFLOAT a0 = F[2 * idx], a1 = F[2 * idx + 1], b0 = W[idx], b1 = 0.0f;
                    padDst[2 * odx] = a0 * b0 - a1 * b1;
                    padDst[2 * odx + 1] = a0 * b1 + a1 * b0;
                }
            }
}

/* VOL_EXTRACT_RL (include/Image/ImageFunctions.h:51-64) followed by the TIK correction
 * (RECONSTRUCTOR_CORRECT_CONVOLUTION_KERNEL, :1781-1802): dst(i,j,k) = pad(i,j,k) /
 * TIK_RL(NORM_3(i,j,k) / (pf * N)).  pad: real P^3, dst: real N^3, both wrapped-index layout. */
void orc_extract_tik(RFLOAT* dst, const RFLOAT* pad, int P, int N, int pf, int corr)
{
    for (long k = -N / 2; k < N / 2; k++)
        for (long j = -N / 2; j < N / 2; j++)
            for (long i = -N / 2; i < N / 2; i++) {
                size_t ip = (size_t)((k >= 0 ? k : k + P) * (long)P * P + (j >= 0 ? j : j + P) * (long)P +
                                     (i >= 0 ? i : i + P));
                size_t id = (size_t)((k >= 0 ? k : k + N) * (long)N * N + (j >= 0 ? j : j + N) * (long)N +
                                     (i >= 0 ? i : i + N));
                RFLOAT v = pad[ip];
                if (corr) v = v / orc_TIK_RL((RFLOAT)(gsl_hypot3_((double)i, (double)j, (double)k) / (pf * N)));
                dst[id] = v;
            }
}

/* a5: Projector::setProjectee real-space stages: VOL_PAD_RL (include/Image/ImageFunctions.h:
 * 176-192) + gridCorrection LINEAR branch (src/Projector.cpp:573-583): pad(i,j,k) = src(i,j,k) /
 * TIK_RL(NORM_3(i,j,k) / (pf * P)) for the N^3 core, zero elsewhere.  NB the divisor uses the
 * PADDED size times pf.  Every padded voxel is divided (zeros stay zero). */
void orc_pad_gridcorr(RFLOAT* pad, const RFLOAT* src, int N, int pf)
{
    int P = N * pf;
    memset(pad, 0, (size_t)P * P * P * sizeof(RFLOAT));
    for (long k = -N / 2; k < N / 2; k++)
        for (long j = -N / 2; j < N / 2; j++)
            for (long i = -N / 2; i < N / 2; i++) {
                size_t ip = (size_t)((k >= 0 ? k : k + P) * (long)P * P + (j >= 0 ? j : j + P) * (long)P +
                                     (i >= 0 ? i : i + P));
                size_t is = (size_t)((k >= 0 ? k : k + N) * (long)N * N + (j >= 0 ? j : j + N) * (long)N +
                                     (i >= 0 ? i : i + N));
                pad[ip] = src[is] / orc_TIK_RL((RFLOAT)(gsl_hypot3_((double)i, (double)j, (double)k) / (pf * P)));
            }
}

/* ------------------------------------------------------------------------------------------ */
/* a16: FSC(A, B) -- src/Functions/Spectrum.cpp:302-337 (Eigen vec = RFLOAT accumulators;       */
/* sequential order here, the reference's omp-atomic order is arbitrary)                        */
/* ------------------------------------------------------------------------------------------ */
void orc_fsc(RFLOAT* dst, int nShell, const RFLOAT* A, const RFLOAT* B, int P)
{
    RFLOAT* vS = (RFLOAT*)calloc(nShell, sizeof(RFLOAT));
    RFLOAT* vA = (RFLOAT*)calloc(nShell, sizeof(RFLOAT));
    RFLOAT* vB = (RFLOAT*)calloc(nShell, sizeof(RFLOAT));
    for (long k = -P / 2; k < P / 2; k++)
        for (long j = -P / 2; j < P / 2; j++)
            for (long i = 0; i <= P / 2; i++) {
                int u = AROUND_(gsl_hypot3_((double)i, (double)j, (double)k));
                if (u < nShell) {
                    size_t idx = iFTHalf3_(i, j, k, P);
                    RFLOAT a0 = A[2 * idx], a1 = A[2 * idx + 1], b0 = B[2 * idx], b1 = -B[2 * idx + 1];
                    vS[u] += a0 * b0 - a1 * b1; /* REAL(A * CONJUGATE(B)) */
                    vA[u] += A[2 * idx] * A[2 * idx] + A[2 * idx + 1] * A[2 * idx + 1];
                    vB[u] += B[2 * idx] * B[2 * idx] + B[2 * idx + 1] * B[2 * idx + 1];
                }
            }
    for (int i = 0; i < nShell; i++) {
        RFLOAT AB = (RFLOAT)sqrt((double)(vA[i] * vB[i]));
        dst[i] = (AB == 0) ? 0 : vS[i] / AB;
    }
    free(vS); free(vA); free(vB);
}

/* ------------------------------------------------------------------------------------------ */
/* CPU-baseline driver (bench.py cpu_baseline leg): the fixed-work iteration of SURVEY 8(d) for  */
/* a block of particles, OpenMP over images exactly as HOT LOOP B / HOT LOOP C                   */
/* (src/Optimiser.cpp:1162, :7038).  Returns nothing; the caller times it.                       */
/* Reconstructor::insertP (src/Reconstructor.cpp:782-863) with the reference's `omp atomic` adds (src/Image/Volume.cpp:565-712):
 * what several threads sharing one F / T run (the CPU baseline and its per-op rate) */
static void insertP_atomic_(RFLOAT* F, RFLOAT* T, int P, const RFLOAT* src, const RFLOAT* ctf, const double* R, RFLOAT w,
                            const int* iColPad, const int* iRowPad, int nPxl)
{
    for (int i = 0; i < nPxl; i++) {
        int ic = iColPad[i], ir = iRowPad[i];
        double ox = R[0] * ic + R[3] * ir, oy = R[1] * ic + R[4] * ir, oz = R[2] * ic + R[5] * ir;
        RFLOAT c = ctf[i];
        RFLOAT vre = src[2 * i] * c * 1.0f * w, vim = src[2 * i + 1] * c * 1.0f * w;
        RFLOAT tv = pow2f_(c) * 1.0f * w;
        RFLOAT x = (RFLOAT)ox, y = (RFLOAT)oy, z = (RFLOAT)oz;
        if (!(x >= 0)) { x = -x; y = -y; z = -z; vim = -vim; }
        RFLOAT wt[2][2][2];
        long x0[3];
        tri_weights_(wt, x0, x, y, z);
        for (int kk = 0; kk < 2; kk++)
            for (int jj = 0; jj < 2; jj++)
                for (int ii = 0; ii < 2; ii++) {
                    size_t idx = iFTHalf3_(x0[0] + ii, x0[1] + jj, x0[2] + kk, P);
                    RFLOAT a = vre * wt[kk][jj][ii], b = vim * wt[kk][jj][ii], t = tv * wt[kk][jj][ii];
#pragma omp atomic
                    F[2 * idx] += a;
#pragma omp atomic
                    F[2 * idx + 1] += b;
#pragma omp atomic
                    T[idx] += t;
                }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* nGroups > 1: the threads are dealt to nGroups groups, each with its OWN pair of accumulators F [g][vol][2], T [g][vol]
 * (threads of a group share theirs under `omp atomic`) -- the reference's deployment: several MPI ranks per node, each an
 * OpenMP team with private _F3D / _T3D, summed by MPI_Allreduce_Large afterwards (src/Parallel.cpp:26-36,
 * src/Reconstructor.cpp:2383,2436; the caller adds the groups up).  nGroups = 1: one team, one pair. */
#include <omp.h>
void orc_baseline_block_groups(const RFLOAT* vol, int P, int pf, int N, const int* iCol, const int* iRow, const int* iColPad,
                               const int* iRowPad, int nPxl, int nImg, const RFLOAT* dat, const RFLOAT* ctf,
                               const RFLOAT* sigRcp, const double* rot /*[nImg][nPhase][nR][9]*/, const double* tran
                               /*[nImg][nPhase][nT][2]*/, int nPhase, int nR, int nT, const double* recoRot /*[nImg][mReco][9]*/,
                               const double* recoTran /*[nImg][mReco][2]*/, int mReco, RFLOAT* Fall, RFLOAT* Tall, RFLOAT* wRout,
                               int nGroups)
{
    const size_t volN = (size_t)P * P * (P / 2 + 1);
    if (nGroups < 1) nGroups = 1;
#pragma omp parallel
  {
    RFLOAT* F = Fall + (size_t)(omp_get_thread_num() % nGroups) * volN * 2;
    RFLOAT* T = Tall + (size_t)(omp_get_thread_num() % nGroups) * volN;
#pragma omp for schedule(dynamic)
    for (int l = 0; l < nImg; l++) {
        double* pR = (double*)malloc(nR * sizeof(double));
        double* pT = (double*)malloc(nT * sizeof(double));
        double pD = 1.0;
        for (int i = 0; i < nR; i++) pR[i] = 1.0;
        for (int i = 0; i < nT; i++) pT[i] = 1.0;
        RFLOAT wC, wD, base;
        RFLOAT* wR = (RFLOAT*)malloc(nR * sizeof(RFLOAT));
        RFLOAT* wT = (RFLOAT*)malloc(nT * sizeof(RFLOAT));
        for (int ph = 0; ph < nPhase; ph++) {
            orc_expect_local(vol, P, pf, N, iCol, iRow, nPxl, dat + (size_t)l * nPxl * 2, ctf + (size_t)l * nPxl, 0,
                             sigRcp + (size_t)l * nPxl, rot + ((size_t)l * nPhase + ph) * nR * 9, nR,
                             tran + ((size_t)l * nPhase + ph) * nT * 2, nT, 1, 1.0, pR, pT, &pD, &wC, wR, wT, &wD,
                             &base, NULL);
        }
        if (wRout) memcpy(wRout + (size_t)l * nR, wR, nR * sizeof(RFLOAT));
        RFLOAT* transImg = (RFLOAT*)malloc((size_t)nPxl * 2 * sizeof(RFLOAT));
        RFLOAT w = 1.0f / mReco; /* w = 1; w /= mReco (src/Optimiser.cpp:7051-7056) */
        for (int m = 0; m < mReco; m++) {
            const double* tr = recoTran + ((size_t)l * mReco + m) * 2;
            orc_translate_src(transImg, dat + (size_t)l * nPxl * 2, (RFLOAT)(-tr[0]), (RFLOAT)(-tr[1]), N, N, iCol, iRow,
                              nPxl);
            /* insertP with atomic adds (the reference's `omp atomic`) */
            insertP_atomic_(F, T, P, transImg, ctf + (size_t)l * nPxl, recoRot + ((size_t)l * mReco + m) * 9, w, iColPad, iRowPad, nPxl);
        }
        free(transImg); free(pR); free(pT); free(wR); free(wT);
    }
  }
}

void orc_baseline_block(const RFLOAT* vol, int P, int pf, int N, const int* iCol, const int* iRow, const int* iColPad,
                        const int* iRowPad, int nPxl, int nImg, const RFLOAT* dat, const RFLOAT* ctf,
                        const RFLOAT* sigRcp, const double* rot, const double* tran, int nPhase, int nR, int nT,
                        const double* recoRot, const double* recoTran, int mReco, RFLOAT* F, RFLOAT* T, RFLOAT* wRout)
{
    orc_baseline_block_groups(vol, P, pf, N, iCol, iRow, iColPad, iRowPad, nPxl, nImg, dat, ctf, sigRcp, rot, tran, nPhase, nR, nT,
                              recoRot, recoTran, mReco, F, T, wRout, 1);
}

/* ========================================================================================== */
/* SURVEY 8 "next" rows f1/f2: re-centre, re-mask, sigma update                               */
/* ========================================================================================== */

/* f2: softMask(Image& mask, r, ew), src/Functions/Mask.cpp:334-350.  mask is [N][N] real with the
 * wrap-around indexing of Image::iRL (include/Image/Image.h:388-394). */
void orc_soft_mask(RFLOAT* mask, int N, RFLOAT r, RFLOAT ew)
{
    for (long j = -N / 2; j < N / 2; j++)
        for (long i = -N / 2; i < N / 2; i++) {
            RFLOAT u = (RFLOAT)gsl_hypot_((double)i, (double)j);
            size_t idx = (size_t)(j >= 0 ? j : j + N) * N + (size_t)(i >= 0 ? i : i + N);
            if (u > r + ew) mask[idx] = 0;
            else if (u >= r) mask[idx] = (RFLOAT)(0.5 + 0.5 * cos((u - r) / ew * M_PI));
            else mask[idx] = 1;
        }
}

/* f2: the real-space part of Optimiser::reMaskImg, src/Optimiser.cpp:6133-6141: the 1/size scale of
 * FFT::bwExecutePlan (src/FFT.cpp:353-354, SCALE_RL multiplies a float by a double) followed by
 * MUL_RL(img, mask) (include/Image/ImageBase.h:178-180).  The two FFTs either side are done by the
 * caller (oracle.py uses scipy.fft in float32; the reference uses FFTW). */
void orc_scale_mul_rl(RFLOAT* rl, const RFLOAT* mask, size_t n)
{
    double s = 1.0 / (double)n;
    for (size_t i = 0; i < n; i++) {
        rl[i] = (RFLOAT)(rl[i] * s);
        rl[i] *= mask[i];
    }
}

/* f2: translate(Image& dst, const Image& src, [r,] tx, ty), src/Image/ImageFunctions.cpp:269-284
 * (whole image; Optimiser::reCentreImg, src/Optimiser.cpp:6078-6082) and :322-339 (inside radius r;
 * r < 0 here selects the whole-image form).  Pixels outside r are not written. */
void orc_translate_image(RFLOAT* dst, const RFLOAT* src, int N, RFLOAT r, RFLOAT nTransCol, RFLOAT nTransRow)
{
    RFLOAT rCol = nTransCol / N, rRow = nTransRow / N;
    int nc = N / 2 + 1;
    for (long j = -N / 2; j < N / 2; j++)
        for (long i = 0; i <= N / 2; i++) {
            if (r >= 0 && !((double)i * (double)i + (double)j * (double)j < pow2f_(r))) continue;
            RFLOAT phase = (RFLOAT)(6.28318530717959 * (i * rCol + j * rRow));
            RFLOAT c = cosf(-phase), s = sinf(-phase);
            size_t idx = (size_t)(j >= 0 ? j : j + N) * nc + i;
            RFLOAT a0 = src[2 * idx], a1 = src[2 * idx + 1];
            dst[2 * idx] = a0 * c - a1 * s;
            dst[2 * idx + 1] = a0 * s + a1 * c;
        }
}

/* f3: translate(Volume& dst, const Volume& src, r, tx, ty, tz), src/Image/ImageFunctions.cpp:363-384
 * (reference re-centring in Optimiser::reconstructRef, src/Optimiser.cpp:7418-7428 / TranslateI,
 * gpu/interface/Interface.h:510). */
void orc_translate_volume(RFLOAT* dst, const RFLOAT* src, int P, RFLOAT r, RFLOAT tx, RFLOAT ty, RFLOAT tz)
{
    RFLOAT rCol = tx / P, rRow = ty / P, rSlc = tz / P;
    for (long k = -P / 2; k < P / 2; k++)
        for (long j = -P / 2; j < P / 2; j++)
            for (long i = 0; i <= P / 2; i++) {
                if (!((double)i * i + (double)j * j + (double)k * k < pow2f_(r))) continue;
                RFLOAT phase = (RFLOAT)(6.28318530717959 * (i * rCol + j * rRow + k * rSlc));
                RFLOAT c = cosf(-phase), s = sinf(-phase);
                size_t idx = iFTHalf3_(i, j, k, P);
                RFLOAT a0 = src[2 * idx], a1 = src[2 * idx + 1];
                dst[2 * idx] = a0 * c - a1 * s;
                dst[2 * idx + 1] = a0 * s + a1 * c;
            }
}

/* f1: the pixel set shared by Projector::project(Image&, mat) (src/Projector.cpp:276-294:
 * IMAGE_FOR_PIXEL_R_FT(r) with QUAD < r*r) and powerSpectrum (src/Functions/Spectrum.cpp:171-184).
 * Unlike allocPreCalIdx it keeps (0, j<0).  Returns the count; arrays may be NULL. */
int orc_disc_list(int N, int r, int* iCol, int* iRow, int* iPxl, int* iSig)
{
    int n = 0;
    for (long j = -r; j < r; j++)
        for (long i = 0; i <= r; i++) {
            if (!((double)i * i + (double)j * j < (double)r * r)) continue;
            if (iCol) iCol[n] = (int)i;
            if (iRow) iRow[n] = (int)j;
            if (iPxl) iPxl[n] = (int)((j >= 0 ? j : j + N) * (N / 2 + 1) + i);
            if (iSig) iSig[n] = AROUND_(gsl_hypot_((double)i, (double)j));
            n++;
        }
    return n;
}

/* powerSpectrum(vec&, const Image&, r, 1), src/Functions/Spectrum.cpp:161-190 (serial order). */
static void power_spectrum_(RFLOAT* dst, const RFLOAT* img, int N, int r)
{
    unsigned* cnt = (unsigned*)calloc(r, sizeof(unsigned));
    int nc = N / 2 + 1;
    for (int i = 0; i < r; i++) dst[i] = 0;
    for (long j = -N / 2; j < N / 2; j++)
        for (long i = 0; i <= N / 2; i++)
            if ((double)i * i + (double)j * j < pow2f_((RFLOAT)r)) {
                int u = AROUND_(gsl_hypot_((double)i, (double)j));
                if (u < r) {
                    size_t idx = (size_t)(j >= 0 ? j : j + N) * nc + i;
                    dst[u] += img[2 * idx] * img[2 * idx] + img[2 * idx + 1] * img[2 * idx + 1]; /* ABS2 */
                    cnt[u] += 1;
                }
            }
    for (int i = 0; i < r; i++) dst[i] /= cnt[i];
    free(cnt);
}
void orc_power_spectrum(RFLOAT* dst, const RFLOAT* img, int N, int r) { power_spectrum_(dst, img, N, r); }

/* f1: per-image part of Optimiser::allReduceSigma, src/Optimiser.cpp:6443-6565, as configured by
 * include/Config.h (OPTIMISER_SIGMA_RANK1ST: one draw = the top pose; OPTIMISER_SIGMA_WHOLE_FREQUENCY:
 * rSig = size/2-1; OPTIMISER_RECENTRE_IMAGE_EACH_ITERATION; OPTIMISER_CTF_ON_THE_FLY; w = 1).
 * projR = the Projector's _maxRadius (Model sets it to _r, src/Model.cpp:1042).
 * Outputs four vec(rSig): sSVD, dSVD, vSigM, vSigN. */
void orc_sigma_image(const RFLOAT* vol, int P, int pf, int N, int projR, int rSig, const double* rot,
                     const double* tran, const double* offset, RFLOAT pixelSize, RFLOAT voltage, RFLOAT defocusU,
                     RFLOAT defocusV, RFLOAT theta, RFLOAT Cs, RFLOAT amplitudeContrast, RFLOAT phaseShift,
                     const RFLOAT* img, const RFLOAT* imgOri, RFLOAT* sSVD, RFLOAT* dSVD, RFLOAT* vSigM,
                     RFLOAT* vSigN)
{
    size_t nFT = (size_t)N * (N / 2 + 1);
    RFLOAT* imgM = (RFLOAT*)calloc(nFT * 2, sizeof(RFLOAT)); /* SET_0_FT */
    RFLOAT* imgN = (RFLOAT*)calloc(nFT * 2, sizeof(RFLOAT));
    int cap = (2 * projR + 1) * (projR + 1);
    int* iCol = (int*)malloc(cap * sizeof(int));
    int* iRow = (int*)malloc(cap * sizeof(int));
    int* iPxl = (int*)malloc(cap * sizeof(int));
    int n = orc_disc_list(N, projR, iCol, iRow, iPxl, NULL);
    RFLOAT* prj = (RFLOAT*)malloc((size_t)n * 2 * sizeof(RFLOAT));
    /* project(imgM, rot3D, tran): slice then translate(dst, dst, _maxRadius, t0, t1), src/Projector.cpp:456-464 */
    orc_project(prj, vol, P, pf, rot, iCol, iRow, n);
    for (int p = 0; p < n; p++) { imgM[2 * iPxl[p]] = prj[2 * p]; imgM[2 * iPxl[p] + 1] = prj[2 * p + 1]; }
    memcpy(imgN, imgM, nFT * 2 * sizeof(RFLOAT));
    orc_translate_image(imgM, imgM, N, (RFLOAT)projR, (RFLOAT)tran[0], (RFLOAT)tran[1]);
    {
        double t0 = tran[0] - (offset ? offset[0] : 0.0), t1 = tran[1] - (offset ? offset[1] : 0.0);
        orc_translate_image(imgN, imgN, N, (RFLOAT)projR, (RFLOAT)t0, (RFLOAT)t1);
    }
    free(iCol); free(iRow); free(iPxl); free(prj);
    /* CTF(ctf, ..., CEIL(rSig) + 1, 1), src/CTF.cpp:68-111, then imgM[i] *= REAL(ctf[i]) */
    {
        int rc = rSig + 1;
        int capc = (2 * rc + 3) * (rc + 2);
        int* cCol = (int*)malloc(capc * sizeof(int));
        int* cRow = (int*)malloc(capc * sizeof(int));
        int* cPxl = (int*)malloc(capc * sizeof(int));
        int m = 0;
        for (long j = -(rc + 1); j < (rc + 1); j++) /* IMAGE_FOR_PIXEL_R_FT(r + 1) */
            for (long i = 0; i <= (rc + 1); i++) {
                RFLOAT v = (RFLOAT)((double)i * i + (double)j * j);
                if (v < pow2f_((RFLOAT)rc) && j >= -N / 2 && j < N / 2 && i <= N / 2) {
                    cCol[m] = (int)i; cRow[m] = (int)j; cPxl[m] = (int)((j >= 0 ? j : j + N) * (N / 2 + 1) + i); m++;
                }
            }
        RFLOAT* c = (RFLOAT*)malloc((size_t)m * sizeof(RFLOAT));
        orc_ctf(c, pixelSize, voltage, defocusU, defocusV, theta, Cs, amplitudeContrast, phaseShift, N, N, cCol, cRow, m);
        for (int p = 0; p < m; p++) {
            imgM[2 * cPxl[p]] *= c[p]; imgM[2 * cPxl[p] + 1] *= c[p];
            imgN[2 * cPxl[p]] *= c[p]; imgN[2 * cPxl[p] + 1] *= c[p];
        }
        free(cCol); free(cRow); free(cPxl); free(c);
    }
    power_spectrum_(sSVD, imgM, N, rSig);
    power_spectrum_(dSVD, img, N, rSig);
    /* NEG_FT; ADD_FT(imgM, _img[l]); ADD_FT(imgN, _imgOri[l]) */
    for (size_t i = 0; i < nFT * 2; i++) { imgM[i] = imgM[i] * -1 + img[i]; imgN[i] = imgN[i] * -1 + imgOri[i]; }
    power_spectrum_(vSigM, imgM, N, rSig);
    power_spectrum_(vSigN, imgN, N, rSig);
    free(imgM); free(imgN);
}

/* Optimiser::normCorrection, per-image part, src/Optimiser.cpp:6201-6358 as include/Config.h configures it (OPTIMISER_NORM_MASK:
 * project(img, rot3D, tran) and ADD_FT(img, _img[l]); OPTIMISER_CTF_ON_THE_FLY: CTF(ctf, ..., CEIL(rNorm) + 1, 1);
 * OPTIMISER_ADJUST_2D_IMAGE_NOISE_ZERO_MEAN off).  img = _img[l] (masked).  Returns norm(_ID[l]): the RFLOAT sum, in
 * IMAGE_FOR_EACH_PIXEL_FT order, of ABS2 over QUAD(i, j) >= pow_2(rL) && QUAD(i, j) < pow_2(rNorm). */
RFLOAT orc_norm_residual(const RFLOAT* vol, int P, int pf, int N, int projR, RFLOAT rL, RFLOAT rNorm, const double* rot,
                         const double* tran, RFLOAT pixelSize, RFLOAT voltage, RFLOAT defocusU, RFLOAT defocusV, RFLOAT theta, RFLOAT Cs,
                         RFLOAT amplitudeContrast, RFLOAT phaseShift, const RFLOAT* img)
{
    size_t nFT = (size_t)N * (N / 2 + 1);
    int nc = N / 2 + 1;
    RFLOAT* im = (RFLOAT*)calloc(nFT * 2, sizeof(RFLOAT)); /* SET_0_FT */
    int cap = (2 * projR + 1) * (projR + 1);
    int* iCol = (int*)malloc(cap * sizeof(int));
    int* iRow = (int*)malloc(cap * sizeof(int));
    int* iPxl = (int*)malloc(cap * sizeof(int));
    int n = orc_disc_list(N, projR, iCol, iRow, iPxl, NULL);
    RFLOAT* prj = (RFLOAT*)malloc((size_t)n * 2 * sizeof(RFLOAT));
    orc_project(prj, vol, P, pf, rot, iCol, iRow, n);
    for (int p = 0; p < n; p++) { im[2 * iPxl[p]] = prj[2 * p]; im[2 * iPxl[p] + 1] = prj[2 * p + 1]; }
    orc_translate_image(im, im, N, (RFLOAT)projR, (RFLOAT)tran[0], (RFLOAT)tran[1]);
    free(iCol); free(iRow); free(iPxl); free(prj);
    {
        int rc = (int)ceil((double)rNorm) + 1;
        RFLOAT* ctf = (RFLOAT*)calloc(nFT, sizeof(RFLOAT));   /* SET_0_FT(ctf): zero outside the radius the CTF is evaluated in */
        int capc = (2 * rc + 3) * (rc + 2);
        int* cCol = (int*)malloc(capc * sizeof(int));
        int* cRow = (int*)malloc(capc * sizeof(int));
        int* cPxl = (int*)malloc(capc * sizeof(int));
        int m = 0;
        for (long j = -(rc + 1); j < (rc + 1); j++) /* IMAGE_FOR_PIXEL_R_FT(r + 1) */
            for (long i = 0; i <= (rc + 1); i++) {
                RFLOAT v = (RFLOAT)((double)i * i + (double)j * j);
                if (v < pow2f_((RFLOAT)rc) && j >= -N / 2 && j < N / 2 && i <= N / 2) {
                    cCol[m] = (int)i; cRow[m] = (int)j; cPxl[m] = (int)((j >= 0 ? j : j + N) * (N / 2 + 1) + i); m++;
                }
            }
        RFLOAT* c = (RFLOAT*)malloc((size_t)m * sizeof(RFLOAT));
        orc_ctf(c, pixelSize, voltage, defocusU, defocusV, theta, Cs, amplitudeContrast, phaseShift, N, N, cCol, cRow, m);
        for (int p = 0; p < m; p++) ctf[cPxl[p]] = c[p];
        for (size_t i = 0; i < nFT; i++) { im[2 * i] *= ctf[i]; im[2 * i + 1] *= ctf[i]; }   /* FOR_EACH_PIXEL_FT(img) img[i] *= REAL(ctf[i]) */
        free(cCol); free(cRow); free(cPxl); free(c); free(ctf);
    }
    for (size_t i = 0; i < nFT * 2; i++) im[i] = im[i] * -1 + img[i];   /* NEG_FT(img); ADD_FT(img, _img[l]) */
    RFLOAT norm = 0;
    for (long j = -N / 2; j < N / 2; j++)
        for (long i = 0; i <= N / 2; i++) {
            double q = (double)i * i + (double)j * j;   /* QUAD(i, j) = gsl_pow_2(i) + gsl_pow_2(j) */
            if (q >= pow2f_(rL) && q < pow2f_(rNorm)) {
                size_t idx = (size_t)(j >= 0 ? j : j + N) * nc + i;
                norm += im[2 * idx] * im[2 * idx] + im[2 * idx + 1] * im[2 * idx + 1];
            }
        }
    free(im);
    return norm;
}

/* median(vec src, n), src/Functions/Functions.cpp:246-252: TSGSL_sort + quantile_from_sorted_data(.., 0.5)
 * (gsl-2.4/statistics/quantiles_source.c); the data are RFLOAT, the interpolation is done in double */
RFLOAT orc_median(const RFLOAT* src, int n)
{
    RFLOAT* v = (RFLOAT*)malloc((size_t)n * sizeof(RFLOAT));
    memcpy(v, src, (size_t)n * sizeof(RFLOAT));
    for (int i = 1; i < n; i++) { RFLOAT x = v[i]; int k = i - 1; while (k >= 0 && v[k] > x) { v[k + 1] = v[k]; k--; } v[k + 1] = x; }
    double index = 0.5 * (n - 1);
    int lhs = (int)index;
    double delta = index - lhs;
    double r = n == 0 ? 0.0 : (lhs == n - 1 ? v[lhs] : (1 - delta) * v[lhs] + delta * v[lhs + 1]);
    free(v);
    return (RFLOAT)r;
}

/* f1: group accumulation + closing arithmetic of allReduceSigma, src/Optimiser.cpp:6567-6707.
 * spec = [nImg][4][rSig] as written by orc_sigma_image (sSVD, dSVD, vSigM, vSigN); groupID is 1-based
 * as in the reference.  sigM/sigN/svd are [nGroup][rSig+1] accumulators (last column = weight sum) that
 * the caller all-reduces between accum and final; sig/sigRcp are [nGroup][rSig]. */
void orc_sigma_accum(RFLOAT* sigM, RFLOAT* sigN, RFLOAT* svd, const RFLOAT* spec, const int* groupID, int nImg,
                     int nGroup, int rSig, int group)
{
    (void)nGroup;
    int nc = rSig + 1;
    for (int l = 0; l < nImg; l++) {
        int g = group ? groupID[l] - 1 : 0;
        const RFLOAT *s = spec + (size_t)l * 4 * rSig, *d = s + rSig, *vm = d + rSig, *vn = vm + rSig;
        RFLOAT w = 1;
        for (int i = 0; i < rSig; i++) sigM[g * nc + i] += w * vm[i] / 2;
        sigM[g * nc + rSig] += w;
        for (int i = 0; i < rSig; i++) sigN[g * nc + i] += w * vn[i] / 2;
        sigN[g * nc + rSig] += w;
        for (int i = 0; i < rSig; i++) svd[g * nc + i] += (RFLOAT)(w * sqrt(s[i] / d[i]));
        svd[g * nc + rSig] += w;
    }
}

void orc_sigma_final(RFLOAT* sig, RFLOAT* sigRcp, RFLOAT* sigM, RFLOAT* sigN, RFLOAT* svd, int nGroup, int rSig,
                     int group, RFLOAT maskRadius, int size, RFLOAT pixelSize)
{
    int nc = rSig + 1;
    for (int g = 0; g < nGroup; g++) {
        int src = group ? g : 0;
        for (int i = 0; i < rSig; i++) {
            if (group || g == 0) {
                sigM[g * nc + i] /= sigM[g * nc + rSig];
                sigN[g * nc + i] /= sigN[g * nc + rSig];
                svd[g * nc + i] /= svd[g * nc + rSig];
            } else {
                sigM[g * nc + i] = sigM[src * nc + i];
                sigN[g * nc + i] = sigN[src * nc + i];
                svd[g * nc + i] = svd[src * nc + i];
            }
        }
    }
    RFLOAT q = maskRadius / (size * pixelSize);
    RFLOAT alpha = (RFLOAT)sqrt(M_PI * (double)q * (double)q);
    for (int g = 0; g < nGroup; g++)
        for (int j = 0; j < rSig; j++) {
            RFLOAT ratio = (RFLOAT)(1.0 < (double)svd[g * nc + j] ? 1.0 : (double)svd[g * nc + j]); /* GSL_MIN_DBL */
            sig[g * rSig + j] = ratio * sigM[g * nc + j] + (1 - ratio) * alpha * sigN[g * nc + j];
            sigRcp[g * rSig + j] = (RFLOAT)(-0.5 / sig[g * rSig + j]);
        }
}

/* a2 (ctf = true branch): Optimiser::allocPreCal, src/Optimiser.cpp:8124-8169 (ExpectPrecal, Interface.h:166-174):
 * _frequency [nPxl], _defocusP [nImg][nPxl] (image-major), _K1/_K2 [nImg].  attr = [nImg][7] floats in CTFAttr order
 * (voltage, defocusU, defocusV, defocusTheta, Cs, amplitudeContrast, phaseShift).  Note lambda's constant here is
 * 12.2643274 (:8164), not CTF.cpp's 12.2643247, and cos() resolves to the float overload (RFLOAT argument). */
void orc_expect_precal(RFLOAT* freq, RFLOAT* def, RFLOAT* k1, RFLOAT* k2, const RFLOAT* attr, int nImg, int size,
                       RFLOAT pixelSize, const int* iCol, const int* iRow, int nPxl)
{
    for (int i = 0; i < nPxl; i++)
        freq[i] = (RFLOAT)(gsl_hypot_((double)iCol[i], (double)iRow[i]) / size / pixelSize);
    for (int l = 0; l < nImg; l++) {
        const RFLOAT* a = attr + 7 * (size_t)l;
        RFLOAT voltage = a[0], dU = a[1], dV = a[2], theta = a[3], Cs = a[4];
        for (int i = 0; i < nPxl; i++) {
            RFLOAT angle = (RFLOAT)(atan2((double)iRow[i], (double)iCol[i]) - theta);
            def[(size_t)l * nPxl + i] = -(dU + dV + (dU - dV) * cosf(2 * angle)) / 2;
        }
        RFLOAT lambda = (RFLOAT)(12.2643274 / sqrt(voltage * (1 + voltage * 0.978466e-6)));
        k1[l] = (RFLOAT)(M_PI * lambda);
        k2[l] = (RFLOAT)(M_PI_2 * Cs * pow3f_(lambda));
    }
}

/* CTF rows of the defocus search, src/Optimiser.cpp:1246-1272: ctfP [nD][nPxl] for one image from the
 * pre-calculated rows and the nD defocus factors d (double, Particle::d). */
void orc_ctf_dsearch(RFLOAT* ctfP, const RFLOAT* freq, const RFLOAT* def, RFLOAT K1, RFLOAT K2, RFLOAT phaseShift,
                     RFLOAT amplitudeContrast, const double* d, int nD, int nPxl)
{
    for (int iD = 0; iD < nD; iD++)
        for (int i = 0; i < nPxl; i++) {
            RFLOAT ki = (RFLOAT)(K1 * def[i] * d[iD] * pow2f_(freq[i]) + K2 * pow4f_(freq[i]) - phaseShift);
            ctfP[(size_t)nPxl * iD + i] = -sqrtf(1 - pow2f_(amplitudeContrast)) * sinf(ki) + amplitudeContrast * cosf(ki);
        }
}

/* CTF(Image& dst, ...) whole-image form, src/CTF.cpp:31-66 (initCTF / GCTFinit, Interface.h:524-528): complex
 * image [N][N/2+1] with the CTF in the real part. */
void orc_ctf_image(RFLOAT* dst, int N, RFLOAT pixelSize, RFLOAT voltage, RFLOAT defocusU, RFLOAT defocusV, RFLOAT theta,
                   RFLOAT Cs, RFLOAT amplitudeContrast, RFLOAT phaseShift)
{
    int nc = N / 2 + 1;
    int n = N * nc;
    int* iCol = (int*)malloc(n * sizeof(int));
    int* iRow = (int*)malloc(n * sizeof(int));
    RFLOAT* c = (RFLOAT*)malloc(n * sizeof(RFLOAT));
    int m = 0;
    for (long j = -N / 2; j < N / 2; j++)
        for (long i = 0; i <= N / 2; i++) { iCol[m] = (int)i; iRow[m] = (int)j; m++; }
    orc_ctf(c, pixelSize, voltage, defocusU, defocusV, theta, Cs, amplitudeContrast, phaseShift, N, N, iCol, iRow, n);
    for (int p = 0; p < n; p++) {
        size_t idx = (size_t)(iRow[p] >= 0 ? iRow[p] : iRow[p] + N) * nc + iCol[p];
        dst[2 * idx] = c[p];
        dst[2 * idx + 1] = 0;
    }
    free(iCol); free(iRow); free(c);
}

/* ========================================================================================== */
/* image ingestion (Optimiser::initImg, src/Optimiser.cpp:4608-4800)          */
/* ========================================================================================== */

/* gsl_stats_float_mean / gsl_stats_float_sd_m, external/packages/gsl-2.4/statistics/mean_source.c:21-36 and
 * variance_source.c:29-45,95-102 (long double recurrences), through TSGSL_stats_mean / TSGSL_stats_sd_m
 * (src/Precision.cpp:435-481) which narrow the result to RFLOAT. */
static RFLOAT stats_mean_(const RFLOAT* d, size_t n)
{
    long double mean = 0;
    for (size_t i = 0; i < n; i++) mean += (d[i] - mean) / (i + 1);
    return (RFLOAT)(double)mean;
}
static RFLOAT stats_sd_m_(const RFLOAT* d, size_t n, RFLOAT meanf)
{
    const double mean = meanf;
    long double variance = 0;
    for (size_t i = 0; i < n; i++) {
        const long double delta = (d[i] - mean);
        variance += (delta * delta - variance) / (i + 1);
    }
    return (RFLOAT)sqrt((double)variance * ((double)n / (double)(n - 1)));
}

/* collects img.getRL(i, j) over IMAGE_FOR_EACH_PIXEL_RL (j outer, i inner, both from -N/2) where sel(i, j) holds */
static size_t collect_(RFLOAT* out, const RFLOAT* img, int N, int outside, RFLOAT r)
{
    size_t m = 0;
    for (long j = -N / 2; j < N / 2; j++)
        for (long i = -N / 2; i < N / 2; i++) {
            double q = (double)i * i + (double)j * j;
            int in = outside ? (q > pow2f_(r)) : 1;
            if (in) out[m++] = img[(size_t)(j >= 0 ? j : j + N) * N + (size_t)(i >= 0 ? i : i + N)];
        }
    return m;
}

/* bgMeanStddev(mean, stddev, const Image&, r), src/Image/ImageFunctions.cpp:607-621 */
void orc_bg_mean_stddev(RFLOAT* mean, RFLOAT* sd, const RFLOAT* img, int N, RFLOAT r)
{
    RFLOAT* bg = (RFLOAT*)malloc((size_t)N * N * sizeof(RFLOAT));
    size_t m = collect_(bg, img, N, 1, r);
    *mean = stats_mean_(bg, m);
    *sd = stats_sd_m_(bg, m, *mean);
    free(bg);
}

/* bgStddev(mean, const Image&, r) :585-596 and stddev(mean, const Image&) :543-547 */
RFLOAT orc_bg_stddev(RFLOAT mean, const RFLOAT* img, int N, RFLOAT r)
{
    RFLOAT* bg = (RFLOAT*)malloc((size_t)N * N * sizeof(RFLOAT));
    size_t m = collect_(bg, img, N, 1, r);
    RFLOAT s = stats_sd_m_(bg, m, mean);
    free(bg);
    return s;
}
RFLOAT orc_stddev(RFLOAT mean, const RFLOAT* img, int N) { return stats_sd_m_(img, (size_t)N * N, mean); }

/* regionMean(const Image&, rU, rL, nThread), src/Functions/Mask.cpp:102-127 (serial order) */
RFLOAT orc_region_mean(const RFLOAT* img, int N, RFLOAT rU, RFLOAT rL)
{
    RFLOAT weightSum = 0, sum = 0;
    for (long j = -N / 2; j < N / 2; j++)
        for (long i = -N / 2; i < N / 2; i++) {
            RFLOAT u = (RFLOAT)gsl_hypot_((double)i, (double)j);
            if ((u < rU) && (u >= rL)) {
                weightSum += 1;
                sum += img[(size_t)(j >= 0 ? j : j + N) * N + (size_t)(i >= 0 ? i : i + N)];
            }
        }
    return sum / weightSum;
}

/* Optimiser::substractBgImg, src/Optimiser.cpp:4928-4962 (OPTIMISER_INIT_IMG_NORMALISE_OUT_MASK_REGION): in place */
void orc_subtract_bg(RFLOAT* img, int N, RFLOAT r)
{
    RFLOAT m, s;
    orc_bg_mean_stddev(&m, &s, img, N, r);
    for (size_t i = 0; i < (size_t)N * N; i++) { img[i] -= m; img[i] /= s; }
}

/* per-image terms of Optimiser::statImg, src/Optimiser.cpp:4810-4875: out = {regionMean(img, r, 0), bgStddev(0, img, r),
 * stddev(0, img), bgStddev(0, img, r)^2 (gsl_pow_2 in double)} as doubles */
void orc_stat_img(double* out, const RFLOAT* img, int N, RFLOAT r)
{
    out[0] = orc_region_mean(img, N, r, 0);
    RFLOAT b = orc_bg_stddev(0, img, N, r);
    out[1] = b;
    out[2] = orc_stddev(0, img, N);
    out[3] = (double)b * (double)b;
}

/* softMask(dst, src, r, ew, bg, nThread), src/Functions/Mask.cpp:363-385 (Optimiser::maskImg passes bg = 0) */
void orc_soft_mask_bg(RFLOAT* dst, const RFLOAT* src, int N, RFLOAT r, RFLOAT ew, RFLOAT bg)
{
    for (long j = -N / 2; j < N / 2; j++)
        for (long i = -N / 2; i < N / 2; i++) {
            RFLOAT u = (RFLOAT)gsl_hypot_((double)i, (double)j);
            size_t idx = (size_t)(j >= 0 ? j : j + N) * N + (size_t)(i >= 0 ? i : i + N);
            if (u > r + ew) dst[idx] = bg;
            else if (u >= r) {
                RFLOAT w = (RFLOAT)(0.5 - 0.5 * cos((u - r) / ew * M_PI));
                dst[idx] = bg * w + src[idx] * (1 - w);
            } else dst[idx] = src[idx];
        }
}

/* ========================================================================================== */
/* the particle filter's deterministic arithmetic (src/Particle.cpp,           */
/* src/Geometry/DirectionalStat.cpp); random draws (perturbation, shuffle, u0) are inputs here. */
/* ========================================================================================== */

/* 4x4 inverse by cofactors (Eigen's fixed-size dmat44::inverse() is the same closed form) */
static void inv4_(double* o, const double* m)
{
    double inv[16];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    for (int i = 0; i < 16; i++) o[i] = inv[i] / det;
}
static double det4_(const double* m)
{
    double c0 = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    double c1 = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    double c2 = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    double c3 = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    return m[0] * c0 + m[1] * c1 + m[2] * c2 + m[3] * c3;
}
static double quad4_(const double* x, const double* M)
{
    double s = 0;
    for (int j = 0; j < 4; j++) {
        double t = 0;
        for (int k = 0; k < 4; k++) t += M[j * 4 + k] * x[k];
        s += x[j] * t;
    }
    return s;
}

/* inferACG(dmat44& dst, const dmat4& src), src/Geometry/DirectionalStat.cpp:93-145; A row-major [4][4]; returns the
 * number of fixed-point rounds */
int orc_infer_acg(double* A, const double* q, int n)
{
    double B[16], Ainv[16];
    for (int i = 0; i < 16; i++) B[i] = (i % 5 == 0) ? 1.0 : 0.0;
    int rounds = 0;
    double diff;
    do {
        memcpy(A, B, sizeof(B));
        memset(B, 0, sizeof(B));
        double nf = 0;
        inv4_(Ainv, A);
        for (int i = 0; i < n; i++) {
            const double* x = q + 4 * (size_t)i;
            double u = quad4_(x, Ainv);
            for (int j = 0; j < 4; j++)
                for (int k = 0; k < 4; k++) B[j * 4 + k] += (x[j] * x[k]) / u;
            nf += 1.0 / u;
        }
        for (int i = 0; i < 16; i++) B[i] *= 4.0 / nf;
        diff = 0;
        for (int i = 0; i < 16; i++) diff += fabs(A[i] - B[i]);
        rounds++;
    } while (diff > 1e-3 && rounds < 100000);
    return rounds;
}

/* pdfACG(x, sig), :19-24 */
double orc_pdf_acg(const double* x, const double* sig)
{
    double inv[16];
    inv4_(inv, sig);
    return pow(det4_(sig), -0.5) * pow(quad4_(x, inv), -2);
}

/* eigenvector of the largest eigenvalue of a symmetric 4x4 (inferACG(dvec4& mean, ...), :224-262 uses Eigen's
 * SelfAdjointEigenSolver; cyclic Jacobi here -- same vector up to sign and rounding) */
void orc_sym4_top_eigvec(double* v, const double* Ain)
{
    double A[16], V[16];
    memcpy(A, Ain, sizeof(A));
    for (int i = 0; i < 16; i++) V[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 64; sweep++) {
        double off = 0;
        for (int p = 0; p < 4; p++) for (int r = p + 1; r < 4; r++) off += A[p * 4 + r] * A[p * 4 + r];
        if (off < 1e-300) break;
        for (int p = 0; p < 4; p++)
            for (int r = p + 1; r < 4; r++) {
                if (fabs(A[p * 4 + r]) < 1e-300) continue;
                double theta = (A[r * 4 + r] - A[p * 4 + p]) / (2 * A[p * 4 + r]);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                double c = 1 / sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < 4; k++) {
                    double akp = A[k * 4 + p], akr = A[k * 4 + r];
                    A[k * 4 + p] = c * akp - s * akr; A[k * 4 + r] = s * akp + c * akr;
                }
                for (int k = 0; k < 4; k++) {
                    double apk = A[p * 4 + k], ark = A[r * 4 + k];
                    A[p * 4 + k] = c * apk - s * ark; A[r * 4 + k] = s * apk + c * ark;
                }
                for (int k = 0; k < 4; k++) {
                    double vkp = V[k * 4 + p], vkr = V[k * 4 + r];
                    V[k * 4 + p] = c * vkp - s * vkr; V[k * 4 + r] = s * vkp + c * vkr;
                }
            }
    }
    int im = 0;
    for (int i = 1; i < 4; i++) if (A[i * 4 + i] > A[im * 4 + im]) im = i;
    double nrm = 0;
    for (int k = 0; k < 4; k++) nrm += V[k * 4 + im] * V[k * 4 + im];
    nrm = sqrt(nrm);
    for (int k = 0; k < 4; k++) v[k] = V[k * 4 + im] / nrm;
}

/* quaternion_mul(dst, a, b) and quaternion_conj, src/Geometry/Euler.cpp (Hamilton product) */
static void qmul_(double* d, const double* a, const double* b)
{
    double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
    d[0] = w; d[1] = x; d[2] = y; d[3] = z;
}

/* Particle::calVari(PAR_R), MODE_3D with PARTICLE_ROT_MEAN_USING_STAT_CAL_VARI, src/Particle.cpp:1020-1080:
 * mean = inferACG(mean, _r); every quaternion is LEFT-multiplied by conj(mean) (quaternion_mul(quat, quaternion_conj(mean),
 * quat), :1052-1058; quaternion_mul(dst, a, b) is the Hamilton product a * b, src/Geometry/Euler.cpp:13-26), k1..k3 =
 * inferACG(k1, k2, k3, _r), then left-multiplied by mean again (:1066-1074).  k[3] = (k1, k2, k3); q [n][4] is rotated to
 * the mean frame and back in place, as the reference does.  (The random anchor of :1039-1043 only feeds symmetrise(),
 * which returns at once for C1.) */
void orc_cal_vari_R(double* k, double* mean, double* q, int n)
{
    double A[16], cm[4];
    orc_infer_acg(A, q, n);
    orc_sym4_top_eigvec(mean, A);
    cm[0] = mean[0]; cm[1] = -mean[1]; cm[2] = -mean[2]; cm[3] = -mean[3];
    for (int i = 0; i < n; i++) { double t[4]; qmul_(t, cm, q + 4 * i); memcpy(q + 4 * i, t, sizeof(t)); }
    orc_infer_acg(A, q, n);
    k[0] = A[5] / A[0]; k[1] = A[10] / A[0]; k[2] = A[15] / A[0];
    for (int i = 0; i < n; i++) { double t[4]; qmul_(t, mean, q + 4 * i); memcpy(q + 4 * i, t, sizeof(t)); }
}

/* Particle::perturb(pf, PAR_R), MODE_3D, src/Particle.cpp:1185-1243 (PARTICLE_ROTATION_KAPPA off,
 * PARTICLE_ROT_MEAN_USING_STAT_PERTURB on), WITHOUT the closing balanceWeight (orc_balance_weight_R):
 *   d = sampleACG(pf^2 min(PERTURB_K_MAX = 1, k1), ..k2, ..k3, nR) (src/Geometry/DirectionalStat.cpp:40-88: L = chol(diag(1,
 *   k1', k2', k3')) = the square roots of the diagonal, v = L g, v /= |v|; g [n][4] are the standard normals, an input here);
 *   mean = inferACG(mean, _r);  quat = conj(mean) * quat;  quat = d_i * quat;  quat = mean * quat   (three passes). */
void orc_perturb_R(double* q, int n, const double* k, double pf, const double* g)
{
    double A[16], mean[4], cm[4];
    const double pf2 = pf * pf;   /* gsl_pow_2(pf) */
    const double l1 = sqrt(pf2 * (k[0] < 1.0 ? k[0] : 1.0)), l2 = sqrt(pf2 * (k[1] < 1.0 ? k[1] : 1.0)),
                 l3 = sqrt(pf2 * (k[2] < 1.0 ? k[2] : 1.0));
    orc_infer_acg(A, q, n);
    orc_sym4_top_eigvec(mean, A);
    cm[0] = mean[0]; cm[1] = -mean[1]; cm[2] = -mean[2]; cm[3] = -mean[3];
    for (int i = 0; i < n; i++) { double t[4]; qmul_(t, cm, q + 4 * i); memcpy(q + 4 * i, t, sizeof(t)); }
    for (int i = 0; i < n; i++) {
        double v[4] = {g[4 * i], l1 * g[4 * i + 1], l2 * g[4 * i + 2], l3 * g[4 * i + 3]}, t[4];
        double nrm = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
        for (int c = 0; c < 4; c++) v[c] /= nrm;
        qmul_(t, v, q + 4 * i);
        memcpy(q + 4 * i, t, sizeof(t));
    }
    for (int i = 0; i < n; i++) { double t[4]; qmul_(t, mean, q + 4 * i); memcpy(q + 4 * i, t, sizeof(t)); }
}

/* ---- point-group symmetry of the particle filter and of prepareTF ----
 * Symmetry::init(const char sym[]) (src/Geometry/Symmetry.cpp:61-66,107-122): symmetryGroup + fillSymmetryEntry
 * (src/Geometry/SymmetryFunctions.cpp:13-164) -> fillLR (:146-208; rotations only: the reflexion / inversion branches end in
 * CLOG(FATAL)) -> completePointGroup (:225-278).  Output: the nSym NON-identity elements in the reference's order, R column-major
 * [nSym][9] (what SYMMETRIZE_FT hands to VOL_TRANSFORM_MAT_FT, include/Geometry/Transformation.h:170-194) and
 * Symmetry::quat(i) [nSym][4] (what symmetryCounterpart multiplies with).  Returns nSym, -1 for an unknown group, -2 when cap
 * is too small.  Cast points kept: `RFLOAT angle = 2 * M_PI / fold` and `angle * j` are float (:160-164); the axes are NOT
 * normalised (RotationSO stores them as given, src/Geometry/SymmetryOperation.cpp:12-23). */
static void mat33_mul_(double* d, const double* a, const double* b)   /* row-major helpers of this block */
{
    double t[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) { double s = 0; for (int k = 0; k < 3; k++) s += a[r * 3 + k] * b[k * 3 + c]; t[r * 3 + c] = s; }
    memcpy(d, t, sizeof(t));
}
static int same_matrix_(const double* a, const double* b)   /* SAME_MATRIX, include/Geometry/Symmetry.h:64-73: EQUAL_ACCURACY 1e-2 */
{
    for (int i = 0; i < 9; i++) if (fabs(a[i] - b[i]) > 1e-2) return 0;
    return 1;
}
static void quat_of_matrix_(double* q, const double* m /* row-major */)   /* quaternion(dvec4&, const dmat33&), src/Geometry/Euler.cpp:112-123 */
{
#define M_(r, c) m[(r) * 3 + (c)]
    double v;
    v = 1 + M_(0, 0) + M_(1, 1) + M_(2, 2); q[0] = 0.5 * sqrt(v > 0 ? v : 0);
    v = 1 + M_(0, 0) - M_(1, 1) - M_(2, 2); q[1] = 0.5 * sqrt(v > 0 ? v : 0);
    v = 1 - M_(0, 0) + M_(1, 1) - M_(2, 2); q[2] = 0.5 * sqrt(v > 0 ? v : 0);
    v = 1 - M_(0, 0) - M_(1, 1) + M_(2, 2); q[3] = 0.5 * sqrt(v > 0 ? v : 0);
    q[1] = copysign(q[1], M_(2, 1) - M_(1, 2));
    q[2] = copysign(q[2], M_(0, 2) - M_(2, 0));
    q[3] = copysign(q[3], M_(1, 0) - M_(0, 1));
#undef M_
}
struct sym_entry_ { int fold; double ax[3]; };
static int sym_entries_(struct sym_entry_* e, const char* sym)
{
    /* symmetryGroup: "^C[[:digit:]]+$", "^D[[:digit:]]+$", "T", "O", "I1" .. "I4" */
    int n = 0;
#define ROT_(f, x, y, z) do { e[n].fold = (f); e[n].ax[0] = (x); e[n].ax[1] = (y); e[n].ax[2] = (z); n++; } while (0)
    size_t len = strlen(sym);
    int digits = len > 1;
    for (size_t i = 1; i < len; i++) if (sym[i] < '0' || sym[i] > '9') digits = 0;
    if ((sym[0] == 'C' || sym[0] == 'D') && digits) {
        ROT_(atoi(sym + 1), 0, 0, 1);                       /* PG_CN */
        if (sym[0] == 'D') ROT_(2, 1, 0, 0);                /* PG_DN */
    } else if (!strcmp(sym, "T")) { ROT_(3, 0, 0, 1); ROT_(2, 0, 0.816496, 0.577350); }
    else if (!strcmp(sym, "O")) { ROT_(3, 0.5773502, 0.5773502, 0.5773502); ROT_(4, 0, 0, 1); }
    else if (!strcmp(sym, "I1")) { ROT_(2, 1, 0, 0); ROT_(5, 0.8506508, 0, -0.5257311); ROT_(3, 0.9341724, 0.3568221, 0); }
    else if (!strcmp(sym, "I2")) { ROT_(2, 0, 0, 1); ROT_(5, 0.5257311, 0, 0.8506508); ROT_(3, 0, 0.3568221, 0.9341724); }
    else if (!strcmp(sym, "I3")) { ROT_(2, -0.5257311, 0, 0.8506508); ROT_(5, 0, 0, 1); ROT_(3, -0.4911235, 0.3568221, 0.7946545); }
    else if (!strcmp(sym, "I4")) { ROT_(2, 0.5257311, 0, 0.8506508); ROT_(5, 0.8944272, 0, 0.4472136); ROT_(3, 0.4911235, 0.3568221, 0.7946545); }
    else return -1;
#undef ROT_
    return n;
}
int orc_symmetry(const char* sym, double* Rcm, double* quat, int cap)
{
    struct sym_entry_ e[4];
    const int ne = sym_entries_(e, sym);
    if (ne < 0) return -1;
    double* R = (double*)malloc((size_t)(cap > 0 ? cap : 1) * 9 * sizeof(double));   /* row-major while the group is built */
    int n = 0;
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
#define NOVO_(m) ({ int nv_ = !same_matrix_((m), I); for (int q_ = 0; nv_ && q_ < n; q_++) if (same_matrix_((m), R + 9 * q_)) nv_ = 0; nv_; })
    /* fillLR: for every rotation entry, R = rotate3D(angle * j, axis), j = 1 .. fold - 1, appended when novo */
    for (int i = 0; i < ne; i++) {
        const RFLOAT angle = (RFLOAT)(2 * M_PI / e[i].fold);
        for (int j = 1; j < e[i].fold; j++) {
            const double phi = (double)(angle * (RFLOAT)j);
            /* rotate3D(dst, phi, axis) = rotate3D(dst, quaternion(phi, axis)), src/Geometry/Euler.cpp:102-110,272-281 */
            const double q[4] = {cos(phi / 2), sin(phi / 2) * e[i].ax[0], sin(phi / 2) * e[i].ax[1], sin(phi / 2) * e[i].ax[2]};
            double cm[9], m[9];
            orc_rotate3D(cm, q);
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) m[r * 3 + c] = cm[c * 3 + r];
            if (NOVO_(m)) {
                if (n >= cap) { free(R); return -2; }
                memcpy(R + 9 * n, m, sizeof(m));
                quat_of_matrix_(quat + 4 * n, m);
                n++;
            }
        }
    }
    /* completePointGroup: a table of visited (i, j) pairs that grows with every new element; the first unvisited pair in
     * row-major order is multiplied next */
    {
        unsigned char* table = (unsigned char*)calloc((size_t)cap * cap + 1, 1);
        int dim = n;
        for (;;) {
            int fi = -1, fj = -1;
            for (int r = 0; r < dim && fi < 0; r++)
                for (int c = 0; c < dim; c++)
                    if (!table[(size_t)r * cap + c]) { fi = r; fj = c; table[(size_t)r * cap + c] = 1; break; }
            if (fi < 0) break;
            double m[9];
            mat33_mul_(m, R + 9 * fi, R + 9 * fj);
            if (NOVO_(m)) {
                if (n >= cap) { free(R); free(table); return -2; }
                memcpy(R + 9 * n, m, sizeof(m));
                quat_of_matrix_(quat + 4 * n, m);
                n++;
                dim++;
            }
        }
        free(table);
    }
#undef NOVO_
    for (int s = 0; s < n; s++)
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Rcm[9 * s + c * 3 + r] = R[9 * s + r * 3 + c];
    free(R);
    return n;
}

/* symmetryCounterpart(dvec4& dst, const Symmetry& sym, const dvec4* anchor), src/Geometry/Symmetry.cpp:309-336: among dst and
 * conj(sym.quat(i)) * dst the quaternion with the largest |<., anchor>| (kept in RFLOAT, strict >); anchor NULL = ANCHOR_POINT_2
 * = (1, 0, 0, 0) */
void orc_symmetry_counterpart(double* dst, const double* symQuat, int nSym, const double* anchor)
{
    static const double anchor2[4] = {1, 0, 0, 0};
    if (!anchor) anchor = anchor2;
    double q[4] = {dst[0], dst[1], dst[2], dst[3]};
    RFLOAT s = (RFLOAT)fabs(dst[0] * anchor[0] + dst[1] * anchor[1] + dst[2] * anchor[2] + dst[3] * anchor[3]);
    for (int i = 0; i < nSym; i++) {
        const double cs[4] = {symQuat[4 * i], -symQuat[4 * i + 1], -symQuat[4 * i + 2], -symQuat[4 * i + 3]};
        double p[4];
        qmul_(p, cs, dst);
        RFLOAT t = (RFLOAT)fabs(p[0] * anchor[0] + p[1] * anchor[1] + p[2] * anchor[2] + p[3] * anchor[3]);
        if (t > s) { s = t; memcpy(q, p, sizeof(q)); }
    }
    memcpy(dst, q, sizeof(q));
}

/* Particle::symmetrise(const dvec4* anchor), src/Particle.cpp:2445-2470 (returns at once for C1: nSym == 0) */
void orc_symmetrise(double* q, int n, const double* symQuat, int nSym, const double* anchor)
{
    if (nSym <= 0) return;
    for (int i = 0; i < n; i++) orc_symmetry_counterpart(q + 4 * (size_t)i, symQuat, nSym, anchor);
}

/* Particle::calVari(PAR_R) with a point group, src/Particle.cpp:1020-1080: anch = _r.row(gsl_rng_uniform_int(engine, _nR)) (the
 * draw iAnchor is an input); symmetrise(&anch); then as orc_cal_vari_R */
void orc_cal_vari_R_sym(double* k, double* mean, double* q, int n, const double* symQuat, int nSym, int iAnchor)
{
    if (nSym > 0) {
        const double anch[4] = {q[4 * iAnchor], q[4 * iAnchor + 1], q[4 * iAnchor + 2], q[4 * iAnchor + 3]};
        orc_symmetrise(q, n, symQuat, nSym, anch);
    }
    orc_cal_vari_R(k, mean, q, n);
}

/* Particle::perturb(pf, PAR_R) with a point group: the three passes of orc_perturb_R, then symmetrise(&mean) (:1234) */
void orc_perturb_R_sym(double* q, int n, const double* k, double pf, const double* g, const double* symQuat, int nSym)
{
    double A[16], mean[4];
    if (nSym > 0) { orc_infer_acg(A, q, n); orc_sym4_top_eigvec(mean, A); }   /* the mean orc_perturb_R is about to use */
    orc_perturb_R(q, n, k, pf, g);
    if (nSym > 0) orc_symmetrise(q, n, symQuat, nSym, mean);
}

/* Particle::perturb(pf, PAR_T), src/Particle.cpp:1244-1272, + reCentre (PARTICLE_RECENTRE_TRANSQ), :2473-2495, WITHOUT
 * the closing balanceWeight.  gsl_ran_bivariate_gaussian(engine, s0, s1, rho = 0, &x, &y) (PARTICLE_RHO off: _rho = 0)
 * returns (s0 n0, s1 n1) for two independent standard normals (randist/bigauss.c); g [n][4] = n0, n1 and the two normals
 * of the re-draw.  transM = transS * gsl_cdf_chisq_Qinv(transQ, 2) = transS * (-2 ln transQ) (chi-square, 2 dof:
 * Q(x) = exp(-x / 2)). */
void orc_perturb_T(double* t, int n, double s0, double s1, double pf, double transS, double transQ, const double* g)
{
    const double transM = transS * (-2.0 * log(transQ));
    for (int i = 0; i < n; i++) {
        double x = s0 * g[4 * i], y = s1 * g[4 * i + 1];
        t[2 * i] += x * pf;
        t[2 * i + 1] += y * pf;
    }
    for (int i = 0; i < n; i++)
        if (gsl_hypot_(t[2 * i], t[2 * i + 1]) > transM) {
            t[2 * i] = transS * g[4 * i + 2];
            t[2 * i + 1] = transS * g[4 * i + 3];
        }
}

/* gsl_stats_mean / gsl_stats_sd_m on doubles (statistics/mean_source.c, variance_source.c) */
static double dmean_(const double* d, size_t stride, size_t n)
{
    long double mean = 0;
    for (size_t i = 0; i < n; i++) mean += (d[i * stride] - mean) / (i + 1);
    return (double)mean;
}
static double dsd_m_(const double* d, size_t stride, size_t n, double mean)
{
    long double variance = 0;
    for (size_t i = 0; i < n; i++) {
        const long double delta = (d[i * stride] - mean);
        variance += (delta * delta - variance) / (i + 1);
    }
    return sqrt((double)variance * ((double)n / (double)(n - 1)));
}

/* Particle::calVari(PAR_T), :1096-1112 (gsl_stats_sd per column; PARTICLE_RHO off) */
void orc_cal_vari_T(double* s, const double* t, int n)
{
    s[0] = dsd_m_(t, 2, n, dmean_(t, 2, n));
    s[1] = dsd_m_(t + 1, 2, n, dmean_(t + 1, 2, n));
}

/* Particle::balanceWeight(PAR_R) MODE_3D, :2333-2343 + normW: w_i = 1 / pdfACG(r_i, inferACG(r)), normalised */
void orc_balance_weight_R(double* w, const double* q, int n)
{
    double A[16], sum = 0;
    orc_infer_acg(A, q, n);
    for (int i = 0; i < n; i++) { w[i] = 1.0 / orc_pdf_acg(q + 4 * i, A); sum += w[i]; }
    for (int i = 0; i < n; i++) w[i] /= sum;
}

/* Particle::balanceWeight(PAR_T), :2345-2376 + normW; gsl_ran_bivariate_gaussian_pdf (randist/bigauss.c) with rho = 0 */
void orc_balance_weight_T(double* w, const double* t, int n)
{
    double m0 = dmean_(t, 2, n), m1 = dmean_(t + 1, 2, n);
    double s0 = dsd_m_(t, 2, n, m0), s1 = dsd_m_(t + 1, 2, n, m1), rho = 0, sum = 0;
    for (int i = 0; i < n; i++) {
        double u = (t[2 * i] - m0) / s0, v = (t[2 * i + 1] - m1) / s1, c = 1 - rho * rho;
        double p = (1 / (2 * M_PI * s0 * s1 * sqrt(c))) * exp(-(u * u - 2 * rho * u * v + v * v) / (2 * c));
        w[i] = 1.0 / p;
        sum += w[i];
    }
    for (int i = 0; i < n; i++) w[i] /= sum;
}

/* Particle::keepHalfHeightPeak, :1964-2011 */
void orc_keep_half_height_peak(double* u, int n, double peakFactor)
{
    int im = 0;
    for (int i = 1; i < n; i++) if (u[i] > u[im]) im = i;
    double hh = u[im] * peakFactor;
    for (int i = 0; i < n; i++) { if (u[i] < hh) u[i] = 0; else u[i] -= hh; }
}

/* the systematic resampling of Particle::resample(n, pt), :1333-1372 (PARTICLE_PRIOR_ONE), AFTER the shuffle: inputs
 * are the shuffled w, u and the draw u0 in [0, 1/nOut); idx[j] = source index, wOut normalised (normW) */
void orc_resample(int* idx, double* wOut, const double* w, const double* u, int nIn, int nOut, double u0)
{
    double* cdf = (double*)malloc(nIn * sizeof(double));
    double sum = 0, acc = 0;
    for (int i = 0; i < nIn; i++) sum += w[i] * u[i];
    for (int i = 0; i < nIn; i++) { acc += (w[i] * u[i]) / sum; cdf[i] = acc; }
    for (int i = 0; i < nIn; i++) cdf[i] /= cdf[nIn - 1];
    int i = 0;
    double ws = 0;
    for (int j = 0; j < nOut; j++) {
        double uj = u0 + j * 1.0 / nOut;
        while (uj > cdf[i]) i++;
        idx[j] = i;
        wOut[j] = 1.0 / u[i];
        ws += wOut[j];
    }
    for (int j = 0; j < nOut; j++) wOut[j] /= ws;
    free(cdf);
}

/* ========================================================================================== */
/* Model::compareTwoHemispheres, 3-D mode (src/Model.cpp:307-700) -- element-wise pieces        */
/* ========================================================================================== */

/* softMask(Volume& mask, r, ew), src/Functions/Mask.cpp:470-486; in-memory layout [N][N][N] */
void orc_core_mask(RFLOAT* mask, int N, RFLOAT r, RFLOAT ew)
{
    for (long k = -N / 2; k < N / 2; k++)
        for (long j = -N / 2; j < N / 2; j++)
            for (long i = -N / 2; i < N / 2; i++) {
                RFLOAT u = (RFLOAT)gsl_hypot3_((double)i, (double)j, (double)k);
                RFLOAT v;
                if (u > r + ew) v = 0;
                else if (u >= r) v = (RFLOAT)(0.5 + 0.5 * cos((u - r) / ew * M_PI));
                else v = 1;
                mask[((size_t)(k < 0 ? k + N : k) * N + (j < 0 ? j + N : j)) * N + (i < 0 ? i + N : i)] = v;
            }
}

/* softMask(Volume& dst, const Volume& src, r, ew, bg), src/Functions/Mask.cpp:499-521, in place (the spherical mask of
 * Optimiser::solventFlatten, src/Optimiser.cpp:7958-7975, OPTIMISER_SOLVENT_FLATTEN_MASK_ZERO: bg = 0) */
void orc_soft_mask_volume(RFLOAT* vol, int N, RFLOAT r, RFLOAT ew, RFLOAT bg)
{
    for (long k = -N / 2; k < N / 2; k++)
        for (long j = -N / 2; j < N / 2; j++)
            for (long i = -N / 2; i < N / 2; i++) {
                size_t e = ((size_t)(k < 0 ? k + N : k) * N + (j < 0 ? j + N : j)) * N + (i < 0 ? i + N : i);
                RFLOAT u = (RFLOAT)gsl_hypot3_((double)i, (double)j, (double)k);
                if (u > r + ew) vol[e] = bg;
                else if (u >= r) {
                    RFLOAT w = (RFLOAT)(0.5 - 0.5 * cos((u - r) / ew * M_PI));
                    vol[e] = bg * w + vol[e] * (1 - w);
                }
            }
}

/* softMask(dst, src, alpha, bg), src/Functions/Mask.cpp:510-521 */
void orc_alpha_mask(RFLOAT* dst, const RFLOAT* src, const RFLOAT* alpha, RFLOAT bg, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        RFLOAT w = 1 - alpha[i];
        dst[i] = bg * w + src[i] * (1 - w);
    }
}

/* randomPhase(dst, src, r), src/Functions/Spectrum.cpp:365-386, with the phases (the reference's TSGSL_ran_flat draws, one
 * per stored element in loop order) supplied by the caller: phases[e] for the stored element e */
void orc_random_phase(RFLOAT* dst, const RFLOAT* src, int N, int r, const RFLOAT* phases)
{
    const long nc = N / 2 + 1;
    for (long k = -N / 2; k < N / 2; k++)
        for (long j = -N / 2; j < N / 2; j++)
            for (long i = 0; i <= N / 2; i++) {
                size_t e = ((size_t)(k < 0 ? k + N : k) * N + (j < 0 ? j + N : j)) * nc + i;
                int u = AROUND_(gsl_hypot3_((double)i, (double)j, (double)k));
                RFLOAT re = src[2 * e], im = src[2 * e + 1];
                if (u > r) {
                    RFLOAT c = cosf(phases[e]), s = sinf(phases[e]);   /* COMPLEX_POLAR */
                    RFLOAT nr = re * c - im * s, ni = re * s + im * c;
                    re = nr; im = ni;
                }
                dst[2 * e] = re; dst[2 * e + 1] = im;
            }
}

/* averaging of the two halves: src/Model.cpp:663-674 (inside QUAD_3 < r^2) or :620-627 / :688-696 (r < 0: everywhere) */
void orc_average_halves(RFLOAT* A, RFLOAT* B, int N, int r)
{
    const long nc = N / 2 + 1;
    for (long k = -N / 2; k < N / 2; k++)
        for (long j = -N / 2; j < N / 2; j++)
            for (long i = 0; i <= N / 2; i++) {
                if (r >= 0) {
                    RFLOAT q = (RFLOAT)((double)i * i + (double)j * j + (double)k * k);
                    if (!(q < pow2f_((RFLOAT)r))) continue;
                }
                size_t e = ((size_t)(k < 0 ? k + N : k) * N + (j < 0 ? j + N : j)) * nc + i;
                for (int c = 0; c < 2; c++) {
                    RFLOAT avg = (A[2 * e + c] + B[2 * e + c]) / 2;
                    A[2 * e + c] = avg; B[2 * e + c] = avg;
                }
            }
}


/* ========================================================================================== */
/* Per-op rates of this port on T threads (tools/cpu_port_vs_survey.py: SURVEY 8(d)'s 15 % check */
/* against SURVEY 3.5's compiled-reference rates).  OpenMP over rotations, one call per rotation */
/* and thread as the reference's loops make them (src/Optimiser.cpp:758-781, 7038-7232).         */
/* ========================================================================================== */
#include <omp.h>
double orc_bench_project(const RFLOAT* vol, int P, int pf, const double* mats, int nRot, const int* iCol, const int* iRow, int nPxl,
                         int threads)
{
    RFLOAT* buf = (RFLOAT*)malloc((size_t)threads * nPxl * 2 * sizeof(RFLOAT));
    const double t0 = omp_get_wtime();
#pragma omp parallel for num_threads(threads) schedule(dynamic)
    for (int r = 0; r < nRot; r++)
        orc_project(buf + (size_t)omp_get_thread_num() * nPxl * 2, vol, P, pf, mats + 9 * (size_t)r, iCol, iRow, nPxl);
    const double dt = omp_get_wtime() - t0;
    free(buf);
    return dt;
}
double orc_bench_insertP(RFLOAT* F, RFLOAT* T, int P, const RFLOAT* src, const RFLOAT* ctf, const double* mats, int nRot, RFLOAT w,
                         const int* iColPad, const int* iRowPad, int nPxl, int threads)
{
    const double t0 = omp_get_wtime();
#pragma omp parallel for num_threads(threads) schedule(dynamic)
    for (int r = 0; r < nRot; r++) insertP_atomic_(F, T, P, src, ctf, mats + 9 * (size_t)r, w, iColPad, iRowPad, nPxl);
    return omp_get_wtime() - t0;
}
double orc_bench_logDataVSPrior(const RFLOAT* dat, const RFLOAT* pri, const RFLOAT* ctf, const RFLOAT* sigRcp, int m, int nCall,
                                int threads, double* sink)
{
    double acc = 0;
    const double t0 = omp_get_wtime();
#pragma omp parallel for num_threads(threads) reduction(+ : acc)
    for (int c = 0; c < nCall; c++) acc += orc_logDataVSPrior(dat, pri, ctf, sigRcp, m);
    *sink = acc;
    return omp_get_wtime() - t0;
}
