// thx_pf.hip -- the particle filter of the local search on the device:
// Particle::perturb / balanceWeight / reCentre / keepHalfHeightPeak / calRank1st / calVari / shuffle / resample
// (src/Particle.cpp:990-1480,1880-2012,2202-2420,2473-2495) and the angular-central-Gaussian statistics they rest on
// (src/Geometry/DirectionalStat.cpp:19-262), MODE_3D, as include/Config.h configures them (PARTICLE_PRIOR_ONE,
// PARTICLE_RECENTRE(_TRANSQ), PARTICLE_ROT_MEAN_USING_STAT_*, PARTICLE_BALANCE_WEIGHT_R/T, OPTIMISER_PEAK_FACTOR_R).
//
// One wave per image: the support sets are tiny (mLR = 125 rotations, mLT = 9 shifts), the work is thousands of images.
// The reference draws from GSL's global mt19937 under OpenMP (unordered, so not reproducible run to run); here every
// draw comes from a counter-based Philox4x32-10 stream keyed by (seed, image, call, purpose, index): reproducible and
// independent of launch geometry.  Deterministic arithmetic is checked against the oracle, draws statistically.
#include <math.h>
#include <string.h>

#include <vector>

#include "thx_common.h"
#include "thx_philox.h"

namespace thx {

constexpr int kPfMax = 256;  // support points per image and parameter (mLR, mLT <= 256)

// ---- 4x4 helpers (double); matrices row-major ----
__device__ void inv4(double* o, const double* m, double* detOut)
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
    const double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    for (int i = 0; i < 16; i++) o[i] = inv[i] / det;
    if (detOut) *detOut = det;
}
__device__ __forceinline__ double quad4(const double* x, const double* M)
{
    double s = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        double t = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) t += M[j * 4 + k] * x[k];
        s += x[j] * t;
    }
    return s;
}
__device__ __forceinline__ void qmul(double* d, const double* a, const double* b)
{   // quaternion_mul, src/Geometry/Euler.cpp:13-26
    const double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    const double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    const double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    const double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
    d[0] = w; d[1] = x; d[2] = y; d[3] = z;
}

// symmetryCounterpart(dst, sym, anchor), src/Geometry/Symmetry.cpp:309-336: among q and conj(sym.quat(i)) * q the quaternion
// with the largest |<., anchor>| (the reference keeps that number in RFLOAT and compares with a strict >)
__device__ __forceinline__ void sym_counterpart(double* q, const double* __restrict__ symQ, int nSym, const double* anchor)
{
    double best[4] = {q[0], q[1], q[2], q[3]};
    float s = (float)fabs(q[0] * anchor[0] + q[1] * anchor[1] + q[2] * anchor[2] + q[3] * anchor[3]);
    for (int i = 0; i < nSym; i++) {
        const double cs[4] = {symQ[4 * i], -symQ[4 * i + 1], -symQ[4 * i + 2], -symQ[4 * i + 3]};
        double p[4];
        qmul(p, cs, q);
        const float t = (float)fabs(p[0] * anchor[0] + p[1] * anchor[1] + p[2] * anchor[2] + p[3] * anchor[3]);
        if (t > s) { s = t; best[0] = p[0]; best[1] = p[1]; best[2] = p[2]; best[3] = p[3]; }
    }
    q[0] = best[0]; q[1] = best[1]; q[2] = best[2]; q[3] = best[3];
}

// Particle::calVari(PAR_R), src/Particle.cpp:1020-1080, on the n quaternions in LDS (one wave): with a point group the support
// points are first replaced by their counterparts next to a randomly chosen one of them (anch = _r.row(gsl_rng_uniform_int(
// engine, _nR)); symmetrise(&anch), :1030-1036 -- Philox stream (seed, image, call, 13, 0)); then LEFT multiplication by
// conj(mean), inferACG(k1, k2, k3), left multiplication by mean.  Every lane returns the same k[3].
__device__ void cal_vari_R(double* k, double* sq /*LDS [n][4]*/, int n, int lane, const double* __restrict__ symQ, int nSym,
                           unsigned long long seed, unsigned img, unsigned call);

// inverse of a SYMMETRIC 4x4 for inferACG's rounds: the ten distinct cofactors (the other six are the same products summed in
// another order) times 1 / det -- against inv4's sixteen cofactors and sixteen divisions; differs from it by <= 1 ulp per entry,
// the level of the wave-ordered sums around it.  A round of the fixed point is a dependent chain of f64 latencies (an
// image whose cloud has collapsed runs thousands of them while its launch waits): THX_ACG_FAST=0 restores the plain form.
#ifndef THX_ACG_FAST
#define THX_ACG_FAST 1
#endif
__device__ __forceinline__ void inv4_sym(double* o, const double* m)
{
    const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[5], f = m[6], g = m[7], h = m[10], i = m[11], j = m[15];
    // 2x2 minors of the lower two rows / columns
    const double hj = h * j - i * i, fj = f * j - g * i, fi = f * i - g * h, ej = e * j - g * g, ei = e * i - g * f, eh = e * h - f * f;
    const double cj = c * j - d * i, ci = c * i - d * h, cg = c * g - d * f, bj = b * j - d * g, bi = b * i - d * f, bh = b * h - c * f;
    const double bg = b * g - d * e, bf = b * f - c * e;
    const double c00 = e * hj - f * fj + g * fi;
    const double c01 = -(b * hj - f * cj + g * ci);
    const double c02 = b * fj - e * cj + g * cg;
    const double c03 = -(b * fi - e * ci + f * cg);
    const double c11 = a * hj - c * cj + d * ci;
    const double c12 = -(a * fj - b * cj + d * (c * g - d * f));
    const double c13 = a * fi - b * ci + c * cg;
    const double c22 = a * ej - b * bj + d * bg;
    const double c23 = -(a * ei - b * bi + c * bg);
    const double c33 = a * eh - b * bh + c * bf;
    const double rdet = 1.0 / (a * c00 + b * c01 + c * c02 + d * c03);
    o[0] = c00 * rdet; o[1] = o[4] = c01 * rdet; o[2] = o[8] = c02 * rdet; o[3] = o[12] = c03 * rdet;
    o[5] = c11 * rdet; o[6] = o[9] = c12 * rdet; o[7] = o[13] = c13 * rdet;
    o[10] = c22 * rdet; o[11] = o[14] = c23 * rdet; o[15] = c33 * rdet;
}

// inferACG(dmat44&, const dmat4&), src/Geometry/DirectionalStat.cpp:93-145: fixed point B = 4 sum(x x^T / u) / sum(1 / u),
// u = x^T A^-1 x, until sum|A - B| <= 1e-3; returns the LAST-BUT-ONE iterate A as the reference does.  Wave-cooperative:
// lanes stride over the quaternions in LDS, 11 wave sums per round, every lane ends with the same A.
__device__ void infer_acg(double* A, const double* q /*LDS [n][4]*/, int n, int lane, int* roundsOut)
{
    double B[16];
#pragma unroll
    for (int i = 0; i < 16; i++) B[i] = (i % 5 == 0) ? 1.0 : 0.0;
    int rounds = 0;
    double diff;
    do {
#pragma unroll
        for (int i = 0; i < 16; i++) A[i] = B[i];
        double Ainv[16];
#if THX_ACG_FAST
        inv4_sym(Ainv, A);
#else
        inv4(Ainv, A, nullptr);
#endif
        double s[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, nf = 0;
        for (int i = lane; i < n; i += 64) {
            const double* x = q + 4 * i;
            const double u = quad4(x, Ainv), ru = 1.0 / u;
            int e = 0;
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int k = j; k < 4; k++)
#if THX_ACG_FAST
                    s[e++] += (x[j] * x[k]) * ru;
#else
                    s[e++] += (x[j] * x[k]) / u;
#endif
            nf += ru;
        }
#pragma unroll
        for (int e = 0; e < 10; e++) s[e] = wave_sum(s[e]);
        nf = wave_sum(nf);
        int e = 0;
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int k = j; k < 4; k++) { B[j * 4 + k] = s[e] * (4.0 / nf); B[k * 4 + j] = B[j * 4 + k]; e++; }
        diff = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) diff += fabs(A[i] - B[i]);
        rounds++;
    } while (diff > 1e-3 && rounds < 100000);
    if (roundsOut) *roundsOut = rounds;
}

// eigenvector of the largest eigenvalue of a symmetric 4x4 (cyclic Jacobi), for inferACG(dvec4& mean, ...) :224-262
__device__ void sym4_top_eigvec(double* v, const double* Ain)
{
    double A[16], V[16];
    for (int i = 0; i < 16; i++) { A[i] = Ain[i]; V[i] = (i % 5 == 0) ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 64; sweep++) {
        double off = 0;
        for (int p = 0; p < 4; p++) for (int r = p + 1; r < 4; r++) off += A[p * 4 + r] * A[p * 4 + r];
        if (off < 1e-300) break;
        for (int p = 0; p < 4; p++)
            for (int r = p + 1; r < 4; r++) {
                if (fabs(A[p * 4 + r]) < 1e-300) continue;
                const double theta = (A[r * 4 + r] - A[p * 4 + p]) / (2 * A[p * 4 + r]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                const double c = 1 / sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < 4; k++) {
                    const double akp = A[k * 4 + p], akr = A[k * 4 + r];
                    A[k * 4 + p] = c * akp - s * akr; A[k * 4 + r] = s * akp + c * akr;
                }
                for (int k = 0; k < 4; k++) {
                    const double apk = A[p * 4 + k], ark = A[r * 4 + k];
                    A[p * 4 + k] = c * apk - s * ark; A[r * 4 + k] = s * apk + c * ark;
                }
                for (int k = 0; k < 4; k++) {
                    const double vkp = V[k * 4 + p], vkr = V[k * 4 + r];
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

__device__ void cal_vari_R(double* k, double* sq, int n, int lane, const double* __restrict__ symQ, int nSym,
                           unsigned long long seed, unsigned img, unsigned call)
{
    if (nSym > 0) {
        double u4[4];
        draw_u4(u4, seed, img, call, 13u, 0);
        int iA = (int)(u4[0] * n);
        iA = iA >= n ? n - 1 : iA;
        const double anch[4] = {sq[4 * iA], sq[4 * iA + 1], sq[4 * iA + 2], sq[4 * iA + 3]};
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < n; i += 64) sym_counterpart(sq + 4 * i, symQ, nSym, anch);
        __builtin_amdgcn_wave_barrier();
    }
    double A[16], mean[4], cm[4];
    infer_acg(A, sq, n, lane, nullptr);
    sym4_top_eigvec(mean, A);
    cm[0] = mean[0]; cm[1] = -mean[1]; cm[2] = -mean[2]; cm[3] = -mean[3];
    for (int i = lane; i < n; i += 64) { double o[4]; qmul(o, cm, sq + 4 * i); for (int c = 0; c < 4; c++) sq[4 * i + c] = o[c]; }
    __builtin_amdgcn_wave_barrier();
    infer_acg(A, sq, n, lane, nullptr);
    k[0] = A[5] / A[0]; k[1] = A[10] / A[0]; k[2] = A[15] / A[0];
    for (int i = lane; i < n; i += 64) { double o[4]; qmul(o, mean, sq + 4 * i); for (int c = 0; c < 4; c++) sq[4 * i + c] = o[c]; }
    __builtin_amdgcn_wave_barrier();
}

// balanceWeight(PAR_R) + normW (src/Particle.cpp:2333-2343,815-822): w_i = 1 / pdfACG(r_i, A) normalised
__device__ void balance_weight_R(double* w /*LDS*/, const double* q /*LDS*/, int n, int lane)
{
    double A[16], Ainv[16], det;
    infer_acg(A, q, n, lane, nullptr);
    inv4(Ainv, A, &det);
    const double pd = pow(det, -0.5);
    double sum = 0;
    for (int i = lane; i < n; i += 64) {
        const double p = pd * pow(quad4(q + 4 * i, Ainv), -2.0);
        w[i] = 1.0 / p;
        sum += w[i];
    }
    sum = wave_sum(sum);
    for (int i = lane; i < n; i += 64) w[i] /= sum;
}

// column mean / sd (gsl_stats_mean, gsl_stats_sd_m) as two-pass fp64 wave sums
__device__ void col_mean_sd(double& m, double& sd, const double* t /*LDS [n][2]*/, int col, int n, int lane)
{
    double s = 0;
    for (int i = lane; i < n; i += 64) s += t[2 * i + col];
    m = wave_sum(s) / n;
    double ss = 0;
    for (int i = lane; i < n; i += 64) { const double d = t[2 * i + col] - m; ss += d * d; }
    ss = wave_sum(ss);
    sd = sqrt(ss / n * ((double)n / (double)(n - 1)));
}

// balanceWeight(PAR_T) + normW (:2345-2376): w_i = 1 / N2(t_i - m; s0, s1, rho = 0) normalised
__device__ void balance_weight_T(double* w, const double* t, int n, int lane)
{
    double m0, m1, s0, s1;
    col_mean_sd(m0, s0, t, 0, n, lane);
    col_mean_sd(m1, s1, t, 1, n, lane);
    double sum = 0;
    for (int i = lane; i < n; i += 64) {
        const double u = (t[2 * i] - m0) / s0, v = (t[2 * i + 1] - m1) / s1;
        const double p = (1 / (2 * 3.14159265358979323846 * s0 * s1)) * exp(-(u * u + v * v) / 2);
        w[i] = 1.0 / p;
        sum += w[i];
    }
    sum = wave_sum(sum);
    for (int i = lane; i < n; i += 64) w[i] /= sum;
}

// shuffle + systematic resampling of Particle::resample(n, pt) (:1340-1372 with PARTICLE_PRIOR_ONE): elements of width
// W doubles in LDS `val`, weights w, likelihoods u; results written back in place (n points in, n points out).
template <int W>
__device__ void resample(double* val, double* w, double* u, double* tmpVal, double* tmpW, double* tmpU, double* cdf,
                         unsigned* keys, int n, int lane, unsigned long long seed, unsigned img, unsigned call,
                         unsigned purpose)
{
    // gsl_ran_shuffle: a uniformly random permutation -- rank of a random key per element (ties broken by index)
    for (int i = lane; i < n; i += 64) {
        Philox g{(unsigned)seed, (unsigned)(seed >> 32)};
        unsigned c[4] = {img, call, purpose, (unsigned)i};
        g(c);
        keys[i] = c[0];
    }
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < n; i += 64) {
        const unsigned ki = keys[i];
        int rank = 0;
        for (int j = 0; j < n; j++) rank += (keys[j] < ki) || (keys[j] == ki && j < i);
#pragma unroll
        for (int c = 0; c < W; c++) tmpVal[W * rank + c] = val[W * i + c];
        tmpW[rank] = w[i];
        tmpU[rank] = u[i];
    }
    __builtin_amdgcn_wave_barrier();
    // w *= u; w /= sum; cdf = cumsum(w); cdf /= cdf[n-1]
    double sum = 0;
    for (int i = lane; i < n; i += 64) sum += tmpW[i] * tmpU[i];
    sum = wave_sum(sum);
    if (lane == 0) {
        double acc = 0;
        for (int i = 0; i < n; i++) { acc += (tmpW[i] * tmpU[i]) / sum; cdf[i] = acc; }
        const double last = cdf[n - 1];
        for (int i = 0; i < n; i++) cdf[i] /= last;
    }
    __builtin_amdgcn_wave_barrier();
    double u4[4];
    draw_u4(u4, seed, img, call, purpose + 1, 0);
    const double u0 = u4[0] * (1.0 / n);  // gsl_ran_flat(engine, 0, 1.0 / n)
    double ws = 0;
    for (int j = lane; j < n; j += 64) {
        const double uj = u0 + j * 1.0 / n;
        int lo = 0, hi = n - 1;  // smallest i with !(uj > cdf[i])
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (uj > cdf[mid]) lo = mid + 1; else hi = mid; }
#pragma unroll
        for (int c = 0; c < W; c++) val[W * j + c] = tmpVal[W * lo + c];
        w[j] = 1.0 / tmpU[lo];
        ws += w[j];
    }
    ws = wave_sum(ws);
    for (int j = lane; j < n; j += 64) w[j] /= ws;
    __builtin_amdgcn_wave_barrier();
}

struct PfArgs {
    double* r;        // [nImg][nR][4] quaternions
    double* t;        // [nImg][nT][2] shifts
    double* wR;       // [nImg][nR] priors
    double* wT;       // [nImg][nT]
    double* k123;     // [nImg][3]
    double* s01;      // [nImg][2]
    double* topR;     // [nImg][4]
    double* topT;     // [nImg][2]
    const float* uR;  // [nImg][nR] likelihood weights from the E-step
    const float* uT;  // [nImg][nT]
    int nR, nT;
    double pfR, pfT, transS, transM, peakFactorR;
    unsigned long long seed;
    unsigned call;
    const int* active;   // [nImg] or NULL: images with active[img] == 0 are left untouched (per-image stop rule)
    const double* symQ;  // DEVICE [nSym][4] Symmetry::quat(i), or NULL (C1)
    int nSym;
    unsigned img0;       // the launch's first image in the numbering of the Philox streams (thx_pf_ctx)
};

// Particle::perturb(pf, PAR_R) + perturb(pf, PAR_T), src/Particle.cpp:1149-1272 (MODE_3D)
__global__ __launch_bounds__(64) void k_pf_perturb(PfArgs a)
{
    __shared__ double sq[kPfMax * 4], st[kPfMax * 2], sw[kPfMax];
    const int img = blockIdx.x, lane = threadIdx.x;
    if (a.active && !a.active[img]) return;
    const unsigned pimg = a.img0 + (unsigned)img;
    const int nR = a.nR, nT = a.nT;
    if (nR > 0) {
        double* r = a.r + (size_t)img * nR * 4;
        const double* k = a.k123 + 3 * (size_t)img;
        // sampleACG(d, pf^2 min(1, k1), pf^2 min(1, k2), pf^2 min(1, k3), nR): L = chol(diag(1, ...)) = sqrt of the diagonal
        const double pf2 = a.pfR * a.pfR;
        const double l1 = sqrt(pf2 * fmin(1.0, k[0])), l2 = sqrt(pf2 * fmin(1.0, k[1])), l3 = sqrt(pf2 * fmin(1.0, k[2]));
        // mean = inferACG(mean, _r) of the cloud BEFORE the perturbation (PARTICLE_ROT_MEAN_USING_STAT_PERTURB, :1203-1207)
        for (int i = lane; i < nR * 4; i += 64) sq[i] = r[i];
        __builtin_amdgcn_wave_barrier();
        double A[16], mean[4], cm[4];
        infer_acg(A, sq, nR, lane, nullptr);
        sym4_top_eigvec(mean, A);
        cm[0] = mean[0]; cm[1] = -mean[1]; cm[2] = -mean[2]; cm[3] = -mean[3];
        for (int i = lane; i < nR; i += 64) {
            double g[4];
            draw_n4(g, a.seed, pimg, a.call, 0, i);
            double v[4] = {g[0], l1 * g[1], l2 * g[2], l3 * g[3]};
            const double nrm = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
#pragma unroll
            for (int c = 0; c < 4; c++) v[c] /= nrm;
            // quat = conj(mean) * quat; quat = pert * quat; quat = mean * quat (:1211-1239; quaternion_mul(dst, a, b) = a * b):
            // the anisotropic perturbation acts in the frame calVari estimated k1..k3 in
            double o[4], o2[4];
            qmul(o, cm, sq + 4 * i);
            qmul(o2, v, o);
            qmul(o, mean, o2);
            if (a.nSym > 0) sym_counterpart(o, a.symQ, a.nSym, mean);   // symmetrise(&mean), :1234
#pragma unroll
            for (int c = 0; c < 4; c++) { r[4 * i + c] = o[c]; sq[4 * i + c] = o[c]; }
        }
        __builtin_amdgcn_wave_barrier();
        balance_weight_R(sw, sq, nR, lane);
        for (int i = lane; i < nR; i += 64) a.wR[(size_t)img * nR + i] = sw[i];
        __builtin_amdgcn_wave_barrier();
    }
    if (nT > 0) {
        double* t = a.t + (size_t)img * nT * 2;
        const double s0 = a.s01[2 * (size_t)img], s1 = a.s01[2 * (size_t)img + 1];
        for (int i = lane; i < nT; i += 64) {
            double g[4];
            draw_n4(g, a.seed, pimg, a.call, 1, i);
            // gsl_ran_bivariate_gaussian(engine, s0, s1, rho = 0, &x, &y); t += (x, y) * pf
            double x = t[2 * i] + s0 * g[0] * a.pfT, y = t[2 * i + 1] + s1 * g[1] * a.pfT;
            // reCentre (:2473-2495): points beyond transM are redrawn from N(0, transS^2 I)
            if (gsl_hypot_(x, y) > a.transM) { x = a.transS * g[2]; y = a.transS * g[3]; }
            t[2 * i] = x; t[2 * i + 1] = y;
            st[2 * i] = x; st[2 * i + 1] = y;
        }
        __builtin_amdgcn_wave_barrier();
        balance_weight_T(sw, st, nT, lane);
        for (int i = lane; i < nT; i += 64) a.wT[(size_t)img * nT + i] = sw[i];
    }
}

// after the likelihood: setUR/setUT (+ keepHalfHeightPeak(PAR_R)), calRank1st, calVari, resample for R and T
// (src/Optimiser.cpp:1410-1475)
__global__ __launch_bounds__(64) void k_pf_update(PfArgs a)
{
    __shared__ double sq[kPfMax * 4], sw[kPfMax], su[kPfMax], tq[kPfMax * 4], tw[kPfMax], tu[kPfMax], cdf[kPfMax];
    __shared__ unsigned keys[kPfMax];
    const int img = blockIdx.x, lane = threadIdx.x;
    if (a.active && !a.active[img]) return;
    const unsigned pimg = a.img0 + (unsigned)img;
    const int nR = a.nR, nT = a.nT;
    if (nR > 0) {
        double* r = a.r + (size_t)img * nR * 4;
        double umax = -1.0;
        int imax = 0;
        for (int i = lane; i < nR; i += 64) {
#pragma unroll
            for (int c = 0; c < 4; c++) sq[4 * i + c] = r[4 * i + c];
            sw[i] = a.wR[(size_t)img * nR + i];
            const double u = (double)a.uR[(size_t)img * nR + i];
            su[i] = u;
            if (u > umax) { umax = u; imax = i; }
        }
        // d_value_max_index: first index of the maximum
        for (int o = 32; o > 0; o >>= 1) {
            const double ou = __shfl_xor(umax, o, 64);
            const int oi = __shfl_xor(imax, o, 64);
            if (ou > umax || (ou == umax && oi < imax)) { umax = ou; imax = oi; }
        }
        __builtin_amdgcn_wave_barrier();
        if (a.peakFactorR >= 0) {  // keepHalfHeightPeak(PAR_R), :1964-1990
            const double hh = umax * a.peakFactorR;
            for (int i = lane; i < nR; i += 64) su[i] = (su[i] < hh) ? 0.0 : su[i] - hh;
        }
        __builtin_amdgcn_wave_barrier();
        // calVari(PAR_R), :1020-1080 (with a point group: counterparts next to a random anchor first)
        {
            double kk[3];
            cal_vari_R(kk, sq, nR, lane, a.symQ, a.nSym, a.seed, pimg, a.call);
            if (lane == 0) {
                a.k123[3 * (size_t)img] = kk[0];
                a.k123[3 * (size_t)img + 1] = kk[1];
                a.k123[3 * (size_t)img + 2] = kk[2];
            }
        }
        // calRank1st(PAR_R) (:990-1002), taken again by resample() after calVari's round trip (:1381-1383): _topR = _r.row(iMax)
        if (lane < 4) a.topR[4 * (size_t)img + lane] = sq[4 * imax + lane];
        __builtin_amdgcn_wave_barrier();
        resample<4>(sq, sw, su, tq, tw, tu, cdf, keys, nR, lane, a.seed, pimg, a.call, 2);
        for (int i = lane; i < nR; i += 64) {
#pragma unroll
            for (int c = 0; c < 4; c++) r[4 * i + c] = sq[4 * i + c];
            a.wR[(size_t)img * nR + i] = sw[i];
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (nT > 0) {
        double* t = a.t + (size_t)img * nT * 2;
        double umax = -1.0;
        int imax = 0;
        for (int i = lane; i < nT; i += 64) {
            sq[2 * i] = t[2 * i]; sq[2 * i + 1] = t[2 * i + 1];
            sw[i] = a.wT[(size_t)img * nT + i];
            const double u = (double)a.uT[(size_t)img * nT + i];
            su[i] = u;
            if (u > umax) { umax = u; imax = i; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const double ou = __shfl_xor(umax, o, 64);
            const int oi = __shfl_xor(imax, o, 64);
            if (ou > umax || (ou == umax && oi < imax)) { umax = ou; imax = oi; }
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < 2) a.topT[2 * (size_t)img + lane] = sq[2 * imax + lane];
        double m, s0, s1;  // calVari(PAR_T): gsl_stats_sd per column
        col_mean_sd(m, s0, sq, 0, nT, lane);
        col_mean_sd(m, s1, sq, 1, nT, lane);
        if (lane == 0) { a.s01[2 * (size_t)img] = s0; a.s01[2 * (size_t)img + 1] = s1; }
        __builtin_amdgcn_wave_barrier();
        resample<2>(sq, sw, su, tq, tw, tu, cdf, keys, nT, lane, a.seed, pimg, a.call, 4);
        for (int i = lane; i < nT; i += 64) {
            t[2 * i] = sq[2 * i]; t[2 * i + 1] = sq[2 * i + 1];
            a.wT[(size_t)img * nT + i] = sw[i];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Support points of the local search after a GLOBAL scan, src/Optimiser.cpp:953-1008: the particle holds the nIn scanned grid
// points with their scan weights u (uniform priors 1 / nIn); keepHalfHeightPeak(PAR_R) (peak factor; not for PAR_T), then
// resample(mLR, PAR_R) / resample(mLT, PAR_T) -- shuffle, _top = first largest u of the shuffled list, w *= u, systematic draw of
// nOut < nIn points (src/Particle.cpp:1291-1430) -- then calVari(PAR_R) / calVari(PAR_T) on the new points.
// One wave per image.  The shuffle of 10 000 points is a bitonic sort of (Philox key << 32 | index) in LDS (ties by index, as
// the n^2 rank of resample<> above); the sorted pairs are then compacted in place to the indices, and the second half of the
// same LDS holds the shuffled u for the one sequential pass the cumulative sum needs.
// ---------------------------------------------------------------------------------------------
// gsl_ran_shuffle as a sort: pairs[i] = (Philox key of element i) << 32 | i, sorted ascending (ties by index); NT threads of
// one workgroup (64: one wave, no workgroup barrier)
template <int NT>
__device__ void shuffle_sort(unsigned long long* pairs /*LDS [NP]*/, int NP, int nIn, int tid, unsigned long long seed, unsigned img,
                             unsigned call, unsigned purpose)
{
    auto bar = [] { if (NT > 64) __syncthreads(); else __builtin_amdgcn_wave_barrier(); };
    for (int i = tid; i < NP; i += NT) {
        unsigned long long pr = ~0ull;
        if (i < nIn) {
            Philox g{(unsigned)seed, (unsigned)(seed >> 32)};
            unsigned c[4] = {img, call, purpose, (unsigned)i};
            g(c);
            pr = ((unsigned long long)c[0] << 32) | (unsigned)i;
        }
        pairs[i] = pr;
    }
    bar();
    for (int k = 2; k <= NP; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int q = tid; q < NP / 2; q += NT) {   // the q-th compare-exchange of this pass: (t, t + j)
                const int t = ((q & ~(j - 1)) << 1) | (q & (j - 1)), x = t + j;
                const unsigned long long a = pairs[t], b = pairs[x];
                if ((a > b) == ((t & k) == 0)) { pairs[t] = b; pairs[x] = a; }
            }
            bar();
        }
}

template <int W>
__device__ void resample_from_grid(double* outVal /*LDS [nOut][W]*/, double* outW /*LDS [nOut]*/, double* top /*global [W]*/,
                                   const double* __restrict__ grid /*global [nIn][W]*/, const float* __restrict__ u /*global [nIn]*/,
                                   double peakFactor, int nIn, int nOut, unsigned long long* pairs /*LDS [NP]*/, int NP, int lane,
                                   unsigned long long seed, unsigned img, unsigned call, unsigned purpose, bool presorted)
{
    // keepHalfHeightPeak: hh = max(u) * peakFactor; u < hh -> 0, else u - hh   (:1964-2011)
    double umax = -1.0;
    for (int i = lane; i < nIn; i += 64) umax = fmax(umax, (double)u[i]);
    for (int o = 32; o > 0; o >>= 1) umax = fmax(umax, __shfl_xor(umax, o, 64));
    const double hh = peakFactor >= 0 ? umax * peakFactor : 0.0;
    const bool peak = peakFactor >= 0;
    if (!presorted) shuffle_sort<64>(pairs, NP, nIn, lane, seed, img, call, purpose);
    // pairs -> indices (first half of the buffer), then the shuffled u behind them
    unsigned* idx = reinterpret_cast<unsigned*>(pairs);
    float* uv = reinterpret_cast<float*>(pairs) + NP;
    for (int c0 = 0; c0 < NP; c0 += 64) {
        const unsigned e = (unsigned)pairs[c0 + lane];
        __builtin_amdgcn_wave_barrier();
        idx[c0 + lane] = e;
        __builtin_amdgcn_wave_barrier();
    }
    for (int p = lane; p < nIn; p += 64) uv[p] = u[idx[p]];
    __builtin_amdgcn_wave_barrier();
    auto ud = [&](int p) -> double {   // the shuffled list's weight after keepHalfHeightPeak, in double as Particle holds it
        const double v = (double)uv[p];
        return peak ? (v < hh ? 0.0 : v - hh) : v;
    };
    // _top: first largest u of the shuffled list
    int pTop = INT_MAX;
    for (int p = lane; p < nIn; p += 64)
        if ((double)uv[p] == umax) { pTop = p; break; }
    for (int o = 32; o > 0; o >>= 1) pTop = min(pTop, __shfl_xor(pTop, o, 64));
    if (lane < W) top[lane] = grid[(size_t)idx[pTop] * W + lane];
    // w *= u; w /= sum; cdf = cumsum(w); cdf /= cdf[n - 1]; systematic draw
    const double w0 = 1.0 / nIn;
    double sum = 0;
    for (int p = lane; p < nIn; p += 64) sum += w0 * ud(p);
    sum = wave_sum(sum);
    double u4[4];
    draw_u4(u4, seed, img, call, purpose + 1, 0);
    const double u0 = u4[0] * (1.0 / nOut);   // gsl_ran_flat(engine, 0, 1.0 / n)
    int* src = reinterpret_cast<int*>(outW);  // (the source positions, for the wave)
    // The cumulative sum runs in the reference's order (serial over the shuffled list), 64 points at a time: the divisions of a
    // chunk are done by the 64 lanes, the 64 additions by every lane alike (shuffles), lane i keeping the running sum after point i.
    double last = 0;
    for (int c0 = 0; c0 < nIn; c0 += 64) {
        const int p = c0 + lane;
        const double term = p < nIn ? (w0 * ud(p)) / sum : 0.0;
        const int m = nIn - c0 < 64 ? nIn - c0 : 64;
        for (int i = 0; i < m; i++) last += __shfl(term, i, 64);
    }
    {
        double acc = 0;
        int j = 0;
        for (int c0 = 0; c0 < nIn && j < nOut; c0 += 64) {
            const int p = c0 + lane;
            const double term = p < nIn ? (w0 * ud(p)) / sum : 0.0;
            const int m = nIn - c0 < 64 ? nIn - c0 : 64;
            double mine = 0;
            for (int i = 0; i < m; i++) { acc += __shfl(term, i, 64); if (lane == i) mine = acc; }
            const double cdf = mine / last;
            while (j < nOut) {   // smallest p with !(uj > cdf[p])
                const double uj = u0 + j * 1.0 / nOut;
                const unsigned long long hit = __ballot(lane < m && !(uj > cdf));
                if (!hit) break;
                if (lane == 0) src[2 * j] = c0 + __ffsll((long long)hit) - 1;
                j++;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    double ws = 0;
    int mySrc[4];
    for (int j = lane, q = 0; j < nOut; j += 64, q++) mySrc[q] = src[2 * j];
    __builtin_amdgcn_wave_barrier();
    for (int j = lane, q = 0; j < nOut; j += 64, q++) {
#pragma unroll
        for (int c = 0; c < W; c++) outVal[W * j + c] = grid[(size_t)idx[mySrc[q]] * W + c];
        outW[j] = 1.0 / ud(mySrc[q]);
        ws += outW[j];
    }
    ws = wave_sum(ws);
    for (int j = lane; j < nOut; j += 64) outW[j] /= ws;
    __builtin_amdgcn_wave_barrier();
}

struct ScanSupportArgs {
    double *r, *t, *wR, *wT, *k123, *s01, *topR, *topT;
    const double *gridR, *gridT;
    const float *uR, *uT;   // [nK][nImg][nRin], [nK][nImg][nTin]
    const int* cls;         // [nImg] or NULL (class 0)
    int nImg, nRin, nTin, mLR, mLT, NPR, NPT;
    int rowStride;          // images per class row of uR / uT (the scan weights of a BATCH of a larger shard: its image count)
    double peakFactorR, minK, minS;
    unsigned long long seed;
    unsigned call;
    const double* symQ;
    int nSym;
    unsigned img0;
};

constexpr int kScanSupThreads = 256;   // the 10 000-point shuffle is sorted by four waves; one wave does the rest
__global__ __launch_bounds__(kScanSupThreads) void k_pf_scan_support(ScanSupportArgs a)
{
    extern __shared__ unsigned long long pairs[];
    __shared__ double sq[kPfMax * 4], sw[kPfMax];
    const int img = blockIdx.x, lane = threadIdx.x;
    const unsigned pimg = a.img0 + (unsigned)img;
    const size_t row = (size_t)(a.cls ? a.cls[img] : 0) * a.rowStride + img;
    // ---- rotations ----
    shuffle_sort<kScanSupThreads>(pairs, a.NPR, a.nRin, threadIdx.x, a.seed, pimg, a.call, 2);
    if (threadIdx.x >= 64) return;   // (no workgroup barrier below this line)
    resample_from_grid<4>(sq, sw, a.topR + 4 * (size_t)img, a.gridR, a.uR + row * a.nRin, a.peakFactorR, a.nRin, a.mLR, pairs, a.NPR, lane,
                          a.seed, pimg, a.call, 2, true);
    {   // calVari(PAR_R) on the new points, as in k_pf_update
        double kk[3];
        cal_vari_R(kk, sq, a.mLR, lane, a.symQ, a.nSym, a.seed, pimg, a.call);
        if (lane == 0) {   // setK1..3(max(minimum of the scanning phase, k)), src/Optimiser.cpp:1032-1050
            a.k123[3 * (size_t)img] = fmax(a.minK, kk[0]);
            a.k123[3 * (size_t)img + 1] = fmax(a.minK, kk[1]);
            a.k123[3 * (size_t)img + 2] = fmax(a.minK, kk[2]);
        }
    }
    for (int i = lane; i < a.mLR; i += 64) {
#pragma unroll
        for (int c = 0; c < 4; c++) a.r[((size_t)img * a.mLR + i) * 4 + c] = sq[4 * i + c];
        a.wR[(size_t)img * a.mLR + i] = sw[i];
    }
    __builtin_amdgcn_wave_barrier();
    // ---- shifts ----
    resample_from_grid<2>(sq, sw, a.topT + 2 * (size_t)img, a.gridT, a.uT + row * a.nTin, -1.0, a.nTin, a.mLT, pairs, a.NPT, lane, a.seed,
                          pimg, a.call, 4, false);
    double m, s0, s1;
    col_mean_sd(m, s0, sq, 0, a.mLT, lane);
    col_mean_sd(m, s1, sq, 1, a.mLT, lane);
    if (lane == 0) { a.s01[2 * (size_t)img] = fmax(a.minS, s0); a.s01[2 * (size_t)img + 1] = fmax(a.minS, s1); }   // setS0 / setS1, :1067-1079
    for (int i = lane; i < a.mLT; i += 64) {
        a.t[((size_t)img * a.mLT + i) * 2] = sq[2 * i];
        a.t[((size_t)img * a.mLT + i) * 2 + 1] = sq[2 * i + 1];
        a.wT[(size_t)img * a.mLT + i] = sw[i];
    }
}

// inferACG / statistics probe for the parity tests: A [nImg][16], mean [nImg][4], k123 [nImg][3], wBal [nImg][n]
__global__ __launch_bounds__(64) void k_pf_acg_stats(double* __restrict__ Aout, double* __restrict__ meanOut,
                                                     double* __restrict__ kOut, double* __restrict__ wBal,
                                                     int* __restrict__ roundsOut, const double* __restrict__ q, int n)
{
    __shared__ double sq[kPfMax * 4], sw[kPfMax];
    const int img = blockIdx.x, lane = threadIdx.x;
    for (int i = lane; i < n * 4; i += 64) sq[i] = q[(size_t)img * n * 4 + i];
    __builtin_amdgcn_wave_barrier();
    double A[16], mean[4], cm[4];
    int rounds = 0;
    infer_acg(A, sq, n, lane, &rounds);
    if (lane == 0 && roundsOut) roundsOut[2 * (size_t)img] = rounds;
    if (lane < 16) Aout[16 * (size_t)img + lane] = A[lane];
    sym4_top_eigvec(mean, A);
    if (lane < 4) meanOut[4 * (size_t)img + lane] = mean[lane];
    balance_weight_R(sw, sq, n, lane);
    for (int i = lane; i < n; i += 64) wBal[(size_t)img * n + i] = sw[i];
    cm[0] = mean[0]; cm[1] = -mean[1]; cm[2] = -mean[2]; cm[3] = -mean[3];
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < n; i += 64) { double o[4]; qmul(o, cm, sq + 4 * i); for (int c = 0; c < 4; c++) sq[4 * i + c] = o[c]; }
    __builtin_amdgcn_wave_barrier();
    infer_acg(A, sq, n, lane, &rounds);
    if (lane == 0 && roundsOut) roundsOut[2 * (size_t)img + 1] = rounds;
    if (lane == 0) { kOut[3 * (size_t)img] = A[5] / A[0]; kOut[3 * (size_t)img + 1] = A[10] / A[0]; kOut[3 * (size_t)img + 2] = A[15] / A[0]; }
}

// The per-image stop rule of the local search, src/Optimiser.cpp:1510-1615 (MODE_3D, not OPTIMISER_COMPRESS_CRITERIA): after
// the phase with index `phase` >= MIN_N_PHASE_PER_ITER_LOCAL the variances of the filter (k1..k3, s0, s1; the defocus
// variance `dVari` only under CTF search) are compared with the smallest ever seen; no decrease by PARTICLE_FILTER_DECREASE_
// FACTOR (0.95; squared for k1..k3) in N_PHASE_WITH_NO_VARI_DECREASE = 1 phase ends the image's search.
// state [nImg][8]: k1, k2, k3, s0, s1, dVari minima, nPhaseWithNoVariDecrease, (unused); nP [nImg]: phase at which it stopped.
__global__ void k_pf_stop_rule(int* __restrict__ active, int* __restrict__ nP, double* __restrict__ state,
                               const double* __restrict__ k123, const double* __restrict__ s01, const double* __restrict__ sD,
                               int phase, int nImg, int* __restrict__ nActive)
{
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nImg || !active[l]) return;
    double* st = state + (size_t)l * 8;
    const double k1 = k123[3 * l], k2 = k123[3 * l + 1], k3 = k123[3 * l + 2], s0 = s01[2 * l], s1 = s01[2 * l + 1];
    const double d = sD ? sD[l] : 0.0;
    const double f = 0.95, f2 = 0.95 * 0.95;   // PARTICLE_FILTER_DECREASE_FACTOR, gsl_pow_2 of it
    int noDec = (int)st[6];
    if ((k1 < st[0] * f2) || (k2 < st[1] * f2) || (k3 < st[2] * f2) || (s0 < st[3] * f) || (s1 < st[4] * f) || (d < st[5] * f))
        noDec = 0;   // there is still room for searching
    else
        noDec += 1;
    if (k1 < st[0]) st[0] = k1;
    if (k2 < st[1]) st[1] = k2;
    if (k3 < st[2]) st[2] = k3;
    if (s0 < st[3]) st[3] = s0;
    if (s1 < st[4]) st[4] = s1;
    if (d < st[5]) st[5] = d;
    st[6] = (double)noDec;
    if (noDec == 1) {   // N_PHASE_WITH_NO_VARI_DECREASE
        active[l] = 0;
        nP[l] = phase;
    } else {
        atomicAdd(nActive, 1);
    }
}

__global__ void k_pf_stop_init(int* __restrict__ active, int* __restrict__ nP, double* __restrict__ state, double transS,
                               double ctfRefineS, int nImg)
{
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nImg) return;
    active[l] = 1;
    nP[l] = 0;
    double* st = state + (size_t)l * 8;
    st[0] = st[1] = st[2] = 1.0;                // k1 = k2 = k3 = 1 (src/Optimiser.cpp:1178-1180)
    st[3] = st[4] = 5 * transS;                 // tVariS0 = tVariS1 = 5 * transS
    st[5] = 5 * ctfRefineS;                     // dVari
    st[6] = 0.0; st[7] = 0.0;
}

// The defocus factor of the CTF search (PAR_D), one thread per image (mLD <= 64 support points):
//   mode 0  Particle::initD(nD, sD), src/Particle.cpp:281-311 (PARTICLE_DEFOCUS_INIT_GAUSSIAN): d_i = 1 + N(0, sD^2)
//   mode 1  Particle::perturb(pf, PAR_D), :1273-1287: d_i += N(0, s^2) pf   (s = the last calVari(PAR_D))
// followed by balanceWeight(PAR_D) + normW (:2405-2440): w_i = 1 / N(d_i - mean; sd), 1 when sd == 0.
// Philox stream (seed, image, call, 10, i).
__global__ void k_pf_perturb_d(double* __restrict__ d, double* __restrict__ wD, const double* __restrict__ sD, int nImg, int nD,
                               double scale, int init, unsigned long long seed, unsigned call, const int* __restrict__ active,
                               unsigned img0)
{
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nImg || (active && !active[l])) return;
    double* dd = d + (size_t)l * nD;
    double* w = wD + (size_t)l * nD;
    const double s = init ? scale : sD[l];
    for (int i = 0; i < nD; i++) {
        double g[4];
        draw_n4(g, seed, img0 + (unsigned)l, call, 10u, (unsigned)i);
        dd[i] = init ? 1.0 + s * g[0] : dd[i] + (s * g[0]) * scale;      // gsl_ran_gaussian(engine, sigma) = sigma * n
    }
    // gsl_stats_mean / gsl_stats_sd_m (recurrences in long double in GSL; two-pass in double here: 1e-16)
    double m = 0;
    for (int i = 0; i < nD; i++) m += dd[i];
    m /= nD;
    double sd = 0;
    if (nD > 1) {
        double v = 0;
        for (int i = 0; i < nD; i++) v += (dd[i] - m) * (dd[i] - m);
        sd = sqrt(v / nD * ((double)nD / (double)(nD - 1)));
    }
    double sum = 0;
    for (int i = 0; i < nD; i++) {
        if (sd == 0) w[i] = 1.0;
        else {
            const double u = (dd[i] - m) / fabs(sd);
            w[i] = 1.0 / ((1.0 / (sqrt(2 * 3.14159265358979323846) * fabs(sd))) * exp(-u * u / 2));   // gsl_ran_gaussian_pdf
        }
        sum += w[i];
    }
    for (int i = 0; i < nD; i++) w[i] /= sum;
}

// after the likelihoods of a CTF-search phase (src/Optimiser.cpp:1424-1436,1465-1470): setUD, [keepHalfHeightPeak(PAR_D) --
// OPTIMISER_PEAK_FACTOR_D is off], calRank1st(PAR_D), calVari(PAR_D) (gsl_stats_sd, 0 for one point), resample(mLD, PAR_D).
// Philox streams (seed, image, call, 11 = shuffle keys / 12 = u0).
__global__ void k_pf_update_d(double* __restrict__ d, double* __restrict__ wD, const float* __restrict__ uD, double* __restrict__ sD,
                              double* __restrict__ topD, int nImg, int nD, unsigned long long seed, unsigned call,
                              const int* __restrict__ active, unsigned img0)
{
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nImg || (active && !active[l])) return;
    double dd[64], w[64], u[64], sv[64], sw[64], su[64], cdf[64];
    unsigned key[64];
    int imax = 0;
    for (int i = 0; i < nD; i++) {
        dd[i] = d[(size_t)l * nD + i]; w[i] = wD[(size_t)l * nD + i]; u[i] = (double)uD[(size_t)l * nD + i];
        if (u[i] > u[imax]) imax = i;
    }
    topD[l] = dd[imax];
    double m = 0;
    for (int i = 0; i < nD; i++) m += dd[i];
    m /= nD;
    double v = 0;
    for (int i = 0; i < nD; i++) v += (dd[i] - m) * (dd[i] - m);
    sD[l] = nD > 1 ? sqrt(v / nD * ((double)nD / (double)(nD - 1))) : 0.0;
    for (int i = 0; i < nD; i++) {
        Philox g{(unsigned)seed, (unsigned)(seed >> 32)};
        unsigned c[4] = {img0 + (unsigned)l, call, 11u, (unsigned)i};
        g(c);
        key[i] = c[0];
    }
    for (int i = 0; i < nD; i++) {
        int rank = 0;
        for (int j = 0; j < nD; j++) rank += (key[j] < key[i]) || (key[j] == key[i] && j < i);
        sv[rank] = dd[i]; sw[rank] = w[i]; su[rank] = u[i];
    }
    double sum = 0, acc = 0;
    for (int i = 0; i < nD; i++) sum += sw[i] * su[i];
    for (int i = 0; i < nD; i++) { acc += (sw[i] * su[i]) / sum; cdf[i] = acc; }
    const double last = cdf[nD - 1];
    for (int i = 0; i < nD; i++) cdf[i] /= last;
    double d4[4];
    draw_u4(d4, seed, img0 + (unsigned)l, call, 12u, 0);
    const double u0 = d4[0] * (1.0 / nD);
    int i = 0;
    double ws = 0;
    for (int j = 0; j < nD; j++) {
        const double uj = u0 + j * 1.0 / nD;
        while (uj > cdf[i]) i++;
        d[(size_t)l * nD + j] = sv[i];
        w[j] = 1.0 / su[i];
        ws += w[j];
    }
    for (int j = 0; j < nD; j++) wD[(size_t)l * nD + j] = w[j] / ws;
}

// Class selection after the global scan, src/Optimiser.cpp:925-952: setUC(wC) -> setPeakFactor(PAR_C) (PARTICLE_PEAK_FACTOR_C:
// _peakFactorC = PEAK_FACTOR_C = 1 - 1e-2) -> keepHalfHeightPeak(PAR_C) -> resample(k, PAR_C) (shuffle, w *= u, systematic
// resampling, src/Particle.cpp:1296-1338) -> Particle::rand(cls) (uniform pick among the k resampled classes, :2109-2120).
// One thread per image (k <= 64 classes): Philox streams (seed, image, call, 6 / 7 / 8, .).
constexpr int kMaxClasses = 64;
__global__ void k_pf_class_select(int* __restrict__ cls, const float* __restrict__ uC, const double* __restrict__ wC, int nImg, int nK,
                                  double peakFactorC, unsigned long long seed, unsigned call, unsigned img0)
{
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nImg) return;
    const unsigned pl = img0 + (unsigned)l;
    double u[kMaxClasses], w[kMaxClasses], su[kMaxClasses], sw[kMaxClasses];
    int sc[kMaxClasses];
    unsigned key[kMaxClasses];
    double umax = -1.0;
    for (int i = 0; i < nK; i++) {
        u[i] = (double)uC[(size_t)l * nK + i];
        w[i] = wC ? wC[(size_t)l * nK + i] : 1.0 / nK;
        if (u[i] > umax) umax = u[i];
    }
    const double hh = umax * peakFactorC;
    for (int i = 0; i < nK; i++) u[i] = (u[i] < hh) ? 0.0 : u[i] - hh;
    for (int i = 0; i < nK; i++) {
        Philox g{(unsigned)seed, (unsigned)(seed >> 32)};
        unsigned c[4] = {pl, call, 6u, (unsigned)i};
        g(c);
        key[i] = c[0];
    }
    for (int i = 0; i < nK; i++) {
        int rank = 0;
        for (int j = 0; j < nK; j++) rank += (key[j] < key[i]) || (key[j] == key[i] && j < i);
        sc[rank] = i; sw[rank] = w[i]; su[rank] = u[i];
    }
    double sum = 0;
    for (int i = 0; i < nK; i++) sum += sw[i] * su[i];
    double cdf[kMaxClasses], acc = 0;
    for (int i = 0; i < nK; i++) { acc += (sw[i] * su[i]) / sum; cdf[i] = acc; }
    const double last = cdf[nK - 1];
    for (int i = 0; i < nK; i++) cdf[i] /= last;
    double d4[4];
    draw_u4(d4, seed, pl, call, 7u, 0);
    const double u0 = d4[0] * (1.0 / nK);
    draw_u4(d4, seed, pl, call, 8u, 0);
    int pick = (int)(d4[0] * nK);          // gsl_rng_uniform_int(engine, _nC) among the resampled classes
    pick = pick >= nK ? nK - 1 : pick;
    int i = 0, chosen = sc[0];
    for (int j = 0; j < nK; j++) {
        const double uj = u0 + j * 1.0 / nK;
        while (uj > cdf[i]) i++;
        if (j == pick) chosen = sc[i];
    }
    cls[l] = chosen;
}

// Particle::calVari(PAR_R) + calVari(PAR_T) on their own (Particle::load, src/Particle.cpp:400-520: the filter's spread before its
// first phase): r is replaced by the symmetrised / round-tripped support points as the reference's calVari leaves _r
__global__ __launch_bounds__(64) void k_pf_cal_vari(PfArgs a)
{
    __shared__ double sq[kPfMax * 4];
    const int img = blockIdx.x, lane = threadIdx.x;
    const unsigned pimg = a.img0 + (unsigned)img;
    if (a.nR > 0) {
        double* r = a.r + (size_t)img * a.nR * 4;
        for (int i = lane; i < a.nR * 4; i += 64) sq[i] = r[i];
        __builtin_amdgcn_wave_barrier();
        double kk[3];
        cal_vari_R(kk, sq, a.nR, lane, a.symQ, a.nSym, a.seed, pimg, a.call);
        if (lane < 3) a.k123[3 * (size_t)img + lane] = kk[lane];
        if (a.nSym > 0) for (int i = lane; i < a.nR * 4; i += 64) r[i] = sq[i];
        __builtin_amdgcn_wave_barrier();
    }
    if (a.nT > 0) {
        const double* t = a.t + (size_t)img * a.nT * 2;
        for (int i = lane; i < a.nT * 2; i += 64) sq[i] = t[i];
        __builtin_amdgcn_wave_barrier();
        double m, s0, s1;
        col_mean_sd(m, s0, sq, 0, a.nT, lane);
        col_mean_sd(m, s1, sq, 1, a.nT, lane);
        if (lane == 0) { a.s01[2 * (size_t)img] = s0; a.s01[2 * (size_t)img + 1] = s1; }
    }
}

// Particle::symmetrise(anchor), src/Particle.cpp:2445-2470, one thread per quaternion; anchor [nImg][4] or NULL = ANCHOR_POINT_2
__global__ void k_pf_symmetrise(double* __restrict__ r, const double* __restrict__ anchor, size_t nImg, int nR,
                                const double* __restrict__ symQ, int nSym)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nImg * nR) return;
    const size_t l = e / nR;
    double an[4] = {1, 0, 0, 0};
    if (anchor) { an[0] = anchor[4 * l]; an[1] = anchor[4 * l + 1]; an[2] = anchor[4 * l + 2]; an[3] = anchor[4 * l + 3]; }
    double q[4] = {r[4 * e], r[4 * e + 1], r[4 * e + 2], r[4 * e + 3]};
    sym_counterpart(q, symQ, nSym, an);
    r[4 * e] = q[0]; r[4 * e + 1] = q[1]; r[4 * e + 2] = q[2]; r[4 * e + 3] = q[3];
}

}  // namespace thx

using namespace thx;

namespace {
// Symmetry::init(const char sym[]) on the host -- see thx_symmetry_host below
struct SymEntry { int fold; double ax[3]; };
void mat33_mul(double* d, const double* a, const double* b)
{
    double t[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) { double s = 0; for (int k = 0; k < 3; k++) s += a[r * 3 + k] * b[k * 3 + c]; t[r * 3 + c] = s; }
    memcpy(d, t, sizeof(t));
}
bool same_matrix(const double* a, const double* b)   // SAME_MATRIX, EQUAL_ACCURACY = 1e-2
{
    for (int i = 0; i < 9; i++) if (fabs(a[i] - b[i]) > 1e-2) return false;
    return true;
}
void quat_of_matrix(double* q, const double* m)   // quaternion(dvec4&, const dmat33&), src/Geometry/Euler.cpp:112-123; m row-major
{
    auto M = [&](int r, int c) { return m[r * 3 + c]; };
    q[0] = 0.5 * sqrt(fmax(0.0, 1 + M(0, 0) + M(1, 1) + M(2, 2)));
    q[1] = 0.5 * sqrt(fmax(0.0, 1 + M(0, 0) - M(1, 1) - M(2, 2)));
    q[2] = 0.5 * sqrt(fmax(0.0, 1 - M(0, 0) + M(1, 1) - M(2, 2)));
    q[3] = 0.5 * sqrt(fmax(0.0, 1 - M(0, 0) - M(1, 1) + M(2, 2)));
    q[1] = copysign(q[1], M(2, 1) - M(1, 2));
    q[2] = copysign(q[2], M(0, 2) - M(2, 0));
    q[3] = copysign(q[3], M(1, 0) - M(0, 1));
}
void rot_of_quat(double* m /*row-major*/, const double* q)   // rotate3D(dmat33&, const dvec4&), src/Geometry/Euler.cpp:181-189
{
    const double A[3][3] = {{0, -q[3], q[2]}, {q[3], 0, -q[1]}, {-q[2], q[1], 0}};
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += A[r][k] * A[k][c];
            m[r * 3 + c] = (r == c ? 1.0 : 0.0) + 2 * q[0] * A[r][c] + 2 * s;
        }
}
PfArgs pf_args_ctx(const thx_pf_ctx* ctx)
{
    PfArgs a;
    memset(&a, 0, sizeof(a));
    if (ctx) { a.symQ = ctx->nSym > 0 ? ctx->symQuat : nullptr; a.nSym = ctx->nSym > 0 ? ctx->nSym : 0; a.img0 = ctx->img0; }
    return a;
}
}  // namespace

extern "C" {

// Symmetry::init(const char sym[]) (src/Geometry/Symmetry.cpp:61-278): symmetryGroup + fillSymmetryEntry
// (src/Geometry/SymmetryFunctions.cpp:13-164) -> fillLR (rotations; `RFLOAT angle = 2 * M_PI / fold`, axes as given) ->
// completePointGroup (closure under products, in the order of the reference's visit table).
int thx_symmetry_host(const char* sym, double* symMat, double* symQuat, int cap, int* nSym)
{
    THX_REQUIRE(sym && nSym && cap >= 0, "bad arguments");
    std::vector<SymEntry> e;
    auto rot = [&](int f, double x, double y, double z) { e.push_back({f, {x, y, z}}); };
    const size_t len = strlen(sym);
    bool digits = len > 1;
    for (size_t i = 1; i < len; i++) if (sym[i] < '0' || sym[i] > '9') digits = false;
    if ((sym[0] == 'C' || sym[0] == 'D') && digits) {
        THX_REQUIRE(atoi(sym + 1) >= 1, "point group order must be >= 1");
        rot(atoi(sym + 1), 0, 0, 1);
        if (sym[0] == 'D') rot(2, 1, 0, 0);
    } else if (!strcmp(sym, "T")) { rot(3, 0, 0, 1); rot(2, 0, 0.816496, 0.577350); }
    else if (!strcmp(sym, "O")) { rot(3, 0.5773502, 0.5773502, 0.5773502); rot(4, 0, 0, 1); }
    else if (!strcmp(sym, "I1")) { rot(2, 1, 0, 0); rot(5, 0.8506508, 0, -0.5257311); rot(3, 0.9341724, 0.3568221, 0); }
    else if (!strcmp(sym, "I2")) { rot(2, 0, 0, 1); rot(5, 0.5257311, 0, 0.8506508); rot(3, 0, 0.3568221, 0.9341724); }
    else if (!strcmp(sym, "I3")) { rot(2, -0.5257311, 0, 0.8506508); rot(5, 0, 0, 1); rot(3, -0.4911235, 0.3568221, 0.7946545); }
    else if (!strcmp(sym, "I4")) { rot(2, 0.5257311, 0, 0.8506508); rot(5, 0.8944272, 0, 0.4472136); rot(3, 0.4911235, 0.3568221, 0.7946545); }
    else { set_error("INVALID SYMMTRY INDEX: %s", sym); return -1; }
    std::vector<double> R;   // row-major [n][9]
    std::vector<double> Q;
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    auto novo = [&](const double* m) {
        if (same_matrix(m, I)) return false;
        for (size_t q = 0; q < R.size() / 9; q++) if (same_matrix(m, R.data() + 9 * q)) return false;
        return true;
    };
    auto append = [&](const double* m) { R.insert(R.end(), m, m + 9); double q[4]; quat_of_matrix(q, m); Q.insert(Q.end(), q, q + 4); };
    for (const SymEntry& en : e) {
        const float angle = (float)(2 * M_PI / en.fold);
        for (int j = 1; j < en.fold; j++) {
            const double phi = (double)(angle * (float)j);
            const double q[4] = {cos(phi / 2), sin(phi / 2) * en.ax[0], sin(phi / 2) * en.ax[1], sin(phi / 2) * en.ax[2]};
            double m[9];
            rot_of_quat(m, q);
            if (novo(m)) append(m);
        }
    }
    {
        std::vector<std::vector<unsigned char>> table(R.size() / 9, std::vector<unsigned char>(R.size() / 9, 0));
        for (;;) {
            int fi = -1, fj = -1;
            for (size_t r = 0; r < table.size() && fi < 0; r++)
                for (size_t c = 0; c < table.size(); c++)
                    if (!table[r][c]) { fi = (int)r; fj = (int)c; table[r][c] = 1; break; }
            if (fi < 0) break;
            double m[9];
            mat33_mul(m, R.data() + 9 * fi, R.data() + 9 * fj);
            if (novo(m)) {
                append(m);
                for (auto& row : table) row.push_back(0);
                table.push_back(std::vector<unsigned char>(table.size() + 1, 0));
            }
        }
    }
    const int n = (int)(R.size() / 9);
    *nSym = n;
    if (!symMat && !symQuat) return 0;
    THX_REQUIRE(n <= cap, "more symmetry elements than the caller's arrays hold");
    for (int s = 0; s < n; s++) {
        if (symMat) for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) symMat[9 * s + c * 3 + r] = R[9 * s + r * 3 + c];
        if (symQuat) for (int c = 0; c < 4; c++) symQuat[4 * s + c] = Q[4 * s + c];
    }
    return 0;
}

int thx_pf_symmetrise_dev(double* r, const double* anchor, int nImg, int nR, const double* symQuat, int nSym, void* stream)
{
    if (nImg <= 0 || nR <= 0 || nSym <= 0) return 0;
    THX_REQUIRE(r && symQuat, "NULL pointer");
    const size_t n = (size_t)nImg * nR;
    hipLaunchKernelGGL(k_pf_symmetrise, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), r, anchor, (size_t)nImg, nR,
                       symQuat, nSym);
    THX_LAUNCH_CHECK();
    return 0;
}

int thx_pf_cal_vari_dev(double* r, const double* t, double* k123, double* s01, int nImg, int nR, int nT, unsigned long long seed,
                        unsigned call, const thx_pf_ctx* ctx, void* stream)
{
    if (nImg <= 0) return 0;
    THX_REQUIRE(nR >= 0 && nT >= 0 && nR <= kPfMax && nT <= kPfMax, "nR / nT must be <= 256");
    THX_REQUIRE((nR == 0 || (r && k123 && nR >= 5)) && (nT == 0 || (t && s01 && nT >= 2)), "bad arguments");
    PfArgs a = pf_args_ctx(ctx);
    a.r = r; a.t = const_cast<double*>(t); a.k123 = k123; a.s01 = s01; a.nR = nR; a.nT = nT; a.seed = seed; a.call = call;
    hipLaunchKernelGGL(k_pf_cal_vari, dim3(nImg), dim3(64), 0, as_stream(stream), a);
    THX_LAUNCH_CHECK();
    return 0;
}

int thx_pf_perturb_ex_dev(double* r, double* t, double* wR, double* wT, const double* k123, const double* s01, int nImg,
                          int nR, int nT, double pfR, double pfT, double transS, double transQ, unsigned long long seed,
                          unsigned call, const int* active, const thx_pf_ctx* ctx, void* stream)
{
    if (nImg <= 0) return 0;
    THX_REQUIRE(nR >= 0 && nT >= 0 && nR <= kPfMax && nT <= kPfMax, "nR / nT must be <= 256");
    THX_REQUIRE((nR == 0 || (r && wR && k123)) && (nT == 0 || (t && wT && s01)), "NULL pointer");
    THX_REQUIRE(nR == 0 || nR >= 5, "the ACG statistics need at least 5 rotations");
    THX_REQUIRE(nT == 0 || nT >= 2, "the shift statistics need at least 2 points");
    PfArgs a = pf_args_ctx(ctx);
    a.r = r; a.t = t; a.wR = wR; a.wT = wT; a.k123 = const_cast<double*>(k123); a.s01 = const_cast<double*>(s01);
    a.nR = nR; a.nT = nT; a.pfR = pfR; a.pfT = pfT; a.transS = transS;
    // PARTICLE_RECENTRE_TRANSQ: transM = transS * gsl_cdf_chisq_Qinv(transQ, 2); for 2 degrees of freedom Q(x) = exp(-x/2)
    a.transM = transS * (-2.0 * log(transQ));
    a.seed = seed; a.call = call; a.active = active;
    hipLaunchKernelGGL(k_pf_perturb, dim3(nImg), dim3(64), 0, as_stream(stream), a);
    THX_LAUNCH_CHECK();
    return 0;
}

int thx_pf_perturb_dev(double* r, double* t, double* wR, double* wT, const double* k123, const double* s01, int nImg,
                       int nR, int nT, double pfR, double pfT, double transS, double transQ, unsigned long long seed,
                       unsigned call, const int* active, void* stream)
{
    return thx_pf_perturb_ex_dev(r, t, wR, wT, k123, s01, nImg, nR, nT, pfR, pfT, transS, transQ, seed, call, active, nullptr, stream);
}

int thx_pf_update_ex_dev(double* r, double* t, double* wR, double* wT, const float* uR, const float* uT, double* k123,
                         double* s01, double* topR, double* topT, int nImg, int nR, int nT, double peakFactorR,
                         unsigned long long seed, unsigned call, const int* active, const thx_pf_ctx* ctx, void* stream)
{
    if (nImg <= 0) return 0;
    THX_REQUIRE(nR >= 0 && nT >= 0 && nR <= kPfMax && nT <= kPfMax, "nR / nT must be <= 256");
    THX_REQUIRE((nR == 0 || (r && wR && uR && k123 && topR)) && (nT == 0 || (t && wT && uT && s01 && topT)), "NULL pointer");
    THX_REQUIRE((nR == 0 || nR >= 5) && (nT == 0 || nT >= 2), "too few support points");
    PfArgs a = pf_args_ctx(ctx);
    a.r = r; a.t = t; a.wR = wR; a.wT = wT; a.uR = uR; a.uT = uT; a.k123 = k123; a.s01 = s01; a.topR = topR; a.topT = topT;
    a.nR = nR; a.nT = nT; a.peakFactorR = peakFactorR; a.seed = seed; a.call = call; a.active = active;
    hipLaunchKernelGGL(k_pf_update, dim3(nImg), dim3(64), 0, as_stream(stream), a);
    THX_LAUNCH_CHECK();
    return 0;
}

int thx_pf_update_dev(double* r, double* t, double* wR, double* wT, const float* uR, const float* uT, double* k123,
                      double* s01, double* topR, double* topT, int nImg, int nR, int nT, double peakFactorR,
                      unsigned long long seed, unsigned call, const int* active, void* stream)
{
    return thx_pf_update_ex_dev(r, t, wR, wT, uR, uT, k123, s01, topR, topT, nImg, nR, nT, peakFactorR, seed, call, active, nullptr, stream);
}

int thx_pf_stop_init_dev(int* active, int* nP, double* state, double transS, double ctfRefineS, int nImg, void* stream)
{
    if (nImg <= 0) return 0;
    THX_REQUIRE(active && nP && state, "NULL pointer");
    hipLaunchKernelGGL(k_pf_stop_init, dim3((nImg + 255) / 256), dim3(256), 0, as_stream(stream), active, nP, state, transS, ctfRefineS,
                       nImg);
    THX_LAUNCH_CHECK();
    return 0;
}

int thx_pf_stop_rule_dev(int* active, int* nP, double* state, const double* k123, const double* s01, const double* sD, int phase,
                         int nImg, int* nActive, void* stream)
{
    if (nImg <= 0) return 0;
    THX_REQUIRE(active && nP && state && k123 && s01 && nActive, "NULL pointer");
    hipLaunchKernelGGL(k_pf_stop_rule, dim3((nImg + 255) / 256), dim3(256), 0, as_stream(stream), active, nP, state, k123, s01, sD,
                       phase, nImg, nActive);
    THX_LAUNCH_CHECK();
    return 0;
}

int thx_pf_perturb_d_ex_dev(double* d, double* wD, const double* sD, int nImg, int nD, double scale, int init, unsigned long long seed,
                            unsigned call, const int* active, unsigned img0, void* stream)
{
    if (nImg <= 0) return 0;
    THX_REQUIRE(d && wD && (init || sD) && nD >= 1 && nD <= 64, "bad arguments (at most 64 defocus support points)");
    hipLaunchKernelGGL(k_pf_perturb_d, dim3((nImg + 63) / 64), dim3(64), 0, as_stream(stream), d, wD, sD, nImg, nD, scale, init, seed, call,
                       active, img0);
    THX_LAUNCH_CHECK();
    return 0;
}

int thx_pf_perturb_d_dev(double* d, double* wD, const double* sD, int nImg, int nD, double scale, int init, unsigned long long seed,
                         unsigned call, const int* active, void* stream)
{
    return thx_pf_perturb_d_ex_dev(d, wD, sD, nImg, nD, scale, init, seed, call, active, 0u, stream);
}

int thx_pf_update_d_ex_dev(double* d, double* wD, const float* uD, double* sD, double* topD, int nImg, int nD, unsigned long long seed,
                           unsigned call, const int* active, unsigned img0, void* stream)
{
    if (nImg <= 0) return 0;
    THX_REQUIRE(d && wD && uD && sD && topD && nD >= 1 && nD <= 64, "bad arguments (at most 64 defocus support points)");
    hipLaunchKernelGGL(k_pf_update_d, dim3((nImg + 63) / 64), dim3(64), 0, as_stream(stream), d, wD, uD, sD, topD, nImg, nD, seed, call,
                       active, img0);
    THX_LAUNCH_CHECK();
    return 0;
}

int thx_pf_update_d_dev(double* d, double* wD, const float* uD, double* sD, double* topD, int nImg, int nD, unsigned long long seed,
                        unsigned call, const int* active, void* stream)
{
    return thx_pf_update_d_ex_dev(d, wD, uD, sD, topD, nImg, nD, seed, call, active, 0u, stream);
}

int thx_pf_class_select_ex_dev(int* cls, const float* uC, const double* wC, int nImg, int nK, double peakFactorC,
                               unsigned long long seed, unsigned call, unsigned img0, void* stream)
{
    if (nImg <= 0) return 0;
    THX_REQUIRE(cls && uC && nK >= 1 && nK <= kMaxClasses, "bad arguments (at most 64 classes)");
    hipLaunchKernelGGL(k_pf_class_select, dim3((nImg + 63) / 64), dim3(64), 0, as_stream(stream), cls, uC, wC, nImg, nK, peakFactorC,
                       seed, call, img0);
    THX_LAUNCH_CHECK();
    return 0;
}

int thx_pf_class_select_dev(int* cls, const float* uC, const double* wC, int nImg, int nK, double peakFactorC,
                            unsigned long long seed, unsigned call, void* stream)
{
    return thx_pf_class_select_ex_dev(cls, uC, wC, nImg, nK, peakFactorC, seed, call, 0u, stream);
}

int thx_pf_scan_support_dev(double* r, double* t, double* wR, double* wT, double* k123, double* s01, double* topR, double* topT,
                            const double* gridR, const double* gridT, const float* uR, const float* uT, const int* cls, int nImg,
                            int nRin, int nTin, int mLR, int mLT, double peakFactorR, double minK, double minS, unsigned long long seed,
                            unsigned call, void* stream)
{
    return thx_pf_scan_support_ex_dev(r, t, wR, wT, k123, s01, topR, topT, gridR, gridT, uR, uT, cls, nImg, nImg, nRin, nTin, mLR, mLT,
                                      peakFactorR, minK, minS, seed, call, nullptr, stream);
}

int thx_pf_scan_support_ex_dev(double* r, double* t, double* wR, double* wT, double* k123, double* s01, double* topR, double* topT,
                               const double* gridR, const double* gridT, const float* uR, const float* uT, const int* cls, int nImg,
                               int rowStride, int nRin, int nTin, int mLR, int mLT, double peakFactorR, double minK, double minS,
                               unsigned long long seed, unsigned call, const thx_pf_ctx* ctx, void* stream)
{
    if (nImg <= 0) return 0;
    THX_REQUIRE(r && t && wR && wT && k123 && s01 && topR && topT && gridR && gridT && uR && uT, "NULL pointer");
    THX_REQUIRE(mLR >= 5 && mLR <= kPfMax && mLT >= 2 && mLT <= kPfMax && mLR <= nRin && mLT <= nTin, "bad support sizes");
    THX_REQUIRE(nRin <= 16384 && nTin <= 16384, "at most 16384 scanned points per parameter (the shuffle sorts them in LDS)");
    ScanSupportArgs a;
    a.r = r; a.t = t; a.wR = wR; a.wT = wT; a.k123 = k123; a.s01 = s01; a.topR = topR; a.topT = topT;
    a.gridR = gridR; a.gridT = gridT; a.uR = uR; a.uT = uT; a.cls = cls;
    a.nImg = nImg; a.nRin = nRin; a.nTin = nTin; a.mLR = mLR; a.mLT = mLT;
    a.NPR = 64; while (a.NPR < nRin) a.NPR <<= 1;
    a.NPT = 64; while (a.NPT < nTin) a.NPT <<= 1;
    a.peakFactorR = peakFactorR; a.minK = minK; a.minS = minS; a.seed = seed; a.call = call;
    a.symQ = ctx && ctx->nSym > 0 ? ctx->symQuat : nullptr; a.nSym = ctx && ctx->nSym > 0 ? ctx->nSym : 0; a.img0 = ctx ? ctx->img0 : 0u;
    a.rowStride = rowStride;
    const size_t lds = (size_t)(a.NPR > a.NPT ? a.NPR : a.NPT) * sizeof(unsigned long long);
    THX_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pf_scan_support), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_pf_scan_support, dim3(nImg), dim3(kScanSupThreads), lds, as_stream(stream), a);
    THX_LAUNCH_CHECK();
    return 0;
}

int thx_pf_acg_stats_dev(double* A, double* mean, double* k123, double* wBal, int* rounds, const double* quat, int nImg,
                         int n, void* stream)
{
    if (nImg <= 0) return 0;
    THX_REQUIRE(A && mean && k123 && wBal && quat && n >= 5 && n <= kPfMax, "bad arguments");
    hipLaunchKernelGGL(k_pf_acg_stats, dim3(nImg), dim3(64), 0, as_stream(stream), A, mean, k123, wBal, rounds, quat, n);
    THX_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
