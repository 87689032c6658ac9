// thx_estep.hip -- E-step kernels: rotation matrices, phase ramps, CTF rows, pixel gather,
// Fourier-slice extraction, likelihood and the particle-filter weight marginals.
// Reference behaviour: src/Optimiser.cpp:622-1681 (expectation), src/Projector.cpp:356-374,
// src/CTF.cpp:113-151, src/Image/ImageFunctions.cpp:233-252.  gfx950 only; wavefront = 64.
#include <stdarg.h>
#include <stdlib.h>

#include <map>
#include <mutex>
#include <tuple>
#include <utility>

#include "thx_common.h"

namespace thx {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static Knobs read_knobs()
{
    Knobs v;
    const char* e;
    v.expectNSplit = 0;
    if ((e = getenv("THX_EXPECT_NSPLIT"))) { const int n = atoi(e); if (n >= 1 && n <= 16) v.expectNSplit = n; }
    v.expectWgPerCU = -1;
    if ((e = getenv("THX_EXPECT_WG_PER_CU"))) v.expectWgPerCU = atoi(e);
    e = getenv("THX_EXPECT_ND");
    v.expectNdSweep = e && e[0] == 's';
    v.expectSplit = 0.f;
    if ((e = getenv("THX_EXPECT_SPLIT"))) v.expectSplit = (float)atof(e);
    v.expectOrder = 1;
    if ((e = getenv("THX_EXPECT_ORDER"))) v.expectOrder = atoi(e);
    v.expectWgLater = -1;
    if ((e = getenv("THX_EXPECT_WG_LATER"))) v.expectWgLater = atoi(e);
    e = getenv("THX_SCAN");
    v.scanSimple = e && e[0] == 's';
    v.scanTile = e && e[0] == 't' ? atoi(e + 1) : 0;
    e = getenv("THX_INSERT_PLAIN");
    v.insertPlain = e && e[0] == '1';
    e = getenv("THX_INSERT_SCRATCH_MB");
    v.insertScratchMB = e ? atol(e) : 0;
    e = getenv("THX_INSERT_SEG_CAP");
    v.insertSegCap = e ? atol(e) : 0;
    e = getenv("THX_FFT");
    v.fftRocfft = e && e[0] == 'r';
    v.recoTrace = getenv("THX_RECO_TRACE") != nullptr;
    { const char* w = getenv("THX_FFTZ_WAVES"); v.fftzWaves = w ? atoi(w) : 0; }
    { const char* w = getenv("THX_RECO_WT"); v.recoNatural = w && w[0] == 'n'; }
    { const char* w = getenv("THX_RECO_STOP"); v.recoHostStop = w && w[0] == 'h'; }
    { const char* w = getenv("THX_RECO_OWNERS"); v.recoReplicate = w && w[0] == '0'; }
    e = getenv("THX_COMM_FORCE");
    v.commForce = e && e[0] == '1';
    return v;
}

static Knobs g_knobs = read_knobs();   // once, when the library is loaded
const Knobs& knobs() { return g_knobs; }

// keyed on (device, stream, slot): the Interface.h-shaped *_host entry points run on whichever device gpuIdx names
struct ScratchBuf { void* p = nullptr; size_t n = 0; };
static std::mutex g_scratchMtx;
static std::map<std::tuple<int, hipStream_t, int>, ScratchBuf> g_scratch;
static std::map<std::tuple<int, hipStream_t, int>, ScratchBuf> g_pinned;
void release_io_plans(int dev, hipStream_t st);
void release_next_plans(int dev, hipStream_t st);
extern "C" void thx_release_reco_plans_(int dev, hipStream_t st);
// everything the library caches per (device, stream): scratch buffers and hipFFT plans (which carry their stream).  A caller
// that destroys a stream calls this first -- a recycled handle value must not find a plan bound to the destroyed stream.
int release_stream(hipStream_t stream)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    (void)hipStreamSynchronize(stream);
    {
        std::lock_guard<std::mutex> g(g_scratchMtx);
        for (auto it = g_scratch.begin(); it != g_scratch.end();)
            if (std::get<0>(it->first) == dev && std::get<1>(it->first) == stream) { if (it->second.p) (void)hipFree(it->second.p); it = g_scratch.erase(it); } else ++it;
        for (auto it = g_pinned.begin(); it != g_pinned.end();)
            if (std::get<0>(it->first) == dev && std::get<1>(it->first) == stream) { if (it->second.p) (void)hipHostFree(it->second.p); it = g_pinned.erase(it); } else ++it;
    }
    release_io_plans(dev, stream);
    release_next_plans(dev, stream);
    thx_release_reco_plans_(dev, stream);
    return 0;
}
// page-locked host memory for the few words the host reads back inside a call (grow-only, per (device, stream, slot) like
// scratch(): a read-back into pageable memory makes hipMemcpyAsync stage and block)
void* pinned_host(hipStream_t stream, int slot, size_t bytes)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> g(g_scratchMtx);
    ScratchBuf& b = g_pinned[std::make_tuple(dev, stream, slot)];
    if (b.n < bytes) {
        if (b.p) {
            (void)hipStreamSynchronize(stream);
            (void)hipHostFree(b.p);
        }
        b.p = nullptr;
        b.n = 0;
        const size_t want = bytes < 4096 ? 4096 : bytes;
        if (hipHostMalloc(&b.p, want) != hipSuccess) { b.p = nullptr; return nullptr; }
        b.n = want;
    }
    return b.p;
}
void scratch_release(hipStream_t stream, int slot)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> g(g_scratchMtx);
    auto it = g_scratch.find(std::make_tuple(dev, stream, slot));
    if (it == g_scratch.end()) return;
    (void)hipStreamSynchronize(stream);
    if (it->second.p) (void)hipFree(it->second.p);
    g_scratch.erase(it);
}
size_t scratch_size(hipStream_t stream, int slot)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> g(g_scratchMtx);
    auto it = g_scratch.find(std::make_tuple(dev, stream, slot));
    return it == g_scratch.end() ? 0 : it->second.n;
}
void* scratch(hipStream_t stream, int slot, size_t bytes)
{
    using Buf = ScratchBuf;
    std::mutex& mtx = g_scratchMtx;
    auto& bufs = g_scratch;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> g(mtx);
    Buf& b = bufs[std::make_tuple(dev, stream, slot)];
    if (b.n < bytes) {
        if (b.p) {
            (void)hipStreamSynchronize(stream);   // the old buffer may still be in use by queued work
            (void)hipFree(b.p);
        }
        b.p = nullptr;
        b.n = 0;
        const size_t want = bytes < 4096 ? 4096 : bytes;
        if (hipMalloc(&b.p, want) != hipSuccess) { b.p = nullptr; return nullptr; }
        b.n = want;
    }
    return b.p;
}

// ---------------------------------------------------------------------------------------------
// rotate3D(quaternion), src/Geometry/Euler.cpp:181-189: R = I + 2 q0 A + 2 A*A, column-major out
// ---------------------------------------------------------------------------------------------
__global__ void k_rotmat(const double* __restrict__ quat, double* __restrict__ mat, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    rotate3d_colmajor(quat + 4 * (size_t)i, mat + 9 * (size_t)i);
}

// translate(), src/Image/ImageFunctions.cpp:233-252, nT ramps at once: grid (ceil(nPxl/256), nT)
__global__ void k_translate(float2* __restrict__ traP, const double* __restrict__ trans, const int* __restrict__ iCol,
                            const int* __restrict__ iRow, int nPxl, int idim)
{
    const int t = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nPxl) return;
    const float rCol = (float)trans[2 * t] / idim, rRow = (float)trans[2 * t + 1] / idim;
    traP[(size_t)t * nPxl + p] = ramp_value(rCol, rRow, iCol[p], iRow[p]);
}

// CTF rows, src/CTF.cpp:113-151: grid (ceil(nPxl/256), nImg)
__global__ void k_ctf(float* __restrict__ ctfP, const thx_ctf_attr* __restrict__ attr, const double* __restrict__ dfac,
                      float pixelSize, const int* __restrict__ iCol, const int* __restrict__ iRow, int nPxl, int idim)
{
    const int l = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nPxl) return;
    const CtfConst c = ctf_const(attr[l], dfac ? dfac[l] : 1.0);
    ctfP[(size_t)l * nPxl + p] = ctf_value(c, pixelSize, idim, idim, iCol[p], iRow[p]);
}

// allocPreCal, ctf = true branch, src/Optimiser.cpp:8124-8169 (ExpectPrecal / kernel_ExpectPrectf): grid (ceil(nPxl/256), nImg)
__global__ void k_expect_precal(float* __restrict__ freq, float* __restrict__ def, float* __restrict__ k1,
                                float* __restrict__ k2, const thx_ctf_attr* __restrict__ attr, int size,
                                float pixelSize, const int* __restrict__ iCol, const int* __restrict__ iRow, int nPxl)
{
    const int l = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const thx_ctf_attr a = attr[l];
    if (p == 0) {
        const float lambda = (float)(12.2643274 / sqrt(a.voltage * (1 + a.voltage * 0.978466e-6)));  // sic, :8164
        k1[l] = (float)(3.14159265358979323846 * lambda);
        k2[l] = (float)(1.57079632679489661923 * a.Cs * pow3f_(lambda));
    }
    if (p >= nPxl) return;
    if (l == 0 && freq) freq[p] = (float)(gsl_hypot_((double)iCol[p], (double)iRow[p]) / size / pixelSize);
    const float angle = (float)(atan2((double)iRow[p], (double)iCol[p]) - a.defocusTheta);
    def[(size_t)l * nPxl + p] = -(a.defocusU + a.defocusV + (a.defocusU - a.defocusV) * cosf(2 * angle)) / 2;
}

// defocus-search CTF rows, src/Optimiser.cpp:1246-1272: grid (ceil(nPxl/256), nD, nImg); ctfP [nImg][nD][nPxl]
__global__ void k_ctf_dsearch(float* __restrict__ ctfP, const float* __restrict__ freq, const float* __restrict__ def,
                              const float* __restrict__ k1, const float* __restrict__ k2,
                              const thx_ctf_attr* __restrict__ attr, const double* __restrict__ dpara, int nD, int nPxl)
{
    const int l = blockIdx.z, iD = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nPxl) return;
    const float f = freq[p], ac = attr[l].amplitudeContrast;
    const float ki = (float)(k1[l] * def[(size_t)l * nPxl + p] * dpara[(size_t)l * nD + iD] * pow2f_(f) +
                             k2[l] * pow4f_(f) - attr[l].phaseShift);
    ctfP[((size_t)l * nD + iD) * nPxl + p] = -sqrtf(1 - pow2f_(ac)) * sinf(ki) + ac * cosf(ki);
}

// CTF(Image& dst, ...), src/CTF.cpp:31-66 (GCTFinit): grid (idim, nImg), row per block; complex output, imaginary part 0
__global__ __launch_bounds__(128) void k_ctf_image(float2* __restrict__ dst, const thx_ctf_attr* __restrict__ attr,
                                                   float pixelSize, int idim)
{
    const int row = blockIdx.x, l = blockIdx.y, nc = idim / 2 + 1;
    const int j = row < idim / 2 ? row : row - idim;
    const CtfConst c = ctf_const(attr[l], 1.0);
    for (int i = threadIdx.x; i < nc; i += blockDim.x)
        dst[((size_t)l * idim + row) * nc + i] = make_float2(ctf_value(c, pixelSize, idim, idim, i, j), 0.f);
}

// allocPreCal gather, src/Optimiser.cpp:8055-8075
__global__ void k_gather_pixels(float2* __restrict__ datP, const float2* __restrict__ img, const int* __restrict__ iPxl,
                                int nPxl, size_t imgSize)
{
    const int l = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nPxl) return;
    datP[(size_t)l * nPxl + p] = img[(size_t)l * imgSize + iPxl[p]];
}

// ---------------------------------------------------------------------------------------------
// Projector::project, src/Projector.cpp:356-374: one thread per (pixel, rotation).
// grid (ceil(nPxl/256), nR).  64 B gathered + 8 B written per thread: HBM/L2-bound gather.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_project(const float2* __restrict__ vol, float2* __restrict__ out,
                                                 const double* __restrict__ rotMat, const int* __restrict__ iCol,
                                                 const int* __restrict__ iRow, int pf, int P, int nPxl)
{
    const int r = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nPxl) return;
    const double* m = rotMat + 9 * (size_t)r;
    const double nx = (double)(iCol[p] * pf), ny = (double)(iRow[p] * pf);
    // dvec3 oldCor = mat * dvec3(nx, ny, 0): the z column contributes +-0
    const double ox = m[0] * nx + m[3] * ny;
    const double oy = m[1] * nx + m[4] * ny;
    const double oz = m[2] * nx + m[5] * ny;
    const float x = (float)ox, y = (float)oy, z = (float)oz;
    float2 v = make_float2(0.f, 0.f);
    if (coord_in_grid(x, y, z, P)) v = interp_ft(vol, P, x, y, z);
    out[(size_t)r * nPxl + p] = v;
}

// ---------------------------------------------------------------------------------------------
// logDataVSPrior for nPri prior rows against one image row: one wave per row, lanes stride pixels.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_logdvp(float* __restrict__ out, const float2* __restrict__ dat,
                                                const float2* __restrict__ pri, const float* __restrict__ ctf,
                                                const float* __restrict__ sigRcp, int nPri, int nPxl)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= nPri) return;
    const float2* pr = pri + (size_t)row * nPxl;
    float acc = 0.f;
    for (int i = lane; i < nPxl; i += 64) {
        const float c = ctf[i];
        const float2 d = dat[i], q = pr[i];
        const float a = d.x - c * q.x, b = d.y - c * q.y;
        acc += (a * a + b * b) * sigRcp[i];
    }
    acc = wave_sum(acc);
    if (lane == 0) out[row] = acc;
}

// ---------------------------------------------------------------------------------------------
// Local particle-filter phase (HOT LOOP B body, src/Optimiser.cpp:1225-1406), batched over images.
//
// For one image:  L[r][t] = sum_pix | dat - ctf * tra_t * pri_r |^2 * sigRcp
//                         = sum_pix s|dat|^2  +  sum_pix B |pri_r|^2  -  2 sum_pix Re(A_t pri_r)
//   with s = sigRcp, B = s ctf^2, A_t = s ctf conj(dat) tra_t  (|tra_t| = 1).
// The constant first term is summed once; the per-(r,t) part V[r][t] costs 2 FMAs per pixel-sample
// instead of the reference's 15 flops, and is accumulated without the large constant, so exp(V - Vmax)
// carries less rounding noise than the reference's float sum does.
//
// Mapping: lane <-> rotation (no cross-lane reduction in the hot loop); each wave walks a
// sub-stream of the chunk's pixels; the per-pixel A_t / B table is staged in LDS once per chunk and
// read back as wave-uniform broadcasts.  grid (nSplit, nImg, nD), block 256.
// ---------------------------------------------------------------------------------------------
// cell-packed copy of a projector volume (see interp_ft_packed): one thread per (cell, row pair)
__global__ __launch_bounds__(256) void k_pack_cells(float4* __restrict__ cells, const float2* __restrict__ vol, int P)
{
    const long nc = P / 2 + 1;
    const size_t n = (size_t)P * P * nc * 4;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int kj = (int)(e & 3);
    const size_t cell = e >> 2;
    const int x0 = (int)(cell % nc);
    const size_t row = cell / nc;
    const int yw = (int)(row % P), zw = (int)(row / P);
    const int y1 = (yw + (kj & 1)) % P, z1 = (zw + (kj >> 1)) % P;
    const float2* r = vol + ((size_t)z1 * P + y1) * nc;
    const float2 a = r[x0];
    const float2 b = x0 + 1 < nc ? r[x0 + 1] : make_float2(0.f, 0.f);
    cells[e] = make_float4(a.x, a.y, b.x, b.y);
}

// Ordering of an image's cloud of rotations for the lane <-> rotation mapping of k_expect_local (THX_EXPECT_ORDER; default 1).
// A listed pixel p = (x, y, 0) under R_i = R_0 d_i, d_i ~ 1 + [w_i]x, lands at R_0 (p + w_i x p): in the slice's own frame the
// cloud of one pixel is spread by w_z (-y, x) IN the plane and by (w_x y - w_y x) off it.  Ranking the rotations by one of
// the three components of w puts each of the kernel's 64-rotation waves on one half of the cloud along that direction.
// key: 1 = w_z (in-plane angle), 2 = w_x, 3 = w_y, all relative to the image's first rotation.  One workgroup of 256 threads
// per image; order[img][rank] = rotation.
__global__ __launch_bounds__(256) void k_cloud_order(unsigned char* __restrict__ order, const double* __restrict__ rotMat, int nR,
                                                     int key)
{
    __shared__ unsigned sk[256];
    const int img = blockIdx.x, i = threadIdx.x;
    const double* m0 = rotMat + (size_t)img * nR * 9;
    if (i < nR) {
        const double* m = m0 + (size_t)i * 9;
        // d = R_0^T R_i, d[a][b] = col_a(R_0) . col_b(R_i); w = (d[2][1] - d[1][2], d[0][2] - d[2][0], d[1][0] - d[0][1]) / 2
        auto dot = [&](int a, int b) { return m0[3 * a] * m[3 * b] + m0[3 * a + 1] * m[3 * b + 1] + m0[3 * a + 2] * m[3 * b + 2]; };
        double w;
        if (key == 2) w = dot(2, 1) - dot(1, 2);
        else if (key == 3) w = dot(0, 2) - dot(2, 0);
        else w = dot(1, 0) - dot(0, 1);
        // a TOTAL order whatever the key holds: the float's bits mapped monotonically to an unsigned (negative values reversed), so a
        // NaN key (a degenerate rotation matrix) ranks like any other value instead of comparing false against everything --
        // every rotation gets its own rank and every order[] slot is written (round-5 advisor)
        const unsigned b = __float_as_uint((float)w);
        sk[i] = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    }
    __syncthreads();
    if (i < nR) {
        const unsigned k = sk[i];
        int rank = 0;
        for (int j = 0; j < nR; j++) rank += (sk[j] < k || (sk[j] == k && j < i)) ? 1 : 0;
        order[(size_t)img * nR + rank] = (unsigned char)i;
    }
}

constexpr int kChunk = 256;  // pixels staged per LDS table

struct ExpectLocalArgs {
    const float2* volumes;
    const int* volIdx;
    int P, pf, idim;
    const int* iCol;
    const int* iRow;
    int nPxl, nImg;
    const float2* datP;
    const float* ctfP;
    const float* sigRcpP;
    const double* rotMat;
    int nR;
    const double* trans;
    int nT, nD;
    int nSplit;
    float* partV;  // [nImg][nD][nSplit][nT][nRpad]
    float* partC;  // [nImg][nD][nSplit]
    int nRpad;
    const int* active;   // [nImg] or NULL: images with active[img] == 0 are skipped (outputs untouched)
    const unsigned char* order;   // [nImg][nR] or NULL: lane slot s of an image works on rotation order[s] (k_cloud_order)
    float splitM;        // SPLIT form: half-thickness (voxels) of the slab around the wave's mean slice whose samples are fetched ahead
    int fine;            // != 0: the nSplit workgroups of an image share its pixels in nSplit near-equal PIXEL ranges (the one-image form: a
                         // range may be a fraction of a 256-pixel chunk); 0: in whole chunks (the batched form)
};

template <int NT, bool PACKED, bool SPLIT = false>
__global__ __launch_bounds__(256) void k_expect_local(ExpectLocalArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // per pixel the NT values A_t = s ctf conj(dat) ramp_t, laid out for two-lane FMAs: shifts in pairs (re_t, re_t+1, -im_t,
    // -im_t+1), an odd last one as (re, -im) -- so that acc_t += re_t q.x; acc_t += -im_t q.y is one v_pk_fma_f32 per product
    // for two shifts, with no register shuffling and no sign flips in the loop
    float* sA = reinterpret_cast<float*>(smem_raw);                   // [kChunk][2 NT]
    double* sIc = reinterpret_cast<double*>(sA + kChunk * 2 * NT);    // [kChunk] padded pixel coordinates, as the doubles the
    double* sIr = sIc + kChunk;                                       // [kChunk] rotation multiplies
    float* sB = reinterpret_cast<float*>(sIr + kChunk);               // [kChunk]
    float* sRed = sB + kChunk;                                        // [4 waves] block-reduce scratch

    const int split = blockIdx.x, img = blockIdx.y, d = blockIdx.z;
    if (a.active && !a.active[img]) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = a.P;
    const float2* vol = a.volumes + (size_t)(a.volIdx ? a.volIdx[img] : 0) * ((size_t)P * P * (P / 2 + 1)) * (PACKED ? 8 : 1);
    const float2* dat = a.datP + (size_t)img * a.nPxl;
    const float* ctf = a.ctfP + ((size_t)img * a.nD + d) * a.nPxl;
    const float* sig = a.sigRcpP + (size_t)img * a.nPxl;
    const double* tr = a.trans + (size_t)img * a.nT * 2;

    // rotation groups: nRG waves cover 64*nRG rotation slots per pass, the other waves are pixel sub-streams
    const int nRG0 = (a.nR + 63) >> 6;
    const int nRGp = nRG0 >= 4 ? 4 : (nRG0 >= 2 ? 2 : 1);
    const int nSub = 4 / nRGp;
    const int rg = wave % nRGp, sub = wave / nRGp;
    const int nPass = (nRG0 + nRGp - 1) / nRGp;

    // this block's pixel range (whole chunks)
    const int nChunks = (a.nPxl + kChunk - 1) / kChunk;
    const int c0 = (int)(((long)nChunks * split) / a.nSplit), c1 = (int)(((long)nChunks * (split + 1)) / a.nSplit);
    // [p0, p1): whole chunks (the batched form), or -- fine -- the split's share of the pixels themselves
    const int p0 = a.fine ? (int)(((long)a.nPxl * split) / a.nSplit) : c0 * kChunk;
    const int p1 = a.fine ? (int)(((long)a.nPxl * (split + 1)) / a.nSplit) : min(c1 * kChunk, a.nPxl);

    float cpart = 0.f;

    for (int pass = 0; pass < nPass; pass++) {
        const int rslot = (pass * nRGp + rg) * 64 + lane;
        const bool rvalid = rslot < a.nR;
        // which rotation this lane carries: its slot, or -- with an ordering of the cloud -- the rotation ranked there (the
        // results go back to the rotation's own place: bit-identical outputs whatever the order)
        const int r = (rvalid && a.order) ? (int)a.order[(size_t)img * a.nR + rslot] : rslot;
        double m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0, m5 = 0;
        if (rvalid) {
            const double* m = a.rotMat + ((size_t)img * a.nR + r) * 9;
            m0 = m[0]; m1 = m[1]; m2 = m[2]; m3 = m[3]; m4 = m[4]; m5 = m[5];
        }
        thx_v2f accP[(NT + 1) / 2];   // (acc_t, acc_t+1); the odd last shift in lane 0 of the last pair
        float accB = 0.f;
#pragma unroll
        for (int i = 0; i < (NT + 1) / 2; i++) accP[i] = thx_v2f{0.f, 0.f};

        for (int pbase = p0; pbase < p1; pbase += kChunk) {
            const int clen = min(kChunk, p1 - pbase);
            __syncthreads();  // previous chunk's readers are done
            // ---- stage the per-pixel table ----
            for (int e = tid; e < clen; e += 256) {
                const int p = pbase + e;
                const int ic = a.iCol[p], ir = a.iRow[p];
                sIc[e] = (double)(ic * a.pf);
                sIr[e] = (double)(ir * a.pf);
                const float s = sig[p], cf = ctf[p];
                const float2 dv = dat[p];
                const float g = s * cf;
                sB[e] = g * cf;
                if (pass == 0) cpart = fmaf(s, fmaf(dv.x, dv.x, dv.y * dv.y), cpart);
                const float2 cd = make_float2(dv.x * g, -dv.y * g);  // s ctf conj(dat)
#pragma unroll
                for (int t = 0; t < NT; t++) {
                    if (t < a.nT) {
                        const float rCol = (float)tr[2 * t] / a.idim, rRow = (float)tr[2 * t + 1] / a.idim;
                        const float2 A = cmul(cd, ramp_value(rCol, rRow, ic, ir));
                        float* dst = sA + e * 2 * NT + ((NT & 1) && t == NT - 1 ? 2 * t : 4 * (t >> 1) + (t & 1));
                        dst[0] = A.x;
                        dst[(NT & 1) && t == NT - 1 ? 1 : 2] = -A.y;
                    }
                }
            }
            __syncthreads();
            // ---- hot loop: lane = rotation, walk this wave's pixel sub-stream ----
            if (SPLIT && PACKED) {
                // near-slab / tail form: a sample within splitM voxels of the slab around the wave's MEAN slice (the cells the other
                // rotations of the cloud are touching too: L2 hits once one lane has fetched them) is requested one pixel AHEAD of its
                // use; a sample of the tail keeps the one-at-a-time path -- it is requested first, so that waiting for it leaves the
                // look-ahead requests in flight.  Same cells, same arithmetic, same order of accumulation: bit-identical.
                const float4* cells = reinterpret_cast<const float4*>(vol);
                float nbx = 0.f, nby = 0.f, nbz = 0.f;
                {   // the wave's mean slice normal: sum over its rotations of col0 x col1 (R e_z)
                    float nx = rvalid ? (float)(m1 * m5 - m2 * m4) : 0.f, ny = rvalid ? (float)(m2 * m3 - m0 * m5) : 0.f,
                          nz = rvalid ? (float)(m0 * m4 - m1 * m3) : 0.f;
                    nx = wave_sum(nx); ny = wave_sum(ny); nz = wave_sum(nz);
                    const float nn = sqrtf(nx * nx + ny * ny + nz * nz);
                    if (nn > 0.f) { nbx = nx / nn; nby = ny / nn; nbz = nz / nn; }
                }
                // every iteration issues exactly four "now" loads and four "ahead" loads, unconditionally -- a lane that has nothing to
                // ask for reads the volume's first cell (one broadcast line per instruction) -- so that the compiler can count them:
                // waiting for the tail's data is s_waitcnt vmcnt(4), which leaves the look-ahead in flight
                const float4* dummy = cells;
                float4 pre[4];
                float px = 0.f, py = 0.f, pz = 0.f;
                bool pin = false, pnear = false;
                auto locate = [&](int e, float& x, float& y, float& z, bool& in, bool& near) {
                    const double nx = sIc[e], ny = sIr[e];
                    x = (float)(m0 * nx + m3 * ny);
                    y = (float)(m1 * nx + m4 * ny);
                    z = (float)(m2 * nx + m5 * ny);
                    in = rvalid && coord_in_grid(x, y, z, P);
                    near = in && fabsf(nbx * x + nby * y + nbz * z) <= a.splitM;
                };
                {
                    if (sub < clen) locate(sub, px, py, pz, pin, pnear);
                    const float4* c = pnear ? packed_cell(cells, P, px, py, pz) : dummy;
                    pre[0] = c[0]; pre[1] = c[1]; pre[2] = c[2]; pre[3] = c[3];
                }
                for (int e = sub; e < clen; e += nSub) {
                    const float x = px, y = py, z = pz;
                    const bool in = pin, near = pnear;
                    float4 cur[4] = {pre[0], pre[1], pre[2], pre[3]};
                    float4 far[4];
                    {   // tail: requested now, BEFORE the look-ahead of the next pixel
                        const float4* c = (in && !near) ? packed_cell(cells, P, x, y, z) : dummy;
                        far[0] = c[0]; far[1] = c[1]; far[2] = c[2]; far[3] = c[3];
                    }
                    {
                        pin = false; pnear = false;
                        if (e + nSub < clen) locate(e + nSub, px, py, pz, pin, pnear);
                        const float4* c = pnear ? packed_cell(cells, P, px, py, pz) : dummy;
                        pre[0] = c[0]; pre[1] = c[1]; pre[2] = c[2]; pre[3] = c[3];
                    }
                    if (!near) { cur[0] = far[0]; cur[1] = far[1]; cur[2] = far[2]; cur[3] = far[3]; }
                    float2 q = make_float2(0.f, 0.f);
                    if (in) q = packed_combine(cur, x, y, z);
                    if (rvalid) {
                        accB = fmaf(sB[e], fmaf(q.x, q.x, q.y * q.y), accB);
                        const float* Ap = sA + e * 2 * NT;
                        const thx_v2f qx = {q.x, q.x}, qy = {q.y, q.y};
#pragma unroll
                        for (int i = 0; i < NT / 2; i++) {
                            const float2 re = *reinterpret_cast<const float2*>(Ap + 4 * i), ni = *reinterpret_cast<const float2*>(Ap + 4 * i + 2);
                            accP[i] = __builtin_elementwise_fma(thx_v2f{re.x, re.y}, qx, accP[i]);
                            accP[i] = __builtin_elementwise_fma(thx_v2f{ni.x, ni.y}, qy, accP[i]);
                        }
                        if (NT & 1) {
                            const float2 A = *reinterpret_cast<const float2*>(Ap + 2 * (NT - 1));
                            accP[NT / 2].x = fmaf(A.x, q.x, accP[NT / 2].x);
                            accP[NT / 2].x = fmaf(A.y, q.y, accP[NT / 2].x);
                        }
                    }
                }
            } else if (rvalid) {
                for (int e = sub; e < clen; e += nSub) {
                    const double nx = sIc[e], ny = sIr[e];
                    const float x = (float)(m0 * nx + m3 * ny);
                    const float y = (float)(m1 * nx + m4 * ny);
                    const float z = (float)(m2 * nx + m5 * ny);
                    float2 q = make_float2(0.f, 0.f);
                    if (coord_in_grid(x, y, z, P))
                        q = PACKED ? interp_ft_packed(reinterpret_cast<const float4*>(vol), P, x, y, z) : interp_ft(vol, P, x, y, z);
                    accB = fmaf(sB[e], fmaf(q.x, q.x, q.y * q.y), accB);
                    const float* Ap = sA + e * 2 * NT;
                    const thx_v2f qx = {q.x, q.x}, qy = {q.y, q.y};
#pragma unroll
                    for (int i = 0; i < NT / 2; i++) {
                        const float2 re = *reinterpret_cast<const float2*>(Ap + 4 * i), ni = *reinterpret_cast<const float2*>(Ap + 4 * i + 2);
                        accP[i] = __builtin_elementwise_fma(thx_v2f{re.x, re.y}, qx, accP[i]);
                        accP[i] = __builtin_elementwise_fma(thx_v2f{ni.x, ni.y}, qy, accP[i]);
                    }
                    if (NT & 1) {
                        const float2 A = *reinterpret_cast<const float2*>(Ap + 2 * (NT - 1));
                        accP[NT / 2].x = fmaf(A.x, q.x, accP[NT / 2].x);
                        accP[NT / 2].x = fmaf(A.y, q.y, accP[NT / 2].x);
                    }
                }
            }
        }
        // ---- combine pixel sub-streams (fixed order) and write V for this split ----
        __syncthreads();
        float* sAcc = reinterpret_cast<float*>(smem_raw);  // reuse: [nSub][NT+1][64*nRGp]
        const int slots = 64 * nRGp;
        const int slot = rg * 64 + lane;
#pragma unroll
        for (int t = 0; t < NT; t++) sAcc[(sub * (NT + 1) + t) * slots + slot] = (t & 1) ? accP[t >> 1].y : accP[t >> 1].x;
        sAcc[(sub * (NT + 1) + NT) * slots + slot] = accB;
        __syncthreads();
        if (sub == 0 && rvalid) {
            float b = 0.f;
            for (int s2 = 0; s2 < nSub; s2++) b += sAcc[(s2 * (NT + 1) + NT) * slots + slot];
            float* outV = a.partV + ((((size_t)img * a.nD + d) * a.nSplit + split) * a.nT) * a.nRpad;
#pragma unroll
            for (int t = 0; t < NT; t++) {
                if (t < a.nT) {
                    float v = 0.f;
                    for (int s2 = 0; s2 < nSub; s2++) v += sAcc[(s2 * (NT + 1) + t) * slots + slot];
                    outV[(size_t)t * a.nRpad + r] = b - 2.0f * v;
                }
            }
        }
        __syncthreads();
    }
    // ---- constant term: block reduce ----
    cpart = wave_sum(cpart);
    if (lane == 0) sRed[wave] = cpart;
    __syncthreads();
    if (tid == 0) a.partC[((size_t)img * a.nD + d) * a.nSplit + split] = (sRed[0] + sRed[1]) + (sRed[2] + sRed[3]);
}

// CTF search (nD > 1, src/Optimiser.cpp:1246-1287): the same phase with the defocus factors fused.  The gather of a
// sample does not depend on the defocus factor, so one pass serves all of them:
//   L[r][t][d] - C = sum_p  c_d(p)^2 s(p) |q|^2  - 2 c_d(p) Re(A'_t(p) q),   A'_t = s conj(dat) ramp_t,  q = slice_r(p)
// per sample: u_t = Re(A'_t q) (NT terms), then ND x (NT + 1) FMAs with the ND CTF values of the pixel (LDS broadcasts).
// 9 x fewer gathers than one sweep per defocus factor at nD = 9.  grid (nSplit, nImg), block 256; nT <= NT, nD <= ND.
template <int NT, int ND, bool PACKED>
__global__ __launch_bounds__(256) void k_expect_local_nd(ExpectLocalArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float2* sA = reinterpret_cast<float2*>(smem_raw);                 // [kChunk][NT]  s conj(dat) ramp_t
    float* sC = reinterpret_cast<float*>(sA + kChunk * NT);           // [kChunk][ND]  ctf of the ND defocus factors
    float* sS = sC + kChunk * ND;                                     // [kChunk]      sigRcp
    int* sIc = reinterpret_cast<int*>(sS + kChunk);                   // [kChunk]
    int* sIr = sIc + kChunk;                                          // [kChunk]
    float* sRed = reinterpret_cast<float*>(sIr + kChunk);             // [4 waves]

    const int split = blockIdx.x, img = blockIdx.y;
    if (a.active && !a.active[img]) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = a.P;
    const float2* vol = a.volumes + (size_t)(a.volIdx ? a.volIdx[img] : 0) * ((size_t)P * P * (P / 2 + 1)) * (PACKED ? 8 : 1);
    const float2* dat = a.datP + (size_t)img * a.nPxl;
    const float* ctf = a.ctfP + (size_t)img * a.nD * a.nPxl;   // [nD][nPxl]
    const float* sig = a.sigRcpP + (size_t)img * a.nPxl;
    const double* tr = a.trans + (size_t)img * a.nT * 2;
    const int nRG0 = (a.nR + 63) >> 6;
    const int nRGp = nRG0 >= 4 ? 4 : (nRG0 >= 2 ? 2 : 1);
    const int nSub = 4 / nRGp;
    const int rg = wave % nRGp, sub = wave / nRGp;
    const int nPass = (nRG0 + nRGp - 1) / nRGp;
    const int nChunks = (a.nPxl + kChunk - 1) / kChunk;
    const int c0 = (int)(((long)nChunks * split) / a.nSplit), c1 = (int)(((long)nChunks * (split + 1)) / a.nSplit);
    float cpart = 0.f;

    for (int pass = 0; pass < nPass; pass++) {
        const int r = (pass * nRGp + rg) * 64 + lane;
        const bool rvalid = r < a.nR;
        double m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0, m5 = 0;
        if (rvalid) {
            const double* m = a.rotMat + ((size_t)img * a.nR + r) * 9;
            m0 = m[0]; m1 = m[1]; m2 = m[2]; m3 = m[3]; m4 = m[4]; m5 = m[5];
        }
        float acc[ND][NT];
        float accB[ND];
#pragma unroll
        for (int d = 0; d < ND; d++) {
            accB[d] = 0.f;
#pragma unroll
            for (int t = 0; t < NT; t++) acc[d][t] = 0.f;
        }
        for (int c = c0; c < c1; c++) {
            const int pbase = c * kChunk;
            const int clen = min(kChunk, a.nPxl - pbase);
            __syncthreads();
            for (int e = tid; e < clen; e += 256) {
                const int p = pbase + e;
                const int ic = a.iCol[p], ir = a.iRow[p];
                sIc[e] = ic * a.pf;
                sIr[e] = ir * a.pf;
                const float sv = sig[p];
                const float2 dv = dat[p];
                sS[e] = sv;
                if (pass == 0) cpart = fmaf(sv, fmaf(dv.x, dv.x, dv.y * dv.y), cpart);
                const float2 cd = make_float2(dv.x * sv, -dv.y * sv);  // s conj(dat)
#pragma unroll
                for (int t = 0; t < NT; t++) {
                    if (t < a.nT) {
                        const float rCol = (float)tr[2 * t] / a.idim, rRow = (float)tr[2 * t + 1] / a.idim;
                        sA[e * NT + t] = cmul(cd, ramp_value(rCol, rRow, ic, ir));
                    } else {
                        sA[e * NT + t] = make_float2(0.f, 0.f);
                    }
                }
#pragma unroll
                for (int d = 0; d < ND; d++) sC[e * ND + d] = d < a.nD ? ctf[(size_t)d * a.nPxl + p] : 0.f;
            }
            __syncthreads();
            if (rvalid) {
                for (int e = sub; e < clen; e += nSub) {
                    const double nx = (double)sIc[e], ny = (double)sIr[e];
                    const float x = (float)(m0 * nx + m3 * ny);
                    const float y = (float)(m1 * nx + m4 * ny);
                    const float z = (float)(m2 * nx + m5 * ny);
                    float2 q = make_float2(0.f, 0.f);
                    if (coord_in_grid(x, y, z, P))
                        q = PACKED ? interp_ft_packed(reinterpret_cast<const float4*>(vol), P, x, y, z) : interp_ft(vol, P, x, y, z);
                    const float sq = sS[e] * fmaf(q.x, q.x, q.y * q.y);
                    float u[NT];
                    const float2* Ap = sA + e * NT;
#pragma unroll
                    for (int t = 0; t < NT; t++) {
                        const float2 A = Ap[t];
                        u[t] = fmaf(-A.y, q.y, A.x * q.x);
                    }
                    const float* Cp = sC + e * ND;
#pragma unroll
                    for (int d = 0; d < ND; d++) {
                        const float cv = Cp[d];
                        accB[d] = fmaf(cv * cv, sq, accB[d]);
#pragma unroll
                        for (int t = 0; t < NT; t++) acc[d][t] = fmaf(cv, u[t], acc[d][t]);
                    }
                }
            }
        }
        // ---- combine the pixel sub-streams (fixed order), one defocus factor at a time through the same LDS scratch ----
        float* sAcc = reinterpret_cast<float*>(smem_raw);  // [nSub][NT+1][64*nRGp]
        const int slots = 64 * nRGp;
        const int slot = rg * 64 + lane;
#pragma unroll
        for (int d = 0; d < ND; d++) {
            if (d >= a.nD) break;
            __syncthreads();
#pragma unroll
            for (int t = 0; t < NT; t++) sAcc[(sub * (NT + 1) + t) * slots + slot] = acc[d][t];
            sAcc[(sub * (NT + 1) + NT) * slots + slot] = accB[d];
            __syncthreads();
            if (sub == 0 && rvalid) {
                float b = 0.f;
                for (int s2 = 0; s2 < nSub; s2++) b += sAcc[(s2 * (NT + 1) + NT) * slots + slot];
                float* outV = a.partV + ((((size_t)img * a.nD + d) * a.nSplit + split) * a.nT) * a.nRpad;
#pragma unroll
                for (int t = 0; t < NT; t++) {
                    if (t < a.nT) {
                        float v = 0.f;
                        for (int s2 = 0; s2 < nSub; s2++) v += sAcc[(s2 * (NT + 1) + t) * slots + slot];
                        outV[(size_t)t * a.nRpad + r] = b - 2.0f * v;
                    }
                }
            }
        }
        __syncthreads();
    }
    cpart = wave_sum(cpart);
    if (lane == 0) sRed[wave] = cpart;
    __syncthreads();
    if (tid == 0) {
        const float cs = (sRed[0] + sRed[1]) + (sRed[2] + sRed[3]);
        for (int d = 0; d < a.nD; d++) a.partC[((size_t)img * a.nD + d) * a.nSplit + split] = cs;
    }
}

// The one-image form leaves hundreds of partial sums per (shift, rotation): added up here in a FIXED order before the finalise
// kernel -- lane <-> element, the sixteen waves of a workgroup take a sixteenth of the splits each (loads issued sixteen at a time), wave 0
// adds the sixteen sums in wave order; the result overwrites split 0.  grid (ceil(nT nRpad / 64)), block 1024.  partC likewise (workgroup 0).
constexpr int kReduceWaves = 16;
__global__ __launch_bounds__(64 * kReduceWaves) void k_expect_reduce(float* __restrict__ partV, float* __restrict__ partC, int nSplit, int nElem)
{
    __shared__ float sq[kReduceWaves][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    const int s0 = (int)(((long)nSplit * wave) / kReduceWaves), s1 = (int)(((long)nSplit * (wave + 1)) / kReduceWaves);
    float v = 0.f;
    if (e < nElem) {
        for (int s = s0; s < s1; s += 16) {   // (sixteen loads in flight: three round trips for a wave's ~48 splits)
            float x[16];
#pragma unroll
            for (int u = 0; u < 16; u++) x[u] = (s + u < s1) ? partV[(size_t)(s + u) * nElem + e] : 0.f;
#pragma unroll
            for (int u = 0; u < 16; u++) v += x[u];
        }
    }
    sq[wave][lane] = v;
    __syncthreads();
    if (wave == 0 && e < nElem) {
        float t = sq[0][lane];
#pragma unroll
        for (int w = 1; w < kReduceWaves; w++) t += sq[w][lane];
        partV[e] = t;
    }
    if (blockIdx.x == 0) {   // (partV's writers above touch other memory: no hazard; every reader of partC is this workgroup)
        float c = 0.f;
        for (int s = threadIdx.x; s < nSplit; s += 64 * kReduceWaves) c += partC[s];
        c = wave_sum(c);
        __syncthreads();
        if (lane == 0) sq[0][wave] = c;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = sq[0][0];
            for (int w = 1; w < kReduceWaves; w++) t += sq[0][w];
            partC[0] = t;
        }
    }
}

// Finalise: L = C + V, per-image maximum, exp, marginals (src/Optimiser.cpp:1383-1402 in closed form).
// grid (nImg), block 256, dynamic LDS nD*nT*nR floats + (nR + nT + nD) doubles.
struct ExpectFinalArgs {
    const float* partV;
    const float* partC;
    int nSplit, nR, nRpad, nT, nD;
    const double* pC;
    double pCval;      // the prior of the class when pC is NULL (1 unless a one-image caller hands it over by value)
    const double* pR;
    const double* pT;
    const double* pD;
    float* wC;
    float* wR;
    float* wT;
    float* wD;
    float* baseLine;
    float* logW;
    const int* active;
    unsigned* done;    // (one-image form, may be NULL) host-visible word that receives doneVal once every output of the image is written and
    unsigned doneVal;  // fenced at system scope: a caller that polls it need not wait for the stream's own completion signal
    int par;           // != 0 (the one-image form): the marginals' sums are spread over a wave each instead of walked by one thread --
                       // a launch of ONE workgroup has nothing else to hide a 1 125-term serial loop behind (68 us measured)
};

__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ double block_sum_256(double v, double* sred)
{
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sred[0] + sred[1]) + (sred[2] + sred[3]);
}

__global__ __launch_bounds__(256) void k_expect_final(ExpectFinalArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* sL = reinterpret_cast<float*>(smem_raw);  // [nD][nT][nR] log-likelihoods
    __shared__ double sred[4];
    __shared__ float sfred[5];
    const int img = blockIdx.x, tid = threadIdx.x;
    if (a.active && !a.active[img]) return;
    const int n = a.nD * a.nT * a.nR;
    // the priors go to LDS first, their loads in flight with the partial sums': the marginals below then touch no global memory (a lone
    // workgroup -- the one-image form -- pays every dependent round trip in full)
    double* sPR = reinterpret_cast<double*>(smem_raw + (((size_t)n * sizeof(float) + 7) & ~(size_t)7));
    double* sPT = sPR + a.nR;
    double* sPD = sPT + a.nT;
    {
        const double* gR = a.pR + (size_t)img * a.nR;
        const double* gT = a.pT + (size_t)img * a.nT;
        const double* gD = a.pD + (size_t)img * a.nD;
        for (int i = tid; i < a.nR + a.nT + a.nD; i += 256)
            sPR[i] = i < a.nR ? gR[i] : (i < a.nR + a.nT ? gT[i - a.nR] : gD[i - a.nR - a.nT]);
    }
    float lmax = -INFINITY, cconst = 0.f;
    for (int e = tid; e < n; e += 256) {
        const int r = e % a.nR, t = (e / a.nR) % a.nT, d = e / (a.nR * a.nT);
        float v = 0.f, c = 0.f;
        for (int s = 0; s < a.nSplit; s++) {
            v += a.partV[((((size_t)img * a.nD + d) * a.nSplit + s) * a.nT + t) * a.nRpad + r];
            c += a.partC[((size_t)img * a.nD + d) * a.nSplit + s];
        }
        // L = C + V.  The weights only need V - max V: keeping the (large, sample-independent) constant C out of the
        // exponent's argument avoids losing eps*|C| of every log-likelihood difference.
        sL[e] = v;
        if (a.logW) a.logW[(size_t)img * n + e] = c + v;
        lmax = fmaxf(lmax, v);
        cconst = c;
    }
    lmax = wave_max(lmax);
    if ((tid & 63) == 0) sfred[tid >> 6] = lmax;
    __syncthreads();
    const float vmax = fmaxf(fmaxf(sfred[0], sfred[1]), fmaxf(sfred[2], sfred[3]));
    if (tid == 0) sfred[4] = cconst;   // C = sum s|dat|^2 does not depend on (r, t, d)
    __syncthreads();
    // s = exp(w - baseLine) (src/Optimiser.cpp:1397, float)
    for (int e = tid; e < n; e += 256) sL[e] = expf(sL[e] - vmax);
    __syncthreads();
    const float base = sfred[4] + vmax;
    const double pC = a.pC ? a.pC[img] : a.pCval;
    const double *pR = sPR, *pT = sPT, *pD = sPD;
    // wR
    for (int r = tid; r < a.nR; r += 256) {
        double s = 0;
        for (int d = 0; d < a.nD; d++)
            for (int t = 0; t < a.nT; t++) s += (double)sL[(d * a.nT + t) * a.nR + r] * (pC * pT[t] * pD[d]);
        a.wR[(size_t)img * a.nR + r] = (float)s;
    }
    if (a.par) {   // a wave per shift / defocus factor: lanes stride over the other indices, the 64 partial sums added by butterflies
        const int lane = tid & 63, wave = tid >> 6;
        for (int t = wave; t < a.nT; t += 4) {
            double s = 0;
            for (int q = lane; q < a.nD * a.nR; q += 64) {
                const int d = q / a.nR, r = q - d * a.nR;
                s += (double)sL[(d * a.nT + t) * a.nR + r] * (pC * pR[r] * pD[d]);
            }
            s = wave_sum_f64(s);
            if (lane == 0) a.wT[(size_t)img * a.nT + t] = (float)s;
        }
        for (int d = wave; d < a.nD; d += 4) {
            double s = 0;
            for (int q = lane; q < a.nT * a.nR; q += 64) {
                const int t = q / a.nR, r = q - t * a.nR;
                s += (double)sL[(d * a.nT + t) * a.nR + r] * (pC * pR[r] * pT[t]);
            }
            s = wave_sum_f64(s);
            if (lane == 0) a.wD[(size_t)img * a.nD + d] = (float)s;
        }
    } else {
        for (int t = tid; t < a.nT; t += 256) {
            double s = 0;
            for (int d = 0; d < a.nD; d++)
                for (int r = 0; r < a.nR; r++) s += (double)sL[(d * a.nT + t) * a.nR + r] * (pC * pR[r] * pD[d]);
            a.wT[(size_t)img * a.nT + t] = (float)s;
        }
        for (int d = tid; d < a.nD; d += 256) {
            double s = 0;
            for (int t = 0; t < a.nT; t++)
                for (int r = 0; r < a.nR; r++) s += (double)sL[(d * a.nT + t) * a.nR + r] * (pC * pR[r] * pT[t]);
            a.wD[(size_t)img * a.nD + d] = (float)s;
        }
    }
    double sc = 0;
    for (int e = tid; e < n; e += 256) {
        const int r = e % a.nR, t = (e / a.nR) % a.nT, d = e / (a.nR * a.nT);
        sc += (double)sL[e] * (pR[r] * pT[t] * pD[d]);
    }
    sc = block_sum_256(sc, sred);
    if (tid == 0) {
        a.wC[img] = (float)sc;
        a.baseLine[img] = base;
    }
    if (a.done) {   // (uniform)
        __threadfence_system();
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_store(a.done, a.doneVal, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Global scanning phase (src/Optimiser.cpp:756-894): every image against every (rotation, shift)
// of one class.  Slices and ramps are shared by all images (projected once), so the per-sample cost
// is streaming dat/ctf/sigRcp (16 B per pixel) against L2-resident slices.
// Stage 1: dvp[l][m][n] via the same expansion as above; grid (ceil(nR/64), nImg), lane <-> rotation,
//          pixels walked sequentially, ramps for NT shifts folded into A_t per pixel in LDS.
// Stage 2: per image, fold (baseline, weights) into the carried wC / wR / wT exactly as the reference's
//          running rescale does, but with one maximum per class sweep.
// ---------------------------------------------------------------------------------------------
struct ExpectGlobalArgs {
    const float2* rotPT;   // slices transposed to pixel-major [nPxl][nR]: lanes (= rotations) read consecutive addresses
    const float2* traP;
    const float2* datP;
    const float* ctfP;
    const float* sigRcpP;
    int nR, nT, nPxl, nImg;
    float* dvp;  // [nImg][nT][nR]
};

// (Tried and measured on MI355X, 1024 images x 10 000 rotations x 30 shifts at 866 pixels, per class: this kernel 31 ms =
// 34 TFLOP/s; 4 rotations x 10 shifts per thread (LDS reads amortised, 98 VGPRs) 37 ms; 4 images x 16 shifts per thread with
// the A / B operands as SGPRs through the scalar cache, no LDS, 41 ms.  All three sit at the same level: what is missing is
// the classic two-operand LDS tiling of a GEMM -- see DESIGN.md, what comes next.)
template <int NT>
__global__ __launch_bounds__(256) void k_expect_global(ExpectGlobalArgs a, int t0)
{
    __shared__ float2 sA[kChunk * NT];
    __shared__ float sB[kChunk];
    __shared__ float sRed[4];
    const int img = blockIdx.y;
    const int tid = threadIdx.x;
    const int r = blockIdx.x * 256 + tid;
    const bool rvalid = r < a.nR;
    const float2* dat = a.datP + (size_t)img * a.nPxl;
    const float* ctf = a.ctfP + (size_t)img * a.nPxl;
    const float* sig = a.sigRcpP + (size_t)img * a.nPxl;
    const int nt = min(NT, a.nT - t0);
    float acc[NT];
    float accB = 0.f, cpart = 0.f;
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = 0.f;
    for (int pbase = 0; pbase < a.nPxl; pbase += kChunk) {
        const int clen = min(kChunk, a.nPxl - pbase);
        __syncthreads();
        for (int e = tid; e < clen; e += 256) {
            const int p = pbase + e;
            const float s = sig[p], cf = ctf[p];
            const float2 dv = dat[p];
            const float g = s * cf;
            sB[e] = g * cf;
            cpart = fmaf(s, fmaf(dv.x, dv.x, dv.y * dv.y), cpart);
            const float2 cd = make_float2(dv.x * g, -dv.y * g);
#pragma unroll
            for (int t = 0; t < NT; t++)
                if (t < nt) sA[e * NT + t] = cmul(cd, a.traP[(size_t)(t0 + t) * a.nPxl + p]);
        }
        __syncthreads();
        if (rvalid) {
            const float2* pr = a.rotPT + (size_t)pbase * a.nR + r;
            for (int e = 0; e < clen; e++) {
                const float2 q = pr[(size_t)e * a.nR];
                accB = fmaf(sB[e], fmaf(q.x, q.x, q.y * q.y), accB);
#pragma unroll
                for (int t = 0; t < NT; t++) {
                    const float2 A = sA[e * NT + t];
                    acc[t] = fmaf(A.x, q.x, acc[t]);
                    acc[t] = fmaf(-A.y, q.y, acc[t]);
                }
            }
        }
    }
    cpart = wave_sum(cpart);
    if ((tid & 63) == 0) sRed[tid >> 6] = cpart;
    __syncthreads();
    const float C = (sRed[0] + sRed[1]) + (sRed[2] + sRed[3]);
    if (rvalid) {
#pragma unroll
        for (int t = 0; t < NT; t++)
            if (t < nt) a.dvp[((size_t)img * a.nT + t0 + t) * a.nR + r] = C + (accB - 2.0f * acc[t]);
    }
}

// [nR][nPxl] -> [nPxl][nR] through a padded 32x32 LDS tile (coalesced on both sides)
__global__ __launch_bounds__(256) void k_transpose_c64(float2* __restrict__ dst, const float2* __restrict__ src, int nR, int nPxl,
                                                       long ldd)
{
    __shared__ float2 tile[32][33];
    const int p0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int k = ty; k < 32; k += 8)
        if (r0 + k < nR && p0 + tx < nPxl) tile[k][tx] = src[(size_t)(r0 + k) * nPxl + p0 + tx];
    __syncthreads();
    for (int k = ty; k < 32; k += 8)
        if (p0 + k < nPxl && r0 + tx < nR) dst[(size_t)(p0 + k) * ldd + r0 + tx] = tile[tx][k];
}

// ---------------------------------------------------------------------------------------------
// The same stage as a two-operand LDS-tiled contraction (the form for production sizes: thousands of images x 10^4
// rotations).  Per class
//   acc[m][n]  = sum_k A[k][m] Bq[k][n],   m = (image, shift), n = rotation, k = (pixel, re | im): A[2p] = Re A_t, A[2p+1] = -Im A_t
//   accB[i][n] = sum_p sB[p][i] |q(p, n)|^2
//   dvp[m][n]  = C_image + (accB[image][n] - 2 acc[m][n])
// with every sum running over k in ascending order inside one thread -- the SAME sequence of fmaf's as k_expect_global, so
// the results are bit-identical to it (tests compare both with the oracle).
// ---------------------------------------------------------------------------------------------
#ifndef THX_SCAN_BK
#define THX_SCAN_BK 16
#endif
constexpr int kGM = 128, kGN = 128, kGK = THX_SCAN_BK;   // depth of one LDS step (a multiple of 16)

// A operand, k-major: tabA[2p][m] = Re(s ctf conj(dat) ramp_t), tabA[2p+1][m] = -Im(...); tabS[p][i] = s ctf^2; tabC[i].
// grid (nPxl, ceil(nImg nT / 256)), block 256: lane <-> m = (image, shift), so the two rows of a pixel are written in
// 256-byte runs (the image's three inputs are the same address for the nT lanes of an image: broadcast loads)
__global__ __launch_bounds__(256) void k_scan_tables(float* __restrict__ tabA, float* __restrict__ tabS, ExpectGlobalArgs a, long Mpad,
                                                     long Ipad)
{
    const int p = blockIdx.x, m = blockIdx.y * 256 + threadIdx.x;
    if (m >= a.nImg * a.nT) return;
    const int img = m / a.nT, t = m - img * a.nT;
    const float s = a.sigRcpP[(size_t)img * a.nPxl + p], cf = a.ctfP[(size_t)img * a.nPxl + p];
    const float2 dv = a.datP[(size_t)img * a.nPxl + p];
    const float g = s * cf;
    if (t == 0) tabS[(size_t)p * Ipad + img] = g * cf;
    const float2 cd = make_float2(dv.x * g, -dv.y * g);
    const float2 A = cmul(cd, a.traP[(size_t)t * a.nPxl + p]);
    tabA[(size_t)(2 * p) * Mpad + m] = A.x;
    tabA[(size_t)(2 * p + 1) * Mpad + m] = -A.y;
}

// C_image = sum_p s |dat|^2 in the order k_expect_global sums it (per-thread strided partials, wave tree, 4 waves)
__global__ __launch_bounds__(256) void k_scan_const(float* __restrict__ tabC, ExpectGlobalArgs a)
{
    __shared__ float sRed[4];
    const int img = blockIdx.x, tid = threadIdx.x;
    float cpart = 0.f;
    for (int p = tid; p < a.nPxl; p += 256) {
        const float2 dv = a.datP[(size_t)img * a.nPxl + p];
        cpart = fmaf(a.sigRcpP[(size_t)img * a.nPxl + p], fmaf(dv.x, dv.x, dv.y * dv.y), cpart);
    }
    cpart = wave_sum(cpart);
    if ((tid & 63) == 0) sRed[tid >> 6] = cpart;
    __syncthreads();
    if (tid == 0) tabC[img] = (sRed[0] + sRed[1]) + (sRed[2] + sRed[3]);
}

// |q|^2 of the transposed slices: q2[p][n] = fmaf(q.x, q.x, q.y q.y)
__global__ __launch_bounds__(256) void k_scan_q2(float* __restrict__ q2, const float2* __restrict__ rotPT, size_t n)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const float2 q = rotPT[e];
    q2[e] = fmaf(q.x, q.x, q.y * q.y);
}

// QPAIR = true : B operand is the float2 array rotPT [K/2][ldb] read as rows (2p: re, 2p+1: im); epilogue writes dvp
// QPAIR = false: B operand is the float array q2 [K][ldb]; epilogue writes accB [M][N] (M = images)
// K is a multiple of 16 (zero rows appended) with 16 spare rows allocated behind it, lda / ldb cover whole tiles.
// The products run on v_mfma_f32_32x32x2_f32, whose result is bit-for-bit the k-ordered fmaf chain
// D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)) (one rounding per product, no wider accumulator) at the vector pipe's peak rate
// without its operand traffic: one VGPR per operand per lane, 4 LDS dwords per 4 instructions.  Workgroup = 4 waves as
// 2 x 2, each wave TM x TN tiles of 32 x 32; LDS tiles double-buffered (one barrier per 16-deep step), the next step's
// global loads in flight during the multiply.  Workgroups are numbered so that the 64 that share an XCD's L2 at any
// time form an 8 x 8 patch of the output (8 A panels + 8 B panels for 64 workgroups).
template <bool QPAIR, int TM, int TN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TM * TN <= 4 ? 1 : (TM * TN <= 8 ? 2 : 1), TM * TN <= 4 ? 3 : (TM * TN <= 8 ? 2 : 1)))) void k_scan_gemm(float* __restrict__ out, const float* A, const void* Bv,
                                                   const float* __restrict__ accB, const float* __restrict__ tabC, int M, int N, int K,
                                                   long lda, long ldb, int nT, int tilesM, int tilesN)
{
    constexpr int BM = 64 * TM, BN = 64 * TN;
    typedef float f16v __attribute__((ext_vector_type(16)));
    __shared__ __attribute__((aligned(16))) float As[2][kGK][BM];
    __shared__ __attribute__((aligned(16))) float Bs[2][kGK][BN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware numbering: workgroup b runs on XCD b % 8; its slot b / 8 walks 8 x 8 patches
    int tm, tn;
    {
        const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
        const int patch = (slot >> 6) * 8 + xcd, within = slot & 63;
        const int patchesN = (tilesN + 7) / 8;
        tn = (patch % patchesN) * 8 + (within & 7);
        tm = (patch / patchesN) * 8 + (within >> 3);
        if (tn >= tilesN || tm >= tilesM) return;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    // global -> register staging.  The operands are padded by the launcher (rows to a multiple of 16 with zeros, columns to
    // whole tiles), so a step's loads are unconditional 16-byte loads from pointers that advance by a constant: nothing
    // but the loads themselves is issued per step, and all of them are in flight during the multiply.
    typedef float f4v __attribute__((ext_vector_type(4)));
    typedef float f2v __attribute__((ext_vector_type(2)));
    constexpr int NA = TM * kGK / 16, NB = TN * kGK / 16;   // 16-byte loads per thread and step
    f4v ra[NA], rb[NB];
    size_t oa[NA], ob[NB];   // running offsets (floats) into A / B
    const float* Bf = reinterpret_cast<const float*>(Bv);
#pragma unroll
    for (int u = 0; u < NA; u++) {
        const int idx = tid + 256 * u;
        oa[u] = (size_t)(idx / (BM / 4)) * lda + m0 + (idx % (BM / 4)) * 4;
    }
#pragma unroll
    for (int u = 0; u < NB; u++) {
        const int idx = tid + 256 * u;
        if (QPAIR) ob[u] = 2 * ((size_t)(idx / (BN / 2)) * ldb + n0 + (idx % (BN / 2)) * 2);
        else ob[u] = (size_t)(idx / (BN / 4)) * ldb + n0 + (idx % (BN / 4)) * 4;
    }
    const size_t stepA = (size_t)kGK * lda, stepB = (size_t)kGK * ldb;   // floats per step (QPAIR: 8 rows of float2)
#define THX_SCAN_LOAD()                                                 \
    {                                                                   \
        _Pragma("unroll") for (int u = 0; u < NA; u++) {                \
            ra[u] = *reinterpret_cast<const f4v*>(A + oa[u]);           \
            oa[u] += stepA;                                             \
        }                                                               \
        _Pragma("unroll") for (int u = 0; u < NB; u++) {                \
            rb[u] = *reinterpret_cast<const f4v*>(Bf + ob[u]);          \
            ob[u] += stepB;                                             \
        }                                                               \
    }
#define THX_SCAN_STORE(buf)                                                                                         \
    {                                                                                                               \
        _Pragma("unroll") for (int u = 0; u < NA; u++) {                                                            \
            const int idx = tid + 256 * u;                                                                          \
            *reinterpret_cast<f4v*>(&As[buf][idx / (BM / 4)][(idx % (BM / 4)) * 4]) = ra[u];                        \
        }                                                                                                           \
        _Pragma("unroll") for (int u = 0; u < NB; u++) {                                                            \
            const int idx = tid + 256 * u;                                                                          \
            if (QPAIR) { /* (re, im) of two rotations of pixel pr -> rows 2 pr (re) and 2 pr + 1 (im) */            \
                const int pr = idx / (BN / 2), n = (idx % (BN / 2)) * 2;                                            \
                *reinterpret_cast<f2v*>(&Bs[buf][2 * pr][n]) = f2v{rb[u].x, rb[u].z};                              \
                *reinterpret_cast<f2v*>(&Bs[buf][2 * pr + 1][n]) = f2v{rb[u].y, rb[u].w};                          \
            } else {                                                                                                \
                *reinterpret_cast<f4v*>(&Bs[buf][idx / (BN / 4)][(idx % (BN / 4)) * 4]) = rb[u];                    \
            }                                                                                                       \
        }                                                                                                           \
    }
    f16v acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    // operand lanes: A[i = lane % 32][k = lane / 32], B[k = lane / 32][j = lane % 32]
    const int kh = lane >> 5, l32 = lane & 31;
    const int wm = (wave >> 1) * (32 * TM), wn = (wave & 1) * (32 * TN);
    THX_SCAN_LOAD();
    THX_SCAN_STORE(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += kGK) {
        // the next step's operands, in flight during the multiply (the step after the last reads the 16 spare rows the
        // launcher allocates behind each operand; they are staged and never multiplied)
        THX_SCAN_LOAD();
        asm volatile("" ::: "memory");   // the loads are issued here, not sunk to their use behind the multiplies
        float av[2][TM], bv[2][TN];   // operands of step s + 1 are read while step s multiplies
#pragma unroll
        for (int i = 0; i < TM; i++) av[0][i] = As[buf][kh][wm + 32 * i + l32];
#pragma unroll
        for (int j = 0; j < TN; j++) bv[0][j] = Bs[buf][kh][wn + 32 * j + l32];
#pragma unroll
        for (int s = 0; s < kGK / 2; s++) {
            if (s + 1 < kGK / 2) {
#pragma unroll
                for (int i = 0; i < TM; i++) av[(s + 1) & 1][i] = As[buf][2 * s + 2 + kh][wm + 32 * i + l32];
#pragma unroll
                for (int j = 0; j < TN; j++) bv[(s + 1) & 1][j] = Bs[buf][2 * s + 2 + kh][wn + 32 * j + l32];
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the reads ahead of the multiplies they overlap with
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s & 1][i], bv[s & 1][j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);   // the staging stores stay behind the multiplies (their loads need the time)
        THX_SCAN_STORE(buf ^ 1);   // the other buffer's readers finished before the previous barrier
        __syncthreads();
        buf ^= 1;
    }
#undef THX_SCAN_LOAD
#undef THX_SCAN_STORE
    // epilogue: accumulator register r of lane l is D[8 (r / 4) + 4 (l / 32) + r % 4][l % 32].  Four rows at a time: the
    // loads first (clamped addresses, no branch: 4 x TN in flight), then the predicated stores.
#pragma unroll
    for (int i = 0; i < TM; i++) {
#pragma unroll
        for (int rc = 0; rc < 4; rc++) {
            float ab[4][TN], cc[4];
            const int mb = m0 + wm + 32 * i + 8 * rc + 4 * kh;
            if (QPAIR) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int img = min(mb + r, M - 1) / nT;
                    cc[r] = tabC[img];
#pragma unroll
                    for (int j = 0; j < TN; j++) ab[r][j] = accB[(size_t)img * N + min(n0 + wn + 32 * j + l32, N - 1)];
                }
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    const int n = n0 + wn + 32 * j + l32;
                    const float v = acc[i][j][4 * rc + r];
                    const float o = QPAIR ? cc[r] + (ab[r][j] - 2.0f * v) : v;
                    if (mb + r < M && n < N) out[(size_t)(mb + r) * N + n] = o;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// stage 2: grid (nImg), block 256
__global__ __launch_bounds__(256) void k_expect_global_fold(const float* __restrict__ dvp, const double* __restrict__ pR,
                                                            const double* __restrict__ pT, float* __restrict__ wC,
                                                            float* __restrict__ wR, float* __restrict__ wT,
                                                            float* __restrict__ baseL, int kIdx, int nK, int nR, int nT,
                                                            int nImg)
{
    __shared__ float sfred[4];
    __shared__ double sred[4];
    const int l = blockIdx.x, tid = threadIdx.x;
    const float* d = dvp + (size_t)l * nT * nR;
    const int n = nT * nR;
    float lmax = -INFINITY;
    for (int e = tid; e < n; e += 256) lmax = fmaxf(lmax, d[e]);
    lmax = wave_max(lmax);
    if ((tid & 63) == 0) sfred[tid >> 6] = lmax;
    __syncthreads();
    lmax = fmaxf(fmaxf(sfred[0], sfred[1]), fmaxf(sfred[2], sfred[3]));
    const float old = baseL[l];
    const bool unset = isnan(old);
    const float base = unset ? lmax : fmaxf(old, lmax);
    // rescale what earlier classes / sweeps accumulated (src/Optimiser.cpp:843-871)
    const float nf = unset ? 1.0f : expf(old - base);
    if (!unset && base > old) {
        for (int q = tid; q < nK; q += 256) wC[(size_t)l * nK + q] *= nf;
        for (int td = 0; td < nK; td++) {
            float* a = wR + ((size_t)td * nImg + l) * nR;
            float* b = wT + ((size_t)td * nImg + l) * nT;
            for (int q = tid; q < nR; q += 256) a[q] *= nf;
            for (int q = tid; q < nT; q += 256) b[q] *= nf;
        }
    }
    __syncthreads();
    const double* pr = pR + (size_t)l * nR;
    const double* pt = pT + (size_t)l * nT;
    float* a = wR + ((size_t)kIdx * nImg + l) * nR;
    float* b = wT + ((size_t)kIdx * nImg + l) * nT;
    // wR[m] += sum_n w * pT[n]
    for (int m = tid; m < nR; m += 256) {
        double s = 0;
        for (int t = 0; t < nT; t++) s += (double)expf(d[(size_t)t * nR + m] - base) * pt[t];
        a[m] = (float)((double)a[m] + s);
    }
    // wT[n] += sum_m w * pR[m]; wC += sum w pR pT
    double sc = 0;
    for (int t = 0; t < nT; t++) {
        double s = 0;
        for (int m = tid; m < nR; m += 256) s += (double)expf(d[(size_t)t * nR + m] - base) * pr[m];
        s = block_sum_256(s, sred);
        if (tid == 0) b[t] = (float)((double)b[t] + s);
        sc += s * pt[t];
    }
    if (tid == 0) {
        wC[(size_t)l * nK + kIdx] = (float)((double)wC[(size_t)l * nK + kIdx] + sc);
        baseL[l] = base;
    }
}

// stage 2 for nT <= 32 (every configuration of the reference's scan: 30 shifts at most): the weights of one (shift,
// rotation) are exponentiated once; a thread owns rotations m = tid, tid + 256, ... and carries the nT partial sums of
// wT in registers.  Same summation orders as k_expect_global_fold, so the same bits.
__global__ __launch_bounds__(256) void k_expect_global_fold32(const float* __restrict__ dvp, const double* __restrict__ pR,
                                                              const double* __restrict__ pT, float* __restrict__ wC,
                                                              float* __restrict__ wR, float* __restrict__ wT,
                                                              float* __restrict__ baseL, int kIdx, int nK, int nR, int nT,
                                                              int nImg)
{
    __shared__ float sfred[4];
    __shared__ double sred[4];
    __shared__ double sPt[32];
    const int l = blockIdx.x, tid = threadIdx.x;
    const float* d = dvp + (size_t)l * nT * nR;
    const int n = nT * nR;
    float lmax = -INFINITY;
    if ((((size_t)l * n) & 3) == 0) {   // 16-byte loads when the image's block is aligned
        const float4* d4 = reinterpret_cast<const float4*>(d);
        for (int e = tid; e < n / 4; e += 256) {
            const float4 v = d4[e];
            lmax = fmaxf(lmax, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
        }
        for (int e = (n / 4) * 4 + tid; e < n; e += 256) lmax = fmaxf(lmax, d[e]);
    } else {
        for (int e = tid; e < n; e += 256) lmax = fmaxf(lmax, d[e]);
    }
    lmax = wave_max(lmax);
    if ((tid & 63) == 0) sfred[tid >> 6] = lmax;
    if (tid < 32) sPt[tid] = tid < nT ? pT[(size_t)l * nT + tid] : 0.0;
    __syncthreads();
    lmax = fmaxf(fmaxf(sfred[0], sfred[1]), fmaxf(sfred[2], sfred[3]));
    const float old = baseL[l];
    const bool unset = isnan(old);
    const float base = unset ? lmax : fmaxf(old, lmax);
    const float nf = unset ? 1.0f : expf(old - base);
    if (!unset && base > old) {   // rescale what earlier classes / sweeps accumulated (src/Optimiser.cpp:843-871)
        for (int q = tid; q < nK; q += 256) wC[(size_t)l * nK + q] *= nf;
        for (int td = 0; td < nK; td++) {
            float* a = wR + ((size_t)td * nImg + l) * nR;
            float* b = wT + ((size_t)td * nImg + l) * nT;
            for (int q = tid; q < nR; q += 256) a[q] *= nf;
            for (int q = tid; q < nT; q += 256) b[q] *= nf;
        }
    }
    __syncthreads();
    const double* pr = pR + (size_t)l * nR;
    float* a = wR + ((size_t)kIdx * nImg + l) * nR;
    float* b = wT + ((size_t)kIdx * nImg + l) * nT;
    double sT[32];
#pragma unroll
    for (int t = 0; t < 32; t++) sT[t] = 0;
    for (int m = tid; m < nR; m += 256) {
        const double prm = pr[m];
        float dv[32];
#pragma unroll
        for (int t = 0; t < 32; t++) dv[t] = t < nT ? d[(size_t)t * nR + m] : 0.f;   // all of a rotation's loads in flight
        double s = 0;
#pragma unroll
        for (int t = 0; t < 32; t++) {
            if (t < nT) {
                const double w = (double)expf(dv[t] - base);
                s += w * sPt[t];
                sT[t] += w * prm;
            }
        }
        a[m] = (float)((double)a[m] + s);
    }
    double sc = 0;
#pragma unroll
    for (int t = 0; t < 32; t++) {
        if (t < nT) {   // uniform
            const double s = block_sum_256(sT[t], sred);
            if (tid == 0) b[t] = (float)((double)b[t] + s);
            sc += s * sPt[t];
        }
    }
    if (tid == 0) {
        wC[(size_t)l * nK + kIdx] = (float)((double)wC[(size_t)l * nK + kIdx] + sc);
        baseL[l] = base;
    }
}

}  // namespace thx

using namespace thx;

static int expect_local_nsplit(int nImg)
{
    // enough blocks to fill 256 CUs x ~6 resident blocks when the batch is small
    int s = 1;
    while (s < 16 && (long)nImg * s < 2048) s *= 2;
    if (knobs().expectNSplit) s = knobs().expectNSplit;
    return s;
}

// Workgroups of k_expect_local per CU.  With particle-filter clouds (rotations ~1 degree apart) every wave-load is 64
// scattered 64-byte requests and the kernel runs FASTER with fewer waves in flight: 8 waves per CU (2 workgroups) 162 ms,
// 12 waves 182 ms, 16-20 waves 194 ms, 4 waves 237 ms per 5 000-image launch on MI355X -- beyond ~2 waves per SIMD the
// extra requests only thrash L2 / the memory queues.  With tightly clustered rotations the loads coalesce and the
// kernel wants every wave it can get (82 ms unlimited, 113 ms at 2 workgroups).  The cap is applied by rounding the
// dynamic LDS request up so that one more workgroup does not fit the CU's 160 KB.  0 = unlimited; it is an argument of
// thx_expect_local_dev (negative = this default).
constexpr int kExpectWgPerCUDefault = 2;

static size_t expect_lds_floor(int wg)
{
    if (wg < 0) wg = kExpectWgPerCUDefault;
    if (knobs().expectWgPerCU >= 0) wg = knobs().expectWgPerCU;
    if (wg <= 0 || wg > 8) return 0;
    const size_t f = (size_t)(160 * 1024) / (wg + 1) + 1024;
    return f > 64 * 1024 ? 64 * 1024 : f;   // one workgroup per CU would need > 80 KB: not offered (it is the slowest)
}

template <int NT>
static int launch_expect_local(const ExpectLocalArgs& a, hipStream_t st, bool packed, int wgPerCU)
{
    const int nRG0 = (a.nR + 63) >> 6;
    const int nRGp = nRG0 >= 4 ? 4 : (nRG0 >= 2 ? 2 : 1);
    const int nSub = 4 / nRGp;
    size_t stage = (size_t)kChunk * NT * sizeof(float2) + kChunk * (sizeof(float) + 2 * sizeof(double)) + 4 * sizeof(float);
    size_t red = (size_t)nSub * (NT + 1) * 64 * nRGp * sizeof(float);
    size_t lds = stage > red ? stage : red;
    {
        const size_t f = expect_lds_floor(wgPerCU);
        lds = lds > f ? lds : f;
    }
    // defocus search: one pass serves all nD factors when the accumulators fit (nT, nD <= 9); THX_EXPECT_ND=sweep keeps the
    // one-sweep-per-factor form for A/B runs
    {
        if (NT == 9 && a.nD > 1 && a.nD <= 9 && !knobs().expectNdSweep) {
            constexpr int ND = 9;
            size_t stageN = (size_t)kChunk * (NT * sizeof(float2) + ND * sizeof(float) + sizeof(float) + 2 * sizeof(int)) + 4 * sizeof(float);
            size_t ldsN = stageN > red ? stageN : red;
            if (packed)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_expect_local_nd<9, ND, true>), dim3(a.nSplit, a.nImg), dim3(256), ldsN, st, a);
            else
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_expect_local_nd<9, ND, false>), dim3(a.nSplit, a.nImg), dim3(256), ldsN, st, a);
            THX_LAUNCH_CHECK();
            return 0;
        }
    }
    if (lds > 64 * 1024) {   // NT = 32: 68.6 KB of stage table, above the default dynamic-LDS limit
        THX_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_expect_local<NT, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        THX_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_expect_local<NT, false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    // (dealing the images of a launch to the XCDs in contiguous runs, so that orientation-sorted neighbours share an L2,
    // was measured with view-ordered particles: no gain -- the reuse between neighbouring images happens in the Infinity Cache)
    if (packed && NT == 9 && knobs().expectSplit > 0.f) {   // A/B: the near-slab / tail form (THX_EXPECT_SPLIT = margin in voxels)
        ExpectLocalArgs b = a;
        b.splitM = knobs().expectSplit;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_expect_local<NT, true, true>), dim3(a.nSplit, a.nImg, a.nD), dim3(256), lds, st, b);
    } else if (packed)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_expect_local<NT, true>), dim3(a.nSplit, a.nImg, a.nD), dim3(256), lds, st, a);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_expect_local<NT, false>), dim3(a.nSplit, a.nImg, a.nD), dim3(256), lds, st, a);
    THX_LAUNCH_CHECK();
    return 0;
}

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

int thx_release_stream(void* stream)
{
    return release_stream(as_stream(stream));
}


const char* thx_last_error(void) { return thx::g_err; }
int thx_version(void) { return 200; }

// Test-harness hook: re-reads the THX_* environment switches (they are otherwise read once, when the library is loaded).
// Not for production use: must not run concurrently with launches.
int thx_knobs_reload(void)
{
    thx::g_knobs = thx::read_knobs();
    return 0;
}

int thx_device_count(int* count)
{
    THX_REQUIRE(count, "count is NULL");
    THX_CHECK(hipGetDeviceCount(count));
    return 0;
}

int thx_set_device(int gpuIdx)
{
    THX_CHECK(hipSetDevice(gpuIdx));
    return 0;
}

int thx_rotmat_dev(const double* quat, double* mat, int n, void* stream)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_rotmat, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), quat, mat, n);
    THX_LAUNCH_CHECK();
    return 0;
}

int thx_translate_dev(float* traP, const double* trans, int nT, const int* iCol, const int* iRow, int nPxl, int idim,
                      void* stream)
{
    if (nT <= 0 || nPxl <= 0) return 0;
    hipLaunchKernelGGL(k_translate, dim3((nPxl + 255) / 256, nT), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<float2*>(traP), trans, iCol, iRow, nPxl, idim);
    THX_LAUNCH_CHECK();
    return 0;
}

int thx_ctf_dev(float* ctfP, const thx_ctf_attr* attr, const double* dfac, float pixelSize, const int* iCol,
                const int* iRow, int nPxl, int idim, int nImg, void* stream)
{
    if (nImg <= 0 || nPxl <= 0) return 0;
    for (int l0 = 0; l0 < nImg; l0 += 65535) {
        const int nl = nImg - l0 < 65535 ? nImg - l0 : 65535;
        hipLaunchKernelGGL(k_ctf, dim3((nPxl + 255) / 256, nl), dim3(256), 0, as_stream(stream),
                           ctfP + (size_t)l0 * nPxl, attr + l0, dfac ? dfac + l0 : nullptr, pixelSize, iCol, iRow, nPxl,
                           idim);
    }
    THX_LAUNCH_CHECK();
    return 0;
}

int thx_expect_precal_dev(float* freq, float* def, float* k1, float* k2, const thx_ctf_attr* attr, int idim,
                          float pixelSize, const int* iCol, const int* iRow, int nPxl, int nImg, void* stream)
{
    if (nImg <= 0 || nPxl <= 0) return 0;
    THX_REQUIRE(def && k1 && k2 && attr && iCol && iRow, "NULL pointer");
    for (int l0 = 0; l0 < nImg; l0 += 65535) {
        const int nl = nImg - l0 < 65535 ? nImg - l0 : 65535;
        hipLaunchKernelGGL(k_expect_precal, dim3((nPxl + 255) / 256, nl), dim3(256), 0, as_stream(stream),
                           l0 == 0 ? freq : nullptr, def + (size_t)l0 * nPxl, k1 + l0, k2 + l0, attr + l0, idim, pixelSize,
                           iCol, iRow, nPxl);
    }
    THX_LAUNCH_CHECK();
    return 0;
}

int thx_ctf_dsearch_dev(float* ctfP, const float* freq, const float* def, const float* k1, const float* k2,
                        const thx_ctf_attr* attr, const double* dpara, int nD, int nPxl, int nImg, void* stream)
{
    if (nImg <= 0 || nPxl <= 0 || nD <= 0) return 0;
    THX_REQUIRE(ctfP && freq && def && k1 && k2 && attr && dpara, "NULL pointer");
    THX_REQUIRE(nD <= 65535, "nD too large");
    for (int l0 = 0; l0 < nImg; l0 += 65535) {
        const int nl = nImg - l0 < 65535 ? nImg - l0 : 65535;
        hipLaunchKernelGGL(k_ctf_dsearch, dim3((nPxl + 255) / 256, nD, nl), dim3(256), 0, as_stream(stream),
                           ctfP + (size_t)l0 * nD * nPxl, freq, def + (size_t)l0 * nPxl, k1 + l0, k2 + l0, attr + l0,
                           dpara + (size_t)l0 * nD, nD, nPxl);
    }
    THX_LAUNCH_CHECK();
    return 0;
}

int thx_ctf_image_dev(float* ctfFT, const thx_ctf_attr* attr, float pixelSize, int idim, int nImg, void* stream)
{
    if (nImg <= 0) return 0;
    THX_REQUIRE(ctfFT && attr && idim > 0, "bad arguments");
    for (int l0 = 0; l0 < nImg; l0 += 65535) {
        const int nl = nImg - l0 < 65535 ? nImg - l0 : 65535;
        hipLaunchKernelGGL(k_ctf_image, dim3(idim, nl), dim3(128), 0, as_stream(stream),
                           reinterpret_cast<float2*>(ctfFT) + (size_t)l0 * idim * (idim / 2 + 1), attr + l0, pixelSize,
                           idim);
    }
    THX_LAUNCH_CHECK();
    return 0;
}

int thx_gather_pixels_dev(float* datP, const float* img, const int* iPxl, int nPxl, int idim, int nImg, void* stream)
{
    if (nImg <= 0 || nPxl <= 0) return 0;
    const size_t imgSize = (size_t)idim * (idim / 2 + 1);
    for (int l0 = 0; l0 < nImg; l0 += 65535) {
        const int nl = nImg - l0 < 65535 ? nImg - l0 : 65535;
        hipLaunchKernelGGL(k_gather_pixels, dim3((nPxl + 255) / 256, nl), dim3(256), 0, as_stream(stream),
                           reinterpret_cast<float2*>(datP) + (size_t)l0 * nPxl,
                           reinterpret_cast<const float2*>(img) + (size_t)l0 * imgSize, iPxl, nPxl, imgSize);
    }
    THX_LAUNCH_CHECK();
    return 0;
}

int thx_project_dev(const float* volume, float* rotP, const double* rotMat, const int* iCol, const int* iRow, int nR,
                    int pf, int vdim, int nPxl, void* stream)
{
    if (nR <= 0 || nPxl <= 0) return 0;
    THX_REQUIRE(volume && rotP && rotMat && iCol && iRow, "NULL pointer");
    for (int r0 = 0; r0 < nR; r0 += 65535) {
        const int nr = nR - r0 < 65535 ? nR - r0 : 65535;
        hipLaunchKernelGGL(k_project, dim3((nPxl + 255) / 256, nr), dim3(256), 0, as_stream(stream),
                           reinterpret_cast<const float2*>(volume), reinterpret_cast<float2*>(rotP) + (size_t)r0 * nPxl,
                           rotMat + (size_t)r0 * 9, iCol, iRow, pf, vdim, nPxl);
    }
    THX_LAUNCH_CHECK();
    return 0;
}

int thx_logdatavsprior_dev(float* out, const float* dat, const float* pri, const float* ctf, const float* sigRcp,
                           int nPri, int nPxl, void* stream)
{
    if (nPri <= 0) return 0;
    hipLaunchKernelGGL(k_logdvp, dim3((nPri + 3) / 4), dim3(256), 0, as_stream(stream), out,
                       reinterpret_cast<const float2*>(dat), reinterpret_cast<const float2*>(pri), ctf, sigRcp, nPri,
                       nPxl);
    THX_LAUNCH_CHECK();
    return 0;
}

size_t thx_expect_local_workspace(int nImg, int nR, int nT, int nD)
{
    const int nSplit = expect_local_nsplit(nImg);
    const size_t nRpad = (size_t)((nR + 63) / 64) * 64;
    return ((size_t)nImg * nD * nSplit * nT * nRpad + (size_t)nImg * nD * nSplit) * sizeof(float) + 256;
}

static int expect_local_impl(const float* volumes, const int* volIdx, int vdim, int pf, int idim, const int* iCol,
                             const int* iRow, int nPxl, int nImg, const float* datP, const float* ctfP,
                             const float* sigRcpP, const double* rotMat, int nR, const double* trans, int nT, int nD,
                             const double* pC, const double* pR, const double* pT, const double* pD, float* wC, float* wR,
                             float* wT, float* wD, float* baseLine, float* logW, void* workspace, int wgPerCU, const int* active, void* stream,
                             bool packed, int nSplitForce = 0, bool noOrder = false, double pCval = 1.0, bool fine = false,
                             unsigned* done = nullptr, unsigned doneVal = 0)
{
    if (nImg <= 0) return 0;
    THX_REQUIRE(volumes && iCol && iRow && datP && ctfP && sigRcpP && rotMat && trans, "NULL input pointer");
    THX_REQUIRE(pR && pT && pD && wC && wR && wT && wD && baseLine && workspace, "NULL prior/output/workspace pointer");
    THX_REQUIRE(nR > 0 && nT > 0 && nD > 0 && nPxl > 0, "nR, nT, nD, nPxl must be positive");
    THX_REQUIRE(nT <= 32, "nT > 32 translations per phase is not supported");
    THX_REQUIRE(nImg <= 65535 && nD <= 65535, "nImg and nD must be <= 65535 per call");
    THX_REQUIRE((size_t)nD * nT * nR * sizeof(float) <= 64 * 1024, "nD*nT*nR too large for the finalise kernel");
    hipStream_t st = as_stream(stream);
    ExpectLocalArgs a;
    a.volumes = reinterpret_cast<const float2*>(volumes);
    a.volIdx = volIdx;
    a.P = vdim; a.pf = pf; a.idim = idim;
    a.iCol = iCol; a.iRow = iRow; a.nPxl = nPxl; a.nImg = nImg;
    a.datP = reinterpret_cast<const float2*>(datP);
    a.ctfP = ctfP; a.sigRcpP = sigRcpP; a.rotMat = rotMat; a.nR = nR; a.trans = trans; a.nT = nT; a.nD = nD;
    a.nSplit = nSplitForce > 0 ? nSplitForce : expect_local_nsplit(nImg);
    a.active = active;
    a.splitM = 0.f;
    a.order = nullptr;
    a.fine = (fine && nD == 1) ? 1 : 0;
    if (knobs().expectOrder > 0 && nR > 64 && nR <= 256 && nD == 1 && !noOrder) {   // (one wave holds a cloud of <= 64 rotations whatever the order)
        unsigned char* ord = reinterpret_cast<unsigned char*>(scratch(st, 18, (size_t)nImg * nR));
        THX_REQUIRE(ord, "device scratch allocation failed");
        hipLaunchKernelGGL(k_cloud_order, dim3(nImg), dim3(256), 0, st, ord, rotMat, nR, knobs().expectOrder);
        THX_LAUNCH_CHECK();
        a.order = ord;
    }
    a.nRpad = ((nR + 63) / 64) * 64;
    a.partV = reinterpret_cast<float*>(workspace);
    a.partC = a.partV + (size_t)nImg * nD * a.nSplit * nT * a.nRpad;
    int rc;
    ExpectFinalArgs f;
    f.partV = a.partV; f.partC = a.partC; f.nSplit = a.nSplit;
    if (nT <= 9) rc = launch_expect_local<9>(a, st, packed, wgPerCU);
    else if (nT <= 16) rc = launch_expect_local<16>(a, st, packed, wgPerCU);
    else rc = launch_expect_local<32>(a, st, packed, wgPerCU);
    if (rc) return rc;
    if (a.fine && a.nSplit > 1) {   // (one image, nD = 1: [nSplit][nT][nRpad] partial sums -> split 0)
        const int nElem = nT * a.nRpad;
        hipLaunchKernelGGL(k_expect_reduce, dim3((nElem + 63) / 64), dim3(64 * kReduceWaves), 0, st, a.partV, a.partC, a.nSplit, nElem);
        f.nSplit = 1;
    }
    f.nR = nR; f.nRpad = a.nRpad; f.nT = nT; f.nD = nD;
    f.pC = pC; f.pCval = pCval; f.pR = pR; f.pT = pT; f.pD = pD; f.wC = wC; f.wR = wR; f.wT = wT; f.wD = wD; f.baseLine = baseLine;
    f.logW = logW;
    f.active = active;
    f.par = fine ? 1 : 0;
    f.done = done; f.doneVal = doneVal;
    hipLaunchKernelGGL(k_expect_final, dim3(nImg), dim3(256), (((size_t)nD * nT * nR * sizeof(float) + 7) & ~(size_t)7) + (size_t)(nR + nT + nD) * sizeof(double), st, f);
    THX_LAUNCH_CHECK();
    return 0;
}

int thx_expect_local_dev(const float* volumes, const int* volIdx, int vdim, int pf, int idim, const int* iCol,
                         const int* iRow, int nPxl, int nImg, const float* datP, const float* ctfP,
                         const float* sigRcpP, const double* rotMat, int nR, const double* trans, int nT, int nD,
                         const double* pC, const double* pR, const double* pT, const double* pD, float* wC, float* wR,
                         float* wT, float* wD, float* baseLine, float* logW, void* workspace, int wgPerCU, const int* active,
                         void* stream)
{
    return expect_local_impl(volumes, volIdx, vdim, pf, idim, iCol, iRow, nPxl, nImg, datP, ctfP, sigRcpP, rotMat, nR, trans, nT,
                             nD, pC, pR, pT, pD, wC, wR, wT, wD, baseLine, logW, workspace, wgPerCU, active, stream, false);
}

int thx_expect_local_packed_dev(const float* cells, const int* volIdx, int vdim, int pf, int idim, const int* iCol,
                                const int* iRow, int nPxl, int nImg, const float* datP, const float* ctfP,
                                const float* sigRcpP, const double* rotMat, int nR, const double* trans, int nT, int nD,
                                const double* pC, const double* pR, const double* pT, const double* pD, float* wC, float* wR,
                                float* wT, float* wD, float* baseLine, float* logW, void* workspace, int wgPerCU, const int* active,
                         void* stream)
{
    return expect_local_impl(cells, volIdx, vdim, pf, idim, iCol, iRow, nPxl, nImg, datP, ctfP, sigRcpP, rotMat, nR, trans, nT,
                             nD, pC, pR, pT, pD, wC, wR, wT, wD, baseLine, logW, workspace, wgPerCU, active, stream, true);
}

size_t thx_projector_packed_bytes(int vdim) { return (size_t)vdim * vdim * (vdim / 2 + 1) * 64; }

}  // extern "C"

// ONE image per call -- the granularity of the reference's plug-in surface (ExpectLocalM, gpu/interface/Interface.h:125-139, called
// image by image under a per-GPU lock, src/Optimiser.cpp:2960-3080).  The batched launch gives an image 1 ... 16 workgroups; alone on
// the chip it gets one per 256-pixel chunk (97 at 256^3), the lane <-> rotation ordering launch is left out (the results do not depend
// on it) and the class prior travels by value: two launches per image-phase.  Same kernels, same arithmetic; the partial sums of the
// chunks are added in chunk order (the batched form adds them in groups: equal to rounding, tests/test_iface_gpu.py).
namespace thx {
constexpr int kSinglePixelsPerWg = 32;   // the one-image form: pixels per workgroup (each of its waves then walks 8 - 16 of them; 64: 47 us per image at 256^3, measured)
static int single_nsplit(int nPxl, int nD)
{
    if (knobs().expectNSplit) return knobs().expectNSplit;
    const int per = nD == 1 ? kSinglePixelsPerWg : kChunk;     // (the fused defocus kernel shares its pixels in whole chunks)
    const int n = (nPxl + per - 1) / per;
    return n < 1 ? 1 : (n > 1024 ? 1024 : n);
}
size_t expect_local_single_workspace(int nPxl, int nR, int nT, int nD)
{
    const size_t nRpad = (size_t)((nR + 63) / 64) * 64, nSplit = (size_t)single_nsplit(nPxl, nD);
    return ((size_t)nD * nSplit * nT * nRpad + (size_t)nD * nSplit) * sizeof(float) + 256;
}
int expect_local_single(const float* volOrCells, bool packed, int vdim, int pf, int idim, const int* iCol, const int* iRow, int nPxl,
                        const float* datP, const float* ctfP, const float* sigRcpP, const double* rotMat, int nR, const double* trans,
                        int nT, int nD, double pC, const double* pR, const double* pT, const double* pD, float* wC, float* wR, float* wT,
                        float* wD, float* baseLine, void* workspace, hipStream_t st, unsigned* done, unsigned doneVal)
{
    return expect_local_impl(volOrCells, nullptr, vdim, pf, idim, iCol, iRow, nPxl, 1, datP, ctfP, sigRcpP, rotMat, nR, trans, nT, nD,
                             nullptr, pR, pT, pD, wC, wR, wT, wD, baseLine, nullptr, workspace, 0 /* no occupancy cap: one image */, nullptr, st,
                             packed, single_nsplit(nPxl, nD), true, pC, true, done, doneVal);
}
}  // namespace thx

extern "C" {

int thx_projector_pack_dev(float* cells, const float* volumes, int vdim, int nVol, void* stream)
{
    THX_REQUIRE(cells && volumes && vdim > 0 && nVol >= 0, "bad arguments");
    const size_t nCell = (size_t)vdim * vdim * (vdim / 2 + 1);
    for (int v = 0; v < nVol; v++) {
        hipLaunchKernelGGL(k_pack_cells, dim3((unsigned)((nCell * 4 + 255) / 256)), dim3(256), 0, as_stream(stream),
                           reinterpret_cast<float4*>(cells) + (size_t)v * nCell * 4,
                           reinterpret_cast<const float2*>(volumes) + (size_t)v * nCell, vdim);
    }
    THX_LAUNCH_CHECK();
    return 0;
}

size_t thx_expect_global_workspace(int nImg, int nR, int nT) { return (size_t)nImg * nR * nT * sizeof(float) + 256; }
// (the transposed slices live in the library's own per-stream scratch: nR * nPxl * 8 B)

int thx_expect_global_dev(const float* rotP, const float* traP, const float* datP, const float* ctfP,
                          const float* sigRcpP, const double* pR, const double* pT, float* wC, float* wR, float* wT,
                          float* baseL, int kIdx, int nK, int nR, int nT, int nPxl, int nImg, void* workspace,
                          void* stream)
{
    if (nImg <= 0) return 0;
    THX_REQUIRE(rotP && traP && datP && ctfP && sigRcpP && pR && pT && wC && wR && wT && baseL && workspace,
                "NULL pointer");
    THX_REQUIRE(nImg <= 65535, "nImg must be <= 65535 per call");
    hipStream_t st = as_stream(stream);
    ExpectGlobalArgs a;
    // production sizes: the LDS-tiled contraction (bit-identical sums); small problems: the rotation-per-thread kernel
    const bool tiled = (size_t)nImg * nT >= 256 && nR >= 256 && !knobs().scanSimple;
    const int tileSel = knobs().scanTile;
    const int bm = (tileSel == 42 || tileSel == 44) ? 256 : 128, bn = (tileSel == 24 || tileSel == 44) ? 256 : 128;
    // padded shapes of the tiled form: pixel rows to 16, rotation / (image, shift) / image columns to whole tiles
    const long P16 = ((nPxl + kGK - 1) / kGK) * kGK, Npad = tiled ? (((long)nR + 255) / 256) * 256 : nR;
    float2* rotPT = reinterpret_cast<float2*>(scratch(st, 4, (size_t)(tiled ? P16 + kGK : nPxl) * Npad * sizeof(float2)));
    THX_REQUIRE(rotPT, "device scratch allocation failed");
    if (tiled && P16 > nPxl) THX_CHECK(hipMemsetAsync(rotPT + (size_t)nPxl * Npad, 0, (size_t)(P16 - nPxl) * Npad * sizeof(float2), st));
    hipLaunchKernelGGL(k_transpose_c64, dim3((nPxl + 31) / 32, (nR + 31) / 32), dim3(256), 0, st, rotPT,
                       reinterpret_cast<const float2*>(rotP), nR, nPxl, Npad);
    a.rotPT = rotPT;
    a.traP = reinterpret_cast<const float2*>(traP);
    a.datP = reinterpret_cast<const float2*>(datP);
    a.ctfP = ctfP; a.sigRcpP = sigRcpP; a.nR = nR; a.nT = nT; a.nPxl = nPxl; a.nImg = nImg;
    a.dvp = reinterpret_cast<float*>(workspace);
    if (tiled) {
        const int Mrows = nImg * nT;
        const long Mpad = (((long)Mrows + 255) / 256) * 256, Ipad = (((long)nImg + 127) / 128) * 128;
        const int K = 2 * nPxl, Kpad = ((2 * nPxl + kGK - 1) / kGK) * kGK;
        const size_t nA = (size_t)(Kpad + kGK) * Mpad, nS = (size_t)(P16 + kGK) * Ipad, nQ = (size_t)(P16 + kGK) * Npad, nB = (size_t)nImg * nR;
        float* tabA = reinterpret_cast<float*>(scratch(st, 11, (nA + nS + nQ + nB + nImg + 64) * sizeof(float)));
        THX_REQUIRE(tabA, "device scratch allocation failed");
        float *tabS = tabA + nA, *q2 = tabS + nS, *accB = q2 + nQ, *tabC = accB + nB;
        if (Kpad > K) THX_CHECK(hipMemsetAsync(tabA + (size_t)K * Mpad, 0, (size_t)(Kpad - K) * Mpad * sizeof(float), st));
        if (P16 > nPxl) THX_CHECK(hipMemsetAsync(tabS + (size_t)nPxl * Ipad, 0, (size_t)(P16 - nPxl) * Ipad * sizeof(float), st));
        THX_REQUIRE((Mrows + 255) / 256 <= 65535, "too many (image, shift) rows for one launch");
        hipLaunchKernelGGL(k_scan_tables, dim3(nPxl, (Mrows + 255) / 256), dim3(256), 0, st, tabA, tabS, a, Mpad, Ipad);
        hipLaunchKernelGGL(k_scan_const, dim3(nImg), dim3(256), 0, st, tabC, a);
        hipLaunchKernelGGL(k_scan_q2, dim3((unsigned)(((size_t)P16 * Npad + 255) / 256)), dim3(256), 0, st, q2, rotPT, (size_t)P16 * Npad);
        {
            const int tM = (nImg + 127) / 128, tN = (nR + 127) / 128;
            const int nwg = ((((tM + 7) / 8) * ((tN + 7) / 8) + 7) / 8) * 8 * 64;   // patches rounded up to the 8 XCDs
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_gemm<false, 2, 2>), dim3(nwg), dim3(256), 0, st, accB, tabS, q2, nullptr, nullptr,
                               nImg, nR, (int)P16, Ipad, Npad, 1, tM, tN);
        }
        {
            auto launch = [&](auto kern) {
                const int tM = (Mrows + bm - 1) / bm, tN = (nR + bn - 1) / bn;
                const int nwg = ((((tM + 7) / 8) * ((tN + 7) / 8) + 7) / 8) * 8 * 64;
                hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), 0, st, a.dvp, tabA, rotPT, accB, tabC, Mrows, nR, Kpad, Mpad, Npad, nT, tM,
                                   tN);
            };
            switch (tileSel) {
            case 42: launch(k_scan_gemm<true, 4, 2>); break;
            case 24: launch(k_scan_gemm<true, 2, 4>); break;
            case 44: launch(k_scan_gemm<true, 4, 4>); break;
            default: launch(k_scan_gemm<true, 2, 2>); break;
            }
        }
    } else {
        constexpr int NT = 8;
        for (int t0 = 0; t0 < nT; t0 += NT) {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_expect_global<NT>), dim3((nR + 255) / 256, nImg), dim3(256), 0, st, a, t0);
        }
    }
    THX_LAUNCH_CHECK();
    if (nT <= 32)
        hipLaunchKernelGGL(k_expect_global_fold32, dim3(nImg), dim3(256), 0, st, a.dvp, pR, pT, wC, wR, wT, baseL, kIdx, nK, nR,
                           nT, nImg);
    else
        hipLaunchKernelGGL(k_expect_global_fold, dim3(nImg), dim3(256), 0, st, a.dvp, pR, pT, wC, wR, wT, baseL, kIdx, nK, nR,
                           nT, nImg);
    THX_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
