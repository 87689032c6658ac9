// thx_io.hip -- the data formats either side of the hot path:
//   * MRC stacks / volumes (ImageFile, src/Image/ImageFile.cpp:150-404, include/Image/MRCHeader.h) -- host code
//   * the .thu particle table (Database, include/Database.h:22-287, src/Database.cpp) -- host code
//   * image ingestion on the device (Optimiser::initImg, src/Optimiser.cpp:4608-4800): background normalisation,
//     image statistics, soft mask, 1/stdN scale, batched rocFFT 2-D r2c into the two HBM-resident image stacks.
#include <hipfft/hipfft.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "thx_common.h"

namespace thx {

// ---------------------------------------------------------------------------------------------
// MRC (host)
// ---------------------------------------------------------------------------------------------
struct MrcHeader {  // include/Image/MRCHeader.h: 1024 bytes
    int nx, ny, nz, mode, nxstart, nystart, nzstart, mx, my, mz;
    float cella[3], cellb[3];
    int mapc, mapr, maps;
    float dmin, dmax, dmean;
    int ispg, nsymbt;
    char extra[100];
    float origin[3];
    char map[4];
    int machst;
    float rms;
    int nlabels;
    char label[10][80];
};
static_assert(sizeof(MrcHeader) == 1024, "MRC header must be 1024 bytes");

static int byte_mode(int mode)  // BYTE_MODE, include/Image/ImageFile.h:96-110
{
    switch (mode) {
        case 0: return 1;
        case 1: return 2;
        case 2: return 4;
        default: return -1;  // modes 3, 4, 6 are not read by ImageFile::readImageMRC either (:258-263)
    }
}

struct File {
    FILE* f = nullptr;
    ~File() { if (f) fclose(f); }
};

static int read_header(File& fh, const char* path, MrcHeader& h)
{
    fh.f = fopen(path, "rb");
    if (!fh.f) { set_error("cannot open %s", path); return -2; }
    if (fread(&h, 1, 1024, fh.f) != 1024) { set_error("FAIL TO READ IN MRC HEADER FILE (%s)", path); return -3; }
    if (h.nx <= 0 || h.ny <= 0 || h.nz <= 0 || h.nsymbt < 0) { set_error("bad MRC header in %s", path); return -4; }
    if (byte_mode(h.mode) < 0) { set_error("unsupported MRC mode %d in %s", h.mode, path); return -5; }
    return 0;
}

// IMAGE_READ_CAST / VOLUME_READ_CAST: memory(i, j[, k]) = file[MESH_*_INDEX] (include/Image/ImageFile.h:383-388,418-470):
// the file keeps the origin at the centre, the in-memory layout at index 0 (wrapped).
template <typename T>
static void mesh_plane(float* dst, const T* src, int nx, int ny)
{
    for (int j = 0; j < ny; j++) {
        const T* srow = src + (size_t)((j + ny / 2) % ny) * nx;
        float* drow = dst + (size_t)j * nx;
        for (int i = 0; i < nx; i++) drow[i] = (float)srow[(i + nx / 2) % nx];
    }
}

static void fill_header(MrcHeader& h, int nx, int ny, int nz, float pixelSize, int dims)
{   // ImageFile::fillMRCHeader, src/Image/ImageFile.cpp:171-207
    memset(&h, 0, sizeof(h));
    h.mode = 2;
    h.nx = nx; h.ny = ny; h.nz = nz;
    h.mx = nx; h.my = ny; h.mz = nz;
    h.cella[0] = (float)nx; h.cella[1] = (float)ny; h.cella[2] = (float)nz;
    h.cellb[0] = h.cellb[1] = h.cellb[2] = 90;
    h.mapc = 1; h.mapr = 2; h.maps = 3;
    h.ispg = 1;
    memcpy(h.map, "MAP ", 4);
    for (int d = 0; d < dims; d++) h.cella[d] *= pixelSize;  // writeImageMRC scales 2, writeVolumeMRC / openStack 3 (:309-372)
}

// ---------------------------------------------------------------------------------------------
// ingestion kernels: one workgroup per image, fp64 accumulation, fixed summation order (deterministic)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum(double v, double* s)
{
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) t += s[w];
    return t;
}

// pixel classes of an image in wrapped layout: q = i^2 + j^2 (exact in double), u = (float)hypot(i, j)
__device__ __forceinline__ void pix_ij(int e, int N, int& i, int& j)
{
    const int iw = e % N, jw = e / N;
    i = iw >= N / 2 ? iw - N : iw;
    j = jw >= N / 2 ? jw - N : jw;
}

// substractBgImg (src/Optimiser.cpp:4928-4962): bgMeanStddev over q > r^2 (src/Image/ImageFunctions.cpp:607-621), then
// (x - mean) / stddev.  GSL's running long-double recurrences are replaced by two-pass fp64 sums.
__global__ __launch_bounds__(256) void k_subtract_bg(float* __restrict__ img, int N, float r2)
{
    __shared__ double s[4];
    float* p = img + (size_t)blockIdx.x * N * N;
    const int n = N * N;
    double sum = 0, cnt = 0;
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        int i, j;
        pix_ij(e, N, i, j);
        if ((double)i * i + (double)j * j > (double)r2) { sum += p[e]; cnt += 1; }
    }
    sum = block_sum(sum, s);
    cnt = block_sum(cnt, s);
    const float mean = (float)(sum / cnt);
    double ss = 0;
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        int i, j;
        pix_ij(e, N, i, j);
        if ((double)i * i + (double)j * j > (double)r2) { const double d = (double)p[e] - (double)mean; ss += d * d; }
    }
    ss = block_sum(ss, s);
    const float sd = (float)sqrt(ss / cnt * (cnt / (cnt - 1)));
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        float v = p[e];
        v -= mean;
        v /= sd;
        p[e] = v;
    }
}

// per-image terms of statImg (src/Optimiser.cpp:4838-4870): out[l] = {regionMean(img, r, 0), bgStddev(0, img, r),
// stddev(0, img), bgStddev^2}
__global__ __launch_bounds__(256) void k_stat_img(double* __restrict__ out, const float* __restrict__ img, int N, float r,
                                                  float r2)
{
    __shared__ double s[4];
    const float* p = img + (size_t)blockIdx.x * N * N;
    const int n = N * N;
    double sumIn = 0, cntIn = 0, ssBg = 0, cntBg = 0, ssAll = 0;
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        int i, j;
        pix_ij(e, N, i, j);
        const double v = p[e];
        const float u = (float)gsl_hypot_((double)i, (double)j);
        if (u < r && u >= 0.0f) { sumIn += v; cntIn += 1; }
        if ((double)i * i + (double)j * j > (double)r2) { ssBg += v * v; cntBg += 1; }
        ssAll += v * v;
    }
    sumIn = block_sum(sumIn, s); cntIn = block_sum(cntIn, s);
    ssBg = block_sum(ssBg, s); cntBg = block_sum(cntBg, s);
    ssAll = block_sum(ssAll, s);
    if (threadIdx.x == 0) {
        const float rm = (float)sumIn / (float)cntIn;
        const float b = (float)sqrt(ssBg / cntBg * (cntBg / (cntBg - 1)));
        const float d = (float)sqrt(ssAll / n * ((double)n / (double)(n - 1)));
        out[4 * (size_t)blockIdx.x + 0] = rm;
        out[4 * (size_t)blockIdx.x + 1] = b;
        out[4 * (size_t)blockIdx.x + 2] = d;
        out[4 * (size_t)blockIdx.x + 3] = (double)b * (double)b;
    }
}

// maskImg + normaliseImg (src/Optimiser.cpp:4964-5012): ori = x * scale, masked = softMask(x, bg = 0) * scale.
// mask[] holds the weight 1 - w of softMask(dst, src, r, ew, bg) (src/Functions/Mask.cpp:363-385), built on the host.
__global__ __launch_bounds__(256) void k_mask_scale(float* __restrict__ masked, float* __restrict__ ori,
                                                    const float* __restrict__ keep, const unsigned char* __restrict__ zone,
                                                    size_t nPerImg, size_t total, float scale)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const size_t p = e % nPerImg;
    const float x = ori[e];
    float m;
    const unsigned char z = zone[p];
    if (z == 0) m = x;                                   // u < r: dst = src
    else if (z == 2) m = 0.0f;                           // u > r + ew: dst = bg
    else m = x * keep[p];                                // bg * w + src * (1 - w) with bg = 0, keep = 1 - w
    masked[e] = m * scale;
    ori[e] = x * scale;
}

// one plan per (device, stream, shape): a hipFFT plan carries its stream and work area
static std::mutex g_planIoMtx;
static std::map<std::tuple<int, hipStream_t, int, int>, hipfftHandle> g_planIo;
void release_io_plans(int dev, hipStream_t st)
{
    std::lock_guard<std::mutex> g(g_planIoMtx);
    for (auto it = g_planIo.begin(); it != g_planIo.end();)
        if (std::get<0>(it->first) == dev && std::get<1>(it->first) == st) { (void)hipfftDestroy(it->second); it = g_planIo.erase(it); } else ++it;
}
static int cached_plan_r2c(hipfftHandle* out, int idim, int batch, hipStream_t st)
{
    std::mutex& mtx = g_planIoMtx;
    auto& cache = g_planIo;
    int dev = 0;
    THX_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> g(mtx);
    auto key = std::make_tuple(dev, st, idim, batch);
    auto it = cache.find(key);
    if (it == cache.end()) {
        hipfftHandle p;
        int n[2] = {idim, idim};
        if (hipfftPlanMany(&p, 2, n, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_R2C, batch) != HIPFFT_SUCCESS) {
            set_error("hipfftPlanMany(r2c %d x %d, batch %d) failed", idim, idim, batch);
            return 1001;
        }
        if (hipfftSetStream(p, st) != HIPFFT_SUCCESS) { set_error("hipfftSetStream failed"); return 1002; }
        it = cache.emplace(key, p).first;
    }
    *out = it->second;
    return 0;
}

}  // namespace thx

using namespace thx;

extern "C" {

int thx_mrc_info(const char* path, int* nx, int* ny, int* nz, int* mode, int* nsymbt)
{
    THX_REQUIRE(path, "path is NULL");
    File fh;
    MrcHeader h;
    int rc = read_header(fh, path, h);
    if (rc) return rc;
    if (nx) *nx = h.nx;
    if (ny) *ny = h.ny;
    if (nz) *nz = h.nz;
    if (mode) *mode = h.mode;
    if (nsymbt) *nsymbt = h.nsymbt;
    return 0;
}

int thx_mrc_read_images(const char* path, int first, int count, float* dst)
{
    THX_REQUIRE(path && dst && first >= 0 && count >= 0, "bad arguments");
    File fh;
    MrcHeader h;
    int rc = read_header(fh, path, h);
    if (rc) return rc;
    THX_REQUIRE(first + count <= h.nz, "slice range beyond the end of the stack");
    const size_t plane = (size_t)h.nx * h.ny;
    const int bm = byte_mode(h.mode);
    std::vector<char> buf(plane * bm);
    // SKIP_HEAD(size * iSlc * BYTE_MODE(mode)), include/Image/ImageFile.h:372-374
    if (fseek(fh.f, (long)(1024 + h.nsymbt + plane * (size_t)first * bm), SEEK_SET) != 0) {
        set_error("Fail to read in an image (%s)", path);
        return -6;
    }
    for (int s = 0; s < count; s++) {
        if (fread(buf.data(), bm, plane, fh.f) != plane) { set_error("Fail to read in an image (%s, slice %d)", path, first + s); return -6; }
        float* d = dst + (size_t)s * plane;
        if (h.mode == 0) mesh_plane(d, reinterpret_cast<const signed char*>(buf.data()), h.nx, h.ny);
        else if (h.mode == 1) mesh_plane(d, reinterpret_cast<const short*>(buf.data()), h.nx, h.ny);
        else mesh_plane(d, reinterpret_cast<const float*>(buf.data()), h.nx, h.ny);
    }
    return 0;
}

int thx_mrc_read_volume(const char* path, float* dst)
{
    THX_REQUIRE(path && dst, "bad arguments");
    File fh;
    MrcHeader h;
    int rc = read_header(fh, path, h);
    if (rc) return rc;
    const size_t plane = (size_t)h.nx * h.ny;
    const int bm = byte_mode(h.mode);
    std::vector<char> buf(plane * bm);
    if (fseek(fh.f, 1024 + h.nsymbt, SEEK_SET) != 0) { set_error("Fail to read in a volume (%s)", path); return -6; }
    for (int k = 0; k < h.nz; k++) {  // file slice k lands at memory slice (k - nz/2) mod nz (MESH_VOLUME_INDEX)
        if (fread(buf.data(), bm, plane, fh.f) != plane) { set_error("Fail to read in a volume (%s)", path); return -6; }
        float* d = dst + (size_t)((k + h.nz - h.nz / 2) % h.nz) * plane;
        if (h.mode == 0) mesh_plane(d, reinterpret_cast<const signed char*>(buf.data()), h.nx, h.ny);
        else if (h.mode == 1) mesh_plane(d, reinterpret_cast<const short*>(buf.data()), h.nx, h.ny);
        else mesh_plane(d, reinterpret_cast<const float*>(buf.data()), h.nx, h.ny);
    }
    return 0;
}

static int write_planes(const char* path, const float* src, int nx, int ny, int nz, float pixelSize, int dims, bool meshZ)
{
    File fh;
    fh.f = fopen(path, "wb");
    if (!fh.f) { set_error("cannot create %s", path); return -2; }
    MrcHeader h;
    fill_header(h, nx, ny, nz, pixelSize, dims);
    if (fwrite(&h, 1, 1024, fh.f) != 1024) { set_error("FAIL TO WRITE OUT THIS IMAGE (%s)", path); return -7; }
    const size_t plane = (size_t)nx * ny;
    std::vector<float> buf(plane);
    for (int k = 0; k < nz; k++) {
        // IMAGE_WRITE_CAST / VOLUME_WRITE_CAST: file(i, j, k) = memory[MESH_*_INDEX(i, j, k)] (include/Image/ImageFile.h:
        // 483-560) -- the same +n/2 rotation as on reading (its own inverse for even sizes)
        const float* s = src + (size_t)(meshZ ? (k + nz / 2) % nz : k) * plane;
        for (int j = 0; j < ny; j++) {
            const float* srow = s + (size_t)((j + ny / 2) % ny) * nx;
            float* drow = buf.data() + (size_t)j * nx;
            for (int i = 0; i < nx; i++) drow[i] = srow[(i + nx / 2) % nx];
        }
        if (fwrite(buf.data(), sizeof(float), plane, fh.f) != plane) { set_error("FAIL TO WRITE OUT THIS IMAGE (%s)", path); return -7; }
    }
    return 0;
}

int thx_mrc_write_volume(const char* path, const float* src, int nx, int ny, int nz, float pixelSize)
{
    THX_REQUIRE(path && src && nx > 0 && ny > 0 && nz > 0, "bad arguments");
    return write_planes(path, src, nx, ny, nz, pixelSize, 3, true);
}

int thx_mrc_write_stack(const char* path, const float* src, int size, int nSlc, float pixelSize)
{
    THX_REQUIRE(path && src && size > 0 && nSlc > 0, "bad arguments");
    return write_planes(path, src, size, size, nSlc, pixelSize, 3, false);
}

// ---------------------------------------------------------------------------------------------
// .thu table (host): one particle per line, blank-separated columns in the order of include/Database.h:22-287;
// blank lines and lines whose first non-blank character is '#' are skipped (Database::reGenDatabase, src/Database.cpp:40-100)
// ---------------------------------------------------------------------------------------------
static bool thu_data_line(const char* s)
{
    for (; *s; s++) {
        if (*s == ' ' || *s == '\t' || *s == '\n' || *s == '\r') continue;
        return *s != '#';
    }
    return false;
}

int thx_thu_count(const char* path, int* nParticle, int* nGroup)
{
    THX_REQUIRE(path && nParticle, "bad arguments");
    File fh;
    fh.f = fopen(path, "r");
    if (!fh.f) { set_error("FAIL TO OPEN DATABASE (%s)", path); return -2; }
    std::vector<char> line(1 << 16);
    int n = 0, g = 0;
    while (fgets(line.data(), (int)line.size() - 1, fh.f)) {
        if (!thu_data_line(line.data())) continue;
        n++;
        int col = 0;
        for (char* w = strtok(line.data(), " \t\r\n"); w; w = strtok(nullptr, " \t\r\n"), col++)
            if (col == 11) { const int v = atoi(w); if (v > g) g = v; }  // THU_GROUP_ID, Database::nGroup :152-180
    }
    *nParticle = n;
    if (nGroup) *nGroup = g;
    return 0;
}

int thx_thu_load(const char* path, int nParticle, thx_ctf_attr* ctf, char* particlePath, int pathStride, int* groupID,
                 int* classID, double* quat, double* tran, double* stdT, double* defocusFactor, double* score)
{
    THX_REQUIRE(path && nParticle >= 0, "bad arguments");
    THX_REQUIRE(!particlePath || pathStride > 1, "pathStride must hold at least one character + NUL");
    File fh;
    fh.f = fopen(path, "r");
    if (!fh.f) { set_error("FAIL TO OPEN DATABASE (%s)", path); return -2; }
    std::vector<char> line(1 << 16);
    int l = 0;
    while (l < nParticle && fgets(line.data(), (int)line.size() - 1, fh.f)) {
        if (!thu_data_line(line.data())) continue;
        double v[27];
        for (int c = 0; c < 27; c++) v[c] = 0;
        v[13] = 1.0;   // identity quaternion, unit defocus factor when the columns are absent (short tables)
        v[24] = 1.0;
        std::string ppath;
        int col = 0;
        for (char* w = strtok(line.data(), " \t\r\n"); w && col < 27; w = strtok(nullptr, " \t\r\n"), col++) {
            if (col == 7) ppath = w;
            else if (col != 8) v[col] = atof(w);  // Database::ctf / quat / tran ...: atof / atoi per column
        }
        if (col < 8) { set_error("line %d of %s has only %d columns (need the CTF columns and the particle path)", l + 1, path, col); return -8; }
        if (ctf) {
            ctf[l].voltage = (float)v[0]; ctf[l].defocusU = (float)v[1]; ctf[l].defocusV = (float)v[2];
            ctf[l].defocusTheta = (float)v[3]; ctf[l].Cs = (float)v[4]; ctf[l].amplitudeContrast = (float)v[5];
            ctf[l].phaseShift = (float)v[6];
        }
        if (particlePath) {
            strncpy(particlePath + (size_t)l * pathStride, ppath.c_str(), pathStride - 1);
            particlePath[(size_t)l * pathStride + pathStride - 1] = 0;
        }
        if (groupID) groupID[l] = (int)v[11];
        if (classID) classID[l] = (int)v[12];
        if (quat) for (int c = 0; c < 4; c++) quat[4 * (size_t)l + c] = v[13 + c];
        if (tran) { tran[2 * (size_t)l] = v[20]; tran[2 * (size_t)l + 1] = v[21]; }
        if (stdT) { stdT[2 * (size_t)l] = v[22]; stdT[2 * (size_t)l + 1] = v[23]; }
        if (defocusFactor) defocusFactor[l] = v[24];
        if (score) score[l] = v[26];
        l++;
    }
    if (l != nParticle) { set_error("%s holds %d particles, %d requested", path, l, nParticle); return -9; }
    return 0;
}

// Optimiser::saveDatabase (src/Optimiser.cpp:8251-8416) + writeDescInfo (:8217-8249): the 27-column table the next run
// (or the reference) reads back.  The reference's ranks append in turn (rank 1 opens "w", the others "a", every rank
// writes the '#' description block); `append` selects the mode.  Column formats and order as the reference's fprintf.
int thx_thu_write(const char* path, int append, int nParticle, const thx_ctf_attr* ctf, const char* particlePath, int pathStride,
                  const char* micrographPath, int micStride, const double* coordXY, const int* groupID, const int* classID,
                  const double* quat, const double* k123, const double* tran, const double* stdT, const double* defocusFactor,
                  const double* stdDefocus, const double* score)
{
    THX_REQUIRE(path && nParticle >= 0 && ctf && particlePath && pathStride > 1, "bad arguments");
    File fh;
    fh.f = fopen(path, append ? "a" : "w");
    if (!fh.f) { set_error("FAIL TO OPEN %s FOR WRITING", path); return -2; }
    FILE* file = fh.f;
    static const char* desc[27] = {"#0:VOLTAGE\tFLOAT\t18.9f", "#1:DEFOCUS_U\tFLOAT\t18.9f", "#2:DEFOCUS_V\tFLOAT\t18.9f",
                                   "#3:DEFOCUS_THETA\tFLOAT\t18.9f", "#4:CS\tFLOAT\t18.9f", "#5:AMPLITUTDE_CONTRAST\tFLOAT\t18.9f",
                                   "#6:PHASE_SHIFT\tFLOAT\t18.9f", "#7:PARTICLE_PATH\tSTRING", "#8:MICROGRAPH_PATH\tSTRING",
                                   "#9:COORDINATE_X\tFLOAT\t18.9f", "#10:COORDINATE_Y\tFLOAT\t18.9f", "#11:GROUP_ID\tINT\t6d",
                                   "#12:CLASS_ID\tINT\t6d", "#13QUATERNION_0\tFLOAT\t18.9f", "#14:QUATERNION_1\tFLOAT\t18.9f",
                                   "#15:QUATERNION_2\tFLOAT\t18.9f", "#16:QUATERNION_3\tFLOAT\t18.9f", "#17:K1\tFLOAT\t18.9f",
                                   "#18:K2\tFLOAT\t18.9f", "#19:K3\tFLOAT\t18.9f", "#20:TRANSLATION_X\tFLOAT\t18.9f",
                                   "#21:TRANSLATION_Y\tFLOAT\t18.9f", "#22:STD_TRANSLATION_X\tFLOAT\t18.9f",
                                   "#23:STD_TRANSLATION_Y\tFLOAT\t18.9f", "#24:DEFOCUS_FACTOR\tFLOAT\t18.9f",
                                   "#25:STD_DEFOCUS_FACTOR\tFLOAT\t18.9f", "#26:SCORE\tFLOAT\t18.9f\n"};
    for (int c = 0; c < 27; c++) fprintf(file, "%s\n", desc[c]);
    const char* pad = "                     ";   // the reference's format string continues its lines inside the literal
    for (int l = 0; l < nParticle; l++) {
        const thx_ctf_attr& a = ctf[l];
        const char* mic = micrographPath ? micrographPath + (size_t)l * micStride : "mic.mrc";
        fprintf(file, "%18.9lf %18.9lf %18.9lf %18.9lf %18.9lf %18.9lf %18.9lf %s%s %s %18.9lf %18.9lf %s%6d %6lu %s"
                      "%18.9lf %18.9lf %18.9lf %18.9lf %s%18.9lf %18.9lf %18.9lf %s%18.9lf %18.9lf %18.9lf %18.9lf %s%18.9lf %18.9lf %s"
                      "%18.9lf\n",
                (double)a.voltage, (double)a.defocusU, (double)a.defocusV, (double)a.defocusTheta, (double)a.Cs,
                (double)a.amplitudeContrast, (double)a.phaseShift, pad, particlePath + (size_t)l * pathStride, mic,
                coordXY ? coordXY[2 * (size_t)l] : 0.0, coordXY ? coordXY[2 * (size_t)l + 1] : 0.0, pad, groupID ? groupID[l] : 1,
                (unsigned long)(classID ? classID[l] : 0), pad, quat ? quat[4 * (size_t)l] : 1.0, quat ? quat[4 * (size_t)l + 1] : 0.0,
                quat ? quat[4 * (size_t)l + 2] : 0.0, quat ? quat[4 * (size_t)l + 3] : 0.0, pad, k123 ? k123[3 * (size_t)l] : 1.0,
                k123 ? k123[3 * (size_t)l + 1] : 1.0, k123 ? k123[3 * (size_t)l + 2] : 1.0, pad, tran ? tran[2 * (size_t)l] : 0.0,
                tran ? tran[2 * (size_t)l + 1] : 0.0, stdT ? stdT[2 * (size_t)l] : 0.0, stdT ? stdT[2 * (size_t)l + 1] : 0.0, pad,
                defocusFactor ? defocusFactor[l] : 1.0, stdDefocus ? stdDefocus[l] : 0.0, pad, score ? score[l] : 0.0);
    }
    if (ferror(file)) { set_error("write error on %s", path); return -3; }
    return 0;
}

// the columns thx_thu_load does not return: micrograph path, coordinates, K1..K3, sd of the defocus factor
int thx_thu_load_extra(const char* path, int nParticle, char* micrographPath, int micStride, double* coordXY, double* k123,
                       double* stdDefocus)
{
    THX_REQUIRE(path && nParticle >= 0, "bad arguments");
    File fh;
    fh.f = fopen(path, "r");
    if (!fh.f) { set_error("FAIL TO OPEN DATABASE (%s)", path); return -2; }
    std::vector<char> line(1 << 16);
    int l = 0;
    while (l < nParticle && fgets(line.data(), (int)line.size() - 1, fh.f)) {
        if (!thu_data_line(line.data())) continue;
        double v[27];
        for (int c = 0; c < 27; c++) v[c] = 0;
        std::string mic;
        int col = 0;
        for (char* w = strtok(line.data(), " \t\r\n"); w && col < 27; w = strtok(nullptr, " \t\r\n"), col++) {
            if (col == 8) mic = w;
            else if (col != 7) v[col] = atof(w);
        }
        if (micrographPath && micStride > 1) {
            strncpy(micrographPath + (size_t)l * micStride, mic.c_str(), micStride - 1);
            micrographPath[(size_t)l * micStride + micStride - 1] = 0;
        }
        if (coordXY) { coordXY[2 * (size_t)l] = v[9]; coordXY[2 * (size_t)l + 1] = v[10]; }
        if (k123) for (int c = 0; c < 3; c++) k123[3 * (size_t)l + c] = v[17 + c];
        if (stdDefocus) stdDefocus[l] = v[25];
        l++;
    }
    if (l != nParticle) { set_error("%s holds %d particles, %d requested", path, l, nParticle); return -9; }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// ingestion on the device
// ---------------------------------------------------------------------------------------------
int thx_img_subtract_bg_dev(float* imgRL, int nImg, int idim, float maskRadiusPx, void* stream)
{
    if (nImg <= 0) return 0;
    THX_REQUIRE(imgRL && idim > 0 && (idim % 2) == 0, "bad arguments");
    hipLaunchKernelGGL(k_subtract_bg, dim3(nImg), dim3(256), 0, as_stream(stream), imgRL, idim, pow2f_(maskRadiusPx));
    THX_LAUNCH_CHECK();
    return 0;
}

int thx_img_stats_dev(double* stat, const float* imgRL, int nImg, int idim, float maskRadiusPx, void* stream)
{
    if (nImg <= 0) return 0;
    THX_REQUIRE(stat && imgRL && idim > 0, "bad arguments");
    hipLaunchKernelGGL(k_stat_img, dim3(nImg), dim3(256), 0, as_stream(stream), stat, imgRL, idim, maskRadiusPx,
                       pow2f_(maskRadiusPx));
    THX_LAUNCH_CHECK();
    return 0;
}

int thx_img_mask_normalise_fft_dev(float* imgFT, float* imgOriFT, float* imgRL, float* scratchRL, int nImg, int idim,
                                   float maskRadiusPx, float ew, float scale, void* stream)
{
    if (nImg <= 0) return 0;
    THX_REQUIRE(imgFT && imgOriFT && imgRL && scratchRL && idim > 0 && (idim % 2) == 0, "bad arguments");
    hipStream_t st = as_stream(stream);
    // mask zones / weights, built on the host as the reference's CPU path does (glibc cos in double), cached per call site
    static std::mutex mtx;
    static std::map<std::tuple<int, int, float, float>, std::pair<float*, unsigned char*>> cache;
    int dev = 0;
    THX_CHECK(hipGetDevice(&dev));
    float* keepDev = nullptr;
    unsigned char* zoneDev = nullptr;
    {
        std::lock_guard<std::mutex> g(mtx);
        auto key = std::make_tuple(dev, idim, maskRadiusPx, ew);
        auto it = cache.find(key);
        if (it == cache.end()) {
            const size_t n = (size_t)idim * idim;
            std::vector<float> keep(n);
            std::vector<unsigned char> zone(n);
            for (long j = -idim / 2; j < idim / 2; j++)
                for (long i = -idim / 2; i < idim / 2; i++) {
                    const float u = (float)gsl_hypot_((double)i, (double)j);
                    const size_t idx = (size_t)(j >= 0 ? j : j + idim) * idim + (size_t)(i >= 0 ? i : i + idim);
                    if (u > maskRadiusPx + ew) { zone[idx] = 2; keep[idx] = 0; }
                    else if (u >= maskRadiusPx) {
                        const float w = (float)(0.5 - 0.5 * cos((u - maskRadiusPx) / ew * 3.14159265358979323846));
                        zone[idx] = 1; keep[idx] = 1 - w;
                    } else { zone[idx] = 0; keep[idx] = 1; }
                }
            THX_CHECK(hipMalloc(&keepDev, n * sizeof(float)));
            THX_CHECK(hipMalloc(&zoneDev, n));
            THX_CHECK(hipMemcpy(keepDev, keep.data(), n * sizeof(float), hipMemcpyHostToDevice));
            THX_CHECK(hipMemcpy(zoneDev, zone.data(), n, hipMemcpyHostToDevice));
            it = cache.emplace(key, std::make_pair(keepDev, zoneDev)).first;
        }
        keepDev = it->second.first;
        zoneDev = it->second.second;
    }
    const size_t nPer = (size_t)idim * idim, nFT = (size_t)idim * (idim / 2 + 1) * 2;
    const int kBatch = 1024;
    for (int b = 0; b < nImg; b += kBatch) {
        const int nb = nImg - b < kBatch ? nImg - b : kBatch;
        float* rl = imgRL + (size_t)b * nPer;
        const size_t total = (size_t)nb * nPer;
        hipLaunchKernelGGL(k_mask_scale, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, scratchRL, rl, keepDev, zoneDev,
                           nPer, total, scale);
        THX_LAUNCH_CHECK();
        hipfftHandle plan;
        int rc = cached_plan_r2c(&plan, idim, nb, st);
        if (rc) return rc;
        if (hipfftExecR2C(plan, scratchRL, reinterpret_cast<hipfftComplex*>(imgFT + (size_t)b * nFT)) != HIPFFT_SUCCESS ||
            hipfftExecR2C(plan, rl, reinterpret_cast<hipfftComplex*>(imgOriFT + (size_t)b * nFT)) != HIPFFT_SUCCESS) {
            set_error("hipfftExecR2C failed");
            return 1003;
        }
    }
    return 0;
}

}  // extern "C"
