// thx_comm.hip -- the half-set exchange in native code: RCCL communicators and the reduction of the Fourier accumulators.
// Reference behaviour: gpu/src/cuthunder.cu:4192-4206 (ncclGetUniqueId on the hemisphere's root, MPI_Bcast of the id,
// ncclCommInitRank), :4972-5067 (ncclAllReduce of F, T, O, counter over the hemisphere), and the CPU path
// Reconstructor::allReduceF / allReduceT (src/Reconstructor.cpp:2350-2484, MPI_Allreduce_Large over _hemi).
// One process per GPU; xGMI is point-to-point, so the reduce is ONE large ring all-reduce per iteration and half:
// F (re, im) and T of the voxels inside the sample sphere only, packed into one contiguous buffer (50 % of the grid).
#include <rccl/rccl.h>

#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "thx_common.h"

struct thx_comm {
    ncclComm_t c = nullptr;
    int rank = 0, size = 1, dev = 0;
};

namespace thx {

#define THX_NCCL_CHECK(expr)                                                                                   \
    do {                                                                                                       \
        ncclResult_t _r = (expr);                                                                              \
        if (_r != ncclSuccess) {                                                                               \
            thx::set_error("%s failed: %s (%s:%d)", #expr, ncclGetErrorString(_r), __FILE__, __LINE__);        \
            return 2000 + (int)_r;                                                                             \
        }                                                                                                      \
    } while (0)

// rows (k, j) of the [P][P][P/2+1] half grid cut to the sample sphere: row r = kw * P + jw holds voxels i in [0, len(r));
// rowOff[r] = number of sphere voxels before row r (rowOff[P*P] = total)
struct SphereRows {
    long* rowOff = nullptr;   // device [P*P + 1]
    long total = 0;
};

static int sphere_rows(SphereRows* out, int P, int R)
{
    static std::mutex mtx;
    static std::map<std::tuple<int, int, int>, SphereRows> cache;
    int dev = 0;
    THX_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> g(mtx);
    auto key = std::make_tuple(dev, P, R);
    auto it = cache.find(key);
    if (it == cache.end()) {
        std::vector<long> off((size_t)P * P + 1);
        long run = 0;
        const long R2 = (long)R * R;
        for (int kw = 0; kw < P; kw++) {
            const long k = kw >= P / 2 ? kw - P : kw;
            for (int jw = 0; jw < P; jw++) {
                const long j = jw >= P / 2 ? jw - P : jw;
                off[(size_t)kw * P + jw] = run;
                const long rem = R2 - k * k - j * j;
                if (rem >= 0) {
                    long len = (long)floor(sqrt((double)rem)) + 1;
                    if (len > P / 2 + 1) len = P / 2 + 1;
                    run += len;
                }
            }
        }
        off[(size_t)P * P] = run;
        SphereRows s;
        s.total = run;
        THX_CHECK(hipMalloc(reinterpret_cast<void**>(&s.rowOff), off.size() * sizeof(long)));
        THX_CHECK(hipMemcpy(s.rowOff, off.data(), off.size() * sizeof(long), hipMemcpyHostToDevice));
        it = cache.emplace(key, s).first;
    }
    *out = it->second;
    return 0;
}

// pack (DIR = 0) / unpack (DIR = 1) the sphere voxels of F (complex) and T (real) into buf = [2 total | total] floats.
// One wave per row: coalesced on both sides.  grid (ceil(P*P / 4)), block 256.
template <int DIR>
__global__ __launch_bounds__(256) void k_sphere_pack(float2* __restrict__ F, float* __restrict__ T, float* __restrict__ buf,
                                                     const long* __restrict__ rowOff, int P, long total)
{
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)P * P) return;
    const int lane = threadIdx.x & 63;
    const long o = rowOff[row];
    const int len = (int)(rowOff[row + 1] - o);
    float2* bF = reinterpret_cast<float2*>(buf) + o;
    float* bT = buf + 2 * total + o;
    float2* f = F + row * (P / 2 + 1);
    float* t = T + row * (P / 2 + 1);
    for (int i = lane; i < len; i += 64) {
        if (DIR == 0) { bF[i] = f[i]; bT[i] = t[i]; }
        else { f[i] = bF[i]; t[i] = bT[i]; }
    }
}

// the same for the 64-bit fixed-point accumulators of the insertion (acc = [F re, im per voxel | T per voxel], thx_mstep.hip):
// buf = [2 total | total] long long
template <int DIR>
__global__ __launch_bounds__(256) void k_sphere_pack_acc(long long* __restrict__ accF, long long* __restrict__ accT,
                                                         long long* __restrict__ buf, const long* __restrict__ rowOff, int P, long total)
{
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)P * P) return;
    const int lane = threadIdx.x & 63;
    const long o = rowOff[row];
    const int len = (int)(rowOff[row + 1] - o);
    longlong2* bF = reinterpret_cast<longlong2*>(buf) + o;
    long long* bT = buf + 2 * total + o;
    longlong2* f = reinterpret_cast<longlong2*>(accF) + row * (P / 2 + 1);
    long long* t = accT + row * (P / 2 + 1);
    for (int i = lane; i < len; i += 64) {
        if (DIR == 0) { bF[i] = f[i]; bT[i] = t[i]; }
        else { f[i] = bF[i]; t[i] = bT[i]; }
    }
}

}  // namespace thx

using namespace thx;

extern "C" {

int thx_comm_unique_id(void* id128)
{
    THX_REQUIRE(id128, "id128 is NULL");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    THX_NCCL_CHECK(ncclGetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return 0;
}

int thx_comm_init(thx_comm** out, const void* id128, int rank, int size)
{
    THX_REQUIRE(out && id128 && size >= 1 && rank >= 0 && rank < size, "bad arguments");
    thx_comm* c = new thx_comm;
    c->rank = rank;
    c->size = size;
    if (hipGetDevice(&c->dev) != hipSuccess) { delete c; set_error("hipGetDevice failed"); return -1; }
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclResult_t r = ncclCommInitRank(&c->c, size, id, rank);
    if (r != ncclSuccess) {
        set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, size, ncclGetErrorString(r));
        delete c;
        return 2000 + (int)r;
    }
    *out = c;
    return 0;
}

int thx_comm_destroy(thx_comm* c)
{
    if (!c) return 0;
    if (c->c) (void)ncclCommDestroy(c->c);
    delete c;
    return 0;
}

int thx_comm_rank(const thx_comm* c) { return c ? c->rank : 0; }
int thx_comm_size(const thx_comm* c) { return c ? c->size : 1; }

int thx_comm_allreduce_f32(thx_comm* c, float* buf, size_t count, void* stream)
{
    if (!c || (c->size == 1 && !knobs().commForce) || count == 0) return 0;
    THX_NCCL_CHECK(ncclAllReduce(buf, buf, count, ncclFloat, ncclSum, c->c, as_stream(stream)));
    return 0;
}

int thx_comm_allreduce_f64(thx_comm* c, double* buf, size_t count, void* stream)
{
    if (!c || (c->size == 1 && !knobs().commForce) || count == 0) return 0;
    THX_NCCL_CHECK(ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, c->c, as_stream(stream)));
    return 0;
}

int thx_comm_allreduce_i32(thx_comm* c, int* buf, size_t count, void* stream)
{
    if (!c || (c->size == 1 && !knobs().commForce) || count == 0) return 0;
    THX_NCCL_CHECK(ncclAllReduce(buf, buf, count, ncclInt32, ncclSum, c->c, as_stream(stream)));
    return 0;
}

int thx_comm_allreduce_max_f64(thx_comm* c, double* buf, size_t count, void* stream)
{
    if (!c || (c->size == 1 && !knobs().commForce) || count == 0) return 0;
    THX_NCCL_CHECK(ncclAllReduce(buf, buf, count, ncclDouble, ncclMax, c->c, as_stream(stream)));
    return 0;
}

int thx_comm_broadcast(thx_comm* c, void* buf, size_t bytes, int root, void* stream)
{
    if (!c || (c->size == 1 && !knobs().commForce) || bytes == 0) return 0;
    THX_REQUIRE(root >= 0 && root < c->size, "root out of range");
    THX_NCCL_CHECK(ncclBroadcast(buf, buf, bytes, ncclChar, root, c->c, as_stream(stream)));
    return 0;
}

size_t thx_reco_allreduce_workspace(int dim, int maxRadius, int pf)
{
    // sphere voxels x 3 floats; an upper bound that needs no device: the cylinder of radius R over the half grid
    const double R = (double)maxRadius * pf + 2;
    const double vox = 3.14159265358979 * R * R * (dim / 2 + 1) + 4.0 * dim * dim;
    const double grid = (double)dim * dim * (dim / 2 + 1);
    return (size_t)((vox < grid ? vox : grid) * 3 * sizeof(float)) + 256;
}

int thx_reco_allreduce(thx_comm* hemi, float* F, float* T, double* O, int* counter, int dim, int maxRadius, int pf,
                       void* workspace, void* stream)
{
    if (!hemi || (hemi->size == 1 && !knobs().commForce)) return 0;   // the half lives on one rank: nothing to exchange
    THX_REQUIRE(F && T && workspace && dim > 0 && maxRadius > 0 && pf > 0, "bad arguments");
    hipStream_t st = as_stream(stream);
    SphereRows sr;
    const int R = maxRadius * pf + 2;   // inserted samples lie inside maxRadius * pf; their trilinear cells reach + 1
    THX_RC(sphere_rows(&sr, dim, R));
    THX_REQUIRE((size_t)sr.total * 3 * sizeof(float) <= thx_reco_allreduce_workspace(dim, maxRadius, pf), "workspace too small");
    float* buf = reinterpret_cast<float*>(workspace);
    const unsigned blocks = (unsigned)(((long)dim * dim + 3) / 4);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sphere_pack<0>), dim3(blocks), dim3(256), 0, st, reinterpret_cast<float2*>(F), T, buf,
                       sr.rowOff, dim, sr.total);
    THX_LAUNCH_CHECK();
    THX_NCCL_CHECK(ncclAllReduce(buf, buf, (size_t)sr.total * 3, ncclFloat, ncclSum, hemi->c, st));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sphere_pack<1>), dim3(blocks), dim3(256), 0, st, reinterpret_cast<float2*>(F), T, buf,
                       sr.rowOff, dim, sr.total);
    THX_LAUNCH_CHECK();
    if (O) THX_NCCL_CHECK(ncclAllReduce(O, O, 3, ncclDouble, ncclSum, hemi->c, st));
    if (counter) THX_NCCL_CHECK(ncclAllReduce(counter, counter, 1, ncclInt32, ncclSum, hemi->c, st));
    return 0;
}

size_t thx_reco_allreduce_acc_workspace(int dim, int maxRadius, int pf)
{
    return 2 * thx_reco_allreduce_workspace(dim, maxRadius, pf);   // 3 x 8 bytes per sphere voxel
}

// The half-set reduce BEFORE the accumulators become floats: every rank of the half has accumulated in the same quanta
// (thx_insert_scale_dev takes the extrema over the hemisphere), integer addition is associative, so the sums of N ranks are
// bit for bit what one rank would have accumulated over all the particles -- summing the converted floats
// (thx_reco_allreduce) depends on how the particles were dealt to the ranks.  One ring all-reduce of the sphere rows
// (ncclInt64: 24 bytes per voxel instead of 12).
int thx_reco_allreduce_acc_class(thx_comm* hemi, void* acc, int nK, int k, double* O, int* counter, int dim, int maxRadius, int pf,
                                 void* workspace, void* stream);

int thx_reco_allreduce_acc(thx_comm* hemi, void* acc, double* O, int* counter, int dim, int maxRadius, int pf, void* workspace,
                           void* stream)
{
    return thx_reco_allreduce_acc_class(hemi, acc, 1, 0, O, counter, dim, maxRadius, pf, workspace, stream);
}

// class k of a session over nK classes (acc = [nK][vol][2] F | [nK][vol] T, thx_insert_acc_bytes(dim, nK)): the classification
// driver reduces its K pairs one after the other through the same workspace
int thx_reco_allreduce_acc_class(thx_comm* hemi, void* acc, int nK, int k, double* O, int* counter, int dim, int maxRadius, int pf,
                                 void* workspace, void* stream)
{
    if (!hemi || (hemi->size == 1 && !knobs().commForce)) return 0;
    THX_REQUIRE(acc && workspace && dim > 0 && maxRadius > 0 && pf > 0 && nK >= 1 && k >= 0 && k < nK, "bad arguments");
    hipStream_t st = as_stream(stream);
    SphereRows sr;
    const int R = maxRadius * pf + 2;
    THX_RC(sphere_rows(&sr, dim, R));
    THX_REQUIRE((size_t)sr.total * 3 * sizeof(long long) <= thx_reco_allreduce_acc_workspace(dim, maxRadius, pf), "workspace too small");
    const size_t volN = (size_t)dim * dim * (dim / 2 + 1);
    long long* accF = reinterpret_cast<long long*>(acc) + (size_t)k * 2 * volN;
    long long* accT = reinterpret_cast<long long*>(acc) + (size_t)nK * 2 * volN + (size_t)k * volN;
    long long* buf = reinterpret_cast<long long*>(workspace);
    const unsigned blocks = (unsigned)(((long)dim * dim + 3) / 4);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sphere_pack_acc<0>), dim3(blocks), dim3(256), 0, st, accF, accT, buf, sr.rowOff, dim, sr.total);
    THX_LAUNCH_CHECK();
    THX_NCCL_CHECK(ncclAllReduce(buf, buf, (size_t)sr.total * 3, ncclInt64, ncclSum, hemi->c, st));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sphere_pack_acc<1>), dim3(blocks), dim3(256), 0, st, accF, accT, buf, sr.rowOff, dim, sr.total);
    THX_LAUNCH_CHECK();
    if (O) THX_NCCL_CHECK(ncclAllReduce(O, O, 3, ncclDouble, ncclSum, hemi->c, st));
    if (counter) THX_NCCL_CHECK(ncclAllReduce(counter, counter, 1, ncclInt32, ncclSum, hemi->c, st));
    return 0;
}

// the pack / unpack pair without a communicator (parity test of the sphere-row tables on one GPU): unpack(pack(F, T))
// must leave every voxel inside the sphere unchanged and touch nothing outside
int thx_reco_sphere_pack_dev(float* F, float* T, int dim, int maxRadius, int pf, void* workspace, int unpack, long* nVoxOut,
                             void* stream)
{
    THX_REQUIRE(F && T && workspace, "NULL pointer");
    SphereRows sr;
    THX_RC(sphere_rows(&sr, dim, maxRadius * pf + 2));
    const unsigned blocks = (unsigned)(((long)dim * dim + 3) / 4);
    if (unpack)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sphere_pack<1>), dim3(blocks), dim3(256), 0, as_stream(stream),
                           reinterpret_cast<float2*>(F), T, reinterpret_cast<float*>(workspace), sr.rowOff, dim, sr.total);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sphere_pack<0>), dim3(blocks), dim3(256), 0, as_stream(stream),
                           reinterpret_cast<float2*>(F), T, reinterpret_cast<float*>(workspace), sr.rowOff, dim, sr.total);
    THX_LAUNCH_CHECK();
    if (nVoxOut) *nVoxOut = sr.total;
    return 0;
}

}  // extern "C"
