// thx_comm.hip -- the half-set exchange in native code: communicators and the reduction of the Fourier accumulators.
// Reference behaviour: gpu/src/cuthunder.cu:4192-4206 (ncclGetUniqueId on the hemisphere's root, MPI_Bcast of the id,
// ncclCommInitRank), :4972-5067 (ncclAllReduce of F, T, O, counter over the hemisphere), and the CPU path
// Reconstructor::allReduceF / allReduceT (src/Reconstructor.cpp:2350-2484, MPI_Allreduce_Large over _hemi).
// One process per GPU; xGMI is point-to-point, so the reduce is ONE large ring collective per iteration, half and class:
// F (re, im) and T of the voxels inside the sample sphere only, packed into one contiguous buffer (50 % of the grid).
//
// A communicator runs over one of two TRANSPORTS behind the same entry points (chosen by the 128-byte id it is built from):
//   rccl  the product path: ncclAllReduce / ncclReduce / ncclBroadcast on the caller's stream;
//   shm   TEST-ONLY (THX_COMM_TRANSPORT=shm when the id is drawn): device -> host copy, a POSIX shared-memory segment with
//         one slot per rank and a sense-reversing barrier, rank-ordered sums on the host, host -> device copy.  RCCL refuses
//         two ranks on one device; this does not, so 2 or 4 processes sharing one GPU run the UNCHANGED multi-rank branches
//         of thx_refine_iterate (tests/test_multirank_gpu.py).  Host-synchronous and slow by construction; never timed.
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "thx_common.h"

namespace thx {

enum class DT { F32, F64, I32, I64, U8 };
enum class OP { SUM, MAX };

static size_t dt_size(DT t) { return t == DT::F32 || t == DT::I32 ? 4 : (t == DT::U8 ? 1 : 8); }

struct Transport {
    virtual ~Transport() {}
    virtual const char* name() const = 0;
    // in place on DEVICE buffers; root < 0: every rank receives the result, otherwise only `root` does (the others' buffers
    // are left as they were)
    virtual int reduce(void* buf, size_t count, DT t, OP op, int root, hipStream_t st) = 0;
    virtual int broadcast(void* buf, size_t bytes, int root, hipStream_t st) = 0;
};

}  // namespace thx

struct thx_comm {
    thx::Transport* tp = nullptr;
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


// ---------------------------------------------------------------------------------------------
// transport 1: RCCL (the product path)
// ---------------------------------------------------------------------------------------------
struct RcclTransport : Transport {
    ncclComm_t c = nullptr;
    ~RcclTransport() override { if (c) (void)ncclCommDestroy(c); }
    const char* name() const override { return "rccl"; }
    static ncclDataType_t nt(DT t)
    {
        switch (t) {
            case DT::F32: return ncclFloat;
            case DT::F64: return ncclDouble;
            case DT::I32: return ncclInt32;
            case DT::I64: return ncclInt64;
            default: return ncclChar;
        }
    }
    int reduce(void* buf, size_t count, DT t, OP op, int root, hipStream_t st) override
    {
        const ncclRedOp_t o = op == OP::SUM ? ncclSum : ncclMax;
        if (root < 0) THX_NCCL_CHECK(ncclAllReduce(buf, buf, count, nt(t), o, c, st));
        else THX_NCCL_CHECK(ncclReduce(buf, buf, count, nt(t), o, root, c, st));
        return 0;
    }
    int broadcast(void* buf, size_t bytes, int root, hipStream_t st) override
    {
        THX_NCCL_CHECK(ncclBroadcast(buf, buf, bytes, ncclChar, root, c, st));
        return 0;
    }
};

// ---------------------------------------------------------------------------------------------
// transport 2 (TEST-ONLY): host shared memory.  Segment = header + one slot of slotBytes per rank.  A collective moves
// the buffer through the slots chunk by chunk: D2H into the own slot, barrier, every receiving rank combines the slots in
// RANK ORDER (so all of them end with the same bits, as an RCCL all-reduce guarantees), barrier, H2D.
// ---------------------------------------------------------------------------------------------
static const char kShmMagic[8] = {'T', 'H', 'X', 'S', 'H', 'M', '1', 0};

struct ShmHeader {
    std::atomic<int> ready;        // set to 0x7458 by rank 0 once the header is initialised
    std::atomic<int> attached;     // ranks that have mapped the segment
    std::atomic<int> arrived;      // barrier: arrivals of the current generation
    std::atomic<int> generation;
    std::atomic<int> failed;       // a rank gave up (time-out / error): everybody else stops waiting
    int size;
    size_t slotBytes;
};

struct ShmTransport : Transport {
    ShmHeader* hdr = nullptr;
    unsigned char* slots = nullptr;
    size_t mapBytes = 0, slotBytes = 0;
    int rank = 0, size = 1;
    char shmName[64] = {0};
    std::vector<unsigned char> host;   // the combined chunk
    double timeoutS = 300.0;

    const char* name() const override { return "shm"; }

    static double now()
    {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
    }

    ~ShmTransport() override
    {
        if (hdr) {
            (void)munmap(hdr, mapBytes);
            if (rank == 0) (void)shm_unlink(shmName);   // (already gone after a complete init: harmless)
        }
    }

    int init(const char* nm, int rank_, int size_)
    {
        rank = rank_; size = size_;
        snprintf(shmName, sizeof(shmName), "%s", nm);
        const char* e = getenv("THX_COMM_SHM_SLOT_MB");
        slotBytes = (size_t)((e && atol(e) > 0) ? atol(e) : 4) << 20;   // (small by default: container /dev/shm can be 64 MiB; buffers travel in chunks of a slot)
        if ((e = getenv("THX_COMM_SHM_TIMEOUT_S")) && atof(e) > 0) timeoutS = atof(e);
        const size_t hdrBytes = 4096;
        mapBytes = hdrBytes + (size_t)size * slotBytes;
        int fd = -1;
        const double t0 = now();
        if (rank == 0) {
            fd = shm_open(shmName, O_CREAT | O_EXCL | O_RDWR, 0600);
            if (fd < 0) { set_error("shm transport: shm_open(%s) failed on rank 0", shmName); return -1; }
            if (ftruncate(fd, (off_t)mapBytes) != 0) { (void)close(fd); (void)shm_unlink(shmName); set_error("shm transport: ftruncate(%zu) failed", mapBytes); return -1; }
        } else {
            while (true) {   // the segment appears when rank 0 has created AND sized it
                fd = shm_open(shmName, O_RDWR, 0600);
                if (fd >= 0) {
                    struct stat sb;
                    if (fstat(fd, &sb) == 0 && (size_t)sb.st_size >= mapBytes) break;
                    (void)close(fd);
                    fd = -1;
                }
                if (now() - t0 > timeoutS) { set_error("shm transport: rank %d timed out waiting for %s", rank, shmName); return -1; }
                usleep(1000);
            }
        }
        void* m = mmap(nullptr, mapBytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        (void)close(fd);
        if (m == MAP_FAILED) { set_error("shm transport: mmap of %zu bytes failed", mapBytes); return -1; }
        hdr = reinterpret_cast<ShmHeader*>(m);
        slots = reinterpret_cast<unsigned char*>(m) + hdrBytes;
        if (rank == 0) {   // (a fresh segment is zero-filled: the atomics start at 0)
            hdr->size = size;
            hdr->slotBytes = slotBytes;
            hdr->ready.store(0x7458, std::memory_order_release);
        } else {
            while (hdr->ready.load(std::memory_order_acquire) != 0x7458) {
                if (now() - t0 > timeoutS) { set_error("shm transport: rank %d timed out waiting for the header", rank); return -1; }
                usleep(200);
            }
            if (hdr->size != size || hdr->slotBytes != slotBytes) { set_error("shm transport: rank %d disagrees with rank 0 on size / slot bytes", rank); return -1; }
        }
        hdr->attached.fetch_add(1);
        THX_RC(barrier());
        if (rank == 0) (void)shm_unlink(shmName);   // every rank holds a mapping: the name can go (nothing is left behind in /dev/shm)
        return 0;
    }

    int barrier()
    {
        const int gen = hdr->generation.load(std::memory_order_acquire);
        if (hdr->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == size) {
            hdr->arrived.store(0, std::memory_order_relaxed);
            hdr->generation.fetch_add(1, std::memory_order_release);
            return 0;
        }
        const double t0 = now();
        for (unsigned spin = 0; hdr->generation.load(std::memory_order_acquire) == gen; spin++) {
            if (hdr->failed.load(std::memory_order_relaxed)) { set_error("shm transport: another rank failed"); return -1; }
            if (spin < 200) sched_yield();
            else {
                usleep(50);
                if ((spin & 1023) == 0 && now() - t0 > timeoutS) {
                    hdr->failed.store(1);
                    set_error("shm transport: rank %d timed out in a barrier (%.0f s)", rank, timeoutS);
                    return -1;
                }
            }
        }
        return 0;
    }

    template <typename T>
    static void combine(T* out, const unsigned char* slots, size_t slotBytes, int size, size_t n, OP op)
    {
        const T* s0 = reinterpret_cast<const T*>(slots);
        for (size_t i = 0; i < n; i++) out[i] = s0[i];
        for (int r = 1; r < size; r++) {
            const T* s = reinterpret_cast<const T*>(slots + (size_t)r * slotBytes);
            if (op == OP::SUM) for (size_t i = 0; i < n; i++) out[i] = out[i] + s[i];
            else for (size_t i = 0; i < n; i++) out[i] = s[i] > out[i] ? s[i] : out[i];
        }
    }

    int fail(int rc) { hdr->failed.store(1); return rc; }

    int reduce(void* buf, size_t count, DT t, OP op, int root, hipStream_t st) override
    {
        if (hipStreamSynchronize(st) != hipSuccess) { set_error("shm transport: stream error before a collective"); return fail(-1); }
        const size_t es = dt_size(t), per = slotBytes / es;
        if (host.size() < slotBytes) host.resize(slotBytes);
        for (size_t o = 0; o < count; o += per) {
            const size_t n = count - o < per ? count - o : per;
            unsigned char* d = reinterpret_cast<unsigned char*>(buf) + o * es;
            if (hipMemcpy(slots + (size_t)rank * slotBytes, d, n * es, hipMemcpyDeviceToHost) != hipSuccess) { set_error("shm transport: D2H failed"); return fail(-1); }
            THX_RC(barrier());
            const bool mine = root < 0 || root == rank;
            if (mine) {
                switch (t) {
                    case DT::F32: combine(reinterpret_cast<float*>(host.data()), slots, slotBytes, size, n, op); break;
                    case DT::F64: combine(reinterpret_cast<double*>(host.data()), slots, slotBytes, size, n, op); break;
                    case DT::I32: combine(reinterpret_cast<int*>(host.data()), slots, slotBytes, size, n, op); break;
                    case DT::I64: combine(reinterpret_cast<long long*>(host.data()), slots, slotBytes, size, n, op); break;
                    default: set_error("shm transport: bad data type"); return fail(-1);
                }
            }
            THX_RC(barrier());   // everybody has read the slots: they may be refilled
            if (mine && hipMemcpy(d, host.data(), n * es, hipMemcpyHostToDevice) != hipSuccess) { set_error("shm transport: H2D failed"); return fail(-1); }
        }
        return 0;
    }

    int broadcast(void* buf, size_t bytes, int root, hipStream_t st) override
    {
        if (hipStreamSynchronize(st) != hipSuccess) { set_error("shm transport: stream error before a broadcast"); return fail(-1); }
        for (size_t o = 0; o < bytes; o += slotBytes) {
            const size_t n = bytes - o < slotBytes ? bytes - o : slotBytes;
            unsigned char* d = reinterpret_cast<unsigned char*>(buf) + o;
            if (rank == root && hipMemcpy(slots, d, n, hipMemcpyDeviceToHost) != hipSuccess) { set_error("shm transport: D2H failed"); return fail(-1); }
            THX_RC(barrier());
            if (rank != root && hipMemcpy(d, slots, n, hipMemcpyHostToDevice) != hipSuccess) { set_error("shm transport: H2D failed"); return fail(-1); }
            THX_RC(barrier());
        }
        return 0;
    }
};

static bool comm_active(const thx_comm* c) { return c && c->tp && (c->size > 1 || knobs().commForce); }

}  // namespace thx

using namespace thx;

extern "C" {

int thx_comm_unique_id(void* id128)
{
    THX_REQUIRE(id128, "id128 is NULL");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    const char* tp = getenv("THX_COMM_TRANSPORT");
    if (tp && strcmp(tp, "shm") == 0) {   // TEST-ONLY transport: the id carries the name of the shared-memory segment
        static std::atomic<unsigned> ctr{0};
        char* id = reinterpret_cast<char*>(id128);
        memset(id, 0, 128);
        memcpy(id, kShmMagic, sizeof(kShmMagic));
        timespec ts;
        clock_gettime(CLOCK_REALTIME, &ts);
        snprintf(id + 8, 64, "/thx_comm_%d_%u_%lx", (int)getpid(), ctr.fetch_add(1), (unsigned long)ts.tv_nsec);
        return 0;
    }
    THX_REQUIRE(!tp || strcmp(tp, "rccl") == 0, "THX_COMM_TRANSPORT: rccl (default) or shm (test-only)");
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
    if (memcmp(id128, kShmMagic, sizeof(kShmMagic)) == 0) {
        // loud, once per process: every collective of this communicator is a host-synchronous D2H + barrier + H2D (round-5 advisor)
        static std::atomic<bool> warned{false};
        if (!warned.exchange(true))
            fprintf(stderr, "thunder_amd: WARNING: THX_COMM_TRANSPORT=shm -- the TEST-ONLY shared-memory transport carries this job's "
                            "collectives through host memory; unset it to use RCCL\n");
        ShmTransport* t = new ShmTransport;
        const int rc = t->init(reinterpret_cast<const char*>(id128) + 8, rank, size);
        if (rc) { delete t; delete c; return rc; }
        c->tp = t;
    } else {
        RcclTransport* t = new RcclTransport;
        ncclUniqueId id;
        memcpy(&id, id128, sizeof(id));
        ncclResult_t r = ncclCommInitRank(&t->c, size, id, rank);
        if (r != ncclSuccess) {
            set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, size, ncclGetErrorString(r));
            t->c = nullptr;
            delete t;
            delete c;
            return 2000 + (int)r;
        }
        c->tp = t;
    }
    *out = c;
    return 0;
}

int thx_comm_destroy(thx_comm* c)
{
    if (!c) return 0;
    delete c->tp;
    delete c;
    return 0;
}

int thx_comm_rank(const thx_comm* c) { return c ? c->rank : 0; }
int thx_comm_size(const thx_comm* c) { return c ? c->size : 1; }
const char* thx_comm_transport(const thx_comm* c) { return c && c->tp ? c->tp->name() : "none"; }

int thx_comm_allreduce_f32(thx_comm* c, float* buf, size_t count, void* stream)
{
    if (!comm_active(c) || count == 0) return 0;
    return c->tp->reduce(buf, count, DT::F32, OP::SUM, -1, as_stream(stream));
}

int thx_comm_allreduce_f64(thx_comm* c, double* buf, size_t count, void* stream)
{
    if (!comm_active(c) || count == 0) return 0;
    return c->tp->reduce(buf, count, DT::F64, OP::SUM, -1, as_stream(stream));
}

int thx_comm_allreduce_i32(thx_comm* c, int* buf, size_t count, void* stream)
{
    if (!comm_active(c) || count == 0) return 0;
    return c->tp->reduce(buf, count, DT::I32, OP::SUM, -1, as_stream(stream));
}

int thx_comm_allreduce_i64(thx_comm* c, long long* buf, size_t count, void* stream)
{
    if (!comm_active(c) || count == 0) return 0;
    return c->tp->reduce(buf, count, DT::I64, OP::SUM, -1, as_stream(stream));
}

int thx_comm_allreduce_max_f64(thx_comm* c, double* buf, size_t count, void* stream)
{
    if (!comm_active(c) || count == 0) return 0;
    return c->tp->reduce(buf, count, DT::F64, OP::MAX, -1, as_stream(stream));
}

int thx_comm_reduce_i64(thx_comm* c, long long* buf, size_t count, int root, void* stream)
{
    if (!comm_active(c) || count == 0) return 0;
    THX_REQUIRE(root >= 0 && root < c->size, "root out of range");
    return c->tp->reduce(buf, count, DT::I64, OP::SUM, root, as_stream(stream));
}

int thx_comm_broadcast(thx_comm* c, void* buf, size_t bytes, int root, void* stream)
{
    if (!comm_active(c) || bytes == 0) return 0;
    THX_REQUIRE(root >= 0 && root < c->size, "root out of range");
    return c->tp->broadcast(buf, bytes, root, as_stream(stream));
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
    if (!comm_active(hemi)) return 0;   // the half lives on one rank: nothing to exchange
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
    THX_RC(hemi->tp->reduce(buf, (size_t)sr.total * 3, DT::F32, OP::SUM, -1, st));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sphere_pack<1>), dim3(blocks), dim3(256), 0, st, reinterpret_cast<float2*>(F), T, buf,
                       sr.rowOff, dim, sr.total);
    THX_LAUNCH_CHECK();
    if (O) THX_RC(hemi->tp->reduce(O, 3, DT::F64, OP::SUM, -1, st));
    if (counter) THX_RC(hemi->tp->reduce(counter, 1, DT::I32, OP::SUM, -1, st));
    return 0;
}

size_t thx_reco_allreduce_acc_workspace(int dim, int maxRadius, int pf)
{
    return 2 * thx_reco_allreduce_workspace(dim, maxRadius, pf);   // 3 x 8 bytes per sphere voxel
}

// The half-set reduce BEFORE the accumulators become floats: every rank of the half has accumulated in the same quanta
// (thx_insert_scale_dev takes the extrema over the hemisphere), integer addition is associative, so the sums of N ranks are
// bit for bit what one rank would have accumulated over all the particles -- summing the converted floats
// (thx_reco_allreduce) depends on how the particles were dealt to the ranks.  One ring collective over the sphere rows
// (ncclInt64: 24 bytes per voxel instead of 12).
int thx_reco_allreduce_acc(thx_comm* hemi, void* acc, double* O, int* counter, int dim, int maxRadius, int pf, void* workspace,
                           void* stream)
{
    return thx_reco_allreduce_acc_class(hemi, acc, 1, 0, O, counter, dim, maxRadius, pf, workspace, stream);
}

// class k of a session over nK classes (acc = [nK][vol][2] F | [nK][vol] T, thx_insert_acc_bytes(dim, nK)): the iteration
// driver reduces its K triples one after the other through the same workspace.  root < 0: all-reduce (every rank of the half
// ends with the sums, the reference's ncclAllReduce); root >= 0: ncclReduce to that rank of the half -- the rank that will
// reconstruct class k -- which about halves the bytes every link of the ring carries; the other ranks' accumulators of the
// class are left as they were (partial sums: not to be used).
int thx_reco_reduce_acc_class(thx_comm* hemi, void* acc, int nK, int k, int root, int dim, int maxRadius, int pf, void* workspace,
                              void* stream)
{
    if (!comm_active(hemi)) return 0;
    THX_REQUIRE(acc && workspace && dim > 0 && maxRadius > 0 && pf > 0 && nK >= 1 && k >= 0 && k < nK, "bad arguments");
    THX_REQUIRE(root < hemi->size, "root out of range");
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
    THX_RC(hemi->tp->reduce(buf, (size_t)sr.total * 3, DT::I64, OP::SUM, root, st));
    if (root < 0 || root == hemi->rank) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sphere_pack_acc<1>), dim3(blocks), dim3(256), 0, st, accF, accT, buf, sr.rowOff, dim, sr.total);
        THX_LAUNCH_CHECK();
    }
    return 0;
}

int thx_reco_allreduce_acc_class(thx_comm* hemi, void* acc, int nK, int k, double* O, int* counter, int dim, int maxRadius, int pf,
                                 void* workspace, void* stream)
{
    if (!comm_active(hemi)) return 0;
    THX_RC(thx_reco_reduce_acc_class(hemi, acc, nK, k, -1, dim, maxRadius, pf, workspace, stream));
    if (O) THX_RC(hemi->tp->reduce(O, 3, DT::F64, OP::SUM, -1, as_stream(stream)));
    if (counter) THX_RC(hemi->tp->reduce(counter, 1, DT::I32, OP::SUM, -1, as_stream(stream)));
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
