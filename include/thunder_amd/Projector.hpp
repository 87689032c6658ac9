// Projector.hpp -- header-only C++ mirror of thuem/THUNDER's Projector (include/Projector.h:85-379) over the C ABI
// of libthunder_amd.so.  Same method names, argument meaning and error behaviour (void + abort) for the methods on the
// E-step hot path; host pointers in and out, the padded FT stays resident in HBM between calls.
//
// Types: `Complex` is {float dat[2]} as in include/Precision.h:100-109; `dmat33` arguments are passed as 9 doubles,
// column-major (Eigen's default layout -- on the reference side pass mat.data()).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "../thunder_amd.h"

namespace thunder_amd {

struct Complex {
    float dat[2];
};

#define THX_ABORT_ON(rc)                                                                     \
    do {                                                                                     \
        if ((rc) != 0) {                                                                     \
            std::fprintf(stderr, "thunder_amd FATAL: %s (%s:%d)\n", thx_last_error(), __FILE__, __LINE__); \
            std::abort(); /* REPORT_ERROR + abort(), include/Logging.h:28-34 */              \
        }                                                                                    \
    } while (0)

class Projector {
public:
    Projector() : _mode(1), _maxRadius(-1), _interp(1), _pf(2), _N(0), _vol(nullptr), _plan(nullptr) {}
    ~Projector() { clear(); }
    Projector(const Projector&) = delete;             // BOOST_MOVABLE_BUT_NOT_COPYABLE, include/Projector.h:87
    Projector& operator=(const Projector&) = delete;
    Projector(Projector&& o) noexcept { steal(o); }
    Projector& operator=(Projector&& o) noexcept { if (this != &o) { clear(); steal(o); } return *this; }
    void swap(Projector& o) { Projector t(std::move(o)); o = std::move(*this); *this = std::move(t); }

    int mode() const { return _mode; }
    void setMode(int mode) { _mode = mode; }
    int maxRadius() const { return _maxRadius; }
    void setMaxRadius(int r) { _maxRadius = r; }
    int interp() const { return _interp; }
    void setInterp(int interp) { _interp = interp; }
    int pf() const { return _pf; }
    void setPf(int pf) { _pf = pf; }
    int vdim() const { return _pf * _N; }
    const float* projectee3D_dev() const { return _vol; }   // device pointer of the padded FT

    // setProjectee(Volume src, nThread), src/Projector.cpp:123-148: src = FT of the N^3 reference (half-complex
    // [N][N][N/2+1]); inverse FFT, zero-pad x pf, TIK grid correction, forward FFT -- all on the device.
    void setProjectee(const Complex* srcFT, int N, unsigned int /*nThread*/ = 1)
    {
        alloc(N);
        const size_t nh = (size_t)N * N * (N / 2 + 1);
        void *ft = nullptr, *rl = nullptr;
        THX_ABORT_ON(thx_malloc_dev(&ft, nh * 2 * sizeof(float)));
        THX_ABORT_ON(thx_malloc_dev(&rl, (size_t)N * N * N * sizeof(float)));
        THX_ABORT_ON(thx_memcpy_h2d(ft, srcFT, nh * 2 * sizeof(float)));
        THX_ABORT_ON(thx_fft3d_bw_dev((float*)ft, (float*)rl, N, nullptr));
        THX_ABORT_ON(thx_reco_set_projectee_dev(_plan, (const float*)rl, _vol, nullptr));
        THX_ABORT_ON(thx_device_sync());
        thx_free_dev(ft);
        thx_free_dev(rl);
        _maxRadius = (_pf * N) / _pf / 2 - 1;   // floor(MIN_3(nCol,nRow,nSlc) / pf / 2 - 1), :131-133
    }

    // setProjectee(Volume src, nThread) with the reference's own Volume (include/Projector.h:235; `_proj[l].setProjectee(_ref[l].copyVolume(),
    // nThread)`, src/Model.cpp:1037): any volume type with nColRL() and operator[] onto its Fourier half [N][N][N/2+1] is taken as it is
    // (the reference passes the Volume by value -- a moved temporary -- and its operator[] is not const: taken by forwarding reference)
    template <class V, class = decltype(std::declval<V&>().nColRL())>
    void setProjectee(V&& src, unsigned int nThread = 1)
    {
        static_assert(sizeof(src[0]) == 2 * sizeof(float), "the volume must be single-precision complex (RFLOAT = float)");
        setProjectee(reinterpret_cast<const Complex*>(&src[0]), (int)src.nColRL(), nThread);
    }

    // same, starting from the real-space map [N][N][N] (wrapped index layout)
    void setProjecteeRL(const float* srcRL, int N)
    {
        alloc(N);
        void* rl = nullptr;
        THX_ABORT_ON(thx_malloc_dev(&rl, (size_t)N * N * N * sizeof(float)));
        THX_ABORT_ON(thx_memcpy_h2d(rl, srcRL, (size_t)N * N * N * sizeof(float)));
        THX_ABORT_ON(thx_reco_set_projectee_dev(_plan, (const float*)rl, _vol, nullptr));
        THX_ABORT_ON(thx_device_sync());
        thx_free_dev(rl);
        _maxRadius = N / 2 - 1;
    }

    // project(Complex* dst, const dmat33& mat, const int* iCol, const int* iRow, int nPxl, unsigned nThread) const,
    // include/Projector.h:293-299, src/Projector.cpp:356-374.  const + stateless: callable from many OpenMP threads.
    void project(Complex* dst, const double* mat, const int* iCol, const int* iRow, int nPxl,
                 unsigned int /*nThread*/ = 1) const
    {
        projectBatch(dst, mat, 1, iCol, iRow, nPxl);
    }

    // The reference's OWN call syntax (include/Projector.h:293-299): `proj.project(priP, rot, _iCol, _iRow, _nPxl, nThread)` with
    // the reference's `Complex*` and Eigen's `dmat33` (src/Optimiser.cpp:775-781,1294-1308) compiles against this overload -- any
    // 8-byte complex type and any 3 x 3 matrix of doubles with a column-major .data() (Eigen's default) are taken as they are.
    template <class C, class M, class = decltype(std::declval<const M&>().data())>
    void project(C* dst, const M& mat, const int* iCol, const int* iRow, int nPxl, unsigned int nThread = 1) const
    {
        static_assert(sizeof(C) == 2 * sizeof(float), "dst must be a single-precision complex type (RFLOAT = float)");
        static_assert(sizeof(M) == 9 * sizeof(double), "mat must be a 3 x 3 matrix of doubles");
        project(reinterpret_cast<Complex*>(dst), static_cast<const double*>(mat.data()), iCol, iRow, nPxl, nThread);
    }

    // project(Image& dst, const dmat33& mat, nThread), include/Projector.h:257-268, src/Projector.cpp:274-292: every pixel of the
    // half-plane with i^2 + j^2 < maxRadius^2 (IMAGE_FOR_PIXEL_R_FT(_maxRadius)) is set through Image::setFT; the rest of dst is
    // left as it is.  dstFT = the image's Fourier half [N][N/2+1] in the reference's layout (row j < 0 at j + N).  What
    // appsrc/thunder_project.cpp:146-236 calls.
    template <class C>
    void projectImage(C* dstFT, int N, const double* mat, unsigned int /*nThread*/ = 1) const
    {
        static_assert(sizeof(C) == 2 * sizeof(float), "dstFT must be a single-precision complex type");
        std::vector<int> iCol, iRow, iPxl;
        const int r = _maxRadius;
        for (int j = -r; j < r; j++)
            for (int i = 0; i <= r; i++)
                if (i * i + j * j < r * r) { iCol.push_back(i); iRow.push_back(j); iPxl.push_back((j >= 0 ? j : j + N) * (N / 2 + 1) + i); }
        std::vector<Complex> row(iCol.size());
        projectBatch(row.data(), mat, 1, iCol.data(), iRow.data(), (int)iCol.size());
        for (size_t p = 0; p < iCol.size(); p++) std::memcpy(&dstFT[iPxl[p]], &row[p], sizeof(Complex));
    }
    template <class C, class M, class = decltype(std::declval<const M&>().data())>
    void projectImage(C* dstFT, int N, const M& mat, unsigned int nThread = 1) const
    {
        static_assert(sizeof(M) == 9 * sizeof(double), "mat must be a 3 x 3 matrix of doubles");
        projectImage(dstFT, N, static_cast<const double*>(mat.data()), nThread);
    }

    // nR matrices at once (what ExpectProject does, Interface.h:210-219): dst [nR][nPxl]
    void projectBatch(Complex* dst, const double* mats, int nR, const int* iCol, const int* iRow, int nPxl) const
    {
        if (!_vol) { std::fprintf(stderr, "thunder_amd FATAL: Projector has no projectee\n"); std::abort(); }
        // device staging from a pool the object owns: a calling thread takes a slot (grow-only buffers), uses it, hands it back -- no
        // hipMalloc / hipFree per call (round-5 review: four of each before), and concurrent calls from the caller's OpenMP threads
        // (`project` is const and called with nThread = 1 from every thread, src/Optimiser.cpp:758-781) do not share buffers
        Slot* sl = take_slot();
        void* dOut = sl->grow(0, (size_t)nR * nPxl * 2 * sizeof(float));
        void* dMat = sl->grow(1, (size_t)nR * 9 * sizeof(double));
        void* dCol = sl->grow(2, (size_t)nPxl * sizeof(int));
        void* dRow = sl->grow(3, (size_t)nPxl * sizeof(int));
        THX_ABORT_ON(thx_memcpy_h2d(dMat, mats, (size_t)nR * 9 * sizeof(double)));
        THX_ABORT_ON(thx_memcpy_h2d(dCol, iCol, (size_t)nPxl * sizeof(int)));
        THX_ABORT_ON(thx_memcpy_h2d(dRow, iRow, (size_t)nPxl * sizeof(int)));
        THX_ABORT_ON(thx_project_dev(_vol, (float*)dOut, (const double*)dMat, (const int*)dCol, (const int*)dRow, nR, _pf,
                                     _pf * _N, nPxl, nullptr));
        THX_ABORT_ON(thx_memcpy_d2h(dst, dOut, (size_t)nR * nPxl * 2 * sizeof(float)));
        give_slot(sl);
    }

    // project(Image& dst, const dmat33& mat, const unsigned int nThread) const UNDER ITS OWN NAME (include/Projector.h:257-268;
    // `proj.project(img, mat, nThread)`, appsrc/thunder_project.cpp:207): any image type with nColRL() and operator[] onto its Fourier
    // half, any 3 x 3 matrix of doubles with .data()
    template <class I, class M, class = decltype(std::declval<I&>().nColRL()), class = decltype(std::declval<const M&>().data())>
    void project(I& dst, const M& mat, unsigned int nThread = 1) const
    {
        projectImage(&dst[0], (int)dst.nColRL(), mat, nThread);
    }

private:
    void alloc(int N)
    {
        if (_plan && N == _N) return;
        clear();
        _N = N;
        THX_ABORT_ON(thx_reco_create(&_plan, N, N, _pf, 1.9f, 15.0f));
        const int P = _pf * N;
        void* v = nullptr;
        THX_ABORT_ON(thx_malloc_dev(&v, (size_t)P * P * (P / 2 + 1) * 2 * sizeof(float)));
        _vol = (float*)v;
    }
    void clear()
    {
        if (_vol) thx_free_dev(_vol);
        if (_plan) thx_reco_destroy(_plan);
        _vol = nullptr;
        _plan = nullptr;
        std::lock_guard<std::mutex> g(_poolMtx);
        for (Slot* sl : _pool) { for (int i = 0; i < 4; i++) if (sl->p[i]) thx_free_dev(sl->p[i]); delete sl; }
        _pool.clear();
    }
    struct Slot {
        void* p[4] = {nullptr, nullptr, nullptr, nullptr};
        size_t cap[4] = {0, 0, 0, 0};
        void* grow(int i, size_t bytes)
        {
            if (bytes > cap[i]) {
                if (p[i]) thx_free_dev(p[i]);    // (the slot's last user waited for its copy back: nothing is in flight on it)
                const size_t c = bytes + bytes / 2 + 256;
                THX_ABORT_ON(thx_malloc_dev(&p[i], c));
                cap[i] = c;
            }
            return p[i];
        }
    };
    Slot* take_slot() const
    {
        std::lock_guard<std::mutex> g(_poolMtx);
        if (_pool.empty()) return new Slot();
        Slot* sl = _pool.back();
        _pool.pop_back();
        return sl;
    }
    void give_slot(Slot* sl) const
    {
        std::lock_guard<std::mutex> g(_poolMtx);
        _pool.push_back(sl);
    }
    void steal(Projector& o)
    {
        _mode = o._mode; _maxRadius = o._maxRadius; _interp = o._interp; _pf = o._pf; _N = o._N;
        _vol = o._vol; _plan = o._plan;
        o._vol = nullptr; o._plan = nullptr;
        std::lock_guard<std::mutex> g(o._poolMtx);
        _pool.swap(o._pool);
    }
    int _mode, _maxRadius, _interp, _pf, _N;
    float* _vol;
    thx_reco* _plan;
    mutable std::mutex _poolMtx;
    mutable std::vector<Slot*> _pool;
};

}  // namespace thunder_amd
