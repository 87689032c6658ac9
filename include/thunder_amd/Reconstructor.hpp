// Reconstructor.hpp -- header-only C++ mirror of thuem/THUNDER's Reconstructor (include/Reconstructor.h:86-781) over
// the C ABI of libthunder_amd.so, MODE_3D.  Same method names, argument meaning and error behaviour (void + abort) for
// the methods on the M-step hot path: allocSpace / freeSpace / reset, setPreCal, insertP, insertDir, prepareTF,
// reconstruct, the set* flags.  F and T live in HBM between allocSpace and freeSpace, as the reference's volumes live in
// host memory between the same calls.
//
// insertP() is thread-safe like the reference's (atomic adds); it buffers nothing and pays one launch per call, so a
// caller that owns whole images x draws should prefer insertBatch() (= InsertFT, Interface.h:267-318).
// The MPI all-reduce inside the reference's prepareTF (MPI_Allreduce_Large over _hemi, src/Reconstructor.cpp:2383,2436)
// is injected as a callback so that this header needs neither MPI nor RCCL headers.
#pragma once
#include <functional>
#include <mutex>
#include <vector>

#include "Projector.hpp"

namespace thunder_amd {

class Reconstructor {
public:
    typedef std::function<void(float* dev, size_t nFloats)> AllReduce;  // in-place sum over the hemisphere

    Reconstructor() : _plan(nullptr), _F(nullptr), _T(nullptr), _iCol(nullptr), _iRow(nullptr) { defaults(); }
    // Reconstructor(mode, size, N, pf = 2, sym = NULL, a = 1.9, alpha = 15), include/Reconstructor.h ctor
    Reconstructor(int mode, int size, int N, int pf = 2, const double* symMat = nullptr, int nSym = 0, float a = 1.9f,
                  float alpha = 15.0f)
        : _plan(nullptr), _F(nullptr), _T(nullptr), _iCol(nullptr), _iRow(nullptr)
    {
        defaults();
        init(mode, size, N, pf, symMat, nSym, a, alpha);
    }
    // The reference's OWN constructor (include/Reconstructor.h:336-349; `Reconstructor recon(MODE_3D, boxsize, boxsize, 2, &sym, 1.9, 15)`,
    // appsrc/thunder_reconstruct.cpp:194): `sym` is any point-group type with nSymmetryElement() and get(L, R, i) (include/Geometry/
    // Symmetry.h:180-190) -- its R matrices are read here, the object is not kept.  NULL: no symmetry.
    template <class Sym, class = decltype(std::declval<const Sym&>().nSymmetryElement())>
    Reconstructor(int mode, int size, int N, int pf, const Sym* sym, double a = 1.9, double alpha = 15)
        : _plan(nullptr), _F(nullptr), _T(nullptr), _iCol(nullptr), _iRow(nullptr)
    {
        defaults();
        init(mode, size, N, pf, sym, a, alpha);
    }
    template <class Sym, class = decltype(std::declval<const Sym&>().nSymmetryElement())>
    void init(int mode, int size, int N, int pf, const Sym* sym, double a = 1.9, double alpha = 15)
    {
        init(mode, size, N, pf, (const double*)nullptr, 0, (float)a, (float)alpha);
        setSymmetry(sym);
    }
    // setSymmetry(const Symmetry* sym), include/Reconstructor.h:438
    template <class Sym, class = decltype(std::declval<const Sym&>().nSymmetryElement())>
    void setSymmetry(const Sym* sym)
    {
        _sym.clear();
        if (!sym) return;
        typedef decltype(mat_of_(&Sym::get)) M;              // the reference's dmat33 (Eigen, column-major .data())
        static_assert(sizeof(M) == 9 * sizeof(double), "Symmetry::get must hand out 3 x 3 matrices of doubles");
        for (int i = 0; i < sym->nSymmetryElement(); i++) {
            M L, R;
            sym->get(L, R, i);                               // SYMMETRIZE_FT transforms with R (include/Geometry/Transformation.h:170-194)
            const double* d = R.data();
            _sym.insert(_sym.end(), d, d + 9);
        }
    }
    // setFSC(const vec& FSC), include/Reconstructor.h:443: any vector type with size() and operator()(i) (Eigen's vec)
    template <class V, class = decltype(std::declval<const V&>()(0)), class = decltype(std::declval<const V&>().size())>
    void setFSC(const V& fsc)
    {
        _FSC.resize((size_t)fsc.size());
        for (size_t i = 0; i < _FSC.size(); i++) _FSC[i] = (float)fsc((int)i);
    }
    // Parallel::setMPIEnv(commSize, commRank, hemi, slav), include/Parallel.h:164-168, src/Parallel.cpp:38-57: what the mirror keeps of
    // it is WHO THE PROCESS IS -- rank 0 is the master (MASTER_ID, include/Parallel.h:42), and `IF_MASTER return;` opens prepareTF,
    // prepareO and reconstruct (src/Reconstructor.cpp:1058,1106,1132).  The communicators themselves are MPI's: the sums over the
    // hemisphere go through setAllReduce / setHemisphereComm.  Without the call the object is a hemisphere lead on its own (rank 1).
    template <class Comm>
    void setMPIEnv(int commSize, int commRank, const Comm& /*hemi*/, const Comm& /*slav*/) { _commSize = commSize; _commRank = commRank; }
    void setMPIEnv(int commSize, int commRank) { _commSize = commSize; _commRank = commRank; }
    bool isMaster() const { return _commRank == 0; }
    int commSize() const { return _commSize; }
    int commRank() const { return _commRank; }

    ~Reconstructor() { freeSpace(); }
    Reconstructor(const Reconstructor&) = delete;
    Reconstructor& operator=(const Reconstructor&) = delete;

    void init(int mode, int size, int N, int pf = 2, const double* symMat = nullptr, int nSym = 0, float a = 1.9f,
              float alpha = 15.0f)
    {
        if (mode != 1) { std::fprintf(stderr, "thunder_amd FATAL: only MODE_3D is implemented\n"); std::abort(); }
        _size = size; _N = N; _pf = pf; _a = a; _alpha = alpha;
        _sym.assign(symMat, symMat + 9 * (size_t)nSym);
        _maxRadius = size / 2 - (int)std::ceil(a);   // src/Reconstructor.cpp:89
    }

    // allocSpace(nThread), src/Reconstructor.cpp:92-136: FFT plans + F, W, C, T volumes, then reset()
    void allocSpace(unsigned int /*nThread*/ = 1)
    {
        freeSpace();
        THX_ABORT_ON(thx_reco_create(&_plan, _size, _N, _pf, _a, _alpha));
        void* p = nullptr;
        THX_ABORT_ON(thx_malloc_dev(&p, nVox() * 2 * sizeof(float)));
        _F = (float*)p;
        THX_ABORT_ON(thx_malloc_dev(&p, nVox() * sizeof(float)));
        _T = (float*)p;
        reset();
    }
    void freeSpace()
    {
        if (_plan) thx_reco_destroy(_plan);
        if (_F) thx_free_dev(_F);
        if (_T) thx_free_dev(_T);
        if (_iCol) thx_free_dev(_iCol);
        if (_iRow) thx_free_dev(_iRow);
        for (int i = 0; i < kStage; i++) { if (_stage[i]) thx_free_dev(_stage[i]); _stage[i] = nullptr; _stageCap[i] = 0; }
        if (_symF) thx_free_dev(_symF);
        if (_symT) thx_free_dev(_symT);
        _plan = nullptr; _F = _T = nullptr; _iCol = _iRow = nullptr; _symF = _symT = nullptr;
    }
    void resizeSpace(int size) { _size = size; }   // src/Reconstructor.cpp:162-176 (allocSpace must follow)
    // reset(nThread), src/Reconstructor.cpp:178-250
    void reset(unsigned int /*nThread*/ = 1)
    {
        _MAP = true; _gridCorr = true; _joinHalf = false;
        _ox = _oy = _oz = 0; _counter = 0; _nPxl = 0;
        if (_F) THX_ABORT_ON(thx_memset_dev(_F, 0, nVox() * 2 * sizeof(float)));
        if (_T) THX_ABORT_ON(thx_memset_dev(_T, 0, nVox() * sizeof(float)));
    }

    void setSymmetry(const double* symMat, int nSym) { _sym.assign(symMat, symMat + 9 * (size_t)nSym); }
    void setFSC(const float* fsc, int n) { _FSC.assign(fsc, fsc + n); }
    void setMAP(bool v) { _MAP = v; }
    void setGridCorr(bool v) { _gridCorr = v; }
    void setJoinHalf(bool v) { _joinHalf = v; }
    int maxRadius() const { return _maxRadius; }
    void setMaxRadius(int r) { _maxRadius = r; }
    double ox() const { return _ox; }
    double oy() const { return _oy; }
    double oz() const { return _oz; }
    int counter() const { return _counter; }
    void setAllReduce(AllReduce f) { _allreduce = f; }
    // setMPIEnv(commSize, commRank, hemi, slav) of Parallel (src/Parallel.cpp:38-57) in RCCL terms: the communicator of this
    // rank's hemisphere (thx_comm_init; NULL = the hemisphere is this rank alone).  prepareTF then reduces F, T, O and the
    // counter over it in native code (thx_reco_allreduce), as the reference's GPU build does with ncclAllReduce
    // (gpu/src/cuthunder.cu:4972-5067).
    void setHemisphereComm(thx_comm* hemi) { _hemi = hemi; }
    // (device pointers; the insert* methods are ASYNCHRONOUS -- call thx_device_sync(), or prepareTF, before reading through these
    // from another stream or handing them to a GPU-aware MPI)
    float* getF_dev() { return _F; }
    float* getT_dev() { return _T; }
    int getModelDim() const { return _pf * _size; }

    // setPreCal(nPxl, iCol, iRow, iPxl, iSig), src/Reconstructor.cpp:381-395.  The reference BORROWS the caller's PADDED
    // index arrays (_iColPad/_iRowPad, src/Optimiser.cpp:6741); they are copied to the device here, so the caller may
    // free them right after (as src/Optimiser.cpp:7760 does).
    void setPreCal(int nPxl, const int* iColPad, const int* iRowPad, const int* /*iPxl*/, const int* /*iSig*/)
    {
        _nPxl = nPxl;
        std::vector<int> c(nPxl), r(nPxl);
        for (int i = 0; i < nPxl; i++) { c[i] = iColPad[i] / _pf; r[i] = iRowPad[i] / _pf; }
        if (_iCol) thx_free_dev(_iCol);
        if (_iRow) thx_free_dev(_iRow);
        void* p = nullptr;
        THX_ABORT_ON(thx_malloc_dev(&p, nPxl * sizeof(int))); _iCol = (int*)p;
        THX_ABORT_ON(thx_malloc_dev(&p, nPxl * sizeof(int))); _iRow = (int*)p;
        THX_ABORT_ON(thx_memcpy_h2d(_iCol, c.data(), nPxl * sizeof(int)));
        THX_ABORT_ON(thx_memcpy_h2d(_iRow, r.data(), nPxl * sizeof(int)));
    }

    // insertDir(ox, oy, oz), src/Reconstructor.cpp:407-422
    void insertDir(double ox, double oy, double oz)
    {
        std::lock_guard<std::mutex> g(_mtx);
        _ox += ox; _oy += oy; _oz += oz; _counter += 1;
    }

    // insertDir(const dvec3& dir), include/Reconstructor.h:545
    template <class D, class = decltype(std::declval<const D&>().data())>
    void insertDir(const D& dir)
    {
        static_assert(sizeof(D) == 3 * sizeof(double), "dir must be a vector of three doubles");
        insertDir(dir.data()[0], dir.data()[1], dir.data()[2]);
    }

    // insertP(const Complex* src, const RFLOAT* ctf, const dmat33& rot, RFLOAT w, const vec* sig = NULL),
    // src/Reconstructor.cpp:782-863: src is the ALREADY TRANSLATED image row on the pixel list.
    void insertP(const Complex* src, const float* ctf, const double* rot, float w)
    {
        const double zero2[2] = {0, 0};
        insertBatch(src, ctf, &w, rot, zero2, 1, 1);
    }

    // The reference's OWN call syntax (include/Reconstructor.h:610): `reco.insertP(transImgP, ctfP, rot, w[, sig])` with the
    // reference's `Complex*`, `RFLOAT*`, Eigen's `dmat33`, `const vec*` (src/Optimiser.cpp:7129-7232) compiles against this overload.
    template <class C, class M, class V = void, class = decltype(std::declval<const M&>().data())>
    void insertP(const C* src, const float* ctf, const M& rot, float w, const V* /*sig*/ = nullptr)
    {
        static_assert(sizeof(C) == 2 * sizeof(float), "src must be a single-precision complex type (RFLOAT = float)");
        static_assert(sizeof(M) == 9 * sizeof(double), "rot must be a 3 x 3 matrix of doubles");
        insertP(reinterpret_cast<const Complex*>(src), ctf, static_cast<const double*>(rot.data()), w);
    }

    // insert(const Image& src, const Image& ctf, const dmat33& rot, RFLOAT w), src/Reconstructor.cpp:493-567 (what
    // appsrc/thunder_reconstruct.cpp:194-284 calls): EVERY pixel of the Fourier half [N][N/2+1] with i^2 + j^2 < maxRadius^2
    // (IMAGE_FOR_EACH_PIXEL_FT -- column i = 0 with both signs of j, unlike the pixel list of allocPreCalIdx) adds
    // src * Re(ctf) * w to F and Re(ctf)^2 * w to T.  srcFT / ctfFT in the reference's layout (row j < 0 at j + N).
    template <class C>
    void insert(const C* srcFT, const C* ctfFT, int N, const double* rot, float w)
    {
        static_assert(sizeof(C) == 2 * sizeof(float), "the images must be single-precision complex");
        if (N != _size) { std::fprintf(stderr, "thunder_amd FATAL: INCORRECT SIZE OF INSERTING IMAGE\n"); std::abort(); }
        std::vector<int> iCol, iRow;
        std::vector<Complex> dat;
        std::vector<float> ctf;
        const float* s = reinterpret_cast<const float*>(srcFT);
        const float* c = reinterpret_cast<const float*>(ctfFT);
        for (int j = -N / 2; j < N / 2; j++)
            for (int i = 0; i <= N / 2; i++)
                if (i * i + j * j < _maxRadius * _maxRadius) {
                    const size_t e = (size_t)(j >= 0 ? j : j + N) * (N / 2 + 1) + i;
                    iCol.push_back(i); iRow.push_back(j);
                    Complex v; v.dat[0] = s[2 * e]; v.dat[1] = s[2 * e + 1];
                    dat.push_back(v);
                    ctf.push_back(c[2 * e]);
                }
        std::lock_guard<std::mutex> g(_stageMtx);
        const int n = (int)iCol.size();
        int* dIdx = (int*)stage(5, 2 * (size_t)n * sizeof(int));
        THX_ABORT_ON(thx_memcpy_h2d(dIdx, iCol.data(), n * sizeof(int)));
        THX_ABORT_ON(thx_memcpy_h2d(dIdx + n, iRow.data(), n * sizeof(int)));
        const double zero2[2] = {0, 0};
        insert_staged(dat.data(), ctf.data(), &w, rot, zero2, 1, 1, dIdx, dIdx + n, n);
    }
    template <class C, class M, class = decltype(std::declval<const M&>().data())>
    void insert(const C* srcFT, const C* ctfFT, int N, const M& rot, float w)
    {
        static_assert(sizeof(M) == 9 * sizeof(double), "rot must be a 3 x 3 matrix of doubles");
        insert(srcFT, ctfFT, N, static_cast<const double*>(rot.data()), w);
    }

    // insert(const Image& src, const Image& ctf, const dmat33& rot, const RFLOAT w) UNDER ITS OWN NAME (include/Reconstructor.h:561-566;
    // `recon.insert(img, ctf, rot, 1)`, appsrc/thunder_reconstruct.cpp:262): any image type with nColRL() and operator[] onto its
    // Fourier half (include/Image/ImageBase.h), any 3 x 3 matrix of doubles with .data()
    // (the reference's ImageBase offers `const Complex* dataFT() const`, include/Image/ImageBase.h:265: read through it)
    template <class I, class M, class = decltype(std::declval<const I&>().dataFT()), class = decltype(std::declval<const M&>().data())>
    void insert(const I& src, const I& ctf, const M& rot, float w)
    {
        insert(src.dataFT(), ctf.dataFT(), (int)src.nColRL(), rot, w);
    }

    // nImg images x mReco draws in one launch (InsertFT): datP [nImg][nPxl] untranslated rows, rot [nImg][mReco][9],
    // tran [nImg][mReco][2] (the image is shifted by -tran on the device), w [nImg] (already divided by mReco).
    // Staging: grow-only device buffers owned by the object (no hipMalloc / hipFree and no device-wide synchronisation per call:
    // the blocking host-to-device copies of the NEXT call are ordered behind this call's kernels on the same stream); calls from
    // several host threads -- the reference inserts from an OpenMP loop -- are serialised on the staging buffers.
    void insertBatch(const Complex* datP, const float* ctfP, const float* w, const double* rot, const double* tran,
                     int nImg, int mReco)
    {
        if (!_F || !_iCol) { std::fprintf(stderr, "thunder_amd FATAL: allocSpace/setPreCal not called\n"); std::abort(); }
        std::lock_guard<std::mutex> g(_stageMtx);
        insert_staged(datP, ctfP, w, rot, tran, nImg, mReco, _iCol, _iRow, _nPxl);
    }

    // ---- members of the -DGPU_VERSION build (include/Reconstructor.h:611-659, 676-680) ----
    // insertI(datP, ctfP, sigP, w, offS, nr, nt, nd, [nc,] ctfaData, pixelSize, cSearch, opf, mReco, idim, imgNum),
    // src/Reconstructor.cpp:865-976 -> InsertFT: whole batches of images x draws; nr = QUATERNIONS [imgNum][mReco][4],
    // nt = shifts [imgNum][mReco][2], nd = defocus factors [imgNum][mReco], offS [imgNum][2]; the orientation sum and the
    // counter advance as insertDir does per draw.  nc (classes) must be NULL or all zero: one Reconstructor is one class.
    void insertI(const Complex* datP, const float* ctfP, const float* /*sigP*/, const float* w, const double* offS,
                 const double* nr, const double* nt, const double* nd, const int* /*nc*/, const thx_ctf_attr* ctfaData,
                 float pixelSize, bool cSearch, int opf, int mReco, int idim, int imgNum)
    {
        if (!_F || !_iCol) { std::fprintf(stderr, "thunder_amd FATAL: allocSpace/setPreCal not called\n"); std::abort(); }
        const size_t nd_ = (size_t)imgNum * mReco;
        // staging: the object's grow-only device buffers (no hipMalloc / hipFree per call -- round-5 review: eleven of each before);
        // calls from several host threads are serialised on them
        std::lock_guard<std::mutex> g(_stageMtx);
        auto up = [&](int slot, const void* h, size_t bytes) -> void* {
            void* d = stage(slot, bytes);
            THX_ABORT_ON(thx_memcpy_h2d(d, h, bytes));
            return d;
        };
        void* dDat = up(0, datP, (size_t)imgNum * _nPxl * 2 * sizeof(float));
        void* dCtf = up(1, ctfP, (size_t)imgNum * _nPxl * sizeof(float));
        void* dW = up(2, w, imgNum * sizeof(float));
        void* dRot = stage(3, nd_ * 9 * sizeof(double));
        void* dTran = up(4, nt, nd_ * 2 * sizeof(double));
        void* dQ = up(6, nr, nd_ * 4 * sizeof(double));
        void* dOff = offS ? up(7, offS, (size_t)imgNum * 2 * sizeof(double)) : nullptr;
        void* dAttr = cSearch ? up(8, ctfaData, imgNum * sizeof(thx_ctf_attr)) : nullptr;
        void* dDf = cSearch ? up(9, nd, nd_ * sizeof(double)) : nullptr;
        char* dSmall = (char*)stage(10, 3 * sizeof(double) + sizeof(int));
        void* dO = dSmall;
        void* dCnt = dSmall + 3 * sizeof(double);
        THX_ABORT_ON(thx_memset_dev(dSmall, 0, 3 * sizeof(double) + sizeof(int)));
        THX_ABORT_ON(thx_rotmat_dev((const double*)dQ, (double*)dRot, (int)nd_, nullptr));
        THX_ABORT_ON(thx_insert_dev(_F, _T, (double*)dO, (int*)dCnt, _pf * _size, 1, (const float*)dDat, (const float*)dCtf,
                                    (const float*)dW, (const double*)dRot, (const double*)dTran, (const double*)dOff, nullptr,
                                    (const thx_ctf_attr*)dAttr, (const double*)dDf, cSearch ? 1 : 0, pixelSize, _iCol, _iRow,
                                    opf, _nPxl, mReco, idim, imgNum, nullptr));
        double o[3];
        int c = 0;
        THX_ABORT_ON(thx_memcpy_d2h(o, dO, sizeof(o)));
        THX_ABORT_ON(thx_memcpy_d2h(&c, dCnt, sizeof(int)));
        {
            std::lock_guard<std::mutex> g2(_mtx);
            _ox += o[0]; _oy += o[1]; _oz += o[2]; _counter += c;
        }
    }
    void insertI(const Complex* datP, const float* ctfP, const float* sigP, const float* w, const double* offS,
                 const double* nr, const double* nt, const double* nd, const thx_ctf_attr* ctfaData, float pixelSize,
                 bool cSearch, int opf, int mReco, int idim, int imgNum)
    {
        insertI(datP, ctfP, sigP, w, offS, nr, nt, nd, nullptr, ctfaData, pixelSize, cSearch, opf, mReco, idim, imgNum);
    }
    int getModelSize() const { return (int)nVox(); }
    // prepareTFG(gpuIdx), src/Reconstructor.cpp:1012-1054: prepareTF with the symmetrisation on the device -- here both are
    void prepareTFG(int gpuIdx) { THX_ABORT_ON(thx_set_device(gpuIdx)); prepareTF(1); }
    // reconstructG(dst, gpuIdx, nThread), src/Reconstructor.cpp:1835-2330
    void reconstructG(float* dstRL, int gpuIdx, unsigned int nThread = 1)
    {
        THX_ABORT_ON(thx_set_device(gpuIdx));
        reconstruct(dstRL, nThread);
    }

    // prepareTF(nThread), src/Reconstructor.cpp:1056-1091: allReduceT (+ 1/T[0] normalisation of T and F), symmetrizeT,
    // allReduceF, symmetrizeF
    void prepareTF(unsigned int /*nThread*/ = 1)
    {
        if (isMaster()) return;   // IF_MASTER return; src/Reconstructor.cpp:1058
        const int dim = _pf * _size;
        // insert / insertP / insertBatch leave their kernels queued (no device-wide wait per call): everything inserted so far must
        // have landed in F / T before a caller-side all-reduce callback, a GPU-aware MPI or another stream reads them
        THX_ABORT_ON(thx_device_sync());
        if (_hemi) {   // one collective for F and T (sphere rows only) + O + counter; order of the sums is immaterial
            void *ws = nullptr, *dO = nullptr, *dC = nullptr;
            THX_ABORT_ON(thx_malloc_dev(&ws, thx_reco_allreduce_workspace(dim, _maxRadius, _pf)));
            THX_ABORT_ON(thx_malloc_dev(&dO, 3 * sizeof(double)));
            THX_ABORT_ON(thx_malloc_dev(&dC, sizeof(int)));
            double o[3] = {_ox, _oy, _oz};
            THX_ABORT_ON(thx_memcpy_h2d(dO, o, sizeof(o)));
            THX_ABORT_ON(thx_memcpy_h2d(dC, &_counter, sizeof(int)));
            THX_ABORT_ON(thx_reco_allreduce(_hemi, _F, _T, (double*)dO, (int*)dC, dim, _maxRadius, _pf, ws, nullptr));
            THX_ABORT_ON(thx_device_sync());
            THX_ABORT_ON(thx_memcpy_d2h(o, dO, sizeof(o)));
            THX_ABORT_ON(thx_memcpy_d2h(&_counter, dC, sizeof(int)));
            _ox = o[0]; _oy = o[1]; _oz = o[2];
            thx_free_dev(ws); thx_free_dev(dO); thx_free_dev(dC);
        }
        if (_allreduce) _allreduce(_T, nVox());
        THX_ABORT_ON(thx_normalise_tf_dev(_F, _T, dim, nullptr));
        const double r = (double)(_maxRadius * _pf + 1);
        const int nSym = (int)(_sym.size() / 9);
        if (nSym > 0) symm(_T, nVox(), 0, r);
        if (_allreduce) _allreduce(_F, 2 * nVox());
        if (nSym > 0) symm(_F, 2 * nVox(), 1, r);
        THX_ABORT_ON(thx_device_sync());
    }

    // prepareO(), src/Reconstructor.cpp:1104-1127 (symmetry sweep of O omitted: C1 or caller-side)
    void prepareO() { if (isMaster()) return; if (_counter) { _ox /= _counter; _oy /= _counter; _oz /= _counter; } }   // IF_MASTER return; :1106

    // reconstruct(Volume& dst, nThread), src/Reconstructor.cpp:1129-1831 -> dst [N][N][N] real, host
    void reconstruct(float* dstRL, unsigned int /*nThread*/ = 1)
    {
        if (isMaster()) return;   // IF_MASTER return; src/Reconstructor.cpp:1132
        void* d = nullptr;
        THX_ABORT_ON(thx_malloc_dev(&d, (size_t)_N * _N * _N * sizeof(float)));
        THX_ABORT_ON(thx_reco_reconstruct_dev(_plan, _F, _T, _maxRadius, _FSC.empty() ? nullptr : _FSC.data(),
                                              (int)_FSC.size(), _joinHalf ? 1 : 0, (_MAP && !_FSC.empty()) ? 1 : 0,
                                              _gridCorr ? 1 : 0, (float*)d, nullptr, nullptr, nullptr));
        THX_ABORT_ON(thx_memcpy_d2h(dstRL, d, (size_t)_N * _N * _N * sizeof(float)));
        thx_free_dev(d);
    }

    // reconstruct(Volume& dst, nThread) / reconstructG(Volume& dst, gpuIdx, nThread) with the reference's own Volume (include/Reconstructor.h:
    // 720,728; `_model.reco(t).reconstruct(ref, _para.nThreadsPerProcess)`, src/Optimiser.cpp:7366-7371): dst is allocated in real
    // space as the reference does (dst.alloc(_N, _N, _N, RL_SPACE), src/Reconstructor.cpp:1796) and filled through operator()
    template <class V, class = decltype(std::declval<V&>().alloc(0L, 0L, 0L, 0))>
    void reconstruct(V& dst, unsigned int nThread = 1)
    {
        if (isMaster()) return;   // (the reference's master returns before dst is touched)
        dst.alloc((long)_N, (long)_N, (long)_N, 0 /* RL_SPACE, include/Image/ImageBase.h:48 */);
        static_assert(sizeof(dst(0)) == sizeof(float), "the volume must be single precision (RFLOAT = float)");
        reconstruct(reinterpret_cast<float*>(&dst(0)), nThread);
    }
    template <class V, class = decltype(std::declval<V&>().alloc(0L, 0L, 0L, 0))>
    void reconstructG(V& dst, int gpuIdx, unsigned int nThread = 1)
    {
        THX_ABORT_ON(thx_set_device(gpuIdx));
        reconstruct(dst, nThread);
    }

private:
    enum { kStage = 11 };
    void* stage(int slot, size_t bytes)   // (under _stageMtx)
    {
        if (bytes > _stageCap[slot]) {
            if (_stage[slot]) { THX_ABORT_ON(thx_device_sync()); thx_free_dev(_stage[slot]); }
            const size_t cap = bytes + bytes / 2 + 256;
            THX_ABORT_ON(thx_malloc_dev(&_stage[slot], cap));
            _stageCap[slot] = cap;
        }
        return _stage[slot];
    }
    void insert_staged(const Complex* datP, const float* ctfP, const float* w, const double* rot, const double* tran, int nImg, int mReco,
                       const int* dCol, const int* dRow, int nPxl)
    {
        const size_t nd = (size_t)nImg * mReco;
        void* dDat = stage(0, (size_t)nImg * nPxl * 2 * sizeof(float));
        void* dCtf = stage(1, (size_t)nImg * nPxl * sizeof(float));
        void* dW = stage(2, nImg * sizeof(float));
        void* dRot = stage(3, nd * 9 * sizeof(double));
        void* dTran = stage(4, nd * 2 * sizeof(double));
        THX_ABORT_ON(thx_memcpy_h2d(dDat, datP, (size_t)nImg * nPxl * 2 * sizeof(float)));
        THX_ABORT_ON(thx_memcpy_h2d(dCtf, ctfP, (size_t)nImg * nPxl * sizeof(float)));
        THX_ABORT_ON(thx_memcpy_h2d(dW, w, nImg * sizeof(float)));
        THX_ABORT_ON(thx_memcpy_h2d(dRot, rot, nd * 9 * sizeof(double)));
        THX_ABORT_ON(thx_memcpy_h2d(dTran, tran, nd * 2 * sizeof(double)));
        THX_ABORT_ON(thx_insert_dev(_F, _T, nullptr, nullptr, _pf * _size, 1, (const float*)dDat, (const float*)dCtf,
                                    (const float*)dW, (const double*)dRot, (const double*)dTran, nullptr, nullptr, nullptr,
                                    nullptr, 0, 1.0f, dCol, dRow, _pf, nPxl, mReco, _N, nImg, nullptr));
    }
    size_t nVox() const { const size_t P = (size_t)_pf * _size; return P * P * (P / 2 + 1); }
    void defaults()
    {
        _size = _N = 0; _pf = 2; _a = 1.9f; _alpha = 15.0f; _maxRadius = 0; _nPxl = 0;
        _MAP = true; _gridCorr = true; _joinHalf = false; _ox = _oy = _oz = 0; _counter = 0;
    }
    // SYMMETRIZE_FT into the object's second volume of the kind, then the two trade places: no allocation per call (round-5 review)
    void symm(float*& vol, size_t nFloats, int isComplex, double r)
    {
        float*& other = isComplex ? _symF : _symT;
        if (!other) {
            void* tmp = nullptr;
            THX_ABORT_ON(thx_malloc_dev(&tmp, nFloats * sizeof(float)));
            other = (float*)tmp;
        }
        THX_ABORT_ON(thx_symmetrize_dev(other, vol, _pf * _size, isComplex, _sym.data(), (int)(_sym.size() / 9), r, nullptr));
        float* t = vol; vol = other; other = t;
    }
    template <class S, class M> static M mat_of_(void (S::*)(M&, M&, int) const);
    thx_reco* _plan;
    float *_F, *_T;
    float *_symF = nullptr, *_symT = nullptr;
    int _commSize = 1, _commRank = 1;
    int *_iCol, *_iRow;
    int _size, _N, _pf, _maxRadius, _nPxl, _counter;
    float _a, _alpha;
    bool _MAP, _gridCorr, _joinHalf;
    double _ox, _oy, _oz;
    std::vector<double> _sym;
    std::vector<float> _FSC;
    AllReduce _allreduce;
    thx_comm* _hemi = nullptr;
    std::mutex _mtx, _stageMtx;
    void* _stage[kStage] = {};
    size_t _stageCap[kStage] = {};
};

}  // namespace thunder_amd
