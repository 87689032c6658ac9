/*
 * thunder_amd.h -- C ABI of the MI355X-native E/M hot path (libthunder_amd.so).
 *
 * Drop-in boundary for thuem/THUNDER v1.4.14's GPU plug-in surface gpu/interface/Interface.h:16-528
 * and, through it, the Projector / Reconstructor methods on the per-iteration hot path
 * (include/Projector.h:293-299, include/Reconstructor.h:530,550,610,703,720).  Every entry point
 * names the reference interface it replaces (paths relative to the reference tree).  The C++
 * marshalling stub a THUNDER maintainer adds on the reference side (Volume& / vec / vector<> ->
 * plain pointers) is shown in INTEGRATION.md; header-only Projector / Reconstructor mirrors live in
 * include/thunder_amd/.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only.  All functions return 0 on success, non-zero on
 *     failure; thx_last_error() gives the message (the reference's own style is void + abort --
 *     the C++ mirrors restore that).
 *   - RFLOAT is float (SINGLE_PRECISION build, CMakeLists.txt:48); Complex is {float re, im}
 *     (include/Precision.h:100-109) and is passed as float* with interleaved re,im.
 *   - Rotation matrices are 9 doubles, column-major (Eigen dmat33, include/Typedef.h:149).
 *   - Volume FT layout: [P][P][P/2+1] complex, index (k<0?k+P:k)*(P/2+1)*P + (j<0?j+P:j)*(P/2+1) + i
 *     (include/Image/Volume.h:567-575).  T volumes are REAL float on this side of the boundary
 *     (the reference stores T complex with a never-written imaginary part; its own GPU interface
 *     also passes RFLOAT* T3D, Interface.h:337-344).
 *   - *_dev functions: every pointer is a DEVICE pointer unless the parameter is documented
 *     "host"; work is enqueued on `stream` (a hipStream_t, NULL = default stream) and the call
 *     returns without synchronising unless stated.
 *   - *_host functions (the Interface.h-shaped ones): every pointer is a caller-owned HOST pointer,
 *     results are written back into the caller's buffers before return, exactly as
 *     gpu/src/cuthunder.cu does.
 */
#ifndef THUNDER_AMD_H
#define THUNDER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* struct CTFAttr, include/Database.h:302-330 (7 RFLOATs, same order) */
typedef struct thx_ctf_attr {
    float voltage;
    float defocusU;
    float defocusV;
    float defocusTheta;
    float Cs;
    float amplitudeContrast;
    float phaseShift;
} thx_ctf_attr;

const char* thx_last_error(void);
int thx_version(void);
/* The THX_* environment switches (A/B selection between tested code paths: THX_INSERT_PLAIN, THX_FFT, THX_EXPECT_ND,
 * THX_EXPECT_KERNEL, ...) are read ONCE, when the library is loaded.  Test harnesses that change the environment
 * afterwards call this to re-read them; it must not run concurrently with launches. */
int thx_knobs_reload(void);

/* getAviDevice(std::vector<int>&), Interface.h:16 -- number of visible gfx950 devices */
int thx_device_count(int* count);
/* selects the device later calls on this thread use (the reference passes gpuIdx per call) */
int thx_set_device(int gpuIdx);

/* device-memory plumbing for hosts that do not link a HIP runtime themselves (the C++ Projector / Reconstructor
 * mirrors in include/thunder_amd/ use only these; cuthunder.cu does the same job with cudaMalloc / cudaMemcpy) */
int thx_malloc_dev(void** ptr, size_t bytes);
int thx_free_dev(void* ptr);
int thx_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes);
int thx_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes);
int thx_memset_dev(void* dst_dev, int value, size_t bytes);
int thx_device_sync(void);

/* ---------------------------------------------------------------------------------------------
 * E-step building blocks
 * ------------------------------------------------------------------------------------------- */

/* rotate3D(dmat33&, const dvec4&), src/Geometry/Euler.cpp:181-189 (kernel_getRotMat):
 * quat [n][4] -> mat [n][9] column-major. */
int thx_rotmat_dev(const double* quat, double* mat, int n, void* stream);

/* translate(Complex* dst, tx, ty, nCol, nRow, iCol, iRow, nPxl), src/Image/ImageFunctions.cpp:233-252
 * for nT shifts at once (ExpectRotran, Interface.h:199-208): trans [nT][2] doubles -> traP [nT][nPxl]. */
int thx_translate_dev(float* traP, const double* trans, int nT, const int* iCol, const int* iRow, int nPxl, int idim,
                      void* stream);

/* CTF(RFLOAT* dst, pixelSize, V, dU, dV, theta, Cs, A, phi, nCol, nRow, iCol, iRow, nPxl), src/CTF.cpp:113-151,
 * for nImg images (GCTFinit Interface.h:524 / allocPreCal src/Optimiser.cpp:8083-8110).
 * dfac (optional, may be NULL) [nImg] multiplies defocusU/V as src/Optimiser.cpp:7183-7202 does. */
int thx_ctf_dev(float* ctfP, const thx_ctf_attr* attr, const double* dfac, float pixelSize, const int* iCol,
                const int* iRow, int nPxl, int idim, int nImg, void* stream);

/* Optimiser::allocPreCal, ctf = true branch, src/Optimiser.cpp:8124-8169 (ExpectPrecal, Interface.h:166-174;
 * kernel_ExpectPrectf): freq [nPxl] (may be NULL), def [nImg][nPxl], k1/k2 [nImg] for the defocus search.
 * (lambda's constant on this branch is 12.2643274, :8164, unlike CTF()'s 12.2643247.) */
int thx_expect_precal_dev(float* freq, float* def, float* k1, float* k2, const thx_ctf_attr* attr, int idim,
                          float pixelSize, const int* iCol, const int* iRow, int nPxl, int nImg, void* stream);

/* CTF rows of the defocus search, src/Optimiser.cpp:1246-1272: ctfP [nImg][nD][nPxl] from the rows of
 * thx_expect_precal_dev and the defocus factors dpara [nImg][nD] (Particle::d, double) -- the ctfP layout
 * thx_expect_local_dev takes when nD > 1. */
int thx_ctf_dsearch_dev(float* ctfP, const float* freq, const float* def, const float* k1, const float* k2,
                        const thx_ctf_attr* attr, const double* dpara, int nD, int nPxl, int nImg, void* stream);

/* CTF(Image& dst, pixelSize, ...), src/CTF.cpp:31-66, for nImg images (Optimiser::initCTF / GCTFinit,
 * Interface.h:524-528): ctfFT [nImg][idim][idim/2+1] complex64, CTF in the real part. */
int thx_ctf_image_dev(float* ctfFT, const thx_ctf_attr* attr, float pixelSize, int idim, int nImg, void* stream);

/* Optimiser::allocPreCal gather (src/Optimiser.cpp:8055-8075): packs full image FTs
 * img [nImg][idim][idim/2+1] into datP [nImg][nPxl] through iPxl. */
int thx_gather_pixels_dev(float* datP, const float* img, const int* iPxl, int nPxl, int idim, int nImg, void* stream);

/* Projector::project(Complex* dst, const dmat33&, iCol, iRow, nPxl, nThread) const, src/Projector.cpp:356-374,
 * for nR matrices at once (ExpectProject, Interface.h:210-219; kernel_Project3D).  volume = padded, grid-corrected
 * FT held by the Projector (vdim = P = pf*N); out [nR][nPxl].  Trilinear only (interp == LINEAR_INTERP).
 * Results are bit-identical to the reference's arithmetic (fp64 matmul -> fp32 trilinear, no contraction). */
int thx_project_dev(const float* volume, float* rotP, const double* rotMat, const int* iCol, const int* iRow, int nR,
                    int pf, int vdim, int nPxl, void* stream);

/* logDataVSPrior(dat, pri, ctf, sigRcp, m), src/Optimiser.cpp:9187-9213 for n (pri) rows against one image row:
 * out[i] = sum_pix |dat - ctf*pri_i|^2 * sigRcp.  Wave-tree summation (the reference's own scalar and AVX
 * variants already differ in order). */
int thx_logdatavsprior_dev(float* out, const float* dat, const float* pri, const float* ctf, const float* sigRcp,
                           int nPri, int nPxl, void* stream);

/* One particle-filter phase for a batch of images: the body of HOT LOOP B, src/Optimiser.cpp:1225-1406
 * (ExpectLocalP/RTD/PreI3D/M, Interface.h:50-145, collapsed into one batched call).
 *   volumes      [nVol] padded FTs back to back (vdim^3 half-complex each); volIdx [nImg] (NULL = 0)
 *   datP, ctfP, sigRcpP  image-major rows [nImg][nPxl]; when nD > 1 ctfP is [nImg][nD][nPxl]
 *                (rows built as src/Optimiser.cpp:1263-1287)
 *   rotMat [nImg][nR][9], trans [nImg][nT][2] (pixels, double as Particle::t)
 *   pC [nImg], pR [nImg][nR], pT [nImg][nT], pD [nImg][nD]  priors (Particle::wC/wR/wT/wD, double)
 * outputs (RFLOAT): wC [nImg], wR [nImg][nR], wT [nImg][nT], wD [nImg][nD], baseLine [nImg],
 *   logW (optional, may be NULL) [nImg][nD][nT][nR] every log-likelihood.
 * Weights are relative to the per-image maximum exactly as the reference's running-baseline rescale leaves
 * them (mathematically identical; float rounding differs -- tolerance in tests/test_parity_gpu.py).
 * workspace: thx_expect_local_workspace() bytes of device scratch.
 * wgPerCU: occupancy of the local-search kernel, workgroups (4 waves each) per CU; 0 = unlimited, negative = the
 *   library default (2).  With the particle filter's clouds of support points (rotations ~1 degree apart, the
 *   reference's production case) the kernel is bound by scattered 64-byte reads and is 17 % faster with 8 waves per CU
 *   than with 20; callers that feed tightly clustered rotations (all within ~0.2 degree) should pass 0.  Results do
 *   not depend on it.
 * active (DEVICE [nImg] ints, may be NULL): images with active[img] == 0 are skipped and their outputs left untouched --
 *   the per-image stop rule of the local search (thx_pf_stop_rule_dev) ends an image's phases this way. */
size_t thx_expect_local_workspace(int nImg, int nR, int nT, int nD);
int thx_expect_local_dev(const float* volumes, const int* volIdx, int vdim, int pf, int idim, const int* iCol,
                         const int* iRow, int nPxl, int nImg, const float* datP, const float* ctfP,
                         const float* sigRcpP, const double* rotMat, int nR, const double* trans, int nT, int nD,
                         const double* pC, const double* pR, const double* pT, const double* pD, float* wC, float* wR,
                         float* wT, float* wD, float* baseLine, float* logW, void* workspace, int wgPerCU, const int* active,
                         void* stream);

/* Cell-packed projector volume: for every cell origin of the half grid the 8 corner values of its trilinear cell stored
 * contiguously (64 bytes), so that one sample's gather is ONE contiguous read instead of four reads from four cache
 * lines -- 8x the memory (4.3 GB at vdim = 512; the MI355X has 288 GB), 2-4x fewer lines through L1/L2 when the rotations of
 * an image are a degree or more apart.  cells: thx_projector_packed_bytes(vdim) bytes per volume, built from the
 * standard layout; rebuild after the volume changes (Projector::setProjectee / Model::refreshProj). */
size_t thx_projector_packed_bytes(int vdim);
int thx_projector_pack_dev(float* cells, const float* volumes, int vdim, int nVol, void* stream);

/* thx_expect_local_dev gathering from cell-packed volumes (bit-identical results). */
int thx_expect_local_packed_dev(const float* cells, const int* volIdx, int vdim, int pf, int idim, const int* iCol,
                                const int* iRow, int nPxl, int nImg, const float* datP, const float* ctfP,
                                const float* sigRcpP, const double* rotMat, int nR, const double* trans, int nT, int nD,
                                const double* pC, const double* pR, const double* pT, const double* pD, float* wC,
                                float* wR, float* wT, float* wD, float* baseLine, float* logW, void* workspace,
                                int wgPerCU, const int* active, void* stream);

/* Global scanning phase for class kIdx: src/Optimiser.cpp:756-894 (ExpectGlobal3D, Interface.h:221-237).
 *   rotP [nR][nPxl] slices (thx_project_dev), traP [nT][nPxl] ramps (thx_translate_dev)
 *   datP/ctfP/sigRcpP image-major [nImg][nPxl]; pR [nImg][nR], pT [nImg][nT] priors
 *   wC [nImg][nK], wR [nK][nImg][nR], wT [nK][nImg][nT], baseL [nImg] are READ-MODIFY-WRITE and carry over
 *   from class to class as in the reference (baseL = NaN means "unset", :737-745). */
size_t thx_expect_global_workspace(int nImg, int nR, int nT);
int thx_expect_global_dev(const float* rotP, const float* traP, const float* datP, const float* ctfP,
                          const float* sigRcpP, const double* pR, const double* pT, float* wC, float* wR, float* wT,
                          float* baseL, int kIdx, int nK, int nR, int nT, int nPxl, int nImg, void* workspace,
                          void* stream);

/* ---------------------------------------------------------------------------------------------
 * M-step
 * ------------------------------------------------------------------------------------------- */

/* HOT LOOP C + Reconstructor::insertP/insertDir, src/Optimiser.cpp:7038-7241, src/Reconstructor.cpp:407-422,
 * 782-863 (InsertFT, Interface.h:267-318; kernel_InsertT/InsertF).
 *   F [nK] complex volumes, T [nK] real volumes (dim^3 half grids, dim = pf*size), accumulated in place
 *   datP [nImg][nPxl] (UNMASKED images, _imgOri), ctfP [nImg][nPxl]; w [nImg] per-image weight (already / mReco)
 *   rotMat [nImg][mReco][9], trans [nImg][mReco][2], offS [nImg][2] (NULL = 0): image shifted by -(tran - offS)
 *   cls [nImg][mReco] (NULL = class 0); iCol/iRow UNPADDED pixel indices, opf = padding factor
 *   cSearch != 0: CTF recomputed per draw from attr [nImg] with defocus * dfac [nImg][mReco] (:7183-7202)
 *   O (3 doubles) += -R*(tran-offS) and counter += 1 per draw (insertDir).
 * Accumulation is 64-bit fixed point (every voxel term rounded once to the call's quanta, integer atomics): F / T come out
 * bit-identical from run to run -- the reference's `omp atomic` float sums do not.  The brick-sorted form behind it
 * (thx_insert_sort.hip) keeps its sample records in library scratch: THX_INSERT_SCRATCH_MB megabytes per (device, stream),
 * default min(32 GiB, 40 % of the free memory), never less than one image's worst case; requires mReco < 4096.  The call (and
 * thx_insert_accumulate_dev) WAITS on the host a few times -- once on `stream` for the images' group counts, once per chunk of
 * images on an event for the number of segment descriptors to sort (the next chunk is already queued behind it: the GPU does
 * not idle) -- so it cannot be captured into a graph; work queued on other streams is not affected. */
int thx_insert_dev(float* F, float* T, double* O, int* counter, int dim, int nK, const float* datP,
                   const float* ctfP, const float* w, const double* rotMat, const double* trans, const double* offS,
                   const int* cls, const thx_ctf_attr* attr, const double* dfac, int cSearch, float pixelSize,
                   const int* iCol, const int* iRow, int opf, int nPxl, int mReco, int idim, int nImg, void* stream);

/* The same insertion as a SESSION over several calls (batches of images, and -- through the hemisphere communicator --
 * several ranks) that accumulate into ONE pair of 64-bit fixed-point volumes before anything becomes a float:
 *   bounds  DEVICE [nImg][2]: per image max(|re| + |im|) of its row and max |ctf| (they bound the image's largest term);
 *   thx_insert_scale_dev: gexp DEVICE [2] = the exponents of the session's quanta from the extrema over `nImg` images of this
 *     rank and, with hemi != NULL, over every rank of the half (one ncclMax of four numbers); nImgHemi = images of the
 *     whole half (head-room of the 64-bit sums);
 *   acc  DEVICE, thx_insert_acc_bytes(dim, nK) bytes, zeroed by the caller before the first accumulate call:
 *     [nK][vol][2] F | [nK][vol] T, long long;
 *   thx_insert_accumulate_dev: thx_insert_dev's arguments without F / T, any number of calls (bounds / w / rows of THAT call's
 *     images);
 *   thx_reco_allreduce_acc (thx_comm.hip): the half-set reduce on the integers -- N ranks give bit for bit the one-rank sums;
 *   thx_insert_finish_dev: F += accF 2^-gexp[0], T += accT 2^-gexp[1].
 * thx_insert_dev is bounds + scale + a zeroed scratch acc + accumulate + finish in one call. */
typedef struct thx_comm thx_comm;   /* RCCL communicator handle, declared with the thx_comm_* calls below */
size_t thx_insert_acc_bytes(int dim, int nK);
int thx_insert_bounds_dev(float* bounds, const float* datP, const float* ctfP, int nPxl, int nImg, void* stream);
int thx_insert_scale_dev(int* gexp, const float* bounds, const float* w, int nImg, int mReco, int cSearch, long nImgHemi,
                         thx_comm* hemi, void* stream);
int thx_insert_accumulate_dev(void* acc, const int* gexp, const float* bounds, double* O, int* counter, int dim, int nK,
                              const float* datP, const float* ctfP, const float* w, const double* rotMat, const double* trans,
                              const double* offS, const int* cls, const thx_ctf_attr* attr, const double* dfac, int cSearch,
                              float pixelSize, const int* iCol, const int* iRow, int opf, int nPxl, int mReco, int idim, int nImg,
                              void* stream);
int thx_insert_finish_dev(float* F, float* T, const void* acc, const int* gexp, int dim, int nK, void* stream);
/* measurement: (image, group) pairs inserted on the current device since the last reset -- a group = the draws of an image
 * that share a rotation [, class, defocus factor] and are inserted as one; the window kernel issues 24 LDS adds per listed
 * pixel and GROUP (8 voxels x re, im, T), which is what its roofline counts (bench.py).  Synchronises the stream. */
int thx_insert_groups_total(unsigned long long* out, int reset, void* stream);

/* Reconstructor::allReduceT tail (RECONSTRUCTOR_NORMALISE_T_F), src/Reconstructor.cpp:2455-2476:
 * sf = 1/T[0]; T *= sf; F *= sf.  (The sum over ranks itself is RCCL, done by the caller on F/T.) */
int thx_normalise_tf_dev(float* F, float* T, int dim, void* stream);

/* Reconstructor::symmetrizeF / symmetrizeT = SYMMETRIZE_FT, include/Geometry/Transformation.h:105-131,170-194,
 * src/Reconstructor.cpp:2676-2690 (PrepareTF, Interface.h:320-326; kernel_SymmetrizeF/T).
 * dst = src + sum_s interp(src, R_s k) inside |R_s k| < r (r = maxRadius*pf + 1); dst != src.
 * symMat host pointer [nSym][9]. isComplex: 1 for F, 0 for T. */
int thx_symmetrize_dev(float* dst, const float* src, int dim, int isComplex, const double* symMat_host, int nSym,
                       double r, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Reconstructor state (allocSpace / freeSpace, src/Reconstructor.cpp:92-160: FFT plans + W, C volumes)
 * ------------------------------------------------------------------------------------------- */
typedef struct thx_reco thx_reco;

/* Reconstructor(mode=MODE_3D, size, N, pf, sym, a, alpha) + allocSpace: include/Reconstructor.h ctor,
 * src/Reconstructor.cpp:43-90,92-136.  Builds the MKB_RL_R2 table (_kernelRL, 1e5 steps), hipFFT plans for
 * (pf*size)^3 and the W / C / scratch volumes. */
int thx_reco_create(thx_reco** out, int size, int N, int pf, float a, float alpha);
int thx_reco_destroy(thx_reco* r);

/* Reconstructor::reconstruct(Volume& dst, nThread), src/Reconstructor.cpp:1129-1831, MODE_3D
 * (ExposePT + ExposeWT + ExposePFW/ExposePF + ExposeCorrF, Interface.h:337-505).
 *   F, T: device volumes AFTER prepareTF (T is modified in place as in the reference: Wiener term, 1e-25 floor)
 *   FSC host pointer [nFSC] (used when MAP != 0), joinHalf, gridCorr as setMAP/setJoinHalf/setGridCorr
 *   dstRL device real volume [N][N][N] (wrapped index layout, origin at [0][0][0])
 *   nIterOut / diffCOut (host, optional): balancing rounds done and last distance.
 * Synchronises the stream once per balancing round (the convergence test of :1530-1551 runs on the host). */
int thx_reco_reconstruct_dev(thx_reco* r, const float* F, float* T, int maxRadius, const float* FSC_host, int nFSC,
                             int joinHalf, int MAP, int gridCorr, float* dstRL, int* nIterOut, float* diffCOut,
                             void* stream);
/* the same without a host synchronisation on the hand-written gridding loop (power-of-two grids 64 .. 1024: the stop rule is
 * evaluated on the device and the queued rounds after it fall through): `result` -- 8 ints of DEVICE or page-locked HOST memory --
 * receives {done, balancing rounds, -, diffC as float bits, -} when the stream gets there */
int thx_reco_reconstruct_async_dev(thx_reco* r, const float* F, float* T, int maxRadius, const float* FSC_host, int nFSC,
                                   int joinHalf, int MAP, int gridCorr, float* dstRL, void* result, void* stream);
/* bounds of the gridding loop of the following reconstructions on this plan (default MAX_N_ITER_BALANCE = 30, MIN_N_ITER_BALANCE
 * = 10, include/Reconstructor.h): two implementations are compared after the SAME number of rounds this way where the stop rule
 * -- a max norm against 0.95 x its previous value -- would let rounding decide (tests at P = 1024) */
int thx_reco_set_balance_rounds(thx_reco* r, int maxIter, int minIter);
/* T = max(T, 1e-25) (src/Reconstructor.cpp:1322-1324) alone: what a reconstruction leaves in the caller's T.  The iteration driver
 * calls it on a rank that runs only the MAP-on reconstruction of a class (the MAP-off one having gone to another rank of the half),
 * so that T sees the reference's sequence floor -> Wiener term -> floor.  Overwrites the plan's W. */
int thx_reco_floor_T_dev(thx_reco* r, float* T, int maxRadius, void* stream);

/* Projector::setProjectee(Volume src, nThread), src/Projector.cpp:123-148 + gridCorrection :524-606, starting from
 * the real-space map (the reference first fft.bw's the FT it is handed): zero-pad x pf, divide by TIK_RL,
 * forward FFT.  refRL device [N][N][N]; volume out [pf*N]^3 half-complex. */
int thx_reco_set_projectee_dev(thx_reco* r, const float* refRL, float* volume, void* stream);

/* forward / backward FFT of an N^3 map on the device (FFT::fw / FFT::bw, src/FFT.cpp:176-232; 1/size on bw) */
int thx_fft3d_fw_dev(const float* rl, float* ft, int n, void* stream);
int thx_fft3d_bw_dev(float* ft, float* rl, int n, void* stream);

/* FSC(vec&, const Volume& A, const Volume& B), src/Functions/Spectrum.cpp:302-337: two half-complex FTs of
 * dim^3 maps -> fsc [nShell] (device). */
int thx_fsc_dev(float* fsc, int nShell, const float* A, const float* B, int dim, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Callers either side of the E/M loop
 * ------------------------------------------------------------------------------------------- */

/* Optimiser::reMaskImg, src/Optimiser.cpp:6093-6149 (ReMask, Interface.h:517-522; cuthunder::reMask
 * gpu/src/cuthunder.cu:9406-9631): for every image FFT::bwExecutePlan (c2r, x 1/size, src/FFT.cpp:346-360),
 * MUL_RL by softMask(mask, r, ew) (src/Functions/Mask.cpp:334-350), FFT::fwExecutePlan (r2c).
 * imgFT [nImg][idim][idim/2+1] complex64 on the device, transformed IN PLACE (the real rows overlay the complex
 * rows, so no second image stack is needed).  maskRadiusPx = maskRadius / pixelSize; ew = EDGE_WIDTH_RL = 6
 * (include/Macro.h:99).  2-D FFTs by rocFFT (batched plans, cached). */
int thx_remask_dev(float* imgFT, int nImg, int idim, float maskRadiusPx, float ew, void* stream);

/* translate(Image& dst, const Image& src, tx, ty, nThread), src/Image/ImageFunctions.cpp:269-284 (r < 0: every
 * stored pixel; Optimiser::reCentreImg src/Optimiser.cpp:6078-6082) and the radius form :322-339 (r >= 0:
 * only i^2+j^2 < r^2 is written; TranslateI2D Interface.h:504-508) for nImg images: trans [nImg][2] doubles on
 * the device.  dst may alias src. */
int thx_translate_image_dev(float* dst, const float* src, const double* trans, int nImg, int idim, float r,
                            void* stream);

/* translate(Volume& dst, const Volume& src, r, tx, ty, tz, nThread), src/Image/ImageFunctions.cpp:363-384
 * (TranslateI, Interface.h:510-515; reference re-centring src/Optimiser.cpp:7418-7428) on a dim^3 half-complex FT;
 * voxels outside r are not written.  dst may alias src. */
int thx_translate_volume_dev(float* dst, const float* src, int dim, float r, double ox, double oy, double oz,
                             void* stream);

/* Optimiser::normCorrection, src/Optimiser.cpp:6201-6394 (called at the head of Optimiser::maximization from the second
 * iteration on, :3405-3413), as include/Config.h configures it (OPTIMISER_NORM_MASK, _CTF_ON_THE_FLY):
 *   thx_norm_residual_dev: norm [nImg] = sum over rL^2 <= i^2 + j^2 < rNorm^2 of |img - ctf . P . ramp(t)|^2 with P the top pose's
 *     slice inside projR = Projector::_maxRadius (img = the MASKED stack _img; rotMat [nImg][9], trans [nImg][2]);
 *     rNorm = min(_r, Model::resolutionP(0.75)) is the caller's (src/Functions/Spectrum.cpp:339-363 on the previous FSC);
 *   thx_median_f32_dev: out (device, 1 float) = median(values, n) as gsl_stats_quantile_from_sorted_data(.., 0.5) -- over the norms
 *     of ALL particles of the job (the reference all-reduces the vector over MPI_COMM_WORLD first);
 *   thx_norm_scale_dev: img[l] *= sqrt(median / norm[l]), imgOri[l] *= the same, in place ([nImg][idim][idim/2+1] complex64).
 * All device pointers. */
int thx_norm_residual_dev(float* norm, const float* volumes, const int* volIdx, int vdim, int pf, int idim, int projR, float rL,
                          float rNorm, const float* img, const thx_ctf_attr* attr, const double* dfac, float pixelSize,
                          const double* rotMat, const double* trans, int nImg, void* stream);
int thx_norm_residual_packed_dev(float* norm, const float* cells, const int* volIdx, int vdim, int pf, int idim, int projR, float rL,
                                 float rNorm, const float* img, const thx_ctf_attr* attr, const double* dfac, float pixelSize,
                                 const double* rotMat, const double* trans, int nImg, void* stream);   /* cell-packed references; bit-identical */
int thx_median_f32_dev(float* out, const float* values, int n, void* stream);
int thx_norm_scale_dev(float* img, float* imgOri, const float* norm, const float* median, int idim, int nImg, void* stream);

/* Per-image part of Optimiser::allReduceSigma, src/Optimiser.cpp:6443-6565, as include/Config.h configures it
 * (OPTIMISER_SIGMA_RANK1ST, _SIGMA_WHOLE_FREQUENCY, _RECENTRE_IMAGE_EACH_ITERATION, _CTF_ON_THE_FLY; w = 1):
 * with P = the top pose's slice (radius projR = Projector::_maxRadius), spec [nImg][4][rSig] receives the shell
 * power spectra (powerSpectrum, src/Functions/Spectrum.cpp:161-190) of
 *   0: ctf.P.ramp(t)   1: img   2: img - ctf.P.ramp(t)   3: imgOri - ctf.P.ramp(t - offset).
 * img/imgOri [nImg][idim][idim/2+1] full image FTs (masked / unmasked), rotMat [nImg][9], trans [nImg][2],
 * offset [nImg][2] or NULL, dfac [nImg] or NULL (defocus factor of SEARCH_TYPE_CTF), volumes/volIdx as in
 * thx_expect_local_dev.  All device pointers. */
int thx_sigma_spectra_dev(float* spec, const float* volumes, const int* volIdx, int vdim, int pf, int idim, int projR,
                          int rSig, const float* img, const float* imgOri, const thx_ctf_attr* attr,
                          const double* dfac, float pixelSize, const double* rotMat, const double* trans,
                          const double* offset, int nImg, void* stream);
/* the same with `cells` = the cell-packed copies of the references (thx_projector_pack_dev): one 64-byte request per sample; the
 * spectra are bit-identical (what the iteration driver calls) */
int thx_sigma_spectra_packed_dev(float* spec, const float* cells, const int* volIdx, int vdim, int pf, int idim, int projR,
                                 int rSig, const float* img, const float* imgOri, const thx_ctf_attr* attr,
                                 const double* dfac, float pixelSize, const double* rotMat, const double* trans,
                                 const double* offset, int nImg, void* stream);

/* Group accumulation of allReduceSigma, src/Optimiser.cpp:6567-6597: sigM/sigN/svd [nGroup][rSig+1] (device,
 * READ-MODIFY-WRITE; last column = weight sum) += this rank's images.  groupID_host [nImg] is the HOST array
 * Optimiser::_groupID (1-based); group == 0 pools everything into row 0.  The caller all-reduces the three
 * tables over the hemisphere (:6608-6650) before thx_sigma_final_dev. */
int thx_sigma_accum_dev(float* sigM, float* sigN, float* svd, const float* spec, const int* groupID_host, int nImg,
                        int nGroup, int rSig, int group, void* stream);

/* Closing arithmetic of allReduceSigma, src/Optimiser.cpp:6654-6707: sig, sigRcp [nGroup][rSig] (device) from the
 * reduced tables; maskRadius (Angstrom), size, pixelSize feed alpha (:6683). */
int thx_sigma_final_dev(float* sig, float* sigRcp, const float* sigM, const float* sigN, const float* svd, int nGroup,
                        int rSig, int group, float maskRadius, int size, float pixelSize, void* stream);

/* Model::compareTwoHemispheres, MODE_3D (src/Model.cpp:307-700) on the two half maps' FTs A, B (DEVICE complex64
 * [N][N][N/2+1], modified only by the averaging):
 *   fscHost (HOST [rU], may be NULL = fscFlag off): the gold-standard FSC (src/Functions/Spectrum.cpp:302-337); with a mask
 *     (maskRL DEVICE [N]^3 = _maskFSC, or maskRL NULL and coreR > 0 = _coreFSC with softMask(mask, coreR, ew),
 *     src/Functions/Mask.cpp:470-486) its mask-corrected form (:424-563): randomPhaseThres = resP(fscUnmask, 0.8, 1, 1),
 *     FSC of the masked phase-randomised halves (randomPhase, src/Functions/Spectrum.cpp:365-386; phases from the Philox
 *     stream (seed, element, call / call + 1, 9) instead of GSL's global generator), FSC of the masked halves,
 *     (fscMask - fscRF) / (1 - fscRF) beyond randomPhaseThres + 2.  *randomPhaseThresOut (HOST, optional).
 *   avgFlag: A = B = (A + B) / 2 inside QUAD_3 < avgR^2 (gold standard, one reference, :629-674; avgR =
 *     min(AROUND(resA2P(1 / A_B_AVERAGE_THRES, size, pixelSize)), r) is the caller's), avgR < 0: everywhere (:688-696).
 * Synchronises the stream (the FSC curves come back to the host). */
int thx_compare_hemispheres_dev(float* A, float* B, int N, int rU, float* fscHost, const float* maskRL, float coreR, float ew,
                                int avgFlag, int avgR, unsigned long long seed, unsigned call, int* randomPhaseThresOut,
                                void* stream);
/* its pieces, exposed for the parity tests: softMask(Volume& mask, r, ew) and randomPhase(dst, src, r) (phases: DEVICE
 * [N][N][N/2+1] floats receiving the angles, may be NULL) */
int thx_core_mask_dev(float* mask, int N, float r, float ew, void* stream);
/* softMask(Volume& dst, const Volume& src, r, ew, bg), src/Functions/Mask.cpp:499-521, in place on a DEVICE [N]^3 map
 * (Optimiser::solventFlatten's spherical mask, src/Optimiser.cpp:7958-7975) */
int thx_soft_mask_volume_dev(float* vol, int N, float r, float ew, float bg, void* stream);
int thx_random_phase_dev(float* dst, const float* src, int N, int r, unsigned long long seed, unsigned call, float* phases,
                         void* stream);

/* ---------------------------------------------------------------------------------------------
 * Data formats either side of the path
 * ------------------------------------------------------------------------------------------- */

/* ImageFile::readMetaDataMRC, src/Image/ImageFile.cpp:209-228 (header = include/Image/MRCHeader.h, 1024 bytes). */
int thx_mrc_info(const char* path, int* nx, int* ny, int* nz, int* mode, int* nsymbt);

/* ImageFile::readImageMRC(Image&, iSlc), src/Image/ImageFile.cpp:248-264, for slices [first, first+count):
 * dst HOST float [count][ny][nx] in the reference's in-memory layout (origin at index 0: IMAGE_READ_CAST moves the
 * file's centre-origin samples with MESH_IMAGE_INDEX, include/Image/ImageFile.h:383-388,418-435).  Modes 0, 1, 2. */
int thx_mrc_read_images(const char* path, int first, int count, float* dst);

/* ImageFile::readVolumeMRC, src/Image/ImageFile.cpp:289-303 (VOLUME_READ_CAST + MESH_VOLUME_INDEX): dst HOST [nz][ny][nx]. */
int thx_mrc_read_volume(const char* path, float* dst);

/* ImageFile::writeVolumeMRC (src/Image/ImageFile.cpp:332-357) and openStack/writeStack/closeStack (:359-404):
 * mode 2, header as ImageFile::fillMRCHeader (:171-207), cell dimensions scaled by pixelSize.  src HOST arrays in the
 * in-memory layout; a stack's slices are images (meshed in x, y only). */
int thx_mrc_write_volume(const char* path, const float* src, int nx, int ny, int nz, float pixelSize);
int thx_mrc_write_stack(const char* path, const float* src, int size, int nSlc, float pixelSize);

/* The .thu particle table (include/Database.h:22-287; Database::nParticle/nGroup/ctf/path/groupID/cls/quat/tran/...,
 * src/Database.cpp:137-640): one particle per line, blank-separated columns; blank and '#' lines are skipped as
 * Database::reGenDatabase does (:40-100).  thx_thu_load fills whichever HOST arrays are non-NULL:
 * ctf [n], particlePath [n][pathStride] ("000001@stack.mrcs" syntax left to the caller, src/Optimiser.cpp:4646-4660),
 * groupID / classID [n], quat [n][4], tran / stdT [n][2], defocusFactor / score [n]. */
int thx_thu_count(const char* path, int* nParticle, int* nGroup);
int thx_thu_load(const char* path, int nParticle, thx_ctf_attr* ctf, char* particlePath, int pathStride, int* groupID,
                 int* classID, double* quat, double* tran, double* stdT, double* defocusFactor, double* score);

/* Optimiser::saveDatabase + writeDescInfo, src/Optimiser.cpp:8217-8416: writes (append = 0) or appends (the reference's
 * ranks append in turn) the 27-column particle table: the '#' description block, then one line per particle in the
 * reference's column order and formats (%18.9lf / %6d / %6lu).  HOST arrays; ctf and particlePath [n][pathStride] are
 * required, the others may be NULL (written as the neutral values the loader assumes): micrographPath [n][micStride],
 * coordXY [n][2], groupID / classID [n], quat [n][4] (Particle::rank1st), k123 [n][3], tran [n][2] (already minus the
 * re-centring offset, OPTIMISER_RECENTRE_IMAGE_EACH_ITERATION), stdT [n][2], defocusFactor / stdDefocus / score [n]. */
int thx_thu_write(const char* path, int append, int nParticle, const thx_ctf_attr* ctf, const char* particlePath, int pathStride,
                  const char* micrographPath, int micStride, const double* coordXY, const int* groupID, const int* classID,
                  const double* quat, const double* k123, const double* tran, const double* stdT, const double* defocusFactor,
                  const double* stdDefocus, const double* score);
/* the columns thx_thu_load leaves out (micrograph path, coordinates, K1..K3, sd of the defocus factor); any may be NULL */
int thx_thu_load_extra(const char* path, int nParticle, char* micrographPath, int micStride, double* coordXY, double* k123,
                       double* stdDefocus);

/* Optimiser::substractBgImg, src/Optimiser.cpp:4928-4962 (OPTIMISER_INIT_IMG_NORMALISE_OUT_MASK_REGION): per image
 * (x - bgMean) / bgStddev over the pixels outside maskRadiusPx (bgMeanStddev, src/Image/ImageFunctions.cpp:607-621).
 * imgRL DEVICE float [nImg][idim][idim], in-memory layout, in place. */
int thx_img_subtract_bg_dev(float* imgRL, int nImg, int idim, float maskRadiusPx, void* stream);

/* Per-image terms of Optimiser::statImg, src/Optimiser.cpp:4838-4870: stat DEVICE double [nImg][4] =
 * {regionMean(img, r, 0), bgStddev(0, img, r), stddev(0, img), bgStddev^2}; the caller sums them over its images,
 * all-reduces over the hemisphere and divides by N as :4877-4915 does. */
int thx_img_stats_dev(double* stat, const float* imgRL, int nImg, int idim, float maskRadiusPx, void* stream);

/* Optimiser::maskImg (zeroMask) + normaliseImg + fwImg, src/Optimiser.cpp:4964-5024: imgOriFT = FFT(x * scale),
 * imgFT = FFT(softMask(x, r, ew, bg = 0) * scale), scale = 1 / stdN.  imgRL is scaled in place; scratchRL holds
 * min(nImg, 1024) images.  Both outputs DEVICE complex64 [nImg][idim][idim/2+1]; batched rocFFT 2-D r2c. */
int thx_img_mask_normalise_fft_dev(float* imgFT, float* imgOriFT, float* imgRL, float* scratchRL, int nImg, int idim,
                                   float maskRadiusPx, float ew, float scale, void* stream);

/* The particle filter of the local search (src/Particle.cpp, src/Geometry/DirectionalStat.cpp), MODE_3D, one wave per
 * image.  State, all DEVICE doubles owned by the caller: r [nImg][nR][4] quaternions, t [nImg][nT][2] shifts (pixels),
 * wR [nImg][nR] / wT [nImg][nT] priors (Particle::_wR/_wT: the pR / pT of thx_expect_local_dev), k123 [nImg][3]
 * (Particle::_k1.._k3), s01 [nImg][2] (_s0, _s1), topR [nImg][4], topT [nImg][2].  nR, nT <= 256.
 * Randomness: Philox4x32-10 keyed by (seed, image, call, purpose, index); give every call of one run a distinct `call`.
 *
 * thx_pf_perturb_dev = Particle::perturb(pfR, PAR_R) + perturb(pfT, PAR_T), src/Particle.cpp:1149-1272: ACG
 *   perturbation of the rotations (sampleACG with pf^2 min(1, k), DirectionalStat.cpp:39-91), Gaussian perturbation of
 *   the shifts, reCentre (:2473-2495, transM = transS chi2inv_Q(transQ, 2)), balanceWeight for both (:2309-2376). */
int thx_pf_perturb_dev(double* r, double* t, double* wR, double* wT, const double* k123, const double* s01, int nImg,
                       int nR, int nT, double pfR, double pfT, double transS, double transQ, unsigned long long seed,
                       unsigned call, const int* active, void* stream);

/* thx_pf_update_dev = what follows the likelihood of a phase, src/Optimiser.cpp:1410-1475: setUR/setUT from the E-step's
 *   wR / wT (uR, uT: DEVICE floats), keepHalfHeightPeak(PAR_R) (peakFactorR < 0 disables it), calRank1st, calVari (ACG
 *   k1..k3 in the mean frame; per-column sd of the shifts), shuffle + systematic resample with PARTICLE_PRIOR_ONE
 *   (src/Particle.cpp:990-1143,1291-1480,1964-1990,2202-2300). */
int thx_pf_update_dev(double* r, double* t, double* wR, double* wT, const float* uR, const float* uT, double* k123,
                      double* s01, double* topR, double* topT, int nImg, int nR, int nT, double peakFactorR,
                      unsigned long long seed, unsigned call, const int* active, void* stream);

/* The per-image stop rule of the local search, src/Optimiser.cpp:1168-1183,1510-1615 (MODE_3D): all DEVICE arrays.
 * thx_pf_stop_init_dev: active[l] = 1, nP[l] = 0, state[l] = {k1 = k2 = k3 = 1, tVariS0 = tVariS1 = 5 transS,
 *   dVari = 5 ctfRefineS, nPhaseWithNoVariDecrease = 0} (state [nImg][8] doubles).
 * thx_pf_stop_rule_dev, after the phase with index `phase` >= MIN_N_PHASE_PER_ITER_LOCAL (3): for every active image
 *   compares k123 / s01 (and sD [nImg], the defocus sd under CTF search, NULL = 0) with the minima seen so far
 *   (PARTICLE_FILTER_DECREASE_FACTOR 0.95, squared for k1..k3), updates them, and on N_PHASE_WITH_NO_VARI_DECREASE = 1
 *   phase without a decrease sets active[l] = 0, nP[l] = phase.  *nActive (DEVICE int, zeroed by the caller) += the
 *   images still active.  thx_pf_perturb / thx_expect_local / thx_pf_update skip images with active == 0. */
int thx_pf_stop_init_dev(int* active, int* nP, double* state, double transS, double ctfRefineS, int nImg, void* stream);
int thx_pf_stop_rule_dev(int* active, int* nP, double* state, const double* k123, const double* s01, const double* sD, int phase,
                         int nImg, int* nActive, void* stream);

/* The defocus factor of the CTF search (PAR_D), d / wD [nImg][nD] support points and priors, sD [nImg] their spread:
 *   thx_pf_perturb_d_dev  init != 0: Particle::initD(nD, scale = ctfRefineS) (src/Particle.cpp:281-311, phase 0 of a CTF search,
 *                         src/Optimiser.cpp:1196-1197); init == 0: Particle::perturb(scale = perturbFactorSCTF, PAR_D) (:1273-1287,
 *                         src/Optimiser.cpp:1209); both followed by balanceWeight(PAR_D).  Philox (seed, image, call, 10, i).
 *   thx_pf_update_d_dev   after the likelihoods (uD = wD of thx_expect_local_dev): calRank1st(PAR_D) -> topD [nImg],
 *                         calVari(PAR_D) -> sD, resample(mLD, PAR_D) (src/Optimiser.cpp:1465-1470).  Philox purposes 11 / 12. */
int thx_pf_perturb_d_dev(double* d, double* wD, const double* sD, int nImg, int nD, double scale, int init, unsigned long long seed,
                         unsigned call, const int* active, void* stream);
int thx_pf_update_d_dev(double* d, double* wD, const float* uD, double* sD, double* topD, int nImg, int nD, unsigned long long seed,
                        unsigned call, const int* active, void* stream);

/* The class of every image after the global scan, src/Optimiser.cpp:925-952: uC [nImg][nK] = the scan's class weights (wC of
 * thx_expect_global_dev), wC [nImg][nK] the filter's class priors (NULL = 1 / nK) -> keepHalfHeightPeak(PAR_C) with
 * peakFactorC (PEAK_FACTOR_C = 1 - 1e-2), resample(k, PAR_C), Particle::rand(cls).  cls [nImg] out.  Philox streams
 * (seed, image, call, 6 = shuffle keys / 7 = u0 / 8 = the pick). */
int thx_pf_class_select_dev(int* cls, const float* uC, const double* wC, int nImg, int nK, double peakFactorC,
                            unsigned long long seed, unsigned call, void* stream);

/* Support points of the local search after a global scan, src/Optimiser.cpp:953-1008: for every image the scanned grid
 * (gridR [nRin][4] quaternions, gridT [nTin][2] shifts, shared by all images; uniform priors) with the scan weights of its
 * class -- uR [nK][nImg][nRin], uT [nK][nImg][nTin] as thx_expect_global_dev leaves them, cls [nImg] from
 * thx_pf_class_select_dev (NULL = class 0) -- goes through keepHalfHeightPeak(PAR_R) (peakFactorR; < 0 = off),
 * resample(mLR, PAR_R), resample(mLT, PAR_T) (src/Particle.cpp:1291-1430: shuffle, top = first largest weight, systematic draw
 * of mLR / mLT of the nRin / nTin points), calVari(PAR_R), calVari(PAR_T), and the minimum spread of the scanning phase:
 * k1..k3 = max(minK, k), s0, s1 = max(minS, s) (:1032-1079; OPTIMISER_SCAN_SET_MIN_STD_WITH_PERTURB: minK = (scanMinStdR /
 * perturbFactorSGlobal)^2 with scanMinStdR = mS^(-1/3) -- nRin^(-1/3) for C1 --, minS = scanMinStdT / perturbFactorSGlobal; 0 = none).  Out: r [nImg][mLR][4], t [nImg][mLT][2], their priors
 * wR / wT, k123 [nImg][3], s01 [nImg][2], topR [nImg][4], topT [nImg][2] -- the state thx_pf_perturb_dev continues from.
 * Philox streams (seed, image, call, 2 / 3 = rotation shuffle keys / u0, 4 / 5 = shift shuffle keys / u0).  nRin, nTin <= 16384. */
int thx_pf_scan_support_dev(double* r, double* t, double* wR, double* wT, double* k123, double* s01, double* topR, double* topT,
                            const double* gridR, const double* gridT, const float* uR, const float* uT, const int* cls, int nImg,
                            int nRin, int nTin, int mLR, int mLT, double peakFactorR, double minK, double minS, unsigned long long seed,
                            unsigned call, void* stream);

/* ---- point-group symmetry of the filter, and the numbering of its Philox streams ----
 * thx_symmetry_host = Symmetry::init(const char sym[]) (src/Geometry/Symmetry.cpp:61-278, src/Geometry/SymmetryFunctions.cpp:
 *   13-164): "C<n>", "D<n>", "T", "O", "I1" .. "I4" -> the *nSym NON-identity elements in the reference's order: symMat
 *   [nSym][9] column-major R (what SYMMETRIZE_FT / thx_symmetrize_dev take), symQuat [nSym][4] = Symmetry::quat(i) (what
 *   symmetryCounterpart multiplies with).  HOST arrays of `cap` elements (either may be NULL; both NULL: only the count).
 * thx_pf_ctx: what the *_ex_dev forms of the filter calls take on top of their plain forms (NULL = C1, img0 = 0):
 *   symQuat DEVICE [nSym][4] -- Particle::perturb(PAR_R) then ends with symmetrise(&mean) (src/Particle.cpp:1234) and
 *   Particle::calVari(PAR_R) starts with symmetrise(&anch), anch a random support point (:1030-1036; Philox purpose 13);
 *   img0 is added to the launch's image index in every Philox counter, so that an image draws the same numbers whatever
 *   batch or rank it sits in (give it the image's index in the whole data set).
 * thx_pf_symmetrise_dev = Particle::symmetrise(anchor) (:2445-2470, symmetryCounterpart src/Geometry/Symmetry.cpp:309-336) on
 *   r [nImg][nR][4]; anchor DEVICE [nImg][4] or NULL = ANCHOR_POINT_2 (1, 0, 0, 0) (what Particle::reset applies to a
 *   freshly drawn scan grid, src/Particle.cpp:168).
 * thx_pf_cal_vari_dev = calVari(PAR_R) + calVari(PAR_T) on their own (Particle::load): k123 / s01 out, r symmetrised in place. */
int thx_symmetry_host(const char* sym, double* symMat, double* symQuat, int cap, int* nSym);
typedef struct thx_pf_ctx {
    const double* symQuat;
    int nSym;
    unsigned img0;
} thx_pf_ctx;
int thx_pf_symmetrise_dev(double* r, const double* anchor, int nImg, int nR, const double* symQuat, int nSym, void* stream);
int thx_pf_cal_vari_dev(double* r, const double* t, double* k123, double* s01, int nImg, int nR, int nT, unsigned long long seed,
                        unsigned call, const thx_pf_ctx* ctx, void* stream);
int thx_pf_perturb_ex_dev(double* r, double* t, double* wR, double* wT, const double* k123, const double* s01, int nImg,
                          int nR, int nT, double pfR, double pfT, double transS, double transQ, unsigned long long seed,
                          unsigned call, const int* active, const thx_pf_ctx* ctx, void* stream);
int thx_pf_update_ex_dev(double* r, double* t, double* wR, double* wT, const float* uR, const float* uT, double* k123,
                         double* s01, double* topR, double* topT, int nImg, int nR, int nT, double peakFactorR,
                         unsigned long long seed, unsigned call, const int* active, const thx_pf_ctx* ctx, void* stream);
int thx_pf_perturb_d_ex_dev(double* d, double* wD, const double* sD, int nImg, int nD, double scale, int init, unsigned long long seed,
                            unsigned call, const int* active, unsigned img0, void* stream);
int thx_pf_update_d_ex_dev(double* d, double* wD, const float* uD, double* sD, double* topD, int nImg, int nD, unsigned long long seed,
                           unsigned call, const int* active, unsigned img0, void* stream);
int thx_pf_class_select_ex_dev(int* cls, const float* uC, const double* wC, int nImg, int nK, double peakFactorC,
                               unsigned long long seed, unsigned call, unsigned img0, void* stream);
/* rowStride: images per class row of uR / uT (nImg for the plain form; the batch's image count when the scan ran batch by batch) */
int thx_pf_scan_support_ex_dev(double* r, double* t, double* wR, double* wT, double* k123, double* s01, double* topR, double* topT,
                               const double* gridR, const double* gridT, const float* uR, const float* uT, const int* cls, int nImg,
                               int rowStride, int nRin, int nTin, int mLR, int mLT, double peakFactorR, double minK, double minS,
                               unsigned long long seed, unsigned call, const thx_pf_ctx* ctx, void* stream);

/* The deterministic ACG statistics on their own (parity probe): for quat [nImg][n][4] -> A [nImg][16] (inferACG,
 * DirectionalStat.cpp:93-145), mean [nImg][4] (:224-262), k123 [nImg][3] (calVari's mean-frame ratios), wBal [nImg][n]
 * (balanceWeight(PAR_R)), rounds [nImg][2] fixed-point rounds of the two inferACG calls (may be NULL). */
int thx_pf_acg_stats_dev(double* A, double* mean, double* k123, double* wBal, int* rounds, const double* quat, int nImg,
                         int n, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Half-set exchange in native code (RCCL over xGMI; one process per GPU)
 * ------------------------------------------------------------------------------------------- */

/* Communicator bootstrap as gpu/src/cuthunder.cu:4192-4206 does it (ncclGetUniqueId on the hemisphere's root, the id
 * shared by the caller's launcher -- MPI_Bcast in the reference --, ncclCommInitRank on every member).  id128: 128 bytes.
 * thx_comm_init binds the communicator to the CURRENT device.  A NULL thx_comm* or a communicator of size 1 makes
 * every collective below a no-op.
 * Transports.  The id decides what the communicator runs over: an RCCL id (the default and the only product path), or --
 * when the environment of the process that DRAWS the id holds THX_COMM_TRANSPORT=shm -- the name of a POSIX shared-memory
 * segment.  The shm transport is TEST-ONLY: every collective becomes device -> host copy, barrier, rank-ordered host sum,
 * host -> device copy; it exists because RCCL refuses two ranks on one device, so that 2 or 4 processes sharing one GPU can
 * run the unchanged multi-rank branches of thx_refine_iterate and be held to the one-rank result
 * (tests/test_multirank_gpu.py).  thx_comm_transport names the transport of a communicator ("rccl", "shm", "none"). */
typedef struct thx_comm thx_comm;
int thx_comm_unique_id(void* id128);
int thx_comm_init(thx_comm** out, const void* id128, int rank, int size);
int thx_comm_destroy(thx_comm* c);
int thx_comm_rank(const thx_comm* c);
int thx_comm_size(const thx_comm* c);
const char* thx_comm_transport(const thx_comm* c);
/* in-place collectives on DEVICE buffers, enqueued on `stream` */
int thx_comm_allreduce_f32(thx_comm* c, float* buf, size_t count, void* stream);
int thx_comm_allreduce_f64(thx_comm* c, double* buf, size_t count, void* stream);
int thx_comm_allreduce_i32(thx_comm* c, int* buf, size_t count, void* stream);
int thx_comm_allreduce_i64(thx_comm* c, long long* buf, size_t count, void* stream);
int thx_comm_allreduce_max_f64(thx_comm* c, double* buf, size_t count, void* stream);
/* ncclReduce: only `root` receives the sums; the other ranks' buffers are left as they were */
int thx_comm_reduce_i64(thx_comm* c, long long* buf, size_t count, int root, void* stream);
int thx_comm_broadcast(thx_comm* c, void* buf, size_t bytes, int root, void* stream);

/* Reconstructor::allReduceF / allReduceT (src/Reconstructor.cpp:2350-2484, MPI_Allreduce_Large over _hemi) and the
 * reference GPU path's ncclAllReduce of F, T, O, counter (gpu/src/cuthunder.cu:4972-5067): sums the accumulators over
 * the ranks of the half, in place.  Only the voxels inside the sample sphere (radius maxRadius * pf + 2: inserted samples
 * lie inside maxRadius * pf, their trilinear cells reach one further) travel: F (re, im) and T of those voxels are packed
 * into ONE contiguous buffer (half of the grid), reduced by ONE ring all-reduce, and unpacked; voxels outside keep their
 * local values (zero after insertion).  O (3 doubles) and counter (1 int) may be NULL.
 * workspace: thx_reco_allreduce_workspace() bytes of device memory. */
size_t thx_reco_allreduce_workspace(int dim, int maxRadius, int pf);
int thx_reco_allreduce(thx_comm* hemi, float* F, float* T, double* O, int* counter, int dim, int maxRadius, int pf,
                       void* workspace, void* stream);
/* the same reduce on the 64-bit fixed-point accumulators of an insertion session (thx_insert_accumulate_dev), BEFORE
 * thx_insert_finish_dev: integer sums are associative, so N ranks end with bit for bit the volumes one rank would have
 * accumulated (gpu/src/cuthunder.cu:4972-5067 reduces floats).  workspace: thx_reco_allreduce_acc_workspace() bytes. */
size_t thx_reco_allreduce_acc_workspace(int dim, int maxRadius, int pf);
int thx_reco_allreduce_acc(thx_comm* hemi, void* acc, double* O, int* counter, int dim, int maxRadius, int pf, void* workspace,
                           void* stream);
/* class k of a session over nK classes (acc as thx_insert_acc_bytes(dim, nK) lays it out); thx_reco_allreduce_acc is nK = 1, k = 0 */
int thx_reco_allreduce_acc_class(thx_comm* hemi, void* acc, int nK, int k, double* O, int* counter, int dim, int maxRadius, int pf,
                                 void* workspace, void* stream);
/* the same towards ONE rank of the half (root >= 0: ncclReduce; root < 0: all-reduce): the rank that reconstructs class k is
 * the only one that needs the sums (thx_refine_iterate with owners, src/Reconstructor.cpp:2383,2436 reduce to everybody); the
 * other ranks' accumulators of the class keep their partial sums */
int thx_reco_reduce_acc_class(thx_comm* hemi, void* acc, int nK, int k, int root, int dim, int maxRadius, int pf, void* workspace,
                              void* stream);
/* the pack (unpack = 0) / unpack (unpack = 1) halves of thx_reco_allreduce on their own (parity probe of the sphere-row
 * tables on one GPU); *nVoxOut (host, optional) = packed voxels */
int thx_reco_sphere_pack_dev(float* F, float* T, int dim, int maxRadius, int pf, void* workspace, int unpack, long* nVoxOut,
                             void* stream);

/* ---------------------------------------------------------------------------------------------
 * The iteration driver in native code: Optimiser::expectation (src/Optimiser.cpp:1141-1660, local phases) +
 * Optimiser::maximization (src/Optimiser.cpp:3405-3480: allReduceSigma :6395-6710, reconstructRef :6711-7766) +
 * Model::compareTwoHemispheres / refreshProj + reCentreImg / reMaskImg (:6065-6149), sequenced on the host side of this
 * library over one rank's HBM-resident shard of particles.  What bench.py times and tests/cpp/iteration.cpp drives.
 * ------------------------------------------------------------------------------------------- */

/* The library caches, per (device, stream), grow-only scratch buffers and hipFFT plans (a plan carries its stream).  A caller
 * that destroys a stream calls this first: it synchronises the stream, frees that stream's scratch and destroys its plans,
 * so that a recycled handle value can never meet a plan bound to a dead stream. */
int thx_release_stream(void* stream);

/* Optimiser::allocPreCalIdx (src/Optimiser.cpp:7991-8041) on the host: the half-plane pixel list rL <= |k| < rU.
 * order: 0 = the reference's row-major order, 1 = Morton (Z-order) visit order (the E-step's default).  Output arrays
 * (may be NULL) hold at least (rU + 2) * (2 rU + 2) ints; *nPxl receives the count. */
int thx_pixel_list_host(int N, int rU, int rL, int order, int* iCol, int* iRow, int* iPxl, int* iSig, int* nPxl);

/* Shard layout: perm [n] (HOST) orders n particles along a Morton curve over their view direction (the normal R e_z of the
 * central slice of quat [n][4], folded to one hemisphere, equal-area map): a caller that stores its particles in this
 * order -- by the previous iteration's top rotation -- puts images that cut the volume along nearly the same plane next to
 * each other in every launch, and the local-search kernel's gathers hit L2 / Infinity Cache more often (-8 % at 100 k
 * particles).  Results do not depend on the order. */
int thx_view_order_host(const double* quat, int n, int* perm);

/* The mReco draws of the insertion (src/Optimiser.cpp:7129-7150) from a RESAMPLED filter (thx_pf_update_dev): uniform
 * picks among the support points r [nImg][nR][4], t [nImg][nT][2] (Particle::rand, src/Particle.cpp:2109-2178) ->
 * recoRot [nImg][mReco][9], recoTran [nImg][mReco][2].  Philox stream (seed, img0 + image, call, 7, draw). */
int thx_draw_reco_dev(double* recoRot, double* recoTran, const double* r, const double* t, int nImg, int nR, int nT, int mReco,
                      unsigned long long seed, unsigned call, unsigned img0, void* stream);

#define THX_SEARCH_LOCAL 0    /* SEARCH_TYPE_LOCAL  (include/Optimiser.h) */
#define THX_SEARCH_GLOBAL 1   /* SEARCH_TYPE_GLOBAL: every image is first scanned against all classes x nR rotations x nT shifts */
#define THX_SEARCH_CTF 2      /* SEARCH_TYPE_CTF: local search with mLD defocus factors per image */

typedef struct thx_refine thx_refine;
typedef struct thx_refine_config {
    int N, pf;                 /* box size, padding factor */
    int nImg;                  /* particles of this rank */
    int halfOfRank;            /* -1: both half-sets live on this rank ([0, nHalfA) = half 0, the rest = half 1);
                                  0 / 1: the whole shard belongs to that half (odd / even ranks, src/Parallel.cpp:26-36) */
    int nHalfA;
    int mLR, mLT, nPhase, mReco;  /* script/demo_3D.json: 125, 9, 3 (MIN_N_PHASE_PER_ITER_LOCAL; _GLOBAL is 10), 100 */
    int maxPhase;              /* <= nPhase: exactly nPhase phases per image (the fixed-work iteration bench.py times);
                                  > nPhase: the reference's per-image stop rule from phase index nPhase on, at most maxPhase
                                  phases (MAX_N_PHASE_PER_ITER = 100, include/Optimiser.h:58) */
    int batch;                 /* images per kernel launch, at most */
    int rL;                    /* lower cut-off of the E-step pixel list (Optimiser::_rL) */
    int nGroup, groupSig;      /* micrograph groups, OPTIMISER_SIGMA_GROUP */
    int pixelOrder;            /* E-step pixel list: 0 row-major, 1 Morton */
    int wgPerCU;               /* occupancy argument of thx_expect_local_dev */
    float pixelSize, maskRadiusPx, sigma2Init;
    double transS, transQ;     /* Optimiser::_para.transS, TRANS_Q (include/Optimiser.h:67) */
    double pfL, pfS;           /* perturbation factors: phase 0 of a local / CTF search (perturbFactorL), the later phases
                                  (perturbFactorSLocal) (script/demo_3D.json:71-73) */
    double peakFactorR;        /* PEAK_FACTOR_MIN */
    unsigned long long seed;
    /* Model::compareTwoHemispheres / Optimiser::solventFlatten between the reconstructions and Model::refreshProj: */
    int coreFSC;               /* "Calculate FSC Using Core Region" (script/demo_3D.json:23: true): mask-corrected FSC with the
                                  core mask of radius AROUND(maskRadius / pixelSize) (src/Optimiser.cpp:188, src/Model.cpp:411-563) */
    int goldenAverage;         /* != 0: compareTwoHemispheres(false, true, AVERAGE_TWO_HEMISPHERE_THRES) after the MAP reconstruction
                                  (src/Optimiser.cpp:7747).  One class and "Using Golden Standard FSC": the two references are
                                  averaged inside r = Model::resolutionP(0.95) of THIS iteration's FSC (MODEL_RESOLUTION_BASE_AVERAGE,
                                  include/Config.h:129-131, src/Model.cpp:616-674); several classes: averaged everywhere (:688-696) */
    int solventFlatten;        /* != 0: Optimiser::solventFlatten's spherical soft mask (maskRadius / pixelSize, EDGE_WIDTH_RL,
                                  background 0; src/Optimiser.cpp:7768-7990, no provided mask) before the projector refresh */
    int normCorrection;        /* != 0: Optimiser::normCorrection (OPTIMISER_NORM_CORRECTION, src/Optimiser.cpp:3405-3413,6201-6394) at
                                  the head of the M-step of every iteration but the first and not after a global search: both image
                                  stacks -- the driver's _img and the caller's _imgOri, IN PLACE -- are rescaled image by image to the
                                  median residual power over ALL particles (one all-reduce of the norm vector over `world`); the
                                  expectations of all local halves then run before the first M-step */
    /* ---- K references (classification) and the global search; 0 in every field = the one-class local search ---- */
    int nK;                    /* classes (<= 16); 0 = 1 */
    int searchType;            /* THX_SEARCH_* of the next iteration (Optimiser::_searchType); thx_refine_set_search_type changes it */
    int nR, nT;                /* the scanned grid: nR = mS / (1 + nSym) rotations (src/Optimiser.cpp:652), nT >= 30 shifts (:663-667);
                                  0 = the handle never scans (no scan buffers) */
    int rScan;                 /* frequency limit of the scan: the LARGEST radius a global search of this handle scans at (its buffers are
                                  sized for that list); the scan of an iteration runs at min(rScan, r) with r the cut-off of
                                  thx_refine_set_cutoff -- r at Nyquist: SURVEY 8d's configs[3] (scan at rScan, local phases on the full
                                  list); rScan >= r: the reference's own form, scan and local phases both on allocPreCalIdx(_r, _rL) */
    int scanBatch;             /* images per scan launch, at most (0 = 2048): workspace nImg * nR * nT floats */
    double pfSGlobal;          /* perturbFactorSGlobal: EVERY local phase of a global-search iteration (phase index starts at 1,
                                  src/Optimiser.cpp:1185-1212; OPTIMISER_GLOBAL_PERTURB_LARGE is off) */
    double peakFactorC;        /* PEAK_FACTOR_C = 1 - 1e-2 */
    double scanMinK, scanMinS; /* minimum spread after the scan (thx_pf_scan_support_dev's minK / minS) */
    int balanceClass;          /* != 0: OPTIMISER_BALANCE_CLASS (src/Optimiser.cpp:5518-5593,7510-7523,7727-7733): after a global search
                                  a class holding less than CLASS_BALANCE_FACTOR / K of the images takes over the reference of a class
                                  drawn from the distribution of the others (Philox stream (seed, 0, call, 14, class)).  Every rank
                                  evaluates that stream itself: `seed` must be the same number on all ranks
                                  (thx_refine_set_particles checks it over `world`) */
    /* ---- point group (thx_symmetry_host): HOST arrays, copied at create; nSym = 0: C1 ---- */
    int nSym;
    const double* symMat;      /* [nSym][9] column-major: prepareTF symmetrises T, then F (src/Reconstructor.cpp:1056-1091) */
    const double* symQuat;     /* [nSym][4]: Particle::symmetrise in perturb / calVari (thx_pf_ctx); set_grid folds the scan grid */
    /* ---- CTF search (THX_SEARCH_CTF) ---- */
    int mLD;                   /* defocus factors per image (script/demo_3D.json: 9); 0 = the handle never runs a CTF search */
    double ctfRefineS, pfSCTF; /* "CTF Refine Standard Deviation" (0.01), "Perturbation Factor (Small, CTF)" (0.5) */
} thx_refine_config;

/* optional per-phase trace of the local search (tests hold the chain against the oracle with it): DEVICE buffers, any may
 * be NULL; index [phase][image of this rank] (phase = the count of phases run, 0-based, also after a scan).  uR / uT: the
 * E-step's weights as handed to the filter (Particle::setUR / setUT); rP / tP: the support points after Particle::perturb,
 * r / t: after Particle::resample; k123 / s01: Particle::calVari of the phase. */
typedef struct thx_refine_capture {
    float *uR, *uT;            /* [nPhase][nImg][mLR], [nPhase][nImg][mLT] */
    double *r, *t;             /* [nPhase][nImg][mLR][4], [nPhase][nImg][mLT][2] */
    double *k123, *s01;        /* [nPhase][nImg][3], [nPhase][nImg][2] */
    float *mapsFsc;            /* [2][nK][N]^3: the MAP-off half maps the FSC is computed from */
    double *rP, *tP;           /* [nPhase][nImg][mLR][4], [nPhase][nImg][mLT][2]: the support points after Particle::perturb */
    double *wRP, *wTP;         /* [nPhase][nImg][mLR], [nPhase][nImg][mLT]: their priors (Particle::balanceWeight) */
    float *Fraw, *Traw;        /* [local halves][nK] complex / real half grids of the reconstructors' CURRENT size, (pf size)^3 (= (pf N)^3 at
                                  Nyquist; thx_refine_set_cutoff), contiguous from the start of the buffer -- size the buffers for (pf N)^3:
                                  the accumulators as the insertion session left them (after the half-set reduce on the integers), before
                                  prepareTF's normalisation */
    /* global search: the scan's weights and what the filter made of them */
    float *scanUC, *scanUR, *scanUT;   /* [nImg][nK], [nImg][nK][nR], [nImg][nK][nT] */
    double *r0, *t0;           /* [nImg][mLR][4], [nImg][mLT][2]: the support points as thx_pf_scan_support_dev left them */
    double *k0, *s0;           /* [nImg][3], [nImg][2]: their spread (after the scanning phase's minimum) */
    float *Fsym, *Tsym;        /* [local halves][nK]: F / T after prepareTF (normalised, symmetrised), before the Wiener term */
    /* CTF search: the defocus factors of every phase */
    float *uD;                 /* [nPhase][nImg][mLD]: the E-step's weights of the defocus factors (Particle::setUD) */
    double *dP, *dR;           /* [nPhase][nImg][mLD]: the factors after initD / perturb(PAR_D), and after resample(mLD, PAR_D) */
    int phases;                /* leading dimension of the per-phase arrays above (0: cfg.nPhase); with the per-image stop rule
                                  (maxPhase > nPhase) a trace of maxPhase phases follows every image to the phase it stops in --
                                  the rows of an image at phases after the one it stopped in are UNDEFINED (whole batches of the
                                  per-batch scratch are copied; view.nP tells where every image stopped) */
} thx_refine_capture;

typedef struct thx_refine_stats {
    double expectMs, insertMs;            /* HIP-event totals of the local-search / insertion launches (timed iterations) */
    long expectLaunches, expectImages, insertLaunches, insertImages;
    double stageMs[8];                    /* rows, expectation (local phases), sigma, insertion, reconstruct (+FSC, refresh),
                                             recentre+remask, normCorrection, global scan (+ class selection, support points) */
    long balancingRounds, iterations;
    long imagePhases;                     /* sum over images of the phases they ran (Optimiser::_nF) */
    int nPxl, nPxlM, batch;
    unsigned long long insertGroups;      /* (image, group) pairs the insertion launches of the timed iterations processed */
    int lastRounds[4];                    /* balancing rounds of the last iteration's reconstructions of class 0: MAP off, local
                                             halves 0 / 1; MAP on, local halves 0 / 1 (0 where this rank holds no such half) */
    float normMedian, normRadius;         /* normCorrection of the last iteration: the median of the norms and rNorm (0 when it did not run) */
    double scanMs;                        /* HIP-event total of the thx_expect_global_dev launches */
    long scanLaunches, scanImages;        /* (a launch = one class against one batch of images) */
    int nPxlS, nK;
    int classCount[16];                   /* images of this rank per class after the last expectation */
    int lastRoundsK[64];                  /* [MAP off / on][local half][class] of the last iteration */
    int balanced[16];                     /* balanceClass of the last iteration: class t took over the reference of balanced[t] (-1: kept its own) */
} thx_refine_stats;

/* hemi: communicator of this rank's half (NULL = the half lives on this rank alone); world: all ranks (NULL = one rank) */
int thx_refine_create(thx_refine** out, const thx_refine_config* cfg, thx_comm* hemi, thx_comm* world);
int thx_refine_destroy(thx_refine* h);
/* imgOri DEVICE complex64 [nImg][N][N/2+1] as Optimiser::initImg leaves _imgOri -- BORROWED until destroy;
 * attr DEVICE [nImg]; groupID HOST [nImg] 1-based; quat0 DEVICE [nImg][mLR][4], tran0 DEVICE [nImg][mLT][2] initial
 * support points of the particle filter (Particle::load). */
int thx_refine_set_particles(thx_refine* h, const float* imgOri, const thx_ctf_attr* attr, const int* groupID_host,
                             const double* quat0, const double* tran0, void* stream);
/* The Philox number of this rank's first image (default, -1: the particles of the world ranks before it).  A caller that
 * deals the particles of a job to the ranks in any other way -- e.g. each half's particles in contiguous parts, so that an
 * N-rank run draws for every image exactly what a one-rank run of the same job draws -- says where its shard starts.  Call
 * before thx_refine_reset. */
int thx_refine_set_image_base(thx_refine* h, long long base);
int thx_refine_set_reference(thx_refine* h, const float* refRL, void* stream);   /* DEVICE [nK][N]^3 initial maps */
/* the class every image starts a LOCAL search in (Particle::c of the loaded filter; a global search assigns its own):
 * DEVICE or HOST [nImg] ints in [0, nK); without this call every image is in class 0 */
int thx_refine_set_classes(thx_refine* h, const int* cls, void* stream);
/* the scanned grid of a global search: quat [nR][4], shifts [nT][2] doubles (host or device), shared by all images --
 * Particle::reset(k, nR, nT, 1) (src/Particle.cpp:60-168) draws them and, with a point group, symmetrise()s the rotations
 * next to ANCHOR_POINT_2: the fold happens here */
int thx_refine_set_grid(thx_refine* h, const double* quat, const double* shifts, void* stream);
int thx_refine_set_search_type(thx_refine* h, int searchType);
/* The frequency cut-offs of the NEXT iteration -- per-iteration inputs, as Model::updateR / updateRU leave them before it starts
 * (the schedule is the caller's, SURVEY 2 out of scope).  Until the first call: r = rU = N / 2 - 2 (Nyquist, _size = _N).
 *   r  = Optimiser::_r  (rL < r <= N / 2 - 1): the expectation's pixel list allocPreCalIdx(_r, _rL) (src/Optimiser.cpp:631,1693);
 *        a global search scans on min(cfg.rScan, r); Projector::_maxRadius = _r (Model::refreshProj, src/Model.cpp:1042) ends the
 *        slices of allReduceSigma and normCorrection; normCorrection's rNorm = min(_r, resolutionP(0.75)) (:6203).
 *   rU = Model::_rU     (0 < rU <= N / 2 - 1): the reconstruction's list allocPreCalIdx(rU, 0) (:6722-6741),
 *        Reconstructor::setMaxRadius(rU) and resizeSpace(min(N, (rU + ceil(a)) * 2)) (Model::resetReco, src/Model.cpp:1100-1125;
 *        src/Reconstructor.cpp:184-198): insertion, the half-set reduce, prepareTF and the gridding loop run on the (pf size)^3 grid,
 *        the last step of reconstruct pads F W into (pf N)^3 (:1677-1701); compareTwoHemispheres' FSC has rU shells, and the
 *        curve handed to the NEXT iteration's MAP reconstruction keeps this iteration's rU entries (Model::resetReco's
 *        setFSC(_FSC.col(l)); shells beyond count as FSC 0, src/Reconstructor.cpp:1244-1247).
 * Re-cuts the M-step rows, both CTF row sets [, the defocus search's rows, the scan's ramps]; rebuilds the reconstruction plans when
 * the size changes (with several ranks EVERY rank of the job makes the same call: the half-set reduce and the owners' plans follow
 * the size); before the first iteration it also resets the reconstructor's FSC to rU ones (Model::initProjReco,
 * src/Model.cpp:1086).  Synchronises the stream.  Capture buffers Fraw / Traw / Fsym / Tsym are filled as [local halves][nK]
 * volumes of the CURRENT (pf size)^3 half grid, contiguous. */
int thx_refine_set_cutoff(thx_refine* h, int r, int rU, void* stream);
/* any pointer may be NULL: r, rU as set; size = Reconstructor::_size; rScan = the radius a global search scans at */
int thx_refine_get_cutoff(const thx_refine* h, int* r, int* rU, int* size, int* rScan);
int thx_refine_reset(thx_refine* h, void* stream);     /* the state before the first iteration */
/* one EM iteration in the reference's order (src/Optimiser.cpp:3595-4073): expectation ([global scan -> class -> support
 * points ->] local phases), [normCorrection,] allReduceSigma, insertion into the F / T of every image's class, prepareTF
 * (reduce, normalise, symmetrise T, symmetrise F), per class reconstruct (MAP off) -> [balanceClass] -> compareTwoHemispheres
 * (FSC of THIS iteration, Model::_FSC), reconstruct (MAP on, joinHalf: OPTIMISER_RECONSTRUCT_JOIN_HALF) with the FSC
 * Model::resetReco handed to the reconstructor at the end of the PREVIOUS iteration (all ones before the first:
 * src/Model.cpp:1086,1122) -> [balanceClass] -> averaging of the two halves, solvent flattening, Model::refreshProj; reCentreImg /
 * reMaskImg unless the search was global (:3790-3800).  fscHost (optional) [max(nK, 1)][N/2] receives this iteration's FSC per
 * class (rU = N/2 - 2 shells, the rest 0); timed != 0 records HIP events for thx_refine_stats.  Synchronises the stream (FSC
 * to the host, the gridding loop's stop rule).
 * Philox numbering: image = the image's index over all ranks (ranks in world order), call = iteration * 1024 + slot with slot
 * 1 = class selection, 2 = support points after the scan, 8 + 2 p / 9 + 2 p = Particle::perturb / the filter update of phase
 * index p, 1000 = the insertion's draws, 1001 = balanceClass; thx_refine_reset's calVari uses call 3. */
int thx_refine_iterate(thx_refine* h, float* fscHost, int timed, void* stream);
/* capture (copied; NULL = off): where the following iterations leave their per-phase trace */
int thx_refine_set_capture(thx_refine* h, const thx_refine_capture* capture);
/* DEVICE [N]^3: the reference of `half` (class k) as Model::refreshProj consumed it (MAP-on map, averaged / flattened) */
int thx_refine_get_map(thx_refine* h, int half, float* dstRL, void* stream);
int thx_refine_get_map_k(thx_refine* h, int half, int k, float* dstRL, void* stream);
/* DEVICE copies of the per-particle state (any pointer may be NULL): offset [nImg][2], topR [nImg][4], topT [nImg][2],
 * sig [local halves][nGroup][N/2-1] */
int thx_refine_get_state(thx_refine* h, double* offset, double* topR, double* topT, float* sig, void* stream);
int thx_refine_get_stats(thx_refine* h, thx_refine_stats* out, int reset);
/* read-only DEVICE views of the handle's resident state (valid until destroy; contents change with every iteration):
 * what tests compare against the oracle and bench.py's cpu_baseline leg copies its sample from.  Volumes are indexed
 * [local half][class]. */
typedef struct thx_refine_view {
    int nImg, nPxl, nPxlM, nVol, vdim, rSig;
    const int *iCol, *iRow, *iPxl, *iSig, *iColM, *iRowM; /* E-step list [nPxl] (in visit order), M-step list [nPxlM] */
    const float *img;                       /* [nImg][N][N/2+1] complex64: re-centred, masked images (_img) */
    const float *datP, *ctfP, *sigRcpP;     /* E-step rows [nImg][nPxl] (complex64 / f32 / f32) */
    const float *datM, *ctfM;               /* M-step rows [nImg][nPxlM] */
    const double *r, *t, *wR, *wT;          /* particle filter: [nImg][mLR][4], [nImg][mLT][2], priors */
    const double *offset;                   /* [nImg][2] */
    const float *vols, *cells;              /* [nVol] projector FTs / their cell-packed copies */
    const float *F, *T;                     /* [nVol] accumulators of the last iteration (after prepareTF / Wiener term): fdim^3 half grids, contiguous */
    const float *sig;                       /* [local halves][nGroup][rSig] */
    const double *recoRot, *recoTran;       /* draws of the LAST inserted local half [n][mReco][9] / [n][mReco][2] */
    const int *nP;                          /* [nImg] phase index at which the stop rule ended the image's search (maxPhase > nPhase) */
    const float *norm;                      /* [nImg] normCorrection's norms of the last iteration that ran it */
    const int *cls;                         /* [nImg] class of every image */
    const double *topR, *topT, *k123, *s01; /* [nImg][4], [nImg][2], [nImg][3], [nImg][2] */
    const double *d, *wD;                   /* [nImg][mLD] defocus factors and their priors (CTF search; NULL without) */
    const float *maps, *mapsMAP;            /* [2][nK][N]^3 MAP-off / final maps of the last iteration */
    int nK, nPxlS;
    int fdim;                               /* grid of F / T: pf * Reconstructor::_size (= vdim at Nyquist; thx_refine_set_cutoff) */
} thx_refine_view;
int thx_refine_get_view(thx_refine* h, thx_refine_view* out);

/* ---------------------------------------------------------------------------------------------
 * Interface.h-shaped HOST-pointer entry points (what -DGPU_VERSION call sites bind to)
 * ------------------------------------------------------------------------------------------- */

/* void ExpectProject(Complex* volume, Complex* rotP, double* rotMat, const int* iCol, const int* iRow,
 *                    int nR, int pf, int interp, int vdim, int npxl)            Interface.h:210-219 */
int thx_ExpectProject_host(const float* volume, float* rotP, const double* rotMat, const int* iCol, const int* iRow,
                           int nR, int pf, int interp, int vdim, int npxl);

/* void ExpectRotran(Complex* traP, double* trans, double* rot, double* rotMat, const int* iCol, const int* iRow,
 *                   int nR, int nT, int idim, int npxl)                          Interface.h:199-208
 * rot = quaternions [nR][4] in, rotMat [nR][9] out, traP [nT][npxl] out. */
int thx_ExpectRotran_host(float* traP, const double* trans, const double* rot, double* rotMat, const int* iCol,
                          const int* iRow, int nR, int nT, int idim, int npxl);

/* void InsertFT(Volume& F3D, Volume& T3D, double* O3D, int* counter, MPI_Comm& hemi, MPI_Comm& slav, Complex* datP,
 *               RFLOAT* ctfP, RFLOAT* sigRcpP, CTFAttr* ctfaData, double* offS, RFLOAT* w, double* nR, double* nT,
 *               double* nD, [int* nC,] const int* iCol, const int* iRow, RFLOAT pixelSize, bool cSearch, int opf,
 *               int npxl, int mReco, int idim, int dimSize, int imgNum)          Interface.h:267-318
 * F3D/T3D are passed as the raw half-complex arrays (&F3D[0], complex T with unused imaginary part as the
 * reference holds it); nR are QUATERNIONS [imgNum][mReco][4] as in the reference.  The MPI communicators of
 * the reference signature are replaced by the caller doing the hemisphere reduction (thx has no MPI dependency):
 * this call accumulates the LOCAL contribution into F3D/T3D/O3D/counter. */
int thx_InsertFT_host(float* F3D, float* T3D_complex, double* O3D, int* counter, const float* datP, const float* ctfP,
                      const thx_ctf_attr* ctfaData, const double* offS, const float* w, const double* nR,
                      const double* nT, const double* nD, const int* nC, const int* iCol, const int* iRow,
                      float pixelSize, int cSearch, int opf, int npxl, int mReco, int idim, int vdim, int nK,
                      int imgNum);

/* InsertFT with the hemisphere reduction INSIDE, as the reference's GPU build has it (ncclAllReduce of F, T, O, counter
 * over commF, gpu/src/cuthunder.cu:4972-5067): the MPI_Comm& hemi of the reference signature becomes the RCCL communicator
 * of the hemisphere (thx_comm_init).  F3D / T3D must hold only THIS rank's contribution on entry (the reference resets
 * them before insertion); on return every rank of the hemisphere holds the sums.  O3D / counter: this rank's running
 * values += the hemisphere's sums of this call.  maxRadius: Reconstructor::_maxRadius (the reduced sphere). */
int thx_InsertFT_hemi_host(thx_comm* hemi, int maxRadius, float* F3D, float* T3D_complex, double* O3D, int* counter,
                           const float* datP, const float* ctfP, const thx_ctf_attr* ctfaData, const double* offS, const float* w,
                           const double* nR, const double* nT, const double* nD, const int* nC, const int* iCol,
                           const int* iRow, float pixelSize, int cSearch, int opf, int npxl, int mReco, int idim, int vdim,
                           int nK, int imgNum);

/* void PrepareTF(int gpuIdx, Volume& F3D, Volume& T3D, double* symMat, int nSymmetryElement, int maxRadius, int pf)
 *                                                                                Interface.h:320-326
 * normalise by 1/T[0] + symmetrise both, on host arrays (T complex as the reference holds it). */
int thx_PrepareTF_host(int gpuIdx, float* F3D, float* T3D_complex, int vdim, const double* symMat,
                       int nSymmetryElement, int maxRadius, int pf);

/* Reconstructor::reconstructG(Volume& dst, int gpuIdx, nThread) (src/Reconstructor.cpp:1835-2315: ExposePT, ExposeWT,
 * ExposePFW, ExposeCorrF chained) on host arrays: F3D complex, T3D complex, dst real [N]^3. */
int thx_ReconstructG_host(int gpuIdx, const float* F3D, const float* T3D_complex, int size, int N, int pf,
                          int maxRadius, float a, float alpha, const float* FSC, int nFSC, int joinHalf, int MAP,
                          int gridCorr, float* dstRL);

/* void ReMask(vector<Image>& img, RFLOAT maskRadius, RFLOAT pixelSize, RFLOAT ew, int idim, int imgNum)
 *                                                                                Interface.h:517-522
 * imgFT = imgNum host pointers (&img[l][0]), each idim*(idim/2+1) complex64, rewritten in place. */
int thx_ReMask_host(float* const* imgFT, float maskRadius, float pixelSize, float ew, int idim, int imgNum);

/* void TranslateI2D(int gpuIdx, Image& img, double ox, double oy, int r)        Interface.h:504-508 */
int thx_TranslateI2D_host(int gpuIdx, float* imgFT, double ox, double oy, int r, int idim);

/* void TranslateI(int gpuIdx, Volume& ref, double ox, double oy, double oz, int r)  Interface.h:510-515 */
int thx_TranslateI_host(int gpuIdx, float* volFT, double ox, double oy, double oz, int r, int dim);

/* void ExpectPrecal(vector<CTFAttr>& ctfAttr, RFLOAT* def, RFLOAT* k1, RFLOAT* k2, const int* iCol, const int* iRow,
 *                   int idim, int npxl, int imgNum)                              Interface.h:166-174
 * ctfAttr = &ctfAttr[0] (imgNum contiguous CTFAttr); pixelSize is not a parameter of the reference call because def
 * does not depend on it. */
int thx_ExpectPrecal_host(const thx_ctf_attr* ctfAttr, float* def, float* k1, float* k2, const int* iCol,
                          const int* iRow, int idim, int npxl, int imgNum);

/* void ExpectGlobal3D(Complex* rotP, Complex* traP, Complex* datP, RFLOAT* ctfP, RFLOAT* sigRcpP, RFLOAT* wC,
 *                     RFLOAT* wR, RFLOAT* wT, double* pR, double* pT, RFLOAT* baseL, int kIdx, int nK, int nR, int nT,
 *                     int npxl, int imgNum)                                      Interface.h:221-237
 * layouts as thx_expect_global_dev; wC/wR/wT/baseL are read-modify-write host arrays. */
int thx_ExpectGlobal3D_host(const float* rotP, const float* traP, const float* datP, const float* ctfP,
                            const float* sigRcpP, float* wC, float* wR, float* wT, const double* pR, const double* pT,
                            float* baseL, int kIdx, int nK, int nR, int nT, int npxl, int imgNum);

/* void GCTFinit(vector<Image>& img, vector<CTFAttr>& ctfAttr, RFLOAT pixelSize, int idim, int imgNum)
 *                                                                                Interface.h:524-528
 * ctfFT = imgNum host pointers (&_ctf[l][0]), each idim*(idim/2+1) complex64. */
int thx_GCTFinit_host(float* const* ctfFT, const thx_ctf_attr* ctfAttr, float pixelSize, int idim, int imgNum);

/* ---------------------------------------------------------------------------------------------
 * The remaining 3-D entry points of gpu/interface/Interface.h, at the reference's own (per-image / per-stage)
 * granularity, so that a replacement Interface.cpp forwards one to one and src/Optimiser.cpp / src/Reconstructor.cpp
 * stay untouched.  They run the same kernels as the batched calls above on one image / one stage at a time;
 * use the batched calls for throughput.  2-D mode twins (ExpectGlobal2D, ExpectLocalV2D/PreI2D, InsertI2D,
 * ExposePT2D/WT2D/PF2D/CorrF2D) are out of scope.
 * ------------------------------------------------------------------------------------------- */

/* ManagedArrayTexture (gpu/include/ManagedArrayTexture.h: Init(mode, vdim, gpuIdx), getDeviceId) and ManagedCalPoint
 * (gpu/include/ManagedCalPoint.h: Init(mode, cSearch, gpuIdx, nR, nT, mD, npxl)) as opaque handles; mode 1 = MODE_3D. */
typedef struct thx_texture thx_texture;
typedef struct thx_calpoint thx_calpoint;
int thx_texture_create(thx_texture** out, int mode, int vdim, int gpuIdx);
int thx_texture_destroy(thx_texture* t);
int thx_texture_device(const thx_texture* t);
int thx_calpoint_create(thx_calpoint** out, int mode, int cSearch, int gpuIdx, int nR, int nT, int mD, int npxl);
int thx_calpoint_destroy(thx_calpoint* c);

/* void ExpectPreidx(int gpuIdx, int** deviCol, int** deviRow, int* iCol, int* iRow, int npxl)      Interface.h:18-23 */
int thx_ExpectPreidx_host(int gpuIdx, int** deviCol, int** deviRow, const int* iCol, const int* iRow, int npxl);
/* void ExpectPrefre(int gpuIdx, RFLOAT** devfreQ, RFLOAT* freQ, int npxl)                           Interface.h:26-29 */
int thx_ExpectPrefre_host(int gpuIdx, float** devfreQ, const float* freQ, int npxl);
/* void ExpectLocalIn(gpuIdx, Complex** devdatP, RFLOAT** devctfP, RFLOAT** devdefO, RFLOAT** devsigP, nPxl, cpyNumL,
 *                    searchType)                                                                    Interface.h:31-38 */
int thx_ExpectLocalIn_host(int gpuIdx, float** devdatP, float** devctfP, float** devdefO, float** devsigP, int nPxl,
                           int cpyNumL, int searchType);
/* void ExpectLocalV3D(int gpuIdx, ManagedArrayTexture* mgr, Complex* volume, int vdim)              Interface.h:45-48 */
int thx_ExpectLocalV3D_host(int gpuIdx, thx_texture* mgr, const float* volume, int vdim);
/* void ExpectLocalP(gpuIdx, devdatP, devctfP, devdefO, devsigP, datP, ctfP, defO, sigP, threadId, imgId, npxl, cSearch)
 *                                                                                                   Interface.h:50-62 */
int thx_ExpectLocalP_host(int gpuIdx, float* devdatP, float* devctfP, float* devdefO, float* devsigP, const float* datP,
                          const float* ctfP, const float* defO, const float* sigP, int threadId, int imgId, int npxl,
                          int cSearch);
/* void ExpectLocalHostA(gpuIdx, RFLOAT** wC, wR, wT, wD, double** oldR, oldT, oldD, trans, rot, dpara, mR, mT, mD,
 *                       cSearch)                                                                    Interface.h:64-78 */
int thx_ExpectLocalHostA_host(int gpuIdx, float** wC, float** wR, float** wT, float** wD, double** oldR, double** oldT,
                              double** oldD, double** trans, double** rot, double** dpara, int mR, int mT, int mD,
                              int cSearch);
/* void ExpectLocalRTD(gpuIdx, ManagedCalPoint* mcp, oldR, oldT, oldD, trans, rot, dpara)            Interface.h:80-87
 * rot = nR quaternions, trans = nT shifts, oldR/oldT/oldD = Particle::wR/wT/wD. */
int thx_ExpectLocalRTD_host(int gpuIdx, thx_calpoint* mcp, const double* oldR, const double* oldT, const double* oldD,
                            const double* trans, const double* rot, const double* dpara);
/* void ExpectLocalPreI3D(gpuIdx, datShift, mgr, mcp, devdefO, devfreQ, deviCol, deviRow, phaseShift, conT, k1, k2, pf,
 *                        idim, vdim, npxl, interp)                                                  Interface.h:107-123 */
int thx_ExpectLocalPreI3D_host(int gpuIdx, int datShift, const thx_texture* mgr, thx_calpoint* mcp, const float* devdefO,
                               const float* devfreQ, const int* deviCol, const int* deviRow, float phaseShift, float conT,
                               float k1, float k2, int pf, int idim, int vdim, int npxl, int interp);
/* void ExpectLocalM(gpuIdx, datShift, mcp, devdatP, devctfP, devsigP, wC, wR, wT, wD, oldC, npxl)   Interface.h:125-139
 * one particle-filter phase of one image (src/Optimiser.cpp:1225-1406): wC [1], wR [nR], wT [nT], wD [mD or 1] on the
 * host; results equal to thx_expect_local_dev on a batch of one. */
int thx_ExpectLocalM_host(int gpuIdx, int datShift, thx_calpoint* mcp, const float* devdatP, const float* devctfP,
                          const float* devsigP, float* wC, float* wR, float* wT, float* wD, double oldC, int npxl);
/* void ExpectLocalHostF(...)                                                                        Interface.h:141-152 */
int thx_ExpectLocalHostF_host(int gpuIdx, float** wC, float** wR, float** wT, float** wD, double** oldR, double** oldT,
                              double** oldD, double** trans, double** rot, double** dpara, int cSearch);
/* void ExpectLocalFin(gpuIdx, devdatP, devctfP, devdefO, devfreQ, devsigP, cSearch)                 Interface.h:154-160 */
int thx_ExpectLocalFin_host(int gpuIdx, float** devdatP, float** devctfP, float** devdefO, float** devfreQ, float** devsigP,
                            int cSearch);
/* void ExpectFreeIdx(int gpuIdx, int** deviCol, int** deviRow)                                      Interface.h:162-164 */
int thx_ExpectFreeIdx_host(int gpuIdx, int** deviCol, int** deviRow);

/* Staged Reconstructor::reconstructG (src/Reconstructor.cpp:1835-2330).  T3D / W3D are the REAL copies the reference
 * makes (volumeT / volumeW, :1858-1880), dim = the FT grid's edge (_T3D.nSlcFT()), all arrays on the host. */

/* void ExposePT(gpuIdx, RFLOAT* T3D, maxRadius, pf, dim, vec FSC, bool joinHalf, const int wienerF)  Interface.h:337-344
 * T /= FSC'(shell) between wienerF*pf and maxRadius*pf (the MAP branch, src/Reconstructor.cpp:1242-1270). */
int thx_ExposePT_host(int gpuIdx, float* T3D, int maxRadius, int pf, int dim, const float* FSC, int nFSC, int joinHalf,
                      int wienerF);
/* void ExposeWT(gpuIdx, T3D, W3D, TabFunction& kernelRL, nf, maxRadius, pf, dim, maxIter, minIter, size)
 *                                                                                                   Interface.h:438-448
 * the whole gridding-weight iteration on the device (FFTs by rocFFT); tab = kernelRL.getData(), tabSize entries of the
 * 1e5-step table on [0, 1]; size = _N.  W3D receives the balanced weights. */
int thx_ExposeWT_host(int gpuIdx, const float* T3D, float* W3D, const float* tab, int tabSize, float nf, int maxRadius,
                      int pf, int dim, int maxIter, int minIter, int size);
/* void ExposeWT(gpuIdx, T3D, W3D, maxRadius, pf, dim)  (no grid correction)                          Interface.h:457-462 */
int thx_ExposeWT_plain_host(int gpuIdx, const float* T3D, float* W3D, int maxRadius, int pf, int dim);
/* AllocDevicePoint / HostDeviceInit / ExposeC / ExposeForConvC / ExposeWC / FreeDevHostPoint        Interface.h:358-436
 * the same iteration with the two FFTs left to the caller (src/Reconstructor.cpp:1985-2087).  stream[] receives
 * streamNum opaque stream handles.  devDiff / devCount (RECONSTRUCTOR_CHECK_C_AVERAGE) come back NULL: the reference
 * is configured for RECONSTRUCTOR_CHECK_C_MAX (include/Config.h:101-103). */
int thx_AllocDevicePoint_host(int gpuIdx, float** dev_C, float** dev_W, float** dev_T, float** dev_tab, float** devDiff,
                              float** devMax, int** devCount, void** stream, int streamNum, int tabSize, int dim);
int thx_HostDeviceInit_host(int gpuIdx, const float* T3D, const float* tab, float* dev_W, float* dev_T, float* dev_tab,
                            void** stream, int streamNum, int tabSize, int maxRadius, int pf, int dim);
/* C3D (complex, FT layout) = T * W */
int thx_ExposeC_host(int gpuIdx, float* C3D, float* dev_C, const float* dev_T, float* dev_W, void** stream, int streamNum,
                     int dim);
/* C3D_rl (real, dim^3, after the caller's inverse FFT incl. its 1/size) *= kernelRL(|x|^2 / (pf*size)^2) / nf;
 * step = kernelRL.getStep(), dim = C3D.nSlcRL() (both read from the objects in the reference's wrapper) */
int thx_ExposeForConvC_host(int gpuIdx, float* C3D_rl, float* dev_C, const float* dev_tab, void** stream, float step,
                            float nf, int streamNum, int tabSize, int pf, int size, int dim);
/* C3D (complex, after the caller's forward FFT): W /= max(|C|, 1e-6) in the sphere; *diffC = max ||C| - 1| there */
int thx_ExposeWC_host(int gpuIdx, const float* C3D, float* dev_C, float* cmax, float* dev_W, float* devMax, void** stream,
                      float* diffC, int streamNum, int maxRadius, int pf, int dim);
/* copies dev_W into volumeW, then frees everything AllocDevicePoint made */
int thx_FreeDevHostPoint_host(int gpuIdx, float** dev_C, float** dev_W, float** dev_T, float** dev_tab, float** devDiff,
                              float** devMax, int** devCount, void** stream, float* volumeW, int streamNum, int dim);
/* void ExposePFW(gpuIdx, Volume& padDst, Volume& F3D, RFLOAT* W3D, maxRadius, pf)                   Interface.h:472-477
 * padDst (complex, pdim grid) = F * W inside the sphere, 0 elsewhere; fdim = F3D's grid. */
int thx_ExposePFW_host(int gpuIdx, float* padDst, const float* F3D, const float* W3D, int maxRadius, int pf, int pdim,
                       int fdim);
/* void ExposePF(gpuIdx, Volume& padDst, Volume& padDstR, Volume& F3D, RFLOAT* W3D, maxRadius, pf)   Interface.h:479-485
 * as ExposePFW, then the inverse FFT (incl. 1/size) into padDstR (real, pdim^3); padDst may be NULL. */
int thx_ExposePF_host(int gpuIdx, float* padDst, float* padDstR, const float* F3D, const float* W3D, int maxRadius, int pf,
                      int pdim, int fdim);
/* void ExposeCorrF(gpuIdx, Volume& dst, RFLOAT* mkbRL, RFLOAT nf)                                   Interface.h:493-496
 * dst (real, dim^3) /= mkbRL[|k|][|j|][|i|] (the RECONSTRUCTOR_TRILINEAR_KERNEL table, include/Config.h:97). */
int thx_ExposeCorrF_host(int gpuIdx, float* dst, const float* mkbRL, float nf, int dim);
/* void ExposeCorrF(gpuIdx, Volume& dstN, Volume& dst, RFLOAT* mkbRL, RFLOAT nf)                     Interface.h:498-502
 * the same correction of dstN followed by the forward FFT into dstFT (complex, dim grid). */
int thx_ExposeCorrF_fft_host(int gpuIdx, const float* dstN, float* dstFT, const float* mkbRL, float nf, int dim);

#ifdef __cplusplus
}
#endif
#endif /* THUNDER_AMD_H */
