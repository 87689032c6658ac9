// integration/callsite_lint.cpp -- REFERENCE-SIDE lint unit (tools/boundary_lint.sh, -fsyntax-only; never linked, never run): the
// hot-path calls of the reference's CPU build, written with the reference's OWN types (Complex, RFLOAT, Eigen's dmat33, vec, from
// its unchanged headers) exactly as src/Optimiser.cpp writes them, against the class mirrors include/thunder_amd/{Projector,
// Reconstructor}.hpp.  It compiles iff those mirrors accept the reference's call syntax -- the adapters VERDICT round 4 asked for.
#include "Typedef.h"      // dmat33, vec (Eigen)
#include "Precision.h"    // RFLOAT, Complex
#include "Volume.h"       // Volume
#include "thunder_amd/Reconstructor.hpp"

void callsites(thunder_amd::Projector& proj, thunder_amd::Reconstructor& reco, Complex* priP, Complex* imgFT, const Complex* ctfFT,
               const dmat33& rot, const int* iCol, const int* iRow, int nPxl, unsigned int nThread, const Complex* transImgP,
               const RFLOAT* ctfP, RFLOAT w, const vec* sig, int N)
{
    proj.project(priP, rot, iCol, iRow, nPxl, nThread);   // _model.proj(t).project(priRotP, rot, _iCol, _iRow, _nPxl, 1), src/Optimiser.cpp:775-781,1294-1308
    reco.insertP(transImgP, ctfP, rot, w, sig);           // _model.reco(cls).insertP(transImgP, ctfP, rot, w, &sig), :7190-7232
    reco.insertP(transImgP, ctfP, rot, w);                // (without the sigma vector, :7210)
    proj.projectImage(imgFT, N, rot, nThread);            // projector.project(image, rot, nThread), appsrc/thunder_project.cpp:146-236
    reco.insert(imgFT, ctfFT, N, rot, w);                 // reconstructor.insert(image, ctf, rot, 1), appsrc/thunder_reconstruct.cpp:194-284
}

// the other callers SURVEY 8 row b lists: src/Optimiser.cpp:6741 (setPreCal), :7190-7232 (insertDir), :7259-7270 (prepareTF), :7366-7371
// (reconstruct into a Volume), src/Model.cpp:1037 (setProjectee from a Volume)
void callsites_b(thunder_amd::Projector& proj, thunder_amd::Reconstructor& reco, Volume& refFT, int nPxl, const int* iColPad,
                 const int* iRowPad, const int* iPxl, const int* iSig, const dvec3& dir, unsigned int nThread, int gpu)
{
    proj.setProjectee(refFT.copyVolume(), nThread);       // _proj[l].setProjectee(_ref[l].copyVolume(), nThread): a moved temporary
    proj.setProjectee(refFT, nThread);
    reco.setPreCal(nPxl, iColPad, iRowPad, iPxl, iSig);   // _model.reco(t).setPreCal(_nPxl, _iColPad, _iRowPad, _iPxl, _iSig)
    reco.insertDir(dir);                                  // _model.reco(cls).insertDir(dir)
    reco.prepareTF(nThread);                              // _model.reco(t).prepareTF(_para.nThreadsPerProcess)
    reco.prepareTFG(gpu);                                 // _model.reco(t).prepareTFG(gpus[omp_get_thread_num()])
    Volume ref;
    reco.reconstruct(ref, nThread);                       // _model.reco(t).reconstruct(ref, _para.nThreadsPerProcess)
    reco.reconstructG(ref, gpu, 1);                       // _model.reco(t).reconstructG(ref, gpus[omp_get_thread_num()], 1)
}

// Round 6: the reference's OWN constructor and setters on the class mirrors, and the call lines of the two standalone applications
// verbatim -- appsrc/thunder_reconstruct.cpp:194-284 and appsrc/thunder_project.cpp:146-236 -- with the reference's Symmetry, Image,
// Volume, vec, dmat33 and MPI_Comm from its unchanged headers.  (`recon.setMPIEnv()` without arguments reads MPI_COMM_WORLD inside
// Parallel, src/Parallel.cpp:26-36: a caller of the mirror states its rank with the four-argument form, as Model does for its
// reconstructors, src/Model.cpp:1060-1100.)
#include "Symmetry.h"
#include "Image.h"
#include <mpi.h>

void callsites_apps(const char* symmetry, int boxsize, int maxRadius, unsigned int nThread, Image& img, Image& ctf, dmat33& rot, Volume& tarVol,
                    Volume& ref, int commSize, int commRank, MPI_Comm hemi, MPI_Comm slav, const vec& FSC, const mat& fscMat, int l)
{
    Symmetry sym(symmetry);
    thunder_amd::Reconstructor recon(MODE_3D, boxsize, boxsize, 2, &sym, 1.9, 15);      // Reconstructor recon(MODE_3D, boxsize, boxsize, 2, &sym, 1.9, 15), appsrc/thunder_reconstruct.cpp:194
    recon.setMPIEnv(commSize, commRank, hemi, slav);                                    // _reco[l]->setMPIEnv(_commSize, _commRank, _hemi, _slav), src/Model.cpp:78
    recon.allocSpace(nThread);                                                          // :198
    recon.setMaxRadius(maxRadius);                                                      // :200
    recon.insert(img, ctf, rot, 1);                                                     // recon.insert(img, ctf, rot, 1), :262
    recon.prepareTF(nThread);                                                           // :271
    recon.setMAP(false);                                                                // :277
    recon.reconstruct(tarVol, nThread);                                                 // :279
    recon.freeSpace();                                                                  // :303
    // Model::initProjReco / refreshReco / resetReco, src/Model.cpp:1060-1125
    thunder_amd::Reconstructor reco;
    reco.init(MODE_3D, boxsize, boxsize, 2, &sym, 1.9, 15);                             // _reco[l]->init(_mode, _size, _size, _pf, _sym, _a, _alpha), :1062-1068
    reco.setSymmetry(&sym);                                                             // include/Reconstructor.h:438
    reco.resizeSpace((maxRadius + 2) * 2);                                              // _reco[l]->resizeSpace((_rU + CEIL(_a)) * 2), :1077
    reco.setFSC(vec::Constant(maxRadius, 1));                                           // _reco[l]->setFSC(vec::Constant(_rU, 1)), :1086
    reco.setFSC(FSC);
    reco.setFSC(fscMat.col(l));                                                         // _reco[l]->setFSC(_FSC.col(l)), :1122
    reco.setMaxRadius(maxRadius);                                                       // :1096,1124
    reco.prepareO();
    // appsrc/thunder_project.cpp:186-207
    int N = ref.nColRL();
    thunder_amd::Projector proj;
    proj.setProjectee(ref.copyVolume(), nThread);                                       // proj.setProjectee(ref.copyVolume(), nThread), :188
    Image pimg(N, N, FT_SPACE);
    proj.project(pimg, rot, nThread);                                                   // proj.project(img, mat, nThread), :207
    (void)N;
}
