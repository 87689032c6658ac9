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
