// ImageFile.hpp / Database -- header-only C++ mirrors of thuem/THUNDER's ImageFile (include/Image/ImageFile.h:117-360,
// src/Image/ImageFile.cpp) and Database (include/Database.h, src/Database.cpp) over the host-side C ABI entry points
// (thx_mrc_*, thx_thu_*).  Same method names and error behaviour (REPORT_ERROR + abort); images and volumes are plain
// float arrays in the reference's in-memory layout (origin at index 0).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../thunder_amd.h"

namespace thunder_amd {

#ifndef THX_ABORT_ON
#define THX_ABORT_ON(rc)                                                                     \
    do {                                                                                     \
        if ((rc) != 0) {                                                                     \
            std::fprintf(stderr, "thunder_amd FATAL: %s (%s:%d)\n", thx_last_error(), __FILE__, __LINE__); \
            std::abort(); /* REPORT_ERROR + abort(), include/Logging.h:28-34 */              \
        }                                                                                    \
    } while (0)
#endif

class ImageFile {
public:
    ImageFile() : _nCol(0), _nRow(0), _nSlc(0), _mode(2), _symSize(0) {}
    // ImageFile(const char* filename, const char* option), include/Image/ImageFile.h:138 (option "rb" / "wb")
    ImageFile(const char* filename, const char* /*option*/) : ImageFile() { _path = filename; }

    // readMetaData(), src/Image/ImageFile.cpp:57-75,209-228
    void readMetaData()
    {
        THX_ABORT_ON(thx_mrc_info(_path.c_str(), &_nCol, &_nRow, &_nSlc, &_mode, &_symSize));
    }
    int mode() const { return _mode; }
    int nCol() const { return _nCol; }
    int nRow() const { return _nRow; }
    int nSlc() const { return _nSlc; }
    int size() const { return _nCol * _nRow * _nSlc; }
    int symmetryDataSize() const { return _symSize; }

    // readImage(Image& dst, int iSlc = 0, fileType "MRC"), src/Image/ImageFile.cpp:96-113,248-264: dst [nRow][nCol]
    void readImage(float* dst, int iSlc = 0) const { THX_ABORT_ON(thx_mrc_read_images(_path.c_str(), iSlc, 1, dst)); }
    // a run of slices of a stack (what Optimiser::initImg reads particle by particle, src/Optimiser.cpp:4646-4660)
    void readImages(float* dst, int first, int count) const { THX_ABORT_ON(thx_mrc_read_images(_path.c_str(), first, count, dst)); }
    // readVolume(Volume& dst), :115-130,289-303: dst [nSlc][nRow][nCol]
    void readVolume(float* dst) const { THX_ABORT_ON(thx_mrc_read_volume(_path.c_str(), dst)); }

    // writeVolume(const char dst[], const Volume& src, RFLOAT pixelSize), :332-357
    static void writeVolume(const char dst[], const float* src, int nCol, int nRow, int nSlc, float pixelSize = 1)
    {
        THX_ABORT_ON(thx_mrc_write_volume(dst, src, nCol, nRow, nSlc, pixelSize));
    }
    // openStack / writeStack / closeStack, :359-404, as one call on a whole stack [nSlc][size][size]
    static void writeStack(const char dst[], const float* src, int size, int nSlc, float pixelSize = 1)
    {
        THX_ABORT_ON(thx_mrc_write_stack(dst, src, size, nSlc, pixelSize));
    }

private:
    std::string _path;
    int _nCol, _nRow, _nSlc, _mode, _symSize;
};

struct CTFAttr {  // include/Database.h:302-330 == thx_ctf_attr
    float voltage, defocusU, defocusV, defocusTheta, Cs, amplitudeContrast, phaseShift;
};

// Database: the .thu particle table (include/Database.h:22-287).  The reference seeks into the file on every accessor;
// here openDatabase() parses it once.
class Database {
public:
    Database() : _n(0), _nGroup(0) {}
    explicit Database(const char database[]) : Database() { openDatabase(database); }

    void openDatabase(const char database[])
    {
        THX_ABORT_ON(thx_thu_count(database, &_n, &_nGroup));
        _ctf.resize(_n); _path.assign((size_t)_n * kPath, 0); _group.resize(_n); _cls.resize(_n);
        _quat.resize(4 * (size_t)_n); _tran.resize(2 * (size_t)_n); _stdT.resize(2 * (size_t)_n); _d.resize(_n); _score.resize(_n);
        THX_ABORT_ON(thx_thu_load(database, _n, reinterpret_cast<thx_ctf_attr*>(_ctf.data()), _path.data(), kPath, _group.data(),
                                  _cls.data(), _quat.data(), _tran.data(), _stdT.data(), _d.data(), _score.data()));
    }
    int nParticle() const { return _n; }                                   // src/Database.cpp:137-150
    int nGroup() const { return _nGroup; }                                 // :152-180
    std::string path(int i) const { return std::string(&_path[(size_t)i * kPath]); }   // :303-318
    void ctf(CTFAttr& dst, int i) const { dst = _ctf[i]; }                 // :385-396
    int groupID(int i) const { return _group[i]; }                         // :283-301
    int cls(int i) const { return _cls[i]; }                               // :398-414
    const double* quat(int i) const { return &_quat[4 * (size_t)i]; }      // :416-446
    const double* tran(int i) const { return &_tran[2 * (size_t)i]; }      // :506-530
    double stdTX(int i) const { return _stdT[2 * (size_t)i]; }
    double stdTY(int i) const { return _stdT[2 * (size_t)i + 1]; }
    double d(int i) const { return _d[i]; }
    double score(int i) const { return _score[i]; }

    // "000017@stack.mrcs" -> (16, "stack.mrcs"); a path without '@' is a single-image file (src/Optimiser.cpp:4646-4660)
    static void splitPath(const std::string& p, int& iSlc, std::string& file)
    {
        const size_t at = p.find('@');
        if (at == std::string::npos) { iSlc = 0; file = p; }
        else { iSlc = std::atoi(p.substr(0, at).c_str()) - 1; file = p.substr(at + 1); }
    }

private:
    static const int kPath = 256;
    int _n, _nGroup;
    std::vector<CTFAttr> _ctf;
    std::vector<char> _path;
    std::vector<int> _group, _cls;
    std::vector<double> _quat, _tran, _stdT, _d, _score;
};

}  // namespace thunder_amd
