// io_roundtrip.cpp -- host code in the reference's style against the ImageFile / Database mirrors: write a stack and a
// volume, read them back (whole and slice-wise), parse a .thu table.  Pure host code: runs without a GPU.
#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

#include "thunder_amd/ImageFile.hpp"

using namespace thunder_amd;

int main(int argc, char** argv)
{
    const std::string dir = argc > 1 ? argv[1] : ".";
    const int N = 12, nSlc = 5;
    std::vector<float> stack((size_t)nSlc * N * N), vol((size_t)N * N * N);
    for (size_t i = 0; i < stack.size(); i++) stack[i] = std::sin(0.37f * (float)i) * 3.f + (float)(i % 7);
    for (size_t i = 0; i < vol.size(); i++) vol[i] = std::cos(0.11f * (float)i);
    const std::string sp = dir + "/s.mrcs", vp = dir + "/v.mrc";
    ImageFile::writeStack(sp.c_str(), stack.data(), N, nSlc, 1.32f);
    ImageFile::writeVolume(vp.c_str(), vol.data(), N, N, N, 1.32f);
    ImageFile imf(sp.c_str(), "rb");
    imf.readMetaData();
    if (imf.nCol() != N || imf.nRow() != N || imf.nSlc() != nSlc || imf.mode() != 2) { std::printf("FAIL meta\n"); return 1; }
    std::vector<float> img((size_t)N * N);
    for (int s = 0; s < nSlc; s++) {
        imf.readImage(img.data(), s);
        for (int i = 0; i < N * N; i++)
            if (img[i] != stack[(size_t)s * N * N + i]) { std::printf("FAIL image %d\n", s); return 1; }
    }
    ImageFile vf(vp.c_str(), "rb");
    vf.readMetaData();
    std::vector<float> back(vol.size());
    vf.readVolume(back.data());
    for (size_t i = 0; i < vol.size(); i++)
        if (back[i] != vol[i]) { std::printf("FAIL volume\n"); return 1; }
    // .thu
    const std::string tp = dir + "/p.thu";
    FILE* f = std::fopen(tp.c_str(), "w");
    std::fprintf(f, "# comment\n\n");
    for (int l = 0; l < 4; l++)
        std::fprintf(f, "%18.9f %18.9f %18.9f %18.9f %18.9f %18.9f %18.9f %06d@s.mrcs mic.mrc %18.9f %18.9f %6d %6d %18.9f %18.9f %18.9f %18.9f\n",
                     300000.0, 15000.0 + l, 15100.0 + l, 0.5, 2.7e7, 0.1, 0.0, l + 1, 0.0, 0.0, l % 2 + 1, 0, 1.0, 0.0, 0.0, 0.0);
    std::fclose(f);
    Database db(tp.c_str());
    if (db.nParticle() != 4 || db.nGroup() != 2) { std::printf("FAIL db counts\n"); return 1; }
    CTFAttr c;
    db.ctf(c, 2);
    int iSlc; std::string file;
    Database::splitPath(db.path(2), iSlc, file);
    if (c.defocusU != 15002.f || iSlc != 2 || file != "s.mrcs" || db.groupID(3) != 2 || db.quat(1)[0] != 1.0) { std::printf("FAIL db fields\n"); return 1; }
    std::printf("OK\n");
    return 0;
}
