// integration/ManagedArrayTexture.h -- REFERENCE-SIDE replacement of gpu/include/ManagedArrayTexture.h (which wraps a cudaArray and
// a texture object).  Same class name and the methods src/Optimiser.cpp calls (:2217-2234: new, Init; delete); the object holds one
// opaque handle of libthunder_amd (a projector volume resident in HBM, cell-packed for the local search).
#ifndef MANAGEDARRAYTEXTURE_H
#define MANAGEDARRAYTEXTURE_H

struct thx_texture;

class ManagedArrayTexture {
public:
    ManagedArrayTexture() : _h(0) {}
    ~ManagedArrayTexture();

    void Init(int mode, int vdim, int gpuIdx);

    int getDeviceId();

    thx_texture* handle() const { return _h; }

private:
    ManagedArrayTexture(const ManagedArrayTexture&);              // (not copyable: owns the handle)
    ManagedArrayTexture& operator=(const ManagedArrayTexture&);

    thx_texture* _h;
};

#endif
