// integration/ManagedCalPoint.h -- REFERENCE-SIDE replacement of gpu/include/ManagedCalPoint.h (per-stream device buffers of the
// local search).  Same class name and the methods src/Optimiser.cpp calls (:2284-2291: new, Init; :3361 delete); the device buffers
// live behind one opaque handle of libthunder_amd.
#ifndef MANAGEDCALPOINT_H
#define MANAGEDCALPOINT_H

struct thx_calpoint;

class ManagedCalPoint {
public:
    ManagedCalPoint() : _h(0) {}
    ~ManagedCalPoint();

    void Init(int mode, int cSearch, int gpuIdx, int nR, int nT, int mD, int npxl);

    thx_calpoint* handle() const { return _h; }

private:
    ManagedCalPoint(const ManagedCalPoint&);
    ManagedCalPoint& operator=(const ManagedCalPoint&);

    thx_calpoint* _h;
};

#endif
