// photon_host.cpp — see photon_host.h.  Median choice, widest-axis rule, Hoare-style partition and the
// placement of one- and two-element tails follow cyPhotonMap.h:262-328 exactly: the balanced order decides
// the order in which LocatePhotons meets photons, which the irradiance estimate depends on.
#include "photon_host.h"

#include <string.h>

namespace bhrt {
namespace {
struct Balancer {
    std::vector<HostPhoton> &ph;
    std::vector<HostPhoton> out;
    explicit Balancer(std::vector<HostPhoton> &p) : ph(p), out(p.size()) { memset(out.data(), 0, sizeof(HostPhoton) * out.size()); }
    void Swap(int i, int j) { HostPhoton t = ph[i]; ph[i] = ph[j]; ph[j] = t; }
    void Segment(const float *bmin, const float *bmax, int index, int start, int end)
    {
        int median = 1;
        while ((4 * median) <= (end - start + 1)) median += median;
        if ((3 * median) <= (end - start + 1)) { median += median; median += start - 1; }
        else median = end - median + 1;
        int axis = 2;
        const float dx = bmax[0] - bmin[0], dy = bmax[1] - bmin[1], dz = bmax[2] - bmin[2];
        if (dx > dy) { if (dx > dz) axis = 0; }
        else if (dy > dz) axis = 1;
        int left = start, right = end;
        while (right > left) {
            const float v = ph[right].pos[axis];
            int i = left - 1, j = right;
            while (ph[++i].pos[axis] < v) {}
            while (ph[--j].pos[axis] > v && j > left) {}
            while (i < j) {
                Swap(i, j);
                while (ph[++i].pos[axis] < v) {}
                while (ph[--j].pos[axis] > v && j > left) {}
            }
            Swap(i, right);
            if (i >= median) right = i - 1;
            if (i <= median) left = i + 1;
        }
        out[index] = ph[median];
        out[index].planeAndDirZ = (uint8_t)((out[index].planeAndDirZ & 0x8) | (axis & 0x3));
        if (median > start) {
            if (start < median - 1) {
                float tmax[3] = {bmax[0], bmax[1], bmax[2]};
                tmax[axis] = out[index].pos[axis];
                Segment(bmin, tmax, 2 * index, start, median - 1);
            } else out[2 * index] = ph[start];
        }
        if (median < end) {
            if (median + 1 < end) {
                float tmin[3] = {bmin[0], bmin[1], bmin[2]};
                tmin[axis] = out[index].pos[axis];
                Segment(tmin, bmax, 2 * index + 1, median + 1, end);
            } else out[2 * index + 1] = ph[end];
        }
    }
};
} // namespace

void BalancePhotons(std::vector<HostPhoton> &photons)
{
    const int n = (int)photons.size() - 1;
    if (n <= 0) return;
    float bmin[3], bmax[3]; // the box starts from the unused slot 0 (cyPhotonMap.h:241-242)
    for (int k = 0; k < 3; k++) bmin[k] = bmax[k] = photons[0].pos[k];
    for (int i = 1; i <= n; i++)
        for (int k = 0; k < 3; k++) {
            if (bmin[k] > photons[i].pos[k]) bmin[k] = photons[i].pos[k];
            if (bmax[k] < photons[i].pos[k]) bmax[k] = photons[i].pos[k];
        }
    Balancer B(photons);
    B.Segment(bmin, bmax, 1, 1, n);
    photons.swap(B.out);
}

} // namespace bhrt
