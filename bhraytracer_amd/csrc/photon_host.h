// photon_host.h — host part of the caustic photon map: the left-balanced kd-tree build
// (PhotonMap::PrepareForIrradianceEstimation / BalanceSegment, DataStructure/cyPhotonMap.h:236-328).
#pragma once
#include <stdint.h>
#include <vector>

namespace bhrt {
struct HostPhoton { // 24-byte record, cyPhotonMap.h:72-90
    float pos[3];
    float power;
    uint8_t color[3];
    uint8_t planeAndDirZ;
    int16_t dirX, dirY;
};
static_assert(sizeof(HostPhoton) == 24, "photon record must be 24 bytes");
// in: photons[1..n] in emission order (slot 0 = zeros); out: heap-ordered balanced array, same size
void BalancePhotons(std::vector<HostPhoton> &photons);
} // namespace bhrt
