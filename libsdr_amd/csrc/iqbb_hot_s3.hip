// iqbb_hot_s3.hip — explicit instantiations of the hot kernel (iqbb_hot.hpp) for S = 3 K steps (orders up to 33), complex<int16> and complex<uint8> input; one translation unit per filter-length
// class so that the build compiles them in parallel.
#include "iqbb_hot.hpp"

namespace sdrhip {
void hot_launch_s3(bool cu8, int range, bool rot, int epi, const HotLaunch &hl, const HotArgs &ha, const IqbbArgs &b) {
  switch (range) {
    case 0: if (cu8) hot_launch_one<3, 1, 2, true, 4>(rot, epi, hl, ha, b); else hot_launch_one<3, 1, 2, false, 4>(rot, epi, hl, ha, b); break;
    default: if (cu8) hot_launch_one<3, 0, 3, true, 4>(rot, epi, hl, ha, b); else hot_launch_one<3, 0, 3, false, 4>(rot, epi, hl, ha, b); break;
  }
}
}  // namespace sdrhip
