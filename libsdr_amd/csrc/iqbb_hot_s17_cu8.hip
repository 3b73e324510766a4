// iqbb_hot_s17_cu8.hip — explicit instantiations of the hot kernel (iqbb_hot.hpp) for S = 17 K steps (orders up to 257), complex<uint8> input; one translation unit per filter-length
// class so that the build compiles them in parallel.
#include "iqbb_hot.hpp"

namespace sdrhip {
void hot_launch_s17_cu8(int range, bool rot, int epi, const HotLaunch &hl, const HotArgs &ha, const IqbbArgs &b) {
  switch (range) {
    case 0: hot_launch_one<17, 6, 5, HOT_CU8, 8>(rot, epi, hl, ha, b); break;
    case 1: hot_launch_one<17, 4, 9, HOT_CU8, 8>(rot, epi, hl, ha, b); break;
    default: hot_launch_one<17, 0, 17, HOT_CU8, 16>(rot, epi, hl, ha, b); break;
  }
}
}  // namespace sdrhip
