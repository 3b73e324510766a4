// iqbb_hot_s9_cs16.hip — explicit instantiations of the hot kernel (iqbb_hot.hpp) for S = 9 K steps (orders up to 129), complex<int16> input; one translation unit per filter-length
// class so that the build compiles them in parallel.
#include "iqbb_hot.hpp"

namespace sdrhip {
void hot_launch_s9_cs16(int range, bool rot, int epi, const HotLaunch &hl, const HotArgs &ha, const IqbbArgs &b) {
  switch (range) {
    case 0: hot_launch_one<9, 3, 3, HOT_CS16, 4>(rot, epi, hl, ha, b); break;
    case 1: hot_launch_one<9, 2, 5, HOT_CS16, 4>(rot, epi, hl, ha, b); break;
    case 2: hot_launch_one<9, 1, 7, HOT_CS16, 4>(rot, epi, hl, ha, b); break;
    default: hot_launch_one<9, 0, 9, HOT_CS16, 4>(rot, epi, hl, ha, b); break;
  }
}
}  // namespace sdrhip
