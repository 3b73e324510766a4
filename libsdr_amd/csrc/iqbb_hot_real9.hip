// iqbb_hot_real9.hip — explicit instantiations of the hot kernel (iqbb_hot.hpp) for the real-input BaseBand<int16_t>,
// S = 9 K steps of 32 real samples (orders up to 273).
#include "iqbb_hot.hpp"

namespace sdrhip {
void hot_launch_real9(int range, bool rot, int epi, const HotLaunch &hl, const HotArgs &ha, const IqbbArgs &b) {
  switch (range) {
    case 0: hot_launch_one<9, 3, 3, HOT_REAL, 4>(rot, epi, hl, ha, b); break;
    case 1: hot_launch_one<9, 2, 5, HOT_REAL, 4>(rot, epi, hl, ha, b); break;
    case 2: hot_launch_one<9, 1, 7, HOT_REAL, 4>(rot, epi, hl, ha, b); break;
    default: hot_launch_one<9, 0, 9, HOT_REAL, 4>(rot, epi, hl, ha, b); break;
  }
}
}  // namespace sdrhip
