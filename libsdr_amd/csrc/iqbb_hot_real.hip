// iqbb_hot_real.hip — explicit instantiations of the hot kernel (iqbb_hot.hpp) for the real-input BaseBand<int16_t>
// (src/baseband.hh:425-460): S = 3 and 5 K steps of 32 real samples (orders up to 81 / 145).
#include "iqbb_hot.hpp"

namespace sdrhip {
void hot_launch_real9(int range, bool rot, int epi, const HotLaunch &hl, const HotArgs &ha, const IqbbArgs &b);   // iqbb_hot_real9.hip
void hot_launch_real(int S, int range, bool rot, int epi, const HotLaunch &hl, const HotArgs &ha, const IqbbArgs &b) {
  if (S == 3) {
    switch (range) {
      case 0: hot_launch_one<3, 1, 2, HOT_REAL, 4>(rot, epi, hl, ha, b); break;
      default: hot_launch_one<3, 0, 3, HOT_REAL, 4>(rot, epi, hl, ha, b); break;
    }
  } else if (S == 5) {
    switch (range) {
      case 0: hot_launch_one<5, 1, 3, HOT_REAL, 4>(rot, epi, hl, ha, b); break;
      default: hot_launch_one<5, 0, 5, HOT_REAL, 4>(rot, epi, hl, ha, b); break;
    }
  } else {
    hot_launch_real9(range, rot, epi, hl, ha, b);
  }
}
}  // namespace sdrhip
