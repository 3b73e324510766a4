// iqbb_hot_real_anyd.hip — explicit instantiations of the hot kernel's any-decimation form (iqbb_hot.hpp, DG) for the
// real-input BaseBand<int16_t> (src/baseband.hh:425-460): S = 3 and 5 K steps of 32 real samples (orders up to 81 / 145),
// decimations 9 ... 512. S = 9: iqbb_hot_real_anyd9.hip.
#include "iqbb_hot.hpp"

namespace sdrhip {
void hot_launch_real_anyd(int S, int range, bool rot, int epi, const HotLaunch &hl, const HotArgs &ha, const IqbbArgs &b) {
  if (S == 3) {
    if (range == 0) hot_launch_anyd_one<3, 1, 2, HOT_REAL>(rot, epi, hl, ha, b); else hot_launch_anyd_one<3, 0, 3, HOT_REAL>(rot, epi, hl, ha, b);
  } else if (S == 5) {
    if (range == 0) hot_launch_anyd_one<5, 1, 3, HOT_REAL>(rot, epi, hl, ha, b); else hot_launch_anyd_one<5, 0, 5, HOT_REAL>(rot, epi, hl, ha, b);
  } else {
    hot_launch_real_anyd9(range, rot, epi, hl, ha, b);
  }
}
}  // namespace sdrhip
