// iqbb_hot_real_anyd9.hip — the real-input any-decimation form (iqbb_hot.hpp, DG) for S = 9 K steps (orders 146 ... 273).
#include "iqbb_hot.hpp"

namespace sdrhip {
void hot_launch_real_anyd9(int range, bool rot, int epi, const HotLaunch &hl, const HotArgs &ha, const IqbbArgs &b) {
  switch (range) {
    case 0: hot_launch_anyd_one<9, 3, 3, HOT_REAL>(rot, epi, hl, ha, b); break;
    case 1: hot_launch_anyd_one<9, 2, 5, HOT_REAL>(rot, epi, hl, ha, b); break;
    case 2: hot_launch_anyd_one<9, 1, 7, HOT_REAL>(rot, epi, hl, ha, b); break;
    default: hot_launch_anyd_one<9, 0, 9, HOT_REAL>(rot, epi, hl, ha, b); break;
  }
}
}  // namespace sdrhip
