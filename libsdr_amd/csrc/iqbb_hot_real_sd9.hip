// iqbb_hot_real_sd9.hip — the real-input small-decimation form (iqbb_hot.hpp, SD) for S = 9 K steps (orders 146 ... 273).
#include "iqbb_hot.hpp"

namespace sdrhip {
int hot_launch_real_sd9(int range, bool rot, int epi, const HotLaunch &hl, const HotArgs &ha, const IqbbArgs &b, bool dry) {
  switch (range) {
    case 0: return hot_launch_sd_one<9, 3, 3, HOT_REAL>(rot, epi, hl, ha, b, dry);
    case 1: return hot_launch_sd_one<9, 2, 5, HOT_REAL>(rot, epi, hl, ha, b, dry);
    case 2: return hot_launch_sd_one<9, 1, 7, HOT_REAL>(rot, epi, hl, ha, b, dry);
    default: return hot_launch_sd_one<9, 0, 9, HOT_REAL>(rot, epi, hl, ha, b, dry);
  }
}
}  // namespace sdrhip
