// iqbb_hot_real_sd.hip — explicit instantiations of the hot kernel's small-decimation form (iqbb_hot.hpp, SD: decimations
// 1 ... 7) for the real-input BaseBand<int16_t>: S = 3 and 5 K steps. S = 9: iqbb_hot_real_sd9.hip.
#include "iqbb_hot.hpp"

namespace sdrhip {
int hot_launch_real_sd(int S, int range, bool rot, int epi, const HotLaunch &hl, const HotArgs &ha, const IqbbArgs &b, bool dry) {
  if (S == 3) return range == 0 ? hot_launch_sd_one<3, 1, 2, HOT_REAL>(rot, epi, hl, ha, b, dry) : hot_launch_sd_one<3, 0, 3, HOT_REAL>(rot, epi, hl, ha, b, dry);
  if (S == 5) return range == 0 ? hot_launch_sd_one<5, 1, 3, HOT_REAL>(rot, epi, hl, ha, b, dry) : hot_launch_sd_one<5, 0, 5, HOT_REAL>(rot, epi, hl, ha, b, dry);
  return hot_launch_real_sd9(range, rot, epi, hl, ha, b, dry);
}
}  // namespace sdrhip
