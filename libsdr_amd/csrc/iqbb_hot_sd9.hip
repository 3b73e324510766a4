// iqbb_hot_sd9.hip — the hot kernel's small-decimation form (iqbb_hot.hpp, SD: decimations 2 ... 7) for S = 9 K steps
// (orders up to 129; plans without a shift carry two sample arrays and run in 8-wave workgroups: hot_sd_nw).
#include "iqbb_hot.hpp"

namespace sdrhip {
int hot_launch_sd9(int in, int range, bool rot, int epi, const HotLaunch &hl, const HotArgs &ha, const IqbbArgs &b, bool dry) {
  const bool cu8 = in == HOT_CU8;
  switch (range) {
    case 0: return cu8 ? hot_launch_sd_one<9, 3, 3, HOT_CU8>(rot, epi, hl, ha, b, dry) : hot_launch_sd_one<9, 3, 3, HOT_CS16>(rot, epi, hl, ha, b, dry);
    case 1: return cu8 ? hot_launch_sd_one<9, 2, 5, HOT_CU8>(rot, epi, hl, ha, b, dry) : hot_launch_sd_one<9, 2, 5, HOT_CS16>(rot, epi, hl, ha, b, dry);
    case 2: return cu8 ? hot_launch_sd_one<9, 1, 7, HOT_CU8>(rot, epi, hl, ha, b, dry) : hot_launch_sd_one<9, 1, 7, HOT_CS16>(rot, epi, hl, ha, b, dry);
    default: return cu8 ? hot_launch_sd_one<9, 0, 9, HOT_CU8>(rot, epi, hl, ha, b, dry) : hot_launch_sd_one<9, 0, 9, HOT_CS16>(rot, epi, hl, ha, b, dry);
  }
}
}  // namespace sdrhip
