// iqbb_hot_anyd33_cs16.hip — the hot kernel's any-decimation form (iqbb_hot.hpp, DG) for S = 33 K steps (orders 258 ... 513),
// complex<int16> input: one 8-wave workgroup per CU, as the /8 kernel of this class.
#include "iqbb_hot.hpp"

namespace sdrhip {
void hot_launch_anyd33_cs16(int range, bool rot, int epi, const HotLaunch &hl, const HotArgs &ha, const IqbbArgs &b) {
  switch (range) {
    case 0: hot_launch_anyd_one<33, 12, 9, HOT_CS16, 8>(rot, epi, hl, ha, b); break;
    case 1: hot_launch_anyd_one<33, 16, 9, HOT_CS16, 8>(rot, epi, hl, ha, b); break;
    case 2: hot_launch_anyd_one<33, 18, 9, HOT_CS16, 8>(rot, epi, hl, ha, b); break;
    case 3: hot_launch_anyd_one<33, 8, 17, HOT_CS16, 8>(rot, epi, hl, ha, b); break;
    default: hot_launch_anyd_one<33, 0, 33, HOT_CS16, 8>(rot, epi, hl, ha, b); break;
  }
}
}  // namespace sdrhip
