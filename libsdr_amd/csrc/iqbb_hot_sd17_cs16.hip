// iqbb_hot_sd17_cs16.hip — the hot kernel's small-decimation form (iqbb_hot.hpp, SD: decimations 2 ... 7) for S = 17 K steps
// (orders 130 ... 257), complex<int16> input: 8- and 16-wave workgroups (hot_sd_nw).
#include "iqbb_hot.hpp"

namespace sdrhip {
int hot_launch_sd17_cs16(int range, bool rot, int epi, const HotLaunch &hl, const HotArgs &ha, const IqbbArgs &b, bool dry) {
  switch (range) {
    case 0: return hot_launch_sd_one<17, 6, 5, HOT_CS16, 8>(rot, epi, hl, ha, b, dry);
    case 1: return hot_launch_sd_one<17, 4, 9, HOT_CS16, 8>(rot, epi, hl, ha, b, dry);
    default: return hot_launch_sd_one<17, 0, 17, HOT_CS16, 16>(rot, epi, hl, ha, b, dry);
  }
}
}  // namespace sdrhip
