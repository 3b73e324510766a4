// iqbb_hot_s33_cu8.hip — explicit instantiations of the hot kernel (iqbb_hot.hpp) for S = 33 K steps (orders 258 ... 513),
// complex<uint8> input: one 8-wave workgroup per CU (33 ... 66 KB of tap fragments, 1024-sample windows).
#include "iqbb_hot.hpp"

namespace sdrhip {
void hot_launch_s33_cu8(int range, bool rot, int epi, const HotLaunch &hl, const HotArgs &ha, const IqbbArgs &b) {
  switch (range) {
    case 0: hot_launch_one<33, 12, 9, HOT_CU8, 8>(rot, epi, hl, ha, b); break;
    case 1: hot_launch_one<33, 16, 9, HOT_CU8, 8>(rot, epi, hl, ha, b); break;
    case 2: hot_launch_one<33, 18, 9, HOT_CU8, 8>(rot, epi, hl, ha, b); break;
    case 3: hot_launch_one<33, 8, 17, HOT_CU8, 8>(rot, epi, hl, ha, b); break;
    default: hot_launch_one<33, 0, 33, HOT_CU8, 8>(rot, epi, hl, ha, b); break;
  }
}
}  // namespace sdrhip
