// iqbb_hot_anyd17_cu8.hip — the hot kernel's any-decimation form (iqbb_hot.hpp, DG) for S = 17 K steps (orders 130 ... 257),
// complex<uint8> input: 8- and 16-wave workgroups sharing one LDS copy of the tap fragments, as the /8 kernel of this class.
#include "iqbb_hot.hpp"

namespace sdrhip {
void hot_launch_anyd17_cu8(int range, bool rot, int epi, const HotLaunch &hl, const HotArgs &ha, const IqbbArgs &b) {
  switch (range) {
    case 0: hot_launch_anyd_one<17, 6, 5, HOT_CU8, 8>(rot, epi, hl, ha, b); break;
    case 1: hot_launch_anyd_one<17, 4, 9, HOT_CU8, 8>(rot, epi, hl, ha, b); break;
    default: hot_launch_anyd_one<17, 0, 17, HOT_CU8, 16>(rot, epi, hl, ha, b); break;
  }
}
}  // namespace sdrhip
