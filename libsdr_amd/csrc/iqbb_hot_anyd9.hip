// iqbb_hot_anyd9.hip — the hot kernel's any-decimation form (iqbb_hot.hpp, DG) for S = 9 K steps (orders up to 129).
#include "iqbb_hot.hpp"

namespace sdrhip {
void hot_launch_anyd9(int in, int range, bool rot, int epi, const HotLaunch &hl, const HotArgs &ha, const IqbbArgs &b) {
  const bool cu8 = in == HOT_CU8;
  switch (range) {
    case 0: if (cu8) hot_launch_anyd_one<9, 3, 3, HOT_CU8>(rot, epi, hl, ha, b); else hot_launch_anyd_one<9, 3, 3, HOT_CS16>(rot, epi, hl, ha, b); break;
    case 1: if (cu8) hot_launch_anyd_one<9, 2, 5, HOT_CU8>(rot, epi, hl, ha, b); else hot_launch_anyd_one<9, 2, 5, HOT_CS16>(rot, epi, hl, ha, b); break;
    case 2: if (cu8) hot_launch_anyd_one<9, 1, 7, HOT_CU8>(rot, epi, hl, ha, b); else hot_launch_anyd_one<9, 1, 7, HOT_CS16>(rot, epi, hl, ha, b); break;
    default: if (cu8) hot_launch_anyd_one<9, 0, 9, HOT_CU8>(rot, epi, hl, ha, b); else hot_launch_anyd_one<9, 0, 9, HOT_CS16>(rot, epi, hl, ha, b); break;
  }
}
}  // namespace sdrhip
