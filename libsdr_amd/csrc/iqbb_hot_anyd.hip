// iqbb_hot_anyd.hip — explicit instantiations of the hot kernel's any-decimation form (iqbb_hot.hpp, DG) for S = 2, 3 and 5
// K steps (orders up to 65; the reference's receivers use 16 and 21 taps at decimation 62 and 125), complex<int16> and
// complex<uint8> input. S = 9: iqbb_hot_anyd9.hip; S = 17: iqbb_hot_anyd17_*.hip.
#include "iqbb_hot.hpp"

namespace sdrhip {
void hot_launch_anyd(int S, int in, int range, bool rot, int epi, const HotLaunch &hl, const HotArgs &ha, const IqbbArgs &b) {
  if (in == HOT_REAL) { hot_launch_real_anyd(S, range, rot, epi, hl, ha, b); return; }
  if (in == HOT_CS8) { hot_launch_anyd_cs8(S, range, rot, epi, hl, ha, b); return; }
  const bool cu8 = in == HOT_CU8;
  if (S == 2) {
    if (cu8) hot_launch_anyd_one<2, 0, 2, HOT_CU8>(rot, epi, hl, ha, b); else hot_launch_anyd_one<2, 0, 2, HOT_CS16>(rot, epi, hl, ha, b);
  } else if (S == 3) {
    if (range == 0) { if (cu8) hot_launch_anyd_one<3, 1, 2, HOT_CU8>(rot, epi, hl, ha, b); else hot_launch_anyd_one<3, 1, 2, HOT_CS16>(rot, epi, hl, ha, b); }
    else { if (cu8) hot_launch_anyd_one<3, 0, 3, HOT_CU8>(rot, epi, hl, ha, b); else hot_launch_anyd_one<3, 0, 3, HOT_CS16>(rot, epi, hl, ha, b); }
  } else if (S == 5) {
    if (range == 0) { if (cu8) hot_launch_anyd_one<5, 1, 3, HOT_CU8>(rot, epi, hl, ha, b); else hot_launch_anyd_one<5, 1, 3, HOT_CS16>(rot, epi, hl, ha, b); }
    else { if (cu8) hot_launch_anyd_one<5, 0, 5, HOT_CU8>(rot, epi, hl, ha, b); else hot_launch_anyd_one<5, 0, 5, HOT_CS16>(rot, epi, hl, ha, b); }
  } else if (S == 9) {
    hot_launch_anyd9(in, range, rot, epi, hl, ha, b);
  } else if (S == 17) {
    if (cu8) hot_launch_anyd17_cu8(range, rot, epi, hl, ha, b); else hot_launch_anyd17_cs16(range, rot, epi, hl, ha, b);
  } else {
    if (cu8) hot_launch_anyd33_cu8(range, rot, epi, hl, ha, b); else hot_launch_anyd33_cs16(range, rot, epi, hl, ha, b);
  }
}
}  // namespace sdrhip
