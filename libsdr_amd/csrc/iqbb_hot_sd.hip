// iqbb_hot_sd.hip — explicit instantiations of the hot kernel's small-decimation form (iqbb_hot.hpp, SD: decimations 2 ... 7)
// for S = 2, 3 and 5 K steps (orders up to 65), complex<int16> and complex<uint8> input. S = 9: iqbb_hot_sd9.hip; S = 17: iqbb_hot_sd17_*.hip.
#include "iqbb_hot.hpp"

namespace sdrhip {
int hot_launch_sd(int S, int in, int range, bool rot, int epi, const HotLaunch &hl, const HotArgs &ha, const IqbbArgs &b, bool dry) {
  if (in == HOT_REAL) return hot_launch_real_sd(S, range, rot, epi, hl, ha, b, dry);
  if (in == HOT_CS8) return 0;   // (the int8 chain has no small-decimation form: the VALU kernel)
  const bool cu8 = in == HOT_CU8;
  if (S == 2) {
    return cu8 ? hot_launch_sd_one<2, 0, 2, HOT_CU8>(rot, epi, hl, ha, b, dry) : hot_launch_sd_one<2, 0, 2, HOT_CS16>(rot, epi, hl, ha, b, dry);
  } else if (S == 3) {
    if (range == 0) return cu8 ? hot_launch_sd_one<3, 1, 2, HOT_CU8>(rot, epi, hl, ha, b, dry) : hot_launch_sd_one<3, 1, 2, HOT_CS16>(rot, epi, hl, ha, b, dry);
    return cu8 ? hot_launch_sd_one<3, 0, 3, HOT_CU8>(rot, epi, hl, ha, b, dry) : hot_launch_sd_one<3, 0, 3, HOT_CS16>(rot, epi, hl, ha, b, dry);
  } else if (S == 5) {
    if (range == 0) return cu8 ? hot_launch_sd_one<5, 1, 3, HOT_CU8>(rot, epi, hl, ha, b, dry) : hot_launch_sd_one<5, 1, 3, HOT_CS16>(rot, epi, hl, ha, b, dry);
    return cu8 ? hot_launch_sd_one<5, 0, 5, HOT_CU8>(rot, epi, hl, ha, b, dry) : hot_launch_sd_one<5, 0, 5, HOT_CS16>(rot, epi, hl, ha, b, dry);
  }
  if (S == 9) return hot_launch_sd9(in, range, rot, epi, hl, ha, b, dry);
  if (S >= 33) return 0;   // (orders 258 ... 513 at decimations 1 ... 7: no hot form — the VALU kernel)
  return cu8 ? hot_launch_sd17_cu8(range, rot, epi, hl, ha, b, dry) : hot_launch_sd17_cs16(range, rot, epi, hl, ha, b, dry);
}
}  // namespace sdrhip
