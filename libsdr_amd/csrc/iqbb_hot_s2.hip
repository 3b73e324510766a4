// iqbb_hot_s2.hip — explicit instantiations of the hot kernel (iqbb_hot.hpp) for S = 2 K steps (orders up to 17), complex<int16> and complex<uint8> input; one translation unit per filter-length
// class so that the build compiles them in parallel.
#include "iqbb_hot.hpp"

namespace sdrhip {
void hot_launch_s2(int in, int range, bool rot, int epi, const HotLaunch &hl, const HotArgs &ha, const IqbbArgs &b) {
  const bool cu8 = in == HOT_CU8;
  switch (range) {
    default: if (cu8) hot_launch_one<2, 0, 2, HOT_CU8, 4>(rot, epi, hl, ha, b); else hot_launch_one<2, 0, 2, HOT_CS16, 4>(rot, epi, hl, ha, b); break;
  }
}
}  // namespace sdrhip
