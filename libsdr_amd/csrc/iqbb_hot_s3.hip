// iqbb_hot_s3.hip — explicit instantiations of the hot kernel (iqbb_hot.hpp) for S = 3 K steps (orders up to 33), complex<int16> and complex<uint8> input; one translation unit per filter-length
// class so that the build compiles them in parallel.
#include "iqbb_hot.hpp"

namespace sdrhip {
void hot_launch_s3(int in, int range, bool rot, int epi, const HotLaunch &hl, const HotArgs &ha, const IqbbArgs &b) {
  const bool cu8 = in == HOT_CU8;
  switch (range) {
    case 0: if (cu8) hot_launch_one<3, 1, 2, HOT_CU8, 4>(rot, epi, hl, ha, b); else hot_launch_one<3, 1, 2, HOT_CS16, 4>(rot, epi, hl, ha, b); break;
    default: if (cu8) hot_launch_one<3, 0, 3, HOT_CU8, 4>(rot, epi, hl, ha, b); else hot_launch_one<3, 0, 3, HOT_CS16, 4>(rot, epi, hl, ha, b); break;
  }
}
}  // namespace sdrhip
