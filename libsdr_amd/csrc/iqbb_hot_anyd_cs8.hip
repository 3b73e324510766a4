// iqbb_hot_anyd_cs8.hip — the hot kernel's any-decimation form (iqbb_hot.hpp, DG) for IQBaseBand<int8_t> (complex<int8> input): the
// reference's documented chain is 16 taps, unshifted, 2.4 MS/s to 100 kS/s (src/sdr.hh:225-240: decimation 24). S = 2, 3, 5, 9.
#include "iqbb_hot.hpp"

namespace sdrhip {
void hot_launch_anyd_cs8(int S, int range, bool rot, int epi, const HotLaunch &hl, const HotArgs &ha, const IqbbArgs &b) {
  if (S == 2) hot_launch_anyd_one<2, 0, 2, HOT_CS8>(rot, epi, hl, ha, b);
  else if (S == 3) { if (range == 0) hot_launch_anyd_one<3, 1, 2, HOT_CS8>(rot, epi, hl, ha, b); else hot_launch_anyd_one<3, 0, 3, HOT_CS8>(rot, epi, hl, ha, b); }
  else if (S == 5) { if (range == 0) hot_launch_anyd_one<5, 1, 3, HOT_CS8>(rot, epi, hl, ha, b); else hot_launch_anyd_one<5, 0, 5, HOT_CS8>(rot, epi, hl, ha, b); }
  else {
    switch (range) {
      case 0: hot_launch_anyd_one<9, 3, 3, HOT_CS8>(rot, epi, hl, ha, b); break;
      case 1: hot_launch_anyd_one<9, 2, 5, HOT_CS8>(rot, epi, hl, ha, b); break;
      case 2: hot_launch_anyd_one<9, 1, 7, HOT_CS8>(rot, epi, hl, ha, b); break;
      default: hot_launch_anyd_one<9, 0, 9, HOT_CS8>(rot, epi, hl, ha, b); break;
    }
  }
}
}  // namespace sdrhip
