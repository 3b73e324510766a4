// iqbb_hot_cs8.hip — explicit instantiations of the hot kernel (iqbb_hot.hpp) for IQBaseBand<int8_t> at decimation 8 (complex<int8>
// input, reference src/sdr.hh:225-240's chain): S = 2, 3, 5 and 9 K steps (orders up to 129), no demodulator or FMDemod<int8_t,int16_t>.
#include "iqbb_hot.hpp"

namespace sdrhip {
void hot_launch_cs8(int S, int range, bool rot, int epi, const HotLaunch &hl, const HotArgs &ha, const IqbbArgs &b) {
  if (S == 2) hot_launch_one<2, 0, 2, HOT_CS8, 4>(rot, epi, hl, ha, b);
  else if (S == 3) { if (range == 0) hot_launch_one<3, 1, 2, HOT_CS8, 4>(rot, epi, hl, ha, b); else hot_launch_one<3, 0, 3, HOT_CS8, 4>(rot, epi, hl, ha, b); }
  else if (S == 5) { if (range == 0) hot_launch_one<5, 1, 3, HOT_CS8, 4>(rot, epi, hl, ha, b); else hot_launch_one<5, 0, 5, HOT_CS8, 4>(rot, epi, hl, ha, b); }
  else {
    switch (range) {
      case 0: hot_launch_one<9, 3, 3, HOT_CS8, 4>(rot, epi, hl, ha, b); break;
      case 1: hot_launch_one<9, 2, 5, HOT_CS8, 4>(rot, epi, hl, ha, b); break;
      case 2: hot_launch_one<9, 1, 7, HOT_CS8, 4>(rot, epi, hl, ha, b); break;
      default: hot_launch_one<9, 0, 9, HOT_CS8, 4>(rot, epi, hl, ha, b); break;
    }
  }
}
}  // namespace sdrhip
