// bdx_pairs.hip — the pairs-mode instantiations of the wave-autonomous kernel (bdx_wave.hip, KB > 0) and their launcher,
// in a translation unit of their own so that the two sets of instantiations compile side by side.
#define BDX_WAVE_TU_PAIRS 1
#include "bdx_wave.hip"
