// bdx_wave_end.hip — the known-end instantiations of the wave-autonomous kernel (bdx_wave.hip, KEND) and their launcher, in
// a translation unit of their own so that the sets of instantiations compile side by side.
#define BDX_WAVE_TU_KEND 1
#include "bdx_wave.hip"
