// bdx_wave_rev.hip — the known-trim instantiations of the wave-autonomous kernel with REVERSED sweeps (bdx_wave.hip, KEND = 2:
// configs with a trim_side = 3 pass) and their launchers, in a translation unit of their own so that the sets of
// instantiations compile side by side.
#define BDX_WAVE_TU_KREV 1
#include "bdx_wave.hip"
