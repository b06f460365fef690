// bdx_wave_win.hip — the WINDOW-mode instantiations of the wave-autonomous kernel (bdx_wave.hip, WINM: single-pass known-score
// configs whose ref_search_range window is much shorter than their reads — scattered tiles that hold only the windows) and
// their launcher, in a translation unit of their own so that the sets of instantiations compile side by side.
#define BDX_WAVE_TU_WIN 1
#include "bdx_wave.hip"
