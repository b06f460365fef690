// bdx_wave_aln.hip — the known-ALIGNMENT instantiations of the wave-autonomous kernel (bdx_wave.hip, KEND = 3: start and end of
// every pass's winner by anchored sweeps — per-pass position outputs and the DemuxStats histograms without the exact kernel)
// and their launcher, in a translation unit of their own so that the sets of instantiations compile side by side.
#define BDX_WAVE_TU_ALN 1
#include "bdx_wave.hip"
