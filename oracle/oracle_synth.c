/* oracle_synth.c — exports the header-only synthetic generator (include/wm_synth.h) from the oracle .so so
 * tests can check the numpy restatement (whisper.mojo_amd/synth.py) against it.  TEST INFRASTRUCTURE ONLY. */
#include "wm_synth.h"
size_t wo_synth_count(const wm_dims* c) { return wm_synth_count(c); }
size_t wo_synth_fill(const wm_dims* c, uint64_t seed, float* out) { return wm_synth_fill(c, seed, out); }
void wo_synth_mel(uint64_t seed, int n_mels, int n_frames, float* out) { wm_synth_mel(seed, n_mels, n_frames, out); }
