// Host-only translation unit of libabcsmc_hip.so: the serial Walker alias build (alias_host.h), kept out of the HIP
// sources because its vector helper is multiversioned (AVX2 / baseline), which the device pass cannot parse.
#include "alias_host.h"

void abc_alias_preproc(size_t K, const double* w, double* F, uint32_t* A, double* E, uint32_t* smalls, uint32_t* bigs, bool knuth) {
    alias_preproc(K, w, F, A, E, smalls, bigs, knuth);
}
