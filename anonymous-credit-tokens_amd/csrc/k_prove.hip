// placeholder until the prove_spend kernels land (this round)
#include "../../include/act_mi355x.h"
extern "C" int act_prove_spend_batch(act_ctx*, size_t, int, const uint8_t*, const uint8_t*, const uint8_t*, uint8_t*, uint8_t*, uint8_t*) { return ACT_ERR_ARG; }
