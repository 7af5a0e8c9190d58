// placeholder until the client-verifier kernels land (this round)
#include "../../include/act_mi355x.h"
extern "C" int act_issuance_to_credit_token_batch(act_ctx*, size_t, int, const uint8_t*, const uint8_t*, const uint8_t*, const uint8_t*, uint8_t*, uint8_t*) { return ACT_ERR_ARG; }
extern "C" int act_refund_to_credit_token_batch(act_ctx*, size_t, int, const uint8_t*, const uint8_t*, const uint8_t*, const uint8_t*, uint8_t*, uint8_t*) { return ACT_ERR_ARG; }
