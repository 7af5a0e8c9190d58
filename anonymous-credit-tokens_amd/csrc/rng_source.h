// rng_source.h — ACT_RNG_CALLBACK (include/act_mi355x.h): the crate hands its methods an `impl CryptoRngCore`
// (/root/reference/src/lib.rs:781-786) and draws e, alpha only after the checks have passed (:842-852).  A caller that must leave
// its generator in the state a sequential loop would cannot pre-draw bytes for lanes that may be rejected, so the entry points that
// know every verdict before they sign accept the generator itself: one draw(ctx, dst, 128 * k) for the k lanes that will be signed,
// after which the bytes are an ordinary ACT_RNG_SEQUENTIAL stream.  Shared by engine.hip and node.cpp.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>
#include "../../include/act_mi355x.h"

namespace act {
struct DrawnRng {
  std::vector<uint8_t> buf;
  ~DrawnRng() { wipe(); }
  void wipe() {                                    // signing-nonce seeds: not left on the heap
    if (buf.empty()) return;
    volatile uint8_t* p = buf.data();
    for (size_t i = 0; i < buf.size(); i++) p[i] = 0;
  }
  // on return (rng, mode) name host bytes in ACT_RNG_PER_LANE / ACT_RNG_SEQUENTIAL form; `signed_lanes` = lanes with verdict 0
  int resolve(const uint8_t*& rng, int& mode, size_t signed_lanes) {
    if (mode != ACT_RNG_CALLBACK) return (mode == ACT_RNG_PER_LANE || mode == ACT_RNG_SEQUENTIAL) ? ACT_OK : ACT_ERR_ARG;
    const act_rng_source* src = reinterpret_cast<const act_rng_source*>(rng);
    if (!src || !src->draw) return ACT_ERR_ARG;
    buf.assign(signed_lanes * 128 + 16, 0);        // never empty: the signing calls want a non-null pointer
    // a draw that failed must never turn into a signature over zero nonces (e = alpha = 0 makes z = gamma * x: the key): fail the call
    if (signed_lanes && src->draw(src->rng_ctx, buf.data(), signed_lanes * 128) != 0) { wipe(); return ACT_ERR_RNG; }
    rng = buf.data(); mode = ACT_RNG_SEQUENTIAL;
    return ACT_OK;
  }
};
}  // namespace act
