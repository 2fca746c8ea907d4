// Fused on-chip backend (placeholder: plan creation succeeds, nothing is supported yet).
#include "ldpc_common.hpp"

namespace ldpc {
struct FusedPlan {};
int fused_plan_create(Decoder*) { return LDPC_OK; }
void fused_plan_destroy(Decoder*) {}
bool fused_supported(const Decoder*) { return false; }
int fused_decode(Decoder*, const void*, const uint8_t*, int64_t, int32_t, uint32_t, uint8_t*, int32_t*, hipStream_t) {
    set_error("fused backend not available");
    return LDPC_E_UNSUPPORTED;
}
}  // namespace ldpc
