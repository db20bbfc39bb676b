// placeholder until the fixed-base kernels land (this round)
#include "ozk_common.h"
using namespace ozk;
extern "C" {
int ozk_fixed_batch_msm_host(int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, const uint8_t*, const uint8_t*, int32_t, int32_t, uint8_t*) { return fail(OZK_E_INTERNAL, "fixed-base MSM not implemented yet"); }
int ozk_fixed_double_batch_msm_host(int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, const uint8_t*, const uint8_t*, const uint8_t*, int32_t, uint8_t*) { return fail(OZK_E_INTERNAL, "fixed-base MSM not implemented yet"); }
int ozk_field_batch_mul_host(const uint8_t*, int32_t, int32_t, uint8_t*) { return fail(OZK_E_INTERNAL, "not implemented yet"); }
size_t ozk_fixed_batch_msm_workspace_bytes(int32_t, int32_t, int32_t, int32_t) { return 0; }
int ozk_fixed_batch_msm_dev(int32_t, int32_t, int32_t, const void*, const void*, int32_t, void*, void*, size_t, void*) { return fail(OZK_E_INTERNAL, "not implemented yet"); }
int ozk_field_batch_mul_dev(const void*, int32_t, void*, void*) { return fail(OZK_E_INTERNAL, "not implemented yet"); }
}
