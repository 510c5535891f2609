// Error plumbing and the small host-side helpers of the C ABI.
#include <math.h>
#include <stdarg.h>

#include "tmf_common.h"

namespace tmf {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return TMF_OK;
    set_error("%s: %s", what, hipGetErrorString(e));
    return TMF_E_LAUNCH;
}

}  // namespace tmf

extern "C" int tmf_version(void) { return TMF_VERSION; }

extern "C" const char* tmf_last_error(void) { return tmf::g_err; }

extern "C" int tmf_padded_ld(int n_components) { return tmf::row_geom(n_components).ld; }

extern "C" int tmf_padded_ld_bf16(int n_components) { return tmf::row_geom_bf16(n_components).ld; }

extern "C" tmf_adam tmf_adam_fresh(float lr) {
    // fp32 throughout, like tf.keras.optimizers.Adam at iterations == 0 (beta^1 = beta)
    const float one = 1.0f, b1 = 0.9f, b2 = 0.999f;
    tmf_adam a;
    a.one_minus_b1 = one - b1;
    a.one_minus_b2 = one - b2;
    a.alpha = lr * sqrtf(a.one_minus_b2) / a.one_minus_b1;
    a.eps = 1e-7f;
    return a;
}

extern "C" tmf_adam tmf_adam_step(float lr, int step) {
    // Keras Adam at iteration `step` (1-based) in fp32: alpha_t = lr * sqrt(1 - b2^t) / (1 - b1^t); step 1 == tmf_adam_fresh
    const float one = 1.0f, b1 = 0.9f, b2 = 0.999f;
    tmf_adam a;
    a.one_minus_b1 = one - b1;
    a.one_minus_b2 = one - b2;
    const float b1p = step <= 1 ? b1 : powf(b1, (float)step), b2p = step <= 1 ? b2 : powf(b2, (float)step);
    a.alpha = lr * sqrtf(one - b2p) / (one - b1p);
    a.eps = 1e-7f;
    return a;
}
