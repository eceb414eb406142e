// Library-wide state of the C-ABI: last-error string, version, dropout RNG state helpers.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

extern "C" void vqh_set_error(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* vqh_last_error(void) { return g_err; }
extern "C" int vqh_abi_version(void) { return 1; }

__global__ void rng_advance_kernel(unsigned long long* st) {
    if (threadIdx.x == 0 && blockIdx.x == 0) st[1] += 1ull;
}
// rng_state = {seed, step}: advance the step once per training step (captured inside the step graph).
extern "C" int vqh_rng_advance(unsigned long long* rng_state, hipStream_t stream) {
    VQH_CHECK_ARG(rng_state != nullptr, "vqh_rng_advance: null state");
    hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(64), 0, stream, rng_state);
    VQH_LAUNCH_CHECK();
    return VQH_OK;
}
