// Shared helpers for the D-VAE HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

namespace dvae {

// thread-local last-error text, returned by dvae_last_error()
char* err_buf();
int set_err(int code, const char* fmt, ...);
void count_launch();  // api.cpp: kernel launches of this process (dvae_debug_launch_count)

#define DVAE_OK 0
#define DVAE_ERR_ARG -1
#define DVAE_ERR_HIP -2
#define DVAE_ERR_SIZE -3

#define DVAE_REQUIRE(cond, ...)                                   \
    do {                                                          \
        if (!(cond)) return ::dvae::set_err(DVAE_ERR_ARG, __VA_ARGS__); \
    } while (0)

#define DVAE_HIP(call)                                                          \
    do {                                                                        \
        hipError_t e__ = (call);                                                \
        if (e__ != hipSuccess)                                                  \
            return ::dvae::set_err(DVAE_ERR_HIP, "%s failed: %s (%s:%d)", #call, \
                                   hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

#define DVAE_LAUNCH_CHECK()                                                              \
    do {                                                                                 \
        ::dvae::count_launch();                                                          \
        hipError_t e__ = hipGetLastError();                                              \
        if (e__ != hipSuccess)                                                           \
            return ::dvae::set_err(DVAE_ERR_HIP, "kernel launch failed: %s (%s:%d)",     \
                                   hipGetErrorString(e__), __FILE__, __LINE__);          \
    } while (0)

#define DVAE_TRY(expr)            \
    do {                          \
        int rc__ = (expr);        \
        if (rc__ != 0) return rc__; \
    } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace dvae
