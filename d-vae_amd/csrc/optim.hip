// Adam over ONE flat fp32 buffer (parameters, gradients and both moments laid out alike), the
// update rule of torch.optim.Adam as the reference uses it (train.py:174: lr 1e-4, betas
// (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad):
//   m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2
//   p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// HBM-bound: 4 reads + 3 writes of 4 bytes per parameter, one launch per step.
#include "common.h"
#include "dvae_hip.h"
#include <math.h>

namespace dvae {

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long n4, float b1,
                                                   float b2, float eps, float step_size, float inv_sqrt_bc2,
                                                   int zero_grad) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 P = reinterpret_cast<float4*>(p)[i], G = reinterpret_cast<const float4*>(g)[i];
        float4 M = reinterpret_cast<float4*>(m)[i], V = reinterpret_cast<float4*>(v)[i];
#define ADAM1(f)                                              \
    {                                                         \
        M.f = M.f + (1.f - b1) * (G.f - M.f);                 \
        V.f = b2 * V.f + (1.f - b2) * G.f * G.f;              \
        float denom = sqrtf(V.f) * inv_sqrt_bc2 + eps;        \
        P.f -= step_size * (M.f / denom);                     \
    }
        ADAM1(x) ADAM1(y) ADAM1(z) ADAM1(w)
#undef ADAM1
        reinterpret_cast<float4*>(p)[i] = P;
        reinterpret_cast<float4*>(m)[i] = M;
        reinterpret_cast<float4*>(v)[i] = V;
        if (zero_grad) reinterpret_cast<float4*>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);  // the next step's zero_grad
    }
}

// The same update with lr and the step count read from DEVICE memory, so that the launch carries no per-step scalar
// and a captured step can be replayed: state = {lr (float), step count so far (int32, as float bits are NOT used: the
// second word is an int)}.  One lane per workgroup turns the count into the two bias-correction factors (in double,
// as the host path does); adam_advance_kernel, launched behind this one, increments the count.
__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v, long n4, float b1,
                                                       float b2, float eps, const float* __restrict__ lr_dev,
                                                       const int* __restrict__ step_dev, int zero_grad) {
    __shared__ float hyp[2];
    if (threadIdx.x == 0) {
        const double t = (double)(step_dev[0] + 1);
        const double bc1 = 1.0 - pow((double)b1, t), bc2 = 1.0 - pow((double)b2, t);
        hyp[0] = (float)((double)lr_dev[0] / bc1);
        hyp[1] = (float)(1.0 / sqrt(bc2));
    }
    __syncthreads();
    const float step_size = hyp[0], inv_sqrt_bc2 = hyp[1];
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 P = reinterpret_cast<float4*>(p)[i], G = reinterpret_cast<const float4*>(g)[i];
        float4 M = reinterpret_cast<float4*>(m)[i], V = reinterpret_cast<float4*>(v)[i];
#define ADAM1(f)                                              \
    {                                                         \
        M.f = M.f + (1.f - b1) * (G.f - M.f);                 \
        V.f = b2 * V.f + (1.f - b2) * G.f * G.f;              \
        float denom = sqrtf(V.f) * inv_sqrt_bc2 + eps;        \
        P.f -= step_size * (M.f / denom);                     \
    }
        ADAM1(x) ADAM1(y) ADAM1(z) ADAM1(w)
#undef ADAM1
        reinterpret_cast<float4*>(p)[i] = P;
        reinterpret_cast<float4*>(m)[i] = M;
        reinterpret_cast<float4*>(v)[i] = V;
        if (zero_grad) reinterpret_cast<float4*>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
__global__ void adam_advance_kernel(int* step_dev) { step_dev[0] += 1; }

}  // namespace dvae

extern "C" int dvae_adam_step_dev_f32(float* p, float* g, float* m, float* v, int64_t n, const float* lr_dev,
                                      float beta1, float beta2, float eps, int32_t* step_dev, int32_t zero_grad,
                                      void* stream) {
    using namespace dvae;
    if (n == 0) return DVAE_OK;
    DVAE_REQUIRE(p && g && m && v && lr_dev && step_dev, "adam_step_dev: null pointer");
    DVAE_REQUIRE(n > 0 && n % 4 == 0, "adam_step_dev: the flat length must be a multiple of 4 floats");
    DVAE_REQUIRE(((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16 == 0, "adam_step_dev: buffers must be 16-byte aligned");
    long n4 = n / 4;
    long blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adam_dev_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n4, beta1,
                       beta2, eps, lr_dev, step_dev, zero_grad != 0);
    DVAE_LAUNCH_CHECK();
    hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev);
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}

extern "C" int dvae_adam_step_f32(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1,
                                  float beta2, float eps, int32_t step, int32_t zero_grad, void* stream) {
    using namespace dvae;
    if (n == 0) return DVAE_OK;
    DVAE_REQUIRE(p && g && m && v, "adam_step: null pointer");
    DVAE_REQUIRE(n > 0 && n % 4 == 0, "adam_step: the flat length must be a multiple of 4 floats");
    DVAE_REQUIRE(((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16 == 0, "adam_step: buffers must be 16-byte aligned");
    DVAE_REQUIRE(step >= 1, "adam_step: step counts from 1");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    long n4 = n / 4;
    long blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n4, beta1,
                       beta2, eps, step_size, inv_sqrt_bc2, zero_grad != 0);
    DVAE_LAUNCH_CHECK();
    return DVAE_OK;
}
