// Flat-buffer optimizer step for the data-parallel training loop: all trainable parameters live in ONE
// fp32 buffer (and their gradients in another), so gradient clipping + update + refresh of the bf16
// compute copies is two launches instead of one small launch per tensor (458 tensors at
// wav2vec2-base + bart-base).  The caller (HF Trainer in the reference, ref:train.py:291-330) clips to
// max_grad_norm and steps the optimizer after the data-parallel gradient all-reduce.
#include "smx_common.h"

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long long n, float* __restrict__ out) {
    __shared__ float sh[16];
    float s = 0.f;
    long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const long long stride = (long long)gridDim.x * blockDim.x * 4;
    for (; i + 4 <= n; i += stride) {
        const float4 v = *reinterpret_cast<const float4*>(g + i);
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (i < n && i + 4 > n)
        for (long long j = i; j < n; ++j) s += g[j] * g[j];
    s = block_sum(s, sh);
    if (threadIdx.x == 0) atomicAdd(out, s);
}
// out (one float, zeroed here) = sum g^2
extern "C" int smx_sumsq(const float* g, long long n, float* out, hipStream_t stream) {
    (void)hipGetLastError();  // drop stale errors left by other runtime users
    hipMemsetAsync(out, 0, sizeof(float), stream);
    if (n <= 0) return SMX_OK;
    long long blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, stream, g, n, out);
    SMX_CHECK_LAUNCH();
}

struct SmxOptParams {
    float* p;                 // fp32 master parameters
    const float* g;           // fp32 gradients (already summed over ranks)
    float* m;                 // first moment (AdamW) / momentum (SGD) or null
    float* v;                 // second moment (AdamW) or null
    void* shadow;             // bf16 compute copy of p (same offsets) or null
    const float* gnorm_sq;    // device scalar: sum g^2 over the whole buffer (for clipping) or null
    long long n;
    float lr, beta1, beta2, eps, weight_decay;
    float bias_c1, bias_c2;   // 1 - beta^t
    float grad_scale;         // multiplies g (1 / world_size, 1 / grad_accum ...)
    float max_grad_norm;      // <= 0: no clipping
    int kind;                 // 0: SGD(+momentum beta1), 1: AdamW
};

__global__ __launch_bounds__(256) void opt_kernel(SmxOptParams o) {
    float clip = o.grad_scale;
    if (o.max_grad_norm > 0.f && o.gnorm_sq) {
        const float nrm = sqrtf(*o.gnorm_sq) * o.grad_scale;
        clip *= fminf(1.f, o.max_grad_norm / (nrm + 1e-6f));
    }
    bf16_t* sh = reinterpret_cast<bf16_t*>(o.shadow);
    long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const long long stride = (long long)gridDim.x * blockDim.x * 4;
    for (; i < o.n; i += stride) {
        const int cnt = (int)min((long long)4, o.n - i);
        for (int j = 0; j < cnt; ++j) {
            const long long q = i + j;
            const float gi = o.g[q] * clip;
            float pi = o.p[q];
            if (o.kind == 1) {
                const float mi = o.beta1 * o.m[q] + (1.f - o.beta1) * gi;
                const float vi = o.beta2 * o.v[q] + (1.f - o.beta2) * gi * gi;
                o.m[q] = mi;
                o.v[q] = vi;
                pi -= o.lr * o.weight_decay * pi;
                pi -= o.lr * (mi / o.bias_c1) / (sqrtf(vi / o.bias_c2) + o.eps);
            } else {
                float d = gi + o.weight_decay * pi;
                if (o.m) {
                    d = o.beta1 * o.m[q] + d;
                    o.m[q] = d;
                }
                pi -= o.lr * d;
            }
            o.p[q] = pi;
            if (sh) sh[q] = f2bf(pi);
        }
    }
}
extern "C" int smx_optimizer_step(const SmxOptParams* op, hipStream_t stream) {
    (void)hipGetLastError();  // drop stale errors left by other runtime users
    SmxOptParams o = *op;
    if (o.n <= 0) return SMX_OK;
    if (o.kind == 1 && (!o.m || !o.v)) return SMX_EINVAL;
    long long blocks = (o.n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(opt_kernel, dim3(blocks), dim3(256), 0, stream, o);
    SMX_CHECK_LAUNCH();
}

// ABI self-description (checked by the ctypes binding against its struct mirrors)
extern "C" int smx_sizeof_SmxOptParams(void) { return (int)sizeof(SmxOptParams); }
