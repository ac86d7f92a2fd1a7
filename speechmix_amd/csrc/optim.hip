// Flat-buffer optimizer step for the data-parallel training loop: all trainable parameters live in ONE
// fp32 buffer (and their gradients in another), so gradient clipping + update + refresh of the bf16
// compute copies is two launches instead of one small launch per tensor (458 tensors at
// wav2vec2-base + bart-base).  The caller (HF Trainer in the reference, ref:train.py:291-330) clips to
// max_grad_norm and steps the optimizer after the data-parallel gradient all-reduce.
#include "smx_common.h"

// Sum of squares in a FIXED order (round 3): data-parallel replicas derive the clip coefficient from it and must get the same
// bits from the same all-reduced gradient; the block sums used to meet in one fp32 atomic, in arrival order.  Each block now
// stores its sum and a second one-block launch adds the stored sums in block order (a "last block adds" form in one launch
// measured 220 instead of 156 us: its per-block device-scope fence writes the L2 back 2 048 times).
#define SUMSQ_MAX_BLOCKS 2048
static __device__ float smx_sumsq_part[SUMSQ_MAX_BLOCKS];

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long long n) {
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
    if (threadIdx.x == 0) smx_sumsq_part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void sumsq_fold_kernel(int nblocks, float* __restrict__ out) {
    __shared__ float sh[16];
    float a = 0.f;                                         // thread t: blocks t, t + 256, ... in order; then the fixed block tree
    for (int b = threadIdx.x; b < nblocks; b += 256) a += smx_sumsq_part[b];
    a = block_sum(a, sh);
    if (threadIdx.x == 0) *out = a;
}
// out (one float) = sum g^2.  One call in flight per device (the partials are a per-device static).
extern "C" int smx_sumsq(const float* g, long long n, float* out, hipStream_t stream) {
    (void)hipGetLastError();  // drop stale errors left by other runtime users
    if (n <= 0) { hipMemsetAsync(out, 0, sizeof(float), stream); return SMX_OK; }
    long long blocks = (n / 4 + 255) / 256;
    if (blocks > SUMSQ_MAX_BLOCKS) blocks = SUMSQ_MAX_BLOCKS;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, stream, g, n);
    hipLaunchKernelGGL(sumsq_fold_kernel, dim3(1), dim3(256), 0, stream, (int)blocks, out);
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

__device__ __forceinline__ float opt_one(const SmxOptParams& o, float pi, float gi, float& mi, float& vi) {
    if (o.kind == 1) {
        mi = o.beta1 * mi + (1.f - o.beta1) * gi;
        vi = o.beta2 * vi + (1.f - o.beta2) * gi * gi;
        pi -= o.lr * o.weight_decay * pi;
        pi -= o.lr * (mi / o.bias_c1) / (sqrtf(vi / o.bias_c2) + o.eps);
    } else {
        float d = gi + o.weight_decay * pi;
        if (o.m) {
            d = o.beta1 * mi + d;
            mi = d;
        }
        pi -= o.lr * d;
    }
    return pi;
}

// 16-B accesses on every stream (p, g, m, v: float4; bf16 copy: 8 B); ranges are 64-element aligned.
__global__ __launch_bounds__(256) void opt_kernel(SmxOptParams o) {
    float clip = o.grad_scale;
    if (o.max_grad_norm > 0.f && o.gnorm_sq) {
        const float nrm = sqrtf(*o.gnorm_sq) * o.grad_scale;
        clip *= fminf(1.f, o.max_grad_norm / (nrm + 1e-6f));
    }
    bf16_t* sh = reinterpret_cast<bf16_t*>(o.shadow);
    const bool vec = ((reinterpret_cast<uintptr_t>(o.p) | reinterpret_cast<uintptr_t>(o.g) | reinterpret_cast<uintptr_t>(o.m) |
                       reinterpret_cast<uintptr_t>(o.v)) & 15) == 0 && (reinterpret_cast<uintptr_t>(o.shadow) & 7) == 0;
    long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const long long stride = (long long)gridDim.x * blockDim.x * 4;
    for (; i < o.n; i += stride) {
        if (vec && i + 4 <= o.n) {
            float4 p4 = *reinterpret_cast<float4*>(o.p + i);
            const float4 g4 = *reinterpret_cast<const float4*>(o.g + i);
            float4 m4 = o.m ? *reinterpret_cast<float4*>(o.m + i) : make_float4(0, 0, 0, 0);
            float4 v4 = o.v ? *reinterpret_cast<float4*>(o.v + i) : make_float4(0, 0, 0, 0);
            p4.x = opt_one(o, p4.x, g4.x * clip, m4.x, v4.x);
            p4.y = opt_one(o, p4.y, g4.y * clip, m4.y, v4.y);
            p4.z = opt_one(o, p4.z, g4.z * clip, m4.z, v4.z);
            p4.w = opt_one(o, p4.w, g4.w * clip, m4.w, v4.w);
            *reinterpret_cast<float4*>(o.p + i) = p4;
            if (o.m) *reinterpret_cast<float4*>(o.m + i) = m4;
            if (o.v) *reinterpret_cast<float4*>(o.v + i) = v4;
            if (sh) *reinterpret_cast<uint2*>(sh + i) = make_uint2(pack_bf2(p4.x, p4.y), pack_bf2(p4.z, p4.w));
        } else {
            const int cnt = (int)min((long long)4, o.n - i);
            for (int j = 0; j < cnt; ++j) {
                const long long q = i + j;
                float mi = o.m ? o.m[q] : 0.f, vi = o.v ? o.v[q] : 0.f;
                const float pi = opt_one(o, o.p[q], o.g[q] * clip, mi, vi);
                o.p[q] = pi;
                if (o.m) o.m[q] = mi;
                if (o.v) o.v[q] = vi;
                if (sh) sh[q] = f2bf(pi);
            }
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
