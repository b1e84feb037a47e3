// Accuracy of the hardware sine (v_sin_f32: sin(2 pi x), preceded by v_fract_f32; the product feeds it revolutions
// directly, here x = w / 2 for a w in half-revolutions so that both candidates see the same argument) against a double-precision sine, next to
// the product's polynomial (film_sine in csrc/mlp.hpp: half-revolutions, rint, degree-4 minimax in r^2).  The SDF trunk
// feeds sin(30 (f (W x + b) + phi)) through six layers into a root finder with a 1e-5 m threshold: what matters is the
// absolute error of one activation in [-1, 1].
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void k_sin(const float* w, int n, float* hw, float* poly, float* hwc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = w[i];                                   // half-revolutions: value = sin(pi x)
    hw[i] = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(0.5f * x));
    hwc[i] = __builtin_amdgcn_cosf(__builtin_amdgcn_fractf(0.5f * x));
    const float q = rintf(x), r = x - q, r2 = r * r;
    float p = fmaf(r2, 0.0772201280771219f, -0.5980451736306471f);
    p = fmaf(p, r2, 2.550031377188653f);
    p = fmaf(p, r2, -5.167706878920042f);
    p = fmaf(p, r2, 3.1415925800446054f);
    poly[i] = __uint_as_float(__float_as_uint(p * r) ^ ((unsigned)(int)q << 31));
}

int main() {
    const int n = 1 << 22;
    std::vector<float> w(n), a(n), b(n), c(n);
    unsigned s = 12345u;
    for (int i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        const float u = (float)(s >> 8) / 16777216.0f;      // [0,1)
        const float range = (i & 3) == 0 ? 2.0f : ((i & 3) == 1 ? 20.0f : ((i & 3) == 2 ? 100.0f : 600.0f));
        w[i] = (2.0f * u - 1.0f) * range;
    }
    float *dw, *da, *db, *dc;
    hipMalloc(&dw, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4);
    hipMemcpy(dw, w.data(), n * 4, hipMemcpyHostToDevice);
    k_sin<<<(n + 255) / 256, 256>>>(dw, n, da, db, dc);
    hipMemcpy(a.data(), da, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), db, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
    const char* names[4] = {"|w| < 2", "|w| < 20", "|w| < 100", "|w| < 600"};
    for (int k = 0; k < 4; ++k) {
        double eh = 0, ep = 0, sh = 0, sp = 0, ec = 0;
        long cnt = 0;
        for (int i = k; i < n; i += 4) {
            const double ref = sin(M_PI * (double)w[i]);
            const double dh = fabs((double)a[i] - ref), dp = fabs((double)b[i] - ref);
            const double dc2 = fabs((double)c[i] - cos(M_PI * (double)w[i]));
            ec = dc2 > ec ? dc2 : ec;
            eh = dh > eh ? dh : eh; ep = dp > ep ? dp : ep; sh += dh * dh; sp += dp * dp; ++cnt;
        }
        printf("%-10s  v_fract+v_sin: max abs err %.3e rms %.3e   polynomial: max abs err %.3e rms %.3e   v_fract+v_cos: max abs err %.3e\n",
               names[k], eh, sqrt(sh / cnt), ep, sqrt(sp / cnt), ec);
    }
    return 0;
}
