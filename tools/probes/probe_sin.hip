// Accuracy of the hardware v_sin_f32 (input in revolutions) against sin(2*pi*y) in double, on y in [-0.5, 0.5] and on the
// folded range [-0.25, 0.25], next to the polynomial of njf_device.h::sin_accurate.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float* y, float* hw, float* hw_fold, float* poly, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = y[i];
  hw[i] = __builtin_amdgcn_sinf(v);
  float f = v;
  if (fabsf(f) > 0.25f) f = copysignf(0.5f, f) - f;
  hw_fold[i] = __builtin_amdgcn_sinf(f);
  const float y2 = f * f;
  float p = 3.8199525848482803f;
  p = fmaf(p, y2, -15.094642576822984f);
  p = fmaf(p, y2, 42.058693944897634f);
  p = fmaf(p, y2, -76.70585975306136f);
  p = fmaf(p, y2, 81.60524927607504f);
  p = fmaf(p, y2, -41.341702240399755f);
  p = fmaf(p, y2, 6.283185307179586f);
  poly[i] = f * p;
}
int main() {
  const int n = 1 << 22;
  std::vector<float> y(n);
  for (int i = 0; i < n; ++i) y[i] = -0.5f + (float)i / (float)(n - 1);
  float *dy, *d1, *d2, *d3;
  hipMalloc(&dy, 4 * n); hipMalloc(&d1, 4 * n); hipMalloc(&d2, 4 * n); hipMalloc(&d3, 4 * n);
  hipMemcpy(dy, y.data(), 4 * n, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dy, d1, d2, d3, n);
  std::vector<float> a(n), b(n), c(n);
  hipMemcpy(a.data(), d1, 4 * n, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d2, 4 * n, hipMemcpyDeviceToHost); hipMemcpy(c.data(), d3, 4 * n, hipMemcpyDeviceToHost);
  double e1 = 0, e2 = 0, e3 = 0;
  for (int i = 0; i < n; ++i) {
    const double r = sin(2.0 * M_PI * (double)y[i]);
    e1 = fmax(e1, fabs(a[i] - r)); e2 = fmax(e2, fabs(b[i] - r)); e3 = fmax(e3, fabs(c[i] - r));
  }
  printf("max abs error vs sin(2 pi y), y in [-0.5, 0.5]: v_sin_f32 %.3e   v_sin_f32 on the folded argument %.3e   degree-13 polynomial %.3e\n", e1, e2, e3);
  return 0;
}
