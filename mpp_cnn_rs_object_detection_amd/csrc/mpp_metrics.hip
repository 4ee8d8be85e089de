// mpp_metrics.hip -- rotated-rectangle IoU matrix for the DOTA task-1 (oriented boxes) evaluation.
//
// Replaces the SWIG `polyiou.iou_poly` + the axis-aligned pre-filter of DOTA_devkit's
// dota_evaluation_task1.voc_eval, which the reference calls from metrics/dota_eval.py:37-47 (the devkit is an
// un-vendored clone, README.md:22-30).  One lane per (detection, ground-truth) pair; the clipper is the
// Sutherland-Hodgman device function of the sampler's overlap prior.
#include "mpp_device.hpp"

// quad q[8] = x1 y1 .. x4 y4 -> counter-clockwise corner arrays, |area| returned
__device__ __forceinline__ double load_ccw(const double *q, double *x, double *y) {
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int j = (i + 1) & 3;
    s += q[2 * i] * q[2 * j + 1] - q[2 * j] * q[2 * i + 1];
  }
  const bool ccw = s >= 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int k = ccw ? i : 3 - i;
    x[i] = q[2 * k]; y[i] = q[2 * k + 1];
  }
  return 0.5 * fabs(s);
}

__global__ __launch_bounds__(256) void k_quad_iou(int n, const double *__restrict__ a, int m, const double *__restrict__ b,
                                                  double *__restrict__ out) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)n * m) return;
  const int i = (int)(idx / m), j = (int)(idx % m);
  double qa[8], qb[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { qa[k] = a[(size_t)i * 8 + k]; qb[k] = b[(size_t)j * 8 + k]; }
  // axis-aligned pre-filter with the devkit's inclusive-pixel (+1) extents
  double axmin = qa[0], axmax = qa[0], aymin = qa[1], aymax = qa[1], bxmin = qb[0], bxmax = qb[0], bymin = qb[1], bymax = qb[1];
#pragma unroll
  for (int k = 1; k < 4; ++k) {
    axmin = fmin(axmin, qa[2 * k]); axmax = fmax(axmax, qa[2 * k]); aymin = fmin(aymin, qa[2 * k + 1]); aymax = fmax(aymax, qa[2 * k + 1]);
    bxmin = fmin(bxmin, qb[2 * k]); bxmax = fmax(bxmax, qb[2 * k]); bymin = fmin(bymin, qb[2 * k + 1]); bymax = fmax(bymax, qb[2 * k + 1]);
  }
  const double iw = fmax(fmin(axmax, bxmax) - fmax(axmin, bxmin) + 1.0, 0.0);
  const double ih = fmax(fmin(aymax, bymax) - fmax(aymin, bymin) + 1.0, 0.0);
  if (!(iw * ih > 0.0)) { out[idx] = -1.0; return; }
  double ax[4], ay[4], bx[4], by[4];
  const double A = load_ccw(qa, ax, ay), B = load_ccw(qb, bx, by);
  const double inter = clip_area(ax, ay, bx, by);
  const double uni = A + B - inter;
  out[idx] = uni == 0.0 ? (inter + 1.0) / (uni + 1.0) : inter / uni;
}

extern "C" void mpp_launch_quad_iou(hipStream_t st, int n, const double *a, int m, const double *b, double *out) {
  const long long total = (long long)n * m;
  if (total <= 0) return;
  const int block = 256;
  hipLaunchKernelGGL(k_quad_iou, dim3((unsigned)((total + block - 1) / block)), dim3(block), 0, st, n, a, m, b, out);
}
