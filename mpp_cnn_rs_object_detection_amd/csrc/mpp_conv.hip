// mpp_conv.hip -- the 3x3 convolutions of the U-Nets' full-resolution level (32 output channels) on the matrix cores.
//
// DoubleConv of model_parts/unet/unet_parts.py:12-31: Conv2d(3x3, padding_mode='reflect') + BatchNorm2d(eval) + ReLU, twice;
// Up (:48-67) feeds it cat([skip, up]).  On a 4096 x 4096 image the four such convolutions per network that produce 32
// channels at full resolution (32 -> 32 in the first level, 64 -> 32 and 32 -> 32 in the last) are 2.5 of the forward's
// 9.5 TFLOP and took 56 of its 147 ms: the library's kernels for N = 32 reach 35 - 60 TFLOP/s (profiles/r03_unet_pmc.md),
// its N >= 64 kernels 125 - 130.  This kernel is written for exactly that shape:
//   * implicit GEMM on v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: an exact k-ordered fmaf chain, the vector rate of
//     157 TFLOP/s): M = 32 consecutive pixels of a row, N = the 32 output channels, K = 9 taps x C_in;
//   * a workgroup (4 waves, one per SIMD) computes 8 rows x 64 columns of the output; a wave 2 rows = four 32 x 32
//     accumulators (64 VGPRs) that share every B (weight) fragment;
//   * the input tile with its one-pixel halo -- 10 x 66 pixels x 32 channels -- is staged in LDS once per 32 input channels
//     (pixel stride 33 floats: the 32 lanes of an A fragment read 32 different banks), the 9 x 32 x 32 weights beside it;
//     REFLECT padding is index arithmetic at the load (no padded copy of the activation), the previous layer's BatchNorm +
//     ReLU can be applied at the load as well, the concat of Up is a second source pointer (channels 32..63);
//   * the epilogue applies this layer's folded BatchNorm / bias (scale, shift) and ReLU to the accumulators and writes
//     NHWC rows of 128 B.
// HBM traffic: one read of the input (x 1.29 for the halo) and one write of the output; 72 FLOP per byte: MFMA-bound.
#include "mpp_device.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CV_ROWS 8            // output rows of a workgroup
#define CV_COLS 64           // output columns of a workgroup
#define CV_PIX 33            // floats per pixel in LDS (32 channels + 1: bank-conflict-free A fragments)
#define CV_TW (CV_COLS + 2)
#define CV_TH (CV_ROWS + 2)
#define CV_TILE_FLOATS (CV_TH * CV_TW * CV_PIX)
#define CV_W_FLOATS (9 * 32 * 32)

__device__ __forceinline__ int reflect_idx(int i, int n) {       // F.pad(mode='reflect') by one pixel
  i = i < 0 ? -i : i;
  return i >= n ? 2 * n - 2 - i : i;
}

// x0, x1: [H][W][32] float32 (x1 = nullptr: 32 input channels); wp: [C_in / 32][9][32 in][32 out]; y: [H][W][32].
// in_scale / in_shift (or nullptr): x0 <- max(0, x0 * in_scale[c] + in_shift[c]) at the load (the producer's BatchNorm + ReLU).
// out_scale / out_shift (or nullptr) and relu: the epilogue.
// Persistent workgroups: each takes a contiguous range of tiles; a "stage" = (tile, source); while the MFMAs of a stage run,
// the 21 float4 a thread contributes to the NEXT stage's LDS tile are already on their way from HBM (registers); the
// weights of both sources stay in LDS for the whole launch.
#define CV_LD ((CV_TH * CV_TW * 8 + 255) / 256)               // float4 loads per thread and stage (21)
__global__ __launch_bounds__(256, 1) void k_conv3x3_c32(const float *__restrict__ x0, const float *__restrict__ x1, int H, int W,
                                                        const float *__restrict__ wp, const float *__restrict__ in_scale,
                                                        const float *__restrict__ in_shift, const float *__restrict__ out_scale,
                                                        const float *__restrict__ out_shift, int relu, float *__restrict__ y,
                                                        int tiles_x, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *tile = lds, *wl = lds + CV_TILE_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_src = x1 ? 2 : 1;
  // consecutive workgroup ids go to different XCDs: workgroup (xcd, k) takes the k-th share of the xcd-th eighth of the tiles
  // (neighbouring tiles share halo rows in that XCD's L2)
  const int xcd = (int)(blockIdx.x % 8), k_in = (int)(blockIdx.x / 8), per_xcd = (int)(gridDim.x / 8);
  const int e0 = (int)((long long)n_tiles * xcd / 8), e1 = (int)((long long)n_tiles * (xcd + 1) / 8);
  const int t_begin = e0 + (int)((long long)(e1 - e0) * k_in / per_xcd), t_end = e0 + (int)((long long)(e1 - e0) * (k_in + 1) / per_xcd);
  for (int q = tid; q < n_src * CV_W_FLOATS / 4; q += 256) *(float4 *)(wl + 4 * q) = *(const float4 *)(wp + 4 * q);
  const int n_stage = (t_end - t_begin) * n_src;
  if (n_stage <= 0) return;

  float4 pre[CV_LD];
  // the loads of stage `st` (tile t_begin + st / n_src, source st % n_src) into registers
#define CV_ISSUE(st_)                                                                                                  \
  do {                                                                                                                  \
    const int t_ = t_begin + (st_) / n_src, ty_ = t_ / tiles_x, tx_ = t_ - ty_ * tiles_x;                               \
    const float *x_ = ((st_) % n_src) == 0 ? x0 : x1;                                                                   \
    _Pragma("unroll") for (int u = 0; u < CV_LD; ++u) {                                                                 \
      const int q = tid + 256 * u, pix = q >> 3, j = q & 7;                                                             \
      const int pr = pix / CV_TW, pc = pix - pr * CV_TW;                                                                \
      const int gr = reflect_idx(min(ty_ * CV_ROWS - 1 + pr, H), H), gc = reflect_idx(min(tx_ * CV_COLS - 1 + pc, W), W); \
      if (q < CV_TH * CV_TW * 8) pre[u] = *(const float4 *)(x_ + ((size_t)gr * W + gc) * 32 + 4 * j);                   \
    }                                                                                                                   \
  } while (0)
  CV_ISSUE(0);

  f32x16 acc[4];
  for (int st = 0; st < n_stage; ++st) {
    const int src = st % n_src, t = t_begin + st / n_src;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int r0 = ty * CV_ROWS, c0 = tx * CV_COLS;
    if (src == 0) {
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
    }
    __syncthreads();                                          // the previous stage has read its tile (and the weights are in)
    const bool aff = src == 0 && in_scale != nullptr;
#pragma unroll
    for (int u = 0; u < CV_LD; ++u) {
      const int q = tid + 256 * u, pix = q >> 3, j = q & 7;
      if (q < CV_TH * CV_TW * 8) {
        float4 v = pre[u];
        if (aff) {
          const float4 sc = *(const float4 *)(in_scale + 4 * j), sh = *(const float4 *)(in_shift + 4 * j);
          v.x = fmaxf(0.f, v.x * sc.x + sh.x); v.y = fmaxf(0.f, v.y * sc.y + sh.y);
          v.z = fmaxf(0.f, v.z * sc.z + sh.z); v.w = fmaxf(0.f, v.w * sc.w + sh.w);
        }
        float *d = tile + pix * CV_PIX + 4 * j;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      }
    }
    __syncthreads();
    if (st + 1 < n_stage) CV_ISSUE(st + 1);                   // in flight while the MFMAs below run
    // ---- 9 taps x 16 channel pairs: one B fragment, four A fragments, four MFMAs
    const int m = lane & 31, kh = lane >> 5;
    const float *arow = tile + ((2 * wave) * CV_TW + m) * CV_PIX + kh;       // block b: row 2*wave + (b >> 1), columns 32*(b & 1) + m
    const float *bw = wl + src * CV_W_FLOATS + kh * 32 + m;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap % 3;
#pragma unroll
      for (int c = 0; c < 32; c += 2) {
        const float bf = bw[(tap * 32 + c) * 32];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const float af = arow[(((b >> 1) + dy) * CV_TW + 32 * (b & 1) + dx) * CV_PIX + c];
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[b], 0, 0, 0);
        }
      }
    }
    if (src + 1 < n_src) continue;
    // ---- epilogue: C/D layout col = lane & 31 (output channel), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (pixel)
    const int n = lane & 31;
    const float sc = out_scale ? out_scale[n] : 1.f, sh = out_shift ? out_shift[n] : 0.f;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int row = r0 + 2 * wave + (b >> 1);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int col = c0 + 32 * (b & 1) + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
        float v = acc[b][i] * sc + sh;
        if (relu) v = fmaxf(0.f, v);
        if (row < H && col < W) y[((size_t)row * W + col) * 32 + n] = v;
      }
    }
  }
#undef CV_ISSUE
}

extern "C" int mpp_launch_conv3x3_c32(hipStream_t st, const float *x0, const float *x1, int H, int W, const float *wp,
                                      const float *in_scale, const float *in_shift, const float *out_scale, const float *out_shift,
                                      int relu, float *y) {
  if (H < 2 || W < 2) return -1;
  const int tiles_x = (W + CV_COLS - 1) / CV_COLS, tiles_y = (H + CV_ROWS - 1) / CV_ROWS, n_tiles = tiles_x * tiles_y;
  const size_t lds = (size_t)(CV_TILE_FLOATS + (x1 ? 2 : 1) * CV_W_FLOATS) * sizeof(float);
  // one workgroup per CU (its LDS footprint allows no more), eight at a time to the eight XCDs
  static int cus = 0;                                   // (queried once: hipGetDeviceProperties costs about a millisecond)
  if (cus == 0) {
    if (hipFuncSetAttribute((const void *)k_conv3x3_c32, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(CV_TILE_FLOATS + 2 * CV_W_FLOATS) * 4) != hipSuccess) return -2;
    int dev = 0, n = 0;
    cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
  }
  int grid = (cus / 8) * 8;
  if (grid > ((n_tiles + 7) / 8) * 8) grid = ((n_tiles + 7) / 8) * 8;
  if (grid < 8) grid = 8;
  hipLaunchKernelGGL(k_conv3x3_c32, dim3(grid), dim3(256), lds, st, x0, x1, H, W, wp, in_scale, in_shift, out_scale, out_shift, relu, y,
                     tiles_x, n_tiles);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
