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
  // where in the tile the u-th float4 of this thread lands never changes: (row << 8 | column) of its pixel, worked out once
  int pp[CV_LD];
#pragma unroll
  for (int u = 0; u < CV_LD; ++u) {
    const int pix = (tid + 256 * u) >> 3, pr = pix / CV_TW;
    pp[u] = (pr << 8) | (pix - pr * CV_TW);
  }
  const int j4 = 4 * (tid & 7);
  // the u-th load of a stage whose tile starts at (rb + 1, cb + 1) from source xs (already offset by j4)
#define CV_ISSUE1(u_, xs_, rb_, cb_)                                                                                    \
  do {                                                                                                                  \
    const int gr = reflect_idx(min((rb_) + (pp[u_] >> 8), H), H), gc = reflect_idx(min((cb_) + (pp[u_] & 255), W), W);  \
    if (tid + 256 * (u_) < CV_TH * CV_TW * 8) pre[u_] = *(const float4 *)((xs_) + ((size_t)gr * W + gc) * 32);          \
  } while (0)
  {
    const int ty_ = t_begin / tiles_x, tx_ = t_begin - ty_ * tiles_x;
#pragma unroll
    for (int u = 0; u < CV_LD; ++u) CV_ISSUE1(u, x0 + j4, ty_ * CV_ROWS - 1, tx_ * CV_COLS - 1);
  }

  const int m = lane & 31, kh = lane >> 5;
  const float o_sc = out_scale ? out_scale[m] : 1.f, o_sh = out_shift ? out_shift[m] : 0.f;
  f32x16 acc[4];
  // a finished tile's outputs wait here (scaled, shifted, clamped) and are stored one value per "unit" of the NEXT stage's
  // MFMA loop: the stores, like the loads of the stage after, are issued while the matrix cores work, and the barrier at
  // the top of a stage (s_waitcnt vmcnt(0): on gfx9 stores count too) finds them long done
  float outv[64];
  bool pend = false;
  float *pend_y = nullptr;                 // &y[row r0 + 2 * wave][column c0 + 4 * kh][channel m] of the waiting tile
  int pend_r = 0, pend_c = 0;              // its first row and column (bounds of the image's last tiles)
#define CV_STORE1(q_)                                                                                                   \
  do {                                                                                                                  \
    const int b_ = (q_) >> 4, i_ = (q_) & 15, dr_ = b_ >> 1, dc_ = 32 * (b_ & 1) + (i_ & 3) + 8 * (i_ >> 2);            \
    if (pend_r + dr_ < H && pend_c + dc_ < W) pend_y[((size_t)dr_ * W + dc_) * 32] = outv[q_];                          \
  } while (0)
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
      const int q = tid + 256 * u, pix = q >> 3;
      if (q < CV_TH * CV_TW * 8) {
        float4 v = pre[u];
        if (aff) {
          const float4 sc = *(const float4 *)(in_scale + j4), sh = *(const float4 *)(in_shift + j4);
          v.x = fmaxf(0.f, v.x * sc.x + sh.x); v.y = fmaxf(0.f, v.y * sc.y + sh.y);
          v.z = fmaxf(0.f, v.z * sc.z + sh.z); v.w = fmaxf(0.f, v.w * sc.w + sh.w);
        }
        float *d = tile + pix * CV_PIX + j4;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      }
    }
    __syncthreads();
    // the next stage: its source and the corner of its tile
    const bool has_next = st + 1 < n_stage;
    const int tn = t_begin + (st + 1) / n_src, tyn = tn / tiles_x, txn = tn - tyn * tiles_x;
    const float *xn = (((st + 1) % n_src) == 0 ? x0 : x1) + j4;
    const int rbn = tyn * CV_ROWS - 1, cbn = txn * CV_COLS - 1;
    // ---- 9 taps x 16 channel pairs: one B fragment, four A fragments, four MFMAs.  A "unit" = one tap, four channels
    // (two channel pairs): 2 B + 8 A values, 8 MFMAs = 512 matrix-core cycles.  The fragments of unit u + 1 are requested
    // BEFORE the MFMAs of unit u are issued (two register sets), so that with one wave per SIMD an LDS read has a whole
    // unit to arrive in.  Between the two halves of a unit's MFMAs goes one load of the next stage (address arithmetic
    // and all), after them one store of the previous tile: vector work that runs while the matrix cores are busy.
    const float *arow = tile + ((2 * wave) * CV_TW + m) * CV_PIX + kh;       // block b: row 2*wave + (b >> 1), columns 32*(b & 1) + m
    const float *bw = wl + src * CV_W_FLOATS + kh * 32 + m;
    float fa[2][4][2], fb[2][2];
#define CV_LOADU(buf_, u_)                                                                                              \
  do {                                                                                                                  \
    const int tap_ = (u_) / 8, cq_ = (u_) % 8, dy_ = tap_ / 3, dx_ = tap_ % 3;                                          \
    _Pragma("unroll") for (int e = 0; e < 2; ++e) fb[buf_][e] = bw[(tap_ * 32 + 4 * cq_ + 2 * e) * 32];                 \
    _Pragma("unroll") for (int b = 0; b < 4; ++b)                                                                       \
      _Pragma("unroll") for (int e = 0; e < 2; ++e)                                                                     \
        fa[buf_][b][e] = arow[(((b >> 1) + dy_) * CV_TW + 32 * (b & 1) + dx_) * CV_PIX + 4 * cq_ + 2 * e];              \
  } while (0)
    CV_LOADU(0, 0);
#pragma unroll
    for (int u = 0; u < 72; ++u) {
      if (u + 1 < 72) CV_LOADU((u + 1) & 1, u + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[u & 1][b][0], fb[u & 1][0], acc[b], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (u % 3 == 0 && u / 3 < CV_LD && has_next) CV_ISSUE1(u / 3, xn, rbn, cbn);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[u & 1][b][1], fb[u & 1][1], acc[b], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (u < 64 && pend) CV_STORE1(u);
      __builtin_amdgcn_sched_barrier(0);
    }
#undef CV_LOADU
    pend = false;
    if (src + 1 < n_src) continue;
    // ---- epilogue: C/D layout col = lane & 31 (output channel), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (pixel)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float v = acc[b][i] * o_sc + o_sh;
        outv[16 * b + i] = relu ? fmaxf(0.f, v) : v;
      }
    pend = true;
    pend_r = r0 + 2 * wave; pend_c = c0 + 4 * kh;
    pend_y = y + ((size_t)pend_r * W + pend_c) * 32 + m;
  }
  if (pend) {
#pragma unroll
    for (int q = 0; q < 64; ++q) CV_STORE1(q);
  }
#undef CV_STORE1
#undef CV_ISSUE1
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


// ---- ShapeNet's three 1x1 heads + bias + softmax in one pass (model_parts/shape_net.py:12-46: three Conv2d(32, 32, 1x1);
// the softmax of shape_net_model.py's inference) ----------------------------------------------------------------------
// h [ldh][ldw][32] float32 (the backbone's last activation, channels-last) -> marks[k] [H][W][32] for k = size, ratio, angle.
// Separately (library 1x1 convolution, its bias add, the softmax epilogue) the three heads move 38 GB on a 4096 x 4096
// image; fused they read h once and write the three mark maps: 8.6 GB.
// A wave takes 32 consecutive pixels of an image row: their 32 x 32 activations go through LDS (pixel stride 33, as above)
// into B fragments; A fragments are the head's weights (kept in registers); D[class][pixel] puts 16 classes of one pixel
// in a lane and the other 16 in lane ^ 32, so the softmax is 16 in-lane steps and one exchange.
#define HD_PIX 33
__global__ __launch_bounds__(256) void k_shapenet_heads(const float *__restrict__ h, int H, int W, int ldw, const float *__restrict__ wh,
                                                        const float *__restrict__ bh, float *__restrict__ m0, float *__restrict__ m1,
                                                        float *__restrict__ m2, int groups_per_row, int n_groups) {
  __shared__ float tiles[4][32 * HD_PIX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, kh = lane >> 5;
  float *tile = tiles[wave];
  // weights of the three heads as A fragments: step s covers channels 2s, 2s + 1; this lane: class n, channel 2s + kh
  float wa[3][16], bias[3][16];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
#pragma unroll
    for (int s = 0; s < 16; ++s) wa[k][s] = wh[(k * 32 + n) * 32 + 2 * s + kh];
#pragma unroll
    for (int r = 0; r < 16; ++r) bias[k][r] = bh[k * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh];
  }
  const int n_waves = (int)gridDim.x * 4;
  for (int g = (int)blockIdx.x * 4 + wave; g < n_groups; g += n_waves) {
    const int gx = g / groups_per_row, gy0 = (g - gx * groups_per_row) * 32;
    const int nv = min(32, W - gy0);
    const float *src = h + ((size_t)gx * ldw + gy0) * 32;
    float4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = lane + 64 * i, p = idx >> 3;
      v[i] = p < nv ? *(const float4 *)(src + (size_t)idx * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = lane + 64 * i;
      float *d = tile + (idx >> 3) * HD_PIX + 4 * (idx & 7);
      d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float bf[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) bf[s] = tile[n * HD_PIX + 2 * s + kh];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                 // (the next group's stores to the tile come after these reads)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = bias[k][r];
#pragma unroll
      for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[k][s], bf[s], acc, 0, 0, 0);
      float mx = acc[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, acc[r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float e[16], sum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { e[r] = expf(acc[r] - mx); sum += e[r]; }
      sum += __shfl_xor(sum, 32, 64);
      if (n < nv) {
        float *dst = (k == 0 ? m0 : (k == 1 ? m1 : m2)) + ((size_t)gx * W + gy0 + n) * 32 + 4 * kh;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *(float4 *)(dst + 8 * q) = make_float4(e[4 * q] / sum, e[4 * q + 1] / sum, e[4 * q + 2] / sum, e[4 * q + 3] / sum);
      }
    }
  }
}

extern "C" int mpp_launch_shapenet_heads(hipStream_t st, const float *h, int H, int W, int ldw, const float *wh, const float *bh,
                                         float *m0, float *m1, float *m2) {
  if (H < 1 || W < 1 || ldw < W) return -1;
  if (((uintptr_t)h & 15) || ((uintptr_t)m0 & 15) || ((uintptr_t)m1 & 15) || ((uintptr_t)m2 & 15)) return -2;
  const int gpr = (W + 31) / 32;
  const long long n_groups = (long long)gpr * H;
  if (n_groups > 0x7fffffffLL) return -1;
  int grid = (int)((n_groups + 3) / 4);
  if (grid > 2048) grid = 2048;                       // eight workgroups per CU, each wave strides over the groups
  hipLaunchKernelGGL(k_shapenet_heads, dim3(grid), dim3(256), 0, st, h, H, W, ldw, wh, bh, m0, m1, m2, gpr, (int)n_groups);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}


// ---- the stem: Conv2d(3 -> 32, 3x3, reflect) + folded BatchNorm + ReLU of the first DoubleConv (unet_parts.py:12-31) ---------
// x [H][W][3] float32 -> y [H][W][32].  29 GFLOP on 4096 x 4096 against 2.3 GB of traffic: HBM-bound, so plain FMAs: a
// workgroup stages an 18 x 18 x 3 input tile in LDS, a thread computes the 32 channels of one pixel (weights are
// wave-uniform: scalar loads), applies scale / shift / ReLU and writes its 128 bytes.  The library needed a padded copy of
// the picture, a zero fill of the output and a kernel at 0.7 ms for this; the BatchNorm + ReLU then cost the next
// convolution's load phase.
#define ST_T 16
__global__ __launch_bounds__(256) void k_conv3x3_stem(const float *__restrict__ x, int H, int W, const float *__restrict__ wp,
                                                      const float *__restrict__ scale, const float *__restrict__ shift,
                                                      float *__restrict__ y) {
  __shared__ float tile[(ST_T + 2) * (ST_T + 2) * 3];
  const int tid = threadIdx.x, r0 = blockIdx.y * ST_T, c0 = blockIdx.x * ST_T;
  for (int q = tid; q < (ST_T + 2) * (ST_T + 2); q += 256) {
    const int pr = q / (ST_T + 2), pc = q - pr * (ST_T + 2);
    const int gr = reflect_idx(min(r0 - 1 + pr, H), H), gc = reflect_idx(min(c0 - 1 + pc, W), W);
    const float *px = x + ((size_t)gr * W + gc) * 3;
    tile[3 * q] = px[0]; tile[3 * q + 1] = px[1]; tile[3 * q + 2] = px[2];
  }
  __syncthreads();
  const int lr = tid >> 4, lc = tid & 15, row = r0 + lr, col = c0 + lc;
  float acc[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) acc[c] = 0.f;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const float *t = tile + ((lr + tap / 3) * (ST_T + 2) + lc + tap % 3) * 3;
#pragma unroll
    for (int ci = 0; ci < 3; ++ci) {
      const float xv = t[ci];
      const float *w = wp + (tap * 3 + ci) * 32;                 // (wave-uniform address: scalar loads)
#pragma unroll
      for (int c = 0; c < 32; ++c) acc[c] = fmaf(xv, w[c], acc[c]);
    }
  }
  if (row < H && col < W) {
    float4 *dst = (float4 *)(y + ((size_t)row * W + col) * 32);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = fmaxf(0.f, acc[4 * q + k] * scale[4 * q + k] + shift[4 * q + k]);
      dst[q] = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}
extern "C" int mpp_launch_conv3x3_stem(hipStream_t st, const float *x, int H, int W, const float *wp, const float *scale,
                                       const float *shift, float *y) {
  if (H < 2 || W < 2) return -1;
  if ((uintptr_t)y & 15) return -2;
  hipLaunchKernelGGL(k_conv3x3_stem, dim3((W + ST_T - 1) / ST_T, (H + ST_T - 1) / ST_T), dim3(256), 0, st, x, H, W, wp, scale, shift, y);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
