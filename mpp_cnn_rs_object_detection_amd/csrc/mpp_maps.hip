// mpp_maps.hip -- per-tile map preparation and the score-map epilogues of the two U-Nets.
//
//  * birth CDF tables for the data-driven birth kernel (replaces the O(H*W) cumsum the reference
//    redoes on every proposal, utils/sampler2d.py:43);
//  * naive_detection (sample_rjmcmc.py:23-35 + utils/nms.py:68-110) as one workgroup per tile;
//  * PosNet epilogue: sigmoid(mask) , divergence of the vector field, 1x1 "div_clf" conv, sigmoid
//    (pos_net_model.py:186-200, :338-346; torch_div.py:8-43), fused, one read of 3 ch, one write;
//  * ShapeNet epilogue: softmax over 32 classes + CHW -> HWC transpose through LDS so that both
//    the read (64 consecutive pixels of one channel) and the write (64 pixels x 32 classes = 8 KiB
//    contiguous) are fully coalesced (shape_net_model.py:139-141 + data_loaders.py:54).
#include "mpp_device.hpp"

// ---- birth CDF ---------------------------------------------------------------------------------
// (all tiles of a context in one launch: blockIdx.y / blockIdx.x / blockIdx.z = tile)
__global__ void k_row_partial(const float *det_all, int H, int W, double *rowpart_all, double *rowtot_all) {
  const size_t hw = (size_t)H * W;
  const float *det = det_all + blockIdx.y * hw;
  double *rowpart = rowpart_all + blockIdx.y * hw, *rowtot = rowtot_all + (size_t)blockIdx.y * H;
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= H) return;
  double s = 0.0;
  for (int j = 0; j < W; ++j) { s += (double)det[(size_t)r * W + j]; rowpart[(size_t)r * W + j] = s; }
  rowtot[r] = s;
}
__global__ void k_row_base(const double *rowtot_all, int H, double *rowbase_all) {
  const double *rowtot = rowtot_all + (size_t)blockIdx.x * H;
  double *rowbase = rowbase_all + (size_t)blockIdx.x * (H + 1);
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int r = 0; r < H; ++r) { rowbase[r] = s; s += rowtot[r]; }
    rowbase[H] = s;
  }
}
// sum of det over the (2*md+1)^2 window clipped to the tile, around every pixel: the normaliser of the
// data-driven translation kernel (transform_kernels.py:70-76,94-99).  Rows are added top to bottom,
// each row segment taken from the per-row prefix table -- the same order the sampler's draw uses.
__global__ void k_boxsum(const double *rowpart_all, int H, int W, int md, double *boxsum_all) {
  const size_t hw = (size_t)H * W;
  const double *rowpart = rowpart_all + blockIdx.z * hw;
  double *boxsum = boxsum_all + blockIdx.z * hw;
  int y = blockIdx.x * blockDim.x + threadIdx.x, x = blockIdx.y;
  if (y >= W || x >= H) return;
  int x0 = max(0, x - md), x1 = min(x + md + 1, H), y0 = max(0, y - md), y1 = min(y + md + 1, W);
  double s = 0.0;
  for (int r = x0; r < x1; ++r) {
    const double *rp = rowpart + (size_t)r * W;
    s += rp[y1 - 1] - (y0 > 0 ? rp[y0 - 1] : 0.0);
  }
  boxsum[(size_t)x * W + y] = s;
}
extern "C" void mpp_launch_boxsum(hipStream_t st, int n_tiles, const double *rowpart, int H, int W, int md, double *boxsum) {
  for (int t0 = 0; t0 < n_tiles; t0 += 65535) {            // gridDim.z limit
    const int nt = n_tiles - t0 < 65535 ? n_tiles - t0 : 65535;
    const size_t off = (size_t)t0 * H * W;
    hipLaunchKernelGGL(k_boxsum, dim3((W + 63) / 64, H, nt), dim3(64), 0, st, rowpart + off, H, W, md, boxsum + off);
  }
}
extern "C" void mpp_launch_cdf(hipStream_t st, int n_tiles, const float *det, int H, int W, double *rowpart, double *rowbase,
                               double *scratch_rowtot) {
  for (int t0 = 0; t0 < n_tiles; t0 += 65535) {            // gridDim.y limit
    const int nt = n_tiles - t0 < 65535 ? n_tiles - t0 : 65535;
    const size_t off = (size_t)t0 * H * W;
    hipLaunchKernelGGL(k_row_partial, dim3((H + 63) / 64, nt), dim3(64), 0, st, det + off, H, W, rowpart + off,
                       scratch_rowtot + (size_t)t0 * H);
  }
  hipLaunchKernelGGL(k_row_base, dim3(n_tiles), dim3(64), 0, st, scratch_rowtot, H, rowbase);
}

// ---- naive detection ------------------------------------------------------------------------------
// keys: (float bits << 32) | flat index ; det >= 0 so the bit pattern orders like the value; ties go to
// the larger index (the convention of the CPU oracle)
__global__ __launch_bounds__(256) void k_naive_init(const DevParams *P, const TileRef *tiles, double threshold,
                                                    double nms_dist, unsigned long long *cand_all, int cand_cap) {
  __shared__ unsigned long long best[256];
  __shared__ int s_count;
  TileRef t = tiles[blockIdx.x];
  unsigned long long *cand = cand_all + (size_t)blockIdx.x * cand_cap;
  const int hw = P->H * P->W;
  if (threadIdx.x == 0) s_count = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < hw; i += blockDim.x) {
    float v = t.det[i];
    if ((double)v >= threshold) {
      int k = atomicAdd(&s_count, 1);
      if (k < cand_cap) cand[k] = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned int)i;
    }
  }
  __syncthreads();
  int nc = min(s_count, cand_cap);
  int n_out = 0, err = s_count > cand_cap ? 2 : 0;
  while (true) {
    unsigned long long m = 0;
    for (int i = threadIdx.x; i < nc; i += blockDim.x) m = max(m, cand[i]);
    best[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) best[threadIdx.x] = max(best[threadIdx.x], best[threadIdx.x + s]);
      __syncthreads();
    }
    unsigned long long b = best[0];
    __syncthreads();
    if (b == 0) break;
    int idx = (int)(b & 0xffffffffu), bx = idx / P->W, by = idx % P->W;
    if (n_out < P->cap) {
      if (threadIdx.x < 3) {
        int k = threadIdx.x;
        const float *row = mark_row(P, t, k, bx, by);
        int am = 0;
        for (int i = 1; i < MPP_NCLASS; ++i) if (row[i] > row[am]) am = i;
        double v = P->maps.edges[k][am];
        if (k == 0) t.ps[n_out] = v; else if (k == 1) t.pr[n_out] = v; else t.pa[n_out] = v;
      }
      if (threadIdx.x == 3) { t.px[n_out] = bx; t.py[n_out] = by; }
    } else err = 2;
    ++n_out;
    for (int i = threadIdx.x; i < nc; i += blockDim.x) {
      unsigned long long c = cand[i];
      if (c == 0) continue;
      int ci = (int)(c & 0xffffffffu);
      double dx = (double)(ci / P->W - bx), dy = (double)(ci % P->W - by);
      if (!(sqrt(dx * dx + dy * dy) > nms_dist)) cand[i] = 0;   // utils/nms.py:103-105 keeps only d > threshold
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { *t.n = min(n_out, P->cap); if (err) *t.err = err; }
}
extern "C" void mpp_launch_naive_init(hipStream_t st, const DevParams *P, const TileRef *tiles, int n_tiles,
                                      double threshold, double nms_dist, unsigned long long *cand, int cand_cap) {
  hipLaunchKernelGGL(k_naive_init, dim3(n_tiles), dim3(256), 0, st, P, tiles, threshold, nms_dist, cand, cand_cap);
}

// ---- PosNet epilogue --------------------------------------------------------------------------------
// out: [3][ldh][ldw] float32 (vec0 = d/d row component, vec1 = d/d col component, mask logit);
// det[x][y] = sigmoid(w * (d vec0/dx + d vec1/dy) * sigmoid(mask) + b), central differences inside,
// one-sided at the borders of the H x W region (torch.gradient semantics).
__device__ __forceinline__ float posnet_pixel(const float *v0, const float *v1, const float *mk, int x, int y, int H,
                                              int W, int ldw, float w, float b) {
  float g0, g1;
  if (H == 1) g0 = 0.f;
  else if (x == 0) g0 = v0[(size_t)1 * ldw + y] - v0[y];
  else if (x == H - 1) g0 = v0[(size_t)x * ldw + y] - v0[(size_t)(x - 1) * ldw + y];
  else g0 = (v0[(size_t)(x + 1) * ldw + y] - v0[(size_t)(x - 1) * ldw + y]) / 2.0f;
  if (W == 1) g1 = 0.f;
  else if (y == 0) g1 = v1[(size_t)x * ldw + 1] - v1[(size_t)x * ldw];
  else if (y == W - 1) g1 = v1[(size_t)x * ldw + y] - v1[(size_t)x * ldw + y - 1];
  else g1 = (v1[(size_t)x * ldw + y + 1] - v1[(size_t)x * ldw + y - 1]) / 2.0f;
  float mask = 1.0f / (1.0f + expf(-mk[(size_t)x * ldw + y]));
  float score = w * ((g0 + g1) * mask) + b;
  return 1.0f / (1.0f + expf(-score));
}
// each thread produces 4 consecutive pixels of a row: 16-byte loads of the rows above/below, of the row
// itself (plus its two neighbours) and of the mask, one 16-byte store
__global__ __launch_bounds__(256) void k_posnet_epilogue(const float *out, int H, int W, int ldh, int ldw, float w,
                                                         float b, float *det, int vec_ok) {
  const int y4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4, x = blockIdx.y;
  if (y4 >= W || x >= H) return;
  const size_t plane = (size_t)ldh * ldw;
  const float *v0 = out, *v1 = out + plane, *mk = out + 2 * plane;
  if (vec_ok && x > 0 && x < H - 1 && y4 > 0 && y4 + 4 < W) {
    const float4 up = *(const float4 *)(v0 + (size_t)(x - 1) * ldw + y4), dn = *(const float4 *)(v0 + (size_t)(x + 1) * ldw + y4);
    const float4 c = *(const float4 *)(v1 + (size_t)x * ldw + y4), m4 = *(const float4 *)(mk + (size_t)x * ldw + y4);
    const float left = v1[(size_t)x * ldw + y4 - 1], right = v1[(size_t)x * ldw + y4 + 4];
    const float g0[4] = {(dn.x - up.x) / 2.0f, (dn.y - up.y) / 2.0f, (dn.z - up.z) / 2.0f, (dn.w - up.w) / 2.0f};
    const float g1[4] = {(c.y - left) / 2.0f, (c.z - c.x) / 2.0f, (c.w - c.y) / 2.0f, (right - c.z) / 2.0f};
    const float mm[4] = {m4.x, m4.y, m4.z, m4.w};
    float r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float mask = 1.0f / (1.0f + expf(-mm[i]));
      float score = w * ((g0[i] + g1[i]) * mask) + b;
      r[i] = 1.0f / (1.0f + expf(-score));
    }
    *(float4 *)(det + (size_t)x * W + y4) = make_float4(r[0], r[1], r[2], r[3]);
  } else {
    for (int y = y4; y < min(y4 + 4, W); ++y) det[(size_t)x * W + y] = posnet_pixel(v0, v1, mk, x, y, H, W, ldw, w, b);
  }
}
extern "C" void mpp_launch_posnet_epilogue(hipStream_t st, const float *out, int H, int W, int ldh, int ldw, float w,
                                           float b, float *det) {
  int vec_ok = (ldw % 4 == 0) && (W % 4 == 0) && (((uintptr_t)out & 15) == 0) && (((uintptr_t)det & 15) == 0) &&
               (((size_t)ldh * ldw) % 4 == 0);
  hipLaunchKernelGGL(k_posnet_epilogue, dim3((W + 1023) / 1024, H), dim3(256), 0, st, out, H, W, ldh, ldw, w, b, det,
                     vec_ok);
}

// ---- ShapeNet epilogue --------------------------------------------------------------------------------
// logits: [32][ldh][ldw] -> marks [H][W][32] = softmax over classes.  One block = 64 pixels of a row:
// 16-byte loads (4 pixels of one channel per lane), transpose through LDS, 2 x 16-byte stores per lane
// (one pixel's 32 classes = 128 contiguous bytes from 4 lanes).
__global__ __launch_bounds__(256) void k_shapenet_epilogue(const float *logits, int H, int W, int ldh, int ldw,
                                                           float *marks, int vec_ok) {
  __shared__ float tile[MPP_NCLASS][WAVE + 1];
  const int x = blockIdx.y, y0 = blockIdx.x * WAVE;
  const size_t plane = (size_t)ldh * ldw;
  if (vec_ok && y0 + WAVE <= W) {
    const int px4 = (threadIdx.x & 15) * 4, ch0 = threadIdx.x >> 4;      // 16 lanes cover the 64 pixels of a channel
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int ch = ch0 + 16 * h;
      const float4 v = *(const float4 *)(logits + ch * plane + (size_t)x * ldw + y0 + px4);
      tile[ch][px4] = v.x; tile[ch][px4 + 1] = v.y; tile[ch][px4 + 2] = v.z; tile[ch][px4 + 3] = v.w;
    }
  } else {
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    for (int ch = grp; ch < MPP_NCLASS; ch += 4) {
      int y = y0 + lane;
      tile[ch][lane] = y < W ? logits[ch * plane + (size_t)x * ldw + y] : 0.f;
    }
  }
  __syncthreads();
  // pixel p = threadIdx.x / 4 handles 8 classes: threads of one pixel sit in one wave -> shuffles
  const int p = threadIdx.x >> 2, q = threadIdx.x & 3;
  float v[8], m = -INFINITY;
#pragma unroll
  for (int i = 0; i < 8; ++i) { v[i] = tile[q * 8 + i][p]; m = fmaxf(m, v[i]); }
  m = fmaxf(m, __shfl_xor(m, 1, WAVE)); m = fmaxf(m, __shfl_xor(m, 2, WAVE));
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) { v[i] = expf(v[i] - m); s += v[i]; }
  s += __shfl_xor(s, 1, WAVE); s += __shfl_xor(s, 2, WAVE);
  if (y0 + p < W) {
    float4 *dst = (float4 *)(marks + ((size_t)x * W + y0 + p) * MPP_NCLASS + q * 8);
    dst[0] = make_float4(v[0] / s, v[1] / s, v[2] / s, v[3] / s);
    dst[1] = make_float4(v[4] / s, v[5] / s, v[6] / s, v[7] / s);
  }
}
extern "C" void mpp_launch_shapenet_epilogue(hipStream_t st, const float *logits, int H, int W, int ldh, int ldw,
                                             float *marks) {
  int vec_ok = (ldw % 4 == 0) && (((uintptr_t)logits & 15) == 0) && (((size_t)ldh * ldw) % 4 == 0);
  hipLaunchKernelGGL(k_shapenet_epilogue, dim3((W + WAVE - 1) / WAVE, H), dim3(256), 0, st, logits, H, W, ldh, ldw,
                     marks, vec_ok);
}

// ---- conv epilogue of the U-Nets' DoubleConv blocks ----------------------------------------------------------
// y = max(0, x * scale[c] + shift[c]) in place on a [planes][hw] tensor (plane p belongs to channel p % C):
// BatchNorm(eval) folded with the convolution bias, plus the ReLU, in ONE pass over the activation instead of the
// three (bias add, batch norm, ReLU) PyTorch runs after each MIOpen convolution (unet_parts.py:12-31).
// 16-byte accesses: 4 floats or 8 bf16 per lane (hw is a multiple of 8: the image is padded to 2^depth).
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float((unsigned int)h << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {          // round to nearest even, as torch does
  unsigned int u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__global__ __launch_bounds__(256) void k_affine_relu_f32(float *x, int C, size_t hw, size_t total4, const float *scale,
                                                         const float *shift) {
  const size_t i4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i4 >= total4) return;
  const int ch = (int)((i4 * 4 / hw) % (size_t)C);
  const float s = scale[ch], t = shift[ch];
  float4 v = ((float4 *)x)[i4];
  v.x = fmaxf(0.f, v.x * s + t); v.y = fmaxf(0.f, v.y * s + t); v.z = fmaxf(0.f, v.z * s + t); v.w = fmaxf(0.f, v.w * s + t);
  ((float4 *)x)[i4] = v;
}
__global__ __launch_bounds__(256) void k_affine_relu_bf16(unsigned short *x, int C, size_t hw, size_t total8, const float *scale,
                                                          const float *shift) {
  const size_t i8 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i8 >= total8) return;
  const int ch = (int)((i8 * 8 / hw) % (size_t)C);
  const float s = scale[ch], t = shift[ch];
  uint4 v = ((uint4 *)x)[i8];
  unsigned int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float lo = fmaxf(0.f, bf16_to_f32((unsigned short)(w[k] & 0xffffu)) * s + t);
    float hi = fmaxf(0.f, bf16_to_f32((unsigned short)(w[k] >> 16)) * s + t);
    w[k] = (unsigned int)f32_to_bf16(lo) | ((unsigned int)f32_to_bf16(hi) << 16);
  }
  ((uint4 *)x)[i8] = make_uint4(w[0], w[1], w[2], w[3]);
}
// planes whose size is not a multiple of the vector width (the deepest levels of small images): one element per lane
__global__ __launch_bounds__(256) void k_affine_relu_any(void *x, int C, size_t hw, size_t total, int elem_bytes,
                                                         const float *scale, const float *shift) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int ch = (int)((i / hw) % (size_t)C);
  if (elem_bytes == 4) {
    float *p = (float *)x;
    p[i] = fmaxf(0.f, p[i] * scale[ch] + shift[ch]);
  } else {
    unsigned short *p = (unsigned short *)x;
    p[i] = f32_to_bf16(fmaxf(0.f, bf16_to_f32(p[i]) * scale[ch] + shift[ch]));
  }
}
extern "C" int mpp_launch_affine_relu(hipStream_t st, void *x, int planes, int C, size_t hw, int elem_bytes, const float *scale,
                                      const float *shift) {
  const size_t total = (size_t)planes * hw;
  if (elem_bytes != 4 && elem_bytes != 2) return -1;
  if (hw % (16 / elem_bytes) || ((uintptr_t)x & 15)) {
    hipLaunchKernelGGL(k_affine_relu_any, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, C, hw, total, elem_bytes,
                       scale, shift);
    return 0;
  }
  if (elem_bytes == 4) {
    hipLaunchKernelGGL(k_affine_relu_f32, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, st, (float *)x, C, hw,
                       total / 4, scale, shift);
  } else {
    hipLaunchKernelGGL(k_affine_relu_bf16, dim3((unsigned)((total / 8 + 255) / 256)), dim3(256), 0, st, (unsigned short *)x, C,
                       hw, total / 8, scale, shift);
  }
  return 0;
}

// ---- channels-last (NHWC) glue of the U-Nets: everything between two convolutions in ONE pass ------------------------
// y[H+2p][W+2p][C0+C1] <- reflect-pad_p( f( pool( cat(x0, x1) ) ) ),  f(v) = max(0, v * scale[c] + shift[c]) (or the
// identity when scale is null), pool = 2x2 max-pool of the [2H][2W] sources (or none), cat along the channels.
// Stands for `F.pad(mode="reflect")` inside Conv2d(padding_mode="reflect") together with the BatchNorm+ReLU before it
// (unet_parts.py:12-31), MaxPool2d(2) (`:34-45`) and torch.cat([skip, up]) (`:48-67`).  In NHWC one pixel's channels are
// contiguous, so every lane moves 16 bytes (4 floats / 8 bf16) and both streams are fully coalesced.  HBM-bound.
template <int EB> struct NhwcVec;
template <> struct NhwcVec<4> {
  static constexpr int N = 4;
  __device__ static void load(const void *p, size_t i, float *v) { float4 t = *(const float4 *)((const float *)p + i); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
  __device__ static void store(void *p, size_t i, const float *v) { *(float4 *)((float *)p + i) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct NhwcVec<2> {
  static constexpr int N = 8;
  __device__ static void load(const void *p, size_t i, float *v) {
    uint4 t = *(const uint4 *)((const unsigned short *)p + i);
    unsigned int w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[2 * k] = bf16_to_f32((unsigned short)(w[k] & 0xffffu)); v[2 * k + 1] = bf16_to_f32((unsigned short)(w[k] >> 16)); }
  }
  __device__ static void store(void *p, size_t i, const float *v) {
    unsigned int w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] = (unsigned int)f32_to_bf16(v[2 * k]) | ((unsigned int)f32_to_bf16(v[2 * k + 1]) << 16);
    *(uint4 *)((unsigned short *)p + i) = make_uint4(w[0], w[1], w[2], w[3]);
  }
};
__device__ __forceinline__ int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

template <int EB>
__global__ __launch_bounds__(256) void k_nhwc_glue_vec(const void *x0, const void *x1, void *y, int H, int W, int C0, int C1,
                                                        int pad, int pool, const float *scale, const float *shift, size_t total) {
  constexpr int N = NhwcVec<EB>::N;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int C = C0 + C1, cv = C / N, Wo = W + 2 * pad;
  const int c = (int)(i % (size_t)cv) * N;
  const size_t pix = i / (size_t)cv;
  const int wo = (int)(pix % (size_t)Wo), ho = (int)(pix / (size_t)Wo);
  const int h = reflect1(ho - pad, H), w = reflect1(wo - pad, W);
  const void *src = c < C0 ? x0 : x1;
  const int Cs = c < C0 ? C0 : C1, cs = c < C0 ? c : c - C0;
  float s[N], t[N], v[N];
  if (scale) {
#pragma unroll
    for (int k = 0; k < N; ++k) { s[k] = scale[c + k]; t[k] = shift[c + k]; }
  }
  if (pool) {
    const int Ws = 2 * W;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float u[N];
      NhwcVec<EB>::load(src, ((size_t)(2 * h + (q >> 1)) * Ws + (2 * w + (q & 1))) * Cs + cs, u);
#pragma unroll
      for (int k = 0; k < N; ++k) {
        float a = scale ? fmaxf(0.f, u[k] * s[k] + t[k]) : u[k];
        v[k] = q == 0 ? a : fmaxf(v[k], a);
      }
    }
  } else {
    NhwcVec<EB>::load(src, ((size_t)h * W + w) * Cs + cs, v);
    if (scale) {
#pragma unroll
      for (int k = 0; k < N; ++k) v[k] = fmaxf(0.f, v[k] * s[k] + t[k]);
    }
  }
  NhwcVec<EB>::store(y, i * N, v);
}
// any channel count / mixed element types (the 3-channel stem: float32 image in, bfloat16 or float32 out): one element per lane
__global__ __launch_bounds__(256) void k_nhwc_glue_any(const void *x0, const void *x1, void *y, int H, int W, int C0, int C1,
                                                        int pad, int pool, int in_bytes, int out_bytes, const float *scale,
                                                        const float *shift, size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int C = C0 + C1, Wo = W + 2 * pad;
  const int c = (int)(i % (size_t)C);
  const size_t pix = i / (size_t)C;
  const int wo = (int)(pix % (size_t)Wo), ho = (int)(pix / (size_t)Wo);
  const int h = reflect1(ho - pad, H), w = reflect1(wo - pad, W);
  const void *src = c < C0 ? x0 : x1;
  const int Cs = c < C0 ? C0 : C1, cs = c < C0 ? c : c - C0;
  auto rd = [&](size_t j) -> float { return in_bytes == 4 ? ((const float *)src)[j] : bf16_to_f32(((const unsigned short *)src)[j]); };
  float v = 0.f;
  if (pool) {
    const int Ws = 2 * W;
    for (int q = 0; q < 4; ++q) {
      float a = rd(((size_t)(2 * h + (q >> 1)) * Ws + (2 * w + (q & 1))) * Cs + cs);
      if (scale) a = fmaxf(0.f, a * scale[c] + shift[c]);
      v = q == 0 ? a : fmaxf(v, a);
    }
  } else {
    v = rd(((size_t)h * W + w) * Cs + cs);
    if (scale) v = fmaxf(0.f, v * scale[c] + shift[c]);
  }
  if (out_bytes == 4) ((float *)y)[i] = v; else ((unsigned short *)y)[i] = f32_to_bf16(v);
}
extern "C" int mpp_launch_nhwc_glue(hipStream_t st, const void *x0, const void *x1, void *y, int H, int W, int C0, int C1, int pad,
                                    int pool, int in_bytes, int out_bytes, const float *scale, const float *shift) {
  if ((in_bytes != 4 && in_bytes != 2) || (out_bytes != 4 && out_bytes != 2)) return -1;
  const int C = C0 + C1;
  const size_t elems = (size_t)(H + 2 * pad) * (W + 2 * pad) * C;
  const int n = 16 / in_bytes;
  const bool aligned = (((uintptr_t)x0 | (uintptr_t)x1 | (uintptr_t)y) & 15) == 0;
  if (in_bytes == out_bytes && aligned && C0 % n == 0 && C1 % n == 0) {
    const size_t total = elems / n;
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (in_bytes == 4) hipLaunchKernelGGL(k_nhwc_glue_vec<4>, dim3(grid), dim3(256), 0, st, x0, x1, y, H, W, C0, C1, pad, pool, scale, shift, total);
    else hipLaunchKernelGGL(k_nhwc_glue_vec<2>, dim3(grid), dim3(256), 0, st, x0, x1, y, H, W, C0, C1, pad, pool, scale, shift, total);
  } else {
    hipLaunchKernelGGL(k_nhwc_glue_any, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, st, x0, x1, y, H, W, C0, C1, pad, pool,
                       in_bytes, out_bytes, scale, shift, elems);
  }
  return 0;
}

// ---- the two epilogues on channels-last network outputs (float32 or bfloat16) ----------------------------------------
// Same arithmetic, in the same order, as k_posnet_epilogue / k_shapenet_epilogue; only the addressing differs:
// out [ldh][ldw][3] and logits [ldh][ldw][32].  A pixel's 32 logits are contiguous, so the softmax needs no transpose:
// 4 lanes per pixel, 8 classes each (one or two 16-byte loads), two 16-byte stores.
template <int EB> __device__ __forceinline__ float ld_elem(const void *p, size_t i) {
  return EB == 4 ? ((const float *)p)[i] : bf16_to_f32(((const unsigned short *)p)[i]);
}
template <int EB>
__global__ __launch_bounds__(256) void k_posnet_epilogue_nhwc(const void *out, int H, int W, int ldw, float w, float b, float *det) {
  const int y = blockIdx.x * blockDim.x + threadIdx.x, x = blockIdx.y;
  if (y >= W || x >= H) return;
  auto at = [&](int i, int j, int ch) -> float { return ld_elem<EB>(out, ((size_t)i * ldw + j) * 3 + ch); };
  float g0, g1;
  if (H == 1) g0 = 0.f;
  else if (x == 0) g0 = at(1, y, 0) - at(0, y, 0);
  else if (x == H - 1) g0 = at(x, y, 0) - at(x - 1, y, 0);
  else g0 = (at(x + 1, y, 0) - at(x - 1, y, 0)) / 2.0f;
  if (W == 1) g1 = 0.f;
  else if (y == 0) g1 = at(x, 1, 1) - at(x, 0, 1);
  else if (y == W - 1) g1 = at(x, y, 1) - at(x, y - 1, 1);
  else g1 = (at(x, y + 1, 1) - at(x, y - 1, 1)) / 2.0f;
  float mask = 1.0f / (1.0f + expf(-at(x, y, 2)));
  float score = w * ((g0 + g1) * mask) + b;
  det[(size_t)x * W + y] = 1.0f / (1.0f + expf(-score));
}
template <int EB>
__global__ __launch_bounds__(256) void k_shapenet_epilogue_nhwc(const void *logits, int H, int W, int ldw, float *marks) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t pix = t >> 2;
  const int q = (int)(t & 3);
  const bool live = pix < (size_t)H * W;           // the 4 lanes of a pixel are live together; dead lanes still shuffle
  const int x = live ? (int)(pix / (size_t)W) : 0, y = live ? (int)(pix % (size_t)W) : 0;
  const size_t src = ((size_t)x * ldw + y) * MPP_NCLASS + q * 8;
  float v[8], m = -INFINITY;
  if (EB == 4) {
    const float4 a = *(const float4 *)((const float *)logits + src), c = *(const float4 *)((const float *)logits + src + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
  } else {
    const uint4 a = *(const uint4 *)((const unsigned short *)logits + src);
    const unsigned int wd[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[2 * k] = bf16_to_f32((unsigned short)(wd[k] & 0xffffu)); v[2 * k + 1] = bf16_to_f32((unsigned short)(wd[k] >> 16)); }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) m = fmaxf(m, v[i]);
  m = fmaxf(m, __shfl_xor(m, 1, WAVE)); m = fmaxf(m, __shfl_xor(m, 2, WAVE));
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) { v[i] = expf(v[i] - m); s += v[i]; }
  s += __shfl_xor(s, 1, WAVE); s += __shfl_xor(s, 2, WAVE);
  if (live) {
    float4 *dst = (float4 *)(marks + pix * MPP_NCLASS + q * 8);
    dst[0] = make_float4(v[0] / s, v[1] / s, v[2] / s, v[3] / s);
    dst[1] = make_float4(v[4] / s, v[5] / s, v[6] / s, v[7] / s);
  }
}
extern "C" int mpp_launch_posnet_epilogue_nhwc(hipStream_t st, const void *out, int elem_bytes, int H, int W, int ldw, float w,
                                               float b, float *det) {
  const dim3 grid((W + 255) / 256, H);
  if (elem_bytes == 4) hipLaunchKernelGGL(k_posnet_epilogue_nhwc<4>, grid, dim3(256), 0, st, out, H, W, ldw, w, b, det);
  else if (elem_bytes == 2) hipLaunchKernelGGL(k_posnet_epilogue_nhwc<2>, grid, dim3(256), 0, st, out, H, W, ldw, w, b, det);
  else return -1;
  return 0;
}
extern "C" int mpp_launch_shapenet_epilogue_nhwc(hipStream_t st, const void *logits, int elem_bytes, int H, int W, int ldw,
                                                 float *marks) {
  if (((uintptr_t)logits & 15) || ((uintptr_t)marks & 15)) return -2;
  const size_t threads = (size_t)H * W * 4;
  const unsigned grid = (unsigned)((threads + 255) / 256);
  if (elem_bytes == 4) hipLaunchKernelGGL(k_shapenet_epilogue_nhwc<4>, dim3(grid), dim3(256), 0, st, logits, H, W, ldw, marks);
  else if (elem_bytes == 2) hipLaunchKernelGGL(k_shapenet_epilogue_nhwc<2>, dim3(grid), dim3(256), 0, st, logits, H, W, ldw, marks);
  else return -1;
  return 0;
}


// ---- remapped mark tables: out[i] = -2 * sigmoid(coef * P[i] + icpt) + 1 for every (pixel, class) of one mark map, in
// the arithmetic of unit_value(MPP_U_SHAPE_REMAP) (the reference: apply_remap_param_dist, energy_calibration.py:134-139,
// once per tile in energy_setup_legacy.py:142-147).  HBM-bound: 4 B read + 8 B written per entry.
__global__ __launch_bounds__(256) void k_remap_table(const float *__restrict__ m, size_t n, double coef, double icpt,
                                                     double *__restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double p = (double)m[i];
    const double e = exp(-(p * coef + icpt));
    out[i] = -2.0 * (1.0 / (1.0 + e)) + 1.0;
  }
}
extern "C" void mpp_launch_remap_table(hipStream_t st, const float *m, size_t n, double coef, double icpt, double *out) {
  const size_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(k_remap_table, dim3((unsigned)(blocks < 65536 * 16 ? blocks : 65536 * 16)), dim3(256), 0, st, m, n, coef, icpt, out);
}
