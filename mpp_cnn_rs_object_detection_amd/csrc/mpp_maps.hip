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
