// Classic image energies of the contrast energy setup (models/mpp/energies/classics.py:100-238,
// energy_setups/energy_setup_contrast.py:29-105): unit terms that look at the PICTURE under a rectangle instead of at a
// score map.  Included by mpp_device.hpp.
//
// The reference rasterises with scikit-image 0.18.1 (draw.polygon: every pixel of the clipped bounding box put to the
// even-odd crossing test of _shared/geometry.pyx; draw.polygon_perimeter: corners rounded half to even, one Bresenham
// line per edge) and grows / shrinks the pixel set with its own 4-neighbour dilation (utils/morpho.py:9-19).  Here ONE
// THREAD evaluates a rectangle on bit rows -- row i of the bounding box (plus a margin) is a 128-bit word:
//  * the fill mask of a row is the XOR of "all columns left of the crossing" prefixes of the edges that span the row,
//    which is exactly the parity the crossing test computes pixel by pixel (same float64 expression for the crossing);
//  * a dilation step is  row | row << 1 | row >> 1 | row above | row below, clipped to the image;
//  * the statistics walk the set bits in row-major order, two passes (mean, then variance), as numpy does.
// That one-thread form serves the from-scratch kernels and the chain's set-up; the steps of a chain use the
// wave-cooperative form further down (rows, then columns, across lanes; registers only), which returns the same bits.
#pragma once

#define MPP_CL_ROWS 96        // bounding box of the largest rectangle the mappings allow (size <= 32) + margins
#define MPP_CL_COLS 128
#define MPP_CL_OUTLINE 320    // pixels of four Bresenham edges of such a rectangle

struct Bits128 { unsigned long long lo, hi; };
__device__ __forceinline__ Bits128 b_or(Bits128 a, Bits128 b) { return Bits128{a.lo | b.lo, a.hi | b.hi}; }
__device__ __forceinline__ Bits128 b_and(Bits128 a, Bits128 b) { return Bits128{a.lo & b.lo, a.hi & b.hi}; }
__device__ __forceinline__ Bits128 b_andn(Bits128 a, Bits128 b) { return Bits128{a.lo & ~b.lo, a.hi & ~b.hi}; }
__device__ __forceinline__ Bits128 b_xor(Bits128 a, Bits128 b) { return Bits128{a.lo ^ b.lo, a.hi ^ b.hi}; }
__device__ __forceinline__ Bits128 b_shl1(Bits128 a) { return Bits128{a.lo << 1, (a.hi << 1) | (a.lo >> 63)}; }
__device__ __forceinline__ Bits128 b_shr1(Bits128 a) { return Bits128{(a.lo >> 1) | (a.hi << 63), a.hi >> 1}; }
__device__ __forceinline__ bool b_any(Bits128 a) { return (a.lo | a.hi) != 0ull; }
// bits [0, k)
__device__ __forceinline__ Bits128 b_prefix(int k) {
  if (k <= 0) return Bits128{0ull, 0ull};
  if (k >= 128) return Bits128{~0ull, ~0ull};
  if (k >= 64) return Bits128{~0ull, k == 64 ? 0ull : ((1ull << (k - 64)) - 1ull)};
  return Bits128{(1ull << k) - 1ull, 0ull};
}

struct ClGrid { int r0, c0, nr; Bits128 cols; };     // origin, rows in use, columns that lie inside the image

// rect_to_poly's vertex order (base/shapes/rectangle.py:85-88): (+,+) (+,-) (-,-) (-,+) = corners 0, 3, 2, 1 of geo_corners()
__device__ inline void cl_ref_corners(const Geo &g, double *r, double *c) {
  double x[4], y[4];
  geo_corners(g, x, y);
  r[0] = x[0]; c[0] = y[0]; r[1] = x[3]; c[1] = y[3]; r[2] = x[2]; c[2] = y[2]; r[3] = x[1]; c[3] = y[1];
}

// utils/morpho.py:9-19, n_iter steps
__device__ inline void cl_dilate(const ClGrid &G, int H, Bits128 *a, Bits128 *tmp, int n_iter) {
  for (int it = 0; it < n_iter; ++it) {
    for (int i = 0; i < G.nr; ++i) tmp[i] = a[i];
    for (int i = 0; i < G.nr; ++i) {
      const int row = G.r0 + i;
      if (row < 0 || row >= H) continue;
      Bits128 v = b_or(tmp[i], b_or(b_shl1(tmp[i]), b_shr1(tmp[i])));
      if (i > 0) v = b_or(v, tmp[i - 1]);
      if (i + 1 < G.nr) v = b_or(v, tmp[i + 1]);
      a[i] = b_and(v, G.cols);
    }
  }
}
__device__ inline int cl_count(const ClGrid &G, const Bits128 *a) {
  int n = 0;
  for (int i = 0; i < G.nr; ++i) n += __popcll(a[i].lo) + __popcll(a[i].hi);
  return n;
}
// Sums over the pixels of a mask for up to three channels: out[ch] = sum x (mu == nullptr) or sum (x - mu[ch])^2.  The ORDER of
// the additions is the one the wave-cooperative form below can reproduce with one lane per column: every column adds its
// pixels from the first row to the last, column j and column j + 64 are added, and the 64 values are combined by the
// butterfly of a wave reduction (partner lane ^ 32, ^ 16, ... ^ 1) -- so a value computed by one thread here (from-scratch
// kernels, the chain's set-up) and by a wave there (the chain's steps) is the same bit pattern.  The reference adds in
// row-major order (numpy, two passes); the results agree to a few 1e-16 relative (tests: 1e-12).
__device__ inline int cl_sums(const ClGrid &G, const Bits128 *m, const MPP_GLOBAL float *img, int W, int C, const double *mu,
                              double *acc) {
  int n = 0;
  for (int i = 0; i < G.nr; ++i) n += __popcll(m[i].lo) + __popcll(m[i].hi);
  for (int ch = 0; ch < 3; ++ch) {
    acc[ch] = 0.0;
    if (ch >= C) continue;
    double col[MPP_CL_COLS];
    for (int j = 0; j < MPP_CL_COLS; ++j) col[j] = 0.0;
    for (int i = 0; i < G.nr; ++i)
      for (int h = 0; h < 2; ++h) {
        unsigned long long w = h ? m[i].hi : m[i].lo;
        while (w) {
          const int j = __ffsll((long long)w) - 1 + 64 * h;
          w &= w - 1;
          const double x = (double)img[((size_t)(G.r0 + i) * W + (G.c0 + j)) * (size_t)C + ch];
          if (mu) { const double d = x - mu[ch]; col[j] += d * d; } else col[j] += x;
        }
      }
    double t[64], t2[64];
    for (int l = 0; l < 64; ++l) t[l] = col[l] + col[l + 64];
    for (int o = 32; o > 0; o >>= 1) {
      for (int l = 0; l < 64; ++l) t2[l] = t[l] + t[l ^ o];
      for (int l = 0; l < 64; ++l) t[l] = t2[l];
    }
    acc[ch] = t[0];
  }
  return n;
}
// mean and (population) variance per channel over the pixels of a mask, two passes as numpy does
__device__ inline int cl_stats(const ClGrid &G, const Bits128 *m, const MPP_GLOBAL float *img, int W, int C, double *mean,
                               double *var) {
  double acc[3];
  const int n = cl_sums(G, m, img, W, C, nullptr, acc);
  for (int ch = 0; ch < 3; ++ch) mean[ch] = acc[ch] / (double)n;
  cl_sums(G, m, img, W, C, mean, acc);
  for (int ch = 0; ch < 3; ++ch) var[ch] = acc[ch] / (double)n;
  return n;
}
// the contrast measures, classics.py:13-97
__device__ inline double cl_measure(int type, double mi, double mo, double vi, double vo, int ni, int no) {
  const double eps = 1e-8, d = mi - mo;
  switch (type) {
    case 0: return sqrt((vo + vi) / ((double)(no + ni) * (d * d) + eps));                               // lafarge
    case 1: return (d * d) / (4.0 * sqrt(vi + vo)) + (-0.5 * log((2.0 * sqrt(vi * vo)) / (vi + vo)));   // craciun
    case 2: return (d * d) / (4.0 * sqrt(vi + vo) + eps);                                               // craciun2
    case 3: return d * d;                                                                               // mean
    case 4: return fabs(d) / sqrt((vi / (double)ni) + (vo / (double)no) + eps);                         // t-test
    default: return fabs(d);                                                                            // debug
  }
}

// ContrastEnergy.compute (classics.py:151-196); u.p = {measure, dilation, gap, erode, thresh, fac, default_value}
__device__ __noinline__ double classic_contrast(const mpp_unit_term &u, const MPP_GLOBAL float *img, int C, int H, int W,
                                                const Geo &g) {
  double r[4], c[4];
  cl_ref_corners(g, r, c);
  double rmin = r[0], rmax = r[0], cmin = c[0], cmax = c[0];
  for (int i = 1; i < 4; ++i) {
    rmin = r[i] < rmin ? r[i] : rmin; rmax = r[i] > rmax ? r[i] : rmax;
    cmin = c[i] < cmin ? c[i] : cmin; cmax = c[i] > cmax ? c[i] : cmax;
  }
  // skimage/draw/_draw.pyx _polygon: int(max(0, min)), int(ceil(max)), clipped to the shape
  const int minr = (int)(rmin > 0.0 ? rmin : 0.0), minc = (int)(cmin > 0.0 ? cmin : 0.0);
  int maxr = (int)ceil(rmax), maxc = (int)ceil(cmax);
  maxr = maxr > H - 1 ? H - 1 : maxr; maxc = maxc > W - 1 ? W - 1 : maxc;
  if (maxr < minr || maxc < minc) return u.p[6];
  const int dil = (int)u.p[1], gap = (int)u.p[2], ero = (int)u.p[3];
  const int reach = (2 + ero) > (gap + dil) ? (2 + ero) : (gap + dil);
  const int M = 1 + reach;
  ClGrid G;
  G.r0 = minr - M; G.c0 = minc - M; G.nr = maxr - minr + 1 + 2 * M;
  const int nc = maxc - minc + 1 + 2 * M;
  if (G.nr > MPP_CL_ROWS || nc > MPP_CL_COLS) return nan("");      // excluded by mpp_set_model (size range of the mappings)
  {
    // columns of the grid that are columns of the image
    const int lo = G.c0 < 0 ? -G.c0 : 0, hi = (W - G.c0) < nc ? (W - G.c0) : nc;
    G.cols = b_andn(b_prefix(hi), b_prefix(lo));
  }
  Bits128 fill[MPP_CL_ROWS], a[MPP_CL_ROWS], b[MPP_CL_ROWS];
  const Bits128 window = b_andn(b_prefix(maxc - G.c0 + 1), b_prefix(minc - G.c0));
  for (int i = 0; i < G.nr; ++i) {
    const int row = G.r0 + i;
    Bits128 acc{0ull, 0ull};
    if (row >= minr && row <= maxr) {
      const double y = (double)row;
      int j = 3;
      for (int e = 0; e < 4; ++e) {
        if (((r[e] <= y) && (y < r[j])) || ((r[j] <= y) && (y < r[e]))) {
          const double xc = (c[j] - c[e]) * (y - r[e]) / (r[j] - r[e]) + c[e];      // point_in_polygon's crossing
          double k = ceil(xc) - (double)G.c0;            // columns x < xc  <=>  bit index < ceil(xc) - c0
          k = k < 0.0 ? 0.0 : (k > 128.0 ? 128.0 : k);
          acc = b_xor(acc, b_prefix((int)k));
        }
        j = e;
      }
      acc = b_and(acc, window);
    }
    fill[i] = acc;
  }
  if (cl_count(G, fill) == 0) return u.p[6];
  if (ero > 0) {                                   // classics.py:178-182
    for (int i = 0; i < G.nr; ++i) a[i] = fill[i];
    cl_dilate(G, H, a, b, 2);
    for (int i = 0; i < G.nr; ++i) a[i] = b_andn(a[i], fill[i]);
    cl_dilate(G, H, a, b, ero);
    for (int i = 0; i < G.nr; ++i) fill[i] = b_andn(fill[i], a[i]);
    if (cl_count(G, fill) == 0) return u.p[6];
  }
  // a <- rim
  if (gap > 0) {                                   // :187-190
    Bits128 rim[MPP_CL_ROWS];
    for (int i = 0; i < G.nr; ++i) a[i] = fill[i];
    cl_dilate(G, H, a, b, gap);
    for (int i = 0; i < G.nr; ++i) rim[i] = a[i];
    cl_dilate(G, H, rim, b, dil);
    for (int i = 0; i < G.nr; ++i) a[i] = b_andn(rim[i], a[i]);
  } else {                                         // :191-193
    for (int i = 0; i < G.nr; ++i) a[i] = fill[i];
    cl_dilate(G, H, a, b, dil);
    for (int i = 0; i < G.nr; ++i) a[i] = b_andn(a[i], fill[i]);
  }
  double mi[3], mo[3], vi[3], vo[3];
  const int ni = cl_stats(G, fill, img, W, C, mi, vi), no = cl_stats(G, a, img, W, C, mo, vo);
  double val = 0.0;
  for (int ch = 0; ch < C; ++ch) val += u.p[5] * cl_measure((int)u.p[0], mi[ch], mo[ch], vi[ch], vo[ch], ni, no);
  return val - u.p[4];
}

// ---- ContrastEnergy.compute by a whole wave (all 64 lanes, the same rectangle in every lane) ---------------------------------
// Rows of the grid across lanes -- lane l holds rows l and l + 64 of every mask in registers (no private arrays: the
// one-thread form above keeps 6 KB of them in scratch memory and a chain's step took 15 us) -- for the fill and the
// dilations (the rows above and below come from the neighbouring lanes), then COLUMNS across lanes for the statistics:
// the masks of a row are broadcast, lane l looks at columns l and l + 64 of it (consecutive lanes read consecutive pixels)
// and adds in the order cl_sums() spells out; the same values bit for bit.
struct Rows2 { Bits128 a, b; };                       // rows lane and lane + 64
__device__ __forceinline__ Bits128 b_shfl(Bits128 v, int src) {
  return Bits128{(unsigned long long)__shfl((long long)v.lo, src, 64), (unsigned long long)__shfl((long long)v.hi, src, 64)};
}
__device__ __forceinline__ Bits128 b_readlane(Bits128 v, int src) {          // src wave-uniform
  const unsigned int x0 = (unsigned int)__builtin_amdgcn_readlane((int)(v.lo & 0xffffffffull), src),
                     x1 = (unsigned int)__builtin_amdgcn_readlane((int)(v.lo >> 32), src),
                     x2 = (unsigned int)__builtin_amdgcn_readlane((int)(v.hi & 0xffffffffull), src),
                     x3 = (unsigned int)__builtin_amdgcn_readlane((int)(v.hi >> 32), src);
  return Bits128{((unsigned long long)x1 << 32) | x0, ((unsigned long long)x3 << 32) | x2};
}
__device__ __forceinline__ int rows_count(const Rows2 &m) {
  int n = __popcll(m.a.lo) + __popcll(m.a.hi) + __popcll(m.b.lo) + __popcll(m.b.hi);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
  return n;
}
// utils/morpho.py:9-19, n_iter steps; rows at or beyond G.nr are zero and stay zero, rows outside the image keep their value
__device__ inline void rows_dilate(const ClGrid &G, int H, Rows2 &m, int n_iter, int lane) {
  const Bits128 zero{0ull, 0ull};
  const bool two = G.nr > 64;                          // (wave-uniform)
  const int row_a = G.r0 + lane, row_b = G.r0 + 64 + lane;
  const bool upd_a = lane < G.nr && row_a >= 0 && row_a < H, upd_b = 64 + lane < G.nr && row_b >= 0 && row_b < H;
  for (int it = 0; it < n_iter; ++it) {
    const Bits128 t = m.a;
    Bits128 up = b_shfl(t, lane > 0 ? lane - 1 : 0), dn = b_shfl(t, lane < 63 ? lane + 1 : 63);
    if (lane == 0) up = zero;
    Bits128 first_b = zero;
    if (two) first_b = b_readlane(m.b, 0);
    if (lane == 63) dn = first_b;
    Bits128 v = b_or(b_or(t, b_or(b_shl1(t), b_shr1(t))), b_or(up, dn));
    if (two) {
      const Bits128 tb = m.b, last_a = b_readlane(t, 63);
      Bits128 upb = b_shfl(tb, lane > 0 ? lane - 1 : 0), dnb = b_shfl(tb, lane < 63 ? lane + 1 : 63);
      if (lane == 0) upb = last_a;
      if (lane == 63) dnb = zero;
      const Bits128 vb = b_or(b_or(tb, b_or(b_shl1(tb), b_shr1(tb))), b_or(upb, dnb));
      if (upd_b) m.b = b_and(vb, G.cols);
    }
    if (upd_a) m.a = b_and(v, G.cols);
  }
}
// 64 x 64 bit transpose across the wave: lane r brings row r (bit c = column c), lane c leaves with column c (bit r = row r)
__device__ __forceinline__ unsigned long long bit_transpose64(unsigned long long x, int lane) {
#define MPP_TR_STEP(s_, m_)                                                                                             \
  {                                                                                                                     \
    const unsigned long long m = (m_), y = (unsigned long long)__shfl_xor((long long)x, (s_), 64);                       \
    x = (lane & (s_)) ? (((y >> (s_)) & m) | (x & ~m)) : ((x & m) | ((y << (s_)) & ~m));                                 \
  }
  MPP_TR_STEP(32, 0x00000000FFFFFFFFull) MPP_TR_STEP(16, 0x0000FFFF0000FFFFull) MPP_TR_STEP(8, 0x00FF00FF00FF00FFull)
  MPP_TR_STEP(4, 0x0F0F0F0F0F0F0F0Full) MPP_TR_STEP(2, 0x3333333333333333ull) MPP_TR_STEP(1, 0x5555555555555555ull)
#undef MPP_TR_STEP
  return x;
}
// a mask by columns: [h][set] = rows 64 * set .. of column 64 * h + lane
struct Cols4 { unsigned long long w[2][2]; };
__device__ inline Cols4 rows_to_cols(const Rows2 &m, bool two, bool wide, int lane) {
  Cols4 o;
  o.w[0][0] = bit_transpose64(m.a.lo, lane);
  o.w[0][1] = two ? bit_transpose64(m.b.lo, lane) : 0ull;
  o.w[1][0] = wide ? bit_transpose64(m.a.hi, lane) : 0ull;
  o.w[1][1] = (wide && two) ? bit_transpose64(m.b.hi, lane) : 0ull;
  return o;
}
// per-channel sums over two disjoint masks at once (f: the rectangle, r: its rim), lane = column (and column + 64): every
// lane walks the rows of ITS column from the first to the last -- the masks arrive transposed, so there is nothing to
// broadcast -- four pixels at a time (their loads are issued together, then added in row order: a wave alone on its SIMD
// hides no latency).  out[0..2] fill, [3..5] rim; the order of the additions is the one cl_sums() spells out.
__device__ inline void cols_sums(const ClGrid &G, const Cols4 &f, const Cols4 &r, bool two, bool wide, const MPP_GLOBAL float *img,
                                 int W, int C, const double *mu_f, const double *mu_r, double *out, int lane) {
  double acc_f[2][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}}, acc_r[2][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (h == 1 && !wide) continue;                                 // (wave-uniform)
    const MPP_GLOBAL float *colp = img + ((size_t)G.r0 * W + (G.c0 + 64 * h + lane)) * (size_t)C;
    const size_t row_stride = (size_t)W * (size_t)C;
#pragma unroll
    for (int set = 0; set < 2; ++set) {
      if (set == 1 && !two) continue;                              // (wave-uniform)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const unsigned int ff = (unsigned int)(half ? f.w[h][set] >> 32 : f.w[h][set]);
        unsigned int w = ff | (unsigned int)(half ? r.w[h][set] >> 32 : r.w[h][set]);
        const int base = 64 * set + 32 * half;
        while (__ballot(w != 0u) != 0ull) {
          float x[4][3];
          bool ok[4], isf[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            ok[q] = w != 0u;
            const int b = ok[q] ? __ffs((int)w) - 1 : 0;
            isf[q] = (ff >> b) & 1u;
            w &= w - 1u;
            const MPP_GLOBAL float *px = colp + (size_t)(base + b) * row_stride;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) x[q][ch] = ok[q] ? px[ch < C ? ch : 0] : 0.f;
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (!ok[q]) continue;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
              if (ch >= C) continue;
              double v = (double)x[q][ch];
              if (mu_f) { const double d = v - (isf[q] ? mu_f[ch] : mu_r[ch]); v = d * d; }
              if (isf[q]) acc_f[h][ch] += v; else acc_r[h][ch] += v;
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    double a = acc_f[0][ch] + acc_f[1][ch], b = acc_r[0][ch] + acc_r[1][ch];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a = a + __shfl_xor(a, o, 64); b = b + __shfl_xor(b, o, 64); }
    out[ch] = a; out[3 + ch] = b;
  }
}
__device__ inline double classic_contrast_wave(const mpp_unit_term &u, const MPP_GLOBAL float *img, int C, int H, int W,
                                               const Geo &g, int lane) {
  double r[4], c[4];
  cl_ref_corners(g, r, c);
  double rmin = r[0], rmax = r[0], cmin = c[0], cmax = c[0];
  for (int i = 1; i < 4; ++i) {
    rmin = r[i] < rmin ? r[i] : rmin; rmax = r[i] > rmax ? r[i] : rmax;
    cmin = c[i] < cmin ? c[i] : cmin; cmax = c[i] > cmax ? c[i] : cmax;
  }
  const int minr = (int)(rmin > 0.0 ? rmin : 0.0), minc = (int)(cmin > 0.0 ? cmin : 0.0);
  int maxr = (int)ceil(rmax), maxc = (int)ceil(cmax);
  maxr = maxr > H - 1 ? H - 1 : maxr; maxc = maxc > W - 1 ? W - 1 : maxc;
  if (maxr < minr || maxc < minc) return u.p[6];
  const int dil = (int)u.p[1], gap = (int)u.p[2], ero = (int)u.p[3];
  const int reach = (2 + ero) > (gap + dil) ? (2 + ero) : (gap + dil);
  const int M = 1 + reach;
  ClGrid G;
  G.r0 = minr - M; G.c0 = minc - M; G.nr = maxr - minr + 1 + 2 * M;
  const int nc = maxc - minc + 1 + 2 * M;
  if (G.nr > MPP_CL_ROWS || nc > MPP_CL_COLS) return nan("");
  {
    const int lo = G.c0 < 0 ? -G.c0 : 0, hi = (W - G.c0) < nc ? (W - G.c0) : nc;
    G.cols = b_andn(b_prefix(hi), b_prefix(lo));
  }
  const Bits128 window = b_andn(b_prefix(maxc - G.c0 + 1), b_prefix(minc - G.c0));
  const Bits128 zero{0ull, 0ull};
  auto fill_row = [&](int i) -> Bits128 {
    const int row = G.r0 + i;
    Bits128 acc{0ull, 0ull};
    if (i < G.nr && row >= minr && row <= maxr) {
      const double y = (double)row;
      int j = 3;
      for (int e = 0; e < 4; ++e) {
        if (((r[e] <= y) && (y < r[j])) || ((r[j] <= y) && (y < r[e]))) {
          const double xc = (c[j] - c[e]) * (y - r[e]) / (r[j] - r[e]) + c[e];
          double k = ceil(xc) - (double)G.c0;
          k = k < 0.0 ? 0.0 : (k > 128.0 ? 128.0 : k);
          acc = b_xor(acc, b_prefix((int)k));
        }
        j = e;
      }
      acc = b_and(acc, window);
    }
    return acc;
  };
  Rows2 fill{fill_row(lane), G.nr > 64 ? fill_row(64 + lane) : zero};
  if (rows_count(fill) == 0) return u.p[6];
  Rows2 a;
  if (ero > 0) {                                   // classics.py:178-182
    a = fill;
    rows_dilate(G, H, a, 2, lane);
    a.a = b_andn(a.a, fill.a); a.b = b_andn(a.b, fill.b);
    rows_dilate(G, H, a, ero, lane);
    fill.a = b_andn(fill.a, a.a); fill.b = b_andn(fill.b, a.b);
    if (rows_count(fill) == 0) return u.p[6];
  }
  a = fill;
  if (gap > 0) {                                   // :187-190
    rows_dilate(G, H, a, gap, lane);
    Rows2 rim = a;
    rows_dilate(G, H, rim, dil, lane);
    a.a = b_andn(rim.a, a.a); a.b = b_andn(rim.b, a.b);
  } else {                                         // :191-193
    rows_dilate(G, H, a, dil, lane);
    a.a = b_andn(a.a, fill.a); a.b = b_andn(a.b, fill.b);
  }
  const int ni = rows_count(fill), no = rows_count(a);
  double sums[6], mi[3], mo[3], vi[3], vo[3];
  const bool two = G.nr > 64, wide = nc > 64;                       // (wave-uniform)
  const Cols4 cf = rows_to_cols(fill, two, wide, lane), cr = rows_to_cols(a, two, wide, lane);
  cols_sums(G, cf, cr, two, wide, img, W, C, nullptr, nullptr, sums, lane);
  for (int ch = 0; ch < 3; ++ch) { mi[ch] = sums[ch] / (double)ni; mo[ch] = sums[3 + ch] / (double)no; }
  cols_sums(G, cf, cr, two, wide, img, W, C, mi, mo, sums, lane);
  for (int ch = 0; ch < 3; ++ch) { vi[ch] = sums[ch] / (double)ni; vo[ch] = sums[3 + ch] / (double)no; }
  double val = 0.0;
  for (int ch = 0; ch < C; ++ch) val += u.p[5] * cl_measure((int)u.p[0], mi[ch], mo[ch], vi[ch], vo[ch], ni, no);
  return val - u.p[4];
}

// skimage/draw/_draw.pyx _line; appends to (pr, pc) and returns the number of points
__device__ inline int cl_line(int r0, int c0, int r1, int c1, short *pr, short *pc, int room) {
  bool steep = false;
  int r = r0, c = c0, dr = abs(r1 - r0), dc = abs(c1 - c0);
  int sc = (c1 - c) > 0 ? 1 : -1, sr = (r1 - r) > 0 ? 1 : -1;
  if (dr > dc) { steep = true; int t = c; c = r; r = t; t = dc; dc = dr; dr = t; t = sc; sc = sr; sr = t; }
  if (dc + 1 > room) return -1;
  int d = 2 * dr - dc;
  for (int i = 0; i < dc; ++i) {
    if (steep) { pr[i] = (short)c; pc[i] = (short)r; } else { pr[i] = (short)r; pc[i] = (short)c; }
    while (d >= 0) { r += sr; d -= 2 * dc; }
    c += sc; d += 2 * dr;
  }
  pr[dc] = (short)r1; pc[dc] = (short)c1;
  return dc + 1;
}
// GradientEnergy.compute (classics.py:207-232); img = np.gradient of the picture, [H][W][C/2][2]; u.p = {thresh, eps}
__device__ __noinline__ double classic_gradient(const mpp_unit_term &u, const MPP_GLOBAL float *img, int C, int H, int W,
                                                const Geo &g) {
  double r[5], c[5];
  cl_ref_corners(g, r, c);
  r[4] = r[0]; c[4] = c[0];                                   // polygon_clip closes the polygon ...
  const int nv = (r[4] == r[3] && c[4] == c[3]) ? 4 : 5;      // ... and drops a repeated last vertex
  short pr[MPP_CL_OUTLINE], pc[MPP_CL_OUTLINE];
  int n = 0;
  for (int i = 0; i + 1 < nv; ++i) {
    const int k = cl_line((int)rint(r[i]), (int)rint(c[i]), (int)rint(r[i + 1]), (int)rint(c[i + 1]), pr + n, pc + n,
                          MPP_CL_OUTLINE - n);
    if (k < 0) return nan("");
    n += k;
  }
  int m = 0;                                                   // _coords_inside_image
  for (int i = 0; i < n; ++i)
    if (pr[i] >= 0 && pr[i] < H && pc[i] >= 0 && pc[i] < W) { pr[m] = pr[i]; pc[m] = pc[i]; ++m; }
  const double eps = u.p[1];
  double s = 0.0;
  // (8 outline pixels at a time: their up to 48 gradient values are requested together, then used in outline order)
  for (int i0 = 0; i0 < m; i0 += 8) {
    float gv[8][6];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = i0 + k < m ? i0 + k : i0;
      const MPP_GLOBAL float *gp = img + ((size_t)pr[i] * W + pc[i]) * C;
#pragma unroll
      for (int q = 0; q < 6; ++q) gv[k][q] = gp[q < C ? q : 0];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = i0 + k;
      if (i >= m) break;
      const int nx = i + 1 < m ? i + 1 : 0, pv = i > 0 ? i - 1 : m - 1;
      const double t1r = (double)(pr[nx] - pr[i]), t1c = (double)(pc[nx] - pc[i]);
      const double t2r = (double)(pr[pv] - pr[i]), t2c = (double)(pc[pv] - pc[i]);
      const double n1r = -t1c, n1c = t1r, n2r = t2c, n2c = -t2r;
      const double l1 = sqrt(n1r * n1r + n1c * n1c) + eps, l2 = sqrt(n2r * n2r + n2c * n2c) + eps;
      const double nr = 0.5 * (n1r / l1 + n2r / l2), ncl = 0.5 * (n1c / l1 + n2c / l2);
#pragma unroll
      for (int q = 0; q < 6; ++q) if (q < C) s += (double)gv[k][q] * ((q & 1) ? ncl : nr);
    }
  }
  const double mean = s / ((double)m * (double)C);
  return -fabs(mean) - u.p[0];
}
