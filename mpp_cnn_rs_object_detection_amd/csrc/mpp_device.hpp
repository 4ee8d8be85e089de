// mpp_device.hpp -- device-side data model shared by the sampler and the from-scratch
// energy kernels (gfx950 / wave64 only).
//
// Arithmetic policy: score maps are float32 in HBM; every energy, density and
// acceptance quantity is computed in float64 from those reads (the reference works
// in NumPy float64 on float32 maps).  The library is compiled with
// -ffp-contract=off so that no FMA is formed that the CPU oracle does not form.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mpp_hip.h"

#define WAVE 64
#define MPP_PI 3.14159265358979323846
#define MPP_TWO_PI 6.283185307179586476925286766559
#define EPS_GREEN 1e-16       // rjmcmc.py:15
#define AREA_EPS 1e-6         // prior_energies.py:18
#define DEGENERATE_AREA 1e-12 // a zero-width rectangle is a segment: intersection area 0

// everything a chain needs that is the same for all tiles of a ctx (lives in HBM, read
// through scalar loads: it is wave-uniform)
struct DevParams {
  mpp_model model;
  mpp_mappings maps;
  mpp_kernels kern;
  double p_cum[MPP_NKERNEL];
  double max_inter;     // largest pair max_dist (energy_graph.py:26-29)
  double res;           // cell size = max(max_inter, 32) (point_set.py:58)
  int32_t H, W, nx, ny; // grid dims (point_set.py:59-61)
  int32_t cap, cell_cap, n_tiles, res_int;
  int32_t res_shift;    // log2(res_int) if it is a power of two, else -1
  int32_t uniform_bins; // the mark edges are linspace(vmin,vmax,33)[:-1] (checked on the host)
  int32_t maxd2[MPP_MAX_PAIR];   // floor(max_dist^2): d <= max_dist  <=>  d2 <= maxd2 for integer d2
  int32_t conflict_d2;  // (2*max_inter)^2 + 1: two proposals further apart cannot influence each other
  int32_t n_kernels;    // 8, or 10 when the split / merge kernels carry probability
  int32_t rowbase_lds;  // the birth CDF's row level ([H+1] doubles) is staged in LDS (set per launch by the host)
  int32_t force_accept; // apply every proposal (perturbation_sampler.py:162-166 walks kernels without accept/reject)
  int32_t handover;     // (one wave per step, 8 waves) > 0: stop with ERR_HANDOVER once the smoothed number of steps committed
                        // per round reaches handover / 256 (~6 of 8): the host then continues the chain in deep rounds (set per
                        // launch by the host)
  double inv_step[3];   // 32 / (vmax - vmin)
};

// Pointers that are read back from a struct in memory carry no address space, and the device code would use
// flat loads for them; in device compilation the score-map pointers are declared global (same size and layout).
#if defined(__HIP_DEVICE_COMPILE__)
#define MPP_GLOBAL __attribute__((address_space(1)))
#else
#define MPP_GLOBAL
#endif

// per-tile device pointers
struct TileRef {
  const MPP_GLOBAL float *det;
  const MPP_GLOBAL float *m[3];
  const MPP_GLOBAL double *rowpart;   // [H][W] inclusive partial sums of det within each row
  const MPP_GLOBAL double *rowbase;   // [H+1] exclusive prefix of row totals; rowbase[H] = sum(det)
  const MPP_GLOBAL double *boxsum;    // [H][W] sum of det over the (2*max_delta+1)^2 window clipped to the tile
  // optional [H][W][32] tables of the remapped mark probabilities -2*sigmoid(coef_k*P_k+icpt_k)+1 (the reference builds
  // these maps once per tile, energy_setup_legacy.py:142-147); nullptr: the three sigmoids are evaluated per proposal
  const MPP_GLOBAL double *rm[3];
  // the picture behind the classic image energies (mpp_set_image): [H][W][img_c] float32, or nullptr
  const MPP_GLOBAL float *img;
  int32_t img_c, _pad_img;
  // point configuration, dense slots (capacity cap)
  int32_t *px, *py;
  double *ps, *pr, *pa;
  int32_t *n;
  double *T;               // current temperature
  int64_t *step;           // steps done so far
  int32_t *err;            // sticky error code of the chain
  double intensity;
  // The chain's own Philox key and chain id (mpp_set_chain_keys): tiles of SEVERAL images sampled in one launch keep the
  // seed of their image and their tile index within it, i.e. the chain they would run in a launch of their own.
  // key_on == 0: the launch's seed and chain0 + tile.
  uint64_t key_seed;
  uint32_t key_chain, key_on;
};

struct Rect {
  int x, y;
  double s, r, a;
};

// derived geometry of a rectangle (base/shapes/rectangle.py:20-31,69-100):
// length = 2*size/(1+ratio), width = ratio*length, corners = R(angle+pi/2)*(+-length/2,+-width/2)+centre
struct Geo {
  int x, y;
  double hl, hw, ca, sa;
};

__device__ __forceinline__ Geo make_geo(const Rect &q) {
  Geo g;
  g.x = q.x; g.y = q.y;
  double length = (2.0 * q.s) / (1.0 + q.r);
  double width = q.r * length;
  g.hl = length / 2.0; g.hw = width / 2.0;
  double al = q.a + MPP_PI / 2.0;
  g.ca = cos(al); g.sa = sin(al);
  return g;
}
__device__ __forceinline__ double geo_area(const Geo &g) { return (2.0 * g.hl) * (2.0 * g.hw); }

__device__ __forceinline__ void geo_corners(const Geo &g, double *px, double *py) {
  // counter-clockwise: (+,+) (-,+) (-,-) (+,-)
  const double sx[4] = {1, -1, -1, 1}, sy[4] = {1, 1, -1, -1};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    double vx = sx[i] * g.hl, vy = sy[i] * g.hw;
    px[i] = g.ca * vx - g.sa * vy + (double)g.x;
    py[i] = g.sa * vx + g.ca * vy + (double)g.y;
  }
}

// Sutherland-Hodgman: area of (convex quad S) clipped by (convex ccw quad C)
__device__ inline double clip_area(const double *sx, const double *sy, const double *cx, const double *cy) {
  double ax[8], ay[8], bx[8], by[8];
  int na = 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) { ax[i] = sx[i]; ay[i] = sy[i]; }
  for (int e = 0; e < 4 && na > 0; ++e) {
    double x0 = cx[e], y0 = cy[e], x1 = cx[(e + 1) & 3], y1 = cy[(e + 1) & 3];
    double ex = x1 - x0, ey = y1 - y0;
    int nb = 0;
    double px = ax[na - 1], py = ay[na - 1];
    double sp = ex * (py - y0) - ey * (px - x0);
    for (int i = 0; i < na; ++i) {
      double qx = ax[i], qy = ay[i];
      double sq = ex * (qy - y0) - ey * (qx - x0);
      if (sq >= 0) {
        if (sp < 0 && nb < 8) {
          double t = sp / (sp - sq);
          bx[nb] = px + t * (qx - px); by[nb] = py + t * (qy - py); ++nb;
        }
        if (nb < 8) { bx[nb] = qx; by[nb] = qy; ++nb; }
      } else if (sp >= 0 && nb < 8) {
        double t = sp / (sp - sq);
        bx[nb] = px + t * (qx - px); by[nb] = py + t * (qy - py); ++nb;
      }
      px = qx; py = qy; sp = sq;
    }
    for (int i = 0; i < nb; ++i) { ax[i] = bx[i]; ay[i] = by[i]; }
    na = nb;
  }
  if (na < 3) return 0.0;
  double s = 0.0;
  for (int i = 0; i < na; ++i) {
    int j = (i + 1 == na) ? 0 : i + 1;
    s += ax[i] * ay[j] - ax[j] * ay[i];
  }
  return 0.5 * fabs(s);
}

// a strict total order on rectangles; the pair energy always clips the smaller against the
// larger, which makes it a function of the unordered pair (the reference shares one PairEnergy
// object between both endpoints, energy_graph.py:74-77)
__device__ __forceinline__ bool rect_less(int ax, int ay, double as, double ar, double aa, int bx, int by, double bs,
                                          double br, double ba) {
  if (ax != bx) return ax < bx;
  if (ay != by) return ay < by;
  if (as != bs) return as < bs;
  if (ar != br) return ar < br;
  return aa < ba;
}

#ifdef MPP_PROFILE
__device__ unsigned long long g_clip_count;
#endif
// RectangleOverlapEnergy (prior_energies.py:11-24); `u_first` tells whether u is the subject;
// ru, rv: circumscribed-circle radii, d2: squared centre distance
__device__ inline double overlap_energy_r(const Geo &u, const Geo &v, bool u_first, double ru, double rv, double d2) {
  double A = geo_area(u), B = geo_area(v);
  double mn = A < B ? A : B;
  if (mn < DEGENERATE_AREA) return 0.0;
  double reach = ru + rv;
  if (d2 > reach * reach * 1.0000001) return 0.0;                  // circumscribed circles apart
#ifdef MPP_PROFILE
  atomicAdd(&g_clip_count, 1ull);
#endif
  double ax[4], ay[4], bx[4], by[4];
  if (u_first) { geo_corners(u, ax, ay); geo_corners(v, bx, by); }
  else { geo_corners(v, ax, ay); geo_corners(u, bx, by); }
  return clip_area(ax, ay, bx, by) / (mn + AREA_EPS);
}
__device__ __forceinline__ double geo_radius(const Geo &g) { return sqrt(g.hl * g.hl + g.hw * g.hw); }
__device__ inline double overlap_energy(const Geo &u, const Geo &v, bool u_first) {
  double dx = (double)(u.x - v.x), dy = (double)(u.y - v.y);
  return overlap_energy_r(u, v, u_first, geo_radius(u), geo_radius(v), dx * dx + dy * dy);
}

__device__ __forceinline__ double sigmoid_d(double x) { return 1.0 / (1.0 + exp(-x)); }

// mappings.py:44-62: class = max{i : v >= edge_i}.  `edges` = the 32 lower bin edges of mark k
// (HBM copy in DevParams or the chain's LDS copy).  For linspace bins the class is computed and
// then corrected against the actual table, so the result is exactly the table's.
__device__ __forceinline__ int value_to_class_tab(const DevParams *P, const double *edges, int k, double v) {
  if (P->uniform_bins) {
    int g = (int)floor((v - P->maps.vmin[k]) * P->inv_step[k]);
    g = g < 0 ? 0 : (g > MPP_NCLASS - 1 ? MPP_NCLASS - 1 : g);
    double e0 = edges[g], e1 = edges[g < MPP_NCLASS - 1 ? g + 1 : g];
    if (v < e0) { if (g > 0) --g; }
    else if (g < MPP_NCLASS - 1 && v >= e1) ++g;
    return g;
  }
  int lo = 0, hi = MPP_NCLASS;
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (v >= edges[mid]) lo = mid; else hi = mid;
  }
  return lo;
}
__device__ __forceinline__ int value_to_class(const DevParams *P, int k, double v) {
  return value_to_class_tab(P, P->maps.edges[k], k, v);
}
// (values first, then the choice: a conditional over the lvalues becomes a pointer phi and pins the struct in scratch)
__device__ __forceinline__ double mark_of(const Rect &q, int k) {
  const double s = q.s, r = q.r, a = q.a;
  return k == 0 ? s : (k == 1 ? r : a);
}
__device__ __forceinline__ void set_mark(Rect &q, int k, double v) {
  const double s = q.s, r = q.r, a = q.a;
  q.s = k == 0 ? v : s; q.r = k == 1 ? v : r; q.a = k == 2 ? v : a;
}
__device__ __forceinline__ const float *mark_row_w(int W, const TileRef &t, int k, int x, int y) {
  const float *m0 = t.m[0], *m1 = t.m[1], *m2 = t.m[2];
  const float *b = k == 0 ? m0 : (k == 1 ? m1 : m2);   // no runtime-indexed private array
  return b + ((size_t)x * W + y) * MPP_NCLASS;
}
__device__ __forceinline__ const float *mark_row(const DevParams *P, const TileRef &t, int k, int x, int y) {
  return mark_row_w(P->W, t, k, x, y);
}

// the score-map values a rectangle's unit terms can ask for: det at its pixel and, per mark, the probability of the
// mark's class there.  Fetched together (four independent loads, one memory latency) before the terms are evaluated.
struct MapVals { float det, m0, m1, m2; double r0, r1, r2; int tab; };
// (W: the tile width, passed separately so that the chain can hand over its register copy instead of a scalar load)
__device__ __forceinline__ MapVals load_map_vals_w(const DevParams *P, int W, const TileRef &t, const double *edges, const Rect &q) {
  const size_t px = (size_t)q.x * W + q.y, pix = px * MPP_NCLASS;
  const int c0 = value_to_class_tab(P, edges, 0, q.s), c1 = value_to_class_tab(P, edges + MPP_NCLASS, 1, q.r),
            c2 = value_to_class_tab(P, edges + 2 * MPP_NCLASS, 2, q.a);
  MapVals v;
  v.det = t.det[px];
  if (t.rm[0] != nullptr) {        // (the host builds the tables only for models whose sole use of the marks is SHAPE_REMAP)
    v.r0 = t.rm[0][pix + c0]; v.r1 = t.rm[1][pix + c1]; v.r2 = t.rm[2][pix + c2];
    v.m0 = v.m1 = v.m2 = 0.f; v.tab = 1;
  } else {
    v.m0 = t.m[0][pix + c0]; v.m1 = t.m[1][pix + c1]; v.m2 = t.m[2][pix + c2];
    v.r0 = v.r1 = v.r2 = 0.0; v.tab = 0;
  }
  return v;
}
__device__ __forceinline__ MapVals load_map_vals(const DevParams *P, const TileRef &t, const double *edges, const Rect &q) {
  return load_map_vals_w(P, P->W, t, edges, q);
}
__device__ __forceinline__ float map_mark(const MapVals &v, int k) {
  const float a = v.m0, b = v.m1, c = v.m2;
  return k == 0 ? a : (k == 1 ? b : c);
}

#include "mpp_classics.hpp"

// one unit energy term of a rectangle.  CL: the classic image energies are compiled in (the chain kernel's extended
// instantiations and the from-scratch kernels); elsewhere the host never selects a kernel for a model that has them.
// `one_lane`: the whole wave asks for the same rectangle (the chain's wave mode): the contrast term is then evaluated by
// the wave together (classic_contrast_wave); the gradient term by lane 0 alone -- its outline lives in private memory, 64
// lanes doing the same would move 64 times the bytes -- which hands the value to the others.
// `contrast_pre`: the value of the rectangle's ContrastEnergy term, already computed by the whole wave
// (classic_contrast_wave; the deep-round kernel evaluates the terms of its steps one rectangle at a time before the lanes
// go their own ways).
template <bool CL = false>
__device__ inline double unit_value(const mpp_unit_term &u, const Rect &q, const Geo &g, const MapVals &mv,
                                    const TileRef &t, int H, int W, bool one_lane = false, const double *contrast_pre = nullptr) {
  if (CL) {
    if (u.kind == MPP_U_CONTRAST && contrast_pre) return *contrast_pre;
    // (wave mode: every lane is here with the same rectangle -- the wave works on it together)
    if (u.kind == MPP_U_CONTRAST && one_lane) return classic_contrast_wave(u, t.img, t.img_c, H, W, g, (int)(threadIdx.x & 63));
    if (u.kind == MPP_U_CONTRAST || u.kind == MPP_U_GRADIENT) {
      double v = 0.0;
      if (!one_lane || (threadIdx.x & 63) == 0)
        v = u.kind == MPP_U_CONTRAST ? classic_contrast(u, t.img, t.img_c, H, W, g) : classic_gradient(u, t.img, t.img_c, H, W, g);
      if (one_lane) {
        long long b = __double_as_longlong(v);
        const int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffll)), hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
        v = __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
      }
      return v;
    }
  }
  switch (u.kind) {
    case MPP_U_POSITION: {
      float e = -2.0f * (mv.det - (float)u.p[0]);   // float32, as numpy does
      return (double)e;
    }
    case MPP_U_SHAPE_REMAP: {
      if (mv.tab) {                  // table entries hold exactly the three summands formed below
        double acc = 0.0;
        acc += mv.r0; acc += mv.r1; acc += mv.r2;
        return acc / 3.0;
      }
      // the three sigmoids are independent chains: keep them side by side
      double p0 = (double)mv.m0, p1 = (double)mv.m1, p2 = (double)mv.m2;
      double e0 = exp(-(p0 * u.p[0] + u.p[3])), e1 = exp(-(p1 * u.p[1] + u.p[4])), e2 = exp(-(p2 * u.p[2] + u.p[5]));
      double acc = 0.0;
      acc += -2.0 * (1.0 / (1.0 + e0)) + 1.0;
      acc += -2.0 * (1.0 / (1.0 + e1)) + 1.0;
      acc += -2.0 * (1.0 / (1.0 + e2)) + 1.0;
      return acc / 3.0;
    }
    case MPP_U_MARK_NEG: return -(double)map_mark(mv, (int)u.p[0]);
    case MPP_U_MARK_REMAP: {
      double p = (double)map_mark(mv, (int)u.p[0]);
      return -2.0 * sigmoid_d(p * u.p[1] + u.p[2]) + 1.0;
    }
    case MPP_U_AREA: {
      double A = geo_area(g), lo = u.p[0] - A, hi = A - u.p[1];
      double m = lo > hi ? lo : hi;
      return m > 0.0 ? m : 0.0;
    }
    case MPP_U_RATIO_PRIOR: return fabs(u.p[0] - q.r);
    case MPP_U_CONST: return u.p[0];
  }
  return 0.0;
}

// the part of a point's combined energy that does not depend on its neighbours:
// lin = lin0 + sum_units coef*g*v ; gate = [v_gate <= thr]
// (two entry points: with the score-map values already fetched -- the chain issues those loads as early as it knows the
// pixel and the marks, so that their latency overlaps the proposal densities and the trigonometry -- and without)
template <bool CL = false>
__device__ inline void unit_part_mv(const DevParams *P, const TileRef &t, const MapVals &mv, const Rect &q, const Geo &g,
                                    double *lin, int *gate, double *vec_or_null, bool one_lane = false,
                                    const double *contrast_pre = nullptr) {
  const mpp_model &M = P->model;
  const int H = P->H, W = P->W;
  // the gating term first (no local array: a runtime-indexed one would live in scratch memory)
  double vg = 0.0;
  int gt = 1;
  if (M.gate_term >= 0) {
    vg = unit_value<CL>(M.unit[M.gate_term], q, g, mv, t, H, W, one_lane, contrast_pre);
    gt = (vg <= M.gate_thr) ? 1 : 0;
  }
  double l = M.lin0;
  for (int k = 0; k < M.n_unit; ++k) {
    double v = (k == M.gate_term) ? vg : unit_value<CL>(M.unit[k], q, g, mv, t, H, W, one_lane, contrast_pre);
    if (vec_or_null) vec_or_null[k] = v;
    l += M.unit[k].coef * ((M.unit[k].gated ? (double)gt : 1.0)) * v;
  }
  *lin = l; *gate = gt;
}
template <bool CL = false>
__device__ inline void unit_part(const DevParams *P, const TileRef &t, const double *edges, const Rect &q,
                                 const Geo &g, double *lin, int *gate, double *vec_or_null, bool one_lane = false) {
  const MapVals mv = load_map_vals(P, t, edges, q);
  unit_part_mv<CL>(P, t, mv, q, g, lin, gate, vec_or_null, one_lane);
}
__device__ __forceinline__ double pair_part(const DevParams *P, int gate, double r0, double r1) {
  const mpp_model &M = P->model;
  double l = 0.0;
  if (M.n_pair > 0) l += M.pair[0].coef * (M.pair[0].gated ? (double)gate : 1.0) * r0;
  if (M.n_pair > 1) l += M.pair[1].coef * (M.pair[1].gated ? (double)gate : 1.0) * r1;
  return l;
}
__device__ __forceinline__ double finish_energy(const DevParams *P, double lin) {
  return P->model.combinator == MPP_C_LOGISTIC ? 2.0 * sigmoid_d(lin) - 1.0 : lin;
}
__device__ __forceinline__ double reduce2(int mode, double a, double b) {
  return mode == MPP_REDUCE_MAX ? (a > b ? a : b) : (a < b ? a : b);
}

// ---- wave64 helpers -------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
  return v;
}
__device__ __forceinline__ double wave_reduce(int mode, double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = reduce2(mode, v, __shfl_xor(v, o, WAVE));
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
  return v;
}

// ---- Philox4x32-10 --------------------------------------------------------------------------
__host__ __device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ double u53(uint32_t a, uint32_t b) {
  return (double)((((uint64_t)(a >> 5)) << 26) | (uint64_t)(b >> 6)) * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ double u32d(uint32_t a) { return (double)a * (1.0 / 4294967296.0); }
__device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t n) { return (uint32_t)(((uint64_t)a * n) >> 32); }
