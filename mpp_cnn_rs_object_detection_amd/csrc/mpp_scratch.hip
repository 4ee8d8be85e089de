// mpp_scratch.hip -- energies computed from scratch on the configuration in HBM.
//
// These back the EPointsSet facade of the reference (point_set/energy_point_set.py:80-116):
// total_energy(), energy_delta() for perturbations with LISTS of removals/additions (the
// aggregated perturbations of perturbation_sampler.py:176-211) and papangelou().  They are the
// callers' side of the hot path (merge_patches, final scoring, weight learning), not the chain
// itself, so they favour being obviously right: every point's pair reductions are recomputed by
// a scan over all points.  They double as the independent check of the chain's cached
// reductions (the reference's debug=True invariants, energy_point_set.py:126-152).
#include "mpp_device.hpp"

struct Overlay {
  int n_excl; const int32_t *excl;          // slots removed
  int n_extra; const int32_t *exy; const double *emarks;   // rectangles added
};

__device__ __forceinline__ Rect tile_rect(const TileRef &t, int i) {
  return Rect{t.px[i], t.py[i], t.ps[i], t.pr[i], t.pa[i]};
}
__device__ __forceinline__ Rect extra_rect(const Overlay &o, int e) {
  return Rect{o.exy[2 * e], o.exy[2 * e + 1], o.emarks[3 * e], o.emarks[3 * e + 1], o.emarks[3 * e + 2]};
}
__device__ __forceinline__ bool excluded(const Overlay &o, int slot) {
  for (int i = 0; i < o.n_excl; ++i) if (o.excl[i] == slot) return true;
  return false;
}
__device__ double pair_value_rects(const mpp_pair_term &pt, const Rect &u, const Geo &gu, const Rect &v, double d) {
  switch (pt.kind) {
    case MPP_P_OVERLAP: {
      Geo gv = make_geo(v);
      bool uf = rect_less(u.x, u.y, u.s, u.r, u.a, v.x, v.y, v.s, v.r, v.a);
      return overlap_energy(gu, gv, uf);
    }
    case MPP_P_ALIGN: {
      Geo gv = make_geo(v);
      return 1.0 - fabs(gu.ca * gv.ca + gu.sa * gv.sa) - (pt.p[0] != 0.0 ? 1.0 : 0.0);
    }
    case MPP_P_DIST_LE: return d <= pt.max_dist ? 1.0 : 0.0;
    case MPP_P_DIST_LT: return d < pt.max_dist ? 1.0 : 0.0;
  }
  return 0.0;
}

// energy vector (unit terms then pair reductions) of rectangle u in the state
// (configuration - excluded + extras); energy_graph.py:108-137
__device__ double point_energy(const DevParams *P, const TileRef &t, int n, const Rect &u, int self_slot,
                               int self_extra, const Overlay &o, double *vec_or_null) {
  const mpp_model &M = P->model;
  Geo gu = make_geo(u);
  double lin; int gate;
  unit_part(P, t, &P->maps.edges[0][0], u, gu, &lin, &gate, vec_or_null);
  double red[MPP_MAX_PAIR] = {0.0, 0.0};
  const int mi = (int)ceil(P->max_inter);
  for (int v = 0; v < n + o.n_extra; ++v) {
    Rect q;
    if (v < n) {
      if (v == self_slot || excluded(o, v)) continue;
      q.x = t.px[v]; q.y = t.py[v];
    } else {
      if (v - n == self_extra) continue;
      q.x = o.exy[2 * (v - n)]; q.y = o.exy[2 * (v - n) + 1];
    }
    // (integer box test first: with thousands of points almost every candidate is rejected here, without the sqrt)
    const int ix = u.x - q.x, iy = u.y - q.y;
    if (ix > mi || ix < -mi || iy > mi || iy < -mi) continue;
    double dx = (double)ix, dy = (double)iy;
    double d = sqrt(dx * dx + dy * dy);
    if (d > P->max_inter) continue;
    if (v < n) { q.s = t.ps[v]; q.r = t.pr[v]; q.a = t.pa[v]; }
    else { q.s = o.emarks[3 * (v - n)]; q.r = o.emarks[3 * (v - n) + 1]; q.a = o.emarks[3 * (v - n) + 2]; }
    for (int p = 0; p < M.n_pair; ++p)
      if (d <= M.pair[p].max_dist) red[p] = reduce2(M.pair[p].reduce, red[p], pair_value_rects(M.pair[p], u, gu, q, d));
  }
  if (vec_or_null) for (int p = 0; p < M.n_pair; ++p) vec_or_null[M.n_unit + p] = red[p];
  return finish_energy(P, lin + pair_part(P, gate, red[0], red[1]));
}

__global__ void k_point_energies(const DevParams *P, const TileRef *tiles, int tile, double *e_pts, double *vectors) {
  TileRef t = tiles[tile];
  int n = *t.n, i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Overlay none{0, nullptr, 0, nullptr, nullptr};
  int nt = P->model.n_unit + P->model.n_pair;
  double vec[MPP_MAX_UNIT + MPP_MAX_PAIR];
  e_pts[i] = point_energy(P, t, n, tile_rect(t, i), i, -1, none, vec);
  if (vectors) for (int k = 0; k < nt; ++k) vectors[(size_t)i * nt + k] = vec[k];
}

// one workgroup per perturbation:  dE = sum_u [e_u(after)-e_u(before)] + sum e_added - sum e_removed
__global__ __launch_bounds__(256) void k_delta_batch(const DevParams *P, const TileRef *tiles, int tile,
                                                     const int32_t *rem_off, const int32_t *rem,
                                                     const int32_t *add_off, const int32_t *add_xy,
                                                     const double *add_marks, double *dE) {
  __shared__ double part[256];
  TileRef t = tiles[tile];
  const int n = *t.n, cs = blockIdx.x;
  Overlay o;
  o.n_excl = rem_off[cs + 1] - rem_off[cs]; o.excl = rem + rem_off[cs];
  o.n_extra = add_off[cs + 1] - add_off[cs]; o.exy = add_xy + 2 * (size_t)add_off[cs];
  o.emarks = add_marks + 3 * (size_t)add_off[cs];
  Overlay none{0, nullptr, 0, nullptr, nullptr};
  double acc = 0.0;
  for (int i = threadIdx.x; i < n + o.n_extra; i += blockDim.x) {
    if (i < n) {
      Rect u = tile_rect(t, i);
      if (excluded(o, i)) { acc -= point_energy(P, t, n, u, i, -1, none, nullptr); continue; }
      bool touched = false;                      // does any changed point interact with u?
      for (int k = 0; k < o.n_excl && !touched; ++k) {
        double dx = (double)(u.x - t.px[o.excl[k]]), dy = (double)(u.y - t.py[o.excl[k]]);
        touched = sqrt(dx * dx + dy * dy) <= P->max_inter;
      }
      for (int k = 0; k < o.n_extra && !touched; ++k) {
        double dx = (double)(u.x - o.exy[2 * k]), dy = (double)(u.y - o.exy[2 * k + 1]);
        touched = sqrt(dx * dx + dy * dy) <= P->max_inter;
      }
      if (touched) acc += point_energy(P, t, n, u, i, -1, o, nullptr) - point_energy(P, t, n, u, i, -1, none, nullptr);
    } else {
      acc += point_energy(P, t, n, extra_rect(o, i - n), -1, i - n, o, nullptr);
    }
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int s = (int)blockDim.x / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) dE[cs] = part[0];
}

// The same walk, but the per-point energy VECTORS (unit terms, then pair reductions) before and after the
// perturbation are written out instead of being combined: the weight-learning criterion differentiates the
// combinator on them (train_ordering_criterion.py:101-118 with EnergyComputeTorch, :27-40).
// Row cs*stride + i: i < n existing slot i, i >= n the (i-n)-th added rectangle.
// mask: 0 untouched, 1 neighbour of a change (both rows), 2 removed (before only), 3 added (after only).
__global__ __launch_bounds__(256) void k_delta_vectors(const DevParams *P, const TileRef *tiles, int tile,
                                                       const int32_t *rem_off, const int32_t *rem,
                                                       const int32_t *add_off, const int32_t *add_xy,
                                                       const double *add_marks, int stride, double *before,
                                                       double *after, unsigned char *mask) {
  TileRef t = tiles[tile];
  const int n = *t.n, cs = blockIdx.x;
  const int nt = P->model.n_unit + P->model.n_pair;
  Overlay o;
  o.n_excl = rem_off[cs + 1] - rem_off[cs]; o.excl = rem + rem_off[cs];
  o.n_extra = add_off[cs + 1] - add_off[cs]; o.exy = add_xy + 2 * (size_t)add_off[cs];
  o.emarks = add_marks + 3 * (size_t)add_off[cs];
  Overlay none{0, nullptr, 0, nullptr, nullptr};
  for (int i = threadIdx.x; i < stride; i += blockDim.x) {
    const size_t row = (size_t)cs * stride + i;
    double vb[MPP_MAX_UNIT + MPP_MAX_PAIR], va[MPP_MAX_UNIT + MPP_MAX_PAIR];
    for (int k = 0; k < nt; ++k) vb[k] = va[k] = 0.0;
    unsigned char mk = 0;
    if (i < n) {
      Rect u = tile_rect(t, i);
      if (excluded(o, i)) { point_energy(P, t, n, u, i, -1, none, vb); mk = 2; }
      else {
        bool touched = false;
        for (int k = 0; k < o.n_excl && !touched; ++k) {
          double dx = (double)(u.x - t.px[o.excl[k]]), dy = (double)(u.y - t.py[o.excl[k]]);
          touched = sqrt(dx * dx + dy * dy) <= P->max_inter;
        }
        for (int k = 0; k < o.n_extra && !touched; ++k) {
          double dx = (double)(u.x - o.exy[2 * k]), dy = (double)(u.y - o.exy[2 * k + 1]);
          touched = sqrt(dx * dx + dy * dy) <= P->max_inter;
        }
        if (touched) { point_energy(P, t, n, u, i, -1, none, vb); point_energy(P, t, n, u, i, -1, o, va); mk = 1; }
      }
    } else if (i - n < o.n_extra) {
      point_energy(P, t, n, extra_rect(o, i - n), -1, i - n, o, va);
      mk = 3;
    }
    mask[row] = mk;
    for (int k = 0; k < nt; ++k) { before[row * nt + k] = vb[k]; after[row * nt + k] = va[k]; }
  }
}

extern "C" void mpp_launch_delta_vectors(hipStream_t st, const DevParams *P, const TileRef *tiles, int tile,
                                         int n_cases, const int32_t *rem_off, const int32_t *rem,
                                         const int32_t *add_off, const int32_t *add_xy, const double *add_marks,
                                         int stride, double *before, double *after, unsigned char *mask) {
  if (n_cases <= 0 || stride <= 0) return;
  hipLaunchKernelGGL(k_delta_vectors, dim3(n_cases), dim3(256), 0, st, P, tiles, tile, rem_off, rem, add_off, add_xy,
                     add_marks, stride, before, after, mask);
}
extern "C" void mpp_launch_point_energies(hipStream_t st, const DevParams *P, const TileRef *tiles, int tile, int n,
                                          double *e_pts, double *vectors) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_point_energies, dim3((n + 127) / 128), dim3(128), 0, st, P, tiles, tile, e_pts, vectors);
}
extern "C" void mpp_launch_delta_batch(hipStream_t st, const DevParams *P, const TileRef *tiles, int tile,
                                       int n_cases, const int32_t *rem_off, const int32_t *rem,
                                       const int32_t *add_off, const int32_t *add_xy, const double *add_marks,
                                       double *dE) {
  if (n_cases <= 0) return;
  // one wave per perturbation: only the few points near the change do real work, so small workgroups keep more
  // perturbations in flight per CU than 256-thread ones (62 -> see DESIGN.md 6 ms for 5000 single-point removals)
  hipLaunchKernelGGL(k_delta_batch, dim3(n_cases), dim3(64), 0, st, P, tiles, tile, rem_off, rem, add_off, add_xy,
                     add_marks, dE);
}
