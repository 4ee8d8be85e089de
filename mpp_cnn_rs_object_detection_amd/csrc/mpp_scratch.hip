// mpp_scratch.hip -- energies computed from scratch on the configuration in HBM.
//
// These back the EPointsSet facade of the reference (point_set/energy_point_set.py:80-116):
// total_energy(), energy_delta() for perturbations with LISTS of removals/additions (the
// aggregated perturbations of perturbation_sampler.py:176-211) and papangelou().  They are the
// callers' side of the hot path (merge_patches, final scoring, weight learning), not the chain
// itself, so they favour being obviously right: every point's pair reductions are recomputed from
// scratch over all points within range.  They double as the independent check of the chain's cached
// reductions (the reference's debug=True invariants, energy_point_set.py:126-152).
// Candidates come from a scan of the whole configuration, or -- for configurations of GRID_MIN_POINTS points or more
// (a merged image: thousands) -- from the cells around the point of a uniform grid built on the device just before
// the launch (PointsSet.get_potential_neighbors, point_set.py:111-145; the reductions are max / min, so the visiting
// order does not change a bit of the result).
#include "mpp_device.hpp"

struct Grid {                 // CSR over the P->nx x P->ny cells of the tile: items[start[c] .. start[c+1]) = slots in cell c
  const int32_t *start;       // nullptr: no grid, scan all points
  const int32_t *items;
};
__device__ __forceinline__ int grid_coord(const DevParams *P, int x) {
  return P->res_shift >= 0 ? (x >> P->res_shift) : (x / P->res_int);
}

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
                               int self_extra, const Overlay &o, double *vec_or_null, const Grid &g) {
  const mpp_model &M = P->model;
  Geo gu = make_geo(u);
  double lin; int gate;
  unit_part<true>(P, t, &P->maps.edges[0][0], u, gu, &lin, &gate, vec_or_null);
  double red[MPP_MAX_PAIR] = {0.0, 0.0};
  const int mi = (int)ceil(P->max_inter);
  auto visit = [&](int v) {                       // v < n: slot of the configuration, else added rectangle v - n
    Rect q;
    if (v < n) {
      if (v == self_slot || excluded(o, v)) return;
      q.x = t.px[v]; q.y = t.py[v];
    } else {
      if (v - n == self_extra) return;
      q.x = o.exy[2 * (v - n)]; q.y = o.exy[2 * (v - n) + 1];
    }
    // (integer box test first: with thousands of points almost every candidate is rejected here, without the sqrt)
    const int ix = u.x - q.x, iy = u.y - q.y;
    if (ix > mi || ix < -mi || iy > mi || iy < -mi) return;
    double dx = (double)ix, dy = (double)iy;
    double d = sqrt(dx * dx + dy * dy);
    if (d > P->max_inter) return;
    if (v < n) { q.s = t.ps[v]; q.r = t.pr[v]; q.a = t.pa[v]; }
    else { q.s = o.emarks[3 * (v - n)]; q.r = o.emarks[3 * (v - n) + 1]; q.a = o.emarks[3 * (v - n) + 2]; }
    for (int p = 0; p < M.n_pair; ++p)
      if (d <= M.pair[p].max_dist) red[p] = reduce2(M.pair[p].reduce, red[p], pair_value_rects(M.pair[p], u, gu, q, d));
  };
  if (g.start) {
    const int r = (int)ceil(P->max_inter / P->res);
    const int ci = grid_coord(P, u.x), cj = grid_coord(P, u.y);
    for (int i = max(ci - r, 0); i <= min(ci + r, P->nx - 1); ++i)
      for (int j = max(cj - r, 0); j <= min(cj + r, P->ny - 1); ++j) {
        const int c = j + i * P->ny;
        for (int k = g.start[c]; k < g.start[c + 1]; ++k) visit(g.items[k]);
      }
    for (int e = 0; e < o.n_extra; ++e) visit(n + e);
  } else {
    for (int v = 0; v < n + o.n_extra; ++v) visit(v);
  }
  if (vec_or_null) for (int p = 0; p < M.n_pair; ++p) vec_or_null[M.n_unit + p] = red[p];
  return finish_energy(P, lin + pair_part(P, gate, red[0], red[1]));
}

// ---- the grid: count, scan, fill, sort each cell (deterministic order) ------------------------------------------------
// (blockIdx.y: the launch builds the grids of gridDim.y consecutive tiles, `sstride` / `istride` entries apart -- 1 tile: 0)
__global__ void k_grid_count(const DevParams *P, const TileRef *tiles, int tile, int32_t *start, int sstride) {
  tile += blockIdx.y; start += (size_t)blockIdx.y * sstride;
  TileRef t = tiles[tile];
  const int n = *t.n, i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicAdd(&start[1 + grid_coord(P, t.py[i]) + grid_coord(P, t.px[i]) * P->ny], 1);
}
__global__ __launch_bounds__(1024) void k_grid_scan(int32_t *start, int len, int sstride) {       // inclusive scan in place, one workgroup
  __shared__ int part[1024];
  start += (size_t)blockIdx.y * sstride;
  const int per = (len + 1023) / 1024, lo = threadIdx.x * per, hi = min(lo + per, len);
  int s = 0;
  for (int i = lo; i < hi; ++i) s += start[i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    int v = (int)threadIdx.x >= o ? part[threadIdx.x - o] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int run = threadIdx.x ? part[threadIdx.x - 1] : 0;
  for (int i = lo; i < hi; ++i) { run += start[i]; start[i] = run; }
}
__global__ void k_grid_fill(const DevParams *P, const TileRef *tiles, int tile, const int32_t *start, int32_t *cursor,
                            int32_t *items, int sstride, int istride) {
  tile += blockIdx.y; start += (size_t)blockIdx.y * sstride; cursor += (size_t)blockIdx.y * (sstride > 0 ? sstride - 1 : 0);
  items += (size_t)blockIdx.y * istride;
  TileRef t = tiles[tile];
  const int n = *t.n, i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = grid_coord(P, t.py[i]) + grid_coord(P, t.px[i]) * P->ny;
  items[start[c] + atomicAdd(&cursor[c], 1)] = i;
}
__global__ void k_grid_sort(const int32_t *start, int32_t *items, int ncell, int sstride, int istride) {
  start += (size_t)blockIdx.y * sstride; items += (size_t)blockIdx.y * istride;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncell) return;
  const int lo = start[c], hi = start[c + 1];
  for (int i = lo + 1; i < hi; ++i) {
    const int v = items[i];
    int k = i - 1;
    while (k >= lo && items[k] > v) { items[k + 1] = items[k]; --k; }
    items[k + 1] = v;
  }
}
// start: [ncell + 1], cursor: [ncell], items: [n]
extern "C" void mpp_launch_grid_build(hipStream_t st, const DevParams *P, const TileRef *tiles, int tile, int n, int ncell,
                                      int32_t *start, int32_t *cursor, int32_t *items) {
  (void)hipMemsetAsync(start, 0, sizeof(int32_t) * ((size_t)ncell + 1), st);
  (void)hipMemsetAsync(cursor, 0, sizeof(int32_t) * (size_t)ncell, st);
  const unsigned gb = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(k_grid_count, dim3(gb), dim3(256), 0, st, P, tiles, tile, start, 0);
  hipLaunchKernelGGL(k_grid_scan, dim3(1), dim3(1024), 0, st, start, ncell + 1, 0);
  hipLaunchKernelGGL(k_grid_fill, dim3(gb), dim3(256), 0, st, P, tiles, tile, (const int32_t *)start, cursor, items, 0, 0);
  hipLaunchKernelGGL(k_grid_sort, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, st, (const int32_t *)start, items, ncell, 0, 0);
}
// the grids of tiles 0 .. n_tiles - 1 at once: start [n_tiles][ncell + 1], cursor [n_tiles][ncell], items [n_tiles][cap]
extern "C" void mpp_launch_grid_build_all(hipStream_t st, const DevParams *P, const TileRef *tiles, int n_tiles, int max_n, int ncell,
                                          int cap, int32_t *start, int32_t *cursor, int32_t *items) {
  (void)hipMemsetAsync(start, 0, sizeof(int32_t) * ((size_t)ncell + 1) * n_tiles, st);
  (void)hipMemsetAsync(cursor, 0, sizeof(int32_t) * (size_t)ncell * n_tiles, st);
  const unsigned gb = (unsigned)((max_n + 255) / 256);
  hipLaunchKernelGGL(k_grid_count, dim3(gb, n_tiles), dim3(256), 0, st, P, tiles, 0, start, ncell + 1);
  hipLaunchKernelGGL(k_grid_scan, dim3(1, n_tiles), dim3(1024), 0, st, start, ncell + 1, ncell + 1);
  hipLaunchKernelGGL(k_grid_fill, dim3(gb, n_tiles), dim3(256), 0, st, P, tiles, 0, (const int32_t *)start, cursor, items, ncell + 1, cap);
  hipLaunchKernelGGL(k_grid_sort, dim3((unsigned)((ncell + 255) / 256), n_tiles), dim3(256), 0, st, (const int32_t *)start, items, ncell,
                     ncell + 1, cap);
}

__global__ void k_point_energies(const DevParams *P, const TileRef *tiles, int tile, double *e_pts, double *vectors, Grid g) {
  TileRef t = tiles[tile];
  int n = *t.n, i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Overlay none{0, nullptr, 0, nullptr, nullptr};
  int nt = P->model.n_unit + P->model.n_pair;
  double vec[MPP_MAX_UNIT + MPP_MAX_PAIR];
  e_pts[i] = point_energy(P, t, n, tile_rect(t, i), i, -1, none, vec, g);
  if (vectors) for (int k = 0; k < nt; ++k) vectors[(size_t)i * nt + k] = vec[k];
}

// one workgroup per perturbation:  dE = sum_u [e_u(after)-e_u(before)] + sum e_added - sum e_removed
__global__ __launch_bounds__(256) void k_delta_batch(const DevParams *P, const TileRef *tiles, int tile,
                                                     const int32_t *rem_off, const int32_t *rem,
                                                     const int32_t *add_off, const int32_t *add_xy,
                                                     const double *add_marks, double *dE, Grid g) {
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
      if (excluded(o, i)) { acc -= point_energy(P, t, n, u, i, -1, none, nullptr, g); continue; }
      bool touched = false;                      // does any changed point interact with u?
      for (int k = 0; k < o.n_excl && !touched; ++k) {
        double dx = (double)(u.x - t.px[o.excl[k]]), dy = (double)(u.y - t.py[o.excl[k]]);
        touched = sqrt(dx * dx + dy * dy) <= P->max_inter;
      }
      for (int k = 0; k < o.n_extra && !touched; ++k) {
        double dx = (double)(u.x - o.exy[2 * k]), dy = (double)(u.y - o.exy[2 * k + 1]);
        touched = sqrt(dx * dx + dy * dy) <= P->max_inter;
      }
      if (touched) acc += point_energy(P, t, n, u, i, -1, o, nullptr, g) - point_energy(P, t, n, u, i, -1, none, nullptr, g);
    } else {
      acc += point_energy(P, t, n, extra_rect(o, i - n), -1, i - n, o, nullptr, g);
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

// ---- merge_patches(method='distance') on the device, for every tile of a ctx at once ------------------------------------
// (data_loaders.py:122-161: the aggregated detections of an image are scored -- Papangelou intensity of each point in the
// full configuration --, points closer than `distance` keep only the best of them, the survivors are scored again.  One
// "tile" of the ctx = one image's aggregated configuration on the image's score maps; a dataset batch = many tiles.)

// Papangelou of every point: dE[tile][i] = E(with u_i) - E(without u_i), i.e. minus the delta of removing it.  One wave per
// point, the body of k_delta_batch with the one removal (same lanes, same reduction tree: the values of mpp_papangelou).
// `g0`: the candidate grids of the tiles, `sstride` / `istride` entries apart (without them every point_energy below scans
// all points of its tile: 8 of the 17 ms a batch of 28 images took), or {nullptr, nullptr}.
__global__ __launch_bounds__(64) void k_papangelou_tiles(const DevParams *P, const TileRef *tiles, int cap, double *dE, Grid g0,
                                                         int sstride, int istride) {
  __shared__ double part[64];
  __shared__ int32_t ex;
  const int tile = blockIdx.y, cs = blockIdx.x;
  TileRef t = tiles[tile];
  const int n = *t.n;
  if (cs >= n) return;
  const Grid g{g0.start ? g0.start + (size_t)tile * sstride : nullptr, g0.items ? g0.items + (size_t)tile * istride : nullptr};
  if (threadIdx.x == 0) ex = cs;
  __syncthreads();
  Overlay o{1, &ex, 0, nullptr, nullptr};
  Overlay none{0, nullptr, 0, nullptr, nullptr};
  // The points whose energy the removal changes -- u_cs itself and its neighbours within max_inter -- are listed first
  // (ascending index), then every lane evaluates ONE of them: a dozen from-scratch energies side by side instead of one or
  // two lanes working through them in seven turns.  The sum is then formed exactly as the plain loop below forms it (lane l
  // adds the points l, l + 64, ... in that order, then the tree), so the values are those of k_delta_batch bit for bit.
  constexpr int LIST = 256;
  __shared__ int32_t lst[LIST];
  __shared__ double val[LIST];
  const int lane = threadIdx.x;
  const unsigned long long below = (1ull << lane) - 1ull;
  int cnt = 0;
  for (int b = 0; b < n; b += 64) {
    const int i = b + lane;
    bool in = false;
    if (i < n) {
      const double dx = (double)(t.px[i] - t.px[cs]), dy = (double)(t.py[i] - t.py[cs]);
      in = i == cs || sqrt(dx * dx + dy * dy) <= P->max_inter;
    }
    const unsigned long long m = __ballot(in);
    const int pos = cnt + __popcll(m & below);
    if (in && pos < LIST) lst[pos] = i;
    cnt += __popcll(m);
  }
  __syncthreads();
  double acc = 0.0;
  if (cnt <= LIST) {
    for (int k0 = 0; k0 < cnt; k0 += 64) {
      const int k = k0 + lane;
      if (k < cnt) {
        const int i = lst[k];
        const Rect u = tile_rect(t, i);
        val[k] = i == cs ? -point_energy(P, t, n, u, i, -1, none, nullptr, g)
                         : point_energy(P, t, n, u, i, -1, o, nullptr, g) - point_energy(P, t, n, u, i, -1, none, nullptr, g);
      }
    }
    __syncthreads();
    for (int k = 0; k < cnt; ++k)
      if ((lst[k] & 63) == lane) acc += val[k];
  } else {                                              // (more than the list holds: the plain loop)
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      Rect u = tile_rect(t, i);
      if (i == cs) { acc -= point_energy(P, t, n, u, i, -1, none, nullptr, g); continue; }
      double dx = (double)(u.x - t.px[cs]), dy = (double)(u.y - t.py[cs]);
      if (sqrt(dx * dx + dy * dy) <= P->max_inter)
        acc += point_energy(P, t, n, u, i, -1, o, nullptr, g) - point_energy(P, t, n, u, i, -1, none, nullptr, g);
    }
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int s = (int)blockDim.x / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) dE[(size_t)tile * cap + cs] = -part[0];
}

__device__ __forceinline__ double wave_max_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const double w = __shfl_xor(v, o, WAVE); v = w > v ? w : v; }
  return v;
}
__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const int w = __shfl_xor(v, o, WAVE); v = w < v ? w : v; }
  return v;
}
// flags[tile][i] = point i has another point within the distance (the others are their own best and never removed: the walk
// below only visits the flagged ones -- 5 % of the points of a 5 000-point image).  One lane per point.
__global__ __launch_bounds__(64) void k_has_neighbour(const TileRef *tiles, int cap, int dist2, int32_t *flags) {
  const int tile = blockIdx.y, i = blockIdx.x * WAVE + threadIdx.x;
  TileRef t = tiles[tile];
  const int n = *t.n;
  if (i >= n) return;
  const int xi = t.px[i], yi = t.py[i];
  int f = 0;
  for (int j = 0; j < n; ++j) {
    const int dx = t.px[j] - xi, dy = t.py[j] - yi;
    f |= (j != i && dx * dx + dy * dy <= dist2) ? 1 : 0;
  }
  flags[(size_t)tile * cap + i] = f;
}

// The dedupe walk of distance_merge (data_loaders.py:140-159 as restated in this repo's data_loaders.distance_merge): in
// index order, every not-yet-removed point keeps, among the not-yet-removed points within `distance` of it (itself
// included), only the one with the best intensity exp(-dE) -- scores equal to 1e-9 are a tie and the first wins; a
// non-finite best score: numpy's argmax (a NaN first, else the first infinity).  Then the removals in index order, each
// moving the last point into the hole (EPointsSet.remove), which fixes the order of the survivors.  One wave per tile.
// work: [T][cap] the neighbour flags of k_has_neighbour; slot_of, tx .. ta: [T][cap] scratch for the removals (permutation, then copies of the configuration).
__global__ __launch_bounds__(64) void k_dedupe_tiles(const TileRef *tiles, int cap, const double *dE, int dist2, const int32_t *work,
                                                     int32_t *slot_of, int32_t *tx, int32_t *ty, double *ts, double *tr,
                                                     double *ta, int32_t *n_removed) {
  const int tile = blockIdx.x, lane = threadIdx.x;
  TileRef t = tiles[tile];
  const int n = *t.n;
  // LDS: alive[cap] flags, then for the points that have a neighbour (ascending): index, packed position, score exp(-dE) --
  // the walk reads them in every inner loop, and one wave alone hides no HBM latency (17 ms for a batch of 28 images when
  // they sat in global memory)
  extern __shared__ unsigned char dd_lds[];
  unsigned char *alive = dd_lds;
  const int cap8 = (cap + 7) & ~7;
  int32_t *fj = (int32_t *)(dd_lds + cap8), *fxy = fj + cap8;
  double *fsc = (double *)(fxy + cap8);
  int32_t *sl = slot_of + (size_t)tile * cap;
  const double *d = dE + (size_t)tile * cap;
  const int32_t *flags = work + (size_t)tile * cap;
  const unsigned long long below = (1ull << lane) - 1ull;
  int nl = 0;
  for (int b = 0; b < n; b += WAVE) {
    const int j = b + lane;
    if (j < n) alive[j] = 1;
    const bool f = j < n && flags[j] != 0;
    const unsigned long long m = __ballot(f);
    if (f) {
      const int pos = nl + __popcll(m & below);
      fj[pos] = j; fxy[pos] = (t.px[j] & 0xffff) | (t.py[j] << 16); fsc[pos] = exp(-d[j]);
    }
    nl += __popcll(m);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  for (int li = 0; li < nl; ++li) {
    const int i = fj[li];
    if (!alive[i]) continue;                                   // (uniform: every lane reads the same flag)
    const int xyi = fxy[li], xi = xyi & 0xffff, yi = (xyi >> 16) & 0xffff;
    // best score among the alive points within the distance (i itself is one of them)
    double top = -INFINITY;
    int first_nan = 0x7fffffff, cnt = 0;
    for (int lj = lane; lj < nl; lj += WAVE) {
      const int j = fj[lj], xy = fxy[lj];
      const int dx = (xy & 0xffff) - xi, dy = ((xy >> 16) & 0xffff) - yi;
      if (alive[j] && dx * dx + dy * dy <= dist2) {
        const double sc = fsc[lj];
        ++cnt;
        if (sc != sc) first_nan = j < first_nan ? j : first_nan;
        else top = sc > top ? sc : top;
      }
    }
    cnt = wave_sum_i(cnt);
    if (cnt <= 1) continue;                                    // alone by now: its own best, nothing to remove
    top = wave_max_d(top);
    first_nan = wave_min_i(first_nan);
    const bool finite = first_nan == 0x7fffffff && top - top == 0.0;
    const double thr = finite ? top - 1e-9 * fabs(top) : top;
    int best = 0x7fffffff;
    if (first_nan != 0x7fffffff) best = first_nan;             // numpy.argmax: the first NaN
    else {
      for (int lj = lane; lj < nl; lj += WAVE) {
        const int j = fj[lj], xy = fxy[lj];
        const int dx = (xy & 0xffff) - xi, dy = ((xy >> 16) & 0xffff) - yi;
        if (alive[j] && dx * dx + dy * dy <= dist2) {
          const double sc = fsc[lj];
          if (finite ? sc >= thr : sc == top) best = j < best ? j : best;
        }
      }
      best = wave_min_i(best);
    }
    for (int lj = lane; lj < nl; lj += WAVE) {
      const int j = fj[lj], xy = fxy[lj];
      const int dx = (xy & 0xffff) - xi, dy = ((xy >> 16) & 0xffff) - yi;
      if (j != best && alive[j] && dx * dx + dy * dy <= dist2) alive[j] = 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  // the survivors' order: removals in index order, the last point takes the hole
  int32_t *orig_at = tx + (size_t)tile * cap;                  // (reused below: first the permutation, then the copy)
  for (int j = lane; j < n; j += WAVE) {
    sl[j] = j;
    ts[(size_t)tile * cap + j] = t.ps[j]; tr[(size_t)tile * cap + j] = t.pr[j]; ta[(size_t)tile * cap + j] = t.pa[j];
    ty[(size_t)tile * cap + j] = (t.px[j] & 0xffff) | (t.py[j] << 16);
    orig_at[j] = j;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  int m = n;
  if (lane == 0) {
    for (int k = 0; k < n; ++k)
      if (!alive[k]) {
        const int slot = sl[k], last = orig_at[m - 1];
        --m;
        if (last != k) { orig_at[slot] = last; sl[last] = slot; }
      }
    *t.n = m;
    n_removed[tile] = n - m;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  m = __shfl(m, 0, WAVE);
  for (int j = lane; j < m; j += WAVE) {
    const int o = orig_at[j];
    const int xy = ty[(size_t)tile * cap + o];
    t.px[j] = xy & 0xffff; t.py[j] = (xy >> 16) & 0xffff;
    t.ps[j] = ts[(size_t)tile * cap + o]; t.pr[j] = tr[(size_t)tile * cap + o]; t.pa[j] = ta[(size_t)tile * cap + o];
  }
}
extern "C" void mpp_launch_papangelou_tiles(hipStream_t st, const DevParams *P, const TileRef *tiles, int n_tiles, int max_n, int cap,
                                            double *dE, const int32_t *grid_start, const int32_t *grid_items, int sstride, int istride) {
  if (n_tiles <= 0 || max_n <= 0) return;
  hipLaunchKernelGGL(k_papangelou_tiles, dim3(max_n, n_tiles), dim3(64), 0, st, P, tiles, cap, dE, Grid{grid_start, grid_items}, sstride,
                     istride);
}
extern "C" void mpp_launch_dedupe_tiles(hipStream_t st, const TileRef *tiles, int n_tiles, int max_n, int cap, const double *dE, int dist2,
                                        int32_t *work, int32_t *slot_of, int32_t *tx, int32_t *ty, double *ts, double *tr,
                                        double *ta, int32_t *n_removed) {
  if (n_tiles <= 0) return;
  hipLaunchKernelGGL(k_has_neighbour, dim3((max_n + WAVE - 1) / WAVE, n_tiles), dim3(64), 0, st, tiles, cap, dist2, work);
  const size_t lds = (size_t)((cap + 7) & ~7) * 17;          // flags + (index, position, score) of the flagged points
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void *)k_dedupe_tiles, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256); attr_set = true; }
  hipLaunchKernelGGL(k_dedupe_tiles, dim3(n_tiles), dim3(64), lds, st, tiles, cap, dE, dist2, (const int32_t *)work, slot_of, tx, ty,
                     ts, tr, ta, n_removed);
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
                                                       double *after, unsigned char *mask, Grid g) {
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
      if (excluded(o, i)) { point_energy(P, t, n, u, i, -1, none, vb, g); mk = 2; }
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
        if (touched) { point_energy(P, t, n, u, i, -1, none, vb, g); point_energy(P, t, n, u, i, -1, o, va, g); mk = 1; }
      }
    } else if (i - n < o.n_extra) {
      point_energy(P, t, n, extra_rect(o, i - n), -1, i - n, o, va, g);
      mk = 3;
    }
    mask[row] = mk;
    for (int k = 0; k < nt; ++k) { before[row * nt + k] = vb[k]; after[row * nt + k] = va[k]; }
  }
}

extern "C" void mpp_launch_delta_vectors(hipStream_t st, const DevParams *P, const TileRef *tiles, int tile,
                                         int n_cases, const int32_t *rem_off, const int32_t *rem,
                                         const int32_t *add_off, const int32_t *add_xy, const double *add_marks,
                                         int stride, double *before, double *after, unsigned char *mask,
                                         const int32_t *grid_start, const int32_t *grid_items) {
  if (n_cases <= 0 || stride <= 0) return;
  hipLaunchKernelGGL(k_delta_vectors, dim3(n_cases), dim3(256), 0, st, P, tiles, tile, rem_off, rem, add_off, add_xy,
                     add_marks, stride, before, after, mask, Grid{grid_start, grid_items});
}
extern "C" void mpp_launch_point_energies(hipStream_t st, const DevParams *P, const TileRef *tiles, int tile, int n,
                                          double *e_pts, double *vectors, const int32_t *grid_start,
                                          const int32_t *grid_items) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_point_energies, dim3((n + 127) / 128), dim3(128), 0, st, P, tiles, tile, e_pts, vectors,
                     Grid{grid_start, grid_items});
}
extern "C" void mpp_launch_delta_batch(hipStream_t st, const DevParams *P, const TileRef *tiles, int tile,
                                       int n_cases, const int32_t *rem_off, const int32_t *rem,
                                       const int32_t *add_off, const int32_t *add_xy, const double *add_marks,
                                       double *dE, const int32_t *grid_start, const int32_t *grid_items) {
  if (n_cases <= 0) return;
  // one wave per perturbation: only the few points near the change do real work, so small workgroups keep more
  // perturbations in flight per CU than 256-thread ones (62 -> see DESIGN.md 6 ms for 5000 single-point removals)
  hipLaunchKernelGGL(k_delta_batch, dim3(n_cases), dim3(64), 0, st, P, tiles, tile, rem_off, rem, add_off, add_xy,
                     add_marks, dE, Grid{grid_start, grid_items});
}
