// mpp_sampler.hip -- the RJMCMC chain of one tile, resident in one workgroup's LDS.
//
// Restates (not translates) the reference's inner loop, models/mpp/rjmcmc_sampler/rjmcmc.py:83-164,
// with its callees energy_graph.py:139-225 (dE), base_kernels.py / transform_kernels.py /
// shape_samplers.py (proposals and their densities).
//
// Design (MI355X-first):
//  * one workgroup = one tile's chain; the whole interacting point set (marks, cached corner
//    trig, per-point unit energy, per-point pair reductions) and the 32-px spatial hash live in LDS
//    for the entire launch; HBM is touched only for score-map reads of the proposals;
//  * one 64-lane wave evaluates one proposal: lanes take the candidate neighbours gathered from
//    the 3x3 cells around the removed and the added point, compute the pair terms (rectangle
//    clipping, alignment) against them and butterfly-reduce dE; mark rows (32 floats = 128 B) and
//    detection-map windows are read one element per lane, i.e. coalesced;
//  * instead of rebuilding edges twice per step like the reference, each point caches the
//    max/min reduction of its pair energies; a removal that takes away a point's extremum
//    triggers a cooperative re-scan of that point's neighbourhood;
//  * SPEC waves evaluate the next SPEC steps of the SAME chain speculatively against the current
//    state; wave 0 then commits them in order and throws away everything after the first accepted
//    step that could have influenced a later one.  The chain is bit-for-bit the sequential one.
#include "mpp_device.hpp"

#define CAND_MAX 640          // >= 18 cells x cell_cap(<=32) + slack; per-wave candidate list
#define ERR_CELL_OVERFLOW 1
#define ERR_POINT_OVERFLOW 2
#define ERR_BAD_TARGET 3
#define ERR_CAND_OVERFLOW 4

#ifdef MPP_PROFILE
// diagnostic build only: cycles per phase of wave 0, summed over the launch (never in the product build)
__device__ unsigned long long g_prof[16];
#define PROF_T0() unsigned long long pt_ = clock64()
#define PROF_ADD(i) do { unsigned long long n_ = clock64(); if (c.wave == 0) prof_[i] += n_ - pt_; pt_ = n_; } while (0)
#else
#define PROF_T0()
#define PROF_ADD(i)
#endif

struct Rec {                  // one speculative step
  int kernel, tidx, tslot, has_rem, has_add, valid;
  int ax, ay, rx, ry, pid, ncls;
  double as, ar, aa, aux0, aux1, u_acc, qf, qb, dE;
};

struct Lds {
  double *s, *r, *a, *ca, *sa, *hl, *hw, *lin, *red0, *red1;
  int *xy;
  unsigned short *order, *cell_items, *cell_cnt, *cand;
  unsigned char *gate;
  Rec *rec;
  int *sh;                    // [0]=n [1]=err [2]=committed
};

__host__ __device__ inline size_t lds_bytes(int cap, int ncell, int cell_cap, int spec) {
  size_t b = 0;
  b += (size_t)10 * cap * sizeof(double);
  b += (size_t)cap * sizeof(int);
  b += (size_t)cap * sizeof(unsigned short);                  // order
  b += (size_t)ncell * cell_cap * sizeof(unsigned short);     // cell items
  b += (size_t)ncell * sizeof(unsigned short);                // cell counts
  b += (size_t)spec * CAND_MAX * sizeof(unsigned short);      // candidate lists
  b += (size_t)cap;                                           // gate
  b = (b + 15) & ~(size_t)15;
  b += (size_t)spec * sizeof(Rec);
  b += 16 * sizeof(int);
  return b + 64;
}

__device__ inline Lds carve(unsigned char *base, int cap, int ncell, int cell_cap, int spec) {
  Lds L;
  double *d = (double *)base;
  L.s = d; d += cap; L.r = d; d += cap; L.a = d; d += cap; L.ca = d; d += cap; L.sa = d; d += cap;
  L.hl = d; d += cap; L.hw = d; d += cap; L.lin = d; d += cap; L.red0 = d; d += cap; L.red1 = d; d += cap;
  L.xy = (int *)d;
  unsigned short *u = (unsigned short *)(L.xy + cap);
  L.order = u; u += cap;
  L.cell_items = u; u += (size_t)ncell * cell_cap;
  L.cell_cnt = u; u += ncell;
  L.cand = u; u += (size_t)spec * CAND_MAX;
  L.gate = (unsigned char *)u;
  size_t off = (size_t)((unsigned char *)u + cap - base);
  off = (off + 15) & ~(size_t)15;
  L.rec = (Rec *)(base + off);
  L.sh = (int *)(base + off + (size_t)spec * sizeof(Rec));
  return L;
}

struct Chain {
  const DevParams *P;
  TileRef t;
  Lds L;
  int lane, wave;
};

__device__ __forceinline__ void wave_lds_fence() {
  // LDS traffic of one wave is executed in order; this only stops the compiler from moving
  // the loads of other lanes' data above the stores that produce them
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int cell_index(const DevParams *P, int x, int y, int *ci, int *cj) {
  int i = (int)floor((double)x / P->res), j = (int)floor((double)y / P->res);
  *ci = i; *cj = j;
  return j + i * P->ny;
}
__device__ __forceinline__ Geo load_geo(const Lds &L, int slot) {
  Geo g;
  int xy = L.xy[slot];
  g.x = xy & 0xffff; g.y = (xy >> 16) & 0xffff;
  g.hl = L.hl[slot]; g.hw = L.hw[slot]; g.ca = L.ca[slot]; g.sa = L.sa[slot];
  return g;
}
__device__ __forceinline__ Rect load_rect(const Lds &L, int slot) {
  Rect q;
  int xy = L.xy[slot];
  q.x = xy & 0xffff; q.y = (xy >> 16) & 0xffff;
  q.s = L.s[slot]; q.r = L.r[slot]; q.a = L.a[slot];
  return q;
}
// is slot u ordered before the rectangle (vx,vy,vs,vr,va)?  marks are read only on coordinate ties
__device__ __forceinline__ bool slot_first(const Lds &L, int u, const Geo &gu, int vx, int vy, double vs, double vr,
                                           double va) {
  if (gu.x != vx) return gu.x < vx;
  if (gu.y != vy) return gu.y < vy;
  return rect_less(gu.x, gu.y, L.s[u], L.r[u], L.a[u], vx, vy, vs, vr, va);
}
__device__ __forceinline__ double pair_value(const mpp_pair_term &pt, const Geo &u, const Geo &v, bool u_first,
                                             double d) {
  switch (pt.kind) {
    case MPP_P_OVERLAP: return overlap_energy(u, v, u_first);
    case MPP_P_ALIGN: return 1.0 - fabs(u.ca * v.ca + u.sa * v.sa) - (pt.p[0] != 0.0 ? 1.0 : 0.0);
    case MPP_P_DIST_LE: return d <= pt.max_dist ? 1.0 : 0.0;
    case MPP_P_DIST_LT: return d < pt.max_dist ? 1.0 : 0.0;
  }
  return 0.0;
}

// reduction of pair term p over the neighbours of slot u, skipping `skip`, optionally including
// an extra rectangle (the proposal's new point).  Whole wave cooperates; result is uniform.
__device__ double rescan_point(const Chain &c, int p, int u, int skip, bool has_add, const Rect &ar, const Geo &ag) {
  const DevParams *P = c.P;
  const Lds &L = c.L;
  const mpp_pair_term &pt = P->model.pair[p];
  Geo gu = load_geo(L, u);
  int ci, cj;
  cell_index(P, gu.x, gu.y, &ci, &cj);
  double acc = 0.0;                       // 0 is neutral for every supported (kind, reduce) pair
  for (int di = -1; di <= 1; ++di)
    for (int dj = -1; dj <= 1; ++dj) {
      int i = ci + di, j = cj + dj;
      if (i < 0 || i >= P->nx || j < 0 || j >= P->ny) continue;
      int cell = j + i * P->ny, cnt = L.cell_cnt[cell];
      for (int e = c.lane; e < cnt; e += WAVE) {
        int w = L.cell_items[cell * P->cell_cap + e];
        if (w == u || w == skip) continue;
        Geo gw = load_geo(L, w);
        double dx = (double)(gu.x - gw.x), dy = (double)(gu.y - gw.y);
        double d = sqrt(dx * dx + dy * dy);
        if (d <= pt.max_dist) {
          bool uf = slot_first(L, u, gu, gw.x, gw.y, L.s[w], L.r[w], L.a[w]);
          acc = reduce2(pt.reduce, acc, pair_value(pt, gu, gw, uf, d));
        }
      }
    }
  if (has_add && c.lane == 0) {
    double dx = (double)(gu.x - ag.x), dy = (double)(gu.y - ag.y);
    double d = sqrt(dx * dx + dy * dy);
    if (d <= pt.max_dist) {
      bool uf = slot_first(L, u, gu, ar.x, ar.y, ar.s, ar.r, ar.a);
      acc = reduce2(pt.reduce, acc, pair_value(pt, gu, ag, uf, d));
    }
  }
  return wave_reduce(pt.reduce, acc);
}

// dE of (remove slot `rem`, add rectangle `ar`) -- energy_graph.py:139-225 -- as
//   sum over neighbours u of [e_u(after) - e_u(before)]  +  e_added - e_removed.
// With APPLY the neighbours' cached reductions are updated in place.
// ra0/ra1: pair reductions of the added point.  Uniform result.
template <bool APPLY>
__device__ double eval_delta(const Chain &c, int rem, bool has_add, const Rect &ar, const Geo &ag, double lin_a,
                             int gate_a, double *ra0_out, double *ra1_out, int *err) {
  const DevParams *P = c.P;
  const Lds &L = c.L;
  const int np = P->model.n_pair;
  const bool has_rem = rem >= 0;
  unsigned short *cand = L.cand + (size_t)c.wave * CAND_MAX;
  Geo gr;
  Rect rr;
  if (has_rem) { gr = load_geo(L, rem); rr = load_rect(L, rem); }
  else { gr = ag; rr = ar; }
  int cir = 0, cjr = 0, cia = 0, cja = 0;
  if (has_rem) cell_index(P, gr.x, gr.y, &cir, &cjr);
  if (has_add) cell_index(P, ag.x, ag.y, &cia, &cja);

  // ---- gather candidate slots from the 3x3 cells around the removed and the added point
  int my_cell = -1;
  if (c.lane < 18) {
    bool second = c.lane >= 9;
    int k = second ? c.lane - 9 : c.lane;
    int i = (second ? cia : cir) + k / 3 - 1, j = (second ? cja : cjr) + k % 3 - 1;
    bool ok = second ? has_add : has_rem;
    if (ok && second && has_rem && abs(i - cir) <= 1 && abs(j - cjr) <= 1) ok = false;   // already listed
    if (ok && i >= 0 && i < P->nx && j >= 0 && j < P->ny) my_cell = j + i * P->ny;
  }
  int my_cnt = my_cell >= 0 ? (int)L.cell_cnt[my_cell] : 0;
  int incl = my_cnt;
#pragma unroll
  for (int o = 1; o < 32; o <<= 1) {
    int tmp = __shfl_up(incl, o, WAVE);
    if (c.lane >= o) incl += tmp;
  }
  int M = __shfl(incl, 31, WAVE);
  if (M > CAND_MAX) { *err = ERR_CAND_OVERFLOW; M = CAND_MAX; }
  int off = incl - my_cnt;
  for (int e = 0; e < my_cnt && off + e < CAND_MAX; ++e) cand[off + e] = L.cell_items[my_cell * P->cell_cap + e];
  wave_lds_fence();

  double sum_de = 0.0, ra[2] = {0.0, 0.0};
  for (int base = 0; base < M; base += WAVE) {
    int j = base + c.lane;
    bool active = j < M;
    int u = active ? (int)cand[j] : 0;
    if (active && u == rem) active = false;
    Geo gu;
    double oldv[2] = {0.0, 0.0}, newv[2] = {0.0, 0.0};
    bool slow[2] = {false, false};
    if (active) {
      gu = load_geo(L, u);
      oldv[0] = L.red0[u]; oldv[1] = L.red1[u];
      double d_r = 0.0, d_a = 0.0;
      if (has_rem) { double dx = (double)(gu.x - gr.x), dy = (double)(gu.y - gr.y); d_r = sqrt(dx * dx + dy * dy); }
      if (has_add) { double dx = (double)(gu.x - ag.x), dy = (double)(gu.y - ag.y); d_a = sqrt(dx * dx + dy * dy); }
      for (int p = 0; p < np; ++p) {
        const mpp_pair_term &pt = P->model.pair[p];
        bool in_r = has_rem && d_r <= pt.max_dist, in_a = has_add && d_a <= pt.max_dist;
        double nv = oldv[p];
        if (in_r && oldv[p] != 0.0) {
          bool uf = slot_first(L, u, gu, rr.x, rr.y, rr.s, rr.r, rr.a);
          double v_r = pair_value(pt, gu, gr, uf, d_r);
          if (v_r == oldv[p]) slow[p] = true;       // the removed point carries u's extremum
        }
        if (in_a) {
          bool uf = slot_first(L, u, gu, ar.x, ar.y, ar.s, ar.r, ar.a);
          double v_a = pair_value(pt, gu, ag, uf, d_a);
          ra[p] = reduce2(pt.reduce, ra[p], v_a);
          nv = reduce2(pt.reduce, nv, v_a);
        }
        newv[p] = nv;
      }
    }
    // cooperative re-scan for the lanes whose extremum goes away
    for (int p = 0; p < np; ++p) {
      unsigned long long mask = __ballot(slow[p]);
      while (mask) {
        int src = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        int ub = __shfl(u, src, WAVE);
        double v = rescan_point(c, p, ub, rem, has_add, ar, ag);
        if (c.lane == src) newv[p] = v;
      }
    }
    if (active) {
      bool changed = false;
      for (int p = 0; p < np; ++p) changed |= (newv[p] != oldv[p]);
      if (changed) {
        double lin = L.lin[u];
        int gt = L.gate[u];
        sum_de += finish_energy(P, lin + pair_part(P, gt, newv[0], newv[1])) -
                  finish_energy(P, lin + pair_part(P, gt, oldv[0], oldv[1]));
        if (APPLY) { L.red0[u] = newv[0]; L.red1[u] = newv[1]; }
      }
    }
  }
  sum_de = wave_sum(sum_de);
  double ra0 = np > 0 ? wave_reduce(P->model.pair[0].reduce, ra[0]) : 0.0;
  double ra1 = np > 1 ? wave_reduce(P->model.pair[1].reduce, ra[1]) : 0.0;
  *ra0_out = ra0; *ra1_out = ra1;
  double dE = sum_de;
  if (has_add) dE += finish_energy(P, lin_a + pair_part(P, gate_a, ra0, ra1));
  if (has_rem) dE -= finish_energy(P, L.lin[rem] + pair_part(P, (int)L.gate[rem], L.red0[rem], L.red1[rem]));
  return dE;
}

// ---- proposal densities (shape_samplers.py:103-108, transform_kernels.py:94-99, :203-225) -------
__device__ double wave_incl_scan(double v, int lane) {
#pragma unroll
  for (int o = 1; o < WAVE; o <<= 1) {
    double tmp = __shfl_up(v, o, WAVE);
    if (lane >= o) v += tmp;
  }
  return v;
}
// normalised probability of class `cls` in the 32-bin row, and (optionally) a class drawn with u
__device__ double row_prob(const Chain &c, int k, int x, int y, int cls, double u, int *drawn) {
  const float *row = mark_row(c.P, c.t, k, x, y);
  double v = c.lane < MPP_NCLASS ? (double)row[c.lane] : 0.0;       // one coalesced 128-B read
  double acc = wave_incl_scan(v, c.lane);
  double tot = __shfl(acc, WAVE - 1, WAVE);
  if (drawn) {
    unsigned long long m = __ballot(c.lane < MPP_NCLASS && acc / tot <= u);
    int d = __popcll(m);
    *drawn = d < MPP_NCLASS ? d : MPP_NCLASS - 1;
    cls = *drawn;
  }
  return __shfl(v, cls, WAVE) / tot;
}
__device__ double birth_density(const Chain &c, const Rect &q) {
  const DevParams *P = c.P;
  double d = (double)c.t.det[(size_t)q.x * P->W + q.y] / c.t.rowbase[P->H];
  for (int k = 0; k < 3; ++k) d *= row_prob(c, k, q.x, q.y, value_to_class(P, k, mark_of(q, k)), 0.0, nullptr);
  return d * ((double)P->H * (double)P->W * 32768.0);
}
// window of the data-driven translation around (x,y): sum of det, optional draw of an element
__device__ double window_sum(const Chain &c, int x, int y, bool draw, double u, int *ex, int *ey) {
  const DevParams *P = c.P;
  int md = P->kern.max_delta;
  int x0 = max(0, x - md), x1 = min(x + md + 1, P->H), y0 = max(0, y - md), y1 = min(y + md + 1, P->W);
  int wc = y1 - y0, cnt = (x1 - x0) * wc;
  int per = (cnt + WAVE - 1) / WAVE;                      // consecutive elements per lane (row-major order)
  double loc[8];
  double run = 0.0;
  for (int i = 0; i < per && i < 8; ++i) {
    int e = c.lane * per + i;
    double v = e < cnt ? (double)c.t.det[(size_t)(x0 + e / wc) * P->W + (y0 + e % wc)] : 0.0;
    run += v; loc[i] = run;
  }
  double incl = wave_incl_scan(run, c.lane);
  double tot = __shfl(incl, WAVE - 1, WAVE);
  if (draw) {
    double before = incl - run;
    int k = 0;
    for (int i = 0; i < per && i < 8; ++i) {
      int e = c.lane * per + i;
      if (e < cnt && (before + loc[i]) / tot <= u) ++k;
    }
    int e = wave_sum_i(k);
    if (e >= cnt) e = cnt - 1;
    *ex = x0 + e / wc; *ey = y0 + e % wc;
  }
  return tot;
}
__device__ double normal_pdf(double x, double sigma) {
  return exp(-(x * x) / (2.0 * sigma * sigma)) / (sigma * sqrt(MPP_TWO_PI));
}
__device__ void box_muller(uint32_t a, uint32_t b, double *z0, double *z1) {
  double u1 = ((double)a + 1.0) * (1.0 / 4294967296.0), u2 = u32d(b);
  double r = sqrt(-2.0 * log(u1)), th = MPP_TWO_PI * u2;
  *z0 = r * cos(th); *z1 = r * sin(th);
}
__device__ double wrap_mark(const DevParams *P, int k, double v) {
  double lo = P->maps.vmin[k], hi = P->maps.vmax[k];
  if (P->maps.cyclic[k]) {
    double range = hi - lo, m = fmod(v, range);
    if (m < 0) m += range;
    return m + lo;
  }
  return v < lo ? lo : (v > hi ? hi : v);
}

// draw the proposal of a step from its 12 Philox words (the same recipe as the oracle's)
__device__ void draw_proposal(const Chain &c, const uint32_t w[12], int n, Rec &r) {
  const DevParams *P = c.P;
  double uk = u53(w[0], w[1]);
  int k = 0;
  while (k < MPP_NKERNEL - 1 && P->p_cum[k] <= uk) ++k;
  r.kernel = k; r.tidx = -1; r.tslot = -1; r.has_rem = 0; r.has_add = 0; r.pid = -1; r.ncls = -1;
  r.aux0 = r.aux1 = 0.0; r.ax = r.ay = 0; r.as = r.ar = r.aa = 0.0; r.rx = r.ry = 0;
  r.u_acc = u53(w[10], w[11]);
  if (k == MPP_K_UBIRTH) {
    r.has_add = 1;
    r.ax = (int)mulhi32(w[3], (uint32_t)P->H); r.ay = (int)mulhi32(w[4], (uint32_t)P->W);
    r.as = P->maps.vmin[0] + (P->maps.vmax[0] - P->maps.vmin[0]) * u32d(w[5]);
    r.ar = P->maps.vmin[1] + (P->maps.vmax[1] - P->maps.vmin[1]) * u32d(w[6]);
    r.aa = P->maps.vmin[2] + (P->maps.vmax[2] - P->maps.vmin[2]) * u32d(w[7]);
    return;
  }
  if (k == MPP_K_DBIRTH) {
    r.has_add = 1;
    double u = u53(w[3], w[4]), tot = c.t.rowbase[P->H];
    int cnt = 0;
    for (int i = c.lane; i < P->H; i += WAVE) cnt += (c.t.rowbase[i + 1] / tot <= u) ? 1 : 0;
    int row = wave_sum_i(cnt);
    if (row >= P->H) row = P->H - 1;
    double base = c.t.rowbase[row];
    const double *part = c.t.rowpart + (size_t)row * P->W;
    cnt = 0;
    for (int j = c.lane; j < P->W; j += WAVE) cnt += ((base + part[j]) / tot <= u) ? 1 : 0;
    int col = wave_sum_i(cnt);
    if (col >= P->W) col = P->W - 1;
    r.ax = row; r.ay = col;
    int cls;
    row_prob(c, 0, row, col, 0, u32d(w[5]), &cls); r.as = P->maps.edges[0][cls];
    row_prob(c, 1, row, col, 0, u32d(w[6]), &cls); r.ar = P->maps.edges[1][cls];
    row_prob(c, 2, row, col, 0, u32d(w[7]), &cls); r.aa = P->maps.edges[2][cls];
    return;
  }
  if (n == 0) return;
  r.tidx = (int)mulhi32(w[2], (uint32_t)n);
  r.tslot = c.L.order[r.tidx];
  r.has_rem = 1;
  Rect q = load_rect(c.L, r.tslot);
  r.rx = q.x; r.ry = q.y;
  if (k == MPP_K_UDEATH || k == MPP_K_DDEATH) return;
  r.has_add = 1;
  if (k == MPP_K_GTRANS) {
    double z0, z1;
    box_muller(w[3], w[4], &z0, &z1);
    double d0 = P->kern.sigma_trans * z0, d1 = P->kern.sigma_trans * z1;
    int nx = (int)((double)q.x + d0), ny = (int)((double)q.y + d1);
    q.x = min(max(nx, 0), P->H - 1); q.y = min(max(ny, 0), P->W - 1);
    r.aux0 = d0; r.aux1 = d1;
  } else if (k == MPP_K_DTRANS) {
    int ex, ey;
    window_sum(c, q.x, q.y, true, u53(w[3], w[4]), &ex, &ey);
    q.x = ex; q.y = ey;
  } else if (k == MPP_K_GTRANSF) {
    int pid = (int)mulhi32(w[3], 3u);
    double z0, z1;
    box_muller(w[4], w[5], &z0, &z1);
    double d = P->kern.sigma_transform * (P->maps.vmax[pid] - P->maps.vmin[pid]) * z0;
    set_mark(q, pid, wrap_mark(P, pid, mark_of(q, pid) + d));
    r.pid = pid; r.aux0 = d;
  } else {
    int pid = (int)mulhi32(w[3], 3u), cls;
    row_prob(c, pid, q.x, q.y, 0, u32d(w[4]), &cls);
    set_mark(q, pid, P->maps.edges[pid][cls]);
    r.pid = pid; r.ncls = cls;
  }
  r.ax = q.x; r.ay = q.y; r.as = q.s; r.ar = q.r; r.aa = q.a;
}

// n-independent parts of the forward / backward proposal probabilities
__device__ void proposal_densities(const Chain &c, Rec &r) {
  const DevParams *P = c.P;
  r.qf = 1.0; r.qb = 1.0;
  Rect add{r.ax, r.ay, r.as, r.ar, r.aa};
  switch (r.kernel) {
    case MPP_K_DBIRTH: r.qf = birth_density(c, add); break;
    case MPP_K_DDEATH:
      if (r.has_rem) r.qb = birth_density(c, load_rect(c.L, r.tslot));
      break;
    case MPP_K_GTRANS:
      if (r.has_rem) r.qf = r.qb = normal_pdf(r.aux0, P->kern.sigma_trans) * normal_pdf(r.aux1, P->kern.sigma_trans);
      break;
    case MPP_K_DTRANS:
      if (r.has_rem) {
        r.qf = (double)c.t.det[(size_t)r.ax * P->W + r.ay] / window_sum(c, r.rx, r.ry, false, 0.0, nullptr, nullptr);
        r.qb = (double)c.t.det[(size_t)r.rx * P->W + r.ry] / window_sum(c, r.ax, r.ay, false, 0.0, nullptr, nullptr);
      }
      break;
    case MPP_K_GTRANSF:
      if (r.has_rem)
        r.qf = r.qb = normal_pdf(r.aux0, P->kern.sigma_transform * (P->maps.vmax[r.pid] - P->maps.vmin[r.pid]));
      break;
    case MPP_K_DTRANSF:
      if (r.has_rem) {
        Rect old = load_rect(c.L, r.tslot);
        r.qf = row_prob(c, r.pid, old.x, old.y, r.ncls, 0.0, nullptr);
        r.qb = row_prob(c, r.pid, old.x, old.y, value_to_class(P, r.pid, mark_of(old, r.pid)), 0.0, nullptr);
      }
      break;
    default: break;
  }
}

// base_kernels.py:55-64,100-115 ; transform_kernels.py forward/backward_probability
__device__ __forceinline__ void green_terms(const DevParams *P, const Rec &r, int n, double intensity, double *fwd,
                                            double *bwd) {
  const double *pk = P->kern.p_kernel;
  int k = r.kernel;
  if (k == MPP_K_UBIRTH || k == MPP_K_DBIRTH) {
    *fwd = pk[k] * r.qf / intensity; *bwd = pk[k + 1] / (double)(n + 1);
  } else if (!r.has_rem) {
    *fwd = pk[k]; *bwd = pk[k];
  } else if (k == MPP_K_UDEATH || k == MPP_K_DDEATH) {
    *fwd = pk[k] / (double)n; *bwd = pk[k - 1] * r.qb / intensity;
  } else {
    *fwd = pk[k] * r.qf / (double)n; *bwd = pk[k] * r.qb / (double)n;
  }
}

// ---- state mutation (energy_point_set.py:118-154), wave 0 only ----------------------------------
__device__ void cell_remove(const Chain &c, int cell, int slot) {
  const Lds &L = c.L;
  int cnt = L.cell_cnt[cell];
  unsigned short *it = L.cell_items + (size_t)cell * c.P->cell_cap;
  unsigned long long m = __ballot(c.lane < cnt && it[c.lane] == slot);
  wave_lds_fence();
  if (m && c.lane == 0) {
    int idx = __ffsll((long long)m) - 1;
    it[idx] = it[cnt - 1];
    L.cell_cnt[cell] = (unsigned short)(cnt - 1);
  }
  wave_lds_fence();
}
__device__ void cell_insert(const Chain &c, int cell, int slot, int *err) {
  const Lds &L = c.L;
  int cnt = L.cell_cnt[cell];
  if (cnt >= c.P->cell_cap) { *err = ERR_CELL_OVERFLOW; return; }
  if (c.lane == 0) {
    L.cell_items[(size_t)cell * c.P->cell_cap + cnt] = (unsigned short)slot;
    L.cell_cnt[cell] = (unsigned short)(cnt + 1);
  }
  wave_lds_fence();
}
__device__ void write_slot(const Chain &c, int slot, const Rect &q, const Geo &g, double lin, int gate, double r0,
                           double r1) {
  const Lds &L = c.L;
  if (c.lane == 0) {
    L.xy[slot] = (q.x & 0xffff) | (q.y << 16);
    L.s[slot] = q.s; L.r[slot] = q.r; L.a[slot] = q.a;
    L.ca[slot] = g.ca; L.sa[slot] = g.sa; L.hl[slot] = g.hl; L.hw[slot] = g.hw;
    L.lin[slot] = lin; L.gate[slot] = (unsigned char)gate; L.red0[slot] = r0; L.red1[slot] = r1;
  }
}

template <int SPEC>
__global__ __launch_bounds__(WAVE *SPEC) void mpp_chain_kernel(const DevParams *P, const TileRef *tiles, int tile0,
                                                                 long long n_steps, unsigned long long seed,
                                                                 unsigned int chain0, const mpp_proposal *tape,
                                                                 int trace_tile, mpp_step_out *out,
                                                                 mpp_proposal *props) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int tile = tile0 + blockIdx.x;
  Chain c;
  c.P = P; c.t = tiles[tile];
  const int ncell = P->nx * P->ny, cap = P->cap;
  c.L = carve(lds_raw, cap, ncell, P->cell_cap, SPEC);
  c.lane = threadIdx.x & (WAVE - 1);
  c.wave = threadIdx.x / WAVE;
  const Lds &L = c.L;
  const int tid = threadIdx.x, nthr = WAVE * SPEC;
  const bool tracing = (out != nullptr || props != nullptr) && tile == trace_tile;

  // ---------------------------------------------------------------- load the configuration
  int n0 = *c.t.n;
  int err = *c.t.err;
  if (n0 > cap) { n0 = cap; err = ERR_POINT_OVERFLOW; }
  for (int i = tid; i < cap; i += nthr) L.order[i] = (unsigned short)i;
  for (int i = tid; i < ncell; i += nthr) L.cell_cnt[i] = 0;
  for (int i = tid; i < n0; i += nthr) {
    Rect q{c.t.px[i], c.t.py[i], c.t.ps[i], c.t.pr[i], c.t.pa[i]};
    Geo g = make_geo(q);
    double lin; int gate;
    unit_part(P, c.t, q, g, &lin, &gate, nullptr);
    L.xy[i] = (q.x & 0xffff) | (q.y << 16);
    L.s[i] = q.s; L.r[i] = q.r; L.a[i] = q.a; L.ca[i] = g.ca; L.sa[i] = g.sa; L.hl[i] = g.hl; L.hw[i] = g.hw;
    L.lin[i] = lin; L.gate[i] = (unsigned char)gate; L.red0[i] = 0.0; L.red1[i] = 0.0;
  }
  __syncthreads();
  if (tid == 0) {                               // serial: keeps the cell order, hence the result, deterministic
    for (int i = 0; i < n0; ++i) {
      int xy = L.xy[i], ci, cj;
      int cell = cell_index(P, xy & 0xffff, (xy >> 16) & 0xffff, &ci, &cj);
      int cnt = L.cell_cnt[cell];
      if (cnt >= P->cell_cap) { err = ERR_CELL_OVERFLOW; break; }
      L.cell_items[(size_t)cell * P->cell_cap + cnt] = (unsigned short)i;
      L.cell_cnt[cell] = (unsigned short)(cnt + 1);
    }
    L.sh[0] = n0; L.sh[1] = err; L.sh[2] = 0;
  }
  __syncthreads();
  err = L.sh[1];
  {
    Rect dummy{0, 0, 0, 0, 0};
    Geo dg{0, 0, 0, 0, 0, 0};
    for (int u = c.wave; u < n0; u += SPEC)
      for (int p = 0; p < P->model.n_pair; ++p) {
        double v = rescan_point(c, p, u, -1, false, dummy, dg);
        if (c.lane == 0) { if (p == 0) L.red0[u] = v; else L.red1[u] = v; }
      }
  }
  __syncthreads();

  // ---------------------------------------------------------------- the chain
  double T = *c.t.T;                                          // wave 0 keeps the authoritative copy
  const double alpha = c.t.T[1], T_target = c.t.T[2];
  long long step0 = *c.t.step, done = 0;
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);

#ifdef MPP_PROFILE
  unsigned long long prof_[16] = {0};
#endif
  while (done < n_steps && err == 0) {
    const int n = L.sh[0];
    PROF_T0();
    // ---- phase A: wave w evaluates step done+w against the current state
    const long long my = done + c.wave;
    Rec r;
    r.valid = 0;
    if (my < n_steps) {
      r.valid = 1;
      if (tape) {
        const mpp_proposal &tp = tape[my];
        r.kernel = tp.kernel; r.tidx = tp.target; r.tslot = -1; r.has_rem = 0; r.has_add = 0;
        r.ax = tp.ax; r.ay = tp.ay; r.as = tp.as; r.ar = tp.ar; r.aa = tp.aa; r.aux0 = tp.aux0; r.aux1 = tp.aux1;
        r.pid = tp.param_id; r.ncls = tp.new_class; r.u_acc = tp.u_accept; r.rx = r.ry = 0;
        bool is_birth = tp.kernel == MPP_K_UBIRTH || tp.kernel == MPP_K_DBIRTH;
        bool is_death = tp.kernel == MPP_K_UDEATH || tp.kernel == MPP_K_DDEATH;
        if (is_birth) r.has_add = 1;
        else if (n > 0 && tp.target >= 0) {
          if (tp.target >= n) { r.valid = 0; r.kernel = -1; }       // reported at commit time
          else {
            r.has_rem = 1; r.has_add = is_death ? 0 : 1;
            r.tslot = L.order[tp.target];
            int xy = L.xy[r.tslot];
            r.rx = xy & 0xffff; r.ry = (xy >> 16) & 0xffff;
          }
        }
      } else {
        uint32_t w[12];
        uint64_t s = (uint64_t)(step0 + my);
        for (uint32_t b = 0; b < 3; ++b)
          philox4x32_10((uint32_t)s, (uint32_t)(s >> 32), b, chain0 + (uint32_t)tile, k0, k1, w + 4 * b);
        draw_proposal(c, w, n, r);
      }
      PROF_ADD(0);
      if (r.valid) {
        if (r.has_add && (r.ax < 0 || r.ax >= P->H || r.ay < 0 || r.ay >= P->W)) { r.valid = 0; r.kernel = -1; }
      }
      if (r.valid) {
        proposal_densities(c, r);
        PROF_ADD(1);
        r.dE = 0.0;
        if (r.has_rem || r.has_add) {
          Rect add{r.ax, r.ay, r.as, r.ar, r.aa};
          Geo ag = make_geo(add);
          double lin_a = 0.0; int gate_a = 1;
          if (r.has_add) unit_part(P, c.t, add, ag, &lin_a, &gate_a, nullptr);
          PROF_ADD(2);
          double ra0, ra1;
          int e2 = 0;
          r.dE = eval_delta<false>(c, r.has_rem ? r.tslot : -1, r.has_add != 0, add, ag, lin_a, gate_a, &ra0, &ra1, &e2);
          if (e2) { r.valid = 0; r.kernel = -2 - e2; }
          PROF_ADD(3);
        }
      }
    }
    if (SPEC > 1) {
      if (c.lane == 0) L.rec[c.wave] = r;
      __syncthreads();
    }
    // ---- phase B: wave 0 commits in order
    if (c.wave == 0) {
      int committed = 0, cur_n = n;
      bool stop = false;
      for (int w = 0; w < SPEC && !stop; ++w) {
        Rec q = (SPEC > 1) ? L.rec[w] : r;
        if (done + w >= n_steps) break;
        if (!q.valid) {
          if (q.kernel == -1) err = ERR_BAD_TARGET;
          else if (q.kernel <= -3) err = -2 - q.kernel;
          // otherwise: invalidated by an earlier accept of this round -> re-evaluate next round
          break;
        }
        double fwd, bwd;
        green_terms(P, q, cur_n, c.t.intensity, &fwd, &bwd);
        double log_alpha = (-q.dE / T) + log(bwd + EPS_GREEN) - log(fwd + EPS_GREEN);
        int accepted = log(q.u_acc + EPS_GREEN) < log_alpha ? 1 : 0;
        PROF_ADD(4);
        if (accepted && (q.has_rem || q.has_add)) {
          Rect add{q.ax, q.ay, q.as, q.ar, q.aa};
          Geo ag = make_geo(add);
          double lin_a = 0.0; int gate_a = 1;
          if (q.has_add) unit_part(P, c.t, add, ag, &lin_a, &gate_a, nullptr);
          double ra0, ra1;
          int e2 = 0;
          eval_delta<true>(c, q.has_rem ? q.tslot : -1, q.has_add != 0, add, ag, lin_a, gate_a, &ra0, &ra1, &e2);
          wave_lds_fence();
          if (q.has_rem && q.has_add) {                      // move / transform: same slot
            int ci, cj;
            int c0 = cell_index(P, q.rx, q.ry, &ci, &cj), c1 = cell_index(P, q.ax, q.ay, &ci, &cj);
            if (c0 != c1) { cell_remove(c, c0, q.tslot); cell_insert(c, c1, q.tslot, &err); }
            write_slot(c, q.tslot, add, ag, lin_a, gate_a, ra0, ra1);
          } else if (q.has_rem) {                            // death: last index takes the hole
            int ci, cj;
            cell_remove(c, cell_index(P, q.rx, q.ry, &ci, &cj), q.tslot);
            if (c.lane == 0) {
              unsigned short last = L.order[cur_n - 1];
              L.order[cur_n - 1] = (unsigned short)q.tslot;
              L.order[q.tidx] = last;
            }
            cur_n -= 1;
          } else {                                           // birth: next free slot
            if (cur_n >= cap) { err = ERR_POINT_OVERFLOW; }
            else {
              int slot = L.order[cur_n], ci, cj;
              cell_insert(c, cell_index(P, q.ax, q.ay, &ci, &cj), slot, &err);
              write_slot(c, slot, add, ag, lin_a, gate_a, ra0, ra1);
              cur_n += 1;
            }
          }
          wave_lds_fence();
          // which later speculative steps are still trustworthy?
          if (SPEC > 1) {
            if (!(q.has_rem && q.has_add)) stop = true;      // n or the index->slot map changed
            else
              for (int w2 = w + 1; w2 < SPEC; ++w2) {
                Rec &o = L.rec[w2];
                if (!o.valid) continue;
                bool bad = o.has_rem && o.tslot == q.tslot;
                const long long D2 = (long long)(4.0 * P->max_inter * P->max_inter) + 1;
                int ox[2] = {o.rx, o.ax}, oy[2] = {o.ry, o.ay}, oh[2] = {o.has_rem, o.has_add};
                int qx[2] = {q.rx, q.ax}, qy[2] = {q.ry, q.ay};
                for (int a = 0; a < 2 && !bad; ++a)
                  for (int b = 0; b < 2 && !bad; ++b)
                    if (oh[a]) {
                      long long dx = ox[a] - qx[b], dy = oy[a] - qy[b];
                      if (dx * dx + dy * dy <= D2) bad = true;
                    }
                if (bad && c.lane == 0) o.valid = 0;
              }
            wave_lds_fence();
          }
        }
        PROF_ADD(5);
        if (tracing && c.lane == 0) {
          long long idx = done + w;
          if (out) {
            mpp_step_out so;
            so.dE = q.dE; so.fwd = fwd; so.bwd = bwd; so.log_alpha = log_alpha; so.T = T; so.accepted = accepted;
            so.n_after = cur_n;
            out[idx] = so;
          }
          if (props) {
            mpp_proposal pp;
            pp.kernel = q.kernel; pp.target = q.has_rem ? q.tidx : -1; pp.ax = q.ax; pp.ay = q.ay; pp.as = q.as;
            pp.ar = q.ar; pp.aa = q.aa; pp.aux0 = q.aux0; pp.aux1 = q.aux1; pp.param_id = q.pid;
            pp.new_class = q.ncls; pp.u_accept = q.u_acc;
            props[idx] = pp;
          }
        }
        if (T > T_target) T *= alpha;                        // rjmcmc.py:158-159
        committed += 1;
        if (err) break;
      }
      if (c.lane == 0) { L.sh[0] = cur_n; L.sh[1] = err; L.sh[2] = committed; }
    }
    __syncthreads();
    err = L.sh[1];
    done += L.sh[2];
    if (SPEC > 1) __syncthreads();
    PROF_ADD(6);
  }
#ifdef MPP_PROFILE
  if (tid == 0) for (int i = 0; i < 16; ++i) g_prof[i] = prof_[i];
#endif

  // ---------------------------------------------------------------- write the configuration back
  const int n_end = L.sh[0];
  for (int i = tid; i < n_end; i += nthr) {
    int slot = L.order[i], xy = L.xy[slot];
    c.t.px[i] = xy & 0xffff; c.t.py[i] = (xy >> 16) & 0xffff;
    c.t.ps[i] = L.s[slot]; c.t.pr[i] = L.r[slot]; c.t.pa[i] = L.a[slot];
  }
  if (tid == 0) {
    *c.t.n = n_end; *c.t.err = err; *c.t.step = step0 + done;
    *c.t.T = T;
  }
}

#ifdef MPP_PROFILE
extern "C" void mpp_debug_read_prof(unsigned long long *out) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * 16);
}
#endif

// ---- host-side launcher ----------------------------------------------------------------------------
extern "C" size_t mpp_chain_lds_bytes(int cap, int ncell, int cell_cap, int spec) {
  return lds_bytes(cap, ncell, cell_cap, spec);
}

template <int SPEC>
static hipError_t launch_spec(hipStream_t st, int grid, size_t lds, const DevParams *P, const TileRef *tiles, int tile0,
                              long long n_steps, unsigned long long seed, unsigned int chain0,
                              const mpp_proposal *tape, int trace_tile, mpp_step_out *out, mpp_proposal *props) {
  hipError_t e = hipFuncSetAttribute((const void *)mpp_chain_kernel<SPEC>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(mpp_chain_kernel<SPEC>, dim3(grid), dim3(WAVE * SPEC), lds, st, P, tiles, tile0, n_steps, seed,
                     chain0, tape, trace_tile, out, props);
  return hipGetLastError();
}

extern "C" hipError_t mpp_launch_chain(hipStream_t st, int spec, int grid, size_t lds, const DevParams *P,
                                       const TileRef *tiles, int tile0, long long n_steps, unsigned long long seed,
                                       unsigned int chain0, const mpp_proposal *tape, int trace_tile,
                                       mpp_step_out *out, mpp_proposal *props) {
  switch (spec) {
    case 1: return launch_spec<1>(st, grid, lds, P, tiles, tile0, n_steps, seed, chain0, tape, trace_tile, out, props);
    case 2: return launch_spec<2>(st, grid, lds, P, tiles, tile0, n_steps, seed, chain0, tape, trace_tile, out, props);
    case 4: return launch_spec<4>(st, grid, lds, P, tiles, tile0, n_steps, seed, chain0, tape, trace_tile, out, props);
    case 8: return launch_spec<8>(st, grid, lds, P, tiles, tile0, n_steps, seed, chain0, tape, trace_tile, out, props);
    case 16: return launch_spec<16>(st, grid, lds, P, tiles, tile0, n_steps, seed, chain0, tape, trace_tile, out, props);
  }
  return hipErrorInvalidValue;
}
