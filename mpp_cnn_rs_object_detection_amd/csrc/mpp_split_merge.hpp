// mpp_split_merge.hpp -- the optional split and merge kernels (reference kernels/split_and_merge_kernels.py:14-178,
// enabled by use_split_merge; both shipped configs run without them).
//
// A split replaces one point by two, a merge two points by one.  Instead of a second energy-delta routine for
// two-point changes, such a step is DECOMPOSED into the one-point changes the chain already evaluates
// incrementally, applied one after the other to the LDS state:
//     split  {-p, +a0, +a1}  =  move p -> a0,  then birth of a1
//     merge  {-p0, -p1, +q}  =  move p0 -> q,  then death of p1
// dE is the sum of the two deltas (the energy is a function of the state), the Green ratio uses the reference's
// forward / backward probabilities, and a rejected step is undone by the two inverse changes (the cached max/min
// reductions are exact functions of the configuration, so they come back bit for bit).  Such a step runs alone, in
// an "apply round" of wave 0 (the mechanism of the stash-overflow fallback); kernels built without SM carry none
// of this code.
#pragma once
#include "mpp_chain.hpp"

// ValueMapping.clip (shape_net/mappings.py:52-58)
__device__ inline double sm_clip_mark(const DevParams *P, int k, double v) {
  const double lo = P->maps.vmin[k], hi = P->maps.vmax[k];
  if (P->maps.cyclic[k]) {
    double range = hi - lo, m = fmod(v - lo, range);
    if (m < 0) m += range;
    return m + lo;
  }
  return v < lo ? lo : (v > hi ? hi : v);
}
__device__ __forceinline__ int sm_clip_int(double v, int hi) { return (int)(v < 0.0 ? 0.0 : (v > (double)hi ? (double)hi : v)); }

// split_and_merge_kernels.py:56-73
__device__ inline void sm_split_rects(const DevParams *P, const Rect &p, double pd0, double pd1, double sd0, double sd1,
                                      double sd2, Rect *a0, Rect *a1) {
  a0->x = sm_clip_int((double)p.x - pd0, P->H - 1); a0->y = sm_clip_int((double)p.y - pd1, P->W - 1);
  a1->x = sm_clip_int((double)p.x + pd0, P->H - 1); a1->y = sm_clip_int((double)p.y + pd1, P->W - 1);
  a0->s = sm_clip_mark(P, 0, p.s - sd0); a0->r = sm_clip_mark(P, 1, p.r - sd1); a0->a = sm_clip_mark(P, 2, p.a - sd2);
  a1->s = sm_clip_mark(P, 0, p.s + sd0); a1->r = sm_clip_mark(P, 1, p.r + sd1); a1->a = sm_clip_mark(P, 2, p.a + sd2);
}
// split_and_merge_kernels.py:128-135 (the column is clipped with shape[0] upstream; reproduced)
__device__ inline void sm_merge_rect(const DevParams *P, const Rect &p0, const Rect &p1, Rect *q) {
  q->x = sm_clip_int(((double)p0.x + (double)p1.x) / 2.0, P->H - 1);
  q->y = sm_clip_int(((double)p0.y + (double)p1.y) / 2.0, P->H - 1);
  q->s = sm_clip_mark(P, 0, (p0.s + p1.s) / 2.0);
  q->r = sm_clip_mark(P, 1, (p0.r + p1.r) / 2.0);
  q->a = sm_clip_mark(P, 2, (p0.a + p1.a) / 2.0);
}
// SplitSampler.pdf (split_and_merge_kernels.py:33-36)
__device__ inline double sm_split_pdf(const DevParams *P, double sd0, double sd1, double sd2) {
  const double R = P->kern.split_radius, sg = P->kern.split_sigma;
  double p = 1.0 / (MPP_PI * R * R);
  p *= normal_pdf(sd0, sg * (P->maps.vmax[0] - P->maps.vmin[0]));
  p *= normal_pdf(sd1, sg * (P->maps.vmax[1] - P->maps.vmin[1]));
  p *= normal_pdf(sd2, sg * (P->maps.vmax[2] - P->maps.vmin[2]));
  return p;
}
// len(get_potential_neighbors(u, radius)) (point_set.py:111-145): all points of the (2*ceil(r/res)+1)^2 cells
__device__ inline int sm_count_potential(const Chain &c, int x, int y) {
  const DevParams *P = c.P;
  const int off = (int)ceil(P->kern.split_radius / P->res);
  int ci, cj, cnt = 0;
  cell_index(c, x, y, &ci, &cj);
  for (int di = -off; di <= off; ++di)
    for (int dj = -off; dj <= off; ++dj) {
      int i = ci + di, j = cj + dj;
      if (i < 0 || i >= P->nx || j < 0 || j >= P->ny) continue;
      cnt += (int)c.L.cell_cnt[j + i * P->ny];
    }
  return cnt;
}
// get_neighbors(p0, radius) (point_set.py:147-149): the slots within split_radius of slot0 are written to this
// record's stash area; returns their number (> STASH: more than the area holds)
__device__ inline int sm_neighbours(const Chain &c, int ri, int slot0, int x0, int y0) {
  const DevParams *P = c.P;
  const Lds &L = c.L;
  const double R = P->kern.split_radius;
  const int off = (int)ceil(R / P->res);
  const unsigned long long below = (1ull << c.lane) - 1ull;
  int ci, cj, cnt = 0;
  cell_index(c, x0, y0, &ci, &cj);
  for (int di = -off; di <= off; ++di)
    for (int dj = -off; dj <= off; ++dj) {
      int i = ci + di, j = cj + dj;
      if (i < 0 || i >= P->nx || j < 0 || j >= P->ny) continue;
      const int cell = j + i * P->ny, n_c = (int)L.cell_cnt[cell];
      for (int e0 = 0; e0 < n_c; e0 += WAVE) {
        const int e = e0 + c.lane;
        bool ok = false;
        int v = 0;
        if (e < n_c) {
          v = (int)L.cell_items[(size_t)cell * P->cell_cap + e];
          const int vxy = L.xy[v], dx = (vxy & 0xffff) - x0, dy = ((vxy >> 16) & 0xffff) - y0;
          ok = v != slot0 && sqrt((double)(dx * dx + dy * dy)) <= R;
        }
        const unsigned long long m = __ballot(ok);
        const int pos = cnt + __popcll(m & below);
        if (ok && pos < STASH) L.stash_slot[ri * STASH + pos] = (unsigned short)v;
        cnt += __popcll(m);
      }
    }
  wave_lds_fence();
  return cnt;
}
// the j-th of those neighbours in the canonical order of this build (ascending x, y, marks)
__device__ inline int sm_pick(const Chain &c, int ri, int cnt, int j) {
  const Lds &L = c.L;
  const bool mine = c.lane < cnt;
  const int v = mine ? (int)L.stash_slot[ri * STASH + c.lane] : 0;
  const Rect q = load_rect(L, v);
  int rank = 0;
  for (int k = 0; k < cnt; ++k) {
    const Rect o = load_rect(L, (int)L.stash_slot[ri * STASH + k]);
    rank += rect_less(o.x, o.y, o.s, o.r, o.a, q.x, q.y, q.s, q.r, q.a) ? 1 : 0;
  }
  const unsigned long long m = __ballot(mine && rank == j);
  const int src = m ? __ffsll((long long)m) - 1 : 0;
  return __builtin_amdgcn_readlane(v, src);
}
// The same choice when there are more neighbours than the stash area holds (dense, hot configurations; the reference's
// list has no limit): no list at all.  Candidates are enumerated as in sm_neighbours(), 64 at a time, one per lane; every
// lane ranks its candidate against ALL neighbours (the enumeration again, one neighbour at a time for the whole wave:
// broadcast LDS reads), and the candidate of rank j is the one.  O(neighbours^2 / 64): only ever run beyond 32 neighbours.
__device__ inline int sm_pick_streamed(const Chain &c, int slot0, int x0, int y0, int j) {
  const DevParams *P = c.P;
  const Lds &L = c.L;
  const double R = P->kern.split_radius;
  const int off = (int)ceil(R / P->res);
  int ci, cj, first = -1;
  cell_index(c, x0, y0, &ci, &cj);
  for (int di = -off; di <= off; ++di)
    for (int dj = -off; dj <= off; ++dj) {
      const int i = ci + di, jj = cj + dj;
      if (i < 0 || i >= P->nx || jj < 0 || jj >= P->ny) continue;
      const int cell = jj + i * P->ny, n_c = (int)L.cell_cnt[cell];
      for (int e0 = 0; e0 < n_c; e0 += WAVE) {
        const int e = e0 + c.lane;
        bool ok = false;
        int v = 0;
        if (e < n_c) {
          v = (int)L.cell_items[(size_t)cell * P->cell_cap + e];
          const int vxy = L.xy[v], dx = (vxy & 0xffff) - x0, dy = ((vxy >> 16) & 0xffff) - y0;
          ok = v != slot0 && sqrt((double)(dx * dx + dy * dy)) <= R;
        }
        const unsigned long long mk = __ballot(ok);
        if (!mk) continue;
        if (first < 0) first = __builtin_amdgcn_readlane(v, __ffsll((long long)mk) - 1);   // (what sm_pick() falls back to)
        const Rect q = load_rect(L, v);
        int rank = 0;
        for (int ei = -off; ei <= off; ++ei)
          for (int ej = -off; ej <= off; ++ej) {
            const int i2 = ci + ei, j2 = cj + ej;
            if (i2 < 0 || i2 >= P->nx || j2 < 0 || j2 >= P->ny) continue;
            const int cell2 = j2 + i2 * P->ny, n2 = (int)L.cell_cnt[cell2];
            for (int k = 0; k < n2; ++k) {
              const int o_slot = (int)L.cell_items[(size_t)cell2 * P->cell_cap + k];
              const Rect o = load_rect(L, o_slot);
              const int dx = o.x - x0, dy = o.y - y0;
              if (o_slot != slot0 && sqrt((double)(dx * dx + dy * dy)) <= R)
                rank += rect_less(o.x, o.y, o.s, o.r, o.a, q.x, q.y, q.s, q.r, q.a) ? 1 : 0;
            }
          }
        const unsigned long long m = __ballot(ok && rank == j);
        if (m) return __builtin_amdgcn_readlane(v, __ffsll((long long)m) - 1);
      }
    }
  return first;
}
__device__ inline int sm_dense_index(const Chain &c, int n, int slot) {
  int idx = -1;
  for (int i0 = 0; i0 < n; i0 += WAVE) {
    const int i = i0 + c.lane;
    const unsigned long long m = __ballot(i < n && (int)c.L.order[i] == slot);
    if (m) idx = i0 + __ffsll((long long)m) - 1;
  }
  return idx;
}

// the rest of draw_proposal() for the two kernels.  w: the step's first 8 words; further Philox blocks (3..10) feed the
// rejection sampling of the split's position delta (split_and_merge_kernels.py:23-31).
__device__ inline void sm_draw(const Chain &c, Rec &r, int ri, int n, const uint32_t w[8], uint32_t k0, uint32_t k1,
                               uint64_t step, uint32_t chain, int *err) {
  const DevParams *P = c.P;
  // (draw_proposal() has run its last branch for these kernel ids: clear what that wrote)
  r.has_add = 0; r.pid = -1; r.ncls = -1; r.ax = r.ay = 0; r.as = r.ar = r.aa = 0.0; r.aux0 = r.aux1 = 0.0;
  if (r.kernel == MPP_K_MERGE && n < 2) { r.has_rem = 0; r.tidx = -1; r.tslot = -1; return; }   // :121
  if (!r.has_rem) return;                                                                        // n == 0
  if (r.kernel == MPP_K_SPLIT) {
    const double R = P->kern.split_radius;
    double px = 0.0, py = 0.0;
    bool found = false;
    for (uint32_t a = 0; a < 16 && !found; ++a) {
      uint32_t e[4];
      philox4x32_10((uint32_t)step, (uint32_t)(step >> 32), 3u + a / 2u, chain, k0, k1, e);
      const uint32_t ex = (a & 1u) ? e[2] : e[0], ey = (a & 1u) ? e[3] : e[1];
      px = R * u32d(ex); py = R * u32d(ey);
      found = !(sqrt(px * px + py * py) > R);
    }
    double z0, z1, z2, z3;
    box_muller(w[3], w[4], &z0, &z1); box_muller(w[5], w[6], &z2, &z3);
    r.aux0 = px; r.aux1 = py;
    r.as = P->kern.split_sigma * (P->maps.vmax[0] - P->maps.vmin[0]) * z0;
    r.ar = P->kern.split_sigma * (P->maps.vmax[1] - P->maps.vmin[1]) * z1;
    r.aa = P->kern.split_sigma * (P->maps.vmax[2] - P->maps.vmin[2]) * z2;
    return;
  }
  const int cnt = sm_neighbours(c, ri, r.tslot, r.rx, r.ry);
  if (cnt == 0) {                                     // p0 has no neighbour: empty perturbation (:124-126)
    // ... if the search ran on the state this step will really see: record 0 of a round always does.  A later record of
    // the round may have searched around a point that an earlier step of the round moves: it stays a merge with a
    // target, i.e. it asks for an apply round, where it is record 0 and is drawn again on the live state.
    if (ri == 0) r.has_rem = 0;
    return;
  }
  const int pick = (int)mulhi32(w[3], (uint32_t)cnt);
  const int slot1 = cnt <= STASH ? sm_pick(c, ri, cnt, pick) : sm_pick_streamed(c, r.tslot, r.rx, r.ry, pick);
  r.pid = sm_dense_index(c, n, slot1);
}

struct SmSlot { int xy, gate; double s, r, a, ca, sa, hl, hw, rad, lin, red0, red1; };
__device__ inline SmSlot sm_save(const Lds &L, int slot) {
  SmSlot o;
  o.xy = L.xy[slot]; o.gate = (int)L.gate[slot];
  o.s = L.s[slot]; o.r = L.r[slot]; o.a = L.a[slot]; o.ca = L.ca[slot]; o.sa = L.sa[slot]; o.hl = L.hl[slot];
  o.hw = L.hw[slot]; o.rad = L.rad[slot]; o.lin = L.lin[slot]; o.red0 = L.red0[slot]; o.red1 = L.red1[slot];
  return o;
}
__device__ inline void sm_restore(const Chain &c, int slot, const SmSlot &o) {
  const Lds &L = c.L;
  if (c.lane == 0) {
    L.xy[slot] = o.xy; L.gate[slot] = (unsigned char)o.gate;
    L.s[slot] = o.s; L.r[slot] = o.r; L.a[slot] = o.a; L.ca[slot] = o.ca; L.sa[slot] = o.sa; L.hl[slot] = o.hl;
    L.hw[slot] = o.hw; L.rad[slot] = o.rad; L.lin[slot] = o.lin; L.red0[slot] = o.red0; L.red1[slot] = o.red1;
  }
  wave_lds_fence();
}
// put rectangle q into `slot`, as a move of the point living there (moving) or as a new point; neighbours' caches,
// cell lists and the slot are updated, the energy change is returned
__device__ inline double sm_place(const Chain &c, int slot, bool moving, const Rect &q, int *err) {
  const DevParams *P = c.P;
  const Lds &L = c.L;
  Geo2 ag;
  ag.g = make_geo(q);
  ag.rad = geo_radius(ag.g);
  double lin_a, ra0 = 0.0, ra1 = 0.0;
  int gate_a, n_stash = 0;
  unit_part<true>(P, c.t, L.edges, q, ag.g, &lin_a, &gate_a, nullptr, true);     // (the whole wave places one rectangle)
  const double dE = eval_delta(c, 0, moving ? slot : -1, true, q, ag, lin_a, gate_a, &ra0, &ra1, &n_stash, true);
  wave_lds_fence();
  int ci, cj;
  const int c1 = cell_index(c, q.x, q.y, &ci, &cj);
  if (moving) {
    const int oxy = L.xy[slot];
    const int c0 = cell_index(c, oxy & 0xffff, (oxy >> 16) & 0xffff, &ci, &cj);
    if (c0 != c1) { cell_remove(c, c0, slot); cell_insert(c, c1, slot, err); }
  } else {
    cell_insert(c, c1, slot, err);
  }
  if (c.lane == 0) {
    L.xy[slot] = (q.x & 0xffff) | (q.y << 16);
    L.s[slot] = q.s; L.r[slot] = q.r; L.a[slot] = q.a;
    L.ca[slot] = ag.g.ca; L.sa[slot] = ag.g.sa; L.hl[slot] = ag.g.hl; L.hw[slot] = ag.g.hw; L.rad[slot] = ag.rad;
    L.lin[slot] = lin_a; L.gate[slot] = (unsigned char)gate_a; L.red0[slot] = ra0; L.red1[slot] = ra1;
  }
  wave_lds_fence();
  return dE;
}
// take the point of `slot` out of the interaction structure (its slot data stays where it is)
__device__ inline double sm_remove(const Chain &c, int slot) {
  const DevParams *P = c.P;
  const Lds &L = c.L;
  Rect none{0, 0, 0.0, 0.0, 0.0};
  Geo2 ag;
  ag.g.x = ag.g.y = 0; ag.g.hl = ag.g.hw = ag.g.ca = ag.g.sa = 0.0; ag.rad = 0.0;
  double ra0, ra1;
  int n_stash = 0, ci, cj;
  const double dE = eval_delta(c, 0, slot, false, none, ag, 0.0, 1, &ra0, &ra1, &n_stash, true);
  wave_lds_fence();
  const int xy = L.xy[slot];
  cell_remove(c, cell_index(c, xy & 0xffff, (xy >> 16) & 0xffff, &ci, &cj), slot);
  return dE;
}

// One whole split / merge step on the live state (wave 0, apply round): evaluation, accept decision, and either
// the completed change (r._pad = population change, applied) or the exact restoration of the previous state.
__device__ inline void sm_step(const Chain &c, Rec &r, int ri, int n, double T, bool tracing, int *err) {
  const DevParams *P = c.P;
  const Lds &L = c.L;
  const double *pk = P->kern.p_kernel;
  const double intensity = c.t.intensity;
  const int slot0 = r.tslot;
  const Rect p0 = load_rect(L, slot0);
  const SmSlot s0 = sm_save(L, slot0);
  double fwd, bwd, dE;
  r._pad = 0; r.n_stash = 0;
  if (r.kernel == MPP_K_SPLIT) {                      // split_and_merge_kernels.py:79-106
    if (n >= P->cap) { *err = ERR_POINT_OVERFLOW; r.accepted = 0; return; }
    Rect a0, a1;
    sm_split_rects(P, p0, r.aux0, r.aux1, r.as, r.ar, r.aa, &a0, &a1);
    {                         // room in the two target cells?  checked before anything changes, so that a chain that
      int ci, cj;             // stops here can be continued with a larger cell capacity (mpp_api.hip: run_chain)
      const int cp = cell_index(c, p0.x, p0.y, &ci, &cj), c0 = cell_index(c, a0.x, a0.y, &ci, &cj),
                c1 = cell_index(c, a1.x, a1.y, &ci, &cj);
      const int need0 = c0 != cp ? 1 : 0, need1 = 1 + (c1 == c0 ? need0 : 0);
      if ((int)L.cell_cnt[c0] + need0 > c.h.cell_cap || (int)L.cell_cnt[c1] + need1 > c.h.cell_cap) {
        *err = ERR_CELL_OVERFLOW; r.accepted = 0; return;
      }
    }
    const int nn0 = sm_count_potential(c, a0.x, a0.y) + 1, nn1 = sm_count_potential(c, a1.x, a1.y) + 1;
    const double nb = (double)(n + 1);
    fwd = pk[MPP_K_SPLIT] * ((1.0 / (double)n) * sm_split_pdf(P, r.as, r.ar, r.aa)) / intensity;
    bwd = pk[MPP_K_MERGE] * ((1.0 / nb) * (1.0 / (double)nn0) + (1.0 / nb) * (1.0 / (double)nn1));
    const int slot1 = (int)L.order[n];
    dE = sm_place(c, slot0, true, a0, err);
    dE += sm_place(c, slot1, false, a1, err);
    const double ratio = (bwd + EPS_GREEN) / (fwd + EPS_GREEN);
    r.accepted = (*err == 0 && (P->force_accept || (r.u_acc + EPS_GREEN) < exp(-dE / T) * ratio)) ? 1 : 0;
    if (r.accepted) r._pad = 1;
    else {
      sm_remove(c, slot1);
      sm_place(c, slot0, true, p0, err);
      sm_restore(c, slot0, s0);
    }
  } else {                                            // merge, split_and_merge_kernels.py:139-170
    const int idx1 = r.pid, slot1 = (int)L.order[idx1];
    const Rect p1 = load_rect(L, slot1);
    const SmSlot s1 = sm_save(L, slot1);
    int n_nb = sm_neighbours(c, ri, slot0, p0.x, p0.y);
    if (n_nb < 1) n_nb = 1;                           // (a replayed pair further apart than the radius)
    Rect q;
    sm_merge_rect(P, p0, p1, &q);
    {
      int ci, cj;
      const int cp = cell_index(c, p0.x, p0.y, &ci, &cj), cq = cell_index(c, q.x, q.y, &ci, &cj);
      if (cq != cp && (int)L.cell_cnt[cq] + 1 > c.h.cell_cap) { *err = ERR_CELL_OVERFLOW; r.accepted = 0; return; }
    }
    fwd = pk[MPP_K_MERGE] * ((1.0 / (double)n) * (1.0 / (double)n_nb));
    bwd = pk[MPP_K_SPLIT] * ((1.0 / (double)(n - 1)) *
                             sm_split_pdf(P, (p0.s - p1.s) / 2.0, (p0.r - p1.r) / 2.0, (p0.a - p1.a) / 2.0)) / intensity;
    dE = sm_place(c, slot0, true, q, err);
    dE += sm_remove(c, slot1);
    const double ratio = (bwd + EPS_GREEN) / (fwd + EPS_GREEN);
    r.accepted = (*err == 0 && (P->force_accept || (r.u_acc + EPS_GREEN) < exp(-dE / T) * ratio)) ? 1 : 0;
    if (r.accepted) {
      if (c.lane == 0) {                              // the last index takes p1's place (as for a death)
        const unsigned short last = L.order[n - 1];
        L.order[n - 1] = (unsigned short)slot1;
        L.order[idx1] = last;
      }
      wave_lds_fence();
      r._pad = -1;
    } else {
      sm_place(c, slot1, false, p1, err);              // (its caches are rebuilt against q, then follow q -> p0)
      sm_place(c, slot0, true, p0, err);
      sm_restore(c, slot1, s1);
      sm_restore(c, slot0, s0);
    }
  }
  r.dE = dE;
  if (tracing) { r.fwd = fwd; r.bwd = bwd; r.log_alpha = (-dE / T) + log(bwd + EPS_GREEN) - log(fwd + EPS_GREEN); }
}
