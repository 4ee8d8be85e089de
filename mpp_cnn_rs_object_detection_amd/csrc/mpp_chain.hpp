// mpp_chain.hpp -- the RJMCMC chain of one tile, resident in one workgroup's LDS (device code shared by the
// instantiations of mpp_sampler.hip).
//
// Restates (not translates) the reference's inner loop, models/mpp/rjmcmc_sampler/rjmcmc.py:83-164,
// with its callees energy_graph.py:139-225 (dE), base_kernels.py / transform_kernels.py /
// shape_samplers.py (proposals and their densities).
//
// Design (MI355X-first):
//  * one workgroup = one tile's chain; the whole interacting point set (marks, cached corner
//    trig, circumradius, per-point unit energy, per-point pair reductions) and the 32-px spatial
//    hash live in LDS for the entire launch; HBM is touched only for score-map reads of proposals;
//  * one 64-lane wave evaluates one proposal: lanes take the candidate neighbours found in the
//    3x3 cells around the removed and the added point and compute the pair terms (rectangle
//    clipping, alignment) against them; only the few lanes whose neighbour actually changes feed
//    the dE sum (ballot + readlane, no 64-wide butterflies);
//  * instead of rebuilding edges twice per step like the reference, each point caches the
//    max/min reduction of its pair energies; a removal that takes away a point's extremum
//    is answered by the added value when that is at least as extreme, otherwise by a re-scan of that point's
//    neighbourhood in its lane; overlap clips run in uniform control flow with the whole wave on one polygon pair;
//  * a single dependent chain of float64 transcendentals is what a step costs, so their number is
//    kept minimal (one exp for the accept test, trig reused when the angle does not change, ...);
//  * SPEC waves evaluate the next SPEC steps of the SAME chain speculatively against the current
//    state, each including its own accept decision and the list of neighbour updates it would
//    cause; wave 0 then commits them in order (a few LDS writes per accepted step) and discards
//    everything after the first accepted step that could have influenced a later one.
//    The chain is bit-for-bit the sequential one for every SPEC.
#pragma once
#include "mpp_device.hpp"

#define STASH 32              // neighbour updates remembered per speculative step
#define MPP_LDS_PARAMS_MIN_WAVES 4   // chains with at least this many waves read the parameter block from an LDS copy
#define ERR_CELL_OVERFLOW 1
#define ERR_POINT_OVERFLOW 2
#define ERR_BAD_TARGET 3
#define ERR_CAND_OVERFLOW 4
#define ERR_HANDOVER 5        // not an error: the chain has cooled down (see DevParams::handover)

#ifdef MPP_PROFILE
// diagnostic build only: cycles per phase of wave 0, summed over the launch (never in the product build)
__device__ unsigned long long g_prof[16];
__device__ unsigned long long g_prof3[16];
__device__ unsigned long long g_prof4[16];
__device__ unsigned long long g_prof2[16];
// straggler statistics of the rounds of 8 speculative steps: [k] rounds whose slowest wave ran kernel k, [8+k] its time,
// [16+k] its lead over the second slowest, [24] rounds, [25] sum of max, [26] sum of mean, [27] rounds whose slowest wave
// re-reduced a neighbour, [28+k] steps of kernel k (all waves), [36+k] their time, [44] steps with a re-reduction, [45] their time
__device__ unsigned long long g_strag[64];
#define PROF_T0() unsigned long long pt_ = clock64()
#define PROF_ADD(i) do { unsigned long long n_ = clock64(); if (c.wave == 0) prof_[i] += n_ - pt_; pt_ = n_; } while (0)
#else
#define PROF_T0()
#define PROF_ADD(i)
#endif

struct Rec {                  // one speculative step, fully evaluated
  int kernel, tidx, tslot, has_rem, has_add, valid;
  int ax, ay, rx, ry, pid, ncls;
  int accepted, n_stash, gate_a, _pad;
  int acls, _pad2;            // class of the proposed angle when it is a class edge (KEEP_EDGE_ANGLE), else unused
  double as, ar, aa, aux0, aux1, u_acc, qf, qb, dE;
  double hl, hw, ca, sa, rad, lin_a, ra0, ra1;   // derived data of the proposed point
  double fwd, bwd, log_alpha;                    // filled only when the tile is traced
};

struct Lds {
  double *s, *r, *a, *ca, *sa, *hl, *hw, *rad, *lin, *red0, *red1;
  double *edges;              // [3][32] copy of the mark bin edges
  double *trig;               // [2][32] cos / sin of (angle-class edge + pi/2): the corner trigonometry of a rectangle whose
                              // angle was drawn from the class distribution (data-driven birth / transform) without a sincos
  double *rowbase;            // [H+1] copy of the birth CDF's row level (H <= 1024), else nullptr
  double *stash_v0, *stash_v1;
  double *clip;               // [waves][CLIP_SLOTS][32] polygon buffers of the rectangle clipper
  int *xy;
  unsigned short *order, *cell_items, *cell_cnt, *stash_slot;
  unsigned char *gate;
  Rec *rec;
  int *sh;                    // [0]=n [1]=err [2]=committed ; sh[4..5] = T (double)
};

#define ROWBASE_LDS_MAX 1024
#define CLIP_SLOTS 4          // lanes of one wave that clip at the same time (the others take the next turn)
__host__ __device__ inline size_t lds_bytes(int cap, int ncell, int cell_cap, int spec, int rowbase_n, int waves) {
  size_t b = 0;
  b += (size_t)waves * CLIP_SLOTS * 32 * sizeof(double);
  b += (size_t)11 * cap * sizeof(double);
  b += (size_t)rowbase_n * sizeof(double);
  b += (size_t)3 * MPP_NCLASS * sizeof(double);
  b += (size_t)2 * MPP_NCLASS * sizeof(double);               // trig
  b += (size_t)2 * spec * STASH * sizeof(double);
  b += (size_t)cap * sizeof(int);
  b += (size_t)cap * sizeof(unsigned short);                  // order
  b += (size_t)ncell * cell_cap * sizeof(unsigned short);     // cell items
  b += (size_t)ncell * sizeof(unsigned short);                // cell counts
  b += (size_t)spec * STASH * sizeof(unsigned short);
  b += (size_t)cap;                                           // gate
  b = (b + 15) & ~(size_t)15;
  b += (size_t)spec * sizeof(Rec);
  b += 16 * sizeof(int);
  return b + 64;
}

__device__ inline Lds carve(unsigned char *base, int cap, int ncell, int cell_cap, int spec, int rowbase_n, int waves) {
  Lds L;
  double *d = (double *)base;
  L.s = d; d += cap; L.r = d; d += cap; L.a = d; d += cap; L.ca = d; d += cap; L.sa = d; d += cap;
  L.hl = d; d += cap; L.hw = d; d += cap; L.rad = d; d += cap; L.lin = d; d += cap; L.red0 = d; d += cap;
  L.red1 = d; d += cap;
  L.edges = d; d += 3 * MPP_NCLASS;
  L.trig = d; d += 2 * MPP_NCLASS;
  L.rowbase = rowbase_n > 0 ? d : nullptr; d += rowbase_n;
  L.stash_v0 = d; d += (size_t)spec * STASH; L.stash_v1 = d; d += (size_t)spec * STASH;
  L.clip = d; d += (size_t)waves * CLIP_SLOTS * 32;
  L.xy = (int *)d;
  unsigned short *u = (unsigned short *)(L.xy + cap);
  L.order = u; u += cap;
  L.cell_items = u; u += (size_t)ncell * cell_cap;
  L.cell_cnt = u; u += ncell;
  L.stash_slot = u; u += (size_t)spec * STASH;
  L.gate = (unsigned char *)u;
  size_t off = (size_t)((unsigned char *)u + cap - base);
  off = (off + 15) & ~(size_t)15;
  L.rec = (Rec *)(base + off);
  L.sh = (int *)(base + off + (size_t)spec * sizeof(Rec));
  return L;
}

// the pair terms' parameters, read ONCE per launch into registers: inside the non-unrolled loops of eval_delta every
// P->model.pair[p].field was a scalar load followed by a wait (the chain kernel is built without machine LICM)
// a wave-uniform value pinned to scalar registers (the parameter block may be an LDS copy, whose reads land in vector
// registers: 20 VGPRs for the pair terms alone); the empty asm keeps the compiler from re-reading or re-deriving it
__device__ __forceinline__ int launder_s(int v) {
  v = __builtin_amdgcn_readfirstlane(v);
  asm volatile("" : "+s"(v));
  return v;
}
__device__ __forceinline__ double launder_d(double v) {
  const long long b = __double_as_longlong(v);
  int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffll)), hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
  asm volatile("" : "+s"(lo), "+s"(hi));
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
struct PairRegs { int kind, reduce, maxd2, gated; double coef, p0, max_dist; };
// The grid facts every step asks for many times.  The parameter block lives in the kernel-argument segment, every
// P->field in the step loop is a scalar load (the kernel is built without machine LICM), and a scalar load's wait --
// s_waitcnt lgkmcnt(0): scalar loads return out of order -- also drains every LDS read in flight.  These seven are read
// ONCE per launch and laundered through an empty asm, so the compiler can neither re-load them nor hoist anything: they
// stay in scalar registers (or in lanes of a spill register, which costs a v_readlane, not a memory wait).
struct HotP { int H, W, nx, ny, cell_cap, res_shift, res_int; };
struct Chain {
  const DevParams *P;
  TileRef t;
  Lds L;
  int lane, wave;
  int np, comb;
  PairRegs pr0, pr1;
  HotP h;
};
__device__ __forceinline__ void load_hot(Chain &c) {
  const DevParams *Q = c.P;
  c.h.H = launder_s(Q->H); c.h.W = launder_s(Q->W); c.h.nx = launder_s(Q->nx); c.h.ny = launder_s(Q->ny);
  c.h.cell_cap = launder_s(Q->cell_cap); c.h.res_shift = launder_s(Q->res_shift); c.h.res_int = launder_s(Q->res_int);
}
__device__ __forceinline__ PairRegs load_pair_regs(const DevParams *P, int p) {
  PairRegs r;
  const mpp_pair_term &t = P->model.pair[p];
  r.kind = launder_s(t.kind); r.reduce = launder_s(t.reduce); r.maxd2 = launder_s(P->maxd2[p]); r.gated = launder_s(t.gated);
  r.coef = launder_d(t.coef); r.p0 = launder_d(t.p[0]); r.max_dist = launder_d(t.max_dist);
  return r;
}
__device__ __forceinline__ void load_model_regs(Chain &c) {
  c.np = launder_s(c.P->model.n_pair); c.comb = launder_s(c.P->model.combinator);
  c.pr0 = load_pair_regs(c.P, 0); c.pr1 = load_pair_regs(c.P, 1);
}
__device__ __forceinline__ PairRegs pair_regs(const Chain &c, int p) {
  PairRegs r;
  const bool z = p == 0;
  r.kind = z ? c.pr0.kind : c.pr1.kind; r.reduce = z ? c.pr0.reduce : c.pr1.reduce;
  r.maxd2 = z ? c.pr0.maxd2 : c.pr1.maxd2; r.gated = z ? c.pr0.gated : c.pr1.gated;
  r.coef = z ? c.pr0.coef : c.pr1.coef; r.p0 = z ? c.pr0.p0 : c.pr1.p0; r.max_dist = z ? c.pr0.max_dist : c.pr1.max_dist;
  return r;
}
__device__ __forceinline__ double pair_part_c(const Chain &c, int gate, double r0, double r1) {
  double l = 0.0;
  if (c.np > 0) l += c.pr0.coef * (c.pr0.gated ? (double)gate : 1.0) * r0;
  if (c.np > 1) l += c.pr1.coef * (c.pr1.gated ? (double)gate : 1.0) * r1;
  return l;
}
__device__ __forceinline__ double finish_energy_c(const Chain &c, double lin) {
  return c.comb == MPP_C_LOGISTIC ? 2.0 * sigmoid_d(lin) - 1.0 : lin;
}

__device__ __forceinline__ void wave_lds_fence() {
  // LDS traffic of one wave is executed in order; this only stops the compiler from moving
  // the loads of other lanes' data above the stores that produce them
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double readlane_d(double v, int lane) {
  long long b = __double_as_longlong(v);
  int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
  int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

__device__ __forceinline__ int cell_coord(const Chain &c, int x) {
  return c.h.res_shift >= 0 ? (x >> c.h.res_shift) : (x / c.h.res_int);
}
__device__ __forceinline__ int cell_index(const Chain &c, int x, int y, int *ci, int *cj) {
  int i = cell_coord(c, x), j = cell_coord(c, y);
  *ci = i; *cj = j;
  return j + i * c.h.ny;
}
struct Geo2 { Geo g; double rad; };
__device__ __forceinline__ Geo2 load_geo(const Lds &L, int slot) {
  Geo2 o;
  int xy = L.xy[slot];
  o.g.x = xy & 0xffff; o.g.y = (xy >> 16) & 0xffff;
  o.g.hl = L.hl[slot]; o.g.hw = L.hw[slot]; o.g.ca = L.ca[slot]; o.g.sa = L.sa[slot];
  o.rad = L.rad[slot];
  return o;
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
// Sutherland-Hodgman with the two 8-vertex polygon buffers in LDS instead of private (scratch) arrays: same
// arithmetic, same order as clip_area() of mpp_device.hpp (which the from-scratch kernels use), so the values
// agree bit for bit.  buf: 32 doubles = ax[8] ay[8] bx[8] by[8].
__device__ inline double clip_area_lds(double *buf, const double *sx, const double *sy, const double *cx,
                                       const double *cy) {
  double *ax = buf, *ay = buf + 8, *bx = buf + 16, *by = buf + 24;
  int na = 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) { ax[i] = sx[i]; ay[i] = sy[i]; }
#pragma unroll
  for (int e = 0; e < 4; ++e) {           // unrolled: the clipper's corners stay in registers (static indices)
    if (na <= 0) break;
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
    double *tx = ax, *ty = ay;           // swap the roles of the two buffers (clip_area copies b back into a)
    ax = bx; ay = by; bx = tx; by = ty;
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
// The same clipper with the whole wave on ONE polygon pair (uniform control flow required): lane i owns
// vertex i of the current polygon, the output positions come from two ballots, the shoelace sum is accumulated in
// vertex order.  Every vertex goes through exactly the arithmetic of clip_area()/clip_area_lds(), so the area is
// bit-identical; a clip costs ~4 short stages instead of ~20 dependent single-lane vertex steps.
__device__ inline double clip_area_wave(const Chain &c, const double *sx, const double *sy, const double *cx,
                                        const double *cy) {
  double *buf = c.L.clip + (size_t)c.wave * CLIP_SLOTS * 32;
  double *ax = buf, *ay = buf + 8, *bx = buf + 16, *by = buf + 24;
  if (c.lane < 4) {
    const int l = c.lane;
    ax[l] = l == 0 ? sx[0] : (l == 1 ? sx[1] : (l == 2 ? sx[2] : sx[3]));
    ay[l] = l == 0 ? sy[0] : (l == 1 ? sy[1] : (l == 2 ? sy[2] : sy[3]));
  }
  wave_lds_fence();
  const unsigned long long below = (1ull << c.lane) - 1ull;
  int na = 4;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (na <= 0) break;
    const double x0 = cx[e], y0 = cy[e], x1 = cx[(e + 1) & 3], y1 = cy[(e + 1) & 3];
    const double ex = x1 - x0, ey = y1 - y0;
    const bool valid = c.lane < na;
    const int ii = valid ? c.lane : 0, pi = ii == 0 ? na - 1 : ii - 1;
    const double qx = ax[ii], qy = ay[ii], px = ax[pi], py = ay[pi];
    const double sq = ex * (qy - y0) - ey * (qx - x0), sp = ex * (py - y0) - ey * (px - x0);
    const bool e_int = valid && (sq >= 0 ? sp < 0 : sp >= 0), e_q = valid && sq >= 0;
    const unsigned long long mi = __ballot(e_int), mq = __ballot(e_q);
    int off = __popcll(mi & below) + __popcll(mq & below);
    if (e_int) {
      if (off < 8) {
        double t = sp / (sp - sq);
        bx[off] = px + t * (qx - px); by[off] = py + t * (qy - py);
      }
      ++off;
    }
    if (e_q && off < 8) { bx[off] = qx; by[off] = qy; }
    int nb = __popcll(mi) + __popcll(mq);
    na = nb < 8 ? nb : 8;
    double *tx = ax, *ty = ay;
    ax = bx; ay = by; bx = tx; by = ty;
    wave_lds_fence();
  }
  if (na < 3) return 0.0;
  const bool valid = c.lane < na;
  const int ii = valid ? c.lane : 0, jj = ii + 1 == na ? 0 : ii + 1;
  const double term = ax[ii] * ay[jj] - ax[jj] * ay[ii];
  double s = 0.0;
  for (int i = 0; i < na; ++i) s += readlane_d(term, i);
  wave_lds_fence();
  return 0.5 * fabs(s);
}
// RectangleOverlapEnergy for the chain: lanes that really have to clip (circumscribed circles meet) take turns
// on the wave's CLIP_SLOTS polygon buffers.  Works in divergent code too: the ballots only see active lanes.
__device__ inline double overlap_energy_chain(const Chain &c, const Geo &u, const Geo &v, bool u_first, double ru,
                                              double rv, double d2) {
  double A = geo_area(u), B = geo_area(v);
  double mn = A < B ? A : B;
  double reach = ru + rv;
  bool need = !(mn < DEGENERATE_AREA) && !(d2 > reach * reach * 1.0000001);
  double area = 0.0;
  unsigned long long m = __ballot(need);
  while (m) {
    int rank = __popcll(m & ((1ull << c.lane) - 1ull));
    if (need && rank < CLIP_SLOTS) {
#ifdef MPP_PROFILE
      if (c.wave == 0) atomicAdd(&g_clip_count, 1ull);
#endif
      // (the empty asm keeps the corner arithmetic inside this rarely taken branch: the compiler would otherwise
      // hoist and speculate it for every candidate)
      Geo uu = u, vv = v;
      asm volatile("" : "+v"(uu.ca), "+v"(uu.sa), "+v"(vv.ca), "+v"(vv.sa));
      double ax[4], ay[4], bx[4], by[4];
      if (u_first) { geo_corners(uu, ax, ay); geo_corners(vv, bx, by); }
      else { geo_corners(vv, ax, ay); geo_corners(uu, bx, by); }
      area = clip_area_lds(c.L.clip + ((size_t)c.wave * CLIP_SLOTS + rank) * 32, ax, ay, bx, by) / (mn + AREA_EPS);
#ifdef MPP_PROFILE
      if (c.wave == 0 && area == 0.0) atomicAdd(&g_prof2[12], 1ull);
#endif
      need = false;
    }
    m = __ballot(need);
  }
  return area;
}

// pair energy of (u, v) when the overlap value has been computed beforehand (eval_delta's uniform clip phase)
__device__ __forceinline__ double pair_value_pre(const PairRegs &pt, const Geo2 &u, const Geo2 &v, int d2, double ovl) {
  switch (pt.kind) {
    case MPP_P_OVERLAP: return ovl;
    case MPP_P_ALIGN: return 1.0 - fabs(u.g.ca * v.g.ca + u.g.sa * v.g.sa) - (pt.p0 != 0.0 ? 1.0 : 0.0);
    case MPP_P_DIST_LE: { asm volatile("" : "+v"(d2)); return sqrt((double)d2) <= pt.max_dist ? 1.0 : 0.0; }
    case MPP_P_DIST_LT: { asm volatile("" : "+v"(d2)); return sqrt((double)d2) < pt.max_dist ? 1.0 : 0.0; }
  }
  return 0.0;
}
// pair energy of (u, v); d2 = squared centre distance (integer valued)
__device__ __forceinline__ double pair_value(const Chain &c, const PairRegs &pt, const Geo2 &u, const Geo2 &v,
                                             bool u_first, int d2) {
  switch (pt.kind) {
    case MPP_P_OVERLAP: return overlap_energy_chain(c, u.g, v.g, u_first, u.rad, v.rad, (double)d2);
    case MPP_P_ALIGN: return 1.0 - fabs(u.g.ca * v.g.ca + u.g.sa * v.g.sa) - (pt.p0 != 0.0 ? 1.0 : 0.0);
    case MPP_P_DIST_LE: { asm volatile("" : "+v"(d2)); return sqrt((double)d2) <= pt.max_dist ? 1.0 : 0.0; }   // (not speculated)
    case MPP_P_DIST_LT: { asm volatile("" : "+v"(d2)); return sqrt((double)d2) < pt.max_dist ? 1.0 : 0.0; }
  }
  return 0.0;
}

// reduction of pair term p over the neighbours of slot u (skipping `skip`, optionally including the
// proposal's new rectangle), computed by ONE lane on its own: inside eval_delta the few lanes whose
// neighbour loses its extremum each walk that neighbour's 3x3 cells, all of them at the same time
__device__ double rescan_lane(const Chain &c, int p, int u, const Geo2 &gu, int skip, bool has_add, const Rect &ar,
                              const Geo2 &ag) {
  const DevParams *P = c.P;
  const Lds &L = c.L;
  const PairRegs pt = pair_regs(c, p);
  int ci, cj;
  cell_index(c, gu.g.x, gu.g.y, &ci, &cj);
  double acc = 0.0;
#ifdef MPP_PROFILE
  if (c.wave == 0) atomicAdd(&g_prof2[10], 1ull);
#endif
  for (int di = -1; di <= 1; ++di)
    for (int dj = -1; dj <= 1; ++dj) {
      int i = ci + di, j = cj + dj;
      if (i < 0 || i >= c.h.nx || j < 0 || j >= c.h.ny) continue;
      int cell = j + i * c.h.ny, cnt = L.cell_cnt[cell];
      for (int e = 0; e < cnt; ++e) {
        int w = L.cell_items[cell * c.h.cell_cap + e];
        if (w == u || w == skip) continue;
        int wxy = L.xy[w];
        int dx = gu.g.x - (wxy & 0xffff), dy = gu.g.y - ((wxy >> 16) & 0xffff), d2 = dx * dx + dy * dy;
        if (d2 <= pt.maxd2) {
          Geo2 gw = load_geo(L, w);
          bool uf = slot_first(L, u, gu.g, gw.g.x, gw.g.y, L.s[w], L.r[w], L.a[w]);
          acc = reduce2(pt.reduce, acc, pair_value(c, pt, gu, gw, uf, d2));
        }
      }
    }
  if (has_add) {
    int dx = gu.g.x - ag.g.x, dy = gu.g.y - ag.g.y, d2 = dx * dx + dy * dy;
    if (d2 <= pt.maxd2) {
      bool uf = slot_first(L, u, gu.g, ar.x, ar.y, ar.s, ar.r, ar.a);
      acc = reduce2(pt.reduce, acc, pair_value(c, pt, gu, ag, uf, d2));
    }
  }
  return acc;
}

// dE of (remove slot `rem`, add rectangle `ar`) -- energy_graph.py:139-225 -- as
//   sum over neighbours u of [e_u(after) - e_u(before)]  +  e_added - e_removed.
// The neighbours whose cached reductions change are written to the wave's stash
// (slot, new0, new1) so that an accepted step is applied without re-evaluation;
// *n_stash > STASH means the stash overflowed.  With APPLY the caches are updated directly.
#ifdef MPP_PROFILE
#define DPROF_T0() unsigned long long dt_ = clock64()
#define DPROF(i) do { unsigned long long n_ = clock64(); if (c.wave == 0 && c.lane == 0) atomicAdd(&g_prof2[i], n_ - dt_); dt_ = n_; } while (0)
#define DCOUNT(i, v) do { if (c.wave == 0 && c.lane == 0) atomicAdd(&g_prof2[i], (unsigned long long)(v)); } while (0)
#else
#define DPROF_T0()
#define DPROF(i)
#define DCOUNT(i, v)
#endif
// `apply`: write the changed reductions straight into the caches instead of the stash (used for the rare
// step whose neighbour updates do not fit the stash; see the kernel's "apply round").
template <bool FAST = false>
__device__ double eval_delta(const Chain &c, int ri, int rem, bool has_add, const Rect &ar, const Geo2 &ag, double lin_a,
                             int gate_a, double *ra0_out, double *ra1_out, int *n_stash, bool apply) {
  const DevParams *P = c.P;
  const Lds &L = c.L;
  const int np = FAST ? 2 : c.np;
  const bool has_rem = rem >= 0;
  DPROF_T0();
  Geo2 gr;
  Rect rr;
  if (has_rem) { gr = load_geo(L, rem); rr = load_rect(L, rem); }
  else { gr = ag; rr = ar; }
  int cir = 0, cjr = 0, cia = 0, cja = 0;
  if (has_rem) cell_index(c, gr.g.x, gr.g.y, &cir, &cjr);
  if (has_add) cell_index(c, ag.g.x, ag.g.y, &cia, &cja);

  // ---- the 3x3 cells around the removed and the added point.  Lane 3k+e (k < 18, e < 3) owns entry e of cell k:
  // with at most 3 points in each of the 18 cells (the usual case) every candidate neighbour has its lane
  // without any cross-lane traffic.  Fuller cells fall back to a flattened index space over all entries.
  int my_cell = -1;
  const int ck = c.lane / 3, ce = c.lane - 3 * ck;
  if (ck < 18) {
    bool second = ck >= 9;
    int k = second ? ck - 9 : ck;
    int i = (second ? cia : cir) + k / 3 - 1, j = (second ? cja : cjr) + k % 3 - 1;
    bool ok = second ? has_add : has_rem;
    if (ok && second && has_rem && abs(i - cir) <= 1 && abs(j - cjr) <= 1) ok = false;   // already listed
    if (ok && i >= 0 && i < c.h.nx && j >= 0 && j < c.h.ny) my_cell = j + i * c.h.ny;
  }
  const int my_cnt = my_cell >= 0 ? (int)L.cell_cnt[my_cell] : 0;
  const bool direct = __ballot(my_cnt > 3) == 0ull;
  int M = WAVE;                 // direct: one pass
  if (!direct) {
    // flattened candidate index space: cell k covers [lo_k, lo_k + cnt_k)  (wave-uniform, scalar registers)
    M = 0;
#pragma unroll
    for (int k = 0; k < 18; ++k) M += __builtin_amdgcn_readlane(my_cnt, 3 * k);
  }

  DPROF(0); DCOUNT(9, M); DCOUNT(11, 1);
  double de_acc = 0.0, ra[2] = {0.0, 0.0};     // per-lane partials, combined after the loop
  bool any_changed = false, any_a = false, nonfinite = false;
  int stash_n = 0;
  for (int base = 0; base < M; base += WAVE) {
    int my_base = my_cell * c.h.cell_cap, my_e = ce;
    bool active = ce < my_cnt;
    if (!direct) {
      const int j = base + c.lane;
      int my_lo = 0, lo = 0;
      my_base = 0;
#pragma unroll
      for (int k = 0; k < 18; ++k) {
        int cnt_k = __builtin_amdgcn_readlane(my_cnt, 3 * k);
        int cell_k = __builtin_amdgcn_readlane(my_cell, 3 * k);
        if (j >= lo && cnt_k > 0) { my_lo = lo; my_base = cell_k * c.h.cell_cap; }
        lo += cnt_k;
      }
      active = j < M;
      my_e = j - my_lo;
    }
    const int u = active ? (int)L.cell_items[my_base + my_e] : 0;
    if (active && u == rem) active = false;
    Geo2 gu;
    gu.g.x = gu.g.y = 0; gu.g.hl = gu.g.hw = gu.g.ca = gu.g.sa = 0.0; gu.rad = 0.0;
    double oldv[2] = {0.0, 0.0}, newv[2] = {0.0, 0.0};
    int d2r = 0, d2a = 0;
    if (active) {
      gu = load_geo(L, u);
      oldv[0] = L.red0[u]; oldv[1] = L.red1[u];
      // (generic instantiation only -- the shipped setups have no way to an infinite energy: a neighbour whose own energy
      // is not finite makes the reference's E(after) - E(before) over the neighbourhood inf - inf = NaN, see below)
      if (!FAST) { const double lu = L.lin[u]; nonfinite = nonfinite || !(lu - lu == 0.0); }
      if (has_rem) { int dx = gu.g.x - gr.g.x, dy = gu.g.y - gr.g.y; d2r = dx * dx + dy * dy; }
      if (has_add) { int dx = gu.g.x - ag.g.x, dy = gu.g.y - ag.g.y; d2a = dx * dx + dy * dy; }
    }
    DPROF(2);
    // ---- overlap terms first, in uniform control flow: every (candidate, removed/added) pair whose circumscribed
    // circles meet is clipped by the whole wave, one pair after the other (there are ~0.2 of them per step)
    double ovl_r0 = 0.0, ovl_a0 = 0.0, ovl_r1 = 0.0, ovl_a1 = 0.0;
    // The (p, which) bodies are macros, expanded with literal arguments in the FAST instantiation -- pair 0 = rectangle
    // overlap / max, pair 1 = alignment / min, what both shipped energy setups use -- so that the kind switches, the
    // reduction modes and the per-pair selects fold away; the generic instantiation loops over runtime p / which as
    // before.  Same arithmetic either way.  (Macros, not lambdas: a closure over the candidate's state pinned that
    // state in private memory -- 2.9 KB of scratch and a third of the speed.)
#define MPP_OVERLAP_PW(p_, which_, pto_)                                                                                 \
    do {                                                                                                                  \
      const double ov = (p_) == 0 ? oldv[0] : oldv[1];                                                                    \
      bool need = active && ((which_) == 0 ? (has_rem && d2r <= (pto_).maxd2 && ov != 0.0) : (has_add && d2a <= (pto_).maxd2)); \
      const Geo2 gv = (which_) == 0 ? gr : ag;                                                                            \
      const Rect rv = (which_) == 0 ? rr : ar;                                                                            \
      const double B = geo_area(gv.g);                                                                                    \
      double mn = 0.0;                                                                                                    \
      bool uf = false;                                                                                                    \
      if (need) {                                                                                                         \
        const double A = geo_area(gu.g), reach = gu.rad + gv.rad, d2 = (double)((which_) == 0 ? d2r : d2a);               \
        mn = A < B ? A : B;                                                                                               \
        need = !(mn < DEGENERATE_AREA) && !(d2 > reach * reach * 1.0000001);                                              \
        if (need) uf = slot_first(L, u, gu.g, rv.x, rv.y, rv.s, rv.r, rv.a);                                              \
      }                                                                                                                   \
      double val = 0.0;                                                                                                   \
      unsigned long long m = __ballot(need);                                                                              \
      while (m) {                                                                                                         \
        const int src = __ffsll((long long)m) - 1;                                                                        \
        m &= m - 1;                                                                                                       \
        Geo bu;                                                                                                           \
        bu.x = __builtin_amdgcn_readlane(gu.g.x, src); bu.y = __builtin_amdgcn_readlane(gu.g.y, src);                     \
        bu.hl = readlane_d(gu.g.hl, src); bu.hw = readlane_d(gu.g.hw, src);                                               \
        bu.ca = readlane_d(gu.g.ca, src); bu.sa = readlane_d(gu.g.sa, src);                                               \
        const bool u_first = __builtin_amdgcn_readlane((int)uf, src) != 0;                                                \
        DCOUNT(8, 1);                                                                                                     \
        double ux[4], uy[4], vx[4], vy[4];                                                                                \
        geo_corners(bu, ux, uy); geo_corners(gv.g, vx, vy);                                                               \
        double sx[4], sy[4], cx[4], cy[4];          /* subject = the smaller rectangle in the canonical order */         \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                   \
          sx[i] = u_first ? ux[i] : vx[i]; sy[i] = u_first ? uy[i] : vy[i];                                               \
          cx[i] = u_first ? vx[i] : ux[i]; cy[i] = u_first ? vy[i] : uy[i];                                               \
        }                                                                                                                 \
        const double area = clip_area_wave(c, sx, sy, cx, cy);                                                            \
        if (c.lane == src) val = area / (mn + AREA_EPS);                                                                  \
      }                                                                                                                   \
      if ((p_) == 0) { if ((which_) == 0) ovl_r0 = val; else ovl_a0 = val; }                                              \
      else { if ((which_) == 0) ovl_r1 = val; else ovl_a1 = val; }                                                        \
    } while (0)
    // which = 0: against the removed point (is it the one carrying u's extremum?)
    // which = 1: against the added point (it may become u's new extremum)
#define MPP_AGAINST(p_, which_, pt_)                                                                                     \
    do {                                                                                                                  \
      const bool in = (which_) == 0 ? (has_rem && d2r <= (pt_).maxd2 && ov != 0.0) : (has_add && d2a <= (pt_).maxd2);     \
      if (in) {                                                                                                           \
        const Geo2 gv = (which_) == 0 ? gr : ag;                                                                          \
        const double pre = (p_) == 0 ? ((which_) == 0 ? ovl_r0 : ovl_a0) : ((which_) == 0 ? ovl_r1 : ovl_a1);             \
        const double v = pair_value_pre((pt_), gu, gv, (which_) == 0 ? d2r : d2a, pre);                                   \
        if ((which_) == 0) carries = (v == ov);                                                                           \
        else {                                                                                                            \
          if ((p_) == 0) ra[0] = reduce2((pt_).reduce, ra[0], v); else ra[1] = reduce2((pt_).reduce, ra[1], v);           \
          v_add = v; got_add = true;                                                                                      \
          any_a = true;                                                                                                   \
        }                                                                                                                 \
      }                                                                                                                   \
    } while (0)
    // u loses the neighbour that carried its extremum: every other neighbour is no more extreme than `ov`, so an added
    // value at least as extreme IS the new extremum (exactly); otherwise u is re-reduced over its own 3x3 cells (in
    // its lane)
// (RESCAN_: the re-reduction call itself; the FAST instantiation defers it to ONE call site after both pairs -- with two
// inlined copies the inliner gave up on rescan_lane, and a real call taking the Chain by reference pinned the chain state
// and the 1.9 KB parameter block in private memory)
#define MPP_PAIR_P(p_, pt_, UNROLLED, RESCAN_)                                                                           \
    do {                                                                                                                  \
      const double ov = (p_) == 0 ? oldv[0] : oldv[1];                                                                    \
      double nv = ov, v_add = 0.0;                                                                                        \
      bool carries = false, got_add = false;                                                                              \
      if (UNROLLED) { MPP_AGAINST(p_, 0, pt_); MPP_AGAINST(p_, 1, pt_); }                                                 \
      else {                                                                                                              \
        _Pragma("clang loop unroll(disable)") for (int which = 0; which < 2; ++which) MPP_AGAINST(p_, which, pt_);        \
      }                                                                                                                   \
      if (carries) {                                                                                                      \
        const bool dominates = got_add && ((pt_).reduce == MPP_REDUCE_MAX ? v_add >= ov : v_add <= ov);                   \
        if (dominates) nv = v_add; else { RESCAN_; }                                                                      \
      } else if (got_add) {                                                                                               \
        nv = reduce2((pt_).reduce, ov, v_add);                                                                            \
      }                                                                                                                   \
      if ((p_) == 0) newv[0] = nv; else newv[1] = nv;                                                                     \
    } while (0)
    PairRegs f0 = c.pr0, f1 = c.pr1;                 // FAST: the kinds and reduction modes are compile-time facts
    f0.kind = MPP_P_OVERLAP; f0.reduce = MPP_REDUCE_MAX; f1.kind = MPP_P_ALIGN; f1.reduce = MPP_REDUCE_MIN;
    if constexpr (FAST) {
      MPP_OVERLAP_PW(0, 0, f0);
      MPP_OVERLAP_PW(0, 1, f0);
    } else {
#pragma clang loop unroll(disable)
      for (int p = 0; p < np; ++p) {
        const PairRegs pto = pair_regs(c, p);
        if (pto.kind != MPP_P_OVERLAP) continue;
#pragma clang loop unroll(disable)
        for (int which = 0; which < 2; ++which) MPP_OVERLAP_PW(p, which, pto);
      }
    }
    DPROF(3);
    if constexpr (FAST) {
      bool resc0 = false, resc1 = false;
      if (active) {
        MPP_PAIR_P(0, f0, true, resc0 = true);
        MPP_PAIR_P(1, f1, true, resc1 = true);
      }
      // ---- a neighbour lost the point that carried its extremum (its alignment minimum: 0.12 times per step; its
      // overlap maximum: rarely, often in crowded scenes) and the added point does not take over: re-reduce it over its
      // own 3x3 cells.  Done in one lane (rescan_lane, what the generic instantiation does) this walk -- nine cells,
      // dependent LDS reads per entry, an inlined clipper -- cost ~50 VGPRs of pressure in the hottest region of the
      // kernel (256 registers and spills instead of ~200: 12 % of the speed) and ~2 000 cycles of one straggling wave.
      // Here the whole wave does it: the neighbour is broadcast (scalar registers), lane 3k+e takes entry e of cell k
      // (fuller cells: a flattened index space, 64 entries per turn), overlaps that need clipping go through the
      // wave-wide clipper one after the other, the extremum is collected with ballots and readlanes.  Same values, same
      // (exact) max / min: the same result as the in-lane walk.
      unsigned long long rm = __ballot(resc0 || resc1);
#ifdef MPP_PROFILE
      if (rm && c.lane == 0) c.L.sh[8 + (c.wave & 7)] = 1;
#endif
      while (rm) {
        const int src = __ffsll((long long)rm) - 1;
        rm &= rm - 1;
        const bool need0 = __builtin_amdgcn_readlane((int)resc0, src) != 0, need1 = __builtin_amdgcn_readlane((int)resc1, src) != 0;
        const int us = __builtin_amdgcn_readlane(u, src);
        Geo2 bu;                                           // the neighbour, wave-uniform
        bu.g.x = __builtin_amdgcn_readlane(gu.g.x, src); bu.g.y = __builtin_amdgcn_readlane(gu.g.y, src);
        bu.g.hl = readlane_d(gu.g.hl, src); bu.g.hw = readlane_d(gu.g.hw, src);
        bu.g.ca = readlane_d(gu.g.ca, src); bu.g.sa = readlane_d(gu.g.sa, src); bu.rad = readlane_d(gu.rad, src);
        int uci, ucj;
        cell_index(c, bu.g.x, bu.g.y, &uci, &ucj);
        int cell2 = -1;
        if (ck < 9) {
          const int i = uci + ck / 3 - 1, j = ucj + ck % 3 - 1;
          if (i >= 0 && i < c.h.nx && j >= 0 && j < c.h.ny) cell2 = j + i * c.h.ny;
        }
        const int cnt2 = cell2 >= 0 ? (int)L.cell_cnt[cell2] : 0;
        const bool direct2 = __ballot(cnt2 > 3) == 0ull;
        int M2 = WAVE;
        if (!direct2) {
          M2 = 0;
#pragma unroll
          for (int k = 0; k < 9; ++k) M2 += __builtin_amdgcn_readlane(cnt2, 3 * k);
        }
        const double rew = f1.p0 != 0.0 ? 1.0 : 0.0, Au = geo_area(bu.g);
        double acc0 = 0.0, acc1 = 0.0;                     // wave-uniform results
        for (int base2 = 0; base2 < M2; base2 += WAVE) {
          int it_base = cell2 * c.h.cell_cap, it_e = ce;
          bool act2 = ce < cnt2;
          if (!direct2) {
            const int j = base2 + c.lane;
            int lo = 0, my_lo = 0;
            it_base = 0;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
              const int cnt_k = __builtin_amdgcn_readlane(cnt2, 3 * k), cell_k = __builtin_amdgcn_readlane(cell2, 3 * k);
              if (j >= lo && cnt_k > 0) { my_lo = lo; it_base = cell_k * c.h.cell_cap; }
              lo += cnt_k;
            }
            act2 = j < M2;
            it_e = j - my_lo;
          }
          const int w = act2 ? (int)L.cell_items[it_base + it_e] : 0;
          if (act2 && (w == us || w == rem)) act2 = false;
          int wx = 0, wy = 0, d2w = 0;
          if (act2) {
            const int wxy = L.xy[w];
            wx = wxy & 0xffff; wy = (wxy >> 16) & 0xffff;
            const int dx = bu.g.x - wx, dy = bu.g.y - wy;
            d2w = dx * dx + dy * dy;
          }
          if (need1) {                                     // alignment, reduced with min
            double v1 = 0.0;
            if (act2 && d2w <= f1.maxd2) v1 = 1.0 - fabs(bu.g.ca * L.ca[w] + bu.g.sa * L.sa[w]) - rew;
            unsigned long long bm = __ballot(v1 < 0.0);
            while (bm) {
              const int l2 = __ffsll((long long)bm) - 1;
              bm &= bm - 1;
              acc1 = reduce2(MPP_REDUCE_MIN, acc1, readlane_d(v1, l2));
            }
          }
          if (need0) {                                     // overlap, reduced with max
            bool needc = act2 && d2w <= f0.maxd2;
            Geo gw;
            gw.x = wx; gw.y = wy; gw.hl = gw.hw = gw.ca = gw.sa = 0.0;
            double mn = 0.0;
            bool uf = false;
            if (needc) {
              gw.hl = L.hl[w]; gw.hw = L.hw[w]; gw.ca = L.ca[w]; gw.sa = L.sa[w];
              const double B = geo_area(gw), reach = bu.rad + L.rad[w], d2 = (double)d2w;
              mn = Au < B ? Au : B;
              needc = !(mn < DEGENERATE_AREA) && !(d2 > reach * reach * 1.0000001);
              if (needc) uf = slot_first(L, us, bu.g, wx, wy, L.s[w], L.r[w], L.a[w]);
            }
            unsigned long long m = __ballot(needc);
            while (m) {
              const int l2 = __ffsll((long long)m) - 1;
              m &= m - 1;
              Geo bw;
              bw.x = __builtin_amdgcn_readlane(gw.x, l2); bw.y = __builtin_amdgcn_readlane(gw.y, l2);
              bw.hl = readlane_d(gw.hl, l2); bw.hw = readlane_d(gw.hw, l2);
              bw.ca = readlane_d(gw.ca, l2); bw.sa = readlane_d(gw.sa, l2);
              const bool u_first = __builtin_amdgcn_readlane((int)uf, l2) != 0;
              const double mn2 = readlane_d(mn, l2);
              double ux[4], uy[4], vx[4], vy[4];
              geo_corners(bu.g, ux, uy); geo_corners(bw, vx, vy);
              double sx[4], sy[4], cx[4], cy[4];
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                sx[i] = u_first ? ux[i] : vx[i]; sy[i] = u_first ? uy[i] : vy[i];
                cx[i] = u_first ? vx[i] : ux[i]; cy[i] = u_first ? vy[i] : uy[i];
              }
              acc0 = reduce2(MPP_REDUCE_MAX, acc0, clip_area_wave(c, sx, sy, cx, cy) / (mn2 + AREA_EPS));
            }
          }
        }
        if (has_add) {                                     // the proposed rectangle is a neighbour too (uniform)
          const int dx = bu.g.x - ag.g.x, dy = bu.g.y - ag.g.y, d2a2 = dx * dx + dy * dy;
          if (need1 && d2a2 <= f1.maxd2)
            acc1 = reduce2(MPP_REDUCE_MIN, acc1, 1.0 - fabs(bu.g.ca * ag.g.ca + bu.g.sa * ag.g.sa) - rew);
          if (need0 && d2a2 <= f0.maxd2) {
            const double B = geo_area(ag.g), mn = Au < B ? Au : B, reach = bu.rad + ag.rad, d2 = (double)d2a2;
            if (!(mn < DEGENERATE_AREA) && !(d2 > reach * reach * 1.0000001)) {
              const bool u_first = slot_first(L, us, bu.g, ar.x, ar.y, ar.s, ar.r, ar.a);
              double ux[4], uy[4], vx[4], vy[4];
              geo_corners(bu.g, ux, uy); geo_corners(ag.g, vx, vy);
              double sx[4], sy[4], cx[4], cy[4];
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                sx[i] = u_first ? ux[i] : vx[i]; sy[i] = u_first ? uy[i] : vy[i];
                cx[i] = u_first ? vx[i] : ux[i]; cy[i] = u_first ? vy[i] : uy[i];
              }
              acc0 = reduce2(MPP_REDUCE_MAX, acc0, clip_area_wave(c, sx, sy, cx, cy) / (mn + AREA_EPS));
            }
          }
        }
        if (c.lane == src) {
          if (need0) newv[0] = acc0;
          if (need1) newv[1] = acc1;
        }
      }
    } else if (active) {
#pragma clang loop unroll(disable)
      for (int p = 0; p < np; ++p) {
        const PairRegs pt = pair_regs(c, p);
        MPP_PAIR_P(p, pt, false, nv = rescan_lane(c, p, u, gu, rem, has_add, ar, ag));
      }
    }
#undef MPP_OVERLAP_PW
#undef MPP_AGAINST
#undef MPP_PAIR_P
    DPROF(5);
    const bool changed = active && ((newv[0] != oldv[0]) || (newv[1] != oldv[1]));
    if (changed) {
      double lin = L.lin[u];
      int gt = L.gate[u];
      de_acc += finish_energy_c(c, lin + pair_part_c(c, gt, newv[0], newv[1])) -
                finish_energy_c(c, lin + pair_part_c(c, gt, oldv[0], oldv[1]));
      any_changed = true;
      if (apply) { L.red0[u] = newv[0]; L.red1[u] = newv[1]; }
    }
    const unsigned long long cm = __ballot(changed);
    if (!apply && changed) {
      int rank = stash_n + __popcll(cm & ((1ull << c.lane) - 1ull));
      if (rank < STASH) {
        L.stash_slot[ri * STASH + rank] = (unsigned short)u;
        L.stash_v0[ri * STASH + rank] = newv[0];
        L.stash_v1[ri * STASH + rank] = newv[1];
      }
    }
    stash_n += __popcll(cm);
    DPROF(6);
  }
  DPROF(1);
  // combine the few lanes that contribute, in ascending lane order (deterministic, wave-uniform result)
  double sum_de = 0.0, ra0 = 0.0, ra1 = 0.0;
  unsigned long long cm = __ballot(any_changed);
  while (cm) {
    int src = __ffsll((long long)cm) - 1;
    cm &= cm - 1;
    sum_de += readlane_d(de_acc, src);
  }
  unsigned long long am = __ballot(any_a);
  while (am) {
    int src = __ffsll((long long)am) - 1;
    am &= am - 1;
    if (np > 0) ra0 = reduce2(c.pr0.reduce, ra0, readlane_d(ra[0], src));
    if (np > 1) ra1 = reduce2(c.pr1.reduce, ra1, readlane_d(ra[1], src));
  }
  *ra0_out = ra0; *ra1_out = ra1;
  *n_stash = stash_n;
  double dE = sum_de;
  if (has_add) dE += finish_energy_c(c, lin_a + pair_part_c(c, gate_a, ra0, ra1));
  if (has_rem) dE -= finish_energy_c(c, L.lin[rem] + pair_part_c(c, (int)L.gate[rem], L.red0[rem], L.red1[rem]));
  // The reference subtracts two sums over ALL points of the 3 x 3 cells around the change (energy_graph.py:139-225); with
  // a point of infinite energy among them (the `craciun` contrast measure on a one-pixel mask) that is inf - inf = NaN
  // and the step is rejected, whatever the terms that change sum to.
  if (!FAST) { if (__ballot(nonfinite)) dE = nan(""); }
  DPROF(4);
  return dE;
}

// ---- proposal densities (shape_samplers.py:103-108, transform_kernels.py:94-99, :203-225) -------
// One 32-bin mark row (128 B): every lane reads the whole row (a broadcast read of one cache line) and
// sums it in index order -- no cross-lane traffic, wave-uniform result, same order as the oracle.
// Returns P[cls]/sum; with `draw` the class is first drawn: #{j : cumsum_j <= u*sum}.
// `cls2` >= 0: *p2 receives P[cls2]/sum of the same row (the backward probability of the data-driven transform).
__device__ double row_prob(const Chain &c, int k, int x, int y, int cls, bool draw, double u, int *drawn, int cls2 = -1,
                           double *p2 = nullptr) {
  const float4 *row = (const float4 *)mark_row_w(c.h.W, c.t, k, x, y);
  float v[MPP_NCLASS];
#pragma unroll
  for (int i = 0; i < 8; ++i) { float4 q = row[i]; v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w; }
  double tot = 0.0;
  if (draw) {
    // the running sums ARE the partial sums of the total (same order, same roundings): one dependent chain of 32
    // additions instead of two, the counts afterwards are independent compares
    double cs[MPP_NCLASS];
#pragma unroll
    for (int i = 0; i < MPP_NCLASS; ++i) { tot += (double)v[i]; cs[i] = tot; }
    const double thr = u * tot;
    int d = 0;
#pragma unroll
    for (int i = 0; i < MPP_NCLASS; ++i) d += (cs[i] <= thr) ? 1 : 0;
    cls = d < MPP_NCLASS ? d : MPP_NCLASS - 1;
    *drawn = cls;
  } else {
#pragma unroll
    for (int i = 0; i < MPP_NCLASS; ++i) tot += (double)v[i];
  }
  float pc = 0.f;
#pragma unroll
  for (int i = 0; i < MPP_NCLASS; ++i) pc = (i == cls) ? v[i] : pc;
  if (cls2 >= 0) {
    float pc2 = 0.f;
#pragma unroll
    for (int i = 0; i < MPP_NCLASS; ++i) pc2 = (i == cls2) ? v[i] : pc2;
    *p2 = (double)pc2 / tot;
  }
  return (double)pc / tot;
}
// `coop`: the wave works on ONE proposal (wave mode).  The three mark rows are then handled by lanes 0, 1 and 2 at
// the same time -- every lane runs the same sequential sums on its own row, so the values are the ones of three
// calls in a row -- and collected with readlane: one row latency and one pass of arithmetic instead of three.
__device__ double birth_density(const Chain &c, const Rect &q, bool coop) {
  const DevParams *P = c.P;
  double d = (double)c.t.det[(size_t)q.x * c.h.W + q.y] / (c.L.rowbase ? c.L.rowbase[c.h.H] : c.t.rowbase[c.h.H]);
  if (coop) {
    const int kk = c.lane < 2 ? c.lane : 2;
    const double pl = row_prob(c, kk, q.x, q.y, value_to_class_tab(P, c.L.edges + kk * MPP_NCLASS, kk, mark_of(q, kk)),
                               false, 0.0, nullptr);
    d *= readlane_d(pl, 0); d *= readlane_d(pl, 1); d *= readlane_d(pl, 2);
  } else {
#pragma clang loop unroll(disable)
    for (int k = 0; k < 3; ++k)
      d *= row_prob(c, k, q.x, q.y, value_to_class_tab(P, c.L.edges + k * MPP_NCLASS, k, mark_of(q, k)), false, 0.0, nullptr);
  }
  return d * ((double)c.h.H * (double)c.h.W * 32768.0);
}
// data-driven translation (transform_kernels.py:77-89): draw a pixel of the (2*max_delta+1)^2 window
// around (x,y) with probability det/sum.  Lane i owns window row i (<= 31 rows): its segment sum comes
// from the per-row prefix table, the row is found by an ordered readlane walk, the column by a ballot.
__device__ void window_draw(const Chain &c, int x, int y, double u, int *ex, int *ey) {
  const DevParams *P = c.P;
  const int md = P->kern.max_delta;
  const int x0 = max(0, x - md), x1 = min(x + md + 1, c.h.H), y0 = max(0, y - md), y1 = min(y + md + 1, c.h.W);
  const int nrow = x1 - x0, wc = y1 - y0;
  const double tot = c.t.boxsum[(size_t)x * c.h.W + y];
  double seg = 0.0;
  if (c.lane < nrow) {
    const double *rp = c.t.rowpart + (size_t)(x0 + c.lane) * c.h.W;
    seg = rp[y1 - 1] - (y0 > 0 ? rp[y0 - 1] : 0.0);
  }
  double before = 0.0;          // sum of the rows above the chosen one
  const double thr = u * tot;   // cdf <= u  <=>  partial sum <= u * total
  int row = 0;
  for (int i = 0; i < nrow; ++i) {
    double nxt = before + readlane_d(seg, i);
    if (nxt <= thr && i < nrow - 1) { before = nxt; row = i + 1; } else break;
  }
  const double *rp = c.t.rowpart + (size_t)(x0 + row) * c.h.W;
  const double lead = y0 > 0 ? rp[y0 - 1] : 0.0;
  bool le = false;
  if (c.lane < wc) le = (before + (rp[y0 + c.lane] - lead)) <= thr;
  int col = __popcll(__ballot(le));
  if (col >= wc) col = wc - 1;
  *ex = x0 + row; *ey = y0 + col;
}
__device__ __forceinline__ double normal_pdf(double x, double sigma) {
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


// =====================================================================================================
// Lane mode: ONE LANE evaluates one speculative step completely on its own -- no cross-lane traffic.
// A wave instruction costs the same with 1 or 64 active lanes, and one dependent float64 chain cannot
// use more than one SIMD, so the steps of a round are spread over the lanes of four waves (= the four
// SIMDs of the CU).  The arithmetic (and its order) is the one of the wave-cooperative functions above,
// so both modes produce byte-identical chains.
// =====================================================================================================
// STASH_ON = false (the deep-round kernel, mpp_deep.hip): nothing is stashed -- a step that commits is evaluated a second
// time with `apply` -- only the number of neighbours that change is counted.
template <bool STASH_ON = true>
__device__ double eval_delta_lane(const Chain &c, int ri, int rem, bool has_add, const Rect &ar, const Geo2 &ag,
                                  double lin_a, int gate_a, double *ra0_out, double *ra1_out, int *n_stash, bool apply) {
  const DevParams *P = c.P;
  const Lds &L = c.L;
  const int np = c.np;
  const bool has_rem = rem >= 0;
  DPROF_T0();
  Geo2 gr;
  Rect rr;
  if (has_rem) { gr = load_geo(L, rem); rr = load_rect(L, rem); }
  else { gr = ag; rr = ar; }
  int cir = 0, cjr = 0, cia = 0, cja = 0;
  if (has_rem) cell_index(c, gr.g.x, gr.g.y, &cir, &cjr);
  if (has_add) cell_index(c, ag.g.x, ag.g.y, &cia, &cja);
  double sum_de = 0.0, ra[2] = {0.0, 0.0};
  int stash_n = 0;
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 0 ? !has_rem : !has_add) continue;
    const int ci0 = pass == 0 ? cir : cia, cj0 = pass == 0 ? cjr : cja;
    for (int k = 0; k < 9; ++k) {
      int i = ci0 + k / 3 - 1, j = cj0 + k % 3 - 1;
      if (i < 0 || i >= c.h.nx || j < 0 || j >= c.h.ny) continue;
      if (pass == 1 && has_rem && abs(i - cir) <= 1 && abs(j - cjr) <= 1) continue;     // already visited
      const int cell = j + i * c.h.ny, cnt = L.cell_cnt[cell];
      for (int e = 0; e < cnt; ++e) {
        const int u = L.cell_items[cell * c.h.cell_cap + e];
        if (u == rem) continue;
        const int uxy = L.xy[u];
        const int ux = uxy & 0xffff, uy = (uxy >> 16) & 0xffff;
        int d2r = 0, d2a = 0;
        if (has_rem) { int dx = ux - gr.g.x, dy = uy - gr.g.y; d2r = dx * dx + dy * dy; }
        if (has_add) { int dx = ux - ag.g.x, dy = uy - ag.g.y; d2a = dx * dx + dy * dy; }
        bool touch = false;
        for (int p = 0; p < np; ++p)
          touch |= (has_rem && d2r <= pair_regs(c, p).maxd2) || (has_add && d2a <= pair_regs(c, p).maxd2);
        if (!touch) continue;                    // too far to interact with either point
        Geo2 gu = load_geo(L, u);
        double oldv[2] = {L.red0[u], L.red1[u]}, newv[2];
        newv[0] = oldv[0]; newv[1] = oldv[1];
        for (int p = 0; p < np; ++p) {
          const PairRegs pt = pair_regs(c, p);
          bool in_r = has_rem && d2r <= pt.maxd2, in_a = has_add && d2a <= pt.maxd2;
          double nv = oldv[p];
          bool slow = false;
          if (in_r && oldv[p] != 0.0) {
            bool uf = slot_first(L, u, gu.g, rr.x, rr.y, rr.s, rr.r, rr.a);
            if (pair_value(c, pt, gu, gr, uf, d2r) == oldv[p]) slow = true;     // the removed point carries u's extremum
          }
          double v_a = 0.0;
          if (in_a) {
            bool uf = slot_first(L, u, gu.g, ar.x, ar.y, ar.s, ar.r, ar.a);
            v_a = pair_value(c, pt, gu, ag, uf, d2a);
            ra[p] = reduce2(pt.reduce, ra[p], v_a);
            nv = reduce2(pt.reduce, nv, v_a);
          }
          // (an added value at least as extreme as the lost extremum is the new extremum: no re-reduction)
          if (slow && !(in_a && (pt.reduce == MPP_REDUCE_MAX ? v_a >= oldv[p] : v_a <= oldv[p])))
            nv = rescan_lane(c, p, u, gu, rem, has_add, ar, ag);
          newv[p] = nv;
        }
        if (newv[0] != oldv[0] || newv[1] != oldv[1]) {
          double lin = L.lin[u];
          int gt = L.gate[u];
          sum_de += finish_energy_c(c, lin + pair_part_c(c, gt, newv[0], newv[1])) -
                    finish_energy_c(c, lin + pair_part_c(c, gt, oldv[0], oldv[1]));
          if (apply) { L.red0[u] = newv[0]; L.red1[u] = newv[1]; }
          else if (STASH_ON && stash_n < STASH) {
            L.stash_slot[ri * STASH + stash_n] = (unsigned short)u;
            L.stash_v0[ri * STASH + stash_n] = newv[0];
            L.stash_v1[ri * STASH + stash_n] = newv[1];
          }
          ++stash_n;
        }
      }
    }
  }
  *ra0_out = ra[0]; *ra1_out = ra[1];
  *n_stash = stash_n;
  double dE = sum_de;
  if (has_add) dE += finish_energy_c(c, lin_a + pair_part_c(c, gate_a, ra[0], ra[1]));
  if (has_rem) dE -= finish_energy_c(c, L.lin[rem] + pair_part_c(c, (int)L.gate[rem], L.red0[rem], L.red1[rem]));
  DPROF(4);
  return dE;
}

// #{i in [0,n) : key(i) <= u} for a non-decreasing key, by repeated 8-way splitting (7 independent loads
// per level instead of a chain of log2(n) dependent ones)
template <typename F>
__device__ int count_le_8ary(int n, F key_le) {
  int lo = 0, hi = n;                 // invariant: all i < lo satisfy key<=u, all i >= hi do not
  while (hi - lo > 8) {
    int step = (hi - lo) >> 3, c = 0;
#pragma unroll
    for (int j = 1; j <= 7; ++j) c += key_le(lo + j * step - 1) ? 1 : 0;
    int nlo = lo + c * step;          // probes are at lo+step-1, lo+2step-1, ...: c of them are <= u
    hi = c < 7 ? lo + (c + 1) * step - 1 : hi;
    lo = nlo;
  }
  // the last (at most 8) entries: probed together -- one memory latency instead of one per entry (key_le of an index
  // beyond hi is never counted; the probe is clamped so that it reads inside the table)
  int c = lo;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int i = lo + j < hi ? lo + j : (hi > 0 ? hi - 1 : 0);
    const bool le = key_le(i);
    c += (lo + j < hi && le) ? 1 : 0;
  }
  return c;
}

__device__ __forceinline__ void window_draw_lane(const Chain &c, int x, int y, double u, int *ex, int *ey) {
  const DevParams *P = c.P;
  const int md = P->kern.max_delta;
  const int x0 = max(0, x - md), x1 = min(x + md + 1, c.h.H), y0 = max(0, y - md), y1 = min(y + md + 1, c.h.W);
  const int nrow = x1 - x0, wc = y1 - y0;
  const double tot = c.t.boxsum[(size_t)x * c.h.W + y];
  double before = 0.0;
  const double thr = u * tot;
  int row = 0;
  if (nrow <= 17 && wc <= 17) {
    // the usual window (max_delta <= 8): the segment sums of all rows are requested together -- two memory latencies for
    // the row, one for the column, instead of two per row walked -- and added in the order of the walk below
    double seg[17];
#pragma unroll
    for (int i = 0; i < 17; ++i) {
      const int xi = x0 + (i < nrow ? i : nrow - 1);
      const MPP_GLOBAL double *rp = c.t.rowpart + (size_t)xi * c.h.W;
      const double hi_ = rp[y1 - 1], lo_ = y0 > 0 ? rp[y0 - 1] : 0.0;
      seg[i] = hi_ - lo_;
    }
    bool go = true;
#pragma unroll
    for (int i = 0; i < 17; ++i) {
      const double nxt = before + seg[i];
      go = go && i < nrow && nxt <= thr && i < nrow - 1;
      if (go) { before = nxt; row = i + 1; }
    }
    const MPP_GLOBAL double *rp = c.t.rowpart + (size_t)(x0 + row) * c.h.W;
    const double lead = y0 > 0 ? rp[y0 - 1] : 0.0;
    double cv[17];
#pragma unroll
    for (int jj = 0; jj < 17; ++jj) cv[jj] = rp[y0 + (jj < wc ? jj : wc - 1)];
    int col = 0;
#pragma unroll
    for (int jj = 0; jj < 17; ++jj) col += (jj < wc && (before + (cv[jj] - lead)) <= thr) ? 1 : 0;
    if (col >= wc) col = wc - 1;
    *ex = x0 + row; *ey = y0 + col;
    return;
  }
  for (int i = 0; i < nrow; ++i) {
    const double *rp = c.t.rowpart + (size_t)(x0 + i) * c.h.W;
    double nxt = before + (rp[y1 - 1] - (y0 > 0 ? rp[y0 - 1] : 0.0));
    if (nxt <= thr && i < nrow - 1) { before = nxt; row = i + 1; } else break;
  }
  const double *rp = c.t.rowpart + (size_t)(x0 + row) * c.h.W;
  const double lead = y0 > 0 ? rp[y0 - 1] : 0.0;
  int col = 0;
  for (int jj = 0; jj < wc; ++jj) col += ((before + (rp[y0 + jj] - lead)) <= thr) ? 1 : 0;
  if (col >= wc) col = wc - 1;
  *ex = x0 + row; *ey = y0 + col;
}

// which parts of the target's cached geometry survive the proposal
#define KEEP_TRIG 1
#define KEEP_SIZE 2
#define KEEP_QF 4             // the forward density was computed while drawing (data-driven birth)
#define KEEP_QFB 8            // forward AND backward probability were computed while drawing (data-driven transform)
#define KEEP_EDGE_ANGLE 16    // the proposed angle is the lower edge of class r.ncls: its cos / sin are in the LDS table
#define KEEP_MV 32            // the score-map values of the proposed rectangle were fetched while drawing

// Wave mode with remap tables: the kernels that draw a mark class from a score-map row (data-driven birth and
// transform) know the PIXEL of the proposed rectangle one memory latency before they know its classes.  The three
// table rows of that pixel (32 float64 each) are requested right then, one entry per lane -- tables 0 and 1 in the two
// halves of the wave, table 2 in a second load -- next to the mark rows, and the entries of the drawn classes are
// picked with readlane afterwards: the unit terms no longer wait for a table lookup that depends on the draw.
struct TabRows { double a, b; };
__device__ __forceinline__ TabRows tab_rows_request(const Chain &c, size_t pix) {
  const MPP_GLOBAL double *t0 = c.t.rm[0], *t1 = c.t.rm[1], *t2 = c.t.rm[2];
  const int e = c.lane & (MPP_NCLASS - 1);
  TabRows t;
  t.a = (c.lane < MPP_NCLASS ? t0 : t1)[pix + e];
  t.b = t2[pix + e];
  return t;
}
__device__ __forceinline__ MapVals tab_rows_pick(const Chain &c, const TabRows &t, float det, const Rect &q) {
  const DevParams *P = c.P;
  // (the classes the unit terms would look up: load_map_vals_w())
  const int c0 = __builtin_amdgcn_readfirstlane(value_to_class_tab(P, c.L.edges, 0, q.s)),
            c1 = __builtin_amdgcn_readfirstlane(value_to_class_tab(P, c.L.edges + MPP_NCLASS, 1, q.r)),
            c2 = __builtin_amdgcn_readfirstlane(value_to_class_tab(P, c.L.edges + 2 * MPP_NCLASS, 2, q.a));
  MapVals v;
  v.det = det; v.m0 = v.m1 = v.m2 = 0.f; v.tab = 1;
  v.r0 = readlane_d(t.a, c0); v.r1 = readlane_d(t.a, MPP_NCLASS + c1); v.r2 = readlane_d(t.b, c2);
  return v;
}

// draw the proposal of a step from its 12 Philox words (the same recipe as the oracle's)
// w: the step's Philox blocks 0 and 1.  The uniform of the accept test comes from words 6, 7 -- except for the
// kernels that use all eight words themselves (the two births, split): only they pay for block 2.
// the kernel of a step and the uniform of its accept test (state-independent)
template <bool LANE>
__device__ __forceinline__ int draw_head(const Chain &c, const uint32_t w[8], Rec &r, uint32_t k0, uint32_t k1,
                                         uint64_t step, uint32_t chain) {
  const DevParams *P = c.P;
  double uk = u53(w[0], w[1]);
  int k = 0;
  while (k < P->n_kernels - 1 && P->p_cum[k] <= uk) ++k;
  r.kernel = k; r.tidx = -1; r.tslot = -1; r.has_rem = 0; r.has_add = 0; r.pid = -1; r.ncls = -1; r.acls = 0; r._pad2 = 0;
  r.aux0 = r.aux1 = 0.0; r.ax = r.ay = 0; r.as = r.ar = r.aa = 0.0; r.rx = r.ry = 0;
  if (k == MPP_K_UBIRTH || k == MPP_K_DBIRTH || k == MPP_K_SPLIT) {
    uint32_t e[4];
    philox4x32_10((uint32_t)step, (uint32_t)(step >> 32), 2u, chain, k0, k1, e);
    r.u_acc = u53(e[2], e[3]);
  } else {
    r.u_acc = u53(w[6], w[7]);
  }
  return k;
}
// the two birth kernels: everything they draw depends on the step's random words and the score maps only, not on the
// configuration.  (Kept apart from draw_proposal(): a build that let waves draw their NEXT birth while they wait for the
// slowest wave of the round lost 6 % -- a waiting wave costs its SIMD nothing, a drawing one competes with the straggler
// it shares the SIMD with -- but the split itself gained 1.4 % on one tile and 3.2 % with 4 096 chains; a real call
// (noinline) costs a third of the speed.  DESIGN.md 6.)
template <bool LANE>
__device__ __forceinline__ void draw_birth(const Chain &c, const uint32_t w[8], int k, Rec &r, int *keep, MapVals *pmv) {
  const DevParams *P = c.P;
  if (k == MPP_K_UBIRTH) {
    r.has_add = 1;
    r.ax = (int)mulhi32(w[3], (uint32_t)c.h.H); r.ay = (int)mulhi32(w[4], (uint32_t)c.h.W);
    r.as = P->maps.vmin[0] + (P->maps.vmax[0] - P->maps.vmin[0]) * u32d(w[5]);
    r.ar = P->maps.vmin[1] + (P->maps.vmax[1] - P->maps.vmin[1]) * u32d(w[6]);
    r.aa = P->maps.vmin[2] + (P->maps.vmax[2] - P->maps.vmin[2]) * u32d(w[7]);
    return;
  }
  if (k == MPP_K_DBIRTH) {
    r.has_add = 1;
#ifdef MPP_PROFILE
    unsigned long long db_t_ = clock64();
#define DBPROF(i) do { unsigned long long n_ = clock64(); if (c.wave == 0 && c.lane == 0) atomicAdd(&g_prof2[i], n_ - db_t_); db_t_ = n_; } while (0)
#else
#define DBPROF(i)
#endif
    const double u = u53(w[3], w[4]), tot = c.L.rowbase ? c.L.rowbase[c.h.H] : c.t.rowbase[c.h.H];
    const double thr = u * tot;                       // cdf <= u <=> partial sum <= u*total
    int row = 0, col = 0;                             // #rows / #columns whose inclusive cdf is <= u (monotone)
    if (LANE) {
      const double *rb = c.L.rowbase ? c.L.rowbase : c.t.rowbase;
      row = count_le_8ary(c.h.H, [&](int i) { return rb[i + 1] <= thr; });
      if (row >= c.h.H) row = c.h.H - 1;
      const double base = rb[row];
      const double *part = c.t.rowpart + (size_t)row * c.h.W;
      col = count_le_8ary(c.h.W, [&](int j) { return (base + part[j]) <= thr; });
    } else {
      // counts of a monotone table: the loads of a chunk are issued together (8 per lane), then counted -- one memory
      // latency per 512 entries instead of one per 64.  The row level sits in LDS when it fits (H <= 1024).
      double base = 0.0;
      auto row_search = [&](auto rb) {
        for (int i0 = 0; i0 < c.h.H; i0 += 8 * WAVE) {
          double v[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) { int i = i0 + k * WAVE + c.lane; v[k] = i < c.h.H ? rb[i + 1] : INFINITY; }
#pragma unroll
          for (int k = 0; k < 8; ++k) row += __popcll(__ballot(v[k] <= thr));
        }
        if (row >= c.h.H) row = c.h.H - 1;
        base = rb[row];
      };
      // the row level in LDS is searched in two steps of one read and one ballot each: lane k looks at the last row of
      // block k (S rows per block, 64 blocks), then the lanes of the block found look at its rows -- the count of a
      // monotone table is the same either way (8 reads and 8 ballots per lane before: 1 300 of the 7 000 cycles of this
      // draw, which is the slowest wave's kernel in 40 % of the rounds)
      if (c.L.rowbase && c.h.H <= WAVE * WAVE) {
        const double *rb = c.L.rowbase;
        const int S = (c.h.H + WAVE - 1) / WAVE;                       // rows per block
        const int last = (c.lane + 1) * S;                             // rb[last] = cdf after the block's last row
        const int blk = __popcll(__ballot(last <= c.h.H && rb[last] <= thr));      // whole blocks at or below the threshold
        const int i = blk * S + c.lane;
        row = blk * S + __popcll(__ballot(c.lane < S && i < c.h.H && rb[i + 1] <= thr));
        if (row >= c.h.H) row = c.h.H - 1;
        base = rb[row];
      } else if (c.L.rowbase) row_search(c.L.rowbase); else row_search(c.t.rowbase);
      DBPROF(13);
      const MPP_GLOBAL double *part = c.t.rowpart + (size_t)row * c.h.W;
      for (int j0 = 0; j0 < c.h.W; j0 += 8 * WAVE) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { int j = j0 + k * WAVE + c.lane; v[k] = j < c.h.W ? part[j] : INFINITY; }
#pragma unroll
        for (int k = 0; k < 8; ++k) col += __popcll(__ballot((base + v[k]) <= thr));
      }
    }
    if (col >= c.h.W) col = c.h.W - 1;
    DBPROF(14);
    r.ax = row; r.ay = col;
    // marks, and on the way the birth density of the drawn point (shape_samplers.py:103-108; same operation
    // order as birth_density())
    const float detv = c.t.det[(size_t)row * c.h.W + col];
    double d = (double)detv / tot;
    const uint32_t w5 = w[5], w6 = w[6], w7 = w[7];
    const bool pre = !LANE && c.t.rm[0] != nullptr;
    TabRows tr{0.0, 0.0};
    if (pre) tr = tab_rows_request(c, ((size_t)row * c.h.W + col) * MPP_NCLASS);
    if (LANE) {
      // (one row after the other: requesting the three rows together, or searching the cumulative tables with 23 probes
      // per level instead of 7, costs more in registers -- 143 spills instead of 106 -- than the two or three memory
      // latencies it saves: 71.8 and 74.9 ms against 67.6 on the config-2 tile, round 3)
#pragma clang loop unroll(disable)
      for (int k = 0; k < 3; ++k) {
        int cls;
        d *= row_prob(c, k, row, col, 0, true, u32d(k == 0 ? w5 : (k == 1 ? w6 : w7)), &cls);
        double val = c.L.edges[k * MPP_NCLASS + cls];
        if (k == 0) r.as = val; else if (k == 1) r.ar = val; else r.aa = val;
      }
    } else {                      // the three rows in lanes 0, 1, 2 at once (see birth_density)
      const int kk = c.lane < 2 ? c.lane : 2;
      int cls_l = 0;
      const double pl = row_prob(c, kk, row, col, 0, true, u32d(kk == 0 ? w5 : (kk == 1 ? w6 : w7)), &cls_l);
      const int c0 = __builtin_amdgcn_readlane(cls_l, 0), c1 = __builtin_amdgcn_readlane(cls_l, 1),
                c2 = __builtin_amdgcn_readlane(cls_l, 2);
      d *= readlane_d(pl, 0); d *= readlane_d(pl, 1); d *= readlane_d(pl, 2);
      r.as = c.L.edges[c0]; r.ar = c.L.edges[MPP_NCLASS + c1]; r.aa = c.L.edges[2 * MPP_NCLASS + c2];
      r.acls = c2;
    }
    r.qf = d * ((double)c.h.H * (double)c.h.W * 32768.0);
    *keep = KEEP_QF | (LANE ? 0 : KEEP_EDGE_ANGLE);
    if (pre) { *pmv = tab_rows_pick(c, tr, detv, Rect{row, col, r.as, r.ar, r.aa}); *keep |= KEEP_MV; }
    DBPROF(15);
    return;
  }
}
template <bool LANE>
__device__ __forceinline__ void draw_proposal(const Chain &c, const uint32_t w[8], int n, Rec &r, int *keep, uint32_t k0, uint32_t k1,
                              uint64_t step, uint32_t chain, MapVals *pmv) {
  const DevParams *P = c.P;
  const int k = draw_head<LANE>(c, w, r, k0, k1, step, chain);
  *keep = 0;
  if (k == MPP_K_UBIRTH || k == MPP_K_DBIRTH) { draw_birth<LANE>(c, w, k, r, keep, pmv); return; }
  if (n == 0) return;
  r.tidx = (int)mulhi32(w[2], (uint32_t)n);
  r.tslot = c.L.order[r.tidx];
  r.has_rem = 1;
  Rect q = load_rect(c.L, r.tslot);
  r.rx = q.x; r.ry = q.y;
  if (k == MPP_K_UDEATH || k == MPP_K_DDEATH) return;
  r.has_add = 1;
  double z0 = 0.0, z1 = 0.0;
  if (k == MPP_K_GTRANS || k == MPP_K_GTRANSF) {     // words (3,4) for the translation, (4,5) for the mark transform
    const uint32_t w3 = w[3], w4 = w[4], w5 = w[5];
    box_muller(k == MPP_K_GTRANS ? w3 : w4, k == MPP_K_GTRANS ? w4 : w5, &z0, &z1);
  }
  if (k == MPP_K_GTRANS) {
    double d0 = P->kern.sigma_trans * z0, d1 = P->kern.sigma_trans * z1;
    int nx = (int)((double)q.x + d0), ny = (int)((double)q.y + d1);
    q.x = min(max(nx, 0), c.h.H - 1); q.y = min(max(ny, 0), c.h.W - 1);
    r.aux0 = d0; r.aux1 = d1;
    *keep = KEEP_TRIG | KEEP_SIZE;
  } else if (k == MPP_K_DTRANS) {
    int ex, ey;
    if (LANE) window_draw_lane(c, q.x, q.y, u53(w[3], w[4]), &ex, &ey);
    else window_draw(c, q.x, q.y, u53(w[3], w[4]), &ex, &ey);
    q.x = ex; q.y = ey;
    *keep = KEEP_TRIG | KEEP_SIZE;
  } else if (k == MPP_K_GTRANSF) {
    int pid = (int)mulhi32(w[3], 3u);
    double d = P->kern.sigma_transform * (P->maps.vmax[pid] - P->maps.vmin[pid]) * z0;
    set_mark(q, pid, wrap_mark(P, pid, mark_of(q, pid) + d));
    r.pid = pid; r.aux0 = d;
    *keep = pid == 2 ? KEEP_SIZE : KEEP_TRIG;
  } else {
    // forward and backward probability come from the row the class is drawn from (transform_kernels.py:203-225): both are
    // taken here, from the one pass over the row, with the sums in the order proposal_densities() forms them
    int pid = (int)mulhi32(w[3], 3u), cls;
    const int oc = value_to_class_tab(P, c.L.edges + pid * MPP_NCLASS, pid, mark_of(q, pid));
    double pb = 0.0;
    const bool pre = !LANE && c.t.rm[0] != nullptr;
    TabRows tr{0.0, 0.0};
    float detv = 0.f;
    if (pre) { detv = c.t.det[(size_t)q.x * c.h.W + q.y]; tr = tab_rows_request(c, ((size_t)q.x * c.h.W + q.y) * MPP_NCLASS); }
    r.qf = row_prob(c, pid, q.x, q.y, 0, true, u32d(w[4]), &cls, oc, &pb);
    r.qb = pb;
    set_mark(q, pid, c.L.edges[pid * MPP_NCLASS + cls]);
    r.pid = pid; r.ncls = cls; r.acls = cls;
    *keep = (pid == 2 ? KEEP_SIZE : KEEP_TRIG) | KEEP_QFB | (pid == 2 ? KEEP_EDGE_ANGLE : 0);
    if (pre) { *pmv = tab_rows_pick(c, tr, detv, q); *keep |= KEEP_MV; }
  }
  // (materialised here through an empty asm: in the traced instantiation the compiler otherwise lost the row of the
  // data-driven transform branch -- r.ax came out as 0 -- when the kernel grew; ROCm 7.2 hipcc)
  int fx = q.x, fy = q.y;
  asm volatile("" : "+v"(fx), "+v"(fy));
  r.ax = fx; r.ay = fy; r.as = q.s; r.ar = q.r; r.aa = q.a;
}

// n-independent parts of the forward / backward proposal probabilities.  The symmetric Gaussian
// kernels have qf == qb, which cancels in the Green ratio: their pdf is evaluated only for traces.
__device__ __forceinline__ void proposal_densities(const Chain &c, Rec &r, bool tracing, int keep, bool coop) {
  const DevParams *P = c.P;
  if (keep & KEEP_QF) { r.qb = 1.0; return; }
  if (keep & KEEP_QFB) return;
  r.qf = 1.0; r.qb = 1.0;
  Rect add{r.ax, r.ay, r.as, r.ar, r.aa};
  switch (r.kernel) {
    case MPP_K_DBIRTH: r.qf = birth_density(c, add, coop); break;
    case MPP_K_DDEATH:
      if (r.has_rem) r.qb = birth_density(c, load_rect(c.L, r.tslot), coop);
      break;
    case MPP_K_GTRANS:
      if (r.has_rem && tracing)
        r.qf = r.qb = normal_pdf(r.aux0, P->kern.sigma_trans) * normal_pdf(r.aux1, P->kern.sigma_trans);
      break;
    case MPP_K_DTRANS:
      if (r.has_rem) {        // transform_kernels.py:94-99: window renormalised around start resp. end
        r.qf = (double)c.t.det[(size_t)r.ax * c.h.W + r.ay] / c.t.boxsum[(size_t)r.rx * c.h.W + r.ry];
        r.qb = (double)c.t.det[(size_t)r.rx * c.h.W + r.ry] / c.t.boxsum[(size_t)r.ax * c.h.W + r.ay];
      }
      break;
    case MPP_K_GTRANSF:
      if (r.has_rem && tracing)
        r.qf = r.qb = normal_pdf(r.aux0, P->kern.sigma_transform * (P->maps.vmax[r.pid] - P->maps.vmin[r.pid]));
      break;
    case MPP_K_DTRANSF:
      if (r.has_rem) {
        Rect old = load_rect(c.L, r.tslot);
        int oc = value_to_class_tab(P, c.L.edges + r.pid * MPP_NCLASS, r.pid, mark_of(old, r.pid));
        const float *row = mark_row_w(c.h.W, c.t, r.pid, old.x, old.y);
        // both classes of the same row: P[new]/sum and P[old]/sum (one row read)
        double tot = 0.0;
        for (int i = 0; i < MPP_NCLASS; ++i) tot += (double)row[i];
        r.qf = (double)row[r.ncls] / tot;
        r.qb = (double)row[oc] / tot;
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
  unsigned short *it = L.cell_items + (size_t)cell * c.h.cell_cap;
  int idx = -1;
  for (int e0 = 0; e0 < cnt && idx < 0; e0 += WAVE) {         // (a cell may hold more entries than a wave has lanes)
    const unsigned long long m = __ballot(e0 + c.lane < cnt && it[e0 + c.lane] == slot);
    if (m) idx = e0 + __ffsll((long long)m) - 1;
  }
  wave_lds_fence();
  if (idx >= 0 && c.lane == 0) {
    it[idx] = it[cnt - 1];
    L.cell_cnt[cell] = (unsigned short)(cnt - 1);
  }
  wave_lds_fence();
}
__device__ void cell_insert(const Chain &c, int cell, int slot, int *err) {
  const Lds &L = c.L;
  int cnt = L.cell_cnt[cell];
  if (cnt >= c.h.cell_cap) { *err = ERR_CELL_OVERFLOW; return; }
  if (c.lane == 0) {
    L.cell_items[(size_t)cell * c.h.cell_cap + cnt] = (unsigned short)slot;
    L.cell_cnt[cell] = (unsigned short)(cnt + 1);
  }
  wave_lds_fence();
}
__device__ void write_slot(const Chain &c, int slot, const Rec &q) {
  const Lds &L = c.L;
  if (c.lane == 0) {
    L.xy[slot] = (q.ax & 0xffff) | (q.ay << 16);
    L.s[slot] = q.as; L.r[slot] = q.ar; L.a[slot] = q.aa;
    L.ca[slot] = q.ca; L.sa[slot] = q.sa; L.hl[slot] = q.hl; L.hw[slot] = q.hw; L.rad[slot] = q.rad;
    L.lin[slot] = q.lin_a; L.gate[slot] = (unsigned char)q.gate_a; L.red0[slot] = q.ra0; L.red1[slot] = q.ra1;
  }
}

// evaluate one step completely (everything but the state mutation): proposal geometry, unit energy,
// dE, and the accept decision for population n at temperature T
#ifdef MPP_PROFILE
#define EPROF(i) do { unsigned long long n_ = clock64(); if (c.wave == 0) prof[i] += n_ - pt_; pt_ = n_; } while (0)
template <bool LANE, bool FAST = false, bool EXT = false>
__device__ void evaluate(const Chain &c, Rec &r, int ri, int keep, int n, double T, bool tracing, bool apply,
                         const MapVals &pmv, unsigned long long *prof) {
  unsigned long long pt_ = clock64();
#else
#define EPROF(i)
template <bool LANE, bool FAST = false, bool EXT = false>
__device__ void evaluate(const Chain &c, Rec &r, int ri, int keep, int n, double T, bool tracing, bool apply,
                         const MapVals &pmv) {
#endif
  const DevParams *P = c.P;
  const Lds &L = c.L;
  // the score-map values of the proposed rectangle: requested first, used after the densities and the trigonometry
  MapVals mv{0.f, 0.f, 0.f, 0.f, 0.0, 0.0, 0.0, 0};
#ifndef MPP_NO_HOIST
  if (keep & KEEP_MV) mv = pmv;
  else if (r.has_add) mv = load_map_vals_w(P, c.h.W, c.t, L.edges, Rect{r.ax, r.ay, r.as, r.ar, r.aa});
#endif
  proposal_densities(c, r, tracing, keep, !LANE);
  EPROF(4);
  r.dE = 0.0; r.n_stash = 0; r.lin_a = 0.0; r.gate_a = 1; r.ra0 = r.ra1 = 0.0;
  r.hl = r.hw = r.ca = r.sa = r.rad = 0.0;
  if (r.has_rem || r.has_add) {
    Rect add{r.ax, r.ay, r.as, r.ar, r.aa};
    Geo2 ag;
    ag.g.x = add.x; ag.g.y = add.y; ag.g.hl = ag.g.hw = ag.g.ca = ag.g.sa = 0.0; ag.rad = 0.0;
    if (r.has_add) {
      if (keep & KEEP_SIZE) { ag.g.hl = L.hl[r.tslot]; ag.g.hw = L.hw[r.tslot]; ag.rad = L.rad[r.tslot]; }
      else {
        double length = (2.0 * add.s) / (1.0 + add.r), width = add.r * length;
        ag.g.hl = length / 2.0; ag.g.hw = width / 2.0;
        ag.rad = geo_radius(ag.g);
      }
      if (keep & KEEP_TRIG) { ag.g.ca = L.ca[r.tslot]; ag.g.sa = L.sa[r.tslot]; }
      else if (keep & KEEP_EDGE_ANGLE) { ag.g.ca = L.trig[r.acls]; ag.g.sa = L.trig[MPP_NCLASS + r.acls]; }
      else { double al = add.a + MPP_PI / 2.0; ag.g.ca = cos(al); ag.g.sa = sin(al); }
      EPROF(5);
#ifdef MPP_NO_HOIST
      mv = load_map_vals_w(P, c.h.W, c.t, L.edges, add);
#endif
      unit_part_mv<EXT>(P, c.t, mv, add, ag.g, &r.lin_a, &r.gate_a, nullptr, !LANE);
      r.hl = ag.g.hl; r.hw = ag.g.hw; r.ca = ag.g.ca; r.sa = ag.g.sa; r.rad = ag.rad;
      EPROF(6);
    }
    if (LANE)
      r.dE = eval_delta_lane(c, ri, r.has_rem ? r.tslot : -1, r.has_add != 0, add, ag, r.lin_a, r.gate_a, &r.ra0, &r.ra1,
                             &r.n_stash, apply);
    else
      r.dE = eval_delta<FAST>(c, ri, r.has_rem ? r.tslot : -1, r.has_add != 0, add, ag, r.lin_a, r.gate_a, &r.ra0, &r.ra1,
                              &r.n_stash, apply);
    EPROF(7);
  }
  double fwd, bwd;
  green_terms(P, r, n, c.t.intensity, &fwd, &bwd);
  // rjmcmc.py:105-113: accept <=> log(u+eps) < -dE/T + log(bwd+eps) - log(fwd+eps)
  //                           <=> u+eps < exp(-dE/T) * (bwd+eps)/(fwd+eps)      (one exp instead of three logs)
  double ratio = (bwd + EPS_GREEN) / (fwd + EPS_GREEN);
  r.accepted = (P->force_accept || (r.u_acc + EPS_GREEN) < exp(-r.dE / T) * ratio) ? 1 : 0;
  if (tracing) { r.fwd = fwd; r.bwd = bwd; r.log_alpha = (-r.dE / T) + log(bwd + EPS_GREEN) - log(fwd + EPS_GREEN); }
  EPROF(8);
}

