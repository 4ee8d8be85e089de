// mpp_deep.hip -- the "deep round" chain kernel: every LANE evaluates one speculative step.
//
// Same chain, same bits as mpp_chain_kernel (mpp_sampler.hip) -- the reference's loop rjmcmc.py:83-164 with the
// proposals drawn from Philox by step index -- organised for what the chain actually does after its first few thousand
// steps: of the ~25 % of proposals the Metropolis test accepts, four out of five re-write a point with the very values it
// already has (a data-driven transform that draws the class the mark sits in, a translation onto the same pixel), and
// only 2-5 % of all steps change the configuration.  A round of 8 steps, one wave each, is then almost always all
// rejects -- and costs a wave's latency per step.  Here a round is up to 64 * WAVES steps:
//   A  the kernel TYPE of each of the round's steps is a function of its Philox words alone; the steps are counting-
//      sorted by type across the workgroup so that the lanes of one wave run (mostly) the same kernel and do not
//      serialise on the draw;
//   B  every active lane draws and evaluates its step on its own against the round's start state (the lane forms of
//      draw_proposal / evaluate: the arithmetic, and its order, are those of the wave forms), decides it, and reports
//      (valid, changes-the-state, slot, positions) in 12 bytes of LDS;
//   C  every wave takes the same commit decision from those reports: steps commit in order; a committed change
//      invalidates the later steps that could have seen it (same slot, same cell, within 2 * max_inter) and the round ends
//      at the first invalid step; an accepted step that writes the bits that are already there changes nothing and ends
//      nothing;
//   D  the few lanes whose step commits with a change evaluate it a second time, writing the cached reductions of the
//      neighbours directly (no stash), and after a barrier move the point between cell lists and re-write its slot.
// The depth of a round follows the number of steps the last rounds committed.  Temperatures are a function of the step
// index alone: a ring in LDS holds those of the next steps, extended by as many entries as a round commits.
#include <cstdlib>

#include "mpp_chain.hpp"

#define DEEP_NMAX_LIMIT 256      // most steps of one round (deeper rounds commit no more: the first conflict ends them)
#define DEEP_CLIST 192         // candidate neighbours a wave collects before it evaluates them (>= 64: one cell's entries fit)
#ifdef MPP_DEEP_PROF
// diagnostic build only (profiles/tools/build_deep_prof.sh): cycles of wave 0 per phase of a round, summed over the launch,
// in stats[16 + 24 * wave + phase]
#define DPH_N 24
#define DPH_T0() dph_t_ = clock64()
#define DPH(i) do { const unsigned long long n_ = clock64(); dph_[i] += n_ - dph_t_; dph_t_ = n_; } while (0)
#define DPH_ARGS , unsigned long long *dph_, unsigned long long &dph_t_
#define DPH_PASS , dph_, dph_t_
#else
#define DPH_T0()
#define DPH(i)
#define DPH_ARGS
#define DPH_PASS
#endif
// info word 0: bits 0..15 slot, then flags
#define DI_VALID (1 << 16)
#define DI_CHG (1 << 17)
#define DI_HR (1 << 18)
#define DI_HA (1 << 19)
#define DI_BAD (1 << 20)     // the proposal could not be formed (reported when the commit reaches it)
#define DI_ACC (1 << 21)
#define DI_FULL (1 << 22)    // the cell the step adds to is full (capacity check, made against the round's start state)
#define DI_NB (1 << 23)      // the step changes cached reductions of neighbours: committing it needs the second pass
#define DI_NBOV (1 << 24)    // ... of more than two neighbours (their positions are not all reported)
#define DI_NB2 (1 << 26)     // ... of at least two neighbours
#define DI_RESC (1 << 25)    // the evaluation re-reduced a neighbour: it looked two interaction ranges away

struct DeepLds {
  uint4 *info;                 // [nmax] (flags | slot, removed xy, added xy, cell coordinates) -- indexed by the step's offset in the round
  uint4 *nb;                   // [nmax] positions (and circumradii, rounded up) of the (at most two) neighbours whose cached
                               // reductions the step changes
  double *st;                  // [nmax][5] ... and their new reductions (2 x 2 values, then the two slots as bits): a step that
                               // commits writes them; only a step that changes more than two neighbours needs a second pass
  uint4 *pw;                   // [nmax] Philox block 0 of the steps, in sorted order
  unsigned short *poff;        // [nmax] sorted position -> offset of the step in the round
  unsigned short *tcnt;        // [WAVES][16] steps of each kernel type per wave
  double *tring;               // [4 * nmax] temperature of step (offset & mask), filled two rounds ahead
  unsigned long long *racc;    // [WAVES][3][64] per step of a wave: max of the overlaps / min of the alignments with the added point;
                               // a candidate neighbour of the step has a non-finite energy (classic image energies only)
  unsigned int *clist;         // [WAVES][DEEP_CLIST] (step << 16 | slot): the neighbours in range of a wave's steps, in order
  unsigned char *ltab;         // [WAVES][128] the 3 x 3 blocks of cells a wave's steps look at (lane | 0x80: the added point's)
};
__host__ __device__ inline size_t deep_extra_bytes(int nmax, int waves, int ext) {
  return (size_t)nmax * 16 + (size_t)4 * nmax * 8 + (size_t)waves * (2 + (ext ? 1 : 0)) * 64 * 8 + (size_t)waves * DEEP_CLIST * 4 + (size_t)nmax * 32 + (size_t)nmax * 40 +
         (size_t)nmax * 2 + (size_t)waves * 16 * 2 + (size_t)waves * 128 + 64;
}
__host__ __device__ inline size_t deep_base_bytes(int cap, int ncell, int cell_cap, int rowbase_n, int waves) {
  return (lds_bytes(cap, ncell, cell_cap, 0, rowbase_n, waves) + 15) & ~(size_t)15;
}
__device__ inline DeepLds deep_carve(unsigned char *base, int nmax, int waves, int ext) {
  DeepLds D;
  D.pw = (uint4 *)base; base += (size_t)nmax * 16;
  D.tring = (double *)base; base += (size_t)4 * nmax * 8;
  D.racc = (unsigned long long *)base; base += (size_t)waves * (2 + (ext ? 1 : 0)) * 64 * 8;
  D.clist = (unsigned int *)base; base += (size_t)waves * DEEP_CLIST * 4;
  D.info = (uint4 *)base; base += (size_t)nmax * 16;
  D.nb = (uint4 *)base; base += (size_t)nmax * 16;
  D.st = (double *)base; base += (size_t)nmax * 40;
  D.poff = (unsigned short *)base; base += (size_t)nmax * 2;
  D.tcnt = (unsigned short *)base; base += (size_t)waves * 16 * 2;
  D.ltab = base;
  return D;
}

template <bool IN_LDS>
__device__ __forceinline__ const DevParams *deep_stage_params(const DevParams &Pv, int nthr) {
  if constexpr (IN_LDS) {
    __shared__ DevParams s_P;
    const int *src = (const int *)&Pv;
    int *dst = (int *)&s_P;
    for (int i = threadIdx.x; i < (int)(sizeof(DevParams) / 4); i += nthr) dst[i] = src[i];
    __syncthreads();
    return &s_P;
  } else {
    return &Pv;
  }
}

// ---- state mutation by ONE lane (energy_point_set.py:118-154); two steps that commit in the same round touch different
// cells and slots
__device__ __forceinline__ void cell_remove_1(const Chain &c, int cell, int slot) {
  const Lds &L = c.L;
  const int cnt = L.cell_cnt[cell];
  unsigned short *it = L.cell_items + (size_t)cell * c.h.cell_cap;
  int idx = -1;
  for (int i = 0; i < cnt; ++i) if (idx < 0 && it[i] == slot) idx = i;
  if (idx >= 0) {
    it[idx] = it[cnt - 1];
    L.cell_cnt[cell] = (unsigned short)(cnt - 1);
  }
}
__device__ __forceinline__ void cell_insert_1(const Chain &c, int cell, int slot) {
  const Lds &L = c.L;
  const int cnt = L.cell_cnt[cell];          // (room was checked when the step was chosen)
  L.cell_items[(size_t)cell * c.h.cell_cap + cnt] = (unsigned short)slot;
  L.cell_cnt[cell] = (unsigned short)(cnt + 1);
}
__device__ __forceinline__ void write_slot_1(const Chain &c, int slot, const Rec &q) {
  const Lds &L = c.L;
  L.xy[slot] = (q.ax & 0xffff) | (q.ay << 16);
  L.s[slot] = q.as; L.r[slot] = q.ar; L.a[slot] = q.aa;
  L.ca[slot] = q.ca; L.sa[slot] = q.sa; L.hl[slot] = q.hl; L.hw[slot] = q.hw; L.rad[slot] = q.rad;
  L.lin[slot] = q.lin_a; L.gate[slot] = (unsigned char)q.gate_a; L.red0[slot] = q.ra0; L.red1[slot] = q.ra1;
}

// evaluate() of mpp_chain.hpp in its lane form, in two halves around the ONE call of eval_delta_lane the kernel has (the
// step's evaluation and, for a step that commits, the second pass that writes the neighbours' reductions go through the
// same call site: with two, the inliner leaves a real call behind and the chain state lives in scratch memory)
// the geometry of the rectangle a step adds (what of it the step keeps from its target comes from the target's cached values)
__device__ __forceinline__ void deep_add_geo(const Chain &c, Rec &r, int keep) {
  const Lds &L = c.L;
  r.hl = r.hw = r.ca = r.sa = r.rad = 0.0;
  if (r.has_add) {
    Geo g;
    g.x = r.ax; g.y = r.ay; g.hl = g.hw = g.ca = g.sa = 0.0;
    double rad = 0.0;
    if (keep & KEEP_SIZE) { g.hl = L.hl[r.tslot]; g.hw = L.hw[r.tslot]; rad = L.rad[r.tslot]; }
    else {
      double length = (2.0 * r.as) / (1.0 + r.ar), width = r.ar * length;
      g.hl = length / 2.0; g.hw = width / 2.0;
      rad = geo_radius(g);
    }
    if (keep & KEEP_TRIG) { g.ca = L.ca[r.tslot]; g.sa = L.sa[r.tslot]; }
    else if (keep & KEEP_EDGE_ANGLE) { g.ca = L.trig[r.acls]; g.sa = L.trig[MPP_NCLASS + r.acls]; }
    else { double al = r.aa + MPP_PI / 2.0; g.ca = cos(al); g.sa = sin(al); }
    r.hl = g.hl; r.hw = g.hw; r.ca = g.ca; r.sa = g.sa; r.rad = rad;
  }
}
// `contrast_pre`: the ContrastEnergy term of the added rectangle, computed by the whole wave beforehand (or nullptr)
template <bool EXT>
__device__ __forceinline__ void deep_pre(const Chain &c, Rec &r, int keep, bool tracing, const MapVals &pmv, const double *contrast_pre) {
  const DevParams *P = c.P;
  const Lds &L = c.L;
  MapVals mv{0.f, 0.f, 0.f, 0.f, 0.0, 0.0, 0.0, 0};
  if (keep & KEEP_MV) mv = pmv;
  else if (r.has_add) mv = load_map_vals_w(P, c.h.W, c.t, L.edges, Rect{r.ax, r.ay, r.as, r.ar, r.aa});
  proposal_densities(c, r, tracing, keep, false);
  r.dE = 0.0; r.n_stash = 0; r.lin_a = 0.0; r.gate_a = 1; r.ra0 = r.ra1 = 0.0;
  if (r.has_add) {
    Rect add{r.ax, r.ay, r.as, r.ar, r.aa};
    Geo g;
    g.x = add.x; g.y = add.y; g.hl = r.hl; g.hw = r.hw; g.ca = r.ca; g.sa = r.sa;
    unit_part_mv<EXT>(P, c.t, mv, add, g, &r.lin_a, &r.gate_a, nullptr, false, contrast_pre);
  }
}
__device__ __forceinline__ void deep_post(const Chain &c, Rec &r, int n, double T, bool tracing) {
  double fwd, bwd;
  green_terms(c.P, r, n, c.t.intensity, &fwd, &bwd);
  // rjmcmc.py:105-113 (one exp instead of three logs, see evaluate())
  double ratio = (bwd + EPS_GREEN) / (fwd + EPS_GREEN);
  r.accepted = ((r.u_acc + EPS_GREEN) < exp(-r.dE / T) * ratio) ? 1 : 0;
  if (tracing) { r.fwd = fwd; r.bwd = bwd; r.log_alpha = (-r.dE / T) + log(bwd + EPS_GREEN) - log(fwd + EPS_GREEN); }
}

// inclusive prefix sum over the 64 lanes (DPP: shifts within the rows of 16, then the two row broadcasts); EXEC must be full
__device__ __forceinline__ int wave_incl_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);     // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);     // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);     // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);     // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);     // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);     // row_bcast:31 -> rows 2, 3
  return v;
}
__device__ __forceinline__ double shfl_d(double v, int lane) {
  const long long b = __double_as_longlong(v);
  const int lo = __shfl((int)(b & 0xffffffffll), lane, WAVE), hi = __shfl((int)(b >> 32), lane, WAVE);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// clip_area_wave() for up to CLIP_SLOTS rectangle pairs at once: the wave works in groups of 16 lanes, lane li < 8 of a group
// owns vertex li of its group's polygon (the group's two polygon buffers are those of one clip slot).  Every vertex goes
// through exactly the arithmetic of clip_area() / clip_area_wave(), the shoelace sum runs in vertex order: the same bits.
// A data-driven birth lands on an object, i.e. on the rectangle that already sits there: nearly every birth asks for a
// clip, ten per wave and round -- one after the other they were a third of the neighbour evaluation.
// `gact`: my group has a pair; sx .. cy: its subject and clip corners (the same in all lanes of the group).
__device__ __forceinline__ double clip_area_groups(const Chain &c, bool gact, const double *sx, const double *sy, const double *cx,
                                                   const double *cy) {
  const int g = c.lane >> 4, li = c.lane & 15;
  double *buf = c.L.clip + ((size_t)c.wave * CLIP_SLOTS + g) * 32;
  double *ax = buf, *ay = buf + 8, *bx = buf + 16, *by = buf + 24;
  if (li < 4) {
    ax[li] = li == 0 ? sx[0] : (li == 1 ? sx[1] : (li == 2 ? sx[2] : sx[3]));
    ay[li] = li == 0 ? sy[0] : (li == 1 ? sy[1] : (li == 2 ? sy[2] : sy[3]));
  }
  wave_lds_fence();
  const unsigned int belowg = (1u << li) - 1u;
  int na = gact ? 4 : 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const double x0 = cx[e], y0 = cy[e], x1 = cx[(e + 1) & 3], y1 = cy[(e + 1) & 3];
    const double ex = x1 - x0, ey = y1 - y0;
    const bool valid = li < na;
    const int ii = valid ? li : 0, pi = ii == 0 ? (na > 0 ? na - 1 : 0) : ii - 1;
    const double qx = ax[ii], qy = ay[ii], px = ax[pi], py = ay[pi];
    const double sq = ex * (qy - y0) - ey * (qx - x0), sp = ex * (py - y0) - ey * (px - x0);
    const bool e_int = valid && (sq >= 0 ? sp < 0 : sp >= 0), e_q = valid && sq >= 0;
    const unsigned int mi = (unsigned int)(__ballot(e_int) >> (16 * g)) & 0xffffu, mq = (unsigned int)(__ballot(e_q) >> (16 * g)) & 0xffffu;
    int off = __popc(mi & belowg) + __popc(mq & belowg);
    if (e_int) {
      if (off < 8) {
        double t = sp / (sp - sq);
        bx[off] = px + t * (qx - px); by[off] = py + t * (qy - py);
      }
      ++off;
    }
    if (e_q && off < 8) { bx[off] = qx; by[off] = qy; }
    const int nb = __popc(mi) + __popc(mq);
    na = nb < 8 ? nb : 8;
    double *tx = ax, *ty = ay;
    ax = bx; ay = by; bx = tx; by = ty;
    wave_lds_fence();
  }
  const bool valid = li < na;
  const int ii = valid ? li : 0, jj = ii + 1 >= na ? 0 : ii + 1;
  if (li < 8) bx[li] = valid ? ax[ii] * ay[jj] - ax[jj] * ay[ii] : 0.0;
  wave_lds_fence();
  double t_[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) t_[i] = bx[i];
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) if (i < na) s += t_[i];
  wave_lds_fence();
  return na < 3 ? 0.0 : 0.5 * fabs(s);
}

// The neighbours' part of dE for ALL steps a wave evaluates (energy_graph.py:139-225; the arithmetic of eval_delta<FAST>,
// mpp_chain.hpp, per neighbour, summed in the same order).  Lane i leads step i (`lead`: it has a point to remove and / or a
// rectangle to add).  One lane walking the 3 x 3 cells around its own step's points is a chain of dependent LDS reads per
// cell and entry (78 k cycles a round); here the work of all the wave's steps is dealt to all 64 lanes, twice:
//   1  the cells to look at -- the 3 x 3 block around a step's removed point, then, unless it is the same block, the one
//      around its added point -- go to the lanes, 128 at a time (two neighbouring cells per lane); a lane looks at its
//      cells' entries, four at a time, and keeps those within the longest interaction range of the removed or the added
//      point; an exclusive prefix sum puts them into the wave's candidate list in (step, cell, entry) order -- the order
//      eval_delta sums in;
//   2  the list goes to the lanes, 64 (step, neighbour) pairs at a time: the neighbour's geometry and cached reductions from
//      LDS, the step's added rectangle from its leader's registers (ds_bpermute); rectangle clips and the re-reduction of a
//      neighbour that loses its extremum are done by the whole wave, one after the other, as in eval_delta<FAST>; the few
//      neighbours whose energy changes are added to their step's sum in list order; the reductions of the added point
//      itself (max / min: order-free) are LDS atomics on the bit patterns.
// `apply` (wave-uniform): the changed reductions are written to the caches (the second pass of a step that commits).
// Model: pair 0 = rectangle overlap / max, pair 1 = alignment / min (FAST).
template <bool EXT>
__device__ __forceinline__ void deep_delta(const Chain &c, const DeepLds &D, bool lead, bool has_rem, bool has_add, int rem,
                                           int rxy, int axy, double a_s, double a_r, double a_a, double a_hl, double a_hw,
                                           double a_ca, double a_sa, double a_rad, bool apply, double *sum_out, double *ra0_out,
                                           double *ra1_out, int *nchg_out, int *nb0_out, int *nb1_out, int *nbr_out, int *nresc_out, int *su_out, double *sv_out, bool *nonfinite_out DPH_ARGS) {
  const Lds &L = c.L;
  unsigned int *clist = D.clist + (size_t)c.wave * DEEP_CLIST;
  unsigned long long *racc = D.racc + (size_t)c.wave * (EXT ? 192 : 128);
  const int flags = (lead ? 4 : 0) | (lead && has_rem ? 1 : 0) | (lead && has_add ? 2 : 0);
  const int maxd2_0 = c.pr0.maxd2, maxd2_1 = c.pr1.maxd2, range2 = maxd2_0 > maxd2_1 ? maxd2_0 : maxd2_1;
  const double rew = c.pr1.p0 != 0.0 ? 1.0 : 0.0;
  const unsigned long long below = (1ull << c.lane) - 1ull;
  // the 3 x 3 blocks to look at: around the removed point, then -- unless it is the same block -- around the added one;
  // ptab lists them in (step, removed before added) order: lane | 0x80 for an added point's block
  unsigned char *ptab = D.ltab + (size_t)c.wave * 128;
  int npos = 0, ntasks = 0;
  {
    const int rcell_i = cell_coord(c, rxy & 0xffff), rcell_j = cell_coord(c, (rxy >> 16) & 0xffff);
    const int acell_i = cell_coord(c, axy & 0xffff), acell_j = cell_coord(c, (axy >> 16) & 0xffff);
    const bool pa_ = lead && has_rem, pb_ = lead && has_add && !(has_rem && rcell_i == acell_i && rcell_j == acell_j);
    const int cnt_ = (pa_ ? 1 : 0) + (pb_ ? 1 : 0);
    const int incl = wave_incl_scan(cnt_);
    int pos = incl - cnt_;
    if (pa_) ptab[pos++] = (unsigned char)c.lane;
    if (pb_) ptab[pos] = (unsigned char)(c.lane | 0x80);
    npos = __builtin_amdgcn_readlane(incl, 63);
    ntasks = 9 * npos;
  }
  racc[c.lane] = 0ull; racc[64 + c.lane] = 0ull;
  if (EXT) racc[128 + c.lane] = 0ull;
  wave_lds_fence();
  double sum = 0.0;
  int nchg = 0, M = 0, t0 = 0, nb0 = 0, nb1 = 0, nbr = 0, nresc = 0, su = 0;
  double sv00 = 0.0, sv01 = 0.0, sv10 = 0.0, sv11 = 0.0;       // new reductions of the first two neighbours that change
  bool pending = false;
  int p_cnt = 0, p_base[2] = {0, 0}, p_i[2] = {0, 0};
  unsigned long long p_mask[2] = {0ull, 0ull};
  for (;;) {
    bool final = false;
    if (__ballot(pending) == 0ull) {
      if (t0 >= ntasks) final = true;
      else {
        // ---- 1: the next 128 (block, cell) pairs, two neighbouring ones per lane
        p_cnt = 0;
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
          const int t = t0 + 2 * c.lane + q2;
          const int ip = (t * 7282) >> 16, k = t - 9 * ip;          // t / 9, t % 9 (t < 1152)
          const int pe = t < ntasks ? (int)ptab[ip] : 0;            // the block's step and kind
          const int i = pe & 63;
          const bool second = (pe & 0x80) != 0;
          const int sfl = __shfl(flags, i, WAVE), srem = __shfl(rem, i, WAVE), srxy = __shfl(rxy, i, WAVE), saxy = __shfl(axy, i, WAVE);
          const bool s_hr = sfl & 1, s_ha = sfl & 2;
          const int rx = srxy & 0xffff, ry = (srxy >> 16) & 0xffff, ax = saxy & 0xffff, ay = (saxy >> 16) & 0xffff;
          const int cir = cell_coord(c, rx), cjr = cell_coord(c, ry), cia = cell_coord(c, ax), cja = cell_coord(c, ay);
          const int k3 = (k * 11) >> 5;                              // k / 3 for k < 9
          const int ci = (second ? cia : cir) + k3 - 1, cj = (second ? cja : cjr) + (k - 3 * k3) - 1;
          bool ok = t < ntasks;
          if (ok && second && s_hr && abs(ci - cir) <= 1 && abs(cj - cjr) <= 1) ok = false;      // already listed
          ok = ok && ci >= 0 && ci < c.h.nx && cj >= 0 && cj < c.h.ny;
          const int cell = ok ? cj + ci * c.h.ny : 0;
          const int cnt = ok ? (int)L.cell_cnt[cell] : 0;
          const int base = cell * c.h.cell_cap;
          unsigned long long mask = 0ull;
          for (int e0 = 0; __ballot(e0 < cnt) != 0ull; e0 += 4) {
            int u_[4], xy_[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) u_[q] = e0 + q < cnt ? (int)L.cell_items[base + e0 + q] : 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) xy_[q] = e0 + q < cnt ? L.xy[u_[q]] : 0;
            if (EXT) {
              // a candidate neighbour (ANY entry of the cells looked at, as eval_delta's generic form does) whose own energy is
              // not finite makes the reference's E(after) - E(before) over the neighbourhood inf - inf = NaN
#pragma unroll
              for (int q = 0; q < 4; ++q)
                if (e0 + q < cnt && u_[q] != (s_hr ? srem : -1)) {
                  const double lu = L.lin[u_[q]];
                  if (!(lu - lu == 0.0)) racc[128 + i] = 1ull;
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int ux = xy_[q] & 0xffff, uy = (xy_[q] >> 16) & 0xffff;
              const int dxr = ux - rx, dyr = uy - ry, dxa = ux - ax, dya = uy - ay;
              const bool inr = (s_hr && dxr * dxr + dyr * dyr <= range2) || (s_ha && dxa * dxa + dya * dya <= range2);
              if (e0 + q < cnt && u_[q] != (s_hr ? srem : -1) && inr) mask |= 1ull << (e0 + q);
            }
          }
          p_mask[q2] = mask; p_base[q2] = base; p_i[q2] = i;
          p_cnt += __popcll(mask);
        }
        t0 += 2 * WAVE;
        pending = p_cnt > 0;
        DPH(12);
      }
    }
    bool flush = final;
    if (!final) {
      // append the pairs of the waiting lanes, in lane order, as far as the list has room
      const int c_ = pending ? p_cnt : 0;
      const int incl = wave_incl_scan(c_);
      const bool fit = pending && M + incl <= DEEP_CLIST;
      const unsigned long long fm = __ballot(fit);
      if (fit) {
        int pos = M + incl - c_;
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
          unsigned long long m = p_mask[q2];
          while (m) {
            const int e = __ffsll((long long)m) - 1;
            m &= m - 1;
            clist[pos++] = ((unsigned int)p_i[q2] << 16) | (unsigned int)L.cell_items[p_base[q2] + e];
          }
        }
        pending = false;
      }
      if (fm) M = __builtin_amdgcn_readlane(M + incl, 63 - __clzll((long long)fm));
      flush = __ballot(pending) != 0ull;                   // what is left did not fit: evaluate the list first
      DPH(13);
    }
    if (flush && M > 0) {
      wave_lds_fence();
      // ---- 2: the list, 64 (step, neighbour) pairs at a time
      for (int j0 = 0; j0 < M; j0 += WAVE) {
        const int j = j0 + c.lane;
        const bool act = j < M;
        const unsigned int cu = act ? clist[j] : 0u;
        const int i = (int)(cu >> 16), u = (int)(cu & 0xffffu);
        const int sfl = __shfl(flags, i, WAVE), srem = __shfl(rem, i, WAVE), saxy = __shfl(axy, i, WAVE);
        const bool s_hr = act && (sfl & 1), s_ha = act && (sfl & 2);
        Geo2 ag;
        ag.g.x = saxy & 0xffff; ag.g.y = (saxy >> 16) & 0xffff;
        ag.g.hl = shfl_d(a_hl, i); ag.g.hw = shfl_d(a_hw, i); ag.g.ca = shfl_d(a_ca, i); ag.g.sa = shfl_d(a_sa, i);
        ag.rad = shfl_d(a_rad, i);
        Geo2 gr, gu;
        gr.g.x = gr.g.y = 0; gr.g.hl = gr.g.hw = gr.g.ca = gr.g.sa = 0.0; gr.rad = 0.0;
        gu = gr;
        double ov0 = 0.0, ov1 = 0.0;
        if (s_hr) gr = load_geo(L, srem);
        if (act) { gu = load_geo(L, u); ov0 = L.red0[u]; ov1 = L.red1[u]; }
        int d2r = 0, d2a = 0;
        if (s_hr) { const int dx = gu.g.x - gr.g.x, dy = gu.g.y - gr.g.y; d2r = dx * dx + dy * dy; }
        if (s_ha) { const int dx = gu.g.x - ag.g.x, dy = gu.g.y - ag.g.y; d2a = dx * dx + dy * dy; }
        // marks of the added rectangle: only the order of two rectangles on the same pixel asks for them
        const bool tie_a = s_ha && gu.g.x == ag.g.x && gu.g.y == ag.g.y;
        double sa_s = 0.0, sa_r = 0.0, sa_a = 0.0;
        if (__ballot(tie_a) != 0ull) { sa_s = shfl_d(a_s, i); sa_r = shfl_d(a_r, i); sa_a = shfl_d(a_a, i); }
        DPH(14);
        // -- overlaps first, in uniform control flow: every pair whose circumscribed circles meet is clipped by the wave
        const bool in_r0 = s_hr && d2r <= maxd2_0 && ov0 != 0.0, in_a0 = s_ha && d2a <= maxd2_0;
        const double Au = geo_area(gu.g);
        double ovl_r = 0.0, ovl_a = 0.0;
#pragma clang loop unroll(disable)
        for (int which = 0; which < 2; ++which) {
          const Geo2 gv = which == 0 ? gr : ag;
          bool need = which == 0 ? in_r0 : in_a0;
          double mn = 0.0;
          bool uf = false;
          if (need) {
            const double B = geo_area(gv.g), reach = gu.rad + gv.rad, d2 = (double)(which == 0 ? d2r : d2a);
            mn = Au < B ? Au : B;
            need = !(mn < DEGENERATE_AREA) && !(d2 > reach * reach * 1.0000001);
            if (need) {
              if (which == 0) uf = slot_first(L, u, gu.g, gv.g.x, gv.g.y, L.s[srem], L.r[srem], L.a[srem]);
              else uf = slot_first(L, u, gu.g, gv.g.x, gv.g.y, sa_s, sa_r, sa_a);
            }
          }
          double val = 0.0;
          unsigned long long m = __ballot(need);
          while (m) {                                              // CLIP_SLOTS pairs at a time, one per group of 16 lanes
            const int rank = __popcll(m & below);                  // my place among the lanes that still wait
            int my_src = -1;
#pragma unroll
            for (int k = 0; k < CLIP_SLOTS; ++k) {
              const int sk = m ? __ffsll((long long)m) - 1 : -1;
              if (m) m &= m - 1;
              if ((c.lane >> 4) == k) my_src = sk;
            }
            const bool gact = my_src >= 0;
            const int sl = gact ? my_src : 0;
            Geo bu, bv;
            bu.x = __shfl(gu.g.x, sl, WAVE); bu.y = __shfl(gu.g.y, sl, WAVE);
            bu.hl = shfl_d(gu.g.hl, sl); bu.hw = shfl_d(gu.g.hw, sl); bu.ca = shfl_d(gu.g.ca, sl); bu.sa = shfl_d(gu.g.sa, sl);
            bv.x = __shfl(gv.g.x, sl, WAVE); bv.y = __shfl(gv.g.y, sl, WAVE);
            bv.hl = shfl_d(gv.g.hl, sl); bv.hw = shfl_d(gv.g.hw, sl); bv.ca = shfl_d(gv.g.ca, sl); bv.sa = shfl_d(gv.g.sa, sl);
            const bool u_first = __shfl((int)uf, sl, WAVE) != 0;
            double ux[4], uy[4], vx[4], vy[4];
            geo_corners(bu, ux, uy); geo_corners(bv, vx, vy);
            double sx[4], sy[4], cx[4], cy[4];          // subject = the smaller rectangle in the canonical order
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              sx[q] = u_first ? ux[q] : vx[q]; sy[q] = u_first ? uy[q] : vy[q];
              cx[q] = u_first ? vx[q] : ux[q]; cy[q] = u_first ? vy[q] : uy[q];
            }
            const double area = clip_area_groups(c, gact, sx, sy, cx, cy);
            const double mine_ = shfl_d(area, (rank & (CLIP_SLOTS - 1)) << 4);
            if (need && rank < CLIP_SLOTS) { val = mine_ / (mn + AREA_EPS); need = false; }
          }
          if (which == 0) ovl_r = val; else ovl_a = val;
        }
        DPH(15);
        // -- the two reductions of the neighbour (MPP_PAIR_P of eval_delta)
        double nv0 = ov0, nv1 = ov1;
        bool resc0 = false, resc1 = false;
        if (act) {
          {                                                        // pair 0: overlap, max
            bool carries = false;
            if (in_r0) carries = ovl_r == ov0;
            if (in_a0 && ovl_a > 0.0) atomicMax(&racc[i], (unsigned long long)__double_as_longlong(ovl_a));
            if (carries) { if (in_a0 && ovl_a >= ov0) nv0 = ovl_a; else resc0 = true; }
            else if (in_a0) nv0 = ov0 > ovl_a ? ov0 : ovl_a;
          }
          {                                                        // pair 1: alignment, min
            const bool in_r1 = s_hr && d2r <= maxd2_1 && ov1 != 0.0, in_a1 = s_ha && d2a <= maxd2_1;
            bool carries = false;
            double v_add = 0.0;
            if (in_r1) carries = (1.0 - fabs(gu.g.ca * gr.g.ca + gu.g.sa * gr.g.sa) - rew) == ov1;
            if (in_a1) {
              v_add = 1.0 - fabs(gu.g.ca * ag.g.ca + gu.g.sa * ag.g.sa) - rew;
              if (v_add < 0.0) atomicMax(&racc[64 + i], (unsigned long long)__double_as_longlong(v_add));
            }
            if (carries) { if (in_a1 && v_add <= ov1) nv1 = v_add; else resc1 = true; }
            else if (in_a1) nv1 = ov1 < v_add ? ov1 : v_add;
          }
        }
        // -- a neighbour lost the point that carried its extremum and the added point does not take over: the wave
        //    re-reduces it over its own 3 x 3 cells (lane 3k+e on entry e of cell k; fuller cells: 64 entries per turn)
        unsigned long long rm = __ballot(resc0 || resc1);
        const int ck = c.lane / 3, ce = c.lane - 3 * ck;
        while (rm) {
          const int src = __ffsll((long long)rm) - 1;
          rm &= rm - 1;
          const bool need0 = __builtin_amdgcn_readlane((int)resc0, src) != 0, need1 = __builtin_amdgcn_readlane((int)resc1, src) != 0;
          const int us = __builtin_amdgcn_readlane(u, src), is = __builtin_amdgcn_readlane(i, src);
          if (c.lane == is) nresc += 1;
          const int rem_s = __builtin_amdgcn_readlane(srem, src);              // (the neighbour lost a removed point: has_rem)
          const bool ha_s = __builtin_amdgcn_readlane((int)s_ha, src) != 0;
          Geo2 bu, ba;                                       // the neighbour and its step's added rectangle, wave-uniform
          bu.g.x = __builtin_amdgcn_readlane(gu.g.x, src); bu.g.y = __builtin_amdgcn_readlane(gu.g.y, src);
          bu.g.hl = readlane_d(gu.g.hl, src); bu.g.hw = readlane_d(gu.g.hw, src);
          bu.g.ca = readlane_d(gu.g.ca, src); bu.g.sa = readlane_d(gu.g.sa, src); bu.rad = readlane_d(gu.rad, src);
          ba.g.x = __builtin_amdgcn_readlane(ag.g.x, src); ba.g.y = __builtin_amdgcn_readlane(ag.g.y, src);
          ba.g.hl = readlane_d(ag.g.hl, src); ba.g.hw = readlane_d(ag.g.hw, src);
          ba.g.ca = readlane_d(ag.g.ca, src); ba.g.sa = readlane_d(ag.g.sa, src); ba.rad = readlane_d(ag.rad, src);
          int uci, ucj;
          cell_index(c, bu.g.x, bu.g.y, &uci, &ucj);
          int cell2 = -1;
          if (ck < 9) {
            const int ii = uci + ck / 3 - 1, jj = ucj + ck % 3 - 1;
            if (ii >= 0 && ii < c.h.nx && jj >= 0 && jj < c.h.ny) cell2 = jj + ii * c.h.ny;
          }
          const int cnt2 = cell2 >= 0 ? (int)L.cell_cnt[cell2] : 0;
          const bool direct2 = __ballot(cnt2 > 3) == 0ull;
          int M2 = WAVE;
          if (!direct2) {
            M2 = 0;
#pragma unroll
            for (int k = 0; k < 9; ++k) M2 += __builtin_amdgcn_readlane(cnt2, 3 * k);
          }
          const double Aus = geo_area(bu.g);
          double acc0 = 0.0, acc1 = 0.0;                     // wave-uniform results
          for (int base2 = 0; base2 < M2; base2 += WAVE) {
            int it_base = cell2 * c.h.cell_cap, it_e = ce;
            bool act2 = ce < cnt2;
            if (!direct2) {
              const int jx = base2 + c.lane;
              int lo = 0, my_lo = 0;
              it_base = 0;
#pragma unroll
              for (int k = 0; k < 9; ++k) {
                const int cnt_k = __builtin_amdgcn_readlane(cnt2, 3 * k), cell_k = __builtin_amdgcn_readlane(cell2, 3 * k);
                if (jx >= lo && cnt_k > 0) { my_lo = lo; it_base = cell_k * c.h.cell_cap; }
                lo += cnt_k;
              }
              act2 = jx < M2;
              it_e = jx - my_lo;
            }
            const int w = act2 ? (int)L.cell_items[it_base + it_e] : 0;
            if (act2 && (w == us || w == rem_s)) act2 = false;
            int wx = 0, wy = 0, d2w = 0;
            if (act2) {
              const int wxy = L.xy[w];
              wx = wxy & 0xffff; wy = (wxy >> 16) & 0xffff;
              const int dx = bu.g.x - wx, dy = bu.g.y - wy;
              d2w = dx * dx + dy * dy;
            }
            if (need1) {                                     // alignment, reduced with min
              double v1 = 0.0;
              if (act2 && d2w <= maxd2_1) v1 = 1.0 - fabs(bu.g.ca * L.ca[w] + bu.g.sa * L.sa[w]) - rew;
              unsigned long long bm = __ballot(v1 < 0.0);
              while (bm) {
                const int l2 = __ffsll((long long)bm) - 1;
                bm &= bm - 1;
                acc1 = reduce2(MPP_REDUCE_MIN, acc1, readlane_d(v1, l2));
              }
            }
            if (need0) {                                     // overlap, reduced with max
              bool needc = act2 && d2w <= maxd2_0;
              Geo gw;
              gw.x = wx; gw.y = wy; gw.hl = gw.hw = gw.ca = gw.sa = 0.0;
              double mn = 0.0;
              bool uf = false;
              if (needc) {
                gw.hl = L.hl[w]; gw.hw = L.hw[w]; gw.ca = L.ca[w]; gw.sa = L.sa[w];
                const double B = geo_area(gw), reach = bu.rad + L.rad[w], d2 = (double)d2w;
                mn = Aus < B ? Aus : B;
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
                for (int q = 0; q < 4; ++q) {
                  sx[q] = u_first ? ux[q] : vx[q]; sy[q] = u_first ? uy[q] : vy[q];
                  cx[q] = u_first ? vx[q] : ux[q]; cy[q] = u_first ? vy[q] : uy[q];
                }
                acc0 = reduce2(MPP_REDUCE_MAX, acc0, clip_area_wave(c, sx, sy, cx, cy) / (mn2 + AREA_EPS));
              }
            }
          }
          if (ha_s) {                                        // the proposed rectangle is a neighbour too (uniform)
            const int dx = bu.g.x - ba.g.x, dy = bu.g.y - ba.g.y, d2a2 = dx * dx + dy * dy;
            if (need1 && d2a2 <= maxd2_1)
              acc1 = reduce2(MPP_REDUCE_MIN, acc1, 1.0 - fabs(bu.g.ca * ba.g.ca + bu.g.sa * ba.g.sa) - rew);
            if (need0 && d2a2 <= maxd2_0) {
              const double B = geo_area(ba.g), mn = Aus < B ? Aus : B, reach = bu.rad + ba.rad, d2 = (double)d2a2;
              if (!(mn < DEGENERATE_AREA) && !(d2 > reach * reach * 1.0000001)) {
                const bool u_first = slot_first(L, us, bu.g, ba.g.x, ba.g.y, readlane_d(a_s, is), readlane_d(a_r, is), readlane_d(a_a, is));
                double ux[4], uy[4], vx[4], vy[4];
                geo_corners(bu.g, ux, uy); geo_corners(ba.g, vx, vy);
                double sx[4], sy[4], cx[4], cy[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  sx[q] = u_first ? ux[q] : vx[q]; sy[q] = u_first ? uy[q] : vy[q];
                  cx[q] = u_first ? vx[q] : ux[q]; cy[q] = u_first ? vy[q] : uy[q];
                }
                acc0 = reduce2(MPP_REDUCE_MAX, acc0, clip_area_wave(c, sx, sy, cx, cy) / (mn + AREA_EPS));
              }
            }
          }
          if (c.lane == src) {
            if (need0) nv0 = acc0;
            if (need1) nv1 = acc1;
          }
        }
        DPH(16);
        // -- the neighbours whose energy changes, added to their step's sum in list order
        const bool changed = act && (nv0 != ov0 || nv1 != ov1);
        double de = 0.0;
        if (changed) {
          const double lin = L.lin[u];
          const int gt = L.gate[u];
          de = finish_energy_c(c, lin + pair_part_c(c, gt, nv0, nv1)) - finish_energy_c(c, lin + pair_part_c(c, gt, ov0, ov1));
          if (apply) { L.red0[u] = nv0; L.red1[u] = nv1; }
        }
        unsigned long long cm = __ballot(changed);
        while (cm) {
          const int src = __ffsll((long long)cm) - 1;
          cm &= cm - 1;
          const int is = __builtin_amdgcn_readlane(i, src);
          const double v = readlane_d(de, src);
          const int uxy = __builtin_amdgcn_readlane((gu.g.x & 0xffff) | (gu.g.y << 16), src);
          const int uu = __builtin_amdgcn_readlane(u, src);
          const double w0 = readlane_d(nv0, src), w1 = readlane_d(nv1, src);
          const int ur = (int)ceil(readlane_d(gu.rad, src));          // its circumradius, rounded up
          if (c.lane == is) {
            sum += v;
            if (nchg == 0) { nb0 = uxy; nbr = ur & 0xff; su = uu; sv00 = w0; sv01 = w1; }
            else if (nchg == 1) { nb1 = uxy; nbr |= (ur & 0xff) << 8; su |= uu << 16; sv10 = w0; sv11 = w1; }
            nchg += 1;
          }
        }
        DPH(17);
      }
      M = 0;
      wave_lds_fence();
    }
    if (final) break;
  }
  wave_lds_fence();
  *sum_out = sum; *nchg_out = nchg; *nb0_out = nb0; *nb1_out = nb1; *nbr_out = nbr; *nresc_out = nresc;
  *su_out = su; sv_out[0] = sv00; sv_out[1] = sv01; sv_out[2] = sv10; sv_out[3] = sv11;
  *ra0_out = __longlong_as_double((long long)racc[c.lane]);
  *ra1_out = __longlong_as_double((long long)racc[64 + c.lane]);
  *nonfinite_out = EXT && racc[128 + c.lane] != 0ull;
}

// the state change of a committed step, done by the lane that evaluated it (n: the population at the round's start)
__device__ __forceinline__ void deep_mutate(const Chain &c, const Rec &r, int n, const double *st) {
  const Lds &L = c.L;
  int ci, cj;
  if (r.n_stash > 0 && r.n_stash <= 2) {               // the neighbours whose reductions change (more: the second pass wrote them)
    const int su = (int)__double_as_longlong(st[4]);
    L.red0[su & 0xffff] = st[0]; L.red1[su & 0xffff] = st[1];
    if (r.n_stash == 2) { L.red0[(su >> 16) & 0xffff] = st[2]; L.red1[(su >> 16) & 0xffff] = st[3]; }
  }
  if (r.has_rem && r.has_add) {                        // move / transform: same slot
    const int c0 = cell_index(c, r.rx, r.ry, &ci, &cj), c1 = cell_index(c, r.ax, r.ay, &ci, &cj);
    if (c0 != c1) { cell_remove_1(c, c0, r.tslot); cell_insert_1(c, c1, r.tslot); }
    write_slot_1(c, r.tslot, r);
  } else if (r.has_rem) {                              // death: last index takes the hole
    cell_remove_1(c, cell_index(c, r.rx, r.ry, &ci, &cj), r.tslot);
    const unsigned short last = L.order[n - 1];
    L.order[n - 1] = (unsigned short)r.tslot;
    L.order[r.tidx] = last;
  } else {                                             // birth: next free slot
    const int slot = L.order[n];
    cell_insert_1(c, cell_index(c, r.ax, r.ay, &ci, &cj), slot);
    write_slot_1(c, slot, r);
  }
}
__device__ __forceinline__ unsigned long long low_mask(int k) { return k <= 0 ? 0ull : (k >= 64 ? ~0ull : ((1ull << k) - 1ull)); }

// EXT: the instantiation that knows the classic image energies (mpp_classics.hpp): every lane rasterises its own rectangle
template <int WAVES, bool DIAG, int OCC, bool EXT>
__global__ __launch_bounds__(WAVE *WAVES, OCC) void mpp_deep_kernel(const DevParams Pv, const TileRef *tiles, int tile0,
                                                                 const long long *until, long long trace_base,
                                                                 unsigned long long seed, unsigned int chain0, int trace_tile,
                                                                 mpp_step_out *out, mpp_proposal *props, int nmax,
                                                                 int fixed_depth, int gain8, unsigned long long *stats) {
  const bool by_type = (gain8 & 0x100) == 0;      // (bit 8 of the gain word: deal the sorted steps in blocks instead -- A/B tests)
  gain8 &= 0xff;
  constexpr int NCH = DEEP_NMAX_LIMIT / 64;              // chunks of 64 step reports a lane may have to look at
  const DevParams *P = deep_stage_params<(WAVES >= MPP_LDS_PARAMS_MIN_WAVES)>(Pv, WAVE * WAVES);
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int tile = tile0 + blockIdx.x;
  Chain c;
  c.P = P; c.t = tiles[tile];
  load_model_regs(c);
  load_hot(c);
  const int ncell = P->nx * P->ny, cap = P->cap;
  const int rowbase_n = P->rowbase_lds ? P->H + 1 : 0;
  c.L = carve(lds_raw, cap, ncell, P->cell_cap, 0, rowbase_n, WAVES);
  const DeepLds D = deep_carve(lds_raw + deep_base_bytes(cap, ncell, P->cell_cap, rowbase_n, WAVES), nmax, WAVES, EXT ? 1 : 0);
  c.lane = threadIdx.x & (WAVE - 1);
  c.wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
  const Lds &L = c.L;
  const int tid = threadIdx.x, nthr = WAVE * WAVES;
  const bool tracing = DIAG && (out != nullptr || props != nullptr) && tile == trace_tile;

  // ---------------------------------------------------------------- load the configuration (as mpp_chain_kernel does)
  int n0 = __builtin_amdgcn_readfirstlane(*c.t.n);
  int err = __builtin_amdgcn_readfirstlane(*c.t.err);
  if (n0 > cap) { n0 = cap; err = ERR_POINT_OVERFLOW; }
  for (int i = tid; i < 3 * MPP_NCLASS; i += nthr) L.edges[i] = P->maps.edges[i / MPP_NCLASS][i % MPP_NCLASS];
  for (int i = tid; i < MPP_NCLASS; i += nthr) {
    const double al = P->maps.edges[2][i] + MPP_PI / 2.0;
    L.trig[i] = cos(al); L.trig[MPP_NCLASS + i] = sin(al);
  }
  for (int i = tid; i < rowbase_n; i += nthr) L.rowbase[i] = c.t.rowbase[i];
  for (int i = tid; i < cap; i += nthr) L.order[i] = (unsigned short)i;
  for (int i = tid; i < ncell; i += nthr) L.cell_cnt[i] = 0;
  __syncthreads();
  for (int i = tid; i < n0; i += nthr) {
    Rect q{c.t.px[i], c.t.py[i], c.t.ps[i], c.t.pr[i], c.t.pa[i]};
    Geo g = make_geo(q);
    double lin; int gate;
    unit_part<EXT>(P, c.t, L.edges, q, g, &lin, &gate, nullptr);
    L.xy[i] = (q.x & 0xffff) | (q.y << 16);
    L.s[i] = q.s; L.r[i] = q.r; L.a[i] = q.a; L.ca[i] = g.ca; L.sa[i] = g.sa; L.hl[i] = g.hl; L.hw[i] = g.hw;
    L.rad[i] = geo_radius(g);
    L.lin[i] = lin; L.gate[i] = (unsigned char)gate; L.red0[i] = 0.0; L.red1[i] = 0.0;
  }
  __syncthreads();
  const double alpha = c.t.T[1], T_target = c.t.T[2];
  const int rmask = 4 * nmax - 1;
  if (tid == 0) {                               // serial: keeps the cell order, hence the result, deterministic
    for (int i = 0; i < n0; ++i) {
      int xy = L.xy[i], ci, cj;
      int cell = cell_index(c, xy & 0xffff, (xy >> 16) & 0xffff, &ci, &cj);
      int cnt = L.cell_cnt[cell];
      if (cnt >= P->cell_cap) { err = ERR_CELL_OVERFLOW; break; }
      L.cell_items[(size_t)cell * P->cell_cap + cnt] = (unsigned short)i;
      L.cell_cnt[cell] = (unsigned short)(cnt + 1);
    }
    L.sh[1] = err;
  }
  if (tid == nthr - 1) {                        // temperatures of the first 2 * nmax steps (rjmcmc.py:158-159, one multiply per step)
    double Tc = *c.t.T;
    for (int i = 0; i < 2 * nmax; ++i) { D.tring[i] = Tc; if (Tc > T_target) Tc *= alpha; }
  }
  __syncthreads();
  err = __builtin_amdgcn_readfirstlane(L.sh[1]);
  {                                             // cached pair reductions of the initial configuration
    Rect dummy{0, 0, 0, 0, 0};
    Geo2 dg;
    dg.g = Geo{0, 0, 0, 0, 0, 0}; dg.rad = 0.0;
    for (int u = tid; u < n0; u += nthr) {
      Geo2 gu = load_geo(L, u);
#pragma clang loop unroll(disable)
      for (int p = 0; p < P->model.n_pair; ++p) {
        double v;
        [[clang::always_inline]] v = rescan_lane(c, p, u, gu, -1, false, dummy, dg);     // (a real call pins the chain state in scratch)
        if (p == 0) L.red0[u] = v; else L.red1[u] = v;
      }
    }
  }
  __syncthreads();

  // ---------------------------------------------------------------- the chain
  const long long step0 = *c.t.step;
  const long long n_steps = until[tile] - step0;
  const long long tr0 = step0 - trace_base;
  const unsigned long long seed_t = c.t.key_on ? (unsigned long long)c.t.key_seed : seed;
  const uint32_t chain_t = c.t.key_on ? c.t.key_chain : chain0 + (uint32_t)tile;
  const uint32_t k0 = (uint32_t)seed_t, k1 = (uint32_t)(seed_t >> 32);
  const unsigned long long below = (1ull << c.lane) - 1ull;
  long long done = 0;
  int n = n0;                                   // the population, tracked by every wave
  int depth = WAVES, ema16 = WAVES * 16;        // steps of the next round; committed steps per round, x16, smoothed
  unsigned long long st_rounds = 0, st_eval = 0, st_apply = 0;
#ifdef MPP_DEEP_PROF
  unsigned long long dph_[DPH_N] = {0}, dph_t_ = 0;
#endif

  // The loop alternates between two stages that share the call of eval_delta_lane: stage 1 draws and evaluates a round's
  // steps and decides which commit; stage 0 (only after a round with a change) applies them.
  int stage = 1;
  Rec r;
  r.valid = 0; r.kernel = 0; r.accepted = 0; r.has_rem = r.has_add = 0; r._pad = 0; r.tslot = -1; r.tidx = -1;
  r.ax = r.ay = r.rx = r.ry = 0; r.as = r.ar = r.aa = 0.0; r.hl = r.hw = r.ca = r.sa = r.rad = 0.0; r.lin_a = 0.0; r.gate_a = 1;
  bool mine = false, my_commit = false;
  int myoff = 0, lim = 0, committed = 0, cur_n = n, nb0 = 0, nb1 = 0, nbr = 0, ring_todo = 0;
  long long ring_from = 0;
  double Tm = 0.0;
  while (stage == 0 || (done < n_steps && err == 0)) {
    bool do_eval = my_commit && r.n_stash > 2;  // stage 0: the steps that commit and change more neighbours than they could note
    DPH_T0();
    if (stage == 1) {
    int N = fixed_depth > 0 ? fixed_depth : depth;
    if (N > nmax) N = nmax;
    const int Lw = N / WAVES;                   // active lanes per wave
    const long long left = n_steps - done;
    lim = left < (long long)N ? (int)left : N;
    const bool act0 = c.lane < Lw;
    const int e = c.wave * Lw + c.lane;         // my offset when the types are computed, my sorted position afterwards

    // ---- A: kernel type of step done + e, counting sort by type over the workgroup
    int kt = 15;
    uint32_t w0[4] = {0u, 0u, 0u, 0u};
    if (act0 && e < lim) {
      const uint64_t s = (uint64_t)(step0 + done + e);
      philox4x32_10((uint32_t)s, (uint32_t)(s >> 32), 0u, chain_t, k0, k1, w0);
      const double uk = u53(w0[0], w0[1]);
      kt = 0;
      while (kt < P->n_kernels - 1 && P->p_cum[kt] <= uk) ++kt;
    }
    unsigned long long same = 0ull;
    int cnt_lane = 0;                           // lane k: steps of type k in this wave
#pragma unroll
    for (int k = 0; k < MPP_NKERNEL; ++k) {
      const unsigned long long m = __ballot(kt == k);
      if (kt == k) same = m;
      if (c.lane == k) cnt_lane = __popcll(m);
    }
    const int rank = __popcll(same & below);
    if (c.lane < 16) D.tcnt[c.wave * 16 + c.lane] = (unsigned short)cnt_lane;
    DPH(0);
    __syncthreads();                            // (1) also: the changes of the previous round are in place
    DPH(1);
    int tot = 0, pre = 0;
    if (c.lane < 16) {
#pragma unroll
      for (int w = 0; w < WAVES; ++w) {
        const int v = D.tcnt[w * 16 + c.lane];
        if (w < c.wave) pre += v;
        tot += v;
      }
    }
    int excl = 0, run = 0;
#pragma unroll
    for (int k = 0; k < MPP_NKERNEL; ++k) {
      const int t_k = __builtin_amdgcn_readlane(tot, k);
      if (c.lane == k) excl = run;
      run += t_k;
    }
    const int first = __shfl(excl + pre, kt & 15, WAVE);
    if (kt != 15) {
      const int p = first + rank;
      D.pw[p] = make_uint4(w0[0], w0[1], w0[2], w0[3]);
      D.poff[p] = (unsigned short)e;
    }
    __syncthreads();                            // (2)
    DPH(2);

    // ---- B: evaluate my step.  Sorted position e of the thread: with eight waves and the eight kernels of the mixture, wave w
    //      takes the steps of kernel w (it then runs ONE kernel's code, not the tail of one type and the head of the next);
    //      otherwise -- or should a type have more than 64 steps -- the sorted steps are dealt to the waves in blocks.
    int es = e;
    mine = act0 && e < lim;
    if (WAVES == 8 && by_type && !EXT) {           // (EXT: the contrast terms are evaluated one rectangle at a time per wave -- equal shares)
      bool fits = true;
#pragma unroll
      for (int k = 0; k < MPP_NKERNEL; ++k) {
        const int t_k = __builtin_amdgcn_readlane(tot, k);
        fits = fits && t_k <= WAVE && (k < WAVES || t_k == 0);
      }
      if (fits) {
        const int first_w = __builtin_amdgcn_readlane(excl, c.wave), tot_w = __builtin_amdgcn_readlane(tot, c.wave);
        mine = c.lane < tot_w;
        es = first_w + c.lane;
      }
    }
    myoff = mine ? (int)D.poff[es] : 0;
    r.valid = 0; r.kernel = 0; r.accepted = 0; r.has_rem = r.has_add = 0; r.tslot = -1; r.tidx = -1;
    int keep_x = 0;                                    // (EXT only: what deep_pre needs after the cooperative pass below)
    MapVals pmv_x{0.f, 0.f, 0.f, 0.f, 0.0, 0.0, 0.0, 0};
    if (mine) {
      Tm = D.tring[(int)((done + myoff) & (long long)rmask)];
      r.valid = 1;
      int keep = 0;
      MapVals pmv{0.f, 0.f, 0.f, 0.f, 0.0, 0.0, 0.0, 0};
      uint32_t w[8];
      const uint4 wv = D.pw[es];
      w[0] = wv.x; w[1] = wv.y; w[2] = wv.z; w[3] = wv.w;
      const uint64_t s = (uint64_t)(step0 + done + myoff);
      philox4x32_10((uint32_t)s, (uint32_t)(s >> 32), 1u, chain_t, k0, k1, w + 4);
      draw_proposal<true>(c, w, n, r, &keep, k0, k1, s, chain_t, &pmv);
      DPH(3);
      if (r.kernel >= MPP_K_SPLIT) { r.valid = 0; r.kernel = -1; }
      if (r.valid && r.has_add && (r.ax < 0 || r.ax >= c.h.H || r.ay < 0 || r.ay >= c.h.W)) { r.valid = 0; r.kernel = -1; }
      if (r.valid) {
        deep_add_geo(c, r, keep);
        if (!EXT) deep_pre<EXT>(c, r, keep, tracing, pmv, nullptr);
      }
      if (EXT) { keep_x = keep; pmv_x = pmv; }
    }
    // the classic contrast term of every rectangle this wave's steps add: one rectangle at a time by the whole wave
    // (csrc/mpp_classics.hpp: rows, then columns, across lanes), before the lanes go their own ways again
    double cpre = 0.0;
    bool have_cpre = false;
    if (EXT) {
      int cterm = -1;
      for (int k = 0; k < P->model.n_unit; ++k) if (P->model.unit[k].kind == MPP_U_CONTRAST) cterm = k;
      if (cterm >= 0) {
        unsigned long long m = __ballot(mine && r.valid && r.has_add);
        while (m) {
          const int l = __ffsll((long long)m) - 1;
          m &= m - 1;
          Geo g;
          g.x = __builtin_amdgcn_readlane(r.ax, l); g.y = __builtin_amdgcn_readlane(r.ay, l);
          g.hl = readlane_d(r.hl, l); g.hw = readlane_d(r.hw, l); g.ca = readlane_d(r.ca, l); g.sa = readlane_d(r.sa, l);
          const double v = classic_contrast_wave(P->model.unit[cterm], c.t.img, c.t.img_c, c.h.H, c.h.W, g, c.lane);
          if (c.lane == l) { cpre = v; have_cpre = true; }
        }
      }
    }
    if (EXT && mine && r.valid) deep_pre<EXT>(c, r, keep_x, tracing, pmv_x, have_cpre ? &cpre : nullptr);
    do_eval = mine && r.valid && (r.has_rem || r.has_add);
    DPH(4);
    }
    // ---- the neighbours' part of dE (energy_graph.py:139-225) for all steps of the wave; stage 0 writes their cached
    //      reductions
    {
      double ra0 = 0.0, ra1 = 0.0, sde = 0.0, sv[4];
      int ns = 0, nresc = 0, su = 0;
      bool nonf = false;
      const bool hr = r.has_rem != 0, ha = r.has_add != 0;
      deep_delta<EXT>(c, D, do_eval, hr, ha, hr ? r.tslot : -1, (r.rx & 0xffff) | (r.ry << 16), (r.ax & 0xffff) | (r.ay << 16), r.as,
                 r.ar, r.aa, r.hl, r.hw, r.ca, r.sa, r.rad, stage == 0, &sde, &ra0, &ra1, &ns, &nb0, &nb1, &nbr, &nresc, &su, sv, &nonf DPH_PASS);
      if (stage == 1 && do_eval) {
        double dE = sde;
        if (ha) dE += finish_energy_c(c, r.lin_a + pair_part_c(c, r.gate_a, ra0, ra1));
        if (hr) dE -= finish_energy_c(c, L.lin[r.tslot] + pair_part_c(c, (int)L.gate[r.tslot], L.red0[r.tslot], L.red1[r.tslot]));
        if (EXT && nonf) dE = nan("");
        r.dE = dE; r.ra0 = ra0; r.ra1 = ra1; r.n_stash = ns; r._pad2 = nresc;
        if (ns > 0 && ns <= 2) {                 // (kept for the commit: written by this lane again, read by no other)
          double *st = D.st + (size_t)5 * myoff;
          st[0] = sv[0]; st[1] = sv[1]; st[2] = sv[2]; st[3] = sv[3]; st[4] = __longlong_as_double((long long)su);
        }
      }
    }
    DPH(stage == 1 ? 5 : 9);
    if (stage == 1) {
    if (mine) {
      if (r.valid) deep_post(c, r, n, Tm, tracing);
      // does the step change the configuration?  An accepted move that writes the values the slot already holds does not
      // (its cached geometry, unit energy and reductions are functions of those values: the same bits)
      bool chg = r.valid && r.accepted && (r.has_rem || r.has_add);
      if (chg && r.has_rem && r.has_add) {
        const int sl = r.tslot;
        chg = !(r.ax == r.rx && r.ay == r.ry && r.as == L.s[sl] && r.ar == L.r[sl] && r.aa == L.a[sl]);
      }
      bool full = false;
      if (chg && r.has_add) {
        int ci, cj;
        const int c1 = cell_index(c, r.ax, r.ay, &ci, &cj), c0 = r.has_rem ? cell_index(c, r.rx, r.ry, &ci, &cj) : -1;
        full = c1 != c0 && (int)L.cell_cnt[c1] >= c.h.cell_cap;
      }
      int ci_r = 0, cj_r = 0, ci_a = 0, cj_a = 0;
      if (r.has_rem) cell_index(c, r.rx, r.ry, &ci_r, &cj_r);
      if (r.has_add) cell_index(c, r.ax, r.ay, &ci_a, &cj_a);
      const int f = (r.tslot & 0xffff) | (r.valid ? DI_VALID : DI_BAD) | (chg ? DI_CHG : 0) | (r.has_rem ? DI_HR : 0) |
                    (r.has_add ? DI_HA : 0) | (r.accepted ? DI_ACC : 0) | (full ? DI_FULL : 0) | (chg && r.n_stash > 0 ? DI_NB : 0) |
                    (chg && r.n_stash > 1 ? DI_NB2 : 0) | (chg && r.n_stash > 2 ? DI_NBOV : 0) | (r.valid && r._pad2 > 0 ? DI_RESC : 0);
      D.info[myoff] = make_uint4((unsigned)f, (unsigned)((r.rx & 0xffff) | (r.ry << 16)), (unsigned)((r.ax & 0xffff) | (r.ay << 16)),
                                 (unsigned)(ci_r | (cj_r << 8) | (ci_a << 16) | (cj_a << 24)));
      // circumradius (rounded up, at most 255) of the largest rectangle the step removes or adds: how far its overlap term reaches
      double rr = r.has_add ? r.rad : 0.0;
      if (r.has_rem) { const double r0_ = L.rad[r.tslot]; rr = r0_ > rr ? r0_ : rr; }
      const int own_r = rr < 254.0 ? (int)ceil(rr) : 255;
      D.nb[myoff] = make_uint4((unsigned)nb0, (unsigned)nb1, (unsigned)nbr, (unsigned)own_r);
    }
    // temperatures of the steps that entered the window with the PREVIOUS round's commit (the ring is two rounds ahead; wave 0
    // runs the cheapest kernels and would wait at the barrier anyway): one multiply per step, in step order, as the chain does
    if (tid == WAVE - 1 && ring_todo > 0) {
      double Tc = D.tring[(int)((ring_from - 1) & (long long)rmask)];
      for (int i = 0; i < ring_todo; ++i) {
        if (Tc > T_target) Tc *= alpha;
        D.tring[(int)((ring_from + i) & (long long)rmask)] = Tc;
      }
    }
    ring_todo = 0;
    DPH(6);
    __syncthreads();                            // (3)
    DPH(7);

    // ---- C: which steps commit (every wave takes the same decision from the same reports).  Steps commit in order.  A
    //      committed move / transform q makes a later report o untrustworthy when o read something q writes:
    //      the slot itself; a rectangle within the interaction range of o's points, before or after (q's slot);
    //      a neighbour whose cached reductions q changes and o used (q reports up to two of them; more: twice the range);
    //      a cell list of o's 3 x 3 blocks (q crosses a cell border: the order of that cell's entries changes);
    //      and, when o re-reduced a neighbour, anything within twice the range.  A birth / death ends the round.
    uint4 o_[NCH];
    int or_[NCH];                               // reach radius of the report's rectangles
    bool ok_[NCH];
    unsigned long long am_[NCH], cm_[NCH];
    const int nch = (lim + 63) >> 6;            // chunks in use
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      o_[ch] = make_uint4(0u, 0u, 0u, 0u); or_[ch] = 0; ok_[ch] = false; am_[ch] = 0ull; cm_[ch] = 0ull;
      if (ch < nch) {
        const int idx = ch * 64 + c.lane;
        const bool in = idx < lim;
        if (in) { o_[ch] = D.info[idx]; or_[ch] = (int)D.nb[idx].w; }
        ok_[ch] = in && (o_[ch].x & DI_VALID);
        am_[ch] = __ballot(in && (o_[ch].x & DI_CHG));
      }
    }
    int cur = 0;
    committed = 0; cur_n = n;
    bool any_commit = false, any_apply = false;     // a step commits with a change; ... and changes reductions of neighbours
    const int range2 = c.pr0.maxd2 > c.pr1.maxd2 ? c.pr0.maxd2 : c.pr1.maxd2, far2 = P->conflict_d2;
    while (true) {
      int first_bad = lim;
#pragma unroll
      for (int ch = NCH - 1; ch >= 0; --ch)
        if (ch < nch) {
          const unsigned long long bm = ~__ballot(ok_[ch]) & low_mask(lim - ch * 64);
          if (bm) first_bad = ch * 64 + __ffsll((long long)bm) - 1;
        }
      int wq = -1;
#pragma unroll
      for (int ch = NCH - 1; ch >= 0; --ch)
        if (ch < nch) {
          const unsigned long long t = am_[ch] & ~low_mask(cur - ch * 64) & low_mask(first_bad - ch * 64);
          if (t) wq = ch * 64 + __ffsll((long long)t) - 1;
        }
      if (wq < 0) {
        committed = first_bad;
        if (first_bad < lim && (D.info[first_bad].x & DI_BAD)) err = ERR_BAD_TARGET;
        // otherwise: invalidated by an earlier commit of this round -> evaluated again next round
        break;
      }
      const uint4 q = D.info[wq];                          // (one address for the wave: a broadcast read)
      const int qf = (int)q.x;
      const bool q_hr = qf & DI_HR, q_ha = qf & DI_HA;
      // capacity checks BEFORE anything of the step is applied: the chain stops in the state before it (see mpp_sampler.hip)
      // (the evaluating lane looked at the cell's count: no state is read here, see "stage")
      if (qf & DI_FULL) { err = ERR_CELL_OVERFLOW; committed = wq; break; }
      if (!(q_hr && q_ha)) {                               // death / birth: ends the round (n and order[] change)
        if (q_hr) cur_n -= 1;
        else if (cur_n >= cap) { err = ERR_POINT_OVERFLOW; committed = wq; break; }
        else cur_n += 1;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) if ((wq >> 6) == ch) cm_[ch] |= 1ull << (wq & 63);
        any_commit = true;
        if (qf & DI_NBOV) any_apply = true;
        committed = wq + 1;
        break;
      }
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) if ((wq >> 6) == ch) cm_[ch] |= 1ull << (wq & 63);
      any_commit = true;
      if (qf & DI_NBOV) any_apply = true;
      // is a later report still trustworthy after this move / transform?
      const int q_ts = qf & 0xffff;
      const int qx[2] = {(int)(q.y & 0xffffu), (int)(q.z & 0xffffu)}, qy[2] = {(int)(q.y >> 16), (int)(q.z >> 16)};
      const int qci[2] = {(int)(q.w & 0xffu), (int)((q.w >> 16) & 0xffu)}, qcj[2] = {(int)((q.w >> 8) & 0xffu), (int)(q.w >> 24)};
      const bool q_cross = qci[0] != qci[1] || qcj[0] != qcj[1];
      const bool q_far = (qf & DI_NBOV) != 0;
      int qnx[2] = {0, 0}, qny[2] = {0, 0}, qnr[2] = {0, 0}, q_nnb = 0;
      const uint4 qn = D.nb[wq];
      const int q_r = (int)qn.w;
      if (qf & DI_NB) {
        qnx[0] = (int)(qn.x & 0xffffu); qny[0] = (int)(qn.x >> 16); qnx[1] = (int)(qn.y & 0xffffu); qny[1] = (int)(qn.y >> 16);
        qnr[0] = (int)(qn.z & 0xffu); qnr[1] = (int)((qn.z >> 8) & 0xffu);
        q_nnb = 1;
      }
      const bool q_nb2 = (qf & DI_NB2) != 0;
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        const int idx = ch * 64 + c.lane;
        if (ch < nch && ch * 64 + 63 > wq && idx > wq && ok_[ch]) {
          const uint4 o = o_[ch];
          const int of = (int)o.x;
          const bool oh[2] = {(of & DI_HR) != 0, (of & DI_HA) != 0};
          const int ox[2] = {(int)(o.y & 0xffffu), (int)(o.z & 0xffffu)}, oy[2] = {(int)(o.y >> 16), (int)(o.z >> 16)};
          const int oci[2] = {(int)(o.w & 0xffu), (int)((o.w >> 16) & 0xffu)}, ocj[2] = {(int)((o.w >> 8) & 0xffu), (int)(o.w >> 24)};
          // how far two rectangles can see each other: the alignment term its range, the overlap term as far as the
          // circumscribed circles meet (radii rounded up, + 1 px for the rounding and the 1e-7 of the circle test), neither
          // beyond the terms' cut-off
          const int o_r = or_[ch];
          auto reach2 = [&](int ra, int rb) { const int t = (ra + rb + 1) * (ra + rb + 1), a2 = c.pr1.maxd2; const int m = t > a2 ? t : a2; return m < range2 ? m : range2; };
          const int lim2 = ((of & (DI_RESC | DI_NBOV)) || q_far) ? far2 : reach2(o_r, q_r);
          bool bad = oh[0] && (of & 0xffff) == q_ts;
#pragma unroll
          for (int a = 0; a < 2; ++a)
            if (oh[a]) {
#pragma unroll
              for (int b = 0; b < 2; ++b) {
                const int dx = ox[a] - qx[b], dy = oy[a] - qy[b];
                if (dx * dx + dy * dy <= lim2) bad = true;
                if (q_cross && abs(oci[a] - qci[b]) <= 1 && abs(ocj[a] - qcj[b]) <= 1) bad = true;
              }
              if (q_nnb > 0) {
                const int dx0 = ox[a] - qnx[0], dy0 = oy[a] - qny[0];
                if (dx0 * dx0 + dy0 * dy0 <= reach2(o_r, qnr[0])) bad = true;
                const int dx1 = ox[a] - qnx[1], dy1 = oy[a] - qny[1];
                if (q_nb2 && dx1 * dx1 + dy1 * dy1 <= reach2(o_r, qnr[1])) bad = true;
              }
            }
          if (bad) ok_[ch] = false;
        }
      }
      cur = wq + 1;
    }

    // ---- D: apply the committed changes
    my_commit = false;
    if (mine) {
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) if ((myoff >> 6) == ch) my_commit = (cm_[ch] >> (myoff & 63)) & 1ull;
    }
    if (DIAG && tracing && mine && myoff < committed) {
      const long long idx = tr0 + done + myoff;
      if (out) {
        mpp_step_out so;
        so.dE = r.dE; so.fwd = r.fwd; so.bwd = r.bwd; so.log_alpha = r.log_alpha; so.T = Tm;
        so.accepted = r.accepted; so.n_after = (my_commit && !(r.has_rem && r.has_add)) ? cur_n : n;
        out[idx] = so;
      }
      if (props) {
        mpp_proposal pp;
        pp.kernel = r.kernel; pp.target = r.has_rem ? r.tidx : -1; pp.ax = r.ax; pp.ay = r.ay; pp.as = r.as;
        pp.ar = r.ar; pp.aa = r.aa; pp.aux0 = r.aux0; pp.aux1 = r.aux1; pp.param_id = r.pid;
        pp.new_class = r.ncls; pp.u_accept = r.u_acc;
        props[idx] = pp;
      }
    }
    DPH(8);
    if (stats) { st_rounds += 1; st_eval += (unsigned long long)lim; if (any_apply) st_apply += 1; }
    // depth of the next round: a multiple (gain8 / 8, default 2) of what the last rounds committed
    ema16 += committed - (ema16 >> 4);
    {
      int want = ((ema16 * gain8) >> 7) + WAVES;  // gain8 / 8 * mean + WAVES
      want = (want + WAVES - 1) / WAVES * WAVES;
      depth = want < WAVES ? WAVES : (want > nmax ? nmax : want);
    }
    // A committed step that changes reductions of its neighbours is evaluated a second time (stage 0).  When none does, the
    // lists and slots change right here: nothing between barrier (3) and the next round's barrier (1) reads them (the commit
    // decision above works on the reports alone).
    if (any_apply) stage = 0;                   // (my_commit of the steps without such neighbours: they just skip the pass)
    else {
      if (any_commit && my_commit) deep_mutate(c, r, n, D.st + (size_t)5 * myoff);
      my_commit = false;
      ring_from = done + 2 * nmax; ring_todo = committed;
      done += committed;
      n = cur_n;
    }
    } else {
      // ---- D: the committed changes (stage 0; their neighbours' reductions were written just above)
      __syncthreads();                          // (4) every reduction is written before a list or a slot changes
      DPH(10);
      if (my_commit) deep_mutate(c, r, n, D.st + (size_t)5 * myoff);
      my_commit = false;
      ring_from = done + 2 * nmax; ring_todo = committed;
      done += committed;
      n = cur_n;
      stage = 1;
      DPH(11);
    }
  }
  __syncthreads();

  // ---------------------------------------------------------------- write the configuration back
  for (int i = tid; i < n; i += nthr) {
    int slot = L.order[i], xy = L.xy[slot];
    c.t.px[i] = xy & 0xffff; c.t.py[i] = (xy >> 16) & 0xffff;
    c.t.ps[i] = L.s[slot]; c.t.pr[i] = L.r[slot]; c.t.pa[i] = L.a[slot];
  }
#ifdef MPP_DEEP_PROF
  if (stats && c.lane == 0) for (int i = 0; i < DPH_N; ++i) atomicAdd(stats + 16 + DPH_N * c.wave + i, dph_[i]);
#endif
  if (tid == 0) {
    *c.t.n = n; *c.t.err = err; *c.t.step = step0 + done;
    *c.t.T = D.tring[(int)(done & (long long)rmask)];
    if (stats) { atomicAdd(stats, st_rounds); atomicAdd(stats + 1, st_eval); atomicAdd(stats + 2, st_apply); atomicAdd(stats + 3, (unsigned long long)done); }
  }
}

// ---- host-side launcher ----------------------------------------------------------------------------
extern "C" size_t mpp_deep_lds_bytes(int cap, int ncell, int cell_cap, int rowbase_n, int waves, int nmax, int ext) {
  return deep_base_bytes(cap, ncell, cell_cap, rowbase_n, waves) + deep_extra_bytes(nmax, waves, ext);
}
extern "C" size_t mpp_deep_static_lds_bytes(int waves) {
  return waves >= MPP_LDS_PARAMS_MIN_WAVES ? ((sizeof(DevParams) + 15) & ~(size_t)15) : 0;
}

template <int WAVES, bool DIAG, int OCC, bool EXT>
static hipError_t launch_deep_d(hipStream_t st, int grid, size_t lds, const DevParams *P, const TileRef *tiles, int tile0,
                                const long long *until, long long trace_base, unsigned long long seed, unsigned int chain0,
                                int trace_tile, mpp_step_out *out, mpp_proposal *props, int nmax, int fixed_depth, int gain8,
                                unsigned long long *stats) {
  hipError_t e = hipFuncSetAttribute((const void *)mpp_deep_kernel<WAVES, DIAG, OCC, EXT>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((mpp_deep_kernel<WAVES, DIAG, OCC, EXT>), dim3(grid), dim3(WAVE * WAVES), lds, st, *P, tiles, tile0, until,
                     trace_base, seed, chain0, trace_tile, out, props, nmax, fixed_depth, gain8, stats);
  return hipGetLastError();
}

// waves = waves per chain (1, 2, 4, 8); nmax = most steps of one round (a power of two, waves <= nmax <= 64 * waves, <= 256);
// ext: a classic image energy among the unit terms (built for 1 and 8 waves, like the one-wave-per-step kernels)
extern "C" hipError_t mpp_launch_deep(hipStream_t st, int waves, int occ, int grid, size_t lds, const DevParams *P,
                                      const TileRef *tiles, int tile0, const long long *until, long long trace_base,
                                      unsigned long long seed, unsigned int chain0, int trace_tile, mpp_step_out *out,
                                      mpp_proposal *props, int nmax, int fixed_depth, int gain8, unsigned long long *stats, int ext) {
  const bool diag = out || props;
#define GO(W, O, X)                                                                                                       \
  return diag ? launch_deep_d<W, true, O, X>(st, grid, lds, P, tiles, tile0, until, trace_base, seed, chain0, trace_tile, out, props, nmax, fixed_depth, gain8, stats) \
              : launch_deep_d<W, false, O, X>(st, grid, lds, P, tiles, tile0, until, trace_base, seed, chain0, trace_tile, out, props, nmax, fixed_depth, gain8, stats)
  if (ext) {
    switch (waves) {
      case 1: GO(1, 1, true);
      case 8: GO(8, 2, true);
    }
    return hipErrorNotSupported;
  }
  switch (waves) {
    case 1: if (occ >= 2) { GO(1, 2, false); } GO(1, 1, false);
    case 2: if (occ >= 2) { GO(2, 2, false); } GO(2, 1, false);
    case 4: if (occ >= 2) { GO(4, 2, false); } GO(4, 1, false);
    case 8: GO(8, 2, false);
  }
#undef GO
  return hipErrorInvalidValue;
}
