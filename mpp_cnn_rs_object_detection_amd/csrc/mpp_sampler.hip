// mpp_sampler.hip -- the chain kernels (see mpp_chain.hpp for the design notes and the shared pieces)
#include <cstdlib>

#include "mpp_chain.hpp"
#include "mpp_split_merge.hpp"

// WAVES waves per chain.  LPW == 0: one speculative step per wave, the wave's lanes cooperate on it.
// LPW > 0 ("lane mode"): lanes 0..LPW-1 of every wave each evaluate their own step.  SPEC steps per round.
// DIAG = false is the production instantiation: proposals from Philox, nothing recorded (the tape and
// trace code is compiled out, which also lowers the register need of the hot loop).
// OCC: minimum waves per SIMD the register allocation must allow (the hot loop needs ~240 VGPRs: two waves per
// SIMD).  Throughput runs over many one-wave chains ask for OCC = 2 explicitly so that two chains share a SIMD and
// hide each other's latencies.
// SM: the instantiation that also knows the split / merge kernels (mpp_split_merge.hpp); the others carry none of it.
template <bool IN_LDS>
__device__ __forceinline__ const DevParams *stage_params(const DevParams &Pv, int nthr) {
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

// FAST: the energy model is (pair 0 = rectangle overlap / max, pair 1 = alignment / min) -- both shipped setups; the pair
// loops of eval_delta are then straight-line code (chosen by the host; the generic instantiations cover everything else).
template <int WAVES, int LPW, bool DIAG, int OCC, bool SM, bool FAST = false>
__global__ __launch_bounds__(WAVE *WAVES, OCC) void mpp_chain_kernel(const DevParams Pv, const TileRef *tiles, int tile0,
                                                                  const long long *until, long long trace_base,
                                                                  unsigned long long seed,
                                                                  unsigned int chain0, const mpp_proposal *tape,
                                                                  int trace_tile, mpp_step_out *out,
                                                                  mpp_proposal *props) {
  constexpr bool LANE = LPW > 0;
  constexpr int SPEC = LANE ? WAVES * LPW : WAVES;
  // the parameter block travels BY VALUE: it then lives in the kernel-argument segment (constant address space),
  // so every P->field is a scalar load the compiler may cache and hoist, not a vector-memory load in the
  // dependency chain of the step
  // Latency-mode chains (4 or more speculative waves: one chain per CU, LDS to spare) read the block from an LDS copy:
  // ds_reads return in order and overlap with the other LDS traffic, while a scalar load's wait (lgkmcnt(0), scalar loads
  // return out of order) drains everything in flight -- 562 k against 549 k proposals/s on the bench tile.  Throughput
  // launches (one or two waves per chain) keep their LDS for occupancy.
  const DevParams *P = stage_params<(WAVES >= MPP_LDS_PARAMS_MIN_WAVES)>(Pv, WAVE * WAVES);
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int tile = tile0 + blockIdx.x;
  Chain c;
  c.P = P; c.t = tiles[tile];
  load_model_regs(c);
  load_hot(c);
  const int ncell = P->nx * P->ny, cap = P->cap;
  const int rowbase_n = P->rowbase_lds ? P->H + 1 : 0;
  c.L = carve(lds_raw, cap, ncell, P->cell_cap, SPEC, rowbase_n, WAVES);
  c.lane = threadIdx.x & (WAVE - 1);
  c.wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);   // wave-uniform: lets Philox etc. run on the scalar unit
  const Lds &L = c.L;
  const int tid = threadIdx.x, nthr = WAVE * WAVES;
  const bool tracing = DIAG && (out != nullptr || props != nullptr) && tile == trace_tile;
  if (!DIAG) tape = nullptr;

  // ---------------------------------------------------------------- load the configuration
  int n0 = __builtin_amdgcn_readfirstlane(*c.t.n);
  int err = __builtin_amdgcn_readfirstlane(*c.t.err);
  if (n0 > cap) { n0 = cap; err = ERR_POINT_OVERFLOW; }
  for (int i = tid; i < 3 * MPP_NCLASS; i += nthr) L.edges[i] = P->maps.edges[i / MPP_NCLASS][i % MPP_NCLASS];
  for (int i = tid; i < MPP_NCLASS; i += nthr) {       // the expression of make_geo() / evaluate(): the same bits
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
    unit_part<SM>(P, c.t, L.edges, q, g, &lin, &gate, nullptr);
    L.xy[i] = (q.x & 0xffff) | (q.y << 16);
    L.s[i] = q.s; L.r[i] = q.r; L.a[i] = q.a; L.ca[i] = g.ca; L.sa[i] = g.sa; L.hl[i] = g.hl; L.hw[i] = g.hw;
    L.rad[i] = geo_radius(g);
    L.lin[i] = lin; L.gate[i] = (unsigned char)gate; L.red0[i] = 0.0; L.red1[i] = 0.0;
  }
  __syncthreads();
  if (tid == 0) {                               // serial: keeps the cell order, hence the result, deterministic
    for (int i = 0; i < n0; ++i) {
      int xy = L.xy[i], ci, cj;
      int cell = cell_index(c, xy & 0xffff, (xy >> 16) & 0xffff, &ci, &cj);
      int cnt = L.cell_cnt[cell];
      if (cnt >= P->cell_cap) { err = ERR_CELL_OVERFLOW; break; }
      L.cell_items[(size_t)cell * P->cell_cap + cnt] = (unsigned short)i;
      L.cell_cnt[cell] = (unsigned short)(cnt + 1);
    }
    L.sh[0] = n0; L.sh[1] = err; L.sh[2] = 0;
    *(double *)(L.sh + 4) = *c.t.T;
  }
  __syncthreads();
  err = L.sh[1];
  {                                             // cached pair reductions of the initial configuration
    Rect dummy{0, 0, 0, 0, 0};
    Geo2 dg;
    dg.g = Geo{0, 0, 0, 0, 0, 0}; dg.rad = 0.0;
    for (int u = tid; u < n0; u += nthr) {
      Geo2 gu = load_geo(L, u);
#pragma clang loop unroll(disable)
      for (int p = 0; p < P->model.n_pair; ++p) {
        double v = rescan_lane(c, p, u, gu, -1, false, dummy, dg);
        if (p == 0) L.red0[u] = v; else L.red1[u] = v;
      }
    }
    if (tid == 0) L.sh[6] = 0;
  }
  __syncthreads();

  // ---------------------------------------------------------------- the chain
  const double alpha = c.t.T[1], T_target = c.t.T[2];
  long long step0 = *c.t.step, done = 0;
  // the launch runs every tile up to ITS absolute step until[tile]; a launch that follows a capacity overflow (the
  // host raised the capacity and re-launched) finds the finished tiles at their end and the stopped one where it stopped
  const long long n_steps = until[tile] - step0;
  // records of a traced tile / tape entries are indexed from the step the host's call started at
  const long long tr0 = step0 - trace_base;
  const unsigned long long seed_t = c.t.key_on ? (unsigned long long)c.t.key_seed : seed;
  const uint32_t chain_t = c.t.key_on ? c.t.key_chain : chain0 + (uint32_t)tile;
  const uint32_t k0 = (uint32_t)seed_t, k1 = (uint32_t)(seed_t >> 32);
#ifdef MPP_PROFILE
  unsigned long long prof_[16] = {0};
#endif

  int ho_ema = 0, ho_rounds = 0;                // (the commit wave) steps committed per round, x 256, smoothed; rounds so far
  while (done < n_steps && err == 0) {
    const int n = __builtin_amdgcn_readfirstlane(L.sh[0]);
    double T = *(double *)(L.sh + 4);
    PROF_T0();
#ifdef MPP_PROFILE
    const unsigned long long pt_round_ = clock64();
#endif
    // ---- phase A: record ri of this round = step done+ri, evaluated against the current state on the
    //      assumption that the steps before it in this round change nothing it depends on
    // "apply round": the previous round met an accepted step whose neighbour updates overflow the stash;
    // it alone is evaluated again (same state, same decision), writing the caches directly
    const bool apply_round = __builtin_amdgcn_readfirstlane(L.sh[6]) != 0;
    const int ri = LANE ? c.wave * LPW + c.lane : c.wave;
    const bool mine = (LANE ? (c.lane < LPW) : true) && (!apply_round || ri == 0);
    const long long my = done + ri;
    Rec r;
    r.valid = 0; r.kernel = 0; r.accepted = 0; r.has_rem = r.has_add = 0; r._pad = 0;
    if (mine && my < n_steps) {
      for (int i = 0; i < ri; ++i) if (T > T_target) T *= alpha;          // temperature of step `my`
      r.valid = 1;
      int keep = 0;
      MapVals pmv{0.f, 0.f, 0.f, 0.f, 0.0, 0.0, 0.0, 0};
      if (tape) {
        const mpp_proposal &tp = tape[tr0 + my];
        r.kernel = tp.kernel; r.tidx = tp.target; r.tslot = -1;
        r.ax = tp.ax; r.ay = tp.ay; r.as = tp.as; r.ar = tp.ar; r.aa = tp.aa; r.aux0 = tp.aux0; r.aux1 = tp.aux1;
        r.pid = tp.param_id; r.ncls = tp.new_class; r.u_acc = tp.u_accept; r.rx = r.ry = 0;
        bool is_birth = tp.kernel == MPP_K_UBIRTH || tp.kernel == MPP_K_DBIRTH;
        bool is_death = tp.kernel == MPP_K_UDEATH || tp.kernel == MPP_K_DDEATH;
        const bool is_sm = tp.kernel == MPP_K_SPLIT || tp.kernel == MPP_K_MERGE;
        if (tp.kernel < 0 || tp.kernel >= MPP_NKERNEL || (is_sm && !SM)) { r.valid = 0; r.kernel = -1; }
        else if (is_birth) r.has_add = 1;
        else if (is_sm) {                  // split: target; merge: target = p0, param_id = p1 (-1: no neighbour)
          const bool empty = tp.target < 0 || n == 0 || (tp.kernel == MPP_K_MERGE && (n < 2 || tp.param_id < 0));
          if (!empty) {
            if (tp.target >= n || (tp.kernel == MPP_K_MERGE && (tp.param_id >= n || tp.param_id == tp.target))) {
              r.valid = 0; r.kernel = -1;
            } else {
              r.has_rem = 1;
              r.tslot = L.order[tp.target];
              int xy = L.xy[r.tslot];
              r.rx = xy & 0xffff; r.ry = (xy >> 16) & 0xffff;
            }
          } else r.tidx = -1;
        }
        else if (n > 0 && tp.target >= 0) {
          if (tp.target >= n) { r.valid = 0; r.kernel = -1; }       // reported at commit time
          else {
            r.has_rem = 1; r.has_add = is_death ? 0 : 1;
            r.tslot = L.order[tp.target];
            int xy = L.xy[r.tslot];
            r.rx = xy & 0xffff; r.ry = (xy >> 16) & 0xffff;
          }
        }
        if (r.valid && !is_sm && (tp.kernel == MPP_K_DTRANSF || tp.kernel == MPP_K_GTRANSF) && r.has_rem &&
            (tp.param_id < 0 || tp.param_id > 2 || (tp.kernel == MPP_K_DTRANSF && (tp.new_class < 0 || tp.new_class >= MPP_NCLASS)))) {
          r.valid = 0; r.kernel = -1;
        }
      } else {
        uint32_t w[8];
        uint64_t s = (uint64_t)(step0 + my);
#pragma unroll
        for (uint32_t b = 0; b < 2; ++b)
          philox4x32_10((uint32_t)s, (uint32_t)(s >> 32), b, chain_t, k0, k1, w + 4 * b);
        draw_proposal<LANE>(c, w, n, r, &keep, k0, k1, s, chain_t, &pmv);
        if (SM && r.kernel >= MPP_K_SPLIT) {
          int e = 0;
          sm_draw(c, r, ri, n, w, k0, k1, s, chain_t, &e);
          if (e) { r.valid = 0; r.kernel = -2 - e; }
        } else if (!SM && r.kernel >= MPP_K_SPLIT) { r.valid = 0; r.kernel = -1; }
      }
#ifdef MPP_PROFILE
      { unsigned long long n_ = clock64(); if (c.wave == 0 && c.lane == 0 && r.kernel >= 0 && r.kernel < 8) atomicAdd(&g_prof4[r.kernel], n_ - pt_); }
#endif
      PROF_ADD(0);
      if (r.valid && r.has_add && (r.ax < 0 || r.ax >= c.h.H || r.ay < 0 || r.ay >= c.h.W)) { r.valid = 0; r.kernel = -1; }
      if (SM && r.valid && r.kernel >= MPP_K_SPLIT && r.has_rem) {
        // a two-point change runs alone on the live state: ask for an apply round, or (in it) do the whole step
        r.dE = 0.0; r.lin_a = 0.0; r.gate_a = 1; r.ra0 = r.ra1 = 0.0; r.hl = r.hw = r.ca = r.sa = r.rad = 0.0;
        r.qf = r.qb = 1.0;
        if (!apply_round) { r.accepted = 1; r.n_stash = STASH + 1; }
        else {
          int e = 0;
          sm_step(c, r, ri, n, T, tracing, &e);
          if (e) { r.valid = 0; r.kernel = -2 - e; }
        }
      } else if (r.valid) {
#ifdef MPP_PROFILE
        evaluate<LANE, FAST, SM>(c, r, ri, keep, n, T, tracing, apply_round, pmv, prof_);
#else
        evaluate<LANE, FAST, SM>(c, r, ri, keep, n, T, tracing, apply_round, pmv);
#endif
      }
#ifdef MPP_PROFILE
      { unsigned long long n_ = clock64(); if (c.wave == 0 && c.lane == 0 && r.kernel >= 0 && r.kernel < 8) { atomicAdd(&g_prof3[r.kernel], n_ - pt_); atomicAdd(&g_prof3[8 + r.kernel], 1ull); } }
#endif
      PROF_ADD(1);
    }
#ifdef MPP_PROFILE
    if (SPEC == 8 && !LANE) {            // this wave's time for its step of the round, and whether it re-reduced a neighbour
      r.fwd = (double)(clock64() - pt_round_); r.bwd = (double)L.sh[8 + (c.wave & 7)];
      if (c.lane == 0) L.sh[8 + (c.wave & 7)] = 0;
    }
#endif
    if (SPEC > 1) {
      if (LANE ? (c.lane < LPW) : (c.lane == 0)) L.rec[ri] = r;    // also the idle ones of an apply round (valid = 0)
      __syncthreads();
    }
#ifdef MPP_PROFILE
    if (SPEC == 8 && !LANE && c.wave == 0 && !tracing) {
      const Rec &pr_ = L.rec[c.lane < 8 ? c.lane : 0];
      const double t_ = c.lane < 8 && pr_.valid ? pr_.fwd : 0.0;
      const int k_ = pr_.kernel, rs_ = (int)pr_.bwd;
      double mx = 0.0, second = 0.0, sum = 0.0; int arg = 0, nv_ = 0;
      for (int i = 0; i < 8; ++i) {
        const double ti = readlane_d(t_, i);
        if (ti > 0) { sum += ti; ++nv_; }
        if (ti > mx) { second = mx; mx = ti; arg = i; } else if (ti > second) second = ti;
      }
      if (nv_ == 8) {
        const int ka = __builtin_amdgcn_readlane(k_, arg), ra = __builtin_amdgcn_readlane(rs_, arg);
        if (c.lane == 0 && ka >= 0 && ka < 8) {
          atomicAdd(&g_strag[ka], 1ull); atomicAdd(&g_strag[8 + ka], (unsigned long long)mx);
          atomicAdd(&g_strag[16 + ka], (unsigned long long)(mx - second));
          atomicAdd(&g_strag[24], 1ull); atomicAdd(&g_strag[25], (unsigned long long)mx); atomicAdd(&g_strag[26], (unsigned long long)(sum / 8));
          if (ra) atomicAdd(&g_strag[27], 1ull);
        }
        if (c.lane < 8 && k_ >= 0 && k_ < 8) {
          atomicAdd(&g_strag[28 + k_], 1ull); atomicAdd(&g_strag[36 + k_], (unsigned long long)t_);
          if (rs_) { atomicAdd(&g_strag[44], 1ull); atomicAdd(&g_strag[45], (unsigned long long)t_); }
        }
      }
    }
#endif
    PROF_ADD(9);
    // ---- phase B, untraced wave mode: wave 0 DECIDES in order which records commit (registers, ballots and readlanes
    //      only), then every chosen record is applied by the wave that evaluated it, all at once.  Two records of one
    //      round that both commit neither share a slot, nor a cell, nor lie within 2*max_inter of each other, so they
    //      touch disjoint cell lists, slots and cached reductions; a birth / death (which also changes n and order[])
    //      is always the last record of its round.
    const bool par_commit = !LANE && SPEC > 1 && !tracing;
    if (par_commit) {
      if (c.wave == 0) {
        int committed = 0, cur_n = n;
        double Tc = *(double *)(L.sh + 4);
        const long long left = n_steps - done;
        const int lim = left < (long long)SPEC ? (int)left : SPEC;
        auto low = [](int k) -> unsigned long long { return k >= 64 ? ~0ull : ((1ull << k) - 1ull); };
        const unsigned long long lim_mask = low(lim);
        const bool in = c.lane < lim;
        const Rec &me = L.rec[in ? c.lane : 0];
        bool ok = in && me.valid;
        const int m_kern = me.kernel, m_hr = me.has_rem, m_ha = me.has_add, m_ts = me.tslot, m_nst = me.n_stash, m_pad = me._pad;
        const int m_rx = me.rx, m_ry = me.ry, m_ax = me.ax, m_ay = me.ay;
        int ci, cj;
        const int m_cr = m_hr ? cell_index(c, m_rx, m_ry, &ci, &cj) : -1, m_ca = m_ha ? cell_index(c, m_ax, m_ay, &ci, &cj) : -2;
        const unsigned long long acc_mask = __ballot(in && me.accepted && (m_hr || m_ha));
        unsigned int commit_mask = 0;
        int cur = 0;
        while (true) {
          const unsigned long long bad_mask = ~__ballot(ok) & lim_mask;
          const int first_bad = bad_mask ? __ffsll((long long)bad_mask) - 1 : lim;
          const unsigned long long todo = acc_mask & ~low(cur) & low(first_bad);
          if (!todo) {
            committed = first_bad;
            if (first_bad < lim) {
              const int kq = __builtin_amdgcn_readlane(m_kern, first_bad);
              if (kq == -1) err = ERR_BAD_TARGET;
              else if (kq <= -3) err = -2 - kq;
              // otherwise: invalidated by an earlier accept of this round -> re-evaluated next round
            }
            break;
          }
          const int w = __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1);
          if (__builtin_amdgcn_readlane(m_nst, w) > STASH && !apply_round) {     // redo this step alone in an apply round
            if (c.lane == 0) L.sh[6] = 1;
            committed = w;
            break;
          }
          const int q_pad = __builtin_amdgcn_readlane(m_pad, w);
          if (SM && q_pad != 0) { cur_n += q_pad; committed = w + 1; break; }    // a split / merge applied by sm_step()
          const int q_hr = __builtin_amdgcn_readlane(m_hr, w), q_ha = __builtin_amdgcn_readlane(m_ha, w);
          // capacity checks BEFORE anything of step w is applied: the chain stops in the state before the step, so a
          // re-launch with a larger capacity continues it as if there had been no limit.  (Cell counts are those of
          // the round's start: an earlier commit of this round that touched the same cell has invalidated record w.)
          const int q_ca_w = __builtin_amdgcn_readlane(m_ca, w), q_cr_w = __builtin_amdgcn_readlane(m_cr, w);
          if (q_ha && q_ca_w != q_cr_w && (int)L.cell_cnt[q_ca_w] >= c.h.cell_cap) { err = ERR_CELL_OVERFLOW; committed = w; break; }
          if (!(q_hr && q_ha)) {                                                 // death / birth: ends the round
            if (q_hr) { commit_mask |= 1u << w; cur_n -= 1; }
            else if (cur_n >= cap) { err = ERR_POINT_OVERFLOW; committed = w; break; }
            else { commit_mask |= 1u << w; cur_n += 1; }
            committed = w + 1;
            break;
          }
          commit_mask |= 1u << w;
          // lane w2 > w: is record w2 still trustworthy after this move / transform?
          const int q_ts = __builtin_amdgcn_readlane(m_ts, w), q_cr = __builtin_amdgcn_readlane(m_cr, w), q_ca = __builtin_amdgcn_readlane(m_ca, w);
          const int qx[2] = {__builtin_amdgcn_readlane(m_rx, w), __builtin_amdgcn_readlane(m_ax, w)};
          const int qy[2] = {__builtin_amdgcn_readlane(m_ry, w), __builtin_amdgcn_readlane(m_ay, w)};
          if (c.lane > w && ok) {
            bool bad = (m_hr && m_ts == q_ts) || m_cr == q_cr || m_cr == q_ca || m_ca == q_cr || m_ca == q_ca;
            const int ox[2] = {m_rx, m_ax}, oy[2] = {m_ry, m_ay}, oh[2] = {m_hr, m_ha};
            for (int a = 0; a < 2; ++a)
              for (int b = 0; b < 2; ++b)
                if (oh[a]) {
                  int dx = ox[a] - qx[b], dy = oy[a] - qy[b];
                  if (dx * dx + dy * dy <= P->conflict_d2) bad = true;
                }
            if (bad) ok = false;
          }
          cur = w + 1;
        }
        for (int i = 0; i < committed; ++i) if (Tc > T_target) Tc *= alpha;      // rjmcmc.py:158-159
        // A hot chain changes its state every few steps: rounds of 8 speculative steps (26 k cycles) suit it better than deep
        // rounds (45 k cycles and more, whatever commits).  Once handover / 256 (5 by default) of 8 steps commit per round -- a step
        // changes the state with probability below ~0.12 -- the chain is handed to the deep-round kernel: this launch ends after the round's
        // commits, like a capacity stop, and the host continues with the very next step.
        if (P->handover && !apply_round) {
          ho_ema += ((committed << 8) - ho_ema) / 16;
          ++ho_rounds;
          if (err == 0 && ho_rounds >= 48 && ho_ema >= P->handover && done + committed < n_steps) err = ERR_HANDOVER;
        }
        if (c.lane == 0) {
          L.sh[0] = cur_n; L.sh[1] = err; L.sh[2] = committed; L.sh[3] = (int)commit_mask; *(double *)(L.sh + 4) = Tc;
          if (apply_round) L.sh[6] = 0;
        }
      }
      __syncthreads();
      if ((((unsigned int)__builtin_amdgcn_readfirstlane(L.sh[3])) >> c.wave) & 1u) {     // my record commits: apply it
        const Rec &q = r;
        int e2 = 0;
        if (!apply_round && c.lane < q.n_stash) {
          int u = L.stash_slot[c.wave * STASH + c.lane];
          L.red0[u] = L.stash_v0[c.wave * STASH + c.lane];
          L.red1[u] = L.stash_v1[c.wave * STASH + c.lane];
        }
        wave_lds_fence();
        int ci, cj;
        if (q.has_rem && q.has_add) {                      // move / transform: same slot
          int c0 = cell_index(c, q.rx, q.ry, &ci, &cj), c1 = cell_index(c, q.ax, q.ay, &ci, &cj);
          if (c0 != c1) { cell_remove(c, c0, q.tslot); cell_insert(c, c1, q.tslot, &e2); }
          write_slot(c, q.tslot, q);
        } else if (q.has_rem) {                            // death: last index takes the hole
          cell_remove(c, cell_index(c, q.rx, q.ry, &ci, &cj), q.tslot);
          if (c.lane == 0) {
            unsigned short last = L.order[n - 1];
            L.order[n - 1] = (unsigned short)q.tslot;
            L.order[q.tidx] = last;
          }
        } else {                                           // birth: next free slot
          int slot = L.order[n];
          cell_insert(c, cell_index(c, q.ax, q.ay, &ci, &cj), slot, &e2);
          write_slot(c, slot, q);
        }
        if (e2 && c.lane == 0) L.sh[1] = e2;
      }
      PROF_ADD(2);
    } else
    // ---- phase B (traced tiles, lane mode, one wave): wave 0 commits in order
    if (c.wave == 0) {
      int committed = 0, cur_n = n;
      double Tc = *(double *)(L.sh + 4);
      // commit record q (an accepted step that changes the configuration); returns true when the round must end
      // after it (population or index->slot map changed, or an error)
      auto commit_one = [&](const Rec &q, int w) -> bool {
        if (SM && q._pad != 0) { cur_n += q._pad; return true; }      // a split / merge applied by sm_step()
        if (q.has_add) {                                   // capacity checks before anything is applied (see above)
          int ci, cj;
          const int c1 = cell_index(c, q.ax, q.ay, &ci, &cj), c0 = q.has_rem ? cell_index(c, q.rx, q.ry, &ci, &cj) : -1;
          if (c1 != c0 && (int)L.cell_cnt[c1] >= c.h.cell_cap) { err = ERR_CELL_OVERFLOW; return true; }
          if (!q.has_rem && cur_n >= cap) { err = ERR_POINT_OVERFLOW; return true; }
        }
        if (!apply_round && c.lane < q.n_stash) {
          int u = L.stash_slot[w * STASH + c.lane];
          L.red0[u] = L.stash_v0[w * STASH + c.lane];
          L.red1[u] = L.stash_v1[w * STASH + c.lane];
        }
        wave_lds_fence();
        bool stop = false;
        if (q.has_rem && q.has_add) {                      // move / transform: same slot
          int ci, cj;
          int c0 = cell_index(c, q.rx, q.ry, &ci, &cj), c1 = cell_index(c, q.ax, q.ay, &ci, &cj);
          if (c0 != c1) { cell_remove(c, c0, q.tslot); cell_insert(c, c1, q.tslot, &err); }
          write_slot(c, q.tslot, q);
        } else if (q.has_rem) {                            // death: last index takes the hole
          int ci, cj;
          cell_remove(c, cell_index(c, q.rx, q.ry, &ci, &cj), q.tslot);
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
            cell_insert(c, cell_index(c, q.ax, q.ay, &ci, &cj), slot, &err);
            write_slot(c, slot, q);
            cur_n += 1;
          }
        }
        wave_lds_fence();
        // which later speculative steps are still trustworthy?
        if (SPEC > 1) {
          if (!(q.has_rem && q.has_add)) stop = true;      // n or the index->slot map changed
          else {
            // lane w2 judges record w2: it is stale if it touches the same slot or anything within
            // 2*max_inter of the positions this step changed
            int w2 = c.lane;
            if (w2 > w && w2 < SPEC) {
              Rec &o = L.rec[w2];
              if (o.valid) {
                bool bad = o.has_rem && o.tslot == q.tslot;
                int ox[2] = {o.rx, o.ax}, oy[2] = {o.ry, o.ay}, oh[2] = {o.has_rem, o.has_add};
                int qx[2] = {q.rx, q.ax}, qy[2] = {q.ry, q.ay};
                for (int a = 0; a < 2; ++a)
                  for (int b = 0; b < 2; ++b)
                    if (oh[a]) {
                      int dx = ox[a] - qx[b], dy = oy[a] - qy[b];
                      if (dx * dx + dy * dy <= P->conflict_d2) bad = true;
                    }
                if (bad) o.valid = 0;
              }
            }
          }
          wave_lds_fence();
        }
        return stop;
      };
      if (SPEC > 1 && !tracing) {
        // Untraced rounds: lane w looks at record w, and the loop only visits the ACCEPTED records (about a
        // quarter of the steps); rejected ones cost nothing but their temperature update.
        const long long left = n_steps - done;
        const int lim = left < (long long)SPEC ? (int)left : SPEC;
        auto low = [](int k) -> unsigned long long { return k >= 64 ? ~0ull : ((1ull << k) - 1ull); };
        const unsigned long long lim_mask = low(lim);
        const bool in = c.lane < lim;
        const bool acc = in && L.rec[in ? c.lane : 0].accepted && (L.rec[in ? c.lane : 0].has_rem || L.rec[in ? c.lane : 0].has_add);
        const unsigned long long acc_mask = __ballot(acc);
        int cur = 0;
        while (true) {
          const unsigned long long bad_mask = ~__ballot(in && L.rec[in ? c.lane : 0].valid) & lim_mask;
          const int first_bad = bad_mask ? __ffsll((long long)bad_mask) - 1 : lim;
          const unsigned long long todo = acc_mask & ~low(cur) & low(first_bad);
          if (!todo) {
            committed = first_bad;
            if (first_bad < lim) {
              const int kq = L.rec[first_bad].kernel;
              if (kq == -1) err = ERR_BAD_TARGET;
              else if (kq <= -3) err = -2 - kq;
              // otherwise: invalidated by an earlier accept of this round -> re-evaluated next round
            }
            break;
          }
          const int w = __ffsll((long long)todo) - 1;
          const Rec q = L.rec[w];
          if (q.n_stash > STASH && !apply_round) {           // see the traced loop below
            if (c.lane == 0) L.sh[6] = 1;
            committed = w;
            break;
          }
          const bool stop = commit_one(q, w);
          cur = w + 1;
          if (err == ERR_CELL_OVERFLOW || err == ERR_POINT_OVERFLOW) { committed = w; break; }    // step w not done
          if (stop || err) { committed = w + 1; break; }
        }
        for (int i = 0; i < committed; ++i) if (Tc > T_target) Tc *= alpha;      // rjmcmc.py:158-159
      } else {
      bool stop = false;
      for (int w = 0; w < SPEC && !stop; ++w) {
        if (done + w >= n_steps) break;
        Rec q = (SPEC > 1) ? L.rec[w] : r;
        if (!q.valid) {
          if (q.kernel == -1) err = ERR_BAD_TARGET;
          else if (q.kernel <= -3) err = -2 - q.kernel;
          // otherwise: invalidated by an earlier accept of this round -> re-evaluated next round
          break;
        }
        if (q.accepted && (q.has_rem || q.has_add)) {
          if (q.n_stash > STASH && !apply_round) {
            // more neighbours change than the stash holds (dense clusters): end the round here and redo
            // this step alone in an apply round
            if (c.lane == 0) L.sh[6] = 1;
            break;
          }
          stop = commit_one(q, w);
          if (err == ERR_CELL_OVERFLOW || err == ERR_POINT_OVERFLOW) break;                      // step w not done
        }
        if (tracing && c.lane == 0) {
          long long idx = tr0 + done + w;
          if (out) {
            mpp_step_out so;
            so.dE = q.dE; so.fwd = q.fwd; so.bwd = q.bwd; so.log_alpha = q.log_alpha; so.T = Tc;
            so.accepted = q.accepted; so.n_after = cur_n;
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
        if (Tc > T_target) Tc *= alpha;                      // rjmcmc.py:158-159
        committed += 1;
        if (err) break;
      }
      }
      if (c.lane == 0) {
        L.sh[0] = cur_n; L.sh[1] = err; L.sh[2] = committed; *(double *)(L.sh + 4) = Tc;
        if (apply_round) L.sh[6] = 0;
      }
      PROF_ADD(2);
    }
    __syncthreads();
    err = __builtin_amdgcn_readfirstlane(L.sh[1]);
    done += __builtin_amdgcn_readfirstlane(L.sh[2]);
    // (no third barrier: the next write to L.sh / L.rec[].valid by the commit wave comes after the next round's
    // barrier, which every wave reaches only after these reads and those at the top of the loop)
    PROF_ADD(3);
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
    *c.t.T = *(double *)(L.sh + 4);
  }
}

#ifdef MPP_PROFILE
extern "C" __attribute__((visibility("default"))) void mpp_debug_read_prof(unsigned long long *out) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * 16);
}
extern "C" __attribute__((visibility("default"))) void mpp_debug_read_strag(unsigned long long *out) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_strag), sizeof(unsigned long long) * 64);
  unsigned long long z[64] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_strag), z, sizeof z);
}
extern "C" __attribute__((visibility("default"))) void mpp_debug_read_prof4(unsigned long long *out) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof4), sizeof(unsigned long long) * 16);
  unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_prof4), z, sizeof z);
}
extern "C" __attribute__((visibility("default"))) void mpp_debug_read_prof3(unsigned long long *out) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof3), sizeof(unsigned long long) * 16);
  unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_prof3), z, sizeof z);
}
extern "C" __attribute__((visibility("default"))) void mpp_debug_read_prof2(unsigned long long *out, int reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof2), sizeof(unsigned long long) * 16);
  (void)hipMemcpyFromSymbol(out + 8, HIP_SYMBOL(g_clip_count), sizeof(unsigned long long));
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_prof2), z, sizeof z);
               (void)hipMemcpyToSymbol(HIP_SYMBOL(g_clip_count), z, sizeof(unsigned long long)); }
}
#endif

// until[t] = step[t] + n_steps for the tiles of a launch (set once per host call; re-launches after a capacity overflow
// keep it)
__global__ void k_set_until(const TileRef *tiles, int tile0, int n, long long n_steps, long long *until) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) until[tile0 + i] = *tiles[tile0 + i].step + n_steps;
}
extern "C" void mpp_launch_set_until(hipStream_t st, const TileRef *tiles, int tile0, int n, long long n_steps, long long *until) {
  hipLaunchKernelGGL(k_set_until, dim3((n + 255) / 256), dim3(256), 0, st, tiles, tile0, n, n_steps, until);
}

// ---- host-side launcher ----------------------------------------------------------------------------
extern "C" size_t mpp_chain_lds_bytes(int cap, int ncell, int cell_cap, int spec, int rowbase_n, int waves) {
  return lds_bytes(cap, ncell, cell_cap, spec, rowbase_n, waves);
}
// static LDS a chain kernel with `waves` waves uses besides its dynamic allocation (the staged parameter block)
extern "C" size_t mpp_chain_static_lds_bytes(int waves) {
  return waves >= MPP_LDS_PARAMS_MIN_WAVES ? ((sizeof(DevParams) + 15) & ~(size_t)15) : 0;
}

template <int WAVES, int LPW, bool DIAG, int OCC, bool SM, bool FAST = false>
static hipError_t launch_spec_d(hipStream_t st, int grid, size_t lds, const DevParams *P, const TileRef *tiles, int tile0,
                              const long long *until, long long trace_base, unsigned long long seed, unsigned int chain0,
                              const mpp_proposal *tape, int trace_tile, mpp_step_out *out, mpp_proposal *props) {
  hipError_t e = hipFuncSetAttribute((const void *)mpp_chain_kernel<WAVES, LPW, DIAG, OCC, SM, FAST>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((mpp_chain_kernel<WAVES, LPW, DIAG, OCC, SM, FAST>), dim3(grid), dim3(WAVE * WAVES), lds, st, *P, tiles, tile0,
                     until, trace_base, seed, chain0, tape, trace_tile, out, props);
  return hipGetLastError();
}
template <int WAVES, int LPW>
static hipError_t launch_spec(hipStream_t st, int grid, size_t lds, const DevParams *P, const TileRef *tiles, int tile0,
                              const long long *until, long long trace_base, unsigned long long seed, unsigned int chain0,
                              const mpp_proposal *tape, int trace_tile, mpp_step_out *out, mpp_proposal *props,
                              int occ) {
  constexpr int BASE = (WAVES + 3) / 4;      // waves per SIMD one workgroup needs anyway
  const bool diag = tape || out || props;
  bool classic = false;                      // a classic image energy among the unit terms (energies/classics.py)
  for (int k = 0; k < P->model.n_unit; ++k)
    classic = classic || P->model.unit[k].kind == MPP_U_CONTRAST || P->model.unit[k].kind == MPP_U_GRADIENT;
  if (P->n_kernels > MPP_K_SPLIT || classic) {   // split / merge kernels in the mixture, or a classic image energy: the
                                                 // extended instantiations, built for 1 and 8 waves
    if constexpr (LPW == 0 && (WAVES == 1 || WAVES == 8)) {
      if (diag) return launch_spec_d<WAVES, LPW, true, BASE, true>(st, grid, lds, P, tiles, tile0, until, trace_base, seed, chain0, tape, trace_tile, out, props);
      return launch_spec_d<WAVES, LPW, false, BASE, true>(st, grid, lds, P, tiles, tile0, until, trace_base, seed, chain0, tape, trace_tile, out, props);
    } else {
      return hipErrorNotSupported;
    }
  }
  if (diag)
    return launch_spec_d<WAVES, LPW, true, BASE, false>(st, grid, lds, P, tiles, tile0, until, trace_base, seed, chain0, tape, trace_tile, out, props);
  // the production launches of the shipped energy setups: pair loops specialised (FAST); MPP_NO_FAST=1 keeps the generic code
  const mpp_model &M = P->model;
  static const bool no_fast = getenv("MPP_NO_FAST") != nullptr;
  const bool fast = !no_fast && M.n_pair == 2 && M.pair[0].kind == MPP_P_OVERLAP && M.pair[0].reduce == MPP_REDUCE_MAX &&
                    M.pair[1].kind == MPP_P_ALIGN && M.pair[1].reduce == MPP_REDUCE_MIN;
  if constexpr (LPW == 0 && WAVES <= 8) {
    if (fast) {
      if (WAVES <= 4 && occ >= 2)
        return launch_spec_d<WAVES, LPW, false, 2, false, true>(st, grid, lds, P, tiles, tile0, until, trace_base, seed, chain0, tape, trace_tile, out, props);
      return launch_spec_d<WAVES, LPW, false, BASE, false, true>(st, grid, lds, P, tiles, tile0, until, trace_base, seed, chain0, tape, trace_tile, out, props);
    }
  }
  if (WAVES <= 4 && LPW == 0 && occ >= 2)
    return launch_spec_d<WAVES, LPW, false, 2, false>(st, grid, lds, P, tiles, tile0, until, trace_base, seed, chain0, tape, trace_tile, out, props);
  return launch_spec_d<WAVES, LPW, false, BASE, false>(st, grid, lds, P, tiles, tile0, until, trace_base, seed, chain0, tape, trace_tile, out, props);
}

// spec = steps evaluated per round; lanes = 0: one wave per step (spec waves); lanes > 0: 4 waves x lanes lanes
extern "C" hipError_t mpp_launch_chain(hipStream_t st, int spec, int lanes, int occ, int grid, size_t lds,
                                       const DevParams *P, const TileRef *tiles, int tile0, const long long *until,
                                       long long trace_base, unsigned long long seed, unsigned int chain0, const mpp_proposal *tape,
                                       int trace_tile, mpp_step_out *out, mpp_proposal *props) {
#define GO(W, L) return launch_spec<W, L>(st, grid, lds, P, tiles, tile0, until, trace_base, seed, chain0, tape, trace_tile, out, props, occ)
  if (lanes == 0) {
    switch (spec) { case 1: GO(1, 0); case 2: GO(2, 0); case 4: GO(4, 0); case 8: GO(8, 0); case 16: GO(16, 0); }
  } else {
    switch (lanes) { case 1: GO(4, 1); case 2: GO(4, 2); case 4: GO(4, 4); case 8: GO(4, 8); case 16: GO(4, 16); }
  }
#undef GO
  return hipErrorInvalidValue;
}
