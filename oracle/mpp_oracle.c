/*
 * mpp_oracle.c -- plain-C, double-precision, single-threaded restatement of the
 * reference's MPP / RJMCMC sampling path.  TEST INFRASTRUCTURE (see mpp_oracle.h).
 *
 * It follows the reference literally where that is cheap: energies of a
 * perturbation are evaluated as E(new subset) - E(initial subset) over the
 * 3x3-cell "potential neighbour" sets with every point's pair reductions
 * recomputed from scratch (energy_graph.py:139-225); no caches, no incremental
 * bookkeeping.  That makes it an independent check of the HIP path, which keeps
 * per-point reductions cached and updates them incrementally.
 *
 * Canonical state layout (the reference's is id()-hashed Python sets and cannot
 * be reproduced): points live in dense slots 0..n-1; a birth appends, a death
 * moves the last slot into the hole, a move/transform rewrites its slot in place.
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC mpp_oracle.c -lm
 */
#include "mpp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define EPS_GREEN 1e-16          /* rjmcmc.py:15 */
#define AREA_EPS 1e-6            /* prior_energies.py:18 */
#define DEGENERATE_AREA 1e-12    /* zero-width rectangle: intersection area is 0 */
#define TWO_PI 6.283185307179586476925286766559
#define PI_ 3.14159265358979323846264338327950288

typedef struct { int32_t x, y; double s, r, a; } rect_t;

struct orc_ctx {
  int H, W;
  float *det, *m[3];
  float *img; int img_c;      /* the image behind the classic energies (orc_set_image): [H][W][img_c] */
  orc_model model;
  orc_kernels kern;
  double p_cum[ORC_NKERNEL];
  int n_active;              /* 8, or 10 with the split / merge kernels */
  double res;                 /* spatial resolution, point_set.py:58 */
  int nx, ny, max_offset;
  double max_inter;           /* energy_graph.py:26-29 */
  int n, cap;
  rect_t *pt;
  int *cell_cnt, *cell_cap, **cell_items;
  double det_sum, *cdf;       /* shape_samplers.py:87 (normalised detection map) + its cdf */
  double T, alpha, T_target;
  int64_t step;
  int forced;            /* -1: Metropolis test; 0 / 1: decision imposed by orc_replay_forced */
  /* scratch */
  int *mark; int mark_gen;
};

/* ------------------------------------------------------------------ philox */
static inline void mulhilo(uint32_t a, uint32_t b, uint32_t *hi, uint32_t *lo) {
  uint64_t p = (uint64_t)a * b;
  *hi = (uint32_t)(p >> 32);
  *lo = (uint32_t)p;
}
void orc_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0, lo0, hi1, lo1;
    mulhilo(0xD2511F53u, c0, &hi0, &lo0);
    mulhilo(0xCD9E8D57u, c2, &hi1, &lo1);
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static inline double u53(uint32_t a, uint32_t b) {
  return (double)(((uint64_t)(a >> 5) << 26) | (uint64_t)(b >> 6)) * (1.0 / 9007199254740992.0);
}
static inline double u32d(uint32_t a) { return (double)a * (1.0 / 4294967296.0); }
static inline uint32_t mulhi(uint32_t a, uint32_t n) { return (uint32_t)(((uint64_t)a * n) >> 32); }

/* ---------------------------------------------------------------- geometry */
/* base/shapes/rectangle.py:20-31, :69-100 : corners = R(angle+pi/2)*(+-length/2, +-width/2) + centre */
static void rect_corners(const rect_t *q, double px[4], double py[4]) {
  double length = (2.0 * q->s) / (1.0 + q->r);
  double width = q->r * length;
  double hl = length / 2.0, hw = width / 2.0;
  double al = q->a + M_PI / 2.0;
  double c = cos(al), s = sin(al);
  static const double sx[4] = {1, -1, -1, 1}, sy[4] = {1, 1, -1, -1}; /* counter-clockwise */
  for (int i = 0; i < 4; ++i) {
    double vx = sx[i] * hl, vy = sy[i] * hw;
    px[i] = c * vx - s * vy + (double)q->x;
    py[i] = s * vx + c * vy + (double)q->y;
  }
}
static double poly_area(const double *x, const double *y, int n) {
  if (n < 3) return 0.0;
  double s = 0.0;
  for (int i = 0; i < n; ++i) {
    int j = (i + 1 == n) ? 0 : i + 1;
    s += x[i] * y[j] - x[j] * y[i];
  }
  return 0.5 * fabs(s);
}
static double rect_area(const rect_t *q) {
  double x[4], y[4];
  rect_corners(q, x, y);
  return poly_area(x, y, 4);
}
/* Sutherland-Hodgman: clip convex subject by convex counter-clockwise clipper */
static double clip_area(const double *sx, const double *sy, const double *cx, const double *cy) {
  double ax[16], ay[16], bx[16], by[16];
  int na = 4;
  memcpy(ax, sx, 4 * sizeof(double));
  memcpy(ay, sy, 4 * sizeof(double));
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
        if (sp < 0) {
          double t = sp / (sp - sq);
          bx[nb] = px + t * (qx - px); by[nb] = py + t * (qy - py); ++nb;
        }
        bx[nb] = qx; by[nb] = qy; ++nb;
      } else if (sp >= 0) {
        double t = sp / (sp - sq);
        bx[nb] = px + t * (qx - px); by[nb] = py + t * (qy - py); ++nb;
      }
      px = qx; py = qy; sp = sq;
    }
    memcpy(ax, bx, nb * sizeof(double));
    memcpy(ay, by, nb * sizeof(double));
    na = nb;
  }
  return poly_area(ax, ay, na);
}
static int rect_less(const rect_t *a, const rect_t *b) {
  if (a->x != b->x) return a->x < b->x;
  if (a->y != b->y) return a->y < b->y;
  if (a->s != b->s) return a->s < b->s;
  if (a->r != b->r) return a->r < b->r;
  return a->a < b->a;
}
/* prior_energies.py:11-24.  The pair value is made a function of the unordered pair by
 * always clipping the lexicographically smaller rectangle against the larger one. */
static double overlap_energy(const rect_t *u, const rect_t *v) {
  const rect_t *a = rect_less(v, u) ? v : u;
  const rect_t *b = (a == u) ? v : u;
  double ax[4], ay[4], bx[4], by[4];
  rect_corners(a, ax, ay);
  rect_corners(b, bx, by);
  double A = poly_area(ax, ay, 4), B = poly_area(bx, by, 4);
  double mn = A < B ? A : B;
  double inter = 0.0;
  if (mn >= DEGENERATE_AREA) inter = clip_area(ax, ay, bx, by);
  return inter / (mn + AREA_EPS);
}
double orc_overlap(const double r1[5], const double r2[5]) {
  rect_t a = {(int32_t)r1[0], (int32_t)r1[1], r1[2], r1[3], r1[4]};
  rect_t b = {(int32_t)r2[0], (int32_t)r2[1], r2[2], r2[3], r2[4]};
  return overlap_energy(&a, &b);
}

/* ------------------------------------------------------------------- marks */
/* mappings.py:44-62 : class = max{i : v >= edge_i} */
static int value_to_class(const orc_ctx *c, int k, double v) {
  int cls = 0;
  for (int i = 0; i < ORC_NCLASS; ++i)
    if (v >= c->kern.edges[k][i]) cls = i;
  return cls;
}
static inline double mark_of(const rect_t *q, int k) { return k == 0 ? q->s : (k == 1 ? q->r : q->a); }
static inline const float *mark_row(const orc_ctx *c, int k, int x, int y) {
  return c->m[k] + ((size_t)x * c->W + y) * ORC_NCLASS;
}
static double sigmoid(double x) { return 1.0 / (1.0 + exp(-x)); } /* utils/math_utils.py:6 */


/* -------------------------------------------------- classic image energies */
/* models/mpp/energies/classics.py:100-238.  The rasterisation lives in scikit-image 0.18.1, which is absent from the
 * container ("parity unpinned" at that primitive, as for shapely): restated here from its published algorithm --
 * skimage/draw/_draw.pyx `_polygon` (every pixel of the clipped bounding box put to the even-odd crossing test
 * `point_in_polygon` of skimage/_shared/geometry.pyx), `_line` (Bresenham) and draw.py `polygon_perimeter` (corners
 * rounded half-to-even, the four edges in vertex order, closed).  utils/morpho.py:9-19 is the reference's own dilation
 * (4-neighbourhood, clipped to the image after every iteration). */
static void ref_corners(const rect_t *q, double r[4], double c[4]) {
  /* rect_to_poly's vertex order (rectangle.py:85-88): (+,+) (+,-) (-,-) (-,+) = corners 0, 3, 2, 1 of rect_corners() */
  double x[4], y[4];
  static const int ord[4] = {0, 3, 2, 1};
  rect_corners(q, x, y);
  for (int i = 0; i < 4; ++i) { r[i] = x[ord[i]]; c[i] = y[ord[i]]; }
}
static int point_in_polygon(const double *xp, const double *yp, double x, double y) {
  int c = 0, j = 3;
  for (int i = 0; i < 4; ++i) {
    if ((((yp[i] <= y) && (y < yp[j])) || ((yp[j] <= y) && (y < yp[i]))) &&
        (x < (xp[j] - xp[i]) * (y - yp[i]) / (yp[j] - yp[i]) + xp[i])) c = !c;
    j = i;
  }
  return c;
}
typedef struct { int r0, c0, nr, nc; unsigned char *fill, *rim, *t0, *t1; } cgrid_t;
static void grid_dilate(const orc_ctx *c, const cgrid_t *g, unsigned char *a, unsigned char *tmp, int n_iter) {
  for (int it = 0; it < n_iter; ++it) {
    memcpy(tmp, a, (size_t)g->nr * g->nc);
    for (int i = 0; i < g->nr; ++i)
      for (int j = 0; j < g->nc; ++j) {
        if (tmp[i * g->nc + j]) continue;
        int on = (i > 0 && tmp[(i - 1) * g->nc + j]) || (i + 1 < g->nr && tmp[(i + 1) * g->nc + j]) ||
                 (j > 0 && tmp[i * g->nc + j - 1]) || (j + 1 < g->nc && tmp[i * g->nc + j + 1]);
        int r = g->r0 + i, cc = g->c0 + j;
        if (on && r >= 0 && r < c->H && cc >= 0 && cc < c->W) a[i * g->nc + j] = 1;
      }
  }
}
/* ContrastEnergy.compute_masks (classics.py:171-193); returns 0 when the fill mask is empty */
static int contrast_masks(const orc_ctx *c, const rect_t *q, int dilation, int gap, int erode, cgrid_t *g) {
  double r[4], cc[4];
  ref_corners(q, r, cc);
  double rmin = r[0], rmax = r[0], cmin = cc[0], cmax = cc[0];
  for (int i = 1; i < 4; ++i) {
    if (r[i] < rmin) rmin = r[i]; if (r[i] > rmax) rmax = r[i];
    if (cc[i] < cmin) cmin = cc[i]; if (cc[i] > cmax) cmax = cc[i];
  }
  int minr = (int)(rmin > 0 ? rmin : 0), maxr = (int)ceil(rmax), minc = (int)(cmin > 0 ? cmin : 0), maxc = (int)ceil(cmax);
  if (maxr > c->H - 1) maxr = c->H - 1;
  if (maxc > c->W - 1) maxc = c->W - 1;
  const int M = 3 + (erode > gap + dilation ? erode : gap + dilation);
  g->r0 = minr - M; g->c0 = minc - M;
  g->nr = (maxr >= minr ? maxr - minr + 1 : 0) + 2 * M; g->nc = (maxc >= minc ? maxc - minc + 1 : 0) + 2 * M;
  size_t sz = (size_t)g->nr * g->nc;
  g->fill = (unsigned char *)calloc(sz, 1); g->rim = (unsigned char *)calloc(sz, 1);
  g->t0 = (unsigned char *)calloc(sz, 1); g->t1 = (unsigned char *)calloc(sz, 1);
  int n_fill = 0;
  for (int rr = minr; rr <= maxr; ++rr)
    for (int c_ = minc; c_ <= maxc; ++c_)
      if (point_in_polygon(cc, r, (double)c_, (double)rr)) { g->fill[(rr - g->r0) * g->nc + (c_ - g->c0)] = 1; ++n_fill; }
  if (n_fill == 0) return 0;
  if (erode > 0) {                       /* :178-182 */
    memcpy(g->t0, g->fill, sz);
    grid_dilate(c, g, g->t0, g->t1, 2);
    for (size_t i = 0; i < sz; ++i) g->t0[i] = g->t0[i] && !g->fill[i];     /* rim of the raw mask */
    grid_dilate(c, g, g->t0, g->t1, erode);
    n_fill = 0;
    for (size_t i = 0; i < sz; ++i) { g->fill[i] = g->fill[i] && !g->t0[i]; n_fill += g->fill[i]; }
    if (n_fill == 0) return 0;
  }
  if (gap > 0) {                         /* :187-190 */
    memcpy(g->t0, g->fill, sz);
    grid_dilate(c, g, g->t0, g->t1, gap);
    memcpy(g->rim, g->t0, sz);
    grid_dilate(c, g, g->rim, g->t1, dilation);
    for (size_t i = 0; i < sz; ++i) g->rim[i] = g->rim[i] && !g->t0[i];
  } else {                               /* :191-193 */
    memcpy(g->rim, g->fill, sz);
    grid_dilate(c, g, g->rim, g->t1, dilation);
    for (size_t i = 0; i < sz; ++i) g->rim[i] = g->rim[i] && !g->fill[i];
  }
  return n_fill;
}
static void grid_free(cgrid_t *g) { free(g->fill); free(g->rim); free(g->t0); free(g->t1); }
static void mask_stats(const orc_ctx *c, const cgrid_t *g, const unsigned char *m, int ch, double *mean, double *var, int *cnt) {
  double s = 0.0; int n = 0;
  for (int i = 0; i < g->nr; ++i) for (int j = 0; j < g->nc; ++j)
    if (m[i * g->nc + j]) { s += (double)c->img[((size_t)(g->r0 + i) * c->W + (g->c0 + j)) * c->img_c + ch]; ++n; }
  double mu = s / (double)n, v = 0.0;
  for (int i = 0; i < g->nr; ++i) for (int j = 0; j < g->nc; ++j)
    if (m[i * g->nc + j]) { double d = (double)c->img[((size_t)(g->r0 + i) * c->W + (g->c0 + j)) * c->img_c + ch] - mu; v += d * d; }
  *mean = mu; *var = v / (double)n; *cnt = n;
}
/* the contrast measures, classics.py:13-97 */
static double contrast_measure(int type, double mi, double mo, double vi, double vo, int ni, int no) {
  const double eps = 1e-8, d = mi - mo;
  switch (type) {
    case 0: return sqrt((vo + vi) / ((double)(no + ni) * (d * d) + eps));                         /* lafarge :13-28 */
    case 1: return (d * d) / (4.0 * sqrt(vi + vo)) + (-0.5 * log((2.0 * sqrt(vi * vo)) / (vi + vo)));   /* craciun :31-51 */
    case 2: return (d * d) / (4.0 * sqrt(vi + vo) + eps);                                          /* craciun2 :67-82 */
    case 3: return d * d;                                                                          /* mean :94-97 */
    case 4: return fabs(d) / sqrt((vi / (double)ni) + (vo / (double)no) + eps);                    /* t-test :85-91 */
    default: return fabs(d);                                                                       /* debug :57-64 */
  }
}
/* ContrastEnergy.compute (classics.py:151-169): p = {measure, dilation, gap, erode, thresh, fac, default_value} */
static double contrast_value(const orc_ctx *c, const orc_unit_term *t, const rect_t *q) {
  cgrid_t g;
  double val;
  if (!contrast_masks(c, q, (int)t->p[1], (int)t->p[2], (int)t->p[3], &g)) val = t->p[6];
  else {
    val = 0.0;
    for (int ch = 0; ch < c->img_c; ++ch) {
      double mi, mo, vi, vo; int ni, no;
      mask_stats(c, &g, g.fill, ch, &mi, &vi, &ni);
      mask_stats(c, &g, g.rim, ch, &mo, &vo, &no);
      val += t->p[5] * contrast_measure((int)t->p[0], mi, mo, vi, vo, ni, no);
    }
    val -= t->p[4];
  }
  grid_free(&g);
  return val;
}
/* skimage/draw/_draw.pyx `_line` */
static int bresenham(int r0, int c0, int r1, int c1, int *rr, int *cc) {
  int steep = 0, r = r0, c = c0, dr = abs(r1 - r0), dc = abs(c1 - c0);
  int sc = (c1 - c) > 0 ? 1 : -1, sr = (r1 - r) > 0 ? 1 : -1;
  if (dr > dc) { steep = 1; int t = c; c = r; r = t; t = dc; dc = dr; dr = t; t = sc; sc = sr; sr = t; }
  int d = 2 * dr - dc;
  for (int i = 0; i < dc; ++i) {
    if (steep) { rr[i] = c; cc[i] = r; } else { rr[i] = r; cc[i] = c; }
    while (d >= 0) { r += sr; d -= 2 * dc; }
    c += sc; d += 2 * dr;
  }
  rr[dc] = r1; cc[dc] = c1;
  return dc + 1;
}
/* GradientEnergy.compute_outline_and_normal (classics.py:218-232); returns the number of outline pixels */
static int outline_and_normals(const orc_ctx *c, const rect_t *q, double eps, int **pr_out, int **pc_out, double **nrm_out) {
  double r[5], cc[5];
  ref_corners(q, r, cc);
  r[4] = r[0]; cc[4] = cc[0];                 /* polygon_clip closes the polygon */
  int nv = 5;
  if (r[4] == r[3] && cc[4] == cc[3]) nv = 4; /* its "last two vertices equal" rule */
  int vr[5], vc[5], total = 0;
  for (int i = 0; i < nv; ++i) { vr[i] = (int)rint(r[i]); vc[i] = (int)rint(cc[i]); }
  for (int i = 0; i + 1 < nv; ++i) { int a = abs(vr[i + 1] - vr[i]), b = abs(vc[i + 1] - vc[i]); total += (a > b ? a : b) + 1; }
  int *pr = (int *)malloc(sizeof(int) * (total + 1)), *pc = (int *)malloc(sizeof(int) * (total + 1));
  int n = 0;
  for (int i = 0; i + 1 < nv; ++i) n += bresenham(vr[i], vc[i], vr[i + 1], vc[i + 1], pr + n, pc + n);
  int m = 0;                                  /* _coords_inside_image */
  for (int i = 0; i < n; ++i)
    if (pr[i] >= 0 && pr[i] < c->H && pc[i] >= 0 && pc[i] < c->W) { pr[m] = pr[i]; pc[m] = pc[i]; ++m; }
  double *nrm = (double *)malloc(sizeof(double) * 2 * (m + 1));
  for (int i = 0; i < m; ++i) {
    int nx = i + 1 < m ? i + 1 : 0, pv = i > 0 ? i - 1 : m - 1;
    double t1r = pr[nx] - pr[i], t1c = pc[nx] - pc[i];      /* tangent_1 */
    double t2r = pr[pv] - pr[i], t2c = pc[pv] - pc[i];      /* tangent_2 after the two flips */
    double n1r = -t1c, n1c = t1r, n2r = t2c, n2c = -t2r;    /* :224, :226 (expanded signs) */
    double l1 = sqrt(n1r * n1r + n1c * n1c) + eps, l2 = sqrt(n2r * n2r + n2c * n2c) + eps;
    nrm[2 * i] = 0.5 * (n1r / l1 + n2r / l2);
    nrm[2 * i + 1] = 0.5 * (n1c / l1 + n2c / l2);
  }
  *pr_out = pr; *pc_out = pc; *nrm_out = nrm;
  return m;
}
/* GradientEnergy.compute (classics.py:207-216): the image holds np.gradient of the picture, [H][W][C/2][2];
 * p = {thresh, eps} */
static double gradient_value(const orc_ctx *c, const orc_unit_term *t, const rect_t *q) {
  int *pr, *pc; double *nrm;
  int m = outline_and_normals(c, q, t->p[1], &pr, &pc, &nrm);
  double s = 0.0;
  for (int i = 0; i < m; ++i) {
    const float *g = c->img + ((size_t)pr[i] * c->W + pc[i]) * c->img_c;
    for (int k = 0; k < c->img_c; ++k) s += (double)g[k] * nrm[2 * i + (k & 1)];
  }
  double mean = s / ((double)m * (double)c->img_c);
  free(pr); free(pc); free(nrm);
  return -fabs(mean) - t->p[0];
}

/* -------------------------------------------------------------- unit terms */
static double unit_value(const orc_ctx *c, const orc_unit_term *t, const rect_t *q) {
  switch (t->kind) {
    case ORC_U_POSITION: {
      /* float32 arithmetic, as numpy does on the float32 map (data_energies.py:17-18) */
      float e = -2.0f * (c->det[(size_t)q->x * c->W + q->y] - (float)t->p[0]);
      return (double)e;
    }
    case ORC_U_SHAPE_REMAP: {
      double acc = 0.0;
      for (int k = 0; k < 3; ++k) {
        double p = mark_row(c, k, q->x, q->y)[value_to_class(c, k, mark_of(q, k))];
        acc += -2.0 * sigmoid(p * t->p[k] + t->p[3 + k]) + 1.0;
      }
      return acc / 3.0;
    }
    case ORC_U_MARK_NEG: {
      int k = (int)t->p[0];
      return -(double)mark_row(c, k, q->x, q->y)[value_to_class(c, k, mark_of(q, k))];
    }
    case ORC_U_MARK_REMAP: {
      int k = (int)t->p[0];
      double p = mark_row(c, k, q->x, q->y)[value_to_class(c, k, mark_of(q, k))];
      return -2.0 * sigmoid(p * t->p[1] + t->p[2]) + 1.0;
    }
    case ORC_U_AREA: {
      double A = rect_area(q), lo = t->p[0] - A, hi = A - t->p[1];
      double m = lo > hi ? lo : hi;
      return m > 0.0 ? m : 0.0;
    }
    case ORC_U_RATIO_PRIOR: return fabs(t->p[0] - q->r);
    case ORC_U_CONST: return t->p[0];
    case ORC_U_CONTRAST: return contrast_value(c, t, q);
    case ORC_U_GRADIENT: return gradient_value(c, t, q);
  }
  return 0.0;
}
static double pair_value(const orc_pair_term *t, const rect_t *u, const rect_t *v, double d) {
  switch (t->kind) {
    case ORC_P_OVERLAP: return overlap_energy(u, v);
    case ORC_P_ALIGN: return 1.0 - fabs(cos(u->a - v->a)) - (t->p[0] != 0.0 ? 1.0 : 0.0);
    case ORC_P_DIST_LE: return d <= t->max_dist ? 1.0 : 0.0;
    case ORC_P_DIST_LT: return d < t->max_dist ? 1.0 : 0.0;
  }
  return 0.0;
}

/* -------------------------------------------------------------------- grid */
static inline int cell_i(const orc_ctx *c, int x) { return (int)floor((double)x / c->res); }
static inline int cell_of(const orc_ctx *c, int x, int y) { return cell_i(c, y) + cell_i(c, x) * c->ny; }
static void cell_add(orc_ctx *c, int cell, int slot) {
  if (c->cell_cnt[cell] == c->cell_cap[cell]) {
    c->cell_cap[cell] = c->cell_cap[cell] ? 2 * c->cell_cap[cell] : 8;
    c->cell_items[cell] = (int *)realloc(c->cell_items[cell], sizeof(int) * c->cell_cap[cell]);
  }
  c->cell_items[cell][c->cell_cnt[cell]++] = slot;
}
static void cell_del(orc_ctx *c, int cell, int slot) {
  int *it = c->cell_items[cell], n = c->cell_cnt[cell];
  for (int i = 0; i < n; ++i)
    if (it[i] == slot) { it[i] = it[n - 1]; c->cell_cnt[cell] = n - 1; return; }
}
static void cell_rename(orc_ctx *c, int cell, int from, int to) {
  int *it = c->cell_items[cell], n = c->cell_cnt[cell];
  for (int i = 0; i < n; ++i)
    if (it[i] == from) { it[i] = to; return; }
}
static int state_add(orc_ctx *c, const rect_t *q) {
  if (c->n == c->cap) {
    c->cap = c->cap ? 2 * c->cap : 256;
    c->pt = (rect_t *)realloc(c->pt, sizeof(rect_t) * c->cap);
    c->mark = (int *)realloc(c->mark, sizeof(int) * c->cap);
    for (int i = c->n; i < c->cap; ++i) c->mark[i] = 0;
  }
  c->pt[c->n] = *q;
  cell_add(c, cell_of(c, q->x, q->y), c->n);
  return c->n++;
}
static void state_remove(orc_ctx *c, int slot) {
  int last = c->n - 1;
  cell_del(c, cell_of(c, c->pt[slot].x, c->pt[slot].y), slot);
  if (slot != last) {
    cell_rename(c, cell_of(c, c->pt[last].x, c->pt[last].y), last, slot);
    c->pt[slot] = c->pt[last];
  }
  c->n = last;
}
static void state_update(orc_ctx *c, int slot, const rect_t *q) {
  int c0 = cell_of(c, c->pt[slot].x, c->pt[slot].y), c1 = cell_of(c, q->x, q->y);
  if (c0 != c1) { cell_del(c, c0, slot); cell_add(c, c1, slot); }
  c->pt[slot] = *q;
}

/* per-point energy vector in the state (current - excluded slots + extra rectangles);
 * self_slot / self_extra identify the point itself (energy_graph.py:108-137). */
typedef struct {
  int n_excl; const int32_t *excl;
  int n_extra; const rect_t *extra;
} overlay_t;

static int is_excluded(const overlay_t *o, int slot) {
  for (int i = 0; i < o->n_excl; ++i) if (o->excl[i] == slot) return 1;
  return 0;
}
static void point_vector(const orc_ctx *c, const rect_t *u, int self_slot, int self_extra, const overlay_t *o,
                         double *vec) {
  const orc_model *m = &c->model;
  for (int k = 0; k < m->n_unit; ++k) vec[k] = unit_value(c, &m->unit[k], u);
  int have[ORC_MAX_PAIR] = {0, 0};
  double red[ORC_MAX_PAIR] = {0, 0};
  int ci = cell_i(c, u->x), cj = cell_i(c, u->y);
  for (int di = -c->max_offset; di <= c->max_offset; ++di)
    for (int dj = -c->max_offset; dj <= c->max_offset; ++dj) {
      int i = ci + di, j = cj + dj;
      if (i < 0 || i >= c->nx || j < 0 || j >= c->ny) continue;
      int cell = j + i * c->ny;
      for (int e = 0; e < c->cell_cnt[cell]; ++e) {
        int v = c->cell_items[cell][e];
        if (v == self_slot || is_excluded(o, v)) continue;
        const rect_t *q = &c->pt[v];
        double dx = (double)(u->x - q->x), dy = (double)(u->y - q->y);
        double d = sqrt(dx * dx + dy * dy);
        for (int p = 0; p < m->n_pair; ++p)
          if (d <= m->pair[p].max_dist) {       /* energy_graph.py:72-77 */
            double val = pair_value(&m->pair[p], u, q, d);
            if (!have[p]) { red[p] = val; have[p] = 1; }
            else if (m->pair[p].reduce == ORC_REDUCE_MAX) { if (val > red[p]) red[p] = val; }
            else if (val < red[p]) red[p] = val;
          }
      }
    }
  for (int e = 0; e < o->n_extra; ++e) {
    if (e == self_extra) continue;
    const rect_t *q = &o->extra[e];
    /* extras interact only if they would be potential neighbours (same 3x3 cells) */
    int qi = cell_i(c, q->x), qj = cell_i(c, q->y);
    if (abs(qi - ci) > c->max_offset || abs(qj - cj) > c->max_offset) continue;
    double dx = (double)(u->x - q->x), dy = (double)(u->y - q->y);
    double d = sqrt(dx * dx + dy * dy);
    for (int p = 0; p < m->n_pair; ++p)
      if (d <= m->pair[p].max_dist) {
        double val = pair_value(&m->pair[p], u, q, d);
        if (!have[p]) { red[p] = val; have[p] = 1; }
        else if (m->pair[p].reduce == ORC_REDUCE_MAX) { if (val > red[p]) red[p] = val; }
        else if (val < red[p]) red[p] = val;
      }
  }
  for (int p = 0; p < m->n_pair; ++p) vec[m->n_unit + p] = have[p] ? red[p] : 0.0;
}
/* combinators: hierarchical.py:21-32 / :41-48, logistic.py:20-26, plain sum energy_graph.py:131 */
static double combine(const orc_model *m, const double *vec) {
  double gate = 1.0;
  if (m->gate_term >= 0) gate = (vec[m->gate_term] <= m->gate_thr) ? 1.0 : 0.0;
  double lin = m->lin0;
  for (int k = 0; k < m->n_unit; ++k) lin += m->unit[k].coef * (m->unit[k].gated ? gate : 1.0) * vec[k];
  for (int p = 0; p < m->n_pair; ++p)
    lin += m->pair[p].coef * (m->pair[p].gated ? gate : 1.0) * vec[m->n_unit + p];
  if (m->combinator == ORC_C_LOGISTIC) return 2.0 * sigmoid(lin) - 1.0;
  return lin;
}

/* ----------------------------------------------------------------- context */
orc_ctx *orc_create(int H, int W, const float *det, const float *m0, const float *m1, const float *m2,
                    const orc_model *model, const orc_kernels *kernels) {
  orc_ctx *c = (orc_ctx *)calloc(1, sizeof(orc_ctx));
  c->H = H; c->W = W;
  size_t hw = (size_t)H * W;
  c->det = (float *)malloc(hw * sizeof(float));
  if (det) memcpy(c->det, det, hw * sizeof(float)); else memset(c->det, 0, hw * sizeof(float));
  const float *ms[3] = {m0, m1, m2};
  for (int k = 0; k < 3; ++k) {
    c->m[k] = (float *)malloc(hw * ORC_NCLASS * sizeof(float));
    if (ms[k]) memcpy(c->m[k], ms[k], hw * ORC_NCLASS * sizeof(float));
    else memset(c->m[k], 0, hw * ORC_NCLASS * sizeof(float));
  }
  c->model = *model;
  if (kernels) c->kern = *kernels;
  double acc = 0.0;
  for (int k = 0; k < ORC_NKERNEL; ++k) { acc += c->kern.p_kernel[k]; c->p_cum[k] = acc; }
  c->n_active = (c->kern.p_kernel[ORC_K_SPLIT] > 0.0 || c->kern.p_kernel[ORC_K_MERGE] > 0.0) ? ORC_NKERNEL : ORC_K_SPLIT;
  double maxd = 0.0;
  for (int p = 0; p < model->n_pair; ++p) if (model->pair[p].max_dist > maxd) maxd = model->pair[p].max_dist;
  /* energy_point_set.py:33-36, point_set.py:58-61, energy_graph.py:26-29 */
  c->res = maxd > 32.0 ? maxd : 32.0;
  c->max_inter = model->n_pair > 0 ? maxd : 1.0;
  c->max_offset = (int)ceil(c->max_inter / c->res);
  c->nx = (int)ceil((double)H / c->res);
  c->ny = (int)ceil((double)W / c->res);
  int nc = c->nx * c->ny;
  c->cell_cnt = (int *)calloc(nc, sizeof(int));
  c->cell_cap = (int *)calloc(nc, sizeof(int));
  c->cell_items = (int **)calloc(nc, sizeof(int *));
  /* normalised detection map and its cdf: shape_samplers.py:87, utils/sampler2d.py:43 */
  c->cdf = (double *)malloc(hw * sizeof(double));
  double s = 0.0;
  for (size_t i = 0; i < hw; ++i) { s += (double)c->det[i]; c->cdf[i] = s; }
  c->det_sum = s;
  if (s > 0) for (size_t i = 0; i < hw; ++i) c->cdf[i] /= s;
  c->T = 1.0; c->alpha = 1.0; c->T_target = 0.0; c->step = 0; c->forced = -1;
  return c;
}
void orc_destroy(orc_ctx *c) {
  if (!c) return;
  int nc = c->nx * c->ny;
  for (int i = 0; i < nc; ++i) free(c->cell_items[i]);
  free(c->cell_items); free(c->cell_cnt); free(c->cell_cap);
  free(c->det); for (int k = 0; k < 3; ++k) free(c->m[k]);
  free(c->img);
  free(c->cdf); free(c->pt); free(c->mark); free(c);
}
int orc_set_image(orc_ctx *c, int C, const float *img) {
  free(c->img);
  size_t n = (size_t)c->H * c->W * C;
  c->img = (float *)malloc(n * sizeof(float));
  memcpy(c->img, img, n * sizeof(float));
  c->img_c = C;
  return 0;
}
/* test hooks: the two pixel sets of ContrastEnergy.compute_masks / the outline of GradientEnergy (row, col pairs) */
int orc_contrast_masks(orc_ctx *c, const double rect[5], int dilation, int gap, int erode, int cap, int32_t *fill_rc,
                       int32_t *n_fill, int32_t *rim_rc, int32_t *n_rim) {
  rect_t q = {(int32_t)rect[0], (int32_t)rect[1], rect[2], rect[3], rect[4]};
  cgrid_t g;
  int nf = 0, nr = 0;
  if (contrast_masks(c, &q, dilation, gap, erode, &g))
    for (int i = 0; i < g.nr; ++i) for (int j = 0; j < g.nc; ++j) {
      if (g.fill[i * g.nc + j]) { if (nf < cap) { fill_rc[2 * nf] = g.r0 + i; fill_rc[2 * nf + 1] = g.c0 + j; } ++nf; }
      if (g.rim[i * g.nc + j]) { if (nr < cap) { rim_rc[2 * nr] = g.r0 + i; rim_rc[2 * nr + 1] = g.c0 + j; } ++nr; }
    }
  grid_free(&g);
  *n_fill = nf; *n_rim = nr;
  return 0;
}
int orc_outline(orc_ctx *c, const double rect[5], double eps, int cap, int32_t *rc, double *normals) {
  rect_t q = {(int32_t)rect[0], (int32_t)rect[1], rect[2], rect[3], rect[4]};
  int *pr, *pc; double *nrm;
  int m = outline_and_normals(c, &q, eps, &pr, &pc, &nrm);
  for (int i = 0; i < m && i < cap; ++i) { rc[2 * i] = pr[i]; rc[2 * i + 1] = pc[i]; normals[2 * i] = nrm[2 * i]; normals[2 * i + 1] = nrm[2 * i + 1]; }
  free(pr); free(pc); free(nrm);
  return m;
}
double orc_unit_value(orc_ctx *c, const orc_unit_term *t, const double rect[5]) {
  rect_t q = {(int32_t)rect[0], (int32_t)rect[1], rect[2], rect[3], rect[4]};
  return unit_value(c, t, &q);
}
int orc_set_points(orc_ctx *c, int n, const int32_t *xy, const double *marks) {
  int nc = c->nx * c->ny;
  for (int i = 0; i < nc; ++i) c->cell_cnt[i] = 0;
  c->n = 0;
  for (int i = 0; i < n; ++i) {
    rect_t q = {xy[2 * i], xy[2 * i + 1], marks[3 * i], marks[3 * i + 1], marks[3 * i + 2]};
    if (q.x < 0 || q.x >= c->H || q.y < 0 || q.y >= c->W) return -1; /* point_set.py:99 */
    state_add(c, &q);
  }
  return 0;
}
int orc_get_points(orc_ctx *c, int cap, int32_t *xy, double *marks) {
  int n = c->n < cap ? c->n : cap;
  for (int i = 0; i < n; ++i) {
    xy[2 * i] = c->pt[i].x; xy[2 * i + 1] = c->pt[i].y;
    marks[3 * i] = c->pt[i].s; marks[3 * i + 1] = c->pt[i].r; marks[3 * i + 2] = c->pt[i].a;
  }
  return c->n;
}
int orc_count(orc_ctx *c) { return c->n; }
int64_t orc_step_index(orc_ctx *c) { return c->step; }
void orc_set_step_index(orc_ctx *c, int64_t step) { c->step = step; }
double orc_temperature(orc_ctx *c) { return c->T; }
void orc_set_temperature(orc_ctx *c, double T, double alpha, double T_target) {
  c->T = T; c->alpha = alpha; c->T_target = T_target;
}

double orc_total_energy(orc_ctx *c, double *vectors) {
  overlay_t o = {0, NULL, 0, NULL};
  int nt = c->model.n_unit + c->model.n_pair;
  double vec[ORC_MAX_UNIT + ORC_MAX_PAIR], E = 0.0;
  for (int i = 0; i < c->n; ++i) {
    point_vector(c, &c->pt[i], i, -1, &o, vec);
    if (vectors) memcpy(vectors + (size_t)i * nt, vec, nt * sizeof(double));
    E += combine(&c->model, vec);
  }
  return E;
}

/* energy_graph.py:139-225 */
static double delta_rects(orc_ctx *c, int n_rem, const int32_t *rem, int n_add, const rect_t *add) {
  /* unchanged = potential neighbours (cells) of every added/removed point, in the CURRENT state */
  int gen = ++c->mark_gen;
  int *unch = (int *)malloc(sizeof(int) * (c->n + 1));
  int n_unch = 0;
  for (int i = 0; i < n_rem; ++i) c->mark[rem[i]] = gen; /* removed points are never "unchanged" */
  for (int pass = 0; pass < 2; ++pass) {
    int cnt = pass == 0 ? n_add : n_rem;
    for (int e = 0; e < cnt; ++e) {
      int x = pass == 0 ? add[e].x : c->pt[rem[e]].x, y = pass == 0 ? add[e].y : c->pt[rem[e]].y;
      int ci = cell_i(c, x), cj = cell_i(c, y);
      for (int di = -c->max_offset; di <= c->max_offset; ++di)
        for (int dj = -c->max_offset; dj <= c->max_offset; ++dj) {
          int i = ci + di, j = cj + dj;
          if (i < 0 || i >= c->nx || j < 0 || j >= c->ny) continue;
          int cell = j + i * c->ny;
          for (int k = 0; k < c->cell_cnt[cell]; ++k) {
            int v = c->cell_items[cell][k];
            if (c->mark[v] != gen) { c->mark[v] = gen; unch[n_unch++] = v; }
          }
        }
    }
  }
  overlay_t none = {0, NULL, 0, NULL};
  overlay_t after = {n_rem, rem, n_add, add};
  double vec[ORC_MAX_UNIT + ORC_MAX_PAIR];
  double e0 = 0.0, e1 = 0.0;
  for (int i = 0; i < n_unch; ++i) {
    point_vector(c, &c->pt[unch[i]], unch[i], -1, &none, vec);
    e0 += combine(&c->model, vec);
  }
  for (int i = 0; i < n_rem; ++i) {
    point_vector(c, &c->pt[rem[i]], rem[i], -1, &none, vec);
    e0 += combine(&c->model, vec);
  }
  for (int i = 0; i < n_unch; ++i) {
    point_vector(c, &c->pt[unch[i]], unch[i], -1, &after, vec);
    e1 += combine(&c->model, vec);
  }
  for (int i = 0; i < n_add; ++i) {
    point_vector(c, &add[i], -1, i, &after, vec);
    e1 += combine(&c->model, vec);
  }
  free(unch);
  return e1 - e0;
}
double orc_delta(orc_ctx *c, int n_rem, const int32_t *rem, int n_add, const int32_t *add_xy,
                 const double *add_marks) {
  rect_t *add = (rect_t *)malloc(sizeof(rect_t) * (n_add + 1));
  for (int i = 0; i < n_add; ++i) {
    add[i].x = add_xy[2 * i]; add[i].y = add_xy[2 * i + 1];
    add[i].s = add_marks[3 * i]; add[i].r = add_marks[3 * i + 1]; add[i].a = add_marks[3 * i + 2];
  }
  double d = delta_rects(c, n_rem, rem, n_add, add);
  free(add);
  return d;
}
void orc_papangelou(orc_ctx *c, double *out) {
  for (int32_t i = 0; i < c->n; ++i) out[i] = -delta_rects(c, 1, &i, 0, NULL);
}

/* --------------------------------------------------------- proposal kernels */
static double row_sum(const float *row) {
  double s = 0.0;
  for (int i = 0; i < ORC_NCLASS; ++i) s += (double)row[i];
  return s;
}
/* shape_samplers.py:103-108 : (det/sum det)[x,y] * prod_k P_k[class_k] * (H*W*32^3); the mark rows
 * are the row-renormalised ones (transform_kernels.py:170-177 aliases the shared list) */
static double birth_density(const orc_ctx *c, const rect_t *q) {
  double d = (double)c->det[(size_t)q->x * c->W + q->y] / c->det_sum;
  for (int k = 0; k < 3; ++k) {
    const float *row = mark_row(c, k, q->x, q->y);
    d *= (double)row[value_to_class(c, k, mark_of(q, k))] / row_sum(row);
  }
  return d * ((double)c->H * c->W * 32768.0);
}
static double normal_pdf(double x, double sigma) {
  return exp(-(x * x) / (2.0 * sigma * sigma)) / (sigma * sqrt(TWO_PI));
}
/* transform_kernels.py:70-76,94-99 */
static void window_bounds(const orc_ctx *c, int x, int y, int *x0, int *x1, int *y0, int *y1) {
  int md = c->kern.max_delta;
  *x0 = x - md > 0 ? x - md : 0; *x1 = x + md + 1 < c->H ? x + md + 1 : c->H;
  *y0 = y - md > 0 ? y - md : 0; *y1 = y + md + 1 < c->W ? y + md + 1 : c->W;
}
static double move_density(const orc_ctx *c, int sx, int sy, int ex, int ey) {
  int x0, x1, y0, y1;
  window_bounds(c, sx, sy, &x0, &x1, &y0, &y1);
  double tot = 0.0;
  for (int x = x0; x < x1; ++x)
    for (int y = y0; y < y1; ++y) tot += (double)c->det[(size_t)x * c->W + y];
  return (double)c->det[(size_t)ex * c->W + ey] / tot;
}
static int sample_class(const float *row, double u) {
  double tot = row_sum(row), acc = 0.0;
  int cls = 0;
  for (int i = 0; i < ORC_NCLASS; ++i) {
    acc += (double)row[i];
    if (acc / tot <= u) cls = i + 1;
  }
  return cls < ORC_NCLASS ? cls : ORC_NCLASS - 1;
}
static void box_muller(uint32_t a, uint32_t b, double *z0, double *z1) {
  double u1 = ((double)a + 1.0) * (1.0 / 4294967296.0), u2 = u32d(b);
  double r = sqrt(-2.0 * log(u1)), th = TWO_PI * u2;
  *z0 = r * cos(th); *z1 = r * sin(th);
}
static double wrap_mark(const orc_ctx *c, int k, double v) {
  double lo = c->kern.vmin[k], hi = c->kern.vmax[k];
  if (c->kern.cyclic[k]) {
    double range = hi - lo, m = fmod(v, range);
    if (m < 0) m += range;               /* python % */
    return m + lo;                       /* transform_kernels.py:137 */
  }
  return v < lo ? lo : (v > hi ? hi : v);
}
static inline void set_mark(rect_t *q, int k, double v) { if (k == 0) q->s = v; else if (k == 1) q->r = v; else q->a = v; }

/* ValueMapping.clip (shape_net/mappings.py:52-58): modulo for cyclic marks, clamp otherwise */
static double clip_mark(const orc_ctx *c, int k, double v) {
  double lo = c->kern.vmin[k], hi = c->kern.vmax[k];
  if (c->kern.cyclic[k]) {
    double range = hi - lo, m = fmod(v - lo, range);
    if (m < 0) m += range;
    return m + lo;
  }
  return v < lo ? lo : (v > hi ? hi : v);
}
static inline int clip_int(double v, int hi) { return (int)(v < 0.0 ? 0.0 : (v > (double)hi ? (double)hi : v)); }
/* the two rectangles of a split (split_and_merge_kernels.py:56-73) */
static void split_rects(const orc_ctx *c, const rect_t *p, const orc_proposal *pr, rect_t *a0, rect_t *a1) {
  const double sd[3] = {pr->as, pr->ar, pr->aa};
  a0->x = clip_int((double)p->x - pr->aux0, c->H - 1); a0->y = clip_int((double)p->y - pr->aux1, c->W - 1);
  a1->x = clip_int((double)p->x + pr->aux0, c->H - 1); a1->y = clip_int((double)p->y + pr->aux1, c->W - 1);
  for (int k = 0; k < 3; ++k) {
    set_mark(a0, k, clip_mark(c, k, mark_of(p, k) - sd[k]));
    set_mark(a1, k, clip_mark(c, k, mark_of(p, k) + sd[k]));
  }
}
/* the rectangle two points merge into (split_and_merge_kernels.py:128-135; the column is clipped with
 * shape[0] upstream, reproduced) */
static void merge_rect(const orc_ctx *c, const rect_t *p0, const rect_t *p1, rect_t *q) {
  q->x = clip_int(((double)p0->x + (double)p1->x) / 2.0, c->H - 1);
  q->y = clip_int(((double)p0->y + (double)p1->y) / 2.0, c->H - 1);
  for (int k = 0; k < 3; ++k) set_mark(q, k, clip_mark(c, k, (mark_of(p0, k) + mark_of(p1, k)) / 2.0));
}
/* SplitSampler.pdf (split_and_merge_kernels.py:33-36) */
static double split_pdf(const orc_ctx *c, const double sd[3]) {
  double R = c->kern.split_radius, p = 1.0 / (PI_ * R * R);
  for (int k = 0; k < 3; ++k) p *= normal_pdf(sd[k], c->kern.split_sigma * (c->kern.vmax[k] - c->kern.vmin[k]));
  return p;
}
/* len(get_potential_neighbors(u, radius)): every point of the (2*ceil(r/res)+1)^2 cells (point_set.py:111-145) */
static int count_potential(const orc_ctx *c, int x, int y, double radius) {
  int off = (int)ceil(radius / c->res), ci = cell_i(c, x), cj = cell_i(c, y), cnt = 0;
  for (int di = -off; di <= off; ++di)
    for (int dj = -off; dj <= off; ++dj) {
      int i = ci + di, j = cj + dj;
      if (i < 0 || i >= c->nx || j < 0 || j >= c->ny) continue;
      cnt += c->cell_cnt[j + i * c->ny];
    }
  return cnt;
}
/* get_neighbors(p0, radius) (point_set.py:147-149) in the canonical order of this build: ascending (x, y, marks) */
static int merge_neighbours(const orc_ctx *c, int i0, int *out) {
  const rect_t *p0 = &c->pt[i0];
  double R = c->kern.split_radius;
  int off = (int)ceil(R / c->res), ci = cell_i(c, p0->x), cj = cell_i(c, p0->y), n = 0;
  for (int di = -off; di <= off; ++di)
    for (int dj = -off; dj <= off; ++dj) {
      int i = ci + di, j = cj + dj;
      if (i < 0 || i >= c->nx || j < 0 || j >= c->ny) continue;
      int cell = j + i * c->ny;
      for (int k = 0; k < c->cell_cnt[cell]; ++k) {
        int v = c->cell_items[cell][k];
        if (v == i0) continue;
        double dx = (double)(c->pt[v].x - p0->x), dy = (double)(c->pt[v].y - p0->y);
        if (sqrt(dx * dx + dy * dy) <= R) out[n++] = v;
      }
    }
  for (int a = 1; a < n; ++a) {                 /* insertion sort by rect_less */
    int v = out[a], b = a - 1;
    while (b >= 0 && rect_less(&c->pt[v], &c->pt[out[b]])) { out[b + 1] = out[b]; --b; }
    out[b + 1] = v;
  }
  return n;
}

/* draw the proposal of step `step` from Philox blocks 0 and 1 of the step (8 words); the accept uniform is words 6, 7
 * except for the kernels that need all eight (births, split: block 2); split draws further blocks 3.. for its
 * rejection sampling */
static void draw_proposal(const orc_ctx *c, const uint32_t w[8], orc_proposal *pr, const uint32_t key[2],
                          uint64_t step, uint32_t chain) {
  memset(pr, 0, sizeof(*pr));
  double uk = u53(w[0], w[1]);
  int k = 0;
  while (k < c->n_active - 1 && c->p_cum[k] <= uk) ++k;  /* Generator.choice: searchsorted(cdf, u, 'right') */
  pr->kernel = k; pr->target = -1; pr->param_id = -1; pr->new_class = -1;
  pr->aux0 = pr->aux1 = 0.0;
  if (k == ORC_K_UBIRTH || k == ORC_K_DBIRTH || k == ORC_K_SPLIT) {   /* these use all eight words themselves */
    uint32_t e[4], ctr[4] = {(uint32_t)step, (uint32_t)(step >> 32), 2u, chain};
    orc_philox(ctr, key, e);
    pr->u_accept = u53(e[2], e[3]);
  } else {
    pr->u_accept = u53(w[6], w[7]);
  }
  int n = c->n;
  if (k == ORC_K_UBIRTH) {               /* shape_samplers.py:136-141 */
    pr->ax = (int32_t)mulhi(w[3], (uint32_t)c->H); pr->ay = (int32_t)mulhi(w[4], (uint32_t)c->W);
    double v[3];
    for (int j = 0; j < 3; ++j) v[j] = c->kern.vmin[j] + (c->kern.vmax[j] - c->kern.vmin[j]) * u32d(w[5 + j]);
    pr->as = v[0]; pr->ar = v[1]; pr->aa = v[2];
    return;
  }
  if (k == ORC_K_DBIRTH) {               /* shape_samplers.py:90-98 */
    double u = u53(w[3], w[4]);
    size_t hw = (size_t)c->H * c->W, lo = 0, hi = hw;
    while (lo < hi) { size_t mid = (lo + hi) >> 1; if (c->cdf[mid] <= u) lo = mid + 1; else hi = mid; }
    if (lo >= hw) lo = hw - 1;
    pr->ax = (int32_t)(lo / c->W); pr->ay = (int32_t)(lo % c->W);
    double v[3];
    for (int j = 0; j < 3; ++j) v[j] = c->kern.edges[j][sample_class(mark_row(c, j, pr->ax, pr->ay), u32d(w[5 + j]))];
    pr->as = v[0]; pr->ar = v[1]; pr->aa = v[2];
    return;
  }
  if (n == 0) return;                     /* nothing to remove / move: empty perturbation */
  if (k == ORC_K_MERGE && n < 2) return;  /* split_and_merge_kernels.py:121 */
  int t = (int)mulhi(w[2], (uint32_t)n);  /* point_set.py:176-185 (uniform pick) */
  pr->target = t;
  rect_t q = c->pt[t];
  if (k == ORC_K_UDEATH || k == ORC_K_DDEATH) return;
  if (k == ORC_K_SPLIT) {                 /* split_and_merge_kernels.py:23-31 */
    double R = c->kern.split_radius, px = 0.0, py = 0.0;
    for (uint32_t a = 0; a < 16; ++a) {   /* uniform in [0,R)^2 until inside the quarter disc */
      uint32_t e[4], ctr[4] = {(uint32_t)step, (uint32_t)(step >> 32), 3u + a / 2u, chain};
      orc_philox(ctr, key, e);
      px = R * u32d(e[2 * (a & 1u)]); py = R * u32d(e[2 * (a & 1u) + 1]);
      if (!(sqrt(px * px + py * py) > R)) break;
    }
    double z[4];
    box_muller(w[3], w[4], &z[0], &z[1]); box_muller(w[5], w[6], &z[2], &z[3]);
    pr->aux0 = px; pr->aux1 = py;
    pr->as = c->kern.split_sigma * (c->kern.vmax[0] - c->kern.vmin[0]) * z[0];
    pr->ar = c->kern.split_sigma * (c->kern.vmax[1] - c->kern.vmin[1]) * z[1];
    pr->aa = c->kern.split_sigma * (c->kern.vmax[2] - c->kern.vmin[2]) * z[2];
    return;
  }
  if (k == ORC_K_MERGE) {                 /* split_and_merge_kernels.py:119-127 */
    int *nb = (int *)malloc(sizeof(int) * (size_t)(n + 1));
    int cnt = merge_neighbours(c, t, nb);
    pr->param_id = cnt > 0 ? nb[mulhi(w[3], (uint32_t)cnt)] : -1;
    if (cnt == 0) pr->target = -1;        /* p0 has no neighbour: an empty perturbation, nothing is removed */
    free(nb);
    return;
  }
  if (k == ORC_K_GTRANS) {                /* transform_kernels.py:24-34 */
    double z0, z1; box_muller(w[3], w[4], &z0, &z1);
    double d0 = c->kern.sigma_trans * z0, d1 = c->kern.sigma_trans * z1;
    int nx = (int)((double)q.x + d0), ny = (int)((double)q.y + d1);
    nx = nx < 0 ? 0 : (nx > c->H - 1 ? c->H - 1 : nx);
    ny = ny < 0 ? 0 : (ny > c->W - 1 ? c->W - 1 : ny);
    q.x = nx; q.y = ny; pr->aux0 = d0; pr->aux1 = d1;
  } else if (k == ORC_K_DTRANS) {         /* transform_kernels.py:77-89 */
    int x0, x1, y0, y1;
    window_bounds(c, q.x, q.y, &x0, &x1, &y0, &y1);
    double tot = 0.0;
    for (int x = x0; x < x1; ++x) for (int y = y0; y < y1; ++y) tot += (double)c->det[(size_t)x * c->W + y];
    double u = u53(w[3], w[4]), acc = 0.0;
    int wc = y1 - y0, cnt = (x1 - x0) * wc, e = 0;
    for (int i = 0; i < cnt; ++i) {
      acc += (double)c->det[(size_t)(x0 + i / wc) * c->W + (y0 + i % wc)];
      if (acc / tot <= u) e = i + 1;
    }
    if (e >= cnt) e = cnt - 1;
    q.x = x0 + e / wc; q.y = y0 + e % wc;
  } else if (k == ORC_K_GTRANSF) {        /* transform_kernels.py:126-144 */
    int pid = (int)mulhi(w[3], 3u);
    double z0, z1; box_muller(w[4], w[5], &z0, &z1);
    double d = c->kern.sigma_transform * (c->kern.vmax[pid] - c->kern.vmin[pid]) * z0;
    set_mark(&q, pid, wrap_mark(c, pid, mark_of(&q, pid) + d));
    pr->param_id = pid; pr->aux0 = d;
  } else {                                /* ORC_K_DTRANSF, transform_kernels.py:179-201 */
    int pid = (int)mulhi(w[3], 3u);
    int cls = sample_class(mark_row(c, pid, q.x, q.y), u32d(w[4]));
    set_mark(&q, pid, c->kern.edges[pid][cls]);
    pr->param_id = pid; pr->new_class = cls;
  }
  pr->ax = q.x; pr->ay = q.y; pr->as = q.s; pr->ar = q.r; pr->aa = q.a;
}

/* one Metropolis-Hastings-Green step with the proposal given (rjmcmc.py:83-164;
 * base_kernels.py:31-122; transform_kernels.py forward/backward_probability) */
static void do_step(orc_ctx *c, const orc_proposal *pr, orc_step_out *out) {
  const orc_kernels *K = &c->kern;
  int k = pr->kernel, n = c->n;
  double pk = K->p_kernel[k], fwd = pk, bwd = pk, dE = 0.0;
  rect_t add[2] = {{pr->ax, pr->ay, pr->as, pr->ar, pr->aa}, {0, 0, 0.0, 0.0, 0.0}};
  int n_add = 0, n_rem = 0;
  int32_t rem[2] = {pr->target, -1};
  if (k == ORC_K_UBIRTH || k == ORC_K_DBIRTH) {
    n_add = 1;
    double dens = k == ORC_K_UBIRTH ? 1.0 : birth_density(c, &add[0]);
    fwd = K->p_kernel[k] * dens / K->intensity;
    bwd = K->p_kernel[k + 1] / (double)(n + 1);
  } else if (k == ORC_K_SPLIT) {          /* split_and_merge_kernels.py:79-106 */
    bwd = K->p_kernel[ORC_K_MERGE];       /* n == 0: forward p_split, backward p_merge */
    if (n > 0 && rem[0] >= 0) {
      n_rem = 1; n_add = 2;
      const double sd[3] = {pr->as, pr->ar, pr->aa};
      split_rects(c, &c->pt[rem[0]], pr, &add[0], &add[1]);
      fwd = pk * ((1.0 / (double)n) * split_pdf(c, sd)) / K->intensity;
      int nn0 = count_potential(c, add[0].x, add[0].y, K->split_radius) + 1;
      int nn1 = count_potential(c, add[1].x, add[1].y, K->split_radius) + 1;
      double nb = (double)(n + 1);
      bwd = K->p_kernel[ORC_K_MERGE] * ((1.0 / nb) * (1.0 / (double)nn0) + (1.0 / nb) * (1.0 / (double)nn1));
    }
  } else if (k == ORC_K_MERGE) {          /* split_and_merge_kernels.py:139-170 */
    bwd = K->p_kernel[ORC_K_SPLIT];       /* fewer than two points, or p0 without neighbour */
    if (n > 1 && rem[0] >= 0 && pr->param_id >= 0) {
      rem[1] = pr->param_id;
      n_rem = 2; n_add = 1;
      const rect_t *p0 = &c->pt[rem[0]], *p1 = &c->pt[rem[1]];
      int *nbuf = (int *)malloc(sizeof(int) * (size_t)(n + 1));
      int n_nb = merge_neighbours(c, rem[0], nbuf);
      free(nbuf);
      merge_rect(c, p0, p1, &add[0]);
      fwd = pk * ((1.0 / (double)n) * (1.0 / (double)n_nb));
      const double sd[3] = {(p0->s - p1->s) / 2.0, (p0->r - p1->r) / 2.0, (p0->a - p1->a) / 2.0};
      bwd = K->p_kernel[ORC_K_SPLIT] * ((1.0 / (double)(n - 1)) * split_pdf(c, sd)) / K->intensity;
    }
  } else if (n > 0 && rem[0] >= 0) {
    n_rem = 1;
    const rect_t *old = &c->pt[rem[0]];
    if (k == ORC_K_UDEATH || k == ORC_K_DDEATH) {
      double dens = k == ORC_K_UDEATH ? 1.0 : birth_density(c, old);
      fwd = K->p_kernel[k] / (double)n;
      bwd = K->p_kernel[k - 1] * dens / K->intensity;
    } else {
      n_add = 1;
      if (k == ORC_K_GTRANS) {
        fwd = bwd = pk * normal_pdf(pr->aux0, K->sigma_trans) * normal_pdf(pr->aux1, K->sigma_trans) / (double)n;
      } else if (k == ORC_K_DTRANS) {
        fwd = pk * move_density(c, old->x, old->y, add[0].x, add[0].y) / (double)n;
        bwd = pk * move_density(c, add[0].x, add[0].y, old->x, old->y) / (double)n;
      } else if (k == ORC_K_GTRANSF) {
        int pid = pr->param_id;
        fwd = bwd = pk * normal_pdf(pr->aux0, K->sigma_transform * (K->vmax[pid] - K->vmin[pid])) / (double)n;
      } else {
        int pid = pr->param_id;
        const float *row = mark_row(c, pid, old->x, old->y);
        double tot = row_sum(row);
        fwd = pk * ((double)row[pr->new_class] / tot) / (double)n;
        bwd = pk * ((double)row[value_to_class(c, pid, mark_of(old, pid))] / tot) / (double)n;
      }
    }
  }
  if (n_add || n_rem) dE = delta_rects(c, n_rem, rem, n_add, add);
  double log_alpha = (-dE / c->T) + log(bwd + EPS_GREEN) - log(fwd + EPS_GREEN);
  int accepted = log(pr->u_accept + EPS_GREEN) < log_alpha;
  if (c->forced >= 0) accepted = c->forced;      /* orc_replay_forced: a decision taken elsewhere (test resync) */
  if (accepted) {
    /* canonical slots: the (first) added point takes the (first) removed point's slot, a second added point is
     * appended, a second removed point is swap-removed */
    if (n_rem && n_add) {
      state_update(c, rem[0], &add[0]);
      if (n_add == 2) state_add(c, &add[1]);
      if (n_rem == 2) state_remove(c, rem[1]);
    } else if (n_rem) state_remove(c, rem[0]);
    else if (n_add) state_add(c, &add[0]);
  }
  if (out) {
    out->dE = dE; out->fwd = fwd; out->bwd = bwd; out->log_alpha = log_alpha; out->T = c->T;
    out->accepted = accepted; out->n_after = c->n;
  }
  c->step++;
  if (c->T > c->T_target) c->T *= c->alpha;  /* rjmcmc.py:158-159 */
}
int orc_replay(orc_ctx *c, int n, const orc_proposal *tape, orc_step_out *out) {
  for (int i = 0; i < n; ++i) {
    if (tape[i].target >= c->n) return -(i + 1);
    if (tape[i].kernel == ORC_K_MERGE && tape[i].param_id >= 0 && (tape[i].param_id >= c->n || tape[i].param_id == tape[i].target)) return -(i + 1);
    do_step(c, &tape[i], out ? &out[i] : NULL);
  }
  return 0;
}
/* replay with the accept decision of every step given (accept[i] != 0): puts the oracle into the state of a chain whose
 * decisions were taken elsewhere -- used by the tests to re-synchronise after a tie within the dE tolerance */
int orc_replay_forced(orc_ctx *c, int n, const orc_proposal *tape, const int32_t *accept, orc_step_out *out) {
  int rc = 0;
  for (int i = 0; i < n && rc == 0; ++i) {
    if (tape[i].target >= c->n) { rc = -(i + 1); break; }
    c->forced = accept[i] ? 1 : 0;
    do_step(c, &tape[i], out ? &out[i] : NULL);
  }
  c->forced = -1;
  return rc;
}
/* Follow a tape while drawing natively: at every step the oracle draws ITS proposal from Philox and its current state
 * (recorded in `native`, never applied) and then performs the step with the tape's proposal and its own Metropolis
 * decision.  With the tape of a kernel run this keeps the two states bit-identical (the device's log / sincos differ
 * from libm's in the last place, so natively drawn Gaussian marks do too), which separates the two parity questions:
 * are the proposals the same (native vs tape), and are dE / the decisions the same given the same proposal. */
int orc_follow(orc_ctx *c, int n, uint64_t seed, uint32_t chain, const orc_proposal *tape, orc_step_out *out,
               orc_proposal *native) {
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  for (int i = 0; i < n; ++i) {
    uint32_t w[8];
    uint64_t s = (uint64_t)c->step;
    for (uint32_t b = 0; b < 2; ++b) {
      uint32_t ctr[4] = {(uint32_t)s, (uint32_t)(s >> 32), b, chain};
      orc_philox(ctr, key, w + 4 * b);
    }
    orc_proposal pr;
    draw_proposal(c, w, &pr, key, s, chain);
    if (native) native[i] = pr;
    if (tape[i].target >= c->n) return -(i + 1);
    do_step(c, &tape[i], out ? &out[i] : NULL);
  }
  return 0;
}
int orc_run(orc_ctx *c, int64_t n_steps, uint64_t seed, uint32_t chain, orc_step_out *out, orc_proposal *props) {
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  for (int64_t i = 0; i < n_steps; ++i) {
    uint32_t w[8];
    uint64_t s = (uint64_t)c->step;
    for (uint32_t b = 0; b < 2; ++b) {
      uint32_t ctr[4] = {(uint32_t)s, (uint32_t)(s >> 32), b, chain};
      orc_philox(ctr, key, w + 4 * b);
    }
    orc_proposal pr;
    draw_proposal(c, w, &pr, key, s, chain);
    if (props) props[i] = pr;
    do_step(c, &pr, out ? &out[i] : NULL);
  }
  return 0;
}

/* sample_rjmcmc.py:23-35 + utils/nms.py:68-110.  Ties broken towards the larger flat index. */
typedef struct { float s; int idx; } cand_t;
static int cand_cmp(const void *a, const void *b) {
  const cand_t *p = (const cand_t *)a, *q = (const cand_t *)b;
  if (p->s != q->s) return p->s < q->s ? -1 : 1;
  return p->idx < q->idx ? -1 : (p->idx > q->idx ? 1 : 0);
}
int orc_naive_detection(orc_ctx *c, double threshold, double nms_dist, int cap, int32_t *xy, double *marks) {
  size_t hw = (size_t)c->H * c->W;
  int nc = 0;
  cand_t *cand = (cand_t *)malloc(sizeof(cand_t) * hw);
  for (size_t i = 0; i < hw; ++i)
    if ((double)c->det[i] >= threshold) { cand[nc].s = c->det[i]; cand[nc].idx = (int)i; ++nc; }
  qsort(cand, nc, sizeof(cand_t), cand_cmp);
  int n_out = 0;
  while (nc > 0) {
    cand_t best = cand[nc - 1];
    int bx = best.idx / c->W, by = best.idx % c->W;
    if (n_out < cap) {
      xy[2 * n_out] = bx; xy[2 * n_out + 1] = by;
      for (int k = 0; k < 3; ++k) {
        const float *row = mark_row(c, k, bx, by);
        int am = 0;
        for (int i = 1; i < ORC_NCLASS; ++i) if (row[i] > row[am]) am = i;
        marks[3 * n_out + k] = c->kern.edges[k][am];
      }
    }
    ++n_out;
    int m = 0;
    for (int i = 0; i < nc - 1; ++i) {
      int x = cand[i].idx / c->W, y = cand[i].idx % c->W;
      double dx = (double)(x - bx), dy = (double)(y - by);
      if (sqrt(dx * dx + dy * dy) > nms_dist) cand[m++] = cand[i];
    }
    nc = m;
  }
  free(cand);
  return n_out;
}
