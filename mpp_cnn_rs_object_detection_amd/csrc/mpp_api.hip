// mpp_api.hip -- host side of the C ABI declared in include/mpp_hip.h.
// Owns device memory, keeps the per-tile pointer table, launches the kernels of
// mpp_sampler.hip / mpp_scratch.hip / mpp_maps.hip on the ctx's HIP stream.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mpp_device.hpp"

extern "C" size_t mpp_chain_lds_bytes(int cap, int ncell, int cell_cap, int spec, int rowbase_n, int waves);
extern "C" size_t mpp_chain_static_lds_bytes(int waves);
extern "C" hipError_t mpp_launch_chain(hipStream_t st, int spec, int lanes, int occ, int grid, size_t lds,
                                       const DevParams *P, const TileRef *tiles, int tile0, const long long *until,
                                       long long trace_base, unsigned long long seed, unsigned int chain0, const mpp_proposal *tape,
                                       int trace_tile, mpp_step_out *out, mpp_proposal *props);
extern "C" size_t mpp_deep_lds_bytes(int cap, int ncell, int cell_cap, int rowbase_n, int waves, int nmax, int ext);
extern "C" size_t mpp_deep_static_lds_bytes(int waves);
extern "C" hipError_t mpp_launch_deep(hipStream_t st, int waves, int occ, int grid, size_t lds, const DevParams *P,
                                      const TileRef *tiles, int tile0, const long long *until, long long trace_base,
                                      unsigned long long seed, unsigned int chain0, int trace_tile, mpp_step_out *out,
                                      mpp_proposal *props, int nmax, int fixed_depth, int gain8, unsigned long long *stats, int ext);
extern "C" void mpp_launch_papangelou_tiles(hipStream_t st, const DevParams *P, const TileRef *tiles, int n_tiles, int max_n, int cap,
                                            double *dE, const int32_t *grid_start, const int32_t *grid_items, int sstride, int istride);
extern "C" void mpp_launch_grid_build_all(hipStream_t st, const DevParams *P, const TileRef *tiles, int n_tiles, int max_n, int ncell,
                                          int cap, int32_t *start, int32_t *cursor, int32_t *items);
extern "C" void mpp_launch_dedupe_tiles(hipStream_t st, const TileRef *tiles, int n_tiles, int max_n, int cap, const double *dE, int dist2,
                                        int32_t *work, int32_t *slot_of, int32_t *tx, int32_t *ty, double *ts, double *tr,
                                        double *ta, int32_t *n_removed);
extern "C" void mpp_launch_remap_table(hipStream_t st, const float *m, size_t n, double coef, double icpt, double *out);
extern "C" void mpp_launch_set_until(hipStream_t st, const TileRef *tiles, int tile0, int n, long long n_steps, long long *until);
extern "C" void mpp_launch_delta_vectors(hipStream_t st, const DevParams *P, const TileRef *tiles, int tile,
                                         int n_cases, const int32_t *rem_off, const int32_t *rem,
                                         const int32_t *add_off, const int32_t *add_xy, const double *add_marks,
                                         int stride, double *before, double *after, unsigned char *mask,
                                         const int32_t *grid_start, const int32_t *grid_items);
extern "C" void mpp_launch_grid_build(hipStream_t st, const DevParams *P, const TileRef *tiles, int tile, int n, int ncell,
                                      int32_t *start, int32_t *cursor, int32_t *items);
extern "C" int mpp_launch_affine_relu(hipStream_t st, void *x, int planes, int C, size_t hw, int elem_bytes, const float *scale,
                                      const float *shift);
extern "C" int mpp_launch_posnet_epilogue_nhwc(hipStream_t st, const void *out, int elem_bytes, int H, int W, int ldw, float w,
                                               float b, float *det);
extern "C" int mpp_launch_shapenet_epilogue_nhwc(hipStream_t st, const void *logits, int elem_bytes, int H, int W, int ldw,
                                                 float *marks);
extern "C" int mpp_launch_nhwc_glue(hipStream_t st, const void *x0, const void *x1, void *y, int H, int W, int C0, int C1, int pad,
                                    int pool, int in_bytes, int out_bytes, const float *scale, const float *shift);
extern "C" int mpp_launch_conv3x3_c32(hipStream_t st, const float *x0, const float *x1, int H, int W, const float *wp,
                                      const float *in_scale, const float *in_shift, const float *out_scale, const float *out_shift,
                                      int relu, float *y);
extern "C" int mpp_launch_conv3x3_stem(hipStream_t st, const float *x, int H, int W, const float *wp, const float *scale,
                                       const float *shift, float *y);
extern "C" int mpp_launch_shapenet_heads(hipStream_t st, const float *h, int H, int W, int ldw, const float *wh, const float *bh,
                                         float *m0, float *m1, float *m2);
extern "C" void mpp_launch_quad_iou(hipStream_t st, int n, const double *a, int m, const double *b, double *out);
extern "C" void mpp_launch_pack_detections(hipStream_t st, const TileRef *tiles, int n_tiles, const int32_t *tile_ids,
                                           const int32_t *anchors, int capacity, double *out);
extern "C" void mpp_launch_point_energies(hipStream_t st, const DevParams *P, const TileRef *tiles, int tile, int n,
                                          double *e_pts, double *vectors, const int32_t *grid_start,
                                          const int32_t *grid_items);
extern "C" void mpp_launch_delta_batch(hipStream_t st, const DevParams *P, const TileRef *tiles, int tile,
                                       int n_cases, const int32_t *rem_off, const int32_t *rem,
                                       const int32_t *add_off, const int32_t *add_xy, const double *add_marks,
                                       double *dE, const int32_t *grid_start, const int32_t *grid_items);
extern "C" void mpp_launch_cdf(hipStream_t st, int n_tiles, const float *det, int H, int W, double *rowpart, double *rowbase,
                               double *scratch_rowtot);
extern "C" void mpp_launch_boxsum(hipStream_t st, int n_tiles, const double *rowpart, int H, int W, int md, double *boxsum);
extern "C" void mpp_launch_naive_init(hipStream_t st, const DevParams *P, const TileRef *tiles, int n_tiles,
                                      double threshold, double nms_dist, unsigned long long *cand, int cand_cap);
extern "C" void mpp_launch_posnet_epilogue(hipStream_t st, const float *out, int H, int W, int ldh, int ldw, float w,
                                           float b, float *det);
extern "C" void mpp_launch_shapenet_epilogue(hipStream_t st, const float *logits, int H, int W, int ldh, int ldw,
                                             float *marks);

#define MPP_LDS_LIMIT (160 * 1024)
#define MPP_CELL_CAP_MAX 2048    // entries of a 32-px cell of the spatial hash (16-bit counts); what fits the LDS decides

struct mpp_ctx {
  int device = 0;
  hipStream_t stream = nullptr, own_stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::string err;
  DevParams hp;
  DevParams *dp = nullptr;
  bool params_dirty = true, tiles_dirty = true, have_model = false, have_kernels = false, have_maps = false;
  int n_tiles = 0, H = 0, W = 0;
  bool maps_borrowed = false;
  float *det = nullptr, *m[3] = {nullptr, nullptr, nullptr};
  float *img = nullptr;              // the picture behind the classic image energies (mpp_set_image), [n_maps][H][W][img_c]
  int img_c = 0;
  bool img_borrowed = false;
  double *rowpart = nullptr, *rowbase = nullptr, *rowtot = nullptr, *boxsum = nullptr;
  bool box_dirty = true;
  bool cdf_ready = false;          // rowpart / rowbase / boxsum exist (made by the first launch that draws births)
  int cap = 1024, cell_cap = 32, spec = 1, lanes = 0;
  // deep rounds (mpp_deep.hip): every lane of the chain's `spec` waves evaluates one step, at most `deep` steps per round
  // (default 128; 0 = off: one wave per step); deep_fixed > 0 pins the number of steps per round (tests); deep_stats: rounds, evaluated
  // steps, rounds with a change, committed steps of the last mpp_run (device counters, read on request)
  int handover = 1;                  // start a chain of 8 waves with one wave per step and hand it to the deep rounds once it has cooled down
  int handover_at = 1280;            // ... when the smoothed steps committed per round of 8 reach this / 256 (5.0: one tile is flat from 4.5 to 6.5, 64 tiles want it early -- 34.8 ms at 5.0, 36.3 at 5.5, 42 at 6.5)
  int handover_tiles = 64;           // ... in launches of at most this many chains (64 tiles of config 4: -6 %; 256 of config 5: +4 %)
  int deep = 128, deep_fixed = 0, deep_gain = 12;   // deep_gain / 8 x the steps the last rounds committed = depth of the next (12: 4 % faster than 16 on the bench tile and on config 5's chains, 10 and 20 slower)
  unsigned long long *deep_stats = nullptr;
  int replicas = 1, n_maps = 0;      // n_tiles = n_maps * replicas chains; chain t samples on the maps of tile t % n_maps
  int32_t *px = nullptr, *py = nullptr, *n = nullptr, *errd = nullptr;
  double *ps = nullptr, *pr = nullptr, *pa = nullptr, *T = nullptr;
  int64_t *step = nullptr;
  long long *until = nullptr;        // per tile: the absolute step the current mpp_run / mpp_replay call runs it to
  double *remap[3] = {nullptr, nullptr, nullptr};   // tables of the remapped marks (chains only), see ensure_remap_tables
  bool remap_dirty = true;
  int remap_mode = -1;               // option "remap_table": -1 auto (when the tables fit remap_budget), 0 never, 1 always
  // 2 GB: a handful of tiles sampled for many steps (BASELINE configs 2 and 3: 0.2 GB per 512-px tile).  With the 256 tiles
  // of a 4096-px image the tables would be 12.9 GB: 3.7 ms less kernel time (5 %) for >= 6 ms of building them and a 13 GB
  // hipMalloc whose cost varies between 0 and 1.4 s (profiles/tools/probe_remap_cost.py) -- not worth it.
  size_t remap_budget = (size_t)2 << 30;
  int auto_grow = 1, grow_events = 0; // capacity overflow -> raise the capacity and continue (see run_chain)
  std::vector<double> intensity;
  std::vector<uint64_t> key_seed;    // per-chain Philox key / chain id (mpp_set_chain_keys); empty: the launch's seed, chain0 + tile
  std::vector<uint32_t> key_chain;
  std::vector<TileRef> h_tiles;
  TileRef *d_tiles = nullptr;
  double sched[3] = {1.0, 1.0, 0.0};
  double last_ms = 0.0;
  // uniform grid over one tile's configuration for the from-scratch energies (built per call; see mpp_scratch.hip)
  int32_t *g_start = nullptr, *g_cursor = nullptr, *g_items = nullptr;
  int g_cells = 0, g_cap = 0, grid_min_points = 256;
};

static int fail(mpp_ctx *c, int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf;
  return code;
}
#define HIPCHK(c, call)                                                                            \
  do {                                                                                             \
    hipError_t e_ = (call);                                                                        \
    if (e_ != hipSuccess) return fail(c, -2, "%s failed: %s", #call, hipGetErrorString(e_));       \
  } while (0)

template <typename T>
static hipError_t dalloc(T **p, size_t count) {
  if (*p) { (void)hipFree(*p); *p = nullptr; }
  if (count == 0) count = 1;
  return hipMalloc((void **)p, count * sizeof(T));
}

// Build the candidate grid of `tile` when its configuration is large enough to pay for four small launches; returns
// the two device arrays through start / items (nullptr, nullptr: the kernels scan the whole configuration).
static int scratch_grid(mpp_ctx *c, int tile, int n, const int32_t **start, const int32_t **items) {
  *start = *items = nullptr;
  if (c->grid_min_points <= 0 || n < c->grid_min_points) return 0;
  const int ncell = c->hp.nx * c->hp.ny;
  if (ncell <= 0) return 0;
  if (ncell > c->g_cells) {
    if (dalloc(&c->g_start, (size_t)ncell + 1) != hipSuccess || dalloc(&c->g_cursor, (size_t)ncell) != hipSuccess) return -2;
    c->g_cells = ncell;
  }
  if (n > c->g_cap) {
    if (dalloc(&c->g_items, (size_t)c->cap) != hipSuccess) return -2;
    c->g_cap = c->cap;
  }
  mpp_launch_grid_build(c->stream, c->dp, c->d_tiles, tile, n, ncell, c->g_start, c->g_cursor, c->g_items);
  *start = c->g_start; *items = c->g_items;
  return 0;
}

static const char *chain_error_text(int e) {
  switch (e) {
    case 1: return "a cell of the spatial hash holds more points than cell_capacity allows and a larger one does not fit the LDS";
    case 2: return "point capacity of the tile exceeded (a larger point_capacity does not fit the chain's LDS budget, or auto_grow is off)";
    case 3: return "proposal refers to a point that does not exist or lies outside the tile";
    case 4: return "candidate list overflow (lower cell_capacity or report)";
  }
  return "unknown chain error";
}

// 2: ten kernels (split, merge), mpp_kernels.split_*; 3: mpp_nhwc_glue, mpp_*_epilogue_nhwc; 4: mpp_pack_detections; 5: mpp_set_chain_keys, options auto_grow / remap_table
extern "C" int mpp_abi_version(void) { return 9; }

extern "C" void mpp_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out);
}

extern "C" int mpp_create(int device_id, mpp_ctx **out) {
  if (!out) return -1;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return -3;   // no GPU: fail loudly, no CPU fallback
  if (device_id < 0 || device_id >= count) return -1;
  mpp_ctx *c = new mpp_ctx();
  c->device = device_id;
  memset(&c->hp, 0, sizeof(DevParams));
  c->hp.n_kernels = MPP_K_SPLIT;
  if (hipSetDevice(device_id) != hipSuccess || hipStreamCreate(&c->own_stream) != hipSuccess ||
      hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
      hipMalloc((void **)&c->dp, sizeof(DevParams)) != hipSuccess) {
    delete c;
    return -2;
  }
  c->stream = c->own_stream;
  *out = c;
  return 0;
}

static void free_tiles(mpp_ctx *c) {
  if (!c->maps_borrowed) {
    if (c->det) (void)hipFree(c->det);
    for (int k = 0; k < 3; ++k) if (c->m[k]) (void)hipFree(c->m[k]);
  }
  c->det = nullptr; c->m[0] = c->m[1] = c->m[2] = nullptr;
  if (c->img && !c->img_borrowed) (void)hipFree(c->img);
  c->img = nullptr; c->img_c = 0; c->img_borrowed = false;
  for (int k = 0; k < 3; ++k) if (c->remap[k]) { (void)hipFree(c->remap[k]); c->remap[k] = nullptr; }
  c->remap_dirty = true;
  if (c->boxsum) { (void)hipFree(c->boxsum); c->boxsum = nullptr; }
  void *ptrs[] = {c->rowpart, c->rowbase, c->rowtot, c->px, c->py, c->n, c->errd, c->ps, c->pr, c->pa, c->T, c->step,
                  c->d_tiles, c->until};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  c->rowpart = c->rowbase = c->rowtot = nullptr; c->px = c->py = c->n = c->errd = nullptr;
  c->ps = c->pr = c->pa = c->T = nullptr; c->step = nullptr; c->d_tiles = nullptr; c->until = nullptr;
}

extern "C" int mpp_destroy(mpp_ctx *c) {
  if (!c) return 0;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  free_tiles(c);
  if (c->deep_stats) (void)hipFree(c->deep_stats);
  if (c->g_start) (void)hipFree(c->g_start);
  if (c->g_cursor) (void)hipFree(c->g_cursor);
  if (c->g_items) (void)hipFree(c->g_items);
  if (c->dp) (void)hipFree(c->dp);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
  return 0;
}

extern "C" const char *mpp_last_error(mpp_ctx *c) { return c ? c->err.c_str() : "no context"; }

extern "C" int mpp_set_stream(mpp_ctx *c, void *s) {
  if (!c) return -1;
  c->stream = s ? (hipStream_t)s : c->own_stream;
  return 0;
}
extern "C" int mpp_synchronize(mpp_ctx *c) {
  if (!c) return -1;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int mpp_set_option(mpp_ctx *c, const char *name, int64_t v) {
  if (!c || !name) return -1;
  if (!strcmp(name, "spec_waves")) {
    if (v != 1 && v != 2 && v != 4 && v != 8 && v != 16) return fail(c, -1, "spec_waves must be 1, 2, 4, 8 or 16");
    c->spec = (int)v;
  } else if (!strcmp(name, "spec_lanes")) {
    // lane mode: 4 waves, `v` lanes of each evaluate one speculative step each (4*v steps per round); 0 = off
    if (v != 0 && v != 1 && v != 2 && v != 4 && v != 8 && v != 16) return fail(c, -1, "spec_lanes must be 0, 1, 2, 4, 8 or 16");
    c->lanes = (int)v;
  } else if (!strcmp(name, "deep")) {
    if (v != 0 && (v < 8 || v > 256 || (v & (v - 1)))) return fail(c, -1, "deep must be 0 or a power of two in 8..256");
    c->deep = (int)v;
  } else if (!strcmp(name, "handover")) {
    if (v != 0 && v != 1) return fail(c, -1, "handover must be 0 or 1");
    c->handover = (int)v;
  } else if (!strcmp(name, "handover_at")) {
    if (v < 256 || v > 2048) return fail(c, -1, "handover_at must be in 256..2048 (steps committed per round of 8, x 256)");
    c->handover_at = (int)v;
  } else if (!strcmp(name, "handover_tiles")) {
    if (v < 1 || v > 65536) return fail(c, -1, "handover_tiles must be in 1..65536");
    c->handover_tiles = (int)v;
  } else if (!strcmp(name, "deep_gain")) {
    if ((v & 0xff) < 8 || (v & 0xff) > 64 || (v & ~0x1ffll)) return fail(c, -1, "deep_gain must be in 8..64 (eighths; + 256: sorted steps dealt in blocks)");
    c->deep_gain = (int)v;
  } else if (!strcmp(name, "deep_fixed")) {
    if (v < 0 || v > 256) return fail(c, -1, "deep_fixed must be in 0..256");
    c->deep_fixed = (int)v;
  } else if (!strcmp(name, "replicas")) {
    // independent replica chains per tile: mpp_set_maps(n_tiles = M) then creates M*v chains, chain t on the maps
    // of tile t % M (several chains of one tile with different chain ids, or a benchmark's many-tile load)
    if (c->have_maps) return fail(c, -1, "replicas must be set before mpp_set_maps");
    if (v < 1 || v > 65536) return fail(c, -1, "replicas out of range");
    c->replicas = (int)v;
  } else if (!strcmp(name, "point_capacity")) {
    if (c->have_maps) return fail(c, -1, "point_capacity must be set before mpp_set_maps");
    if (v < 1 || v > 65535) return fail(c, -1, "point_capacity out of range");
    c->cap = (int)v;
  } else if (!strcmp(name, "scratch_grid_min_points")) {
    // configurations of at least this many points get a candidate grid for the from-scratch energies; 0 = never
    if (v < 0) return fail(c, -1, "scratch_grid_min_points must be >= 0");
    c->grid_min_points = (int)v;
  } else if (!strcmp(name, "cell_capacity")) {
    if (v < 1 || v > MPP_CELL_CAP_MAX) return fail(c, -1, "cell_capacity must be in 1..%d", MPP_CELL_CAP_MAX);
    c->cell_cap = (int)v; c->params_dirty = true;
  } else if (!strcmp(name, "auto_grow")) {
    c->auto_grow = v ? 1 : 0;
  } else if (!strcmp(name, "remap_table")) {
    if (v < -1 || v > 1) return fail(c, -1, "remap_table must be -1 (auto), 0 or 1");
    c->remap_mode = (int)v; c->remap_dirty = true;
  } else if (!strcmp(name, "force_accept")) {
    c->hp.force_accept = v ? 1 : 0; c->params_dirty = true;
  } else return fail(c, -1, "unknown option %s", name);
  return 0;
}
static bool has_classic(const mpp_model &M, int *want_gradient);
extern "C" int64_t mpp_get_option(mpp_ctx *c, const char *name) {
  if (!c || !name) return -1;
  if (!strcmp(name, "spec_waves")) return c->spec;
  if (!strcmp(name, "spec_lanes")) return c->lanes;
  if (!strcmp(name, "deep")) return c->deep;
  if (!strcmp(name, "deep_fixed")) return c->deep_fixed;
  if (!strncmp(name, "deep_stat", 9) && name[9] >= '0' && name[9] <= '9') {      // deep_stat0 .. deep_stat255 (4 counters, then the per-wave phase clocks of the diagnostic build)
    const int i = atoi(name + 9);
    unsigned long long v[256] = {0};
    if (i > 255) return -1;
    if (c->deep_stats && hipMemcpy(v, c->deep_stats, sizeof v, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int64_t)v[i];
  }
  if (!strcmp(name, "point_capacity")) return c->cap;
  if (!strcmp(name, "replicas")) return c->replicas;
  if (!strcmp(name, "n_chains")) return c->n_tiles;
  if (!strcmp(name, "cell_capacity")) return c->cell_cap;
  if (!strcmp(name, "handover")) return c->handover;
  if (!strcmp(name, "handover_tiles")) return c->handover_tiles;
  if (!strcmp(name, "handover_at")) return c->handover_at;
  if (!strcmp(name, "auto_grow")) return c->auto_grow;
  if (!strcmp(name, "remap_table")) return c->remap[0] ? 1 : 0;       // are the tables in use right now?
  if (!strcmp(name, "grow_events")) return c->grow_events;
  if (!strcmp(name, "scratch_grid_min_points")) return c->grid_min_points;
  if (!strcmp(name, "force_accept")) return c->hp.force_accept;
  if (!strcmp(name, "grid_nx")) return c->hp.nx;       // spatial hash dimensions (point_set.py:58-61)
  if (!strcmp(name, "grid_ny")) return c->hp.ny;
  if (!strcmp(name, "grid_res")) return (long long)c->hp.res;
  if (!strcmp(name, "lds_bytes")) {
    int ncell = c->hp.nx * c->hp.ny;
    int spec = c->lanes > 0 ? 4 * c->lanes : c->spec;
    int rb = (c->lanes > 0 && c->H <= 1024) ? c->H + 1 : 0;
    size_t b = mpp_chain_lds_bytes(c->cap, ncell > 0 ? ncell : 1, c->cell_cap, spec, rb, c->lanes > 0 ? 4 : c->spec) +
               mpp_chain_static_lds_bytes(c->lanes > 0 ? 4 : c->spec);
    if (c->deep > 0 && c->lanes == 0 && c->spec <= 8) {        // deep rounds: at least the smallest round has to fit
      const int rbd = c->H <= 1024 ? c->H + 1 : 0;
      const size_t d = mpp_deep_lds_bytes(c->cap, ncell > 0 ? ncell : 1, c->cell_cap, rbd, c->spec, c->spec > 8 ? c->spec : 8, has_classic(c->hp.model, nullptr) ? 1 : 0) +
                       mpp_deep_static_lds_bytes(c->spec);
      if (d > b) b = d;
    }
    return (int64_t)b;
  }
  return -1;
}

static void refresh_grid(mpp_ctx *c) {
  DevParams &P = c->hp;
  double maxd = 0.0;
  for (int p = 0; p < P.model.n_pair; ++p) if (P.model.pair[p].max_dist > maxd) maxd = P.model.pair[p].max_dist;
  P.max_inter = P.model.n_pair > 0 ? maxd : 1.0;                  // energy_graph.py:26-29
  P.res = maxd > 32.0 ? maxd : 32.0;                              // point_set.py:7,58
  P.H = c->H; P.W = c->W;
  P.nx = c->H > 0 ? (int)ceil((double)c->H / P.res) : 0;         // point_set.py:59-61
  P.ny = c->W > 0 ? (int)ceil((double)c->W / P.res) : 0;
  P.cap = c->cap; P.cell_cap = c->cell_cap; P.n_tiles = c->n_tiles;
  P.res_int = (int)P.res;
  P.res_shift = -1;
  for (int s = 0; s < 16; ++s) if ((1 << s) == P.res_int) P.res_shift = s;
  for (int p = 0; p < MPP_MAX_PAIR; ++p)
    P.maxd2[p] = p < P.model.n_pair ? (int)floor(P.model.pair[p].max_dist * P.model.pair[p].max_dist + 1e-9) : 0;
  P.conflict_d2 = (int)(4.0 * P.max_inter * P.max_inter) + 1;
  P.uniform_bins = 1;
  for (int k = 0; k < 3; ++k) {
    double range = P.maps.vmax[k] - P.maps.vmin[k];
    P.inv_step[k] = range > 0 ? (double)MPP_NCLASS / range : 0.0;
    for (int i = 0; i < MPP_NCLASS; ++i)
      if (!(fabs(P.maps.edges[k][i] - (P.maps.vmin[k] + range * i / MPP_NCLASS)) <= 1e-9 * (fabs(range) + 1.0)))
        P.uniform_bins = 0;
  }
  c->params_dirty = true;
}

extern "C" int mpp_set_model(mpp_ctx *c, const mpp_model *model, const mpp_mappings *maps) {
  if (!c || !model || !maps) return -1;
  if (model->n_unit < 0 || model->n_unit > MPP_MAX_UNIT || model->n_pair < 0 || model->n_pair > MPP_MAX_PAIR)
    return fail(c, -1, "bad term counts");
  if (model->gate_term >= model->n_unit) return fail(c, -1, "gate_term must index a unit term");
  for (int p = 0; p < model->n_pair; ++p) {
    const mpp_pair_term &t = model->pair[p];
    bool ok = (t.kind == MPP_P_OVERLAP && t.reduce == MPP_REDUCE_MAX) ||
              (t.kind == MPP_P_ALIGN && ((t.p[0] != 0.0) == (t.reduce == MPP_REDUCE_MIN))) ||
              ((t.kind == MPP_P_DIST_LE || t.kind == MPP_P_DIST_LT) && t.reduce == MPP_REDUCE_MAX);
    if (!ok) return fail(c, -1, "pair term %d: unsupported (kind, reduce) combination", p);
    if (!(t.max_dist > 0.0)) return fail(c, -1, "pair term %d: max_dist must be positive", p);
  }
  {
    double maxd = 0.0;
    for (int p = 0; p < model->n_pair; ++p) if (model->pair[p].max_dist > maxd) maxd = model->pair[p].max_dist;
    if (maxd > 32.0 && maxd != floor(maxd))
      return fail(c, -1, "interaction radius %g > 32 must be integral (it is the cell size of the spatial hash)", maxd);
  }
  for (int k = 0; k < 3; ++k)
    for (int i = 1; i < MPP_NCLASS; ++i)
      if (!(maps->edges[k][i] > maps->edges[k][i - 1])) return fail(c, -1, "mark %d: bin edges must increase", k);
  for (int k = 0; k < model->n_unit; ++k) {
    const mpp_unit_term &t = model->unit[k];
    if (t.kind != MPP_U_CONTRAST && t.kind != MPP_U_GRADIENT) continue;
    // the rasteriser works on a 96 x 128 pixel window: the largest rectangle (size = vmax, ratio -> 0) must fit with its margins
    if (!(maps->vmax[0] <= 36.0)) return fail(c, -1, "unit term %d: the classic image energies need size marks <= 36 px", k);
    if (t.kind == MPP_U_CONTRAST) {
      const int measure = (int)t.p[0], dil = (int)t.p[1], gap = (int)t.p[2], ero = (int)t.p[3];
      if (measure < 0 || measure > 5 || dil < 1 || dil > 4 || gap < 0 || gap > 2 || ero < 0 || ero > 2)
        return fail(c, -1, "unit term %d: contrast energy wants measure 0..5, dilation 1..4, gap 0..2, erode 0..2", k);
    }
  }
  c->hp.model = *model;
  c->hp.maps = *maps;
  c->have_model = true;
  c->remap_dirty = true;
  refresh_grid(c);
  return 0;
}

extern "C" int mpp_set_kernels(mpp_ctx *c, const mpp_kernels *k, const double *intensity) {
  if (!c || !k) return -1;
  double acc = 0.0;
  for (int i = 0; i < MPP_NKERNEL; ++i) {
    if (k->p_kernel[i] < 0) return fail(c, -1, "negative kernel probability");
    acc += k->p_kernel[i]; c->hp.p_cum[i] = acc;
  }
  if (fabs(acc - 1.0) > 1e-8) return fail(c, -1, "kernel probabilities do not sum to 1");   // make_kernels.py:164-172
  if (k->max_delta < 0 || k->max_delta > 15) return fail(c, -1, "max_delta must be in 0..15");
  // 8 kernels unless split / merge carry probability (the cumulative table is searched up to the last active one)
  c->hp.n_kernels = (k->p_kernel[MPP_K_SPLIT] > 0.0 || k->p_kernel[MPP_K_MERGE] > 0.0) ? MPP_NKERNEL : MPP_K_SPLIT;
  if (c->hp.n_kernels == MPP_NKERNEL && !(k->split_radius > 0.0 && k->split_sigma > 0.0))
    return fail(c, -1, "split/merge kernels need split_radius > 0 and split_sigma > 0");
  if (!c->have_kernels || c->hp.kern.max_delta != k->max_delta) c->box_dirty = true;
  c->hp.kern = *k;
  c->have_kernels = true;
  c->params_dirty = true;
  if (intensity) {
    c->intensity.assign(intensity, intensity + (c->n_tiles > 0 ? c->n_tiles : 1));
    c->tiles_dirty = true;
  }
  return 0;
}

extern "C" int mpp_set_maps(mpp_ctx *c, int n_tiles, int H, int W, const float *det, const float *m0, const float *m1,
                            const float *m2, int on_device) {
  if (!c) return -1;
  if (n_tiles <= 0 || H <= 0 || W <= 0 || H > 65535 || W > 65535) return fail(c, -1, "bad tile geometry");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  free_tiles(c);
  c->n_maps = n_tiles; c->n_tiles = n_tiles * c->replicas; c->H = H; c->W = W;
  c->key_seed.clear(); c->key_chain.clear();
  const size_t hw = (size_t)H * W, M = (size_t)n_tiles, T = (size_t)c->n_tiles;
  const float *src[4] = {det, m0, m1, m2};
  float **dst[4] = {&c->det, &c->m[0], &c->m[1], &c->m[2]};
  c->maps_borrowed = on_device != 0;
  for (int k = 0; k < 4; ++k) {
    size_t cnt = M * hw * (k == 0 ? 1 : MPP_NCLASS);
    if (on_device) {
      if (!src[k]) return fail(c, -1, "borrowed device maps must all be given");
      *dst[k] = const_cast<float *>(src[k]);
    } else {
      HIPCHK(c, dalloc(dst[k], cnt));
      if (src[k]) HIPCHK(c, hipMemcpyAsync(*dst[k], src[k], cnt * sizeof(float), hipMemcpyHostToDevice, c->stream));
      else HIPCHK(c, hipMemsetAsync(*dst[k], 0, cnt * sizeof(float), c->stream));
    }
  }
  // (the cumulative tables of the birth kernels -- 16 B per pixel -- are made by the first chain launch: a context that only
  //  scores or merges, e.g. one whole 4096 x 4096 image, never needs them)
  c->cdf_ready = false;
  c->box_dirty = true;
  HIPCHK(c, dalloc(&c->px, T * c->cap)); HIPCHK(c, dalloc(&c->py, T * c->cap));
  HIPCHK(c, dalloc(&c->ps, T * c->cap)); HIPCHK(c, dalloc(&c->pr, T * c->cap)); HIPCHK(c, dalloc(&c->pa, T * c->cap));
  HIPCHK(c, dalloc(&c->n, T)); HIPCHK(c, dalloc(&c->errd, T)); HIPCHK(c, dalloc(&c->T, T * 3));
  HIPCHK(c, dalloc(&c->step, T)); HIPCHK(c, dalloc(&c->d_tiles, T)); HIPCHK(c, dalloc(&c->until, T));
  HIPCHK(c, hipMemsetAsync(c->n, 0, T * sizeof(int32_t), c->stream));
  HIPCHK(c, hipMemsetAsync(c->errd, 0, T * sizeof(int32_t), c->stream));
  HIPCHK(c, hipMemsetAsync(c->step, 0, T * sizeof(int64_t), c->stream));
  std::vector<double> sched(T * 3);
  for (size_t t = 0; t < T; ++t) { sched[3 * t] = c->sched[0]; sched[3 * t + 1] = c->sched[1]; sched[3 * t + 2] = c->sched[2]; }
  HIPCHK(c, hipMemcpyAsync(c->T, sched.data(), sched.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if ((int)c->intensity.size() != c->n_tiles) c->intensity.assign(c->n_tiles, 1.0);
  c->have_maps = true;
  c->tiles_dirty = true;
  refresh_grid(c);
  return 0;
}

static bool has_classic(const mpp_model &M, int *want_gradient = nullptr) {
  bool any = false;
  for (int k = 0; k < M.n_unit; ++k) {
    if (M.unit[k].kind == MPP_U_CONTRAST) any = true;
    if (M.unit[k].kind == MPP_U_GRADIENT) { any = true; if (want_gradient) *want_gradient = 1; }
  }
  return any;
}

extern "C" int mpp_set_image(mpp_ctx *c, int n_tiles, int C, const float *img, int on_device) {
  if (!c) return -1;
  if (!c->have_maps) return fail(c, -1, "mpp_set_maps has not been called");
  if (n_tiles != c->n_maps) return fail(c, -1, "mpp_set_image: %d tiles given, the ctx holds %d", n_tiles, c->n_maps);
  if (!img || (C != 1 && C != 2 && C != 3 && C != 6)) return fail(c, -1, "mpp_set_image: C must be 1, 2, 3 or 6");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->img && !c->img_borrowed) (void)hipFree(c->img);
  c->img = nullptr;
  const size_t cnt = (size_t)c->n_maps * c->H * c->W * C;
  if (on_device) c->img = const_cast<float *>(img);
  else {
    HIPCHK(c, dalloc(&c->img, cnt));
    HIPCHK(c, hipMemcpyAsync(c->img, img, cnt * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  c->img_borrowed = on_device != 0;
  c->img_c = C;
  c->tiles_dirty = true;
  return 0;
}

static int push_state(mpp_ctx *c) {
  if (!c->have_maps) return fail(c, -1, "mpp_set_maps has not been called");
  if (!c->have_model) return fail(c, -1, "mpp_set_model has not been called");
  {
    int grad = 0;
    if (has_classic(c->hp.model, &grad)) {
      if (!c->img) return fail(c, -1, "the model has a classic image energy: mpp_set_image has not been called");
      if (grad ? (c->img_c != 2 && c->img_c != 6) : (c->img_c != 1 && c->img_c != 3))
        return fail(c, -1, "the image has %d channels: the %s energy wants %s", c->img_c, grad ? "gradient" : "contrast",
                    grad ? "2 or 6 (np.gradient)" : "1 or 3");
    }
  }
  HIPCHK(c, hipSetDevice(c->device));
  if (c->tiles_dirty) {
    const size_t hw = (size_t)c->H * c->W;
    c->h_tiles.resize(c->n_tiles);
    for (int t = 0; t < c->n_tiles; ++t) {
      TileRef &r = c->h_tiles[t];
      const size_t m = (size_t)(t % c->n_maps);                    // replica chains share their tile's maps
      r.det = (const MPP_GLOBAL float *)(c->det + m * hw);
      for (int k = 0; k < 3; ++k) r.m[k] = (const MPP_GLOBAL float *)(c->m[k] + m * hw * MPP_NCLASS);
      r.rowpart = c->cdf_ready ? (const MPP_GLOBAL double *)(c->rowpart + m * hw) : nullptr;
      r.rowbase = c->cdf_ready ? (const MPP_GLOBAL double *)(c->rowbase + m * (c->H + 1)) : nullptr;
      r.boxsum = c->cdf_ready ? (const MPP_GLOBAL double *)(c->boxsum + m * hw) : nullptr;
      for (int k = 0; k < 3; ++k)
        r.rm[k] = c->remap[k] ? (const MPP_GLOBAL double *)(c->remap[k] + m * hw * MPP_NCLASS) : nullptr;
      r.img = c->img ? (const MPP_GLOBAL float *)(c->img + m * hw * (size_t)c->img_c) : nullptr;
      r.img_c = c->img_c; r._pad_img = 0;
      r.px = c->px + (size_t)t * c->cap; r.py = c->py + (size_t)t * c->cap;
      r.ps = c->ps + (size_t)t * c->cap; r.pr = c->pr + (size_t)t * c->cap; r.pa = c->pa + (size_t)t * c->cap;
      r.n = c->n + t; r.T = c->T + 3 * (size_t)t; r.step = c->step + t; r.err = c->errd + t;
      r.intensity = c->intensity[t];
      const bool own = (int)c->key_seed.size() == c->n_tiles;
      r.key_on = own ? 1u : 0u; r.key_seed = own ? c->key_seed[t] : 0ull; r.key_chain = own ? c->key_chain[t] : 0u;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_tiles, c->h_tiles.data(), sizeof(TileRef) * c->n_tiles, hipMemcpyHostToDevice,
                             c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->tiles_dirty = false;
  }
  if (c->box_dirty && c->have_kernels && c->cdf_ready) {
    const size_t hw = (size_t)c->H * c->W;
    (void)hw;
    mpp_launch_boxsum(c->stream, c->n_maps, c->rowpart, c->H, c->W, c->hp.kern.max_delta, c->boxsum);
    HIPCHK(c, hipGetLastError());
    c->box_dirty = false;
  }
  if (c->params_dirty) {
    c->hp.cap = c->cap; c->hp.cell_cap = c->cell_cap; c->hp.n_tiles = c->n_tiles;
    HIPCHK(c, hipMemcpyAsync(c->dp, &c->hp, sizeof(DevParams), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->params_dirty = false;
  }
  return 0;
}
static int check_tile(mpp_ctx *c, int tile) {
  if (!c) return -1;
  if (!c->have_maps) return fail(c, -1, "mpp_set_maps has not been called");
  if (tile < 0 || tile >= c->n_tiles) return fail(c, -1, "tile %d out of range", tile);
  return 0;
}

extern "C" int mpp_set_chain_keys(mpp_ctx *c, int n, const uint64_t *seeds, const uint32_t *chains) {
  if (!c) return -1;
  if (!seeds || !chains) {                     // back to the launch's seed and chain0 + tile
    c->key_seed.clear(); c->key_chain.clear(); c->tiles_dirty = true;
    return 0;
  }
  if (!c->have_maps || n != c->n_tiles) return fail(c, -1, "mpp_set_chain_keys: one key per chain of the context (%d)", c->n_tiles);
  c->key_seed.assign(seeds, seeds + n); c->key_chain.assign(chains, chains + n);
  c->tiles_dirty = true;
  return 0;
}

extern "C" int mpp_set_points(mpp_ctx *c, int tile, int n, const int32_t *xy, const double *marks) {
  int rc = check_tile(c, tile);
  if (rc) return rc;
  if (n < 0 || n > c->cap) return fail(c, -4, "%d points exceed point_capacity %d", n, c->cap);
  std::vector<int32_t> x(n), y(n);
  std::vector<double> s(n), r(n), a(n);
  for (int i = 0; i < n; ++i) {
    x[i] = xy[2 * i]; y[i] = xy[2 * i + 1];
    if (x[i] < 0 || x[i] >= c->H || y[i] < 0 || y[i] >= c->W)
      return fail(c, -5, "point %d (%d,%d) is outside the %dx%d tile", i, x[i], y[i], c->H, c->W);  // point_set.py:99
    s[i] = marks[3 * i]; r[i] = marks[3 * i + 1]; a[i] = marks[3 * i + 2];
  }
  HIPCHK(c, hipSetDevice(c->device));
  size_t o = (size_t)tile * c->cap;
  HIPCHK(c, hipMemcpyAsync(c->px + o, x.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->py + o, y.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->ps + o, s.data(), n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->pr + o, r.data(), n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->pa + o, a.data(), n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  int32_t nn = n, zero = 0;
  HIPCHK(c, hipMemcpyAsync(c->n + tile, &nn, sizeof nn, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->errd + tile, &zero, sizeof zero, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int mpp_count(mpp_ctx *c, int tile, int32_t *n) {
  int rc = check_tile(c, tile);
  if (rc) return rc;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(n, c->n + tile, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int mpp_get_points(mpp_ctx *c, int tile, int cap, int32_t *n_out, int32_t *xy, double *marks) {
  int32_t n = 0;
  int rc = mpp_count(c, tile, &n);
  if (rc) return rc;
  if (n_out) *n_out = n;
  int m = n < cap ? n : cap;
  if (m <= 0 || !xy || !marks) return 0;
  std::vector<int32_t> x(m), y(m);
  std::vector<double> s(m), r(m), a(m);
  size_t o = (size_t)tile * c->cap;
  HIPCHK(c, hipMemcpyAsync(x.data(), c->px + o, m * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(y.data(), c->py + o, m * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(s.data(), c->ps + o, m * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(r.data(), c->pr + o, m * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(a.data(), c->pa + o, m * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int i = 0; i < m; ++i) {
    xy[2 * i] = x[i]; xy[2 * i + 1] = y[i];
    marks[3 * i] = s[i]; marks[3 * i + 1] = r[i]; marks[3 * i + 2] = a[i];
  }
  return 0;
}

// every tile's configuration with five strided copies instead of six small ones per tile (256 tiles: 32 ms -> <1 ms)
extern "C" int mpp_get_points_all(mpp_ctx *c, int cap, int32_t *n_out, int32_t *xy, double *marks) {
  if (!c || c->n_tiles <= 0 || !n_out || cap < 0) return fail(c, -1, "bad get_points_all arguments");
  const int T = c->n_tiles;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(n_out, c->n, sizeof(int32_t) * T, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (!xy || !marks || cap == 0) return 0;
  int m = 0;
  for (int t = 0; t < T; ++t) m = n_out[t] > m ? n_out[t] : m;
  m = m < cap ? m : cap;
  m = m < c->cap ? m : c->cap;
  if (m <= 0) return 0;
  std::vector<int32_t> x((size_t)T * m), y((size_t)T * m);
  std::vector<double> s((size_t)T * m), r((size_t)T * m), a((size_t)T * m);
  const size_t wi = (size_t)m * sizeof(int32_t), wd = (size_t)m * sizeof(double);
  const size_t pi = (size_t)c->cap * sizeof(int32_t), pd = (size_t)c->cap * sizeof(double);
  HIPCHK(c, hipMemcpy2DAsync(x.data(), wi, c->px, pi, wi, T, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpy2DAsync(y.data(), wi, c->py, pi, wi, T, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpy2DAsync(s.data(), wd, c->ps, pd, wd, T, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpy2DAsync(r.data(), wd, c->pr, pd, wd, T, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpy2DAsync(a.data(), wd, c->pa, pd, wd, T, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int t = 0; t < T; ++t) {
    const int k = n_out[t] < m ? n_out[t] : m;
    for (int i = 0; i < k; ++i) {
      const size_t src = (size_t)t * m + i, dst = (size_t)t * cap + i;
      xy[2 * dst] = x[src]; xy[2 * dst + 1] = y[src];
      marks[3 * dst] = s[src]; marks[3 * dst + 1] = r[src]; marks[3 * dst + 2] = a[src];
    }
  }
  return 0;
}

extern "C" int mpp_pack_detections(mpp_ctx *c, int n, const int32_t *tile_ids, const int32_t *anchors, int capacity,
                                   double *out_dev, int32_t *count) {
  if (!c || !c->have_maps || n <= 0 || n > c->n_tiles || !tile_ids || !anchors || capacity < 0 || !out_dev)
    return fail(c, -1, "bad pack_detections arguments");
  int rc = push_state(c);
  if (rc) return rc;
  int32_t *d_meta = nullptr;
  HIPCHK(c, dalloc(&d_meta, (size_t)3 * n));
  hipError_t e = hipMemcpyAsync(d_meta, tile_ids, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_meta + n, anchors, (size_t)2 * n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemsetAsync(out_dev, 0, ((size_t)capacity + 1) * 7 * sizeof(double), c->stream);
  double total = 0.0;
  if (e == hipSuccess) {
    mpp_launch_pack_detections(c->stream, c->d_tiles, n, d_meta, d_meta + n, capacity, out_dev);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(&total, out_dev, sizeof(double), hipMemcpyDeviceToHost, c->stream);
  hipError_t e2 = hipStreamSynchronize(c->stream);
  (void)hipFree(d_meta);
  HIPCHK(c, e); HIPCHK(c, e2);
  if (count) *count = (int32_t)total;
  if ((int)total > capacity)
    return fail(c, -4, "%d detections exceed the gather buffer's capacity %d", (int)total, capacity);
  return 0;
}

extern "C" int mpp_total_energy(mpp_ctx *c, int tile, double *energy, double *vectors) {
  int rc = check_tile(c, tile);
  if (rc) return rc;
  if ((rc = push_state(c))) return rc;
  int32_t n = 0;
  if ((rc = mpp_count(c, tile, &n))) return rc;
  double e = 0.0;
  if (n > 0) {
    int nt = c->hp.model.n_unit + c->hp.model.n_pair;
    double *d_e = nullptr, *d_v = nullptr;
    HIPCHK(c, dalloc(&d_e, (size_t)n));
    if (vectors) HIPCHK(c, dalloc(&d_v, (size_t)n * nt));
    const int32_t *gs, *gi;
    if (scratch_grid(c, tile, n, &gs, &gi)) return fail(c, -2, "no device memory for the candidate grid");
    mpp_launch_point_energies(c->stream, c->dp, c->d_tiles, tile, n, d_e, d_v, gs, gi);
    std::vector<double> he(n);
    hipError_t e1 = hipMemcpyAsync(he.data(), d_e, n * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    hipError_t e2 = hipSuccess;
    if (vectors) e2 = hipMemcpyAsync(vectors, d_v, (size_t)n * nt * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    hipError_t e3 = hipStreamSynchronize(c->stream);
    (void)hipFree(d_e);
    if (d_v) (void)hipFree(d_v);
    HIPCHK(c, e1); HIPCHK(c, e2); HIPCHK(c, e3);
    for (int i = 0; i < n; ++i) e += he[i];                      // same order as the reference's np.sum over points
  }
  if (energy) *energy = e;
  return 0;
}

// shared body of mpp_delta_batch (dE != NULL) and mpp_delta_vectors (before/after/mask != NULL)
static int delta_cases(mpp_ctx *c, int tile, int n_cases, const int32_t *rem_off, const int32_t *rem,
                       const int32_t *add_off, const int32_t *add_xy, const double *add_marks, double *dE, int stride,
                       double *before, double *after, unsigned char *mask) {
  int rc = check_tile(c, tile);
  if (rc) return rc;
  if ((rc = push_state(c))) return rc;
  if (n_cases <= 0) return 0;
  int32_t n = 0;
  if ((rc = mpp_count(c, tile, &n))) return rc;
  const int n_rem = rem_off[n_cases], n_add = add_off[n_cases];
  for (int i = 0; i < n_rem; ++i)
    if (rem[i] < 0 || rem[i] >= n) return fail(c, -6, "removal of slot %d: no such point (n=%d)", rem[i], n);  // KeyError
  for (int i = 0; i < n_add; ++i)
    if (add_xy[2 * i] < 0 || add_xy[2 * i] >= c->H || add_xy[2 * i + 1] < 0 || add_xy[2 * i + 1] >= c->W)
      return fail(c, -5, "added point %d is outside the tile", i);
  const int nt = c->hp.model.n_unit + c->hp.model.n_pair;
  if (!dE) {
    for (int i = 0; i < n_cases; ++i)
      if (n + (add_off[i + 1] - add_off[i]) > stride)
        return fail(c, -1, "delta_vectors: stride %d < n + additions of case %d (%d)", stride, i, n + add_off[i + 1] - add_off[i]);
  }
  const int32_t *gs, *gi;
  if (scratch_grid(c, tile, n, &gs, &gi)) return fail(c, -2, "no device memory for the candidate grid");
  int32_t *d_ro = nullptr, *d_r = nullptr, *d_ao = nullptr, *d_axy = nullptr;
  double *d_am = nullptr, *d_out = nullptr, *d_b = nullptr, *d_a = nullptr;
  unsigned char *d_m = nullptr;
  const size_t rows = dE ? 0 : (size_t)n_cases * stride;
  HIPCHK(c, dalloc(&d_ro, (size_t)n_cases + 1)); HIPCHK(c, dalloc(&d_ao, (size_t)n_cases + 1));
  HIPCHK(c, dalloc(&d_r, (size_t)n_rem)); HIPCHK(c, dalloc(&d_axy, (size_t)2 * n_add));
  HIPCHK(c, dalloc(&d_am, (size_t)3 * n_add));
  if (dE) HIPCHK(c, dalloc(&d_out, (size_t)n_cases));
  else { HIPCHK(c, dalloc(&d_b, rows * nt)); HIPCHK(c, dalloc(&d_a, rows * nt)); HIPCHK(c, dalloc(&d_m, rows)); }
  hipError_t e = hipSuccess;
  auto up = [&](void *dst, const void *src, size_t bytes) {
    if (bytes && e == hipSuccess) e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream);
  };
  up(d_ro, rem_off, (n_cases + 1) * sizeof(int32_t)); up(d_ao, add_off, (n_cases + 1) * sizeof(int32_t));
  up(d_r, rem, n_rem * sizeof(int32_t)); up(d_axy, add_xy, 2 * (size_t)n_add * sizeof(int32_t));
  up(d_am, add_marks, 3 * (size_t)n_add * sizeof(double));
  if (e == hipSuccess) {
    if (dE) {
      mpp_launch_delta_batch(c->stream, c->dp, c->d_tiles, tile, n_cases, d_ro, d_r, d_ao, d_axy, d_am, d_out, gs, gi);
      e = hipMemcpyAsync(dE, d_out, n_cases * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    } else {
      mpp_launch_delta_vectors(c->stream, c->dp, c->d_tiles, tile, n_cases, d_ro, d_r, d_ao, d_axy, d_am, stride, d_b, d_a, d_m, gs, gi);
      e = hipMemcpyAsync(before, d_b, rows * nt * sizeof(double), hipMemcpyDeviceToHost, c->stream);
      if (e == hipSuccess) e = hipMemcpyAsync(after, d_a, rows * nt * sizeof(double), hipMemcpyDeviceToHost, c->stream);
      if (e == hipSuccess) e = hipMemcpyAsync(mask, d_m, rows, hipMemcpyDeviceToHost, c->stream);
    }
  }
  hipError_t e2 = hipStreamSynchronize(c->stream);
  void *ptrs[] = {d_ro, d_r, d_ao, d_axy, d_am, d_out, d_b, d_a, d_m};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  HIPCHK(c, e); HIPCHK(c, e2);
  return 0;
}

extern "C" int mpp_delta_batch(mpp_ctx *c, int tile, int n_cases, const int32_t *rem_off, const int32_t *rem,
                               const int32_t *add_off, const int32_t *add_xy, const double *add_marks, double *dE) {
  if (!dE && n_cases > 0) return fail(c, -1, "delta_batch: dE is NULL");
  return delta_cases(c, tile, n_cases, rem_off, rem, add_off, add_xy, add_marks, dE, 0, nullptr, nullptr, nullptr);
}
extern "C" int mpp_delta_vectors(mpp_ctx *c, int tile, int n_cases, const int32_t *rem_off, const int32_t *rem,
                                 const int32_t *add_off, const int32_t *add_xy, const double *add_marks, int stride,
                                 double *before, double *after, unsigned char *mask) {
  if (n_cases > 0 && (!before || !after || !mask || stride <= 0)) return fail(c, -1, "bad delta_vectors arguments");
  return delta_cases(c, tile, n_cases, rem_off, rem, add_off, add_xy, add_marks, nullptr, stride, before, after, mask);
}

extern "C" int mpp_papangelou(mpp_ctx *c, int tile, double *dE) {
  int32_t n = 0;
  int rc = mpp_count(c, tile, &n);
  if (rc) return rc;
  if (n == 0) return 0;
  std::vector<int32_t> ro(n + 1), r(n), ao(n + 1, 0);
  for (int i = 0; i <= n; ++i) ro[i] = i;
  for (int i = 0; i < n; ++i) r[i] = i;
  int32_t dummy_xy[2] = {0, 0};
  double dummy_m[3] = {0, 0, 0};
  rc = mpp_delta_batch(c, tile, n, ro.data(), r.data(), ao.data(), dummy_xy, dummy_m, dE);
  if (rc) return rc;
  for (int i = 0; i < n; ++i) dE[i] = -dE[i];     // E(with u) - E(without u), energy_point_set.py:108-110
  return 0;
}

// merge_patches(method='distance') for every tile of the ctx at once (data_loaders.py:122-161): each tile holds the
// aggregated detections of one image on that image's score maps.  Papangelou of every point, the dedupe walk, the
// removals (EPointsSet.remove order), Papangelou of the survivors -- four launches for the whole batch, one copy back.
#define MPP_MERGE_MAX_POINTS 8192
extern "C" int mpp_merge_score(mpp_ctx *c, double distance, int cap, int32_t *n_out, int32_t *xy, double *marks, double *dE,
                               int32_t *n_removed) {
  if (!c || !n_out || cap < 0 || !(distance >= 0)) return fail(c, -1, "bad merge_score arguments");
  if (!c->have_maps || !c->have_model) return fail(c, -1, "mpp_set_maps / mpp_set_model have not been called");
  HIPCHK(c, hipSetDevice(c->device));
  int rc = push_state(c);
  if (rc) return rc;
  const int T = c->n_tiles;
  std::vector<int32_t> n0(T);
  HIPCHK(c, hipMemcpyAsync(n0.data(), c->n, sizeof(int32_t) * T, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  int max_n = 0;
  for (int t = 0; t < T; ++t) max_n = n0[t] > max_n ? n0[t] : max_n;
  if (max_n > MPP_MERGE_MAX_POINTS || (size_t)((c->cap + 7) & ~7) * 17 > (size_t)MPP_LDS_LIMIT - 256)      // (the walk's working set lives in LDS)
    return fail(c, -4, "merge_score: a tile holds %d points (the device walk takes at most %d): merge it on the host", max_n,
                MPP_MERGE_MAX_POINTS);
  const size_t TC = (size_t)T * c->cap;
  double *d_dE = nullptr, *ts = nullptr, *tr = nullptr, *ta = nullptr;
  int32_t *work = nullptr, *slot_of = nullptr, *tx = nullptr, *ty = nullptr, *d_rem = nullptr;
  hipError_t e = hipSuccess;
  auto A = [&](void **p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes ? bytes : 1); };
  A((void **)&d_dE, TC * 8); A((void **)&ts, TC * 8); A((void **)&tr, TC * 8); A((void **)&ta, TC * 8);
  A((void **)&work, TC * 4); A((void **)&slot_of, TC * 4); A((void **)&tx, TC * 4); A((void **)&ty, TC * 4); A((void **)&d_rem, (size_t)T * 4);
  std::vector<int32_t> h_rem(T, 0);
  void *extra_free[3] = {nullptr, nullptr, nullptr};
  if (e == hipSuccess && max_n > 0) {
    const int dist2 = (int)floor(distance * distance + 1e-9);
    // the from-scratch energies look their neighbours up in per-tile candidate grids, built on the device before each of
    // the two scorings (the removals in between move points)
    const int ncell = c->hp.nx * c->hp.ny;
    int32_t *gs = nullptr, *gc = nullptr, *gi = nullptr;
    if (ncell > 0 && max_n >= 64) {
      A((void **)&gs, (size_t)T * (ncell + 1) * 4); A((void **)&gc, (size_t)T * ncell * 4); A((void **)&gi, TC * 4);
    }
    const bool grid = gs && gc && gi && e == hipSuccess;
    if (e == hipSuccess) {
      if (grid) mpp_launch_grid_build_all(c->stream, c->dp, c->d_tiles, T, max_n, ncell, c->cap, gs, gc, gi);
      mpp_launch_papangelou_tiles(c->stream, c->dp, c->d_tiles, T, max_n, c->cap, d_dE, grid ? gs : nullptr, grid ? gi : nullptr, ncell + 1, c->cap);
      mpp_launch_dedupe_tiles(c->stream, c->d_tiles, T, max_n, c->cap, d_dE, dist2, work, slot_of, tx, ty, ts, tr, ta, d_rem);
      if (grid) mpp_launch_grid_build_all(c->stream, c->dp, c->d_tiles, T, max_n, ncell, c->cap, gs, gc, gi);
      mpp_launch_papangelou_tiles(c->stream, c->dp, c->d_tiles, T, max_n, c->cap, d_dE, grid ? gs : nullptr, grid ? gi : nullptr, ncell + 1, c->cap);
      e = hipGetLastError();
    }
    extra_free[0] = gs; extra_free[1] = gc; extra_free[2] = gi;
    if (e == hipSuccess) e = hipMemcpyAsync(h_rem.data(), d_rem, sizeof(int32_t) * T, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  }
  if (e == hipSuccess) rc = mpp_get_points_all(c, cap, n_out, xy, marks);
  if (e == hipSuccess && rc == 0 && dE && cap > 0 && max_n > 0) {
    const int m = max_n < cap ? max_n : cap;
    std::vector<double> h((size_t)T * m);
    e = hipMemcpy2D(h.data(), (size_t)m * 8, d_dE, (size_t)c->cap * 8, (size_t)m * 8, T, hipMemcpyDeviceToHost);
    for (int t = 0; t < T && e == hipSuccess; ++t)
      for (int i = 0; i < n_out[t] && i < m; ++i) dE[(size_t)t * cap + i] = h[(size_t)t * m + i];
  }
  if (n_removed) for (int t = 0; t < T; ++t) n_removed[t] = h_rem[t];
  void *fr[] = {d_dE, ts, tr, ta, work, slot_of, tx, ty, d_rem, extra_free[0], extra_free[1], extra_free[2]};
  for (void *p : fr) if (p) (void)hipFree(p);
  HIPCHK(c, e);
  return rc;
}

extern "C" int mpp_naive_init(mpp_ctx *c, double threshold, double nms_distance) {
  if (!c) return -1;
  int rc = push_state(c);
  if (rc) return rc;
  const int cand_cap = c->H * c->W;
  unsigned long long *cand = nullptr;
  HIPCHK(c, dalloc(&cand, (size_t)c->n_tiles * cand_cap));
  mpp_launch_naive_init(c->stream, c->dp, c->d_tiles, c->n_tiles, threshold, nms_distance, cand, cand_cap);
  hipError_t e = hipGetLastError(), e2 = hipStreamSynchronize(c->stream);
  (void)hipFree(cand);
  HIPCHK(c, e); HIPCHK(c, e2);
  std::vector<int32_t> herr(c->n_tiles);
  HIPCHK(c, hipMemcpy(herr.data(), c->errd, c->n_tiles * sizeof(int32_t), hipMemcpyDeviceToHost));
  for (int t = 0; t < c->n_tiles; ++t)
    if (herr[t]) return fail(c, -10 - herr[t], "naive init, tile %d: %s", t, chain_error_text(herr[t]));
  return 0;
}

extern "C" int mpp_set_schedule(mpp_ctx *c, double T0, double alpha, double T_target) {
  if (!c) return -1;
  if (!(T0 >= T_target)) return fail(c, -1, "t0 must be >= t_target");          // rjmcmc.py:71
  c->sched[0] = T0; c->sched[1] = alpha; c->sched[2] = T_target;
  if (c->have_maps) {
    std::vector<double> sched((size_t)c->n_tiles * 3);
    for (int t = 0; t < c->n_tiles; ++t) { sched[3 * t] = T0; sched[3 * t + 1] = alpha; sched[3 * t + 2] = T_target; }
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(c->T, sched.data(), sched.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(c->step, 0, c->n_tiles * sizeof(int64_t), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return 0;
}

// More slots per tile: the five configuration arrays are re-allocated with the new stride and copied.
static int grow_points(mpp_ctx *c, int new_cap) {
  const size_t T = (size_t)c->n_tiles;
  int32_t *px = nullptr, *py = nullptr;
  double *ps = nullptr, *pr = nullptr, *pa = nullptr;
  HIPCHK(c, dalloc(&px, T * new_cap)); HIPCHK(c, dalloc(&py, T * new_cap));
  HIPCHK(c, dalloc(&ps, T * new_cap)); HIPCHK(c, dalloc(&pr, T * new_cap)); HIPCHK(c, dalloc(&pa, T * new_cap));
  const size_t wi = (size_t)c->cap * sizeof(int32_t), wd = (size_t)c->cap * sizeof(double);
  const size_t ni = (size_t)new_cap * sizeof(int32_t), nd = (size_t)new_cap * sizeof(double);
  HIPCHK(c, hipMemcpy2DAsync(px, ni, c->px, wi, wi, T, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipMemcpy2DAsync(py, ni, c->py, wi, wi, T, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipMemcpy2DAsync(ps, nd, c->ps, wd, wd, T, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipMemcpy2DAsync(pr, nd, c->pr, wd, wd, T, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipMemcpy2DAsync(pa, nd, c->pa, wd, wd, T, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  (void)hipFree(c->px); (void)hipFree(c->py); (void)hipFree(c->ps); (void)hipFree(c->pr); (void)hipFree(c->pa);
  c->px = px; c->py = py; c->ps = ps; c->pr = pr; c->pa = pa;
  c->cap = new_cap; c->g_cap = 0;
  c->tiles_dirty = true; c->params_dirty = true;
  return 0;
}

static size_t chain_lds(mpp_ctx *c, int cap, int cell_cap) {
  const int ncell = c->hp.nx * c->hp.ny;
  const int spec = c->lanes > 0 ? 4 * c->lanes : c->spec;
  const int rb = c->hp.rowbase_lds ? c->H + 1 : 0;
  return mpp_chain_lds_bytes(cap, ncell, cell_cap, spec, rb, c->lanes > 0 ? 4 : c->spec);
}
// dynamic + static LDS of a chain: what has to fit the 160 KB of a CU
static size_t chain_lds_total(mpp_ctx *c, int cap, int cell_cap) {
  return chain_lds(c, cap, cell_cap) + mpp_chain_static_lds_bytes(c->lanes > 0 ? 4 : c->spec);
}

// The reference's point set has no capacity (Python sets, point_set.py:45-188); a chain here lives in one workgroup's
// LDS with `point_capacity` slots and `cell_capacity` entries per cell of the spatial hash.  A step that would exceed
// either stops the chain BEFORE the step (state, temperature and step counter of that moment are written back);
// with auto_grow (default) the capacity is doubled -- as long as the chain still fits the 160 KB of LDS -- and the same
// launch is issued again: finished tiles return at once, the stopped ones continue with the very next step, so the
// chain is the one an unlimited capacity would have produced.
// A chain evaluates MPP_U_SHAPE_REMAP -- three sigmoids of mark probabilities -- for every proposal that adds a rectangle
// (~10 % of its vector instructions).  The reference builds the remapped maps once per tile
// (energy_setup_legacy.py:142-147); so do chains here: [H][W][32] float64 per mark, holding exactly the summands the
// inline code forms (same expression, same device exp: the chain is byte-identical with and without the tables).  Only
// for models whose sole use of the mark maps is that term, only for contexts that run chains (the from-scratch energies of
// EPointsSet evaluate a few thousand points: inline), and only while 3 x 8 B x 32 per pixel fits the budget (2 GB).
static int ensure_remap_tables(mpp_ctx *c) {
  if (!c->remap_dirty) return 0;
  c->remap_dirty = false;
  const mpp_model &M = c->hp.model;
  int term = -1;
  bool other_mark_use = false;
  for (int k = 0; k < M.n_unit; ++k) {
    if (M.unit[k].kind == MPP_U_SHAPE_REMAP) term = term < 0 ? k : -2;
    if (M.unit[k].kind == MPP_U_MARK_NEG || M.unit[k].kind == MPP_U_MARK_REMAP) other_mark_use = true;
  }
  const size_t n = (size_t)c->n_maps * c->H * c->W * MPP_NCLASS, bytes = 3 * n * sizeof(double);
  const bool want = c->remap_mode != 0 && term >= 0 && !other_mark_use && (c->remap_mode == 1 || bytes <= c->remap_budget);
  bool have = c->remap[0] != nullptr;
  if (!want) {
    if (have) {
      for (int k = 0; k < 3; ++k) { (void)hipFree(c->remap[k]); c->remap[k] = nullptr; }
      c->tiles_dirty = true;
    }
    return 0;
  }
  for (int k = 0; k < 3; ++k) {
    if (!c->remap[k] && hipMalloc((void **)&c->remap[k], n * sizeof(double)) != hipSuccess) {
      (void)hipGetLastError();                          // no room: chains evaluate the sigmoids inline (same values)
      for (int j = 0; j < 3; ++j) if (c->remap[j]) { (void)hipFree(c->remap[j]); c->remap[j] = nullptr; }
      c->tiles_dirty = true;
      return 0;
    }
    mpp_launch_remap_table(c->stream, c->m[k], n, M.unit[term].p[k], M.unit[term].p[3 + k], c->remap[k]);
  }
  HIPCHK(c, hipGetLastError());
  c->tiles_dirty = true;
  return 0;
}

// rowpart / rowbase (the two-level CDF of the detection map a data-driven birth is drawn from) and boxsum (window sums of
// the translation kernel): made when the first chain is launched
static int ensure_birth_tables(mpp_ctx *c) {
  if (c->cdf_ready) return 0;
  const size_t hw = (size_t)c->H * c->W, M = (size_t)c->n_maps;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, dalloc(&c->rowpart, M * hw));
  HIPCHK(c, dalloc(&c->rowbase, M * (c->H + 1)));
  HIPCHK(c, dalloc(&c->rowtot, M * c->H));
  HIPCHK(c, dalloc(&c->boxsum, M * hw));
  mpp_launch_cdf(c->stream, (int)M, c->det, c->H, c->W, c->rowpart, c->rowbase, c->rowtot);
  HIPCHK(c, hipGetLastError());
  c->cdf_ready = true;
  c->box_dirty = true;
  c->tiles_dirty = true;
  return 0;
}

static int run_chain(mpp_ctx *c, int grid, int tile0, int64_t n_steps, uint64_t seed, uint32_t chain0,
                     const mpp_proposal *d_tape, int trace_tile, mpp_step_out *d_out, mpp_proposal *d_props) {
  if (!c->have_kernels) return fail(c, -1, "mpp_set_kernels has not been called");
  if (!c->have_maps) return fail(c, -1, "mpp_set_maps has not been called");
  if (!c->have_model) return fail(c, -1, "mpp_set_model has not been called");
  int rc = ensure_birth_tables(c);
  if (rc) return rc;
  rc = ensure_remap_tables(c);
  if (rc) return rc;
  rc = push_state(c);
  if (rc) return rc;
  // the row level of the birth CDF goes to LDS when it fits and the chain speculates (it shortens the slowest
  // wave of a round); throughput launches of one-wave chains keep their LDS for occupancy
  c->hp.rowbase_lds = (c->H <= 1024 && (c->lanes > 0 || c->spec > 1)) ? 1 : 0;
  if ((c->hp.n_kernels > MPP_K_SPLIT || has_classic(c->hp.model)) && !(c->lanes == 0 && (c->spec == 1 || c->spec == 8)))
    return fail(c, -1, "the split / merge kernels and the classic image energies are built for spec_waves 1 or 8 with spec_lanes 0");
  // deep rounds: chains of the shipped energy setups drawn from Philox (no tape, no split / merge, no classic image energy)
  int deep_nmax = 0;
  {
    const mpp_model &M = c->hp.model;
    const bool fast = M.n_pair == 2 && M.pair[0].kind == MPP_P_OVERLAP && M.pair[0].reduce == MPP_REDUCE_MAX &&
                      M.pair[1].kind == MPP_P_ALIGN && M.pair[1].reduce == MPP_REDUCE_MIN;
    if (c->deep > 0 && c->lanes == 0 && c->spec <= 8 && !d_tape && fast && c->hp.n_kernels <= MPP_K_SPLIT &&
        (!has_classic(M) || c->spec == 1 || c->spec == 8) && !c->hp.force_accept && c->hp.nx < 256 && c->hp.ny < 256) {
      deep_nmax = c->deep < 64 * c->spec ? c->deep : 64 * c->spec;
      if (deep_nmax < c->spec) deep_nmax = c->spec;
      if (c->H <= 1024) c->hp.rowbase_lds = 1;
      if (!c->deep_stats) {
        HIPCHK(c, hipMalloc((void **)&c->deep_stats, 256 * sizeof(unsigned long long)));
      }
      HIPCHK(c, hipMemsetAsync(c->deep_stats, 0, 256 * sizeof(unsigned long long), c->stream));
    }
  }
  mpp_launch_set_until(c->stream, c->d_tiles, tile0, grid, (long long)n_steps, c->until);
  HIPCHK(c, hipGetLastError());
  long long trace_base = 0;
  if (trace_tile >= 0) {
    int64_t s0 = 0;
    HIPCHK(c, hipMemcpyAsync(&s0, c->step + trace_tile, sizeof s0, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    trace_base = s0;
  }
  c->last_ms = 0.0;
  // many chains in one launch: prefer the instantiation that lets two waves share a SIMD
  const int occ = (grid >= 1024) ? 2 : 1;
  // hot start: one wave per step until the chain has cooled down (ERR_HANDOVER), then deep rounds -- the same chain either way
  // (launches of a few chains only: the launch that hands over ends when its LAST chain has cooled down, the others' CUs idle
  //  until then -- 256 tiles of config 5's scene lost 3 ms to that, one tile gains 6)
  bool hot_start = deep_nmax > 0 && c->handover && c->spec == 8 && c->lanes == 0 && trace_tile < 0 && !c->deep_fixed && n_steps >= 4096 &&
                   grid <= c->handover_tiles && chain_lds_total(c, c->cap, c->cell_cap) <= MPP_LDS_LIMIT;
  for (;;) {
    size_t lds = chain_lds(c, c->cap, c->cell_cap);
    // deep rounds need room for their step reports next to the chain state: halve the round until it fits, or do without
    int nmax = deep_nmax;
    const int ncell_ = c->hp.nx * c->hp.ny, rb_ = c->hp.rowbase_lds ? c->H + 1 : 0, ext_ = has_classic(c->hp.model) ? 1 : 0;
    while (nmax >= c->spec && nmax > 0 &&
           mpp_deep_lds_bytes(c->cap, ncell_, c->cell_cap, rb_, c->spec, nmax, ext_) + mpp_deep_static_lds_bytes(c->spec) > MPP_LDS_LIMIT)
      nmax /= 2;
    if (nmax < c->spec || nmax < 8) nmax = 0;
    if (c->cell_cap > 64) nmax = 0;        // (the deep kernel lists a cell's candidates in a 64-bit mask: fuller cells run one step per wave)
    if (hot_start && nmax > 0) nmax = 0;
    else hot_start = false;
    if (c->hp.handover != (hot_start ? c->handover_at : 0)) { c->hp.handover = hot_start ? c->handover_at : 0; c->params_dirty = true; if ((rc = push_state(c))) return rc; }
    if (nmax > 0) lds = mpp_deep_lds_bytes(c->cap, ncell_, c->cell_cap, rb_, c->spec, nmax, ext_);
    else if (chain_lds_total(c, c->cap, c->cell_cap) > MPP_LDS_LIMIT)
      return fail(c, -7, "chain state needs %zu B of LDS (> %d): lower point_capacity/cell_capacity/spec_waves or tile size",
                  lds, MPP_LDS_LIMIT);
    c->hp.cap = c->cap; c->hp.cell_cap = c->cell_cap;
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    if (nmax > 0) {
      int fixed = c->deep_fixed > nmax ? nmax : c->deep_fixed;
      if (fixed > 0) { fixed = fixed / c->spec * c->spec; if (fixed < c->spec) fixed = c->spec; }
      HIPCHK(c, mpp_launch_deep(c->stream, c->spec, occ, grid, lds, &c->hp, c->d_tiles, tile0, c->until, trace_base, seed, chain0,
                                trace_tile, d_out, d_props, nmax, fixed, c->deep_gain, c->deep_stats, has_classic(c->hp.model) ? 1 : 0));
    } else
    HIPCHK(c, mpp_launch_chain(c->stream, c->spec, c->lanes, occ, grid, lds, &c->hp, c->d_tiles, tile0, c->until, trace_base, seed,
                               chain0, d_tape, trace_tile, d_out, d_props));
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->last_ms += ms;
    std::vector<int32_t> herr(grid);
    HIPCHK(c, hipMemcpy(herr.data(), c->errd + tile0, grid * sizeof(int32_t), hipMemcpyDeviceToHost));
    bool cell = false, point = false, cooled = false;
    for (int t = 0; t < grid; ++t) {
      if (herr[t] == 1) cell = true;
      else if (herr[t] == 2) point = true;
      else if (herr[t] == 5) cooled = true;
      else if (herr[t]) return fail(c, -10 - herr[t], "tile %d: %s", tile0 + t, chain_error_text(herr[t]));
    }
    if (cooled) {                          // (a launch that ended for a capacity as well grows first and keeps its hot start)
      for (int t = 0; t < grid; ++t) if (herr[t] == 5) herr[t] = 0;
      HIPCHK(c, hipMemcpy(c->errd + tile0, herr.data(), grid * sizeof(int32_t), hipMemcpyHostToDevice));
      if (!cell && !point) { hot_start = false; continue; }
    }
    if (!cell && !point) return 0;
    const bool can_grow = c->auto_grow != 0;
    int new_cell = c->cell_cap, new_cap = c->cap;
    if (cell) new_cell = c->cell_cap * 2 > MPP_CELL_CAP_MAX ? MPP_CELL_CAP_MAX : c->cell_cap * 2;
    if (point) new_cap = c->cap * 2 > 65535 ? 65535 : c->cap * 2;
    if (!can_grow || (cell && new_cell == c->cell_cap) || (point && new_cap == c->cap) ||
        chain_lds_total(c, new_cap, new_cell) > MPP_LDS_LIMIT) {
      for (int t = 0; t < grid; ++t)
        if (herr[t]) return fail(c, -10 - herr[t], "tile %d: %s", tile0 + t, chain_error_text(herr[t]));
    }
    if (point && (rc = grow_points(c, new_cap))) return rc;
    c->cell_cap = new_cell;
    c->grow_events += 1;
    // clear the two overflow codes (sticky otherwise) and bring the tile table / parameters up to date
    for (int t = 0; t < grid; ++t) if (herr[t] == 1 || herr[t] == 2) herr[t] = 0;
    HIPCHK(c, hipMemcpy(c->errd + tile0, herr.data(), grid * sizeof(int32_t), hipMemcpyHostToDevice));
    c->params_dirty = true;
    if ((rc = push_state(c))) return rc;
  }
}

extern "C" int mpp_replay(mpp_ctx *c, int tile, int n, const mpp_proposal *tape, mpp_step_out *out) {
  int rc = check_tile(c, tile);
  if (rc) return rc;
  if (n <= 0) return 0;
  HIPCHK(c, hipSetDevice(c->device));
  mpp_proposal *d_tape = nullptr;
  mpp_step_out *d_out = nullptr;
  HIPCHK(c, dalloc(&d_tape, (size_t)n));
  if (out) HIPCHK(c, dalloc(&d_out, (size_t)n));
  hipError_t e = hipMemcpyAsync(d_tape, tape, (size_t)n * sizeof(mpp_proposal), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) rc = run_chain(c, 1, tile, n, 0, 0, d_tape, tile, d_out, nullptr);
  if (e == hipSuccess && out)
    e = hipMemcpy(out, d_out, (size_t)n * sizeof(mpp_step_out), hipMemcpyDeviceToHost);
  (void)hipFree(d_tape);
  if (d_out) (void)hipFree(d_out);
  HIPCHK(c, e);
  return rc;
}

extern "C" int mpp_run(mpp_ctx *c, int64_t n_steps, uint64_t seed, uint32_t chain0, int trace_tile, mpp_step_out *out,
                       mpp_proposal *props) {
  if (!c) return -1;
  if (!c->have_maps) return fail(c, -1, "mpp_set_maps has not been called");
  if (n_steps <= 0) return 0;
  HIPCHK(c, hipSetDevice(c->device));
  mpp_step_out *d_out = nullptr;
  mpp_proposal *d_props = nullptr;
  bool tr = trace_tile >= 0 && trace_tile < c->n_tiles;
  if (tr && out) HIPCHK(c, dalloc(&d_out, (size_t)n_steps));
  if (tr && props) HIPCHK(c, dalloc(&d_props, (size_t)n_steps));
  int rc = run_chain(c, c->n_tiles, 0, n_steps, seed, chain0, nullptr, tr ? trace_tile : -1, d_out, d_props);
  hipError_t e = hipSuccess;
  if (d_out) e = hipMemcpy(out, d_out, (size_t)n_steps * sizeof(mpp_step_out), hipMemcpyDeviceToHost);
  if (d_props && e == hipSuccess) e = hipMemcpy(props, d_props, (size_t)n_steps * sizeof(mpp_proposal), hipMemcpyDeviceToHost);
  if (d_out) (void)hipFree(d_out);
  if (d_props) (void)hipFree(d_props);
  HIPCHK(c, e);
  return rc;
}

extern "C" int mpp_step_index(mpp_ctx *c, int tile, int64_t *step) {
  int rc = check_tile(c, tile);
  if (rc) return rc;
  HIPCHK(c, hipMemcpy(step, c->step + tile, sizeof(int64_t), hipMemcpyDeviceToHost));
  return 0;
}
extern "C" int mpp_last_kernel_ms(mpp_ctx *c, double *ms) {
  if (!c || !ms) return -1;
  *ms = c->last_ms;
  return 0;
}

extern "C" int mpp_posnet_epilogue(mpp_ctx *c, int H, int W, int ldh, int ldw, const float *pos_out, double div_w,
                                   double div_b, float *det) {
  if (!c || !pos_out || !det || H <= 0 || W <= 0 || ldh < H || ldw < W) return fail(c, -1, "bad epilogue arguments");
  HIPCHK(c, hipSetDevice(c->device));
  mpp_launch_posnet_epilogue(c->stream, pos_out, H, W, ldh, ldw, (float)div_w, (float)div_b, det);
  HIPCHK(c, hipGetLastError());
  return 0;
}
extern "C" int mpp_affine_relu(mpp_ctx *c, void *x, int planes, int C, int64_t hw, int elem_bytes, const float *scale,
                               const float *shift) {
  if (!c || !x || !scale || !shift || planes <= 0 || C <= 0 || hw <= 0 || planes % C) return fail(c, -1, "bad affine_relu arguments");
  HIPCHK(c, hipSetDevice(c->device));
  if (mpp_launch_affine_relu(c->stream, x, planes, C, (size_t)hw, elem_bytes, scale, shift))
    return fail(c, -1, "affine_relu: element type must be float32 or bfloat16");
  HIPCHK(c, hipGetLastError());
  return 0;
}
extern "C" int mpp_posnet_epilogue_nhwc(mpp_ctx *c, int H, int W, int ldh, int ldw, const void *pos_out, int elem_bytes, double div_w,
                                       double div_b, float *det) {
  if (!c || !pos_out || !det || H <= 0 || W <= 0 || ldh < H || ldw < W) return fail(c, -1, "bad epilogue arguments");
  HIPCHK(c, hipSetDevice(c->device));
  if (mpp_launch_posnet_epilogue_nhwc(c->stream, pos_out, elem_bytes, H, W, ldw, (float)div_w, (float)div_b, det))
    return fail(c, -1, "posnet_epilogue_nhwc: element type must be float32 or bfloat16");
  HIPCHK(c, hipGetLastError());
  return 0;
}
extern "C" int mpp_shapenet_epilogue_nhwc(mpp_ctx *c, int H, int W, int ldh, int ldw, const void *logits, int elem_bytes, float *marks) {
  if (!c || !logits || !marks || H <= 0 || W <= 0 || ldh < H || ldw < W) return fail(c, -1, "bad epilogue arguments");
  HIPCHK(c, hipSetDevice(c->device));
  const int e = mpp_launch_shapenet_epilogue_nhwc(c->stream, logits, elem_bytes, H, W, ldw, marks);
  if (e == -1) return fail(c, -1, "shapenet_epilogue_nhwc: element type must be float32 or bfloat16");
  if (e == -2) return fail(c, -1, "shapenet_epilogue_nhwc: logits and marks must be 16-byte aligned");
  HIPCHK(c, hipGetLastError());
  return 0;
}
extern "C" int mpp_nhwc_glue(mpp_ctx *c, const void *x0, const void *x1, void *y, int H, int W, int C0, int C1, int pad, int pool,
                             int in_bytes, int out_bytes, const float *scale, const float *shift) {
  if (!c || !x0 || !y || H <= 0 || W <= 0 || C0 <= 0 || C1 < 0 || (C1 > 0 && !x1) || (pad != 0 && pad != 1) || (pad && (H < 2 || W < 2)) ||
      (scale == nullptr) != (shift == nullptr))
    return fail(c, -1, "bad nhwc_glue arguments");
  if (y == x0 && (pad || pool || C1)) return fail(c, -1, "nhwc_glue: in place only without pad / pool / cat");
  HIPCHK(c, hipSetDevice(c->device));
  if (mpp_launch_nhwc_glue(c->stream, x0, C1 > 0 ? x1 : x0, y, H, W, C0, C1, pad, pool ? 1 : 0, in_bytes, out_bytes, scale, shift))
    return fail(c, -1, "nhwc_glue: element types must be float32 or bfloat16");
  HIPCHK(c, hipGetLastError());
  return 0;
}
extern "C" int mpp_conv3x3_c32(mpp_ctx *c, const float *x0, const float *x1, int H, int W, const float *wp, const float *in_scale,
                               const float *in_shift, const float *out_scale, const float *out_shift, int relu, float *y) {
  if (!c || !x0 || !wp || !y || H < 2 || W < 2 || (!in_scale) != (!in_shift) || (!out_scale) != (!out_shift))
    return fail(c, -1, "bad conv3x3_c32 arguments");
  HIPCHK(c, hipSetDevice(c->device));
  if (mpp_launch_conv3x3_c32(c->stream, x0, x1, H, W, wp, in_scale, in_shift, out_scale, out_shift, relu, y))
    return fail(c, -2, "conv3x3_c32 launch failed: %s", hipGetErrorString(hipGetLastError()));
  return 0;
}

extern "C" int mpp_conv3x3_stem(mpp_ctx *c, const float *x, int H, int W, const float *wp, const float *scale, const float *shift,
                                float *y) {
  if (!c || !x || !wp || !scale || !shift || !y || H < 2 || W < 2) return fail(c, -1, "bad conv3x3_stem arguments");
  HIPCHK(c, hipSetDevice(c->device));
  if (mpp_launch_conv3x3_stem(c->stream, x, H, W, wp, scale, shift, y))
    return fail(c, -2, "conv3x3_stem launch failed (y must be 16-byte aligned): %s", hipGetErrorString(hipGetLastError()));
  return 0;
}
extern "C" int mpp_shapenet_heads(mpp_ctx *c, int H, int W, int ldh, int ldw, const float *h, const float *w, const float *b,
                                  float *marks_size, float *marks_ratio, float *marks_angle) {
  if (!c || !h || !w || !b || !marks_size || !marks_ratio || !marks_angle || H < 1 || W < 1 || ldh < H || ldw < W)
    return fail(c, -1, "bad shapenet_heads arguments");
  HIPCHK(c, hipSetDevice(c->device));
  const int rc = mpp_launch_shapenet_heads(c->stream, h, H, W, ldw, w, b, marks_size, marks_ratio, marks_angle);
  if (rc == -2 && (((uintptr_t)h | (uintptr_t)marks_size | (uintptr_t)marks_ratio | (uintptr_t)marks_angle) & 15))
    return fail(c, -1, "shapenet_heads: the activations and the mark maps must be 16-byte aligned");
  if (rc) return fail(c, -2, "shapenet_heads launch failed: %s", hipGetErrorString(hipGetLastError()));
  return 0;
}

extern "C" int mpp_quad_iou(mpp_ctx *c, int n, const double *a, int m, const double *b, double *out, int on_device) {
  if (!c || n < 0 || m < 0 || ((long long)n * m > 0 && (!a || !b || !out))) return fail(c, -1, "bad quad_iou arguments");
  if ((long long)n * m == 0) return 0;
  HIPCHK(c, hipSetDevice(c->device));
  if (on_device) {
    mpp_launch_quad_iou(c->stream, n, a, m, b, out);
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  double *da = nullptr, *db = nullptr, *dout = nullptr;
  const size_t sa = (size_t)n * 8 * sizeof(double), sb = (size_t)m * 8 * sizeof(double), so = (size_t)n * m * sizeof(double);
  int rc = 0;
  if (hipMalloc((void **)&da, sa) != hipSuccess || hipMalloc((void **)&db, sb) != hipSuccess ||
      hipMalloc((void **)&dout, so) != hipSuccess) rc = fail(c, -2, "quad_iou: device allocation failed");
  if (!rc && (hipMemcpyAsync(da, a, sa, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
              hipMemcpyAsync(db, b, sb, hipMemcpyHostToDevice, c->stream) != hipSuccess)) rc = fail(c, -2, "quad_iou: upload failed");
  if (!rc) {
    mpp_launch_quad_iou(c->stream, n, da, m, db, dout);
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(out, dout, so, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(c, -2, "quad_iou: kernel or download failed");
  }
  (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
  return rc;
}
extern "C" int mpp_shapenet_epilogue(mpp_ctx *c, int H, int W, int ldh, int ldw, const float *logits, float *marks) {
  if (!c || !logits || !marks || H <= 0 || W <= 0 || ldh < H || ldw < W) return fail(c, -1, "bad epilogue arguments");
  HIPCHK(c, hipSetDevice(c->device));
  mpp_launch_shapenet_epilogue(c->stream, logits, H, W, ldh, ldw, marks);
  HIPCHK(c, hipGetLastError());
  return 0;
}
