// mpp_gather.hip -- the detection sets of all tiles of a ctx packed into ONE fixed-capacity record buffer on the
// device: the send buffer of the all-gather that replaces the result list `Pool.map` hands back in
// models/mpp/mpp_model.py:250-262.  Records carry image coordinates (tile anchor added, as merge_patches does,
// models/mpp/data_loaders.py:133-138) and the global tile id, so that the gathered buffer of all ranks is the
// reference's `results` list in tile order.
//
// Layout: out [capacity + 1][MPP_GATHER_RECORD] float64, row 0 = (count, 0, ...), row 1 + k = record k =
// (tile id, x, y, size, ratio, angle, 0 = room for a score).  HBM-bound and tiny (a few hundred 56-byte records per
// tile); one workgroup per tile, the record offset of a tile is the sum of the counts of the tiles before it.
#include "mpp_device.hpp"

#define GATHER_RECORD 7

__global__ __launch_bounds__(256) void k_pack_detections(const TileRef *tiles, int n_tiles, const int32_t *tile_ids,
                                                         const int32_t *anchors, int capacity, double *out) {
  __shared__ int part[256];
  const int t = blockIdx.x, tid = threadIdx.x;
  int before = 0, total = 0;                 // points of the tiles before mine / of all tiles
  for (int j = tid; j < n_tiles; j += 256) {
    const int nj = *tiles[j].n;
    total += nj;
    if (j < t) before += nj;
  }
  part[tid] = before;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (tid < s) part[tid] += part[tid + s]; __syncthreads(); }
  const int offset = part[0];
  __syncthreads();
  if (t == 0) {
    part[tid] = total;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) part[tid] += part[tid + s]; __syncthreads(); }
    if (tid == 0) out[0] = (double)part[0];                   // the host checks count <= capacity
    if (tid > 0 && tid < GATHER_RECORD) out[tid] = 0.0;
  }
  const TileRef r = tiles[t];
  const int n = *r.n;
  const double id = (double)tile_ids[t], ax = (double)anchors[2 * t], ay = (double)anchors[2 * t + 1];
  for (int i = tid; i < n; i += 256) {
    const int row = offset + i;
    if (row >= capacity) break;                               // never write past the buffer; reported by the host
    double *o = out + (size_t)(1 + row) * GATHER_RECORD;
    o[0] = id; o[1] = (double)r.px[i] + ax; o[2] = (double)r.py[i] + ay;
    o[3] = r.ps[i]; o[4] = r.pr[i]; o[5] = r.pa[i]; o[6] = 0.0;
  }
}

extern "C" void mpp_launch_pack_detections(hipStream_t st, const TileRef *tiles, int n_tiles, const int32_t *tile_ids,
                                           const int32_t *anchors, int capacity, double *out) {
  hipLaunchKernelGGL(k_pack_detections, dim3(n_tiles), dim3(256), 0, st, tiles, n_tiles, tile_ids, anchors, capacity, out);
}
