// grid_kernels.hpp -- device side of GridMap<float>::creatGridMap and ACS_Rank::setPoints.
#pragma once
#include "wa_device.h"

// K7 voxelise.  model_grid_map.hpp:223-268 tests every voxel against every triangle; here one
// thread owns one voxel and streams the triangle list (wave-uniform address => scalar loads),
// so the O(T*N^3) work runs on 256 CUs and the result (an OR over triangles) is order-free.
// Float ops are written exactly as the reference evaluates them (no contraction).
__global__ __launch_bounds__(256) void k_voxelize(const float *__restrict__ tris, int64_t n_tris,
                                                  float precision, WaDims d,
                                                  const float *__restrict__ cx,
                                                  const float *__restrict__ cy,
                                                  const float *__restrict__ cz,
                                                  uint8_t *__restrict__ free_out)
{
    int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= d.n) return;
    int32_t x = (int32_t)(id % d.nx), y = (int32_t)((id / d.nx) % d.ny), z = (int32_t)(id / d.nxy);
    const float px = cx[x], py = cy[y], pz = cz[z];
    const double thr = 1.2 * precision;  // :256 compares in double
    bool isfree = true;
    for (int64_t t = 0; t < n_tris; t++) {
        const float *T = tris + t * 12;
        const float nxn = T[0], nyn = T[1], nzn = T[2];
        float D = -(T[3] * nxn + T[4] * nyn + T[5] * nzn);  // :224-226
        float mnx = T[3], mny = T[4], mnz = T[5], mxx = T[3], mxy = T[4], mxz = T[5];
#pragma unroll
        for (int v = 0; v < 3; v++) {  // :234-242
            float qx = T[3 + v * 3], qy = T[4 + v * 3], qz = T[5 + v * 3];
            mxx = qx > mxx ? qx : mxx; mxy = qy > mxy ? qy : mxy; mxz = qz > mxz ? qz : mxz;
            mnx = qx < mnx ? qx : mnx; mny = qy < mny ? qy : mny; mnz = qz < mnz ? qz : mnz;
        }
        mnx -= precision; mny -= precision; mnz -= precision;  // :243-248
        mxx += precision; mxy += precision; mxz += precision;
        float dist = px * nxn + py * nyn + pz * nzn + D;  // :252-254
        float ad = dist > 0 ? dist : -dist;               // my_abs :23
        if ((double)ad < thr && mnx <= px && px <= mxx && mny <= py && py <= mxy && mnz <= pz && pz <= mxz)
            isfree = false;
    }
    free_out[id] = isfree ? 1 : 0;
}

// K7, triangle-clipped: the reference's own comment says "local points can replace all points" (:250) but it
// still visits every voxel for every triangle.  Here one workgroup owns one triangle (blockIdx.y splits large
// boxes), finds the index box of the voxels whose coordinates lie inside the triangle's bounding box +- precision
// from the axis tables, and runs the same plane-distance / bounding-box test on those voxels only.  Identical
// arithmetic per (voxel, triangle) pair and an order-free OR => the same occupancy as k_voxelize; the work drops
// from T*N^3 tests to the sum of the box volumes.  free_out must be pre-filled with 1.
__global__ __launch_bounds__(256) void k_voxelize_clip(const float *__restrict__ tris, int64_t n_tris, float precision,
                                                       WaDims d, const float *__restrict__ cx, const float *__restrict__ cy,
                                                       const float *__restrict__ cz, uint8_t *__restrict__ free_out)
{
    __shared__ int32_t lo[3], hi[3];
    const int64_t t = blockIdx.x;
    if (t >= n_tris) return;
    const float *T = tris + t * 12;
    const float nxn = T[0], nyn = T[1], nzn = T[2];
    const float D = -(T[3] * nxn + T[4] * nyn + T[5] * nzn);  // :224-226
    float mnx = T[3], mny = T[4], mnz = T[5], mxx = T[3], mxy = T[4], mxz = T[5];
#pragma unroll
    for (int v = 0; v < 3; v++) {  // :234-242
        float qx = T[3 + v * 3], qy = T[4 + v * 3], qz = T[5 + v * 3];
        mxx = qx > mxx ? qx : mxx; mxy = qy > mxy ? qy : mxy; mxz = qz > mxz ? qz : mxz;
        mnx = qx < mnx ? qx : mnx; mny = qy < mny ? qy : mny; mnz = qz < mnz ? qz : mnz;
    }
    mnx -= precision; mny -= precision; mnz -= precision;  // :243-248
    mxx += precision; mxy += precision; mxz += precision;
    if (threadIdx.x < 3) { lo[threadIdx.x] = 0x7fffffff; hi[threadIdx.x] = -1; }
    __syncthreads();
    // index box: smallest / largest index on each axis whose coordinate passes the reference's own comparison
    for (int32_t i = threadIdx.x; i < d.nx; i += blockDim.x)
        if (mnx <= cx[i] && cx[i] <= mxx) { atomicMin(&lo[0], i); atomicMax(&hi[0], i); }
    for (int32_t i = threadIdx.x; i < d.ny; i += blockDim.x)
        if (mny <= cy[i] && cy[i] <= mxy) { atomicMin(&lo[1], i); atomicMax(&hi[1], i); }
    for (int32_t i = threadIdx.x; i < d.nz; i += blockDim.x)
        if (mnz <= cz[i] && cz[i] <= mxz) { atomicMin(&lo[2], i); atomicMax(&hi[2], i); }
    __syncthreads();
    const int32_t x0 = lo[0], y0 = lo[1], z0 = lo[2];
    const int32_t wx = hi[0] - x0 + 1, wy = hi[1] - y0 + 1, wz = hi[2] - z0 + 1;
    if (wx <= 0 || wy <= 0 || wz <= 0) return;   // NaN vertices or a box outside the grid: nothing can match
    const double thr = 1.2 * precision;  // :256 compares in double
    const int64_t box = (int64_t)wx * wy * wz;
    for (int64_t q = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; q < box; q += (int64_t)gridDim.y * blockDim.x) {
        const int32_t x = x0 + (int32_t)(q % wx), y = y0 + (int32_t)((q / wx) % wy), z = z0 + (int32_t)(q / ((int64_t)wx * wy));
        const float px = cx[x], py = cy[y], pz = cz[z];
        const float dist = px * nxn + py * nyn + pz * nzn + D;  // :252-254
        const float ad = dist > 0 ? dist : -dist;               // my_abs :23
        if ((double)ad < thr && mnx <= px && px <= mxx && mny <= py && py <= mxy && mnz <= pz && pz <= mxz)
            free_out[(int64_t)z * d.nxy + (int64_t)y * d.nx + x] = 0;
    }
}

// ACS_Rank::setPoints (ACSRank_3D.hpp:537-565): last free voxel in raster order within
// t = (float)(1.2*precision) of the point on all three axes == max id among matches.
__global__ __launch_bounds__(256) void k_resolve_points(WaDims d, const float *__restrict__ cx,
                                                        const float *__restrict__ cy,
                                                        const float *__restrict__ cz,
                                                        const uint8_t *__restrict__ free_,
                                                        float precision, const float *__restrict__ pts,
                                                        int32_t n_pts, long long *__restrict__ ids)
{
    int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= d.n || !free_[id]) return;
    int32_t x = (int32_t)(id % d.nx), y = (int32_t)((id / d.nx) % d.ny), z = (int32_t)(id / d.nxy);
    const float px = cx[x], py = cy[y], pz = cz[z];
    const float t = (float)(1.2 * (double)precision);  // `float t = 1.2*precision` :544
    for (int32_t p = 0; p < n_pts; p++) {
        float dx = pts[3 * p] - px, dy = pts[3 * p + 1] - py, dz = pts[3 * p + 2] - pz;
        dx = dx > 0 ? dx : -dx; dy = dy > 0 ? dy : -dy; dz = dz > 0 ? dz : -dz;
        if (dx < t && dy < t && dz < t) atomicMax(&ids[p], (long long)id);
    }
}

// Same result with one workgroup per point: the voxels within t of the point on every axis form an index box
// (read off the axis tables with the same |pt - c| < t comparison); the answer is the largest free id inside it.
// ids must be pre-filled with -1.
__global__ __launch_bounds__(256) void k_resolve_points_clip(WaDims d, const float *__restrict__ cx, const float *__restrict__ cy,
                                                             const float *__restrict__ cz, const uint8_t *__restrict__ free_,
                                                             float precision, const float *__restrict__ pts, int32_t n_pts,
                                                             long long *__restrict__ ids)
{
    __shared__ int32_t lo[3], hi[3];
    __shared__ long long best;
    const int32_t p = blockIdx.x;
    if (p >= n_pts) return;
    const float t = (float)(1.2 * (double)precision);  // `float t = 1.2*precision` :544
    const float qx = pts[3 * p], qy = pts[3 * p + 1], qz = pts[3 * p + 2];
    if (threadIdx.x < 3) { lo[threadIdx.x] = 0x7fffffff; hi[threadIdx.x] = -1; }
    if (threadIdx.x == 0) best = -1;
    __syncthreads();
    for (int32_t i = threadIdx.x; i < d.nx; i += blockDim.x) {
        float v = qx - cx[i];
        v = v > 0 ? v : -v;
        if (v < t) { atomicMin(&lo[0], i); atomicMax(&hi[0], i); }
    }
    for (int32_t i = threadIdx.x; i < d.ny; i += blockDim.x) {
        float v = qy - cy[i];
        v = v > 0 ? v : -v;
        if (v < t) { atomicMin(&lo[1], i); atomicMax(&hi[1], i); }
    }
    for (int32_t i = threadIdx.x; i < d.nz; i += blockDim.x) {
        float v = qz - cz[i];
        v = v > 0 ? v : -v;
        if (v < t) { atomicMin(&lo[2], i); atomicMax(&hi[2], i); }
    }
    __syncthreads();
    const int32_t x0 = lo[0], y0 = lo[1], z0 = lo[2];
    const int32_t wx = hi[0] - x0 + 1, wy = hi[1] - y0 + 1, wz = hi[2] - z0 + 1;
    if (wx <= 0 || wy <= 0 || wz <= 0) return;
    const int64_t box = (int64_t)wx * wy * wz;
    long long mine = -1;
    for (int64_t q = threadIdx.x; q < box; q += blockDim.x) {
        const int32_t x = x0 + (int32_t)(q % wx), y = y0 + (int32_t)((q / wx) % wy), z = z0 + (int32_t)(q / ((int64_t)wx * wy));
        const int64_t id = (int64_t)z * d.nxy + (int64_t)y * d.nx + x;
        if (!free_[id]) continue;
        float dx = qx - cx[x], dy = qy - cy[y], dz = qz - cz[z];
        dx = dx > 0 ? dx : -dx; dy = dy > 0 ? dy : -dy; dz = dz > 0 ? dz : -dz;
        if (dx < t && dy < t && dz < t && id > mine) mine = id;
    }
    if (mine >= 0) atomicMax(&best, mine);
    __syncthreads();
    if (threadIdx.x == 0) ids[p] = best;
}

__global__ __launch_bounds__(256) void k_count_free(const uint8_t *__restrict__ free_, int64_t n,
                                                    unsigned long long *__restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long c = 0;
    for (; i < n; i += (int64_t)gridDim.x * blockDim.x) c += free_[i] ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}
