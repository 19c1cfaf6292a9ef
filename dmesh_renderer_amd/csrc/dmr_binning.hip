// dmr_binning.hip -- projection, tile binning and per-tile depth sort (gfx950).
//
// Replaces, for both renderers:
//   preprocessPointCUDA   cuda_rasterizer/forward.cu:17-47   (cuda_renderer/forward.cu:21-52)
//   preprocessFaceCUDA    cuda_rasterizer/forward.cu:76-149  (cuda_renderer/forward.cu:178-260)
//   cub InclusiveSum + duplicateWithKeys + cub SortPairs + identifyTileRanges
//                         cuda_rasterizer/rasterizer_impl.cu:44-124,278-338
//
// The reference emits one (tile<<32 | depth_bits, face) pair per touched tile and runs
// ONE global 64-bit radix sort over all R pairs (k = ceil((32+bit)/8) passes, ~24 B per
// pair per pass).  Here the tile is known at emission time, so pairs are counted and
// scattered straight into their tile's segment (counting sort on the tile digit) and
// each segment is then sorted by (depth_bits, face_id) inside LDS by one workgroup
// (dmr_sort.hpp: the consumer's workgroup of that tile up to 8 192 tiles, k_sort_tiles above).
// Within a (view, tile) a face occurs at most once, so (depth_bits, face_id) is a total
// order and equals the order of the reference's STABLE sort, whose ties keep emission
// order = ascending face id (Q6).  The result -- per-tile face lists and ranges -- is
// bit-identical to the reference's sorted list, whatever order the scatter ran in.
#include <algorithm>

#include "dmr_kernels.hpp"
#include "dmr_sort.hpp"

namespace dmr {

// ---------------------------------------------------------------------------
// 1. per (view, vertex): pixel coordinates, NDC z and the per-view depth attribute packed
//    into one 16-byte record so later stages gather a vertex with a single dwordx4 load.
// ---------------------------------------------------------------------------
// Element k of a [16] column-major matrix whose 4x4 block may be stored transposed (dmr_scene.mats_transposed).
__device__ __forceinline__ float mat_at(const float* __restrict__ m, int k, bool transposed) {
    return m[transposed ? 4 * (k & 3) + (k >> 2) : k];
}

// Also writes the four matrices in contract layout to `mats` ([mv | proj | inv_mv | inv_proj], [B,16] each) in the
// image buffer: every later kernel of the forward and the whole backward read them from there.
__global__ void __launch_bounds__(256)
k_project_verts(int B, int P, const float* __restrict__ verts, const float* __restrict__ mv_mats,
                const float* __restrict__ proj_mats, const float* __restrict__ inv_mv_mats,
                const float* __restrict__ inv_proj_mats, int transposed, const float* __restrict__ verts_depth,
                int W, int H, float4* __restrict__ vproj, float* __restrict__ mats,
                uint32_t* __restrict__ counters, uint32_t ncounters) {
    {   // every block zeroes a slice of the tile counters (tile_count | tile_hits) the next kernels add into
        const uint32_t per = (ncounters + gridDim.x - 1) / gridDim.x;
        const uint32_t z0 = min(ncounters, blockIdx.x * per), z1 = min(ncounters, z0 + per);
        for (uint32_t i = z0 + threadIdx.x; i < z1; i += 256) counters[i] = 0u;
    }
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < 64 * B; i += 256) {
            const int m = i / (16 * B), r = i % (16 * B);
            const float* src = m == 0 ? mv_mats : (m == 1 ? proj_mats : (m == 2 ? inv_mv_mats : inv_proj_mats));
            mats[i] = mat_at(src + 16 * (r >> 4), r & 15, (transposed >> m) & 1);
        }
    }
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)B * P) return;
    const int b = (int)(idx / P), p = (int)(idx % P);
    float mv[16], pr[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        mv[k] = mat_at(mv_mats + 16 * b, k, transposed & 1);
        pr[k] = mat_at(proj_mats + 16 * b, k, (transposed >> 1) & 1);
    }
    V3 pv = xform4x3(load_v3(verts, p), mv);
    V4 pc = xform4x4(pv, pr);
    float p_w = (float)(1.0 / (double)clamp_w(pc.w));  // double divide (forward.cu:38)
    V3 n = {pc.x * p_w, pc.y * p_w, pc.z * p_w};
    vproj[idx] = make_float4(ndc2pix(n.x, W), ndc2pix(n.y, H), n.z, verts_depth[idx]);
}

// ---------------------------------------------------------------------------
// 2 / 4. per (view, face): cull, tile rect, sort depth, count the face into every tile of its rect (setup);
//   scatter every (face, tile) pair into its tile's segment, key = depth_bits << 32 | face_id (scatter).
//   rect is kept (packed 4 x u16) so the scatter pass does not redo the float work.
//   Global u32 atomics on ~3000 hot tile counters were the cost of both passes (69 + 80 us at C4 for
//   0.9 M increments; 0.46 + 0.65 ms at C5).  Here a workgroup takes 256 x FPT consecutive faces, counts them
//   into an LDS histogram (ds_add_u32), and touches global memory once per (workgroup, non-empty tile): the count
//   pass adds the bin, the scatter pass reserves the bin's slots with ONE returning atomic and then hands
//   out slots with returning LDS atomics.  The histogram covers a WINDOW of tiles: the bounding box of the
//   workgroup's face rects in the view of its first face (consecutive faces of a mesh are neighbours on screen, so
//   the box is a few hundred tiles even when the image has a million).  Faces with huge rects, faces of another
//   view and whole workgroups whose box exceeds LDS_HIST_MAX tiles use the global counters directly.
// ---------------------------------------------------------------------------
constexpr int LDS_HIST_MAX = 8192;    // 32 KiB of counters
// Faces per thread (FPT): 4 for large meshes -- a long run of consecutive faces per workgroup amortises the window's
// set-up and global atomics -- fewer when that would leave the chip empty: C3's 50 688 faces were 50 workgroups of
// 1024 faces, each thread looping over four 4x4-tile rects (scatter 34 us, 14 us with one face per thread); C4's
// 500 000 faces: set-up / scatter 14.8 / 17.7 us at FPT 4, 13.2 / 14.5 at 2, 14.5 / 16.9 at 1; C5's 8 M: FPT 4.
__host__ inline int bin_fpt(int64_t n) { return n < 256 * 1024 ? 1 : n < 1024 * 1024 ? 2 : 4; }
constexpr uint32_t BIG_RECT = 256;    // tiles; larger rects go straight to global atomics

struct BinWindow { int x0, y0, wx, wy, view; bool lds; };

// Faces whose rect covers more than BIG_RECT tiles (a ground plane, a hull face: up to every tile of the image) are
// not emitted by their own thread -- one thread looping over 8160 tiles with (returning) atomics was 1.9 + 2.2 ms for
// 64 screen-filling triangles at 800x800 (scripts/time_huge.py) -- but queued in LDS and emitted by the whole
// workgroup, 256 tiles at a time.  The queue holds BIG_MAX faces; further ones fall back to their own thread.
constexpr int BIG_MAX = 128;
struct BigFace { uint32_t rect_lo, rect_hi, tile_base, key_lo, key_hi; };

// tiles i = tid, tid + 256, ... of a rect in row-major order, without a division per tile
struct RectWalk {
    uint32_t w, n, i, x, y, dx, dy;
    __device__ __forceinline__ RectWalk(uint32_t minx, uint32_t miny, uint32_t maxx, uint32_t maxy, uint32_t tid) {
        w = maxx - minx; n = w * (maxy - miny); i = tid;
        y = miny + tid / w; x = minx + tid % w; dy = 256u / w; dx = 256u % w;
        x0 = minx;
    }
    uint32_t x0;
    __device__ __forceinline__ bool valid() const { return i < n; }
    __device__ __forceinline__ void next() {
        i += 256u; y += dy; x += dx;
        if (x >= x0 + w) { x -= w; y++; }
    }
};

// s_box: {min x, min y, max x, max y} of the rects that want the LDS histogram; uniform result.  Threads reduce
// their own faces, waves reduce with shuffles, one lane per wave touches LDS (every thread doing ds_min / ds_max on
// the same four words serialises: 2 cycles per lane and atomic).
template <int FPT>
__device__ __forceinline__ BinWindow bin_window(int* s_box, int view, const uint2* rr, const uint32_t* touched,
                                                const bool* mine, int tid) {
    int bx0 = 0x7fffffff, by0 = 0x7fffffff, bx1 = 0, by1 = 0;
#pragma unroll
    for (int it = 0; it < FPT; it++) {
        if (!mine[it] || touched[it] == 0 || touched[it] > BIG_RECT) continue;
        bx0 = min(bx0, (int)(rr[it].x & 0xffffu)); by0 = min(by0, (int)(rr[it].x >> 16));
        bx1 = max(bx1, (int)(rr[it].y & 0xffffu)); by1 = max(by1, (int)(rr[it].y >> 16));
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        bx0 = min(bx0, __shfl_xor(bx0, d, 64)); by0 = min(by0, __shfl_xor(by0, d, 64));
        bx1 = max(bx1, __shfl_xor(bx1, d, 64)); by1 = max(by1, __shfl_xor(by1, d, 64));
    }
    if (tid == 0) { s_box[0] = 0x7fffffff; s_box[1] = 0x7fffffff; s_box[2] = 0; s_box[3] = 0; }
    __syncthreads();
    if ((tid & 63) == 0) {
        atomicMin(&s_box[0], bx0); atomicMin(&s_box[1], by0); atomicMax(&s_box[2], bx1); atomicMax(&s_box[3], by1);
    }
    __syncthreads();
    BinWindow w;
    w.x0 = s_box[0]; w.y0 = s_box[1]; w.wx = s_box[2] - s_box[0]; w.wy = s_box[3] - s_box[1]; w.view = view;
    w.lds = w.wx > 0 && w.wy > 0 && (int64_t)w.wx * w.wy <= LDS_HIST_MAX;
    if (!w.lds) { w.wx = 0; w.wy = 0; }
    return w;
}

template <bool TET, int FPT>
__global__ void __launch_bounds__(256)
k_setup_faces_lds(int B, int P, int F, const int* __restrict__ faces, const float4* __restrict__ vproj,
                  int gx, int gy, int r0, int r1,
                  uint2* __restrict__ face_rect, float* __restrict__ key_depth, float* __restrict__ max_depth,
                  uint32_t* __restrict__ tiles_touched, uint32_t* __restrict__ tile_count) {
    __shared__ uint32_t s_hist[LDS_HIST_MAX];
    __shared__ int s_box[4];
    __shared__ BigFace s_big[BIG_MAX];
    __shared__ uint32_t s_nbig;
    const int tid = threadIdx.x;
    if (tid == 0) s_nbig = 0u;  // (bin_window's barriers order this before the queueing below)
    const int64_t BF = (int64_t)B * F;
    const int64_t base = (int64_t)blockIdx.x * (256 * FPT);
    const int view = (int)(base / F);
    uint2 rr[FPT];
    uint32_t touched[FPT];
    bool mine[FPT];  // face of the window's view
#pragma unroll
    for (int it = 0; it < FPT; it++) {
        const int64_t idx = base + it * 256 + tid;
        touched[it] = 0u; rr[it] = make_uint2(0, 0); mine[it] = false;
        if (idx >= BF) continue;
        const int b = (int)(idx / F), f = (int)(idx % F);
        mine[it] = b == view;
        const int v0 = faces[3 * f], v1 = faces[3 * f + 1], v2 = faces[3 * f + 2];
        const float4 a0 = vproj[(int64_t)b * P + v0], a1 = vproj[(int64_t)b * P + v1], a2 = vproj[(int64_t)b * P + v2];
        float max_z = a0.z, min_z = a0.z, depth = 0.0f;
        depth += a0.z;
        max_z = fmaxf(max_z, a1.z); min_z = fminf(min_z, a1.z); depth += a1.z;
        max_z = fmaxf(max_z, a2.z); min_z = fminf(min_z, a2.z); depth += a2.z;
        depth = depth / 3.0f;
        Rect r = {0, 0, 0, 0};
        if (!(max_z < -1.0f || min_z > 1.0f)) {
            r = tile_rect({a0.x, a0.y}, {a1.x, a1.y}, {a2.x, a2.y}, gx, gy, r0, r1);
            touched[it] = (r.maxy - r.miny) * (r.maxx - r.minx);
        }
        if (touched[it] == 0) r = {0, 0, 0, 0};
        auto map01 = [](float z) { float d = (z + 1.0f) * 0.5f; if (d < 0.0f) d = 0.0f; if (d > 1.0f) d = 1.0f; return d; };
        rr[it] = make_uint2(r.minx | (r.miny << 16), r.maxx | (r.maxy << 16));
        tiles_touched[idx] = touched[it];
        face_rect[idx] = rr[it];
        key_depth[idx] = touched[it] ? (TET ? map01(min_z) : map01(depth)) : 0.0f;
        if (TET) max_depth[idx] = touched[it] ? map01(max_z) : 0.0f;
    }
    const BinWindow w = bin_window<FPT>(s_box, view, rr, touched, mine, tid);
    const int nw = w.wx * w.wy;
    for (int t = tid; t < nw; t += 256) s_hist[t] = 0u;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < FPT; it++) {
        if (touched[it] == 0) continue;
        const int64_t idx = base + it * 256 + tid;
        const uint32_t tb = (uint32_t)(idx / F) * gx * gy;
        const uint32_t minx = rr[it].x & 0xffffu, miny = rr[it].x >> 16, maxx = rr[it].y & 0xffffu, maxy = rr[it].y >> 16;
        if (touched[it] > BIG_RECT) {
            const uint32_t q = atomicAdd(&s_nbig, 1u);
            if (q < (uint32_t)BIG_MAX) { s_big[q] = BigFace{rr[it].x, rr[it].y, tb, 0u, 0u}; continue; }
        }
        if (w.lds && mine[it] && touched[it] <= BIG_RECT) {
            for (uint32_t y = miny; y < maxy; y++)
                for (uint32_t x = minx; x < maxx; x++) atomicAdd(&s_hist[(y - w.y0) * w.wx + (x - w.x0)], 1u);
        } else {
            for (uint32_t y = miny; y < maxy; y++)
                for (uint32_t x = minx; x < maxx; x++) atomicAdd(&tile_count[tb + y * gx + x], 1u);
        }
    }
    __syncthreads();
    for (int y = tid / 64; y < w.wy; y += 4)
        for (int x = tid % 64; x < w.wx; x += 64) {
            const uint32_t c = s_hist[y * w.wx + x];
            if (c) atomicAdd(&tile_count[(uint32_t)view * gx * gy + (uint32_t)(w.y0 + y) * gx + (uint32_t)(w.x0 + x)], c);
        }
    const uint32_t nbig = min(s_nbig, (uint32_t)BIG_MAX);
    for (uint32_t q = 0; q < nbig; q++) {
        const BigFace f = s_big[q];
        for (RectWalk t(f.rect_lo & 0xffffu, f.rect_lo >> 16, f.rect_hi & 0xffffu, f.rect_hi >> 16, (uint32_t)tid); t.valid(); t.next())
            atomicAdd(&tile_count[f.tile_base + t.y * gx + t.x], 1u);
    }
}

template <int FPT>
__global__ void __launch_bounds__(256)
k_scatter_faces_lds(int B, int F, int gx, int gy, const uint2* __restrict__ face_rect,
                    const float* __restrict__ key_depth, const uint32_t* __restrict__ tiles_touched,
                    uint32_t* __restrict__ tile_cursor, uint64_t* __restrict__ keys, uint32_t capacity,
                    unsigned long long* __restrict__ mask_offset_dst, unsigned long long mask_offset, unsigned long long mask_first) {
    // where the binning buffer of THIS capacity keeps the coverage masks (dmr_kernels.hpp, TriImageState::mask_offset)
    if (mask_offset_dst && blockIdx.x == 0 && threadIdx.x == 0) { mask_offset_dst[0] = mask_offset; mask_offset_dst[1] = mask_first; }
    __shared__ uint32_t s_hist[LDS_HIST_MAX];
    __shared__ int s_box[4];
    __shared__ BigFace s_big[BIG_MAX];
    __shared__ uint32_t s_nbig;
    const int tid = threadIdx.x;
    if (tid == 0) s_nbig = 0u;
    const int64_t BF = (int64_t)B * F;
    const int64_t base = (int64_t)blockIdx.x * (256 * FPT);
    const int view = (int)(base / F);
    uint2 rr[FPT];
    uint32_t touched[FPT];
    bool mine[FPT];
#pragma unroll
    for (int it = 0; it < FPT; it++) {
        const int64_t idx = base + it * 256 + tid;
        touched[it] = idx < BF ? tiles_touched[idx] : 0u;
        rr[it] = touched[it] ? face_rect[idx] : make_uint2(0, 0);
        mine[it] = idx < BF && (int)(idx / F) == view;
    }
    const BinWindow w = bin_window<FPT>(s_box, view, rr, touched, mine, tid);
    const int nw = w.wx * w.wy;
    for (int t = tid; t < nw; t += 256) s_hist[t] = 0u;
    __syncthreads();
    // pass 1: count this workgroup's entries per tile of the window
#pragma unroll
    for (int it = 0; it < FPT; it++) {
        if (!(w.lds && mine[it]) || touched[it] == 0 || touched[it] > BIG_RECT) continue;
        const uint32_t minx = rr[it].x & 0xffffu, miny = rr[it].x >> 16, maxx = rr[it].y & 0xffffu, maxy = rr[it].y >> 16;
        for (uint32_t y = miny; y < maxy; y++)
            for (uint32_t x = minx; x < maxx; x++) atomicAdd(&s_hist[(y - w.y0) * w.wx + (x - w.x0)], 1u);
    }
    __syncthreads();
    // reserve the workgroup's slots of every non-empty tile with one returning atomic; the bin now holds the cursor
    for (int y = tid / 64; y < w.wy; y += 4)
        for (int x = tid % 64; x < w.wx; x += 64) {
            const uint32_t c = s_hist[y * w.wx + x];
            if (c) s_hist[y * w.wx + x] = atomicAdd(&tile_cursor[(uint32_t)view * gx * gy + (uint32_t)(w.y0 + y) * gx + (uint32_t)(w.x0 + x)], c);
        }
    __syncthreads();
    // pass 2: hand out slots
#pragma unroll
    for (int it = 0; it < FPT; it++) {
        if (touched[it] == 0) continue;
        const int64_t idx = base + it * 256 + tid;
        const int f = (int)(idx % F);
        const uint32_t tb = (uint32_t)(idx / F) * gx * gy;
        const uint32_t minx = rr[it].x & 0xffffu, miny = rr[it].x >> 16, maxx = rr[it].y & 0xffffu, maxy = rr[it].y >> 16;
        const uint64_t key = ((uint64_t)__float_as_uint(key_depth[idx]) << 32) | (uint32_t)f;
        if (touched[it] > BIG_RECT) {
            const uint32_t q = atomicAdd(&s_nbig, 1u);
            if (q < (uint32_t)BIG_MAX) { s_big[q] = BigFace{rr[it].x, rr[it].y, tb, (uint32_t)key, (uint32_t)(key >> 32)}; continue; }
        }
        const bool direct = !(w.lds && mine[it]) || touched[it] > BIG_RECT;
        for (uint32_t y = miny; y < maxy; y++)
            for (uint32_t x = minx; x < maxx; x++) {
                const uint32_t slot = direct ? atomicAdd(&tile_cursor[tb + y * gx + x], 1u)
                                             : atomicAdd(&s_hist[(y - w.y0) * w.wx + (x - w.x0)], 1u);
                if (slot < capacity) keys[slot] = key;
            }
    }
    __syncthreads();
    const uint32_t nbig = min(s_nbig, (uint32_t)BIG_MAX);
    for (uint32_t q = 0; q < nbig; q++) {
        const BigFace f = s_big[q];
        const uint64_t key = ((uint64_t)f.key_hi << 32) | f.key_lo;
        // four returning atomics in flight per thread (one at a time: 0.28 ms for 64 screen-filling faces in one workgroup)
        for (RectWalk t(f.rect_lo & 0xffffu, f.rect_lo >> 16, f.rect_hi & 0xffffu, f.rect_hi >> 16, (uint32_t)tid); t.valid();) {
            uint32_t tile[4], slot[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                tile[u] = 0xffffffffu;
                if (t.valid()) { tile[u] = f.tile_base + t.y * gx + t.x; t.next(); }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) slot[u] = tile[u] != 0xffffffffu ? atomicAdd(&tile_cursor[tile[u]], 1u) : 0xffffffffu;
#pragma unroll
            for (int u = 0; u < 4; u++) if (slot[u] < capacity) keys[slot[u]] = key;
        }
    }
}

// ---------------------------------------------------------------------------
// 3. exclusive scan of the per-tile counts -> segment starts (= the reference's `ranges`),
//    cursor copy for the scatter, and R.  One workgroup: n = B * tiles is small (C4: 8160).
// ---------------------------------------------------------------------------
// Also emits tile_order: the tiles bucket-sorted by descending list length (64 buckets of 16 entries).  The
// compositing kernels take their tile from it, longest first: with C4's lengths (64..650 entries, busy tiles
// clustered) dispatch in image order finishes 33 % above the ideal makespan, longest-first 10 % (simulated on the
// measured length distribution).
constexpr int ORDER_BUCKETS = 64;
__device__ __forceinline__ int order_bucket(uint32_t n) { return ORDER_BUCKETS - 1 - (int)min(n >> 4, (uint32_t)(ORDER_BUCKETS - 1)); }
// Neighbouring tiles have lists of similar length, so the lanes of a wave mostly want the SAME bucket, and same-address
// LDS atomics retire one lane at a time (scripts/micro/lds_atomics.hip).  Every bucket therefore has ORDER_COPIES
// counters, picked by lane: at most four lanes of a wave meet on one.  The order inside a bucket is arbitrary anyway.
// (Measured: on its own this changed nothing at C4 -- 16.4 us before and after; the kernel's time was its uncoalesced
// global accesses, see the slab below.  Kept: it bounds the worst case of a frame whose busy tiles share one bucket.)
constexpr int ORDER_COPIES = 16;
constexpr int ORDER_CELLS = ORDER_BUCKETS * ORDER_COPIES;  // + 1 cell for the empty tiles, behind all the others
__device__ __forceinline__ int order_cell(uint32_t n, int lane) { return order_bucket(n) * ORDER_COPIES + (lane & (ORDER_COPIES - 1)); }

// Both scans are one 1024-thread workgroup up to SCAN_SLAB tiles (C4: 8160), SCAN_BATCH tiles per thread.
constexpr int SCAN_BATCH = 8;

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t local, int tid, uint32_t* wave_sum /*[17]*/) {
    const int lane = tid & 63, wave = tid >> 6;
    uint32_t incl = local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    if (tid < 64) {
        const uint32_t w = tid < 16 ? wave_sum[tid] : 0u;
        uint32_t wi = w;
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            uint32_t o = __shfl_up(wi, d, 64);
            if (tid >= d) wi += o;
        }
        if (tid < 16) wave_sum[tid] = wi - w;
        if (tid == 15) wave_sum[16] = wi;
    }
    __syncthreads();
    return wave_sum[wave] + incl - local;  // total in wave_sum[16]
}

// One workgroup has one memory pipeline: a thread touching its 8 consecutive tiles directly (32-byte stride across a
// wave) made every lane a request of its own -- 24 stores + 16 loads per thread, 11 of the kernel's 17 us in the
// store pass alone.  So the tiles go through an LDS slab: coalesced global loads in, 16-byte LDS reads per thread, and
// offsets / order are written back to the slab and leave as coalesced stores.
constexpr int SCAN_SLAB = 1024 * SCAN_BATCH;  // tiles one workgroup scans (LDS: 32 KB)

__device__ __forceinline__ void slab_load(uint32_t* __restrict__ slab, const uint32_t* __restrict__ src, int n, int tid) {
#pragma unroll
    for (int j = 0; j < SCAN_BATCH; j++) { const int i = j * 1024 + tid; slab[i] = i < n ? src[i] : 0u; }
}
__device__ __forceinline__ void slab_read(const uint32_t* __restrict__ slab, int tid, uint32_t (&c)[SCAN_BATCH]) {
    static_assert(SCAN_BATCH == 8, "two 16-byte LDS reads per thread");
    const uint4 lo = *reinterpret_cast<const uint4*>(slab + tid * SCAN_BATCH), hi = *reinterpret_cast<const uint4*>(slab + tid * SCAN_BATCH + 4);
    c[0] = lo.x; c[1] = lo.y; c[2] = lo.z; c[3] = lo.w; c[4] = hi.x; c[5] = hi.y; c[6] = hi.z; c[7] = hi.w;
}
// the thread's exclusive running offsets, starting at `run`, over its own slots of the slab
__device__ __forceinline__ void slab_write_offsets(uint32_t* __restrict__ slab, int tid, uint32_t run, const uint32_t (&c)[SCAN_BATCH]) {
    uint32_t o[SCAN_BATCH];
#pragma unroll
    for (int j = 0; j < SCAN_BATCH; j++) { o[j] = run; run += c[j]; }
    *reinterpret_cast<uint4*>(slab + tid * SCAN_BATCH) = make_uint4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<uint4*>(slab + tid * SCAN_BATCH + 4) = make_uint4(o[4], o[5], o[6], o[7]);
}

__global__ void __launch_bounds__(1024)
k_scan_tiles(int n, const uint32_t* __restrict__ tile_count, uint32_t* __restrict__ tile_offset,
             uint32_t* __restrict__ tile_cursor, int* __restrict__ num_rendered, unsigned long long* __restrict__ host_num_rendered,
             uint32_t host_seq, uint32_t* __restrict__ tile_order, uint32_t capacity, uint32_t* __restrict__ overflow) {
    static_assert(ORDER_CELLS == 1024, "one counter cell per thread");
    __shared__ uint32_t wave_sum[17];
    __shared__ uint32_t bucket[ORDER_CELLS + 1];
    __shared__ __attribute__((aligned(16))) uint32_t slab[SCAN_SLAB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int begin = tid * SCAN_BATCH;  // n <= SCAN_SLAB
    bucket[tid] = 0u;
    slab_load(slab, tile_count, n, tid);
    __syncthreads();
    uint32_t c[SCAN_BATCH], local = 0;
    slab_read(slab, tid, c);
#pragma unroll
    for (int j = 0; j < SCAN_BATCH; j++) {
        local += c[j];
        if (c[j]) atomicAdd(&bucket[order_cell(c[j], lane)], 1u);  // empty tiles: what is left over
    }
    const uint32_t run = block_exclusive_scan(local, tid, wave_sum);
    const uint32_t total = wave_sum[16];
    slab_write_offsets(slab, tid, run, c);
    __syncthreads();  // wave_sum is reused; the offsets are in the slab
    // exclusive scan of the cells (bucket-major): bucket 0 holds the longest lists, empty tiles come last
    const uint32_t cell_base = block_exclusive_scan(bucket[tid], tid, wave_sum);
    bucket[tid] = cell_base;
    if (tid == 0) bucket[ORDER_CELLS] = wave_sum[16];
#pragma unroll
    for (int j = 0; j < SCAN_BATCH; j++) {
        const int i = j * 1024 + tid;
        if (i < n) { const uint32_t o = slab[i]; tile_offset[i] = o; tile_cursor[i] = o; }
    }
    __syncthreads();  // the cells are final; the slab is free
#pragma unroll
    for (int j = 0; j < SCAN_BATCH; j++) {
        const bool in = begin + j < n;
        const bool empty = in && c[j] == 0u;
        const uint64_t em = __ballot(empty);  // empty tiles (most of a frame) share one cell: claimed per wave
        uint32_t ebase = 0;
        if (em) {
            const int leader = __ffsll((long long)em) - 1;
            if (lane == leader) ebase = atomicAdd(&bucket[ORDER_CELLS], (uint32_t)__popcll(em));
            ebase = __shfl(ebase, leader, 64);
        }
        if (empty) slab[ebase + (uint32_t)__popcll(em & ((1ull << lane) - 1ull))] = (uint32_t)(begin + j);
        else if (in) slab[atomicAdd(&bucket[order_cell(c[j], lane)], 1u)] = (uint32_t)(begin + j);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SCAN_BATCH; j++) { const int i = j * 1024 + tid; if (i < n) tile_order[i] = slab[i]; }
    // host_num_rendered: pinned host memory, read by the host after the event recorded behind this kernel
    if (tid == 0) {
        tile_offset[n] = total; *num_rendered = (int)total;
        if (host_num_rendered) *host_num_rendered = host_size_word(host_seq, total);  // ONE 8-byte store: the host polls it
        if (overflow && total > capacity) *overflow = 1u;  // asynchronous call that outgrew its buffer (sticky, pinned host memory)
    }
}

// Records a tile's region of the backward's record buffer must hold.  The per-pixel kernel writes a face's blended pairs
// as one run padded to a multiple of HIT_GROUP records (the hit-parallel kernel takes HIT_GROUP records of ONE list entry
// per lane), so a tile with h blended pairs (counted by the forward) in a list of `len` entries needs at most
// h + (HIT_GROUP - 1) * min(len, h) records; rounded up to whole blocks of HIT_BLOCK records (the layout inside a region,
// dmr_kernels.hpp).
// (record_bound: dmr_kernels.hpp)

// ---- the same for many tiles (B * tiles > SCAN_SINGLE_MAX: several views at 1080p, 4096^2 images): one workgroup
// per 8192 tiles, three small launches (partial sums + bucket sizes | scan of the partials | offsets + order) instead
// of one workgroup streaming everything (0.5 ms for C5's 1 M tiles).
constexpr int SCAN_BLOCK_TILES = 8192;
static_assert(SCAN_SINGLE_MAX == SCAN_SLAB, "up to here one workgroup does it all (k_scan_tiles, k_scan_hits)");

// pass 1: blk_sum[block] = sum of the block's counts; bucket_count[b] += tiles of the block in order bucket b
// (ORDER == false: the scan of the backward's record regions; the counts are record_bound(tile_count, list length))
template <bool ORDER>
__global__ void __launch_bounds__(1024)
k_scan_tiles_partial(int n, const uint32_t* __restrict__ tile_count, const uint32_t* __restrict__ len_offset, uint32_t* __restrict__ blk_sum,
                     uint32_t* __restrict__ bucket_count) {
    __shared__ uint32_t wave_sum[17];
    __shared__ uint32_t bucket[ORDER_CELLS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int begin = min(n, (int)blockIdx.x * SCAN_BLOCK_TILES + tid * SCAN_BATCH), end = min(n, begin + SCAN_BATCH);
    if (ORDER) bucket[tid] = 0u;
    __syncthreads();
    uint32_t c[SCAN_BATCH], local = 0;
#pragma unroll
    for (int j = 0; j < SCAN_BATCH; j++) {
        c[j] = begin + j < end ? tile_count[begin + j] : 0u;
        if (!ORDER && begin + j < end) c[j] = record_bound(c[j], len_offset[begin + j + 1] - len_offset[begin + j]);
    }
#pragma unroll
    for (int j = 0; j < SCAN_BATCH; j++) {
        local += c[j];
        if (ORDER && begin + j < end && c[j]) atomicAdd(&bucket[order_cell(c[j], lane)], 1u);
    }
    block_exclusive_scan(local, tid, wave_sum);  // (its barriers also publish the cells)
    if (tid == 0) blk_sum[blockIdx.x] = wave_sum[16];
    if (ORDER && tid < ORDER_BUCKETS) {
        uint32_t cnt = 0;
#pragma unroll
        for (int q = 0; q < ORDER_COPIES; q++) cnt += bucket[tid * ORDER_COPIES + q];
        if (cnt) atomicAdd(&bucket_count[tid], cnt);
    }
}

// pass 2 (one workgroup): blk_sum -> exclusive scan in place, total -> R; bucket_count -> exclusive scan in place
// (bucket 0 = longest lists first, empty tiles last): the cursors pass 3 claims from
__global__ void __launch_bounds__(1024)
k_scan_tiles_blocks(int nblk, int n, uint32_t* __restrict__ blk_sum, uint32_t* __restrict__ bucket_count,
                    uint32_t* __restrict__ tile_offset, int* __restrict__ num_rendered, unsigned long long* __restrict__ host_num_rendered,
                    unsigned long long* __restrict__ total64, unsigned long long* __restrict__ host_total64, uint32_t host_seq,
                    uint32_t capacity, uint32_t* __restrict__ overflow) {
    __shared__ uint32_t wave_sum[17];
    const int tid = threadIdx.x, lane = tid & 63;
    unsigned long long carry = 0;  // the 32-bit offsets wrap harmlessly when the total does not fit (the host checks it)
    for (int i0 = 0; i0 < nblk; i0 += 1024) {
        const uint32_t v = i0 + tid < nblk ? blk_sum[i0 + tid] : 0u;
        const uint32_t ex = block_exclusive_scan(v, tid, wave_sum);
        if (i0 + tid < nblk) blk_sum[i0 + tid] = (uint32_t)carry + ex;
        carry += wave_sum[16];
        __syncthreads();  // wave_sum is reused
    }
    if (tid == 0) {
        tile_offset[n] = (uint32_t)carry;
        if (num_rendered) { *num_rendered = (int)carry; if (host_num_rendered) *host_num_rendered = host_size_word(host_seq, carry); }
        if (total64) { *total64 = carry; if (host_total64) *host_total64 = host_size_word(host_seq, carry); }
        if (overflow && carry > (unsigned long long)capacity) *overflow = 1u;
    }
    if (bucket_count && tid < 128) {  // waves 0 and 1; only wave 0 holds buckets 0..63, the empty-tile bucket follows them
        const uint32_t c = tid < ORDER_BUCKETS ? bucket_count[tid] : 0u;
        uint32_t bi = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t o = __shfl_up(bi, d, 64);
            if (lane >= d) bi += o;
        }
        if (tid < ORDER_BUCKETS) bucket_count[tid] = bi - c;
        if (tid == ORDER_BUCKETS - 1) bucket_count[ORDER_BUCKETS] = bi;
    }
}

// pass 3: offsets, cursors and the order
template <bool ORDER>
__global__ void __launch_bounds__(1024)
k_scan_tiles_final(int n, const uint32_t* __restrict__ tile_count, const uint32_t* __restrict__ len_offset, const uint32_t* __restrict__ blk_sum,
                   uint32_t* __restrict__ bucket_cursor, uint32_t* __restrict__ tile_offset, uint32_t* __restrict__ tile_cursor,
                   uint32_t* __restrict__ tile_order) {
    __shared__ uint32_t wave_sum[17];
    __shared__ uint32_t bucket[ORDER_CELLS + 1];
    const int tid = threadIdx.x, lane = tid & 63;
    const int begin = min(n, (int)blockIdx.x * SCAN_BLOCK_TILES + tid * SCAN_BATCH), end = min(n, begin + SCAN_BATCH);
    if (ORDER) { bucket[tid] = 0u; if (tid == 0) bucket[ORDER_CELLS] = 0u; }
    __syncthreads();
    uint32_t c[SCAN_BATCH], local = 0;
    int n_empty = 0;
#pragma unroll
    for (int j = 0; j < SCAN_BATCH; j++) {
        c[j] = begin + j < end ? tile_count[begin + j] : 0u;
        if (!ORDER && begin + j < end) {  // record regions: the bound, and tile_used (here: tile_cursor) cleared
            c[j] = record_bound(c[j], len_offset[begin + j + 1] - len_offset[begin + j]);
            tile_cursor[begin + j] = 0u;
        }
    }
#pragma unroll
    for (int j = 0; j < SCAN_BATCH; j++) {
        local += c[j];
        if (ORDER && begin + j < end) {
            if (c[j]) atomicAdd(&bucket[order_cell(c[j], lane)], 1u);
            else n_empty++;
        }
    }
    if (ORDER) {  // empty tiles of the block: one LDS add per wave
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) n_empty += __shfl_xor(n_empty, d, 64);
        if (lane == 0 && n_empty) atomicAdd(&bucket[ORDER_CELLS], (uint32_t)n_empty);
    }
    uint32_t run = blk_sum[blockIdx.x] + block_exclusive_scan(local, tid, wave_sum);
    // claim the block's share of every bucket of the global order; the LDS cells then hand out slots
    if (ORDER && tid < ORDER_BUCKETS) {
        uint32_t cnt = 0;
#pragma unroll
        for (int q = 0; q < ORDER_COPIES; q++) cnt += bucket[tid * ORDER_COPIES + q];
        uint32_t base = cnt ? atomicAdd(&bucket_cursor[tid], cnt) : 0u;
#pragma unroll
        for (int q = 0; q < ORDER_COPIES; q++) { const uint32_t k = bucket[tid * ORDER_COPIES + q]; bucket[tid * ORDER_COPIES + q] = base; base += k; }
    }
    if (ORDER && tid == ORDER_BUCKETS) { const uint32_t cnt = bucket[ORDER_CELLS]; bucket[ORDER_CELLS] = cnt ? atomicAdd(&bucket_cursor[ORDER_BUCKETS], cnt) : 0u; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SCAN_BATCH; j++) {
        const bool in = begin + j < end;
        if (in) { tile_offset[begin + j] = run; if (ORDER) tile_cursor[begin + j] = run; }
        run += c[j];
        if (!ORDER) continue;
        const bool empty = in && c[j] == 0u;
        const uint64_t em = __ballot(empty);
        uint32_t ebase = 0;
        if (em) {
            const int leader = __ffsll((long long)em) - 1;
            if (lane == leader) ebase = atomicAdd(&bucket[ORDER_CELLS], (uint32_t)__popcll(em));
            ebase = __shfl(ebase, leader, 64);
        }
        if (empty) tile_order[ebase + (uint32_t)__popcll(em & ((1ull << lane) - 1ull))] = (uint32_t)(begin + j);
        else if (in) tile_order[atomicAdd(&bucket[order_cell(c[j], lane)], 1u)] = (uint32_t)(begin + j);
    }
}

// exclusive scan of those bounds: every tile's region of the backward's record buffer (u32 offsets; the total is < 2^32 or
// the backward fails) and the total.  Also clears tile_used (records the per-pixel kernel really wrote, per tile).
__global__ void __launch_bounds__(1024)
k_scan_hits(int n, const uint32_t* __restrict__ tile_hits, const uint32_t* __restrict__ tile_offset, uint32_t* __restrict__ hit_offset,
            uint32_t* __restrict__ tile_used, unsigned long long* __restrict__ hit_total, unsigned long long* __restrict__ host_hit_total, uint32_t host_seq,
            uint32_t capacity, uint32_t* __restrict__ overflow) {
    __shared__ uint32_t wave_sum[17];
    __shared__ unsigned long long s_total;
    __shared__ __attribute__((aligned(16))) uint32_t slab[SCAN_SLAB];
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) s_total = 0ull;
#pragma unroll
    for (int j = 0; j < SCAN_BATCH; j++) {  // n <= SCAN_SLAB
        const int i = j * 1024 + tid;
        slab[i] = i < n ? record_bound(tile_hits[i], tile_offset[i + 1] - tile_offset[i]) : 0u;
        if (i < n) tile_used[i] = 0u;
    }
    __syncthreads();
    uint32_t c[SCAN_BATCH], local = 0;
    slab_read(slab, tid, c);
    unsigned long long wide = 0;
#pragma unroll
    for (int j = 0; j < SCAN_BATCH; j++) { local += c[j]; wide += c[j]; }
    // 64-bit total (overflow check on the host); the offsets themselves wrap harmlessly in that case
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) wide += __shfl_xor(wide, d, 64);
    const uint32_t run = block_exclusive_scan(local, tid, wave_sum);
    if (lane == 0) atomicAdd(&s_total, wide);
    slab_write_offsets(slab, tid, run, c);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SCAN_BATCH; j++) { const int i = j * 1024 + tid; if (i < n) hit_offset[i] = slab[i]; }
    if (tid == 0) {
        hit_offset[n] = wave_sum[16]; *hit_total = s_total;
        if (host_hit_total) *host_hit_total = host_size_word(host_seq, s_total);
        if (overflow && s_total > (unsigned long long)capacity) *overflow = 1u;
    }
}

// ---------------------------------------------------------------------------
// 5. per-tile sort by (depth_bits, face_id) as a kernel of its own (the tet path; the tri forward sorts its tile itself):
//    dmr_sort.hpp.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_sort_tiles(uint32_t ntiles, const uint32_t* __restrict__ tile_offset, const uint32_t* __restrict__ tile_order,
             uint64_t* __restrict__ keys, uint32_t* __restrict__ face_list, uint32_t capacity) {
    __shared__ uint64_t s_keys[SORT_LDS_KEYS];
    __shared__ uint32_t s_rank[SORT_LDS_KEYS];
    // grid-stride over tiles: most tiles of a frame are empty, a workgroup launch per tile costs more than the sort
    for (uint32_t ti = blockIdx.x; ti < ntiles; ti += gridDim.x) {
        const uint32_t tile = tile_order[ti];  // longest first
        const uint32_t begin = tile_offset[tile], end = tile_offset[tile + 1];
        // (a list beyond the buffer exists only while a size guess is being refuted: everything is redone then)
        if (end == begin || end > capacity) continue;
        __syncthreads();  // the previous tile's keys are no longer needed
        sort_tile(begin, end - begin, keys, face_list, s_keys, s_rank, threadIdx.x);
    }
}

// ---------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------
void launch_project_verts(const dmr_scene& s, float4* vproj, float* mats, uint32_t* counters, size_t ncounters,
                          hipStream_t st) {
    const int64_t n = (int64_t)s.B * s.P;
    StageScope t(DMR_STAGE_PROJECT, st);
    k_project_verts<<<dim3((unsigned)std::max<int64_t>(1, (n + 255) / 256)), dim3(256), 0, st>>>(
        s.B, s.P, s.verts, s.mv_mats, s.proj_mats, s.inv_mv_mats, s.inv_proj_mats, s.mats_transposed, s.verts_depth,
        s.W, s.H, vproj, mats, counters, (uint32_t)ncounters);
}

void launch_setup_faces(const dmr_scene& s, bool tet, const float4* vproj, int gx, int gy, int r0, int r1,
                        uint2* face_rect, float* key_depth, float* max_depth, uint32_t* tiles_touched,
                        uint32_t* tile_count, hipStream_t st) {
    const int64_t n = (int64_t)s.B * s.F;
    if (n == 0) return;
    StageScope t(DMR_STAGE_SETUP_FACES, st);
    const int fpt = bin_fpt(n);
    dim3 grid((unsigned)((n + 256 * fpt - 1) / (256 * fpt))), block(256);
#define DMR_SETUP(TET, FPT) k_setup_faces_lds<TET, FPT><<<grid, block, 0, st>>>(s.B, s.P, s.F, s.faces, vproj, gx, gy, r0, r1, \
                                face_rect, key_depth, max_depth, tiles_touched, tile_count)
    if (tet) { if (fpt == 1) DMR_SETUP(true, 1); else if (fpt == 2) DMR_SETUP(true, 2); else DMR_SETUP(true, 4); }
    else { if (fpt == 1) DMR_SETUP(false, 1); else if (fpt == 2) DMR_SETUP(false, 2); else DMR_SETUP(false, 4); }
#undef DMR_SETUP
}

void launch_scan_tiles(int ntiles, const uint32_t* tile_count, uint32_t* tile_offset, uint32_t* tile_cursor,
                       int* num_rendered, unsigned long long* host_num_rendered, uint32_t host_seq, uint32_t* tile_order, uint32_t* scan_tmp,
                       uint32_t capacity, uint32_t* overflow, hipStream_t st) {
    StageScope t(DMR_STAGE_SCAN, st);
    if (ntiles <= SCAN_SINGLE_MAX) {
        k_scan_tiles<<<dim3(1), dim3(1024), 0, st>>>(ntiles, tile_count, tile_offset, tile_cursor, num_rendered, host_num_rendered, host_seq, tile_order,
                                                    capacity, overflow);
        return;
    }
    // scan_tmp: [ORDER_BUCKETS + 1 bucket counters, zeroed by k_project_verts | partial sums per block]
    const int nblk = (ntiles + SCAN_BLOCK_TILES - 1) / SCAN_BLOCK_TILES;
    uint32_t* bucket = scan_tmp;
    uint32_t* blk_sum = scan_tmp + SCAN_TMP_BUCKETS;
    k_scan_tiles_partial<true><<<dim3(nblk), dim3(1024), 0, st>>>(ntiles, tile_count, nullptr, blk_sum, bucket);
    k_scan_tiles_blocks<<<dim3(1), dim3(1024), 0, st>>>(nblk, ntiles, blk_sum, bucket, tile_offset, num_rendered, host_num_rendered,
                                                        nullptr, nullptr, host_seq, capacity, overflow);
    k_scan_tiles_final<true><<<dim3(nblk), dim3(1024), 0, st>>>(ntiles, tile_count, nullptr, blk_sum, bucket, tile_offset, tile_cursor, tile_order);
}

size_t scan_tmp_words(int ntiles) { return SCAN_TMP_BUCKETS + (size_t)(ntiles + SCAN_BLOCK_TILES - 1) / SCAN_BLOCK_TILES + 1; }

void launch_scan_hits(int ntiles, const uint32_t* tile_hits, const uint32_t* tile_offset, uint32_t* hit_offset, uint32_t* tile_used,
                      unsigned long long* hit_total, unsigned long long* host_hit_total, uint32_t host_seq, uint32_t* scan_tmp, uint32_t capacity,
                      uint32_t* overflow, hipStream_t st) {
    StageScope t(DMR_STAGE_SCAN, st);
    if (ntiles <= SCAN_SINGLE_MAX) {
        k_scan_hits<<<dim3(1), dim3(1024), 0, st>>>(ntiles, tile_hits, tile_offset, hit_offset, tile_used, hit_total, host_hit_total, host_seq, capacity, overflow);
        return;
    }
    const int nblk = (ntiles + SCAN_BLOCK_TILES - 1) / SCAN_BLOCK_TILES;
    uint32_t* blk_sum = scan_tmp + SCAN_TMP_BUCKETS;  // the forward's partial sums are no longer needed
    k_scan_tiles_partial<false><<<dim3(nblk), dim3(1024), 0, st>>>(ntiles, tile_hits, tile_offset, blk_sum, nullptr);
    k_scan_tiles_blocks<<<dim3(1), dim3(1024), 0, st>>>(nblk, ntiles, blk_sum, nullptr, hit_offset, nullptr, nullptr, hit_total,
                                                        host_hit_total, host_seq, capacity, overflow);
    k_scan_tiles_final<false><<<dim3(nblk), dim3(1024), 0, st>>>(ntiles, tile_hits, tile_offset, blk_sum, nullptr, hit_offset, tile_used, nullptr);
}

void launch_scatter_faces(const dmr_scene& s, int gx, int gy, const uint2* face_rect, const float* key_depth,
                          const uint32_t* tiles_touched, uint32_t* tile_cursor, uint64_t* keys, uint32_t capacity,
                          unsigned long long* mask_offset_dst, unsigned long long mask_offset, unsigned long long mask_first, hipStream_t st) {
    const int64_t n = (int64_t)s.B * s.F;
    if (n == 0) return;
    StageScope t(DMR_STAGE_SCATTER, st);
    const int fpt = bin_fpt(n);
    const dim3 grid((unsigned)((n + 256 * fpt - 1) / (256 * fpt))), block(256);
#define DMR_SCATTER(FPT) k_scatter_faces_lds<FPT><<<grid, block, 0, st>>>(s.B, s.F, gx, gy, face_rect, key_depth, tiles_touched, \
                                                                         tile_cursor, keys, capacity, mask_offset_dst, mask_offset, mask_first)
    if (fpt == 1) DMR_SCATTER(1); else if (fpt == 2) DMR_SCATTER(2); else DMR_SCATTER(4);
#undef DMR_SCATTER
}

void launch_sort_tiles(int ntiles, const uint32_t* tile_offset, const uint32_t* tile_order, uint64_t* keys,
                       uint32_t* face_list, uint32_t capacity, hipStream_t st) {
    if (ntiles == 0) return;
    StageScope t(DMR_STAGE_SORT, st);
    k_sort_tiles<<<dim3((unsigned)std::min(ntiles, 256 * 64)), dim3(256), 0, st>>>((uint32_t)ntiles, tile_offset, tile_order, keys, face_list, capacity);
}

}  // namespace dmr
