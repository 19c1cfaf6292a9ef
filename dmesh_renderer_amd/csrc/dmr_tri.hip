// dmr_tri.hip -- tri renderer: front-to-back alpha compositing, forward + backward (gfx950).
//
// Replaces TRI_FORWARD::renderCUDA (cuda_rasterizer/forward.cu:257-489) and
// TRI_BACKWARD::renderCUDA (cuda_rasterizer/backward.cu:9-421).
//
// One 256-thread workgroup (4 wave64) per 16x16 tile, tiles taken longest list first; wave w owns the 8x8 pixel
// quadrant (w & 1, w >> 1), lane l the pixel (l & 7, l >> 3) inside it.  The tile's depth-sorted face list is
// consumed in 128-face chunks staged through LDS (waves 0-1 build the coverage records, waves 2-3 the shading
// records).  Per chunk:
//
//   A. coverage  -- face-parallel: 256 / CHUNK threads rasterise one staged face each over the rows of its
//      tile-local pixel box, stepping the three fixed-point edge functions incrementally, and OR the face's bit
//      into the covered pixels' mask words in LDS (ds_or_b32 runs at full rate).  All the per-face work of the
//      reference's in_tri (float->fixed conversion, winding swap, edge deltas, top-left bias) was done once when
//      the face was staged.
//   B. shading   -- every lane walks the set bits of ITS OWN mask in list order, so all 64 lanes do useful
//      blending work on (generally different) faces at once instead of a few lanes per face; face records are
//      gathered from LDS with per-lane addresses (112-byte stride = odd number of 16-byte slots, conflict-light
//      for ds_read_b128).
//
// The pixel's result is the same sequence of blends as the reference's loop.  Rays are recomputed per pixel (not
// stored).  Up to 8 192 tiles the forward's workgroup first sorts its tile's list (dmr_sort.hpp).  The backward is two
// kernels: k_tri_backward_pix (same layout, chunks from the back: the per-pixel sequential part, one 16-byte record per
// blended pair, face-major; it also lays out the tiles' record regions) and k_tri_backward_hits (one lane per group of
// four records of a list entry: 21 sums -- the vertex-position gradient as ray moments --, segmented DPP scan, workgroup
// LDS table, packed atomics).  DESIGN.md section 5 has the measurements behind every step.
#include <algorithm>
#include <cstdlib>

#include "dmr_kernels.hpp"
#include "dmr_sort.hpp"

namespace dmr {

constexpr int FWD_CHUNK = 128;
#ifndef DMR_BWD_CHUNK
#define DMR_BWD_CHUNK 128
#endif
constexpr int BWD_CHUNK = DMR_BWD_CHUNK;

// Phase stamps (ablation build only, DMR_ABLATE bit 4096; scripts/phase_times.py reads them): lane 0 of every wave of the
// first PH_BLOCKS workgroups (= the longest tile lists, tile_order) writes s_memtime at the phase boundaries of its first
// PH_CHUNKS chunks / rounds, and wave 0 the 100 MHz s_memrealtime at both ends of the workgroup.  This is how "where does a
// tile's serial chain spend its time" is measured instead of guessed.
#ifdef DMR_ABLATION
constexpr int PH_KERNELS = 3, PH_BLOCKS = 3072, PH_CHUNKS = 6, PH_STAMPS = 10;
__device__ unsigned long long g_phase[PH_KERNELS][PH_BLOCKS][4][PH_CHUNKS][PH_STAMPS];
__device__ unsigned long long g_phase_rt[PH_KERNELS][PH_BLOCKS][2];
#define DMR_STAMP(p, kern, chunk, s)                                                                              \
    do {                                                                                                          \
        if (DMR_DBG(p, 4096) && (threadIdx.x & 63) == 0 && blockIdx.x < PH_BLOCKS && (chunk) < (uint32_t)PH_CHUNKS) \
            g_phase[kern][blockIdx.x][threadIdx.x >> 6][chunk][s] = __builtin_amdgcn_s_memtime();                 \
    } while (0)
#define DMR_STAMP_RT(p, kern, e)                                                                                  \
    do {                                                                                                          \
        if (DMR_DBG(p, 4096) && threadIdx.x == 0 && blockIdx.x < PH_BLOCKS)                                        \
            g_phase_rt[kern][blockIdx.x][e] = __builtin_amdgcn_s_memrealtime();                                   \
    } while (0)
#else
#define DMR_STAMP(p, kern, chunk, s) do {} while (0)
#define DMR_STAMP_RT(p, kern, e) do {} while (0)
#endif

// s_i(x, y) = s0[i] + bx[i] * (x - x0) + by[i] * (y - y0) (mod 2^32), inside iff all three < 0.
// A zero-area face (in_tri returns false, auxiliary.h:201-202) and the padding entries of a
// partially filled 32-face word are stored as the all-zero record, which covers nothing.
// flags: bit 16 = the record covers something at all (0 for zero-area faces and padding); bits 0-15 =
// the tile-local pixel box x0 | x1 << 4 | y0 << 8 | y1 << 12 that can contain covered pixel centres
// (the whole tile when int32 wrap-around, Q7, could make the edge functions claim pixels outside it).
struct alignas(16) CovRec {
    int32_t s0[3]; int32_t flags;
    int32_t bx[3]; int32_t pad0;
    int32_t by[3]; int32_t pad1;
};
constexpr int COV_VALID = 0x10000;
static_assert(sizeof(CovRec) == 48, "CovRec");

// T = ray_o - p0, E1 = p1 - p0, E2 = p2 - p0, Q = cross(T, E1): the pixel-independent part of
// ray_tri_intersection (auxiliary.h:267-272; the ray origin is the same for a whole view).
struct alignas(16) ShadeRec {
    float T[3], E1[3], E2[3], Q[3];
    float c0[3], c1[3], c2[3];
    float d0, d1, d2, opacity, intense;
    float pad[2];
};
static_assert(sizeof(ShadeRec) == 112, "ShadeRec");

struct TriParams {
    int B, P, F, W, H, gx, gy, r0, r1;
#ifdef DMR_ABLATION
    int dbg;  // DMR_ABLATE bits (ablation build only)
#endif
    uint32_t list_capacity;  // entries face_list holds (< R only while a size guess is being refuted)
    const float* verts; const int* faces; const float* verts_color; const float* faces_opacity;
    const float* inv_mv; const float* inv_proj; const float* faces_intense; const float* bg;
    const float4* vproj; const uint32_t* tile_offset; const uint32_t* face_list;
    unsigned long long* keys;  // (depth_bits << 32 | face) of every list entry, unsorted: the forward sorts its tile's
    float* final_T; float* final_prev_T; uint32_t* n_contrib;
    uint32_t* tile_hits; uint32_t* tile_bound; const uint32_t* hit_offset; uint32_t* tile_used; const uint32_t* tile_order;
    const unsigned long long* mask_offset;  // coverage masks: byte offset behind face_list (TriImageState)
};

// The tile's coverage masks, pixel-major, one slot of 256 x 16 bytes per chunk (layout: TriImageState, dmr_kernels.hpp): chunk
// 0 in the slot of the tile's position in tile_order, the later chunks behind the list offset.  The forward writes them
// right after rasterising, the per-pixel backward reads them back instead of staging the coverage records and
// rasterising the chunk a second time (that was 13 k of its 25 k cycles per chunk, profiles/r02/phase_times_c4.txt).
struct ChunkMasks {
    uint4* first; uint4* rest;
    __device__ __forceinline__ uint4* at(uint32_t chunk) const { return chunk == 0u ? first : rest + (size_t)chunk * TILE_PIX; }
};
__device__ __forceinline__ ChunkMasks chunk_masks(const TriParams& p, uint32_t order_pos, uint32_t begin) {
    uint4* base = reinterpret_cast<uint4*>(const_cast<char*>(reinterpret_cast<const char*>(p.face_list)) + p.mask_offset[0]);
    // (chunk c >= 1: slot first_slots + begin / 128 + c - 1; first_slots >= 1 whenever a list has an entry)
    return {base + (size_t)order_pos * TILE_PIX, base + ((size_t)p.mask_offset[1] + (size_t)(begin / (uint32_t)MASK_CHUNK) - 1u) * TILE_PIX};
}

// Staging a list entry is a chain of dependent gathers: face_list -> faces (+ opacity, intensity) -> 3 x
// (projected vertex, position, colour).  While the workgroup waits for it nothing else runs, so the first two
// levels are software-pipelined: the ids of chunk c + 1 are loaded while chunk c is being composited (6 VGPRs),
// and staging a chunk only exposes the last level.
struct FaceIds { int face, v0, v1, v2; float opacity, intense; };

__device__ __forceinline__ FaceIds load_face_ids(const TriParams& p, int b, int face) {
    FaceIds f;
    f.face = face;
    if (face >= 0) {
        f.v0 = p.faces[3 * face]; f.v1 = p.faces[3 * face + 1]; f.v2 = p.faces[3 * face + 2];
        f.opacity = p.faces_opacity[face];
        f.intense = p.faces_intense[(int64_t)b * p.F + face];
    } else { f.v0 = f.v1 = f.v2 = 0; f.opacity = 0.f; f.intense = 0.f; }
    return f;
}

__device__ __forceinline__ void stage_null(CovRec& cov) {
    int4* q = reinterpret_cast<int4*>(&cov);
    q[0] = make_int4(0, 0, 0, 0); q[1] = make_int4(0, 0, 0, 0); q[2] = make_int4(0, 0, 0, 0);
}

// The two halves of staging a face: with 128-face chunks and 256 threads, waves 0-1 build the coverage records while
// waves 2-3 build the shading records (one thread did both before, three waves waiting for the fourth).
__device__ __forceinline__ void stage_cov(const TriParams& p, int b, const FaceIds& f, int x0, int y0, CovRec& cov) {
    const float4 a0 = p.vproj[(int64_t)b * p.P + f.v0];
    const float4 a1 = p.vproj[(int64_t)b * p.P + f.v1];
    const float4 a2 = p.vproj[(int64_t)b * p.P + f.v2];
    EdgeSetup e = edge_setup({a0.x, a0.y}, {a1.x, a1.y}, {a2.x, a2.y}, x0, y0);
#pragma unroll
    for (int i = 0; i < 3; i++) { cov.s0[i] = e.s0[i]; cov.bx[i] = e.bx[i]; cov.by[i] = e.by[i]; }
    const bool some = e.ok && e.x0 <= e.x1 && e.y0 <= e.y1;
    cov.flags = some ? (COV_VALID | e.x0 | (e.x1 << 4) | (e.y0 << 8) | (e.y1 << 12)) : 0;
    cov.pad0 = 0; cov.pad1 = 0;
}

__device__ __forceinline__ void stage_shade(const TriParams& p, int b, const FaceIds& f, V3 ray_o, ShadeRec& sh) {
    const int v0 = f.v0, v1 = f.v1, v2 = f.v2;
    const V3 p0 = load_v3(p.verts, v0), p1 = load_v3(p.verts, v1), p2 = load_v3(p.verts, v2);
    const V3 c0 = load_v3(p.verts_color, v0), c1 = load_v3(p.verts_color, v1), c2 = load_v3(p.verts_color, v2);
    const float d0 = p.vproj[(int64_t)b * p.P + v0].w, d1 = p.vproj[(int64_t)b * p.P + v1].w, d2 = p.vproj[(int64_t)b * p.P + v2].w;
    const V3 T = ray_o - p0, E1 = p1 - p0, E2 = p2 - p0;
    const V3 Q = cross(T, E1);
    sh.T[0] = T.x; sh.T[1] = T.y; sh.T[2] = T.z;
    sh.E1[0] = E1.x; sh.E1[1] = E1.y; sh.E1[2] = E1.z;
    sh.E2[0] = E2.x; sh.E2[1] = E2.y; sh.E2[2] = E2.z;
    sh.Q[0] = Q.x; sh.Q[1] = Q.y; sh.Q[2] = Q.z;
    sh.c0[0] = c0.x; sh.c0[1] = c0.y; sh.c0[2] = c0.z;
    sh.c1[0] = c1.x; sh.c1[1] = c1.y; sh.c1[2] = c1.z;
    sh.c2[0] = c2.x; sh.c2[1] = c2.y; sh.c2[2] = c2.z;
    sh.d0 = d0; sh.d1 = d1; sh.d2 = d2;
    sh.opacity = f.opacity;
    sh.intense = f.intense;
}

// The shading record in two halves, for the per-pixel backward (no coverage records there: two threads per face):
// geometry (positions -> T, E1, E2, Q) and attributes (colours, depths, opacity, intensity).
__device__ __forceinline__ void stage_shade_geom(const TriParams& p, const FaceIds& f, V3 ray_o, ShadeRec& sh) {
    const V3 p0 = load_v3(p.verts, f.v0), p1 = load_v3(p.verts, f.v1), p2 = load_v3(p.verts, f.v2);
    const V3 T = ray_o - p0, E1 = p1 - p0, E2 = p2 - p0;
    const V3 Q = cross(T, E1);
    sh.T[0] = T.x; sh.T[1] = T.y; sh.T[2] = T.z;
    sh.E1[0] = E1.x; sh.E1[1] = E1.y; sh.E1[2] = E1.z;
    sh.E2[0] = E2.x; sh.E2[1] = E2.y; sh.E2[2] = E2.z;
    sh.Q[0] = Q.x; sh.Q[1] = Q.y; sh.Q[2] = Q.z;
}
__device__ __forceinline__ void stage_shade_attr(const TriParams& p, int b, const FaceIds& f, ShadeRec& sh) {
    const V3 c0 = load_v3(p.verts_color, f.v0), c1 = load_v3(p.verts_color, f.v1), c2 = load_v3(p.verts_color, f.v2);
    const float d0 = p.vproj[(int64_t)b * p.P + f.v0].w, d1 = p.vproj[(int64_t)b * p.P + f.v1].w, d2 = p.vproj[(int64_t)b * p.P + f.v2].w;
    sh.c0[0] = c0.x; sh.c0[1] = c0.y; sh.c0[2] = c0.z;
    sh.c1[0] = c1.x; sh.c1[1] = c1.y; sh.c1[2] = c1.z;
    sh.c2[0] = c2.x; sh.c2[1] = c2.y; sh.c2[2] = c2.z;
    sh.d0 = d0; sh.d1 = d1; sh.d2 = d2;
    sh.opacity = f.opacity;
    sh.intense = f.intense;
}

// thread tid stages face (tid mod CHUNK) of the chunk: its coverage record if tid < CHUNK, else its shading record
template <int CHUNK>
__device__ __forceinline__ void stage_chunk(const TriParams& p, int b, const FaceIds& ids, int n, int tid, int x0, int y0,
                                            V3 ray_o, CovRec* __restrict__ s_cov, ShadeRec* __restrict__ s_shade) {
    static_assert(CHUNK == 128, "256 threads = 128 coverage + 128 shading records");
    const int j = tid & (CHUNK - 1);
    if (tid < CHUNK) {
        if (j < n) stage_cov(p, b, ids, x0, y0, s_cov[j]);
        else stage_null(s_cov[j]);
    } else if (j < n) {
        stage_shade(p, b, ids, ray_o, s_shade[j]);
    }
}


// Phase A, face-parallel.  256 / CHUNK threads rasterise one staged face each (interleaved rows of its pixel
// box): the three fixed-point edge functions are stepped incrementally (+bx per pixel) and every covered
// pixel centre gets the face's bit OR-ed into that pixel's mask word in LDS.  ds_or_b32 runs at full rate on
// gfx950 (scripts/micro/lds_atomics.hip), so a 6x6-pixel triangle costs ~150 lane-instructions instead of
// the ~1900 lane-slots of testing all 256 pixels of the tile against it (the earlier pixel-parallel
// versions: every wave reading every record was LDS-issue bound, 157 of 232 us at C4; one wave per
// 32-face block for all four quadrants with quadrant culling, 45-65 us).
template <int CHUNK>
__device__ __forceinline__ void rasterize_faces(const CovRec* __restrict__ cov, int n, int tid,
                                                uint32_t (*__restrict__ pm)[CHUNK / 32]) {
    constexpr int TPF = 256 / CHUNK;  // threads per face
    const int j = tid / TPF, sub = tid % TPF;
    if (j >= n) return;
    const CovRec& c = cov[j];
    const int fl = c.flags;
    if (!(fl & COV_VALID)) return;
    const int x0 = fl & 15, x1 = (fl >> 4) & 15, y0 = (fl >> 8) & 15, y1 = (fl >> 12) & 15;
    const uint32_t bit = 1u << (j & 31);
    const int word = j >> 5;
    const uint32_t bx0 = (uint32_t)c.bx[0], bx1 = (uint32_t)c.bx[1], bx2 = (uint32_t)c.bx[2];
    for (int y = y0 + sub; y <= y1; y += TPF) {
        uint32_t e0 = (uint32_t)c.s0[0] + (uint32_t)c.by[0] * (uint32_t)y + bx0 * (uint32_t)x0;
        uint32_t e1 = (uint32_t)c.s0[1] + (uint32_t)c.by[1] * (uint32_t)y + bx1 * (uint32_t)x0;
        uint32_t e2 = (uint32_t)c.s0[2] + (uint32_t)c.by[2] * (uint32_t)y + bx2 * (uint32_t)x0;
        for (int x = x0; x <= x1; x++) {
            if ((int32_t)(e0 & e1 & e2) < 0) atomicOr(&pm[y * TILE + x][word], bit);
            e0 += bx0; e1 += bx1; e2 += bx2;
        }
    }
}

// (launch_bounds(256, 6): six waves per SIMD is what the kernel's 77 registers and 24.6 KB of LDS give; (256, 4) compiles to the
// same speed -- 76.5-77.3 against 76.9-77.7 us at C4 on one box, profiles/r03/variants_launch_bounds_c4.txt --, (256, 7) = 72
// registers: 86.9)
#ifndef DMR_FWD_WAVES
#define DMR_FWD_WAVES 6
#endif
#ifndef DMR_PIX_WAVES
#define DMR_PIX_WAVES 6
#endif
// The workgroup first SORTS its tile's list (dmr_sort.hpp; the LDS of the sort is the LDS of the compositing loop): as a
// kernel of its own the sort took 22-24 us at C4, nearly all of it the serial chain of the longest tile on an otherwise
// idle chip (an eighth of the tiles took 19 us, profiles/r02/shard_kernel_sums_c4.json); here that chain runs beside the
// other workgroups' compositing.  The sorted ids go to face_list for this kernel's own staging (a block's global writes
// are visible to it behind a barrier) and for the backward.
// (SORT == false: the lists were sorted by k_sort_tiles.  Frames of many tiles keep the chip busy in either kernel and the
// separate sort is the cheaper one there -- C5, 262 144 tiles: 0.39 + 2.01 ms against 2.48 ms fused; dmr_api.hip decides.)
template <int CHUNK, bool SORT>
__global__ void __launch_bounds__(256, DMR_FWD_WAVES)
k_tri_forward(TriParams p, float* __restrict__ out_color, float* __restrict__ out_depth) {
    constexpr int WORDS = CHUNK / 32;
    static_assert(WORDS == 4, "one 32-face block per wave");
    constexpr int COV_BYTES = CHUNK * (int)sizeof(CovRec), SHADE_BYTES = CHUNK * (int)sizeof(ShadeRec);
    constexpr int FWD_BYTES = COV_BYTES + SHADE_BYTES + TILE_PIX * WORDS * (int)sizeof(uint32_t);
    __shared__ __attribute__((aligned(16))) unsigned char s_mem[FWD_BYTES > SORT_LDS_BYTES ? FWD_BYTES : SORT_LDS_BYTES];
    CovRec* const s_cov = reinterpret_cast<CovRec*>(s_mem);
    ShadeRec* const s_shade = reinterpret_cast<ShadeRec*>(s_mem + COV_BYTES);
    // [tile-local pixel y*16+x][32-face word]: coverage bits of the chunk
    uint32_t (*const s_pm)[WORDS] = reinterpret_cast<uint32_t (*)[WORDS]>(s_mem + COV_BYTES + SHADE_BYTES);
    __shared__ uint32_t s_live[2];
    __shared__ uint32_t s_hits[4];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const unsigned long long mask_first_slots = p.mask_offset[1];  // (requested with the tile's id, not behind its list range)
    // tiles are taken longest list first (tile_order, k_scan_tiles); rows outside this shard's band are skipped
    const int tile = (int)p.tile_order[blockIdx.x];
    const int tx = tile % p.gx, ty = (tile / p.gx) % p.gy, b = tile / (p.gx * p.gy);
    if (ty < p.r0 || ty >= p.r1) return;  // uniform
    const int lx = (wave & 1) * 8 + (lane & 7), ly = (wave >> 1) * 8 + (lane >> 3);
    const int px = tx * TILE + lx, py = ty * TILE + ly;
    const bool inside = px < p.W && py < p.H;
    const int64_t HW = (int64_t)p.H * p.W;
    const int64_t pix_id = (int64_t)p.W * py + px;

    // A list that does not fit the binning buffer (only while a size guess is being refuted; everything is redone
    // then) was neither completely scattered nor sorted: its entries are not face ids.  Such a tile renders as empty.
    // (The same for a tile whose position in tile_order has no first-chunk mask slot: more busy tiles than list capacity.)
    uint32_t begin = p.tile_offset[tile], end = p.tile_offset[tile + 1];
    if (end > p.list_capacity) begin = end = 0u;
    if ((unsigned long long)blockIdx.x >= mask_first_slots) begin = end = 0u;
    if (begin == end) {  // an empty tile (most of a frame: 5 244 of C4's 8 160) is background: no rays, no LDS, no barriers
        if (inside) {
            const int64_t bpix = (int64_t)b * HW + pix_id;
            p.final_prev_T[bpix] = 1.0f; p.final_T[bpix] = 1.0f; p.n_contrib[bpix] = 0u;
            out_color[((int64_t)b * 3 + 0) * HW + pix_id] = 0.f + 1.0f * p.bg[0];
            out_color[((int64_t)b * 3 + 1) * HW + pix_id] = 0.f + 1.0f * p.bg[1];
            out_color[((int64_t)b * 3 + 2) * HW + pix_id] = 0.f + 1.0f * p.bg[2];
            out_depth[bpix] = 0.f + 1.0f * 1.0f;
        }
        return;  // uniform
    }

    if (SORT && !DMR_DBG(p, 262144)) {  // (ablation build, bit 262144: unsorted lists -- timing and instruction counts only)
        sort_tile(begin, end - begin, reinterpret_cast<uint64_t*>(p.keys), const_cast<uint32_t*>(p.face_list),
                  reinterpret_cast<uint64_t*>(s_mem), reinterpret_cast<uint32_t*>(s_mem + SORT_LDS_KEYS * sizeof(uint64_t)), (uint32_t)tid);
        __syncthreads();  // face_list[begin, end) is sorted and visible to this workgroup; the LDS is free
    }

    V3 ro = {0, 0, 0}, rd = {0, 0, 0};
    if (inside && !DMR_DBG(p, 32)) pixel_ray<false>(p.inv_mv + 16 * b, p.inv_proj + 16 * b, px, py, p.W, p.H, ro, rd);
    const V3 view_o = {p.inv_mv[16 * b + 12], p.inv_mv[16 * b + 13], p.inv_mv[16 * b + 14]};

    float T = 1.0f, pT = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f;
    uint32_t last_contributor = 0, n_hits = 0, n_skipped = 0;
    bool done = !inside;

    // staging pipeline (threads < CHUNK): ids of the current chunk, face id of the next one
    const int sj = tid & (CHUNK - 1);  // the chunk face this thread stages (see stage_chunk)
    FaceIds ids = load_face_ids(p, b, begin + sj < end ? (int)p.face_list[begin + sj] : -1);
    int face_next = begin + CHUNK + sj < end ? (int)p.face_list[begin + CHUNK + sj] : -1;

    DMR_STAMP_RT(p, 0, 0);
    static_assert(CHUNK == MASK_CHUNK, "one mask slot per chunk");
    const ChunkMasks masks = chunk_masks(p, blockIdx.x, begin);
    AllDone all_done;
    all_done.init(s_live);
    for (uint32_t base = begin; base < end; base += CHUNK) {
        const uint32_t ph = (base - begin) / CHUNK;  // (phase stamps, ablation build)
        DMR_STAMP(p, 0, ph, 0);
        if (all_done.barrier(done)) break;  // also fences LDS reuse
        DMR_STAMP(p, 0, ph, 1);
        const int n = (int)min((uint32_t)CHUNK, end - base);
        if (!DMR_DBG(p, 64)) stage_chunk<CHUNK>(p, b, ids, n, tid, tx * TILE, ty * TILE, view_o, s_cov, s_shade);
        ids = load_face_ids(p, b, face_next);  // in flight while this chunk is composited
        face_next = base + 2 * CHUNK + sj < end ? (int)p.face_list[base + 2 * CHUNK + sj] : -1;
        *reinterpret_cast<uint4*>(&s_pm[tid][0]) = make_uint4(0u, 0u, 0u, 0u);
        DMR_STAMP(p, 0, ph, 2);
        __syncthreads();
        DMR_STAMP(p, 0, ph, 3);
        if (!DMR_DBG(p, 16)) rasterize_faces<CHUNK>(s_cov, n, tid, s_pm);  // A
        DMR_STAMP(p, 0, ph, 4);
        __syncthreads();
        DMR_STAMP(p, 0, ph, 5);
        uint32_t m[WORDS];
        {
            const uint4 mm = *reinterpret_cast<const uint4*>(&s_pm[ly * TILE + lx][0]);
            m[0] = mm.x; m[1] = mm.y; m[2] = mm.z; m[3] = mm.w;
            masks.at(ph)[ly * TILE + lx] = mm;  // kept for the backward (every pixel: it applies its own bound)
        }
#pragma unroll
        for (int w = 0; w < WORDS; w++) if (done || DMR_DBG(p, 8)) m[w] = 0;
        if (__all(done)) continue;  // wave-uniform
        // (Unpacking a pixel's bits into a byte list of face numbers in its own 16-byte row of s_pm first, so that the walk reads a
        // byte per pair instead of picking "the lowest set bit of four words" -- ~22 instructions of the 4.5-cycle kind -- was
        // measured: 84 us against 77-79 for this loop, profiles/r03/dead_ends.md.)
        while (true) {
            int w = -1; uint32_t mw = 0;
#pragma unroll
            for (int q = WORDS - 1; q >= 0; q--) if (m[q]) { w = q; mw = m[q]; }
            if (w < 0) break;
            const int bit = __ffs(mw) - 1;
            const uint32_t clr = mw & (mw - 1);
#pragma unroll
            for (int q = 0; q < WORDS; q++) if (q == w) m[q] = clr;
            const int k = 32 * w + bit;
            const ShadeRec& r = s_shade[k];

            const V3 E1 = {r.E1[0], r.E1[1], r.E1[2]}, E2 = {r.E2[0], r.E2[1], r.E2[2]};
            const V3 Tv = {r.T[0], r.T[1], r.T[2]}, Q = {r.Q[0], r.Q[1], r.Q[2]};
            const V3 Pv = cross(rd, E2);
            const float denom = dot(Pv, E1);
            if (denom == 0.0f) { n_skipped++; continue; }  // "edge case": counted, not blended (forward.cu:429-430)
            const float inv_denom = 1.0f / denom;
            const float iu = dot(Pv, Tv) * inv_denom;
            const float iv = dot(Q, rd) * inv_denom;
            float iuc, ivc; int code;
            clamp_bary_uv(iu, iv, iuc, ivc, code);
            const float i0 = 1 - iuc - ivc, i1 = iuc, i2 = ivc;
            float iC0 = i0 * r.c0[0] + i1 * r.c1[0] + i2 * r.c2[0];
            float iC1 = i0 * r.c0[1] + i1 * r.c1[1] + i2 * r.c2[1];
            float iC2 = i0 * r.c0[2] + i1 * r.c1[2] + i2 * r.c2[2];
            iC0 = iC0 * r.intense; iC1 = iC1 * r.intense; iC2 = iC2 * r.intense;
            const float iD = i0 * r.d0 + i1 * r.d1 + i2 * r.d2;
            const float alpha = r.opacity;
            const float test_T = T * (1 - alpha);
            C0 += iC0 * alpha * T; C1 += iC1 * alpha * T; C2 += iC2 * alpha * T;
            D += iD * alpha * T;
            pT = T; T = test_T;
            last_contributor = (base - begin) + (uint32_t)k + 1u;
            n_hits += 1u + n_skipped; n_skipped = 0u;  // the backward lists every covered pair below n_contrib, skipped ones too
            if (T < T_EPS) { done = true; break; }  // blend first, test after (Q9)
        }
        DMR_STAMP(p, 0, ph, 6);
    }
    DMR_STAMP_RT(p, 0, 1);

    {   // blended (pixel, face) pairs of the tile (sizes the backward's hit-record buffer) and the bound on its records: one
        // workgroup per tile, so both are plain stores (empty tiles keep the zeros of k_project_verts)
#pragma unroll
        for (int dlt = 32; dlt > 0; dlt >>= 1) n_hits += __shfl_xor(n_hits, dlt, 64);
        if (lane == 0) s_hits[wave] = n_hits;
        __syncthreads();
        if (tid == 0) {
            const uint32_t h = s_hits[0] + s_hits[1] + s_hits[2] + s_hits[3];
            const uint32_t bound = record_bound(h, end - begin);
            p.tile_hits[tile] = h;
            p.tile_bound[tile] = bound;
        }
    }
    if (inside && !DMR_DBG(p, 128)) {
        const int64_t bpix = (int64_t)b * HW + pix_id;
        p.final_prev_T[bpix] = pT;
        p.final_T[bpix] = T;
        p.n_contrib[bpix] = last_contributor;
        out_color[((int64_t)b * 3 + 0) * HW + pix_id] = C0 + T * p.bg[0];
        out_color[((int64_t)b * 3 + 1) * HW + pix_id] = C1 + T * p.bg[1];
        out_color[((int64_t)b * 3 + 2) * HW + pix_id] = C2 + T * p.bg[2];
        out_depth[bpix] = D + T * 1.0f;
    }
}

// ---------------------------------------------------------------------------
// backward: shared helpers
// ---------------------------------------------------------------------------
constexpr int NACC = 23;  // 9 dverts, 9 dvcolor, 3 dvdepth, dopacity, dintense
// What a lane of k_tri_backward_hits sums per list entry and the segmented scan carries: the vertex-position gradient as
// the 7 ray moments (B, A2, S3) it is linear in (see there), then 9 dvcolor, 3 dvdepth, dopacity, dintense.
constexpr int NSCAN = 21;

// DPP lane move (VALU, no LDS traffic): a lane whose source is outside its 16-lane row keeps `old`.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xF, false);
}
// 1-ulp reciprocal (v_rcp_f32): gradients are checked to 1e-4, the forward keeps IEEE division
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
constexpr int DPP_ROW_SHR = 0x110;      // + n, n = 1..15

// One level of the segmented inclusive scan over the 21 sums: g += m * g[source lane], m = 1.0 where the
// source lane carries the same list entry, else 0.0.  One v_fmac_f32_dpp per value (scripts/micro/valu_rates.hip:
// 4.8 SIMD cycles, against 9.1 for the v_cndmask_b32_dpp + v_add_f32 pair this replaces; v_add_f32_dpp under an
// EXEC mask cannot be used because DPP does not read EXEC-disabled source lanes).  Lanes without a valid source
// read 0 (bound_ctrl:0); rows excluded by row_mask are not written.  A non-finite value (quirk Q12) would turn
// 0 * inf into NaN for the neighbouring segment: such lanes are taken out of the scan by the caller.
#define DMR_SEG1(N, DPP) "v_fmac_f32_dpp %[g" #N "], %[g" #N "], %[m] " DPP "\n\t"
#define DMR_SEG_LEVEL(DPP)                                                                                     \
    asm volatile("s_nop 1\n\t"                                                                                 \
                 DMR_SEG1(0, DPP) DMR_SEG1(1, DPP) DMR_SEG1(2, DPP) DMR_SEG1(3, DPP) DMR_SEG1(4, DPP) DMR_SEG1(5, DPP)  \
                 DMR_SEG1(6, DPP) DMR_SEG1(7, DPP) DMR_SEG1(8, DPP) DMR_SEG1(9, DPP) DMR_SEG1(10, DPP) DMR_SEG1(11, DPP) \
                 DMR_SEG1(12, DPP) DMR_SEG1(13, DPP) DMR_SEG1(14, DPP) DMR_SEG1(15, DPP) DMR_SEG1(16, DPP)          \
                 DMR_SEG1(17, DPP) DMR_SEG1(18, DPP) DMR_SEG1(19, DPP) DMR_SEG1(20, DPP)                           \
                 : [g0] "+v"(g[0]), [g1] "+v"(g[1]), [g2] "+v"(g[2]), [g3] "+v"(g[3]), [g4] "+v"(g[4]),              \
                   [g5] "+v"(g[5]), [g6] "+v"(g[6]), [g7] "+v"(g[7]), [g8] "+v"(g[8]), [g9] "+v"(g[9]),             \
                   [g10] "+v"(g[10]), [g11] "+v"(g[11]), [g12] "+v"(g[12]), [g13] "+v"(g[13]), [g14] "+v"(g[14]),   \
                   [g15] "+v"(g[15]), [g16] "+v"(g[16]), [g17] "+v"(g[17]), [g18] "+v"(g[18]), [g19] "+v"(g[19]),   \
                   [g20] "+v"(g[20])                                                                            \
                 : [m] "v"(m))

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void seg_scan_level(int k, float (&g)[NSCAN]) {
    static_assert(NSCAN == 21, "the asm lists 21 registers");
    const int ko = dpp_i<CTRL, ROW_MASK>((int)0x80000000, k);
    const float m = (ko == k) ? 1.0f : 0.0f;  // keys are list entries (< 2^31) or negative per-lane ids: never 0x80000000
    if (CTRL == DPP_ROW_SHR + 1) DMR_SEG_LEVEL("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0");
    else if (CTRL == DPP_ROW_SHR + 2) DMR_SEG_LEVEL("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0");
    else if (CTRL == DPP_ROW_SHR + 4) DMR_SEG_LEVEL("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0");
    else DMR_SEG_LEVEL("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0");
}

// ---------------------------------------------------------------------------
// backward, kernel 1 of 2: k_tri_backward_pix -- the per-pixel sequential part.
//
// Per chunk of 128 list entries (the forward's chunks, walked from the back of the tile list):
//   A. coverage: the pixel's 128-bit mask of the chunk comes from the forward (chunk_masks: what its rasterisation found),
//      cut at the pixel's n_contrib (backward.cu:192-194); every pixel adds one to the counter of each face it blended
//      (ds_add_u32).  Meanwhile the chunk's shading records are staged, two threads per face.  (Up to round 2 this kernel
//      staged the coverage records and rasterised every chunk a second time: 13 k of its 25 k cycles per chunk.)
//   S. one wave scans the 128 counts: face-major slot ranges inside the tile's record region (the region itself
//      starts at the scan of the per-tile hit counts the forward accumulated).
//   B. every pixel walks its mask from the back: recover T (Q10), the running accum_rec terms and dL/dalpha
//      (backward.cu:244-308), claim a slot of the face with a returning ds_add_u32 and write the 16-byte
//      HitRecord straight to HBM.  Within a face the records are in claim order, which kernel 2 does not care about.
// Everything that needs 23 accumulators per face happens in kernel 2, so this kernel keeps a small
// register/LDS footprint (79 VGPRs, 18 KB: six workgroups per CU).  History (all measured at C4, see profiles/r01): fused
// versions -- 23 ds_add_f32 per hit: 59 % of wave cycles stalled on LDS issue (1.26 ms); one thread per (face, quadrant)
// accumulating in registers: ~15 % lane utilisation (1.25 ms); hit-parallel phase with a segmented scan inside this kernel:
// 168 VGPRs + 50 KB LDS -> 3 waves/SIMD, 0.56 ms.  Two-kernel versions -- hits parked in an LDS pool per pass of at
// most 8 per pixel, transposed face-major by 64 wave ballots + block scan (0.274 ms), by per-pass counting with
// integer LDS atomics (0.146 ms); counting during rasterisation removed the passes, the pool and 4 of the ~9
// barriers per chunk (0.100 ms); the forward's masks removed the rasterisation (0.081 ms).
// ---------------------------------------------------------------------------
// Bit 31 of HitRecord.pixel (pixel indices fit 31 bits, check_scene): a covered pair the forward skipped (denom == 0,
// forward.cu:429-430).  Its record stays a member of its list entry's run of records -- same key in the hit-parallel
// kernel's segmented scan, all sums zero -- because that scan assumes equal keys are contiguous: a record with a key of
// its own in the middle of a run splits it, and the sum of the first part was staged twice (found by
// tests/tools/fuzz_campaign.py: a sliver face whose denom is 0 at some pixels only).
constexpr uint32_t HIT_SKIPPED = 0x80000000u;

// SCANNED: the record regions come from k_scan_hits (frames above SCAN_SINGLE_MAX tiles, first call); else they are laid out
// here (HitRegions, dmr_kernels.hpp) -- tested at run time in that instantiation: with the test folded away the register
// allocator spills 16 bytes at the kernel's 80-register cap (k_tri_backward_pix 89 -> 91-93 us at C4), as it is it does not;
// the scanned instantiation without the other path's code is 3 % faster at C5 (2.66 -> 2.58 ms).
template <bool SCANNED>
__global__ void __launch_bounds__(256, DMR_PIX_WAVES)
k_tri_backward_pix(TriParams p, const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
                   float4* __restrict__ pixrec, HitRecord* __restrict__ hits, uint32_t capacity,
                   float* __restrict__ work, uint32_t work_floats, HitRegions regions) {
    constexpr int CHUNK = BWD_CHUNK;
    constexpr int WORDS = CHUNK / 32;
    static_assert(CHUNK == 128, "one wave scans the face counters, two per lane; 256 threads stage 128 + 128 records");
    static_assert(CHUNK == MASK_CHUNK, "the forward's chunks: one mask slot each");
    __shared__ ShadeRec s_shade[CHUNK];
    __shared__ uint32_t s_fcnt[CHUNK];              // blended pixels per face of the chunk (zero between chunks)
    __shared__ uint32_t s_fcur[CHUNK];              // exclusive scan of the padded s_fcnt, then the claim cursor per face
    __shared__ uint32_t s_fpad[CHUNK];              // first pad slot of the face's run | number of pad slots << 28
    __shared__ int s_ids[CHUNK][HIT_GROUP];         // face id and its three vertex ids: word q rides in record q of every group
    __shared__ uint32_t s_max_last, s_chunk_hits;
    __shared__ unsigned long long s_before[4];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    {   // every block zeroes its slice of the packed gradient accumulators kernel 2 adds into
        const uint32_t per = (work_floats + gridDim.x - 1) / gridDim.x;
        const uint32_t z0 = min(work_floats, blockIdx.x * per), z1 = min(work_floats, z0 + per);
        for (uint32_t i = z0 + tid; i < z1; i += 256) work[i] = 0.f;
    }
    // tiles are taken longest list first (tile_order, k_scan_tiles); rows outside this shard's band are skipped
    const int tile = (int)p.tile_order[blockIdx.x];
    const int tx = tile % p.gx, ty = (tile / p.gx) % p.gy, b = tile / (p.gx * p.gy);
    const uint32_t begin = p.tile_offset[tile], end = p.tile_offset[tile + 1];
    // The tile's region of the record buffer.  Without a scan kernel (regions.hit_offset, dmr_kernels.hpp) it starts at the
    // sum of the bounds of all tiles before this one -- an empty tile, or one outside this shard's band, has none -- and this
    // workgroup publishes offset and use of its tile for the hit-parallel kernel (every tile has exactly one workgroup here).
    uint32_t region0 = 0u;
    unsigned long long before = 0ull;  // (this thread's part of) the bounds of the tiles before this one
    const bool self = !SCANNED && regions.hit_offset != nullptr, last = tile == p.B * p.gx * p.gy - 1;
    uint32_t my_bound = 0u;
    if (self) {
        static_assert(SCAN_SINGLE_MAX == 8 * 256 * 4, "eight 16-byte loads per thread cover every tile");
        const uint32_t bound = my_bound = p.tile_bound[tile];
        if (bound == 0u && !last) {  // uniform
            if (tid == 0) p.tile_used[tile] = 0u;
            return;
        }
    } else {
        if (ty < p.r0 || ty >= p.r1) return;  // uniform
        if (begin == end) return;  // uniform
        region0 = p.hit_offset[tile];
        if (region0 == p.hit_offset[tile + 1]) return;  // no pixel of the tile blended anything
    }
    if (self) {
        const uint32_t bound = my_bound;
        // this thread's part of the bounds of the tiles before this one: eight 16-byte loads, all in flight at once.
        // (Measured at C4, k_tri_backward_pix / step: this form 90 us / 0.320 ms; the same loads consumed behind the pixel's
        // own loads, 32 registers live across the ray set-up, or the pixel's loads requested first: spills at the kernel's
        // 80-register cap, 95-100 us / 0.33; block sums added by the forward with
        // atomics so that two loads per thread suffice: 85 us but k_tri_forward + 3 us at C4, + 9 us at C2; 96 dword loads of
        // tile_hits / tile_offset: 93 us; the scan kernel this replaces: 85 us + 8 us + a launch, 0.331 ms.)
        const uint4* __restrict__ tb = reinterpret_cast<const uint4*>(p.tile_bound);
        uint4 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int q = 256 * j + tid;
            v[j] = 4 * q < tile ? tb[q] : make_uint4(0u, 0u, 0u, 0u);  // (a last partial quad reads into the next array of the buffer)
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int i = 4 * (256 * j + tid);
            before += (unsigned long long)((i < tile ? v[j].x : 0u)) + (i + 1 < tile ? v[j].y : 0u);
            before += (unsigned long long)((i + 2 < tile ? v[j].z : 0u)) + (i + 3 < tile ? v[j].w : 0u);
        }
        if (bound == 0u) {  // the last tile, and nothing blended in it: only the total is wanted of this workgroup
#pragma unroll
            for (int dlt = 32; dlt > 0; dlt >>= 1) before += __shfl_xor(before, dlt, 64);
            if (lane == 0) s_before[wave] = before;
            __syncthreads();
            if (tid == 0) {
                const unsigned long long total = s_before[0] + s_before[1] + s_before[2] + s_before[3];
                *regions.hit_total = total;
                if (regions.host_hit_total) *regions.host_hit_total = host_size_word(regions.host_seq, total);
                if (regions.overflow && total > (unsigned long long)capacity) *regions.overflow = 1u;
                p.tile_used[tile] = 0u;
            }
            return;
        }
    }
    const int lx = (wave & 1) * 8 + (lane & 7), ly = (wave >> 1) * 8 + (lane >> 3);
    const int px = tx * TILE + lx, py = ty * TILE + ly;
    const bool inside = px < p.W && py < p.H;
    const int64_t HW = (int64_t)p.H * p.W;
    const int64_t pix_id = (int64_t)p.W * py + px;
    const int64_t bpix = (int64_t)b * HW + pix_id;

    V3 ro = {0, 0, 0}, rd = {0, 0, 0};
    if (inside) pixel_ray<false>(p.inv_mv + 16 * b, p.inv_proj + 16 * b, px, py, p.W, p.H, ro, rd);
    const V3 view_o = {p.inv_mv[16 * b + 12], p.inv_mv[16 * b + 13], p.inv_mv[16 * b + 14]};

    const float T_final = inside ? p.final_T[bpix] : 0.f;
    const float prev_T_final = inside ? p.final_prev_T[bpix] : 0.f;
    const uint32_t last_contributor = inside ? p.n_contrib[bpix] : 0u;
    float dpc0 = 0, dpc1 = 0, dpc2 = 0, dpd = 0;
    if (inside) {
        dpc0 = dL_dcolor[((int64_t)b * 3 + 0) * HW + pix_id];
        dpc1 = dL_dcolor[((int64_t)b * 3 + 1) * HW + pix_id];
        dpc2 = dL_dcolor[((int64_t)b * 3 + 2) * HW + pix_id];
        dpd = dL_ddepth[bpix];
    }
    // what kernel 2 needs of a pixel: ray direction and upstream gradient.  Tile-major (the tile's 256 pixels are 8 KB in a
    // row): kernel 2's workgroup stages them in LDS with coalesced loads instead of two 16-byte gathers per pair.
    const int pl = ly * TILE + lx;  // tile-local pixel index
    if (!DMR_DBG(p, 32768)) {
        pixrec[2 * ((int64_t)tile * TILE_PIX + pl)] = make_float4(rd.x, rd.y, rd.z, dpd);
        pixrec[2 * ((int64_t)tile * TILE_PIX + pl) + 1] = make_float4(dpc0, dpc1, dpc2, 0.f);
    }
    // backward.cu:293-298 (loop invariant there)
    float bg_dot = 0.f;
    bg_dot += p.bg[0] * dpc0; bg_dot += p.bg[1] * dpc1; bg_dot += p.bg[2] * dpc2;
    const float bd_dot = 0.f + (float)(1.0 * (double)dpd);

    if (tid == 0) s_max_last = 0;
    if (tid < CHUNK) s_fcnt[tid] = 0u;
    __syncthreads();
    if (last_contributor) atomicMax(&s_max_last, last_contributor);
    if (self) {
#pragma unroll
        for (int dlt = 32; dlt > 0; dlt >>= 1) before += __shfl_xor(before, dlt, 64);
        if (lane == 0) s_before[wave] = before;
    }
    __syncthreads();
    const uint32_t total = s_max_last;  // list positions >= total contribute to no pixel of the tile
    if (self) {
        before = s_before[0] + s_before[1] + s_before[2] + s_before[3];
        region0 = (uint32_t)before;  // (offsets beyond 2^32 wrap harmlessly: the total says so)
        if (tid == 0) {
            regions.hit_offset[tile] = region0;
            if (last) {
                const unsigned long long all = before + p.tile_bound[tile];
                *regions.hit_total = all;
                if (regions.host_hit_total) *regions.host_hit_total = host_size_word(regions.host_seq, all);
                if (regions.overflow && all > (unsigned long long)capacity) *regions.overflow = 1u;
            }
            if (total == 0) p.tile_used[tile] = 0u;  // (cannot happen for a tile with blended pairs; published all the same)
        }
    }
    if (total == 0) return;
    uint32_t hit_cursor = region0;  // the tile's region of the record buffer

    // pixel-thread state of the reverse walk
    float T = prev_T_final;
    bool first_pass = true;
    float acr0 = 0, acr1 = 0, acr2 = 0, acrd = 0;
    float last_alpha = 0, lc0 = 0, lc1 = 0, lc2 = 0, last_depth = 0;
    const uint32_t pixel = (uint32_t)pl;  // records carry the tile-local pixel

    const uint32_t nchunks = (total + CHUNK - 1) / CHUNK;
    // Chunks are the forward's (list positions [c * CHUNK, (c + 1) * CHUNK), cut at `total`), walked from the back;
    // thread t stages position lo + (t mod CHUNK): threads < CHUNK the geometry half of its record, the others the attributes.
    const int sj = tid & (CHUNK - 1);
    auto chunk_face = [&](uint32_t ci) -> int {
        if (ci >= nchunks) return -1;
        const uint32_t lo = (nchunks - 1u - ci) * CHUNK, hi = min(total, lo + (uint32_t)CHUNK);
        return lo + (uint32_t)sj < hi ? (int)p.face_list[begin + lo + sj] : -1;
    };
    const ChunkMasks masks = chunk_masks(p, blockIdx.x, begin);
    FaceIds ids = load_face_ids(p, b, chunk_face(0));
    int face_next = chunk_face(1);
    DMR_STAMP_RT(p, 1, 0);
    for (uint32_t ci = 0; ci < nchunks; ci++) {
        const uint32_t fc = nchunks - 1u - ci;  // the forward's chunk index
        const uint32_t lo = fc * CHUNK, hi = min(total, lo + (uint32_t)CHUNK);  // chunk = list positions [lo, hi)
        const int n = (int)(hi - lo);
        DMR_STAMP(p, 1, ci, 0);
        // ---- A: this pixel's coverage bits of the chunk, as the forward's rasterisation left them (requested before the
        // barrier, in flight across it), restricted to list positions below the pixel's n_contrib (backward.cu:192-194)
        uint4 mm = masks.at(fc)[pl];
        __syncthreads();  // previous chunk is done with the LDS records and cursors
        DMR_STAMP(p, 1, ci, 1);
        if (tid < CHUNK) {
            if (sj < n) stage_shade_geom(p, ids, view_o, s_shade[sj]);
            *reinterpret_cast<int4*>(&s_ids[tid][0]) = make_int4(ids.face, ids.v0, ids.v1, ids.v2);
        } else if (sj < n) {
            stage_shade_attr(p, b, ids, s_shade[sj]);
        }
        ids = load_face_ids(p, b, face_next);  // in flight while this chunk is processed
        face_next = chunk_face(ci + 2);
        uint32_t m[WORDS] = {mm.x, mm.y, mm.z, mm.w};
        {
            const int lim = last_contributor > lo ? (int)min(last_contributor - lo, (uint32_t)CHUNK) : 0;
#pragma unroll
            for (int w = 0; w < WORDS; w++) {
                const int keep = lim - 32 * w;  // bits of word w below the bound
                m[w] = (keep >= 32 && !DMR_DBG(p, 4)) ? m[w] : (keep <= 0 || DMR_DBG(p, 4) ? 0u : (m[w] & ((1u << keep) - 1u)));
            }
            // ... and every face's number of such pixels (the counters are zero between chunks: the scan wave clears them)
#pragma unroll
            for (int w = 0; w < WORDS; w++) {
                uint32_t t = m[w];
                while (t) {
                    const int bit = __ffs(t) - 1;
                    t &= t - 1u;
                    atomicAdd(&s_fcnt[32 * w + bit], 1u);
                }
            }
        }
        DMR_STAMP(p, 1, ci, 2);
        __syncthreads();
        DMR_STAMP(p, 1, ci, 3);
        DMR_STAMP(p, 1, ci, 4);
        DMR_STAMP(p, 1, ci, 5);
        if (wave == 0) {  // ---- S: lane l scans counters [l * PER, (l + 1) * PER)
            // A face's records form ONE run padded to a multiple of HIT_GROUP (the hit-parallel kernel takes HIT_GROUP
            // records of one list entry per lane, so its segmented scan runs once per group instead of once per record);
            // the pad slots are filled here with records that contribute nothing.
            constexpr int PER = CHUNK / 64;
            uint32_t c[PER], cp[PER]; uint32_t sum = 0;
#pragma unroll
            for (int i = 0; i < PER; i++) {
                c[i] = s_fcnt[lane * PER + i];
                s_fcnt[lane * PER + i] = 0u;  // for the next chunk's count (two barriers away)
                cp[i] = (c[i] + (uint32_t)(HIT_GROUP - 1)) & ~(uint32_t)(HIT_GROUP - 1);
                sum += cp[i];
            }
            uint32_t incl = sum;
#pragma unroll
            for (int dlt = 1; dlt < 64; dlt <<= 1) {
                const uint32_t o = __shfl_up(incl, dlt, 64);
                if (lane >= dlt) incl += o;
            }
            uint32_t run = incl - sum;
#pragma unroll
            for (int i = 0; i < PER; i++) {
                s_fcur[lane * PER + i] = run;
                s_fpad[lane * PER + i] = (run + c[i]) | ((cp[i] - c[i]) << 28);  // first pad slot | number of pad slots
                run += cp[i];
            }
            if (lane == 63) s_chunk_hits = incl;
        }
        DMR_STAMP(p, 1, ci, 6);
        __syncthreads();
        DMR_STAMP(p, 1, ci, 7);
        {   // the pad records of face `sj` (<= HIT_GROUP - 1, they contribute nothing): two threads per face, alternating
            const uint32_t fp = s_fpad[sj], slot0 = hit_cursor - region0 + (fp & 0x0fffffffu), npad = fp >> 28;
            HitRecord pad; pad.pixel = HIT_SKIPPED; pad.T = 0.f; pad.dL_dalpha = 0.f;
            for (uint32_t q = (uint32_t)(tid >> 7); q < npad; q += 2u) {
                const uint32_t addr = region0 + hit_slot_address(slot0 + q);
                pad.id = (uint32_t)s_ids[sj][(slot0 + q) & (uint32_t)(HIT_GROUP - 1)];
                if (addr < capacity && !DMR_DBG(p, 16384)) hits[addr] = pad;
            }
        }
        DMR_STAMP(p, 1, ci, 8);
        // ---- B
        while (true) {
            int w = -1; uint32_t mw = 0;
#pragma unroll
            for (int q = 0; q < WORDS; q++) if (m[q]) { w = q; mw = m[q]; }  // highest non-empty word
            if (w < 0) break;
            const int bit = 31 - __clz((int)mw);
#pragma unroll
            for (int q = 0; q < WORDS; q++) if (q == w) m[q] = mw & ~(1u << bit);
            const int k = 32 * w + bit;
            const uint32_t rel = hit_cursor - region0 + atomicAdd(&s_fcur[k], 1u);
            const uint32_t slot = region0 + hit_slot_address(rel);
            HitRecord hr;
            hr.id = (uint32_t)s_ids[k][rel & (uint32_t)(HIT_GROUP - 1)];
            const ShadeRec& r = s_shade[k];
            const V3 E1 = {r.E1[0], r.E1[1], r.E1[2]}, E2 = {r.E2[0], r.E2[1], r.E2[2]};
            const V3 Tv = {r.T[0], r.T[1], r.T[2]}, Q = {r.Q[0], r.Q[1], r.Q[2]};
            const V3 Pv = cross(rd, E2);
            const float denom = dot(Pv, E1);
            if (denom == 0.0f) {  // "edge case": skipped entirely (backward.cu:215-216); its slot says so
                hr.pixel = pixel | HIT_SKIPPED; hr.T = 0.f; hr.dL_dalpha = 0.f;
                if (slot < capacity && !DMR_DBG(p, 16384)) hits[slot] = hr;
                continue;
            }
            const float inv_denom = fast_rcp(denom);
            const float iu = dot(Pv, Tv) * inv_denom;
            const float iv = dot(Q, rd) * inv_denom;
            float iuc, ivc; int code;
            clamp_bary_uv(iu, iv, iuc, ivc, code);
            const float i0 = 1 - iuc - ivc, i1 = iuc, i2 = ivc;
            const float intense = r.intense;
            const float iC0 = (i0 * r.c0[0] + i1 * r.c1[0] + i2 * r.c2[0]) * intense;
            const float iC1 = (i0 * r.c0[1] + i1 * r.c1[1] + i2 * r.c2[1]) * intense;
            const float iC2 = (i0 * r.c0[2] + i1 * r.c1[2] + i2 * r.c2[2]) * intense;
            const float iD = i0 * r.d0 + i1 * r.d1 + i2 * r.d2;
            const float alpha = r.opacity;
            const float inv_1ma = fast_rcp(1.f - alpha);
            if (!first_pass) T = T * inv_1ma;  // Q10
            first_pass = false;
            float dL_dalpha = 0.0f;
            acr0 = last_alpha * lc0 + (1.f - last_alpha) * acr0; lc0 = iC0; dL_dalpha += (iC0 - acr0) * dpc0;
            acr1 = last_alpha * lc1 + (1.f - last_alpha) * acr1; lc1 = iC1; dL_dalpha += (iC1 - acr1) * dpc1;
            acr2 = last_alpha * lc2 + (1.f - last_alpha) * acr2; lc2 = iC2; dL_dalpha += (iC2 - acr2) * dpc2;
            acrd = last_alpha * last_depth + (1.f - last_alpha) * acrd; last_depth = iD;
            dL_dalpha += (iD - acrd) * dpd;
            dL_dalpha *= T;
            last_alpha = alpha;
            if (alpha == 1.0f) {
                dL_dalpha += (-prev_T_final) * bg_dot;
                dL_dalpha += (-prev_T_final) * bd_dot;
            } else {
                dL_dalpha += (-T_final * inv_1ma) * bg_dot;
                dL_dalpha += (-T_final * inv_1ma) * bd_dot;
            }
            hr.pixel = pixel; hr.T = T; hr.dL_dalpha = dL_dalpha;
            if (slot < capacity && !DMR_DBG(p, 16384)) hits[slot] = hr;  // capacity < total only while a size guess is being refuted
        }
        hit_cursor += s_chunk_hits;  // stable until the next chunk's scan, two barriers away
        DMR_STAMP(p, 1, ci, 9);
    }
    DMR_STAMP_RT(p, 1, 1);
    if (tid == 0) p.tile_used[tile] = hit_cursor - region0;  // what the hit-parallel kernel walks (a multiple of HIT_GROUP)
}

// ---------------------------------------------------------------------------
// backward, kernel 2 of 2: k_tri_backward_hits -- one lane per group of four (pixel, face) pairs of one list entry.
//
// One workgroup per tile, no barriers in its loop.  A lane gathers its face once, recomputes each pair's (u, v) and
// clamp region exactly as the forward did, and adds the pair to 21 sums: the colour / depth / opacity / intensity
// gradients of backward.cu:313-382 and, for the vertex positions, the seven RAY MOMENTS ray_tri_intersection_grad is
// linear in (see the loop).  A segmented DPP scan keyed by the face leaves each entry's totals in the last lane of its
// segment; that lane turns the moments into the three position gradients (six cross products per list entry instead of
// ~150 instructions per pair), adds its three vertex rows into the tile's LDS table and sends the face row out.  The
// table leaves as packed global atomics once per tile: against the reference's 23 global atomics per (pixel, face)
// (backward.cu:389-418).
// ---------------------------------------------------------------------------
// From here to k_tri_backward_hits' end the compiler may contract a*b+c to FMA: gradients are checked to 1e-4
// and the reference's own sums are unordered float atomics.  F3 is this block's vector type (the V3 helpers of
// dmr_device.hpp were compiled under -ffp-contract=off and keep that when inlined).  The forward and everything
// that decides an index stay exact.
#ifndef DMR_HITS_NOCONTRACT
#pragma clang fp contract(fast)
#endif
namespace fm {
struct F3 { float x, y, z; };
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ F3 operator-(F3 a) { return {-a.x, -a.y, -a.z}; }
__device__ __forceinline__ F3 operator*(float b, F3 a) { return {b * a.x, b * a.y, b * a.z}; }
__device__ __forceinline__ float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ F3 cross(F3 a, F3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ F3 load3(const float* __restrict__ a, int id) { return {a[3 * id], a[3 * id + 1], a[3 * id + 2]}; }
}  // namespace fm

#ifndef DMR_VTAB
#define DMR_VTAB 640
#endif
#ifndef DMR_HITS_UNROLL
#define DMR_HITS_UNROLL 1
#endif
constexpr int VTAB = DMR_VTAB;   // vertex-row slots per workgroup
constexpr int TAB_PROBES = 16;
constexpr uint32_t TAB_EMPTY = 0xffffffffu;

// Workgroup-level aggregation of the vertex gradient rows (LDS: 640 slots, 38 KB; with the tile's pixel records 46 KB: three
// workgroups per CU.  C4's busiest tiles touch ~500 rows: 512 slots left them with probe sequences at load factor ~1 --
// k_tri_backward_hits 101.4 us with 512 slots, 95.6-98.2 with 576 / 640 / 704, 117 with 768 = two workgroups per CU).
// Global float atomics execute at the memory side at ~20 G 64-byte requests/s chip-wide whatever they carry
// (MI355X_MICROARCH.md, "Global float atomics"): with one request per (segment, row) -- 3 vertex rows + 1 face row
// per list entry -- this kernel was bound by exactly that (0.195 ms with the atomics, 0.110 ms without, 0.195 ms
// with one dword per row).  A workgroup walks the records of ONE tile, i.e. the faces of a few surfaces crossing it,
// whose vertex rows repeat (valence 6: ~0.6 distinct rows per entry over a tile, not 3).  So segment totals are first
// added into an LDS table keyed by row -- insertion with ds_cmpst, sums with ds_add_f64 -- and a row goes to HBM once
// per tile.  The cells are DOUBLES for the atomic's rate, not for precision: ds_add_f32 retires one lane per ~3 cycles
// per CU, ds_add_f64 ten times that (scripts/micro/lds_atomics.hip).  A full table or a long probe sequence falls
// back to the direct atomics for that row.  Face rows (opacity, intensity) are unique per (tile, face): a table cannot
// merge anything but the partial sums of one entry, so they go out directly, one 8-byte request per segment tail.
// What the atomics cost at C4 (timing builds without them, round 2): the table's flush 23 us (0.53 M requests), the
// face rows 11 us (0.9 M) of the kernel's 106; staging the tails through LDS so that every lane takes one row (the
// earlier layout) cost 5 us more than letting the tail lanes add their three rows from registers.
#ifndef DMR_HITS_PIX_LDS
#define DMR_HITS_PIX_LDS 1
#endif
struct HitsLds {
#if DMR_HITS_PIX_LDS
    float4 pix[2 * TILE_PIX];   // the tile's pixels: (ray direction, dL/ddepth), (dL/dcolor, -)
#endif
    uint32_t vkey[VTAB];
    double vval[VTAB][7];   // dx dy dz dr dg db ddepth of row (view, vertex)
    uint32_t frow_stage[4][64][3];  // per wave: {face row id, dopacity, dintense} of the round's segment tails (see the loop)
};

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Rows 2k and 2k + 1 share a 64-byte line of `vrow`; they hash to the two slots of one PAIR of slots (the probe sequence
// moves pair by pair and keeps the row's parity), so the flush -- consecutive slots in consecutive lane groups of one
// instruction -- sends both out in one memory-side request whenever a tile holds both (neighbouring vertex ids are
// neighbouring vertices in most meshes; nothing is lost when they are not): 106.6 -> 102.6 us at C4.
template <int SLOTS>
__device__ __forceinline__ uint32_t tab_home(uint32_t rid) {  // the slot pair a row hashes to
    constexpr int PAIRS = SLOTS / 2;
    constexpr bool POW2 = (PAIRS & (PAIRS - 1)) == 0;
    static_assert(SLOTS % 2 == 0, "pairs of slots");
    const uint32_t h = (rid >> 1) * 2654435761u;
    return POW2 ? ((h >> 8) & (uint32_t)(PAIRS - 1)) : __umulhi(h, (uint32_t)PAIRS);
}
// probes `probes` slot pairs from `pair` on; -1: no slot (the row goes out with direct atomics)
template <int SLOTS>
__device__ __forceinline__ int tab_probe(uint32_t* __restrict__ key, uint32_t rid, uint32_t pair, int probes) {
    constexpr int PAIRS = SLOTS / 2;
    constexpr bool POW2 = (PAIRS & (PAIRS - 1)) == 0;
    for (int i = 0; i < probes; i++) {
        const uint32_t slot = 2u * pair + (rid & 1u);
        const uint32_t prev = atomicCAS(&key[slot], TAB_EMPTY, rid);
        if (prev == TAB_EMPTY || prev == rid) return (int)slot;
        pair = POW2 ? ((pair + 1u) & (uint32_t)(PAIRS - 1)) : (pair + 1u == (uint32_t)PAIRS ? 0u : pair + 1u);
    }
    return -1;
}
template <int SLOTS>
__device__ __forceinline__ int tab_find(uint32_t* __restrict__ key, uint32_t rid) {
    return tab_probe<SLOTS>(key, rid, tab_home<SLOTS>(rid), TAB_PROBES);
}
// The three rows of a list entry: their first probes are in flight together (three LDS round trips one after the other
// were a sizeable part of what a segment tail costs); only a row whose home slot is taken by another row probes on.
template <int SLOTS>
__device__ __forceinline__ void tab_find3(uint32_t* __restrict__ key, const uint32_t (&rid)[3], int (&slot)[3]) {
    constexpr int PAIRS = SLOTS / 2;
    constexpr bool POW2 = (PAIRS & (PAIRS - 1)) == 0;
    uint32_t home[3], prev[3];
#pragma unroll
    for (int w = 0; w < 3; w++) home[w] = tab_home<SLOTS>(rid[w]);
#pragma unroll
    for (int w = 0; w < 3; w++) prev[w] = atomicCAS(&key[2u * home[w] + (rid[w] & 1u)], TAB_EMPTY, rid[w]);
#pragma unroll
    for (int w = 0; w < 3; w++) {
        if (prev[w] == TAB_EMPTY || prev[w] == rid[w]) slot[w] = (int)(2u * home[w] + (rid[w] & 1u));
        else {
            const uint32_t next = POW2 ? ((home[w] + 1u) & (uint32_t)(PAIRS - 1)) : (home[w] + 1u == (uint32_t)PAIRS ? 0u : home[w] + 1u);
            slot[w] = tab_probe<SLOTS>(key, rid[w], next, TAB_PROBES - 1);
        }
    }
}

// One workgroup per tile (longest list first), one lane per GROUP of HIT_GROUP consecutive records -- all of one list
// entry, because k_tri_backward_pix pads every face's run to a multiple of HIT_GROUP.  The lane gathers its face once,
// sums the 21 components of its (up to) HIT_GROUP pairs in registers, and only then enters the segmented scan: one scan,
// one tail hand-off per group instead of per record (the scan was 92 half-rate DPP instructions of ~600 per 64 records),
// and HIT_GROUP independent pixel gathers in flight per lane.
#ifndef DMR_HITS_WAVES
#define DMR_HITS_WAVES 1
#endif
__global__ void __launch_bounds__(256, DMR_HITS_WAVES)
k_tri_backward_hits(TriParams p, const float4* __restrict__ pixrec, const HitRecord* __restrict__ hits, uint32_t capacity,
                    float* __restrict__ vrow, float* __restrict__ frow) {
    const int tile = (int)p.tile_order[blockIdx.x];
    uint32_t nrec = p.tile_used[tile];
    if (nrec == 0u) return;  // uniform: nothing blended in this tile (or outside this shard's band)
    const uint32_t rec0 = p.hit_offset[tile];
    // (fewer only while a size guess is being refuted / an asynchronous call overflowed: whole groups inside the buffer)
    if (rec0 + ((nrec + (uint32_t)(HIT_BLOCK - 1)) & ~(uint32_t)(HIT_BLOCK - 1)) > capacity) return;  // the results are thrown away
    const uint32_t ngroups = nrec / (uint32_t)HIT_GROUP;
    const int b = tile / (p.gx * p.gy);

    __shared__ HitsLds L;
    const int tid = threadIdx.x, lane = tid & 63;
    DMR_STAMP_RT(p, 2, 0);
    DMR_STAMP(p, 2, 0u, 8);
    const V3 view_o = {p.inv_mv[16 * b + 12], p.inv_mv[16 * b + 13], p.inv_mv[16 * b + 14]};

    // record q of a lane's group: the wave's 64 lanes read one contiguous kilobyte (dmr_kernels.hpp, HIT_BLOCK).
    // The four `id` words of a group are the face and its three vertices (k_tri_backward_pix), so the vertex data can
    // be gathered as soon as the records are here: two dependent memory levels per round instead of four
    // (record -> list entry -> face -> vertices) -- and the records of round r + 1 are loaded while round r computes.
    auto load_group = [&](uint32_t gi, uint4 (&r)[HIT_GROUP]) {
        const bool in = gi < ngroups;
        const uint4* src = reinterpret_cast<const uint4*>(hits + rec0 + (uint64_t)((in && !DMR_DBG(p, 65536) ? gi : 0u) / 64u) * HIT_BLOCK + (uint32_t)lane);
#pragma unroll
        for (int q = 0; q < HIT_GROUP; q++) r[q] = in ? src[64 * q] : make_uint4(0u, HIT_SKIPPED, 0u, 0u);
    };
    // the first round's records and the tile's pixels are requested before the table is cleared: the workgroup's set-up
    // costs one memory latency, not one after the other (phase stamps: 5.1 k cycles of a tile's ~55 k were set-up)
    uint4 raw[HIT_GROUP], nxt[HIT_GROUP];
    load_group((uint32_t)tid, nxt);
#if DMR_HITS_PIX_LDS
    const float4 px0 = pixrec[2 * (int64_t)tile * TILE_PIX + tid], px1 = pixrec[2 * (int64_t)tile * TILE_PIX + 256 + tid];
#endif
    for (int i = tid; i < VTAB; i += 256) {
        L.vkey[i] = TAB_EMPTY;
#pragma unroll
        for (int c = 0; c < 7; c++) L.vval[i][c] = 0.0;
    }
#if DMR_HITS_PIX_LDS
    L.pix[tid] = px0;
    L.pix[tid + 256] = px1;
#endif
    __syncthreads();
    DMR_STAMP(p, 2, 0u, 9);
    for (uint32_t g0 = 0; g0 < ngroups; g0 += 256u) {
        const uint32_t gi = g0 + (uint32_t)tid;
        const bool valid = gi < ngroups;   // lanes past the end: no group, unique keys
        DMR_STAMP(p, 2, g0 / 256u, 0);
#pragma unroll
        for (int q = 0; q < HIT_GROUP; q++) raw[q] = nxt[q];
        load_group(gi + 256u, nxt);  // in flight while this round computes (past the end: nothing is loaded)
        int k = -1 - lane;  // invalid lanes: unique keys
        int v0 = 0, v1 = 0, v2 = 0, face = 0;
        float g[NSCAN];
#pragma unroll
        for (int c = 0; c < NSCAN; c++) g[c] = 0.f;
        if (valid) {  // key and row ids: a group of skipped pairs at the end of its run still carries the run's sums to the tables
            face = (int)raw[0].x; v0 = (int)raw[1].x; v1 = (int)raw[2].x; v2 = (int)raw[3].x;
            k = face;  // a face occurs once per tile: as good a segment key as the list entry
        }
        // the face's ray-independent vectors: needed by the pairs below and, after the scan, by the segment tails
        V3 xT = {0.f, 0.f, 0.f}, xE1 = xT, xE2 = xT, xQ = xT, xE12 = xT, xE2T = xT;
        float w2 = 0.f;
        if (valid) {
            using namespace fm;
            // the face, once per group
            const float alpha = p.faces_opacity[face], intense = p.faces_intense[(int64_t)b * p.F + face];
            const F3 cc0 = load3(p.verts_color, v0), cc1 = load3(p.verts_color, v1), cc2 = load3(p.verts_color, v2);
            const float fd0 = p.vproj[(int64_t)b * p.P + v0].w, fd1 = p.vproj[(int64_t)b * p.P + v1].w,
                        fd2 = p.vproj[(int64_t)b * p.P + v2].w;
            const V3 xp0 = load_v3(p.verts, v0), xp1 = load_v3(p.verts, v1), xp2 = load_v3(p.verts, v2);
            xT = view_o - xp0; xE1 = xp1 - xp0; xE2 = xp2 - xp0;
            xQ = dmr::cross(xT, xE1);
            w2 = dmr::dot(xQ, xE2);
            xE12 = dmr::cross(xE1, xE2); xE2T = dmr::cross(xE2, xT);
            // i0 = 1 - uc - vc, i1 = uc, i2 = vc: what multiplies uc and vc in the interpolated colour / depth
            const F3 dc10 = cc1 - cc0, dc20 = cc2 - cc0;
            const float dd10 = fd1 - fd0, dd20 = fd2 - fd0;
#ifdef DMR_ABLATION
            if (DMR_DBG(p, 4096)) {  // (phase stamps: the face's gathers have arrived)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                DMR_STAMP(p, 2, g0 / 256u, 1);
            }
#endif
#pragma unroll
            for (int q = 0; q < HIT_GROUP; q++) {
                if (raw[q].y & HIT_SKIPPED) continue;  // pad, or a pair the forward skipped (denom == 0)
#if DMR_HITS_PIX_LDS
                const float4 pr0 = L.pix[2 * (raw[q].y & 255u)], pr1 = L.pix[2 * (raw[q].y & 255u) + 1];
#else
                const float4* tp = pixrec + 2 * ((int64_t)tile * TILE_PIX + (raw[q].y & 255u));
                const float4 pr0 = tp[0], pr1 = tp[1];
#endif
                const float hT = __uint_as_float(raw[q].z), hdLda = __uint_as_float(raw[q].w);
                // forward quantities of this (pixel, face) pair (backward.cu:206-243).  Exact arithmetic (the V3 helpers
                // are not contracted): the clamp region `code` selects a piecewise-constant Jacobian, so (u, v) must land
                // on the same side of the region borders as in the forward.
                const V3 xd = {pr0.x, pr0.y, pr0.z};
                const V3 xP = dmr::cross(xd, xE2);
                const float denom = dmr::dot(xP, xE1);
                const float inv_denom = 1.0f / denom;  // IEEE like the forward: a 1-ulp v_rcp_f32 flips the region of a
                                                       // (u, v) on a border now and then (4 vertices at C5's 376 M pairs)
                const float nu = dmr::dot(xP, xT);
                const float iu = nu * inv_denom;
                const float iv = dmr::dot(xQ, xd) * inv_denom;
                float iuc, ivc; int code;
                clamp_bary_uv(iu, iv, iuc, ivc, code);
                const float i0 = 1 - iuc - ivc, i1 = iuc, i2 = ivc;
                // dL/dcolor and dL/ddepth of the pair (backward.cu:254-275), the face intensity folded in once
                const float aT = alpha * hT;
                const float dic0 = pr1.x * aT, dic1 = pr1.y * aT, dic2 = pr1.z * aT, did = pr0.w * aT;
                const float dii0 = dic0 * intense, dii1 = dic1 * intense, dii2 = dic2 * intense;
                // dL/d(uc), dL/d(vc) (backward.cu:313-330: dL/d(weights), then i1 - i0 and i2 - i0) and dL/dintensity
                const float e1 = dc10.x * dii0 + dc10.y * dii1 + dc10.z * dii2 + dd10 * did;
                const float e2 = dc20.x * dii0 + dc20.y * dii1 + dc20.z * dii2 + dd20 * did;
                const float dfint = (cc0.x + iuc * dc10.x + ivc * dc20.x) * dic0 + (cc0.y + iuc * dc10.y + ivc * dc20.y) * dic1
                                  + (cc0.z + iuc * dc10.z + ivc * dc20.z) * dic2;
                // through the clamp (Jacobian of auxiliary.h:374-400)
                float duc_du, duc_dv, dvc_du, dvc_dv;
                clamp_bary_uv_grad(code, duc_du, duc_dv, dvc_du, dvc_dv);
                const float dL_diu = e1 * duc_du + e2 * dvc_du;
                const float dL_div = e1 * duc_dv + e2 * dvc_dv;
                const float dinv = inv_denom * inv_denom;
                // The sums below multiply neighbours' partial sums by 0/1 masks in the scan, so a non-finite value must not
                // enter them (0 * inf = NaN would leak into the next list entry).  Such a pair (degenerate face, Q12; opacity
                // exactly 1 behind it) adds its values with plain atomics, as the reference does for every pair.
                const float chk = (dinv * 0.f) + (aT * 0.f) + (hdLda * 0.f);
                if (!(chk == 0.f)) {
                    // ray_tri_intersection_grad as the reference writes it (auxiliary.h:288-333; Q11: the "v" numerator
                    // is t's, Q12: no clamp of denom^2), pair by pair
                    const float dL_di0 = cc0.x * dii0 + cc0.y * dii1 + cc0.z * dii2 + fd0 * did;
                    const float dL_di1 = cc1.x * dii0 + cc1.y * dii1 + cc1.z * dii2 + fd1 * did;
                    const float dL_di2 = cc2.x * dii0 + cc2.y * dii1 + cc2.z * dii2 + fd2 * did;
                    const float f1 = dL_di1 - dL_di0, f2 = dL_di2 - dL_di0;
                    const float rdiu = f1 * duc_du + f2 * dvc_du, rdiv = f1 * duc_dv + f2 * dvc_dv;
                    const float rinv = 1.0f / (denom * denom);
                    const float w0 = nu, w1 = denom;
                    const V3 du_dE1 = (-1.0f * xP * w0) * rinv;
                    const V3 du_dE2 = (dmr::cross(xT, xd) * w1 - w0 * dmr::cross(xE1, xd)) * rinv;
                    const V3 du_dT = (xP * w1) * rinv;
                    const V3 dv_dE1 = ((xE2T * w1) - (w2 * xP)) * rinv;
                    const V3 dv_dE2 = ((xQ * w1) - (w2 * dmr::cross(xE1, xd))) * rinv;
                    const V3 dv_dT = xE12 * w1 * rinv;
                    const V3 du_dp0 = -du_dE1 - du_dE2 - du_dT, dv_dp0 = -dv_dE1 - dv_dE2 - dv_dT;
                    const V3 dp0 = rdiu * du_dp0 + rdiv * dv_dp0;
                    const V3 dp1 = rdiu * du_dE1 + rdiv * dv_dE1;
                    const V3 dp2 = rdiu * du_dE2 + rdiv * dv_dE2;
                    float h[NACC];
                    h[0] = dp0.x; h[1] = dp0.y; h[2] = dp0.z;
                    h[3] = dp1.x; h[4] = dp1.y; h[5] = dp1.z;
                    h[6] = dp2.x; h[7] = dp2.y; h[8] = dp2.z;
                    h[9] = i0 * dii0; h[10] = i0 * dii1; h[11] = i0 * dii2;
                    h[12] = i1 * dii0; h[13] = i1 * dii1; h[14] = i1 * dii2;
                    h[15] = i2 * dii0; h[16] = i2 * dii1; h[17] = i2 * dii2;
                    h[18] = i0 * did; h[19] = i1 * did; h[20] = i2 * did;
                    h[21] = hdLda;
                    h[22] = (i0 * cc0.x + i1 * cc1.x + i2 * cc2.x) * dic0 + (i0 * cc0.y + i1 * cc1.y + i2 * cc2.y) * dic1
                          + (i0 * cc0.z + i1 * cc1.z + i2 * cc2.z) * dic2;
                    float* r0 = vrow + ((int64_t)b * p.P + v0) * VROW; float* r1 = vrow + ((int64_t)b * p.P + v1) * VROW;
                    float* r2 = vrow + ((int64_t)b * p.P + v2) * VROW; float* rf = frow + ((int64_t)b * p.F + face) * FROW;
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        atomicAdd(r0 + c, h[c]); atomicAdd(r1 + c, h[3 + c]); atomicAdd(r2 + c, h[6 + c]);
                        atomicAdd(r0 + 3 + c, h[9 + c]); atomicAdd(r1 + 3 + c, h[12 + c]); atomicAdd(r2 + 3 + c, h[15 + c]);
                    }
                    atomicAdd(r0 + 6, h[18]); atomicAdd(r1 + 6, h[19]); atomicAdd(r2 + 6, h[20]);
                    atomicAdd(rf, h[21]); atomicAdd(rf + 1, h[22]);
                } else {
                    // The vertex-position gradient of ray_tri_intersection_grad is LINEAR in the ray direction d once the
                    // pair's scalars are fixed -- every d enters through d x E2, T x d or E1 x d:
                    //   dL/dp1 = S3 E2T - B x E2,   dL/dp2 = T x A2 - E1 x B + S3 Q,   dL/dT = A2 x E2 + S3 E12,
                    //   dL/dp0 = -(dL/dp1 + dL/dp2 + dL/dT)
                    // with s2 = dL/du / denom, s3 = dL/dv / denom, B = sum (s2 nu + s3 w2) / denom * d, A2 = sum s2 d,
                    // S3 = sum s3 (w2 = Q . E2 is the reference's "v" numerator, Q11).  So a pair adds 7 multiply-adds here
                    // and the six cross products are taken once per list entry, at the segment tails after the scan -- not
                    // ~150 instructions per pair.  Rounding: (s d) x E2 and s (d x E2) lose the same bits.
                    const float s2 = dL_diu * inv_denom, s3 = dL_div * inv_denom;
                    const float sb = (s2 * nu + s3 * w2) * inv_denom;
                    g[0] += sb * xd.x; g[1] += sb * xd.y; g[2] += sb * xd.z;
                    g[3] += s2 * xd.x; g[4] += s2 * xd.y; g[5] += s2 * xd.z;
                    g[6] += s3;
                    g[7] += i0 * dii0; g[8] += i0 * dii1; g[9] += i0 * dii2;
                    g[10] += i1 * dii0; g[11] += i1 * dii1; g[12] += i1 * dii2;
                    g[13] += i2 * dii0; g[14] += i2 * dii1; g[15] += i2 * dii2;
                    g[16] += i0 * did; g[17] += i1 * did; g[18] += i2 * did;
                    g[19] += hdLda; g[20] += dfint;
                }
            }
        }
        // segmented inclusive scan over the wave (groups of one entry are consecutive lanes): four row-local DPP levels.
        // The scan stops at the 16-lane DPP rows: the table takes partial sums just as well, so an entry that crosses a
        // row boundary simply contributes one more partial.
        DMR_STAMP(p, 2, g0 / 256u, 2);
        seg_scan_level<DPP_ROW_SHR + 1, 0xF>(k, g);
        seg_scan_level<DPP_ROW_SHR + 2, 0xF>(k, g);
        seg_scan_level<DPP_ROW_SHR + 4, 0xF>(k, g);
        seg_scan_level<DPP_ROW_SHR + 8, 0xF>(k, g);

        // segment tails hold the totals: the ray moments become the three vertex-position gradients (see above)
        DMR_STAMP(p, 2, g0 / 256u, 3);
        const int kn = __shfl_down(k, 1, 64);
        const bool tail = valid && ((lane & 15) == 15 || kn != k);
        fm::F3 dp0, dp1, dp2;
        {
            using namespace fm;
            const F3 mB = {g[0], g[1], g[2]}, mA = {g[3], g[4], g[5]};
            const F3 fT = {xT.x, xT.y, xT.z}, fE1 = {xE1.x, xE1.y, xE1.z}, fE2 = {xE2.x, xE2.y, xE2.z};
            const F3 fQ = {xQ.x, xQ.y, xQ.z}, fE12 = {xE12.x, xE12.y, xE12.z}, fE2T = {xE2T.x, xE2T.y, xE2T.z};
            dp1 = g[6] * fE2T - cross(mB, fE2);
            dp2 = cross(fT, mA) - cross(fE1, mB) + g[6] * fQ;
            const F3 dT = cross(mA, fE2) + g[6] * fE12;
            dp0 = -(dp1 + dp2 + dT);
        }
        // The face rows (opacity, intensity: two adjacent floats) of the round's tails go out first.  A tail lane sending its two
        // values itself is two memory-side requests (two instructions): 4.6 us of this kernel at C4.  So the tails are compacted
        // through 12 bytes of wave-private LDS each and lanes 2r / 2r + 1 send the two floats of tail r in ONE instruction -- one
        // 8-byte request per tail (up to 32 tails per instruction).
        {
            const bool ft = tail && !DMR_DBG(p, 512);
            const uint64_t tmask = __ballot(ft);
            const int ntail = __popcll(tmask);
            if (ft) {
                uint32_t* st = L.frow_stage[tid >> 6][__popcll(tmask & ((1ull << lane) - 1ull))];
                st[0] = (uint32_t)b * (uint32_t)p.F + (uint32_t)face; st[1] = __float_as_uint(g[19]); st[2] = __float_as_uint(g[20]);
            }
            wave_lds_sync();
            for (int t0 = 0; t0 < ntail; t0 += 32) {
                const int t = t0 + (lane >> 1);
                if (t < ntail) {
                    const uint32_t* st = L.frow_stage[tid >> 6][t];
                    atomicAdd(&frow[(int64_t)st[0] * FROW + (lane & 1)], __uint_as_float(st[1 + (lane & 1)]));
                }
            }
            wave_lds_sync();  // (the stage may be refilled by the next round)
        }
        // a tail lane adds its entry's three vertex rows into the table, from its registers
        if (tail && !DMR_DBG(p, 512)) {
            const float rows[3][7] = {{dp0.x, dp0.y, dp0.z, g[7], g[8], g[9], g[16]},
                                      {dp1.x, dp1.y, dp1.z, g[10], g[11], g[12], g[17]},
                                      {dp2.x, dp2.y, dp2.z, g[13], g[14], g[15], g[18]}};
            const uint32_t rid[3] = {(uint32_t)b * (uint32_t)p.P + (uint32_t)v0, (uint32_t)b * (uint32_t)p.P + (uint32_t)v1,
                                     (uint32_t)b * (uint32_t)p.P + (uint32_t)v2};
            int slot[3];
            tab_find3<VTAB>(L.vkey, rid, slot);
#pragma unroll
            for (int w = 0; w < 3; w++) {
                // (ablation build, DMR_ABLATE bit 2048, tests only: odd rows are refused their slot, which exercises the direct-atomic fallback)
                if (DMR_DBG(p, 2048) && (rid[w] & 1u)) slot[w] = -1;
                if (slot[w] >= 0) {
#pragma unroll
                    for (int c = 0; c < 7; c++) atomicAdd(&L.vval[slot[w]][c], (double)rows[w][c]);
                } else {
#pragma unroll
                    for (int c = 0; c < 7; c++) atomicAdd(&vrow[(int64_t)rid[w] * VROW + c], rows[w][c]);
                }
            }
        }
        DMR_STAMP(p, 2, g0 / 256u, 4);
    }
    // every row of the table goes out once: 8 lanes per vertex row (7 used)
    DMR_STAMP(p, 2, 0u, 5);
    __syncthreads();
    DMR_STAMP(p, 2, 0u, 6);
    if (DMR_DBG(p, 1024)) return;
    // (eight slots per lane are read before the first atomic goes out: the LDS latencies overlap instead of adding up)
    constexpr int FLUSH_BATCH = VTAB % 256 == 0 ? 8 : (VTAB % 128 == 0 ? 4 : 2);
    static_assert(VTAB % (32 * FLUSH_BATCH) == 0, "table size");
    const int comp = tid & 7;
    for (int s0 = 0; s0 < VTAB; s0 += 32 * FLUSH_BATCH) {
        uint32_t rid[FLUSH_BATCH]; float val[FLUSH_BATCH];
#pragma unroll
        for (int i = 0; i < FLUSH_BATCH; i++) {
            const int slot = s0 + 32 * i + (tid >> 3);
            rid[i] = L.vkey[slot];
            val[i] = (float)L.vval[slot][comp < 7 ? comp : 0];
        }
#pragma unroll
        for (int i = 0; i < FLUSH_BATCH; i++)
            if (rid[i] != TAB_EMPTY && comp < 7) atomicAdd(&vrow[(int64_t)rid[i] * VROW + comp], val[i]);
    }
    DMR_STAMP(p, 2, 0u, 7);
    DMR_STAMP_RT(p, 2, 1);
}

#pragma clang fp contract(off)

// packed accumulators -> the five gradient tensors of render.cu:166-171
__global__ void __launch_bounds__(256)
k_tri_unpack(int B, int P, int F, const float* __restrict__ vrow, const float* __restrict__ frow,
             float* __restrict__ dL_dverts, float* __restrict__ dL_dvcolor, float* __restrict__ dL_dfopacity,
             float* __restrict__ dL_dvdepth, float* __restrict__ dL_dfintense) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < P) {
        float a[6] = {0, 0, 0, 0, 0, 0};
        for (int b = 0; b < B; b++) {
            const float4 lo = *reinterpret_cast<const float4*>(vrow + ((int64_t)b * P + idx) * VROW);
            const float4 hi = *reinterpret_cast<const float4*>(vrow + ((int64_t)b * P + idx) * VROW + 4);
            a[0] += lo.x; a[1] += lo.y; a[2] += lo.z; a[3] += lo.w; a[4] += hi.x; a[5] += hi.y;
            dL_dvdepth[(int64_t)b * P + idx] = hi.z;
        }
        dL_dverts[3 * idx] = a[0]; dL_dverts[3 * idx + 1] = a[1]; dL_dverts[3 * idx + 2] = a[2];
        dL_dvcolor[3 * idx] = a[3]; dL_dvcolor[3 * idx + 1] = a[4]; dL_dvcolor[3 * idx + 2] = a[5];
    }
    if (idx < F) {
        float o = 0.f;
        for (int b = 0; b < B; b++) {
            const float2 r = *reinterpret_cast<const float2*>(frow + ((int64_t)b * F + idx) * FROW);
            o += r.x;
            dL_dfintense[(int64_t)b * F + idx] = r.y;
        }
        dL_dfopacity[idx] = o;
    }
}

static TriParams make_params(const dmr_scene& s, int gx, int gy, int r0, int r1, const float4* vproj,
                             const uint32_t* tile_offset, const uint32_t* face_list, TriImageState img) {
    TriParams p;
    p.B = s.B; p.P = s.P; p.F = s.F; p.W = s.W; p.H = s.H; p.gx = gx; p.gy = gy; p.r0 = r0; p.r1 = r1;
#ifdef DMR_ABLATION
    { static const int dbg = getenv("DMR_ABLATE") ? atoi(getenv("DMR_ABLATE")) : 0; p.dbg = dbg; }  // ablation build only
#endif
    p.verts = s.verts; p.faces = s.faces; p.verts_color = s.verts_color; p.faces_opacity = s.faces_opacity;
    p.inv_mv = s.inv_mv_mats; p.inv_proj = s.inv_proj_mats; p.faces_intense = s.faces_intense; p.bg = s.background;
    p.vproj = vproj; p.tile_offset = tile_offset; p.face_list = face_list; p.keys = nullptr;
    p.final_T = img.final_T; p.final_prev_T = img.final_prev_T; p.n_contrib = img.n_contrib;
    p.tile_hits = img.tile_hits; p.tile_bound = img.tile_bound; p.hit_offset = img.hit_offset; p.tile_used = img.tile_used; p.tile_order = img.tile_order;
    p.mask_offset = img.mask_offset;
    p.list_capacity = 0xffffffffu;
    return p;
}

void launch_tri_forward(const dmr_scene& s, int gx, int gy, int r0, int r1, const float4* vproj,
                        const uint32_t* tile_offset, uint64_t* keys, uint32_t* face_list, uint32_t capacity, TriImageState img,
                        float* out_color, float* out_depth, hipStream_t st) {
    if (r1 <= r0) return;
    TriParams p = make_params(s, gx, gy, r0, r1, vproj, tile_offset, face_list, img);
    p.keys = reinterpret_cast<unsigned long long*>(keys);
    p.list_capacity = capacity;
    StageScope t(DMR_STAGE_TRI_FORWARD, st);
    const dim3 grid((unsigned)(s.B * gx * gy)), block(256);
    if (keys) k_tri_forward<FWD_CHUNK, true><<<grid, block, 0, st>>>(p, out_color, out_depth);
    else k_tri_forward<FWD_CHUNK, false><<<grid, block, 0, st>>>(p, out_color, out_depth);
}

void launch_tri_backward_pix(const dmr_scene& s, int gx, int gy, int r0, int r1, const float4* vproj,
                             const uint32_t* tile_offset, const uint32_t* face_list, TriImageState img,
                             const float* dL_dcolor, const float* dL_ddepth, float4* pixrec, HitRecord* hits,
                             uint32_t capacity, float* work, size_t work_floats, HitRegions regions, hipStream_t st) {
    // (an empty band never gets here: dmr_tri_backward zeroes the gradients itself -- this kernel is also what zeroes `work`)
    TriParams p = make_params(s, gx, gy, r0, r1, vproj, tile_offset, face_list, img);
    StageScope t(DMR_STAGE_TRI_BACKWARD, st);
    dim3 grid((unsigned)(s.B * gx * gy)), block(256);
#ifdef DMR_ABLATION
    if (DMR_DBG(p, 131072)) grid.x = std::min(grid.x, 3072u);  // timing experiment (results invalid): only the busiest tiles' workgroups
#endif
    if (regions.hit_offset)
        k_tri_backward_pix<false><<<grid, block, 0, st>>>(p, dL_dcolor, dL_ddepth, pixrec, hits, capacity, work, (uint32_t)work_floats, regions);
    else
        k_tri_backward_pix<true><<<grid, block, 0, st>>>(p, dL_dcolor, dL_ddepth, pixrec, hits, capacity, work, (uint32_t)work_floats, regions);
}

void launch_tri_backward_hits(const dmr_scene& s, int gx, int gy, const float4* vproj, const uint32_t* face_list, TriImageState img,
                              const float4* pixrec, const HitRecord* hits, uint32_t capacity, float* vrow, float* frow,
                              hipStream_t st) {
    if (capacity == 0) return;
    TriParams p = make_params(s, gx, gy, 0, 0, vproj, nullptr, face_list, img);
    StageScope t(DMR_STAGE_TRI_BACKWARD_HITS, st);
    unsigned nblocks = (unsigned)(s.B * gx * gy);
#ifdef DMR_ABLATION
    if (DMR_DBG(p, 131072)) nblocks = std::min(nblocks, 3072u);  // timing experiment, as above
#endif
    k_tri_backward_hits<<<dim3(nblocks), dim3(256), 0, st>>>(p, pixrec, hits, capacity, vrow, frow);
}

void launch_tri_unpack(const dmr_scene& s, const float* vrow, const float* frow, float* dL_dverts,
                       float* dL_dvcolor, float* dL_dfopacity, float* dL_dvdepth, float* dL_dfintense,
                       hipStream_t st) {
    const int64_t n = s.P > s.F ? s.P : s.F;
    if (n == 0) return;
    StageScope t(DMR_STAGE_TRI_UNPACK, st);
    k_tri_unpack<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(
        s.B, s.P, s.F, vrow, frow, dL_dverts, dL_dvcolor, dL_dfopacity, dL_dvdepth, dL_dfintense);
}

}  // namespace dmr

#ifdef DMR_ABLATION
// ablation build only (not in include/dmesh_renderer_amd.h): copy out / clear the phase stamps.  which = 0: g_phase, 1: g_phase_rt
extern "C" __attribute__((visibility("default"))) long long dmr_debug_phase(int which, void* dst, long long bytes, int reset) {
    const size_t have = which == 0 ? sizeof(dmr::g_phase) : sizeof(dmr::g_phase_rt);
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (dst) {
        const size_t n = std::min((size_t)bytes, have);
        const hipError_t e = which == 0 ? hipMemcpyFromSymbol(dst, HIP_SYMBOL(dmr::g_phase), n)
                                        : hipMemcpyFromSymbol(dst, HIP_SYMBOL(dmr::g_phase_rt), n);
        if (e != hipSuccess) return -1;
    }
    if (reset) {
        void* sym = nullptr;
        const hipError_t e = which == 0 ? hipGetSymbolAddress(&sym, HIP_SYMBOL(dmr::g_phase))
                                        : hipGetSymbolAddress(&sym, HIP_SYMBOL(dmr::g_phase_rt));
        if (e != hipSuccess || hipMemset(sym, 0, have) != hipSuccess) return -1;
    }
    return (long long)have;
}
#endif
