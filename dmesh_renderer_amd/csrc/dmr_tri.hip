// dmr_tri.hip -- tri renderer: front-to-back alpha compositing, forward + backward (gfx950).
//
// Replaces TRI_FORWARD::renderCUDA (cuda_rasterizer/forward.cu:257-489) and
// TRI_BACKWARD::renderCUDA (cuda_rasterizer/backward.cu:9-421).
//
// One 256-thread workgroup (4 wave64) per 16x16 tile; wave w owns the 8x8 pixel quadrant
// (w & 1, w >> 1), lane l the pixel (l & 7, l >> 3) inside it.  The tile's depth-sorted face
// list is consumed in chunks staged through LDS.  Per chunk each wave runs two phases:
//
//   A. coverage  -- the face index is wave-uniform: every lane evaluates the three
//      fixed-point edge functions of face j for its own pixel from a 40-byte LDS record
//      (broadcast reads) and records the result as bit j of a per-lane bit mask.  All the
//      per-face work of the reference's in_tri (float->fixed conversion, winding swap, edge
//      deltas, top-left bias) was done once when the face was staged.
//   B. shading   -- every lane walks the set bits of ITS OWN mask in list order, so all
//      64 lanes do useful blending work on (generally different) faces at once instead of
//      a few lanes per face; face records are gathered from LDS with per-lane addresses
//      (112-byte stride = odd number of 16-byte slots, conflict-light for ds_read_b128).
//
// The pixel's result is the same sequence of blends as the reference's loop.  Rays are
// recomputed per pixel (not stored).  The backward walks the chunks from the back, adds each
// hit's 23 gradient components into per-face LDS accumulators (ds_add_f32), and flushes a
// chunk with packed atomics: 3 vertex rows + 1 face row per (tile, face) instead of the
// reference's 23 global atomics per (pixel, face).
#include <algorithm>
#include <cstdlib>

#include "dmr_kernels.hpp"

namespace dmr {

#ifndef DMR_COV_UNROLL
#define DMR_COV_UNROLL 1
#endif
constexpr int FWD_CHUNK = 128;
constexpr int BWD_CHUNK = 64;

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
    int B, P, F, W, H, gx, gy, r0, dbg;
    const float* verts; const int* faces; const float* verts_color; const float* faces_opacity;
    const float* inv_mv; const float* inv_proj; const float* faces_intense; const float* bg;
    const float4* vproj; const uint32_t* tile_offset; const uint32_t* face_list;
    float* final_T; float* final_prev_T; uint32_t* n_contrib;
    uint32_t* tile_hits; const uint32_t* hit_offset;
};

__device__ __forceinline__ void stage_face(const TriParams& p, int b, int face, int x0, int y0, V3 ray_o,
                                          CovRec& cov, ShadeRec& sh, int* vid) {
    const int v0 = p.faces[3 * face], v1 = p.faces[3 * face + 1], v2 = p.faces[3 * face + 2];
    const float4 a0 = p.vproj[(int64_t)b * p.P + v0];
    const float4 a1 = p.vproj[(int64_t)b * p.P + v1];
    const float4 a2 = p.vproj[(int64_t)b * p.P + v2];
    const V3 p0 = load_v3(p.verts, v0), p1 = load_v3(p.verts, v1), p2 = load_v3(p.verts, v2);
    const V3 c0 = load_v3(p.verts_color, v0), c1 = load_v3(p.verts_color, v1), c2 = load_v3(p.verts_color, v2);
    EdgeSetup e = edge_setup({a0.x, a0.y}, {a1.x, a1.y}, {a2.x, a2.y}, x0, y0);
#pragma unroll
    for (int i = 0; i < 3; i++) { cov.s0[i] = e.s0[i]; cov.bx[i] = e.bx[i]; cov.by[i] = e.by[i]; }
    const bool some = e.ok && e.x0 <= e.x1 && e.y0 <= e.y1;
    cov.flags = some ? (COV_VALID | e.x0 | (e.x1 << 4) | (e.y0 << 8) | (e.y1 << 12)) : 0;
    cov.pad0 = 0; cov.pad1 = 0;
    const V3 T = ray_o - p0, E1 = p1 - p0, E2 = p2 - p0;
    const V3 Q = cross(T, E1);
    sh.T[0] = T.x; sh.T[1] = T.y; sh.T[2] = T.z;
    sh.E1[0] = E1.x; sh.E1[1] = E1.y; sh.E1[2] = E1.z;
    sh.E2[0] = E2.x; sh.E2[1] = E2.y; sh.E2[2] = E2.z;
    sh.Q[0] = Q.x; sh.Q[1] = Q.y; sh.Q[2] = Q.z;
    sh.c0[0] = c0.x; sh.c0[1] = c0.y; sh.c0[2] = c0.z;
    sh.c1[0] = c1.x; sh.c1[1] = c1.y; sh.c1[2] = c1.z;
    sh.c2[0] = c2.x; sh.c2[1] = c2.y; sh.c2[2] = c2.z;
    sh.d0 = a0.w; sh.d1 = a1.w; sh.d2 = a2.w;
    sh.opacity = p.faces_opacity[face];
    sh.intense = p.faces_intense[(int64_t)b * p.F + face];
    if (vid) { vid[0] = v0; vid[1] = v1; vid[2] = v2; vid[3] = face; }
}

__device__ __forceinline__ void stage_null(CovRec& cov) {
    int4* q = reinterpret_cast<int4*>(&cov);
    q[0] = make_int4(0, 0, 0, 0); q[1] = make_int4(0, 0, 0, 0); q[2] = make_int4(0, 0, 0, 0);
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

template <int CHUNK>
__global__ void __launch_bounds__(256, 4)
k_tri_forward(TriParams p, float* __restrict__ out_color, float* __restrict__ out_depth) {
    constexpr int WORDS = CHUNK / 32;
    static_assert(WORDS == 4, "one 32-face block per wave");
    __shared__ CovRec s_cov[CHUNK];
    __shared__ ShadeRec s_shade[CHUNK];
    __shared__ uint32_t s_pm[TILE_PIX][WORDS];  // [tile-local pixel y*16+x][32-face word]: coverage bits of the chunk

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tx = blockIdx.x, ty = blockIdx.y + p.r0, b = blockIdx.z;
    const int lx = (wave & 1) * 8 + (lane & 7), ly = (wave >> 1) * 8 + (lane >> 3);
    const int px = tx * TILE + lx, py = ty * TILE + ly;
    const bool inside = px < p.W && py < p.H;
    const int64_t HW = (int64_t)p.H * p.W;
    const int64_t pix_id = (int64_t)p.W * py + px;

    V3 ro = {0, 0, 0}, rd = {0, 0, 0};
    if (inside) pixel_ray<false>(p.inv_mv + 16 * b, p.inv_proj + 16 * b, px, py, p.W, p.H, ro, rd);
    const V3 view_o = {p.inv_mv[16 * b + 12], p.inv_mv[16 * b + 13], p.inv_mv[16 * b + 14]};

    const int tile = (b * p.gy + ty) * p.gx + tx;
    const uint32_t begin = p.tile_offset[tile], end = p.tile_offset[tile + 1];

    float T = 1.0f, pT = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f;
    uint32_t last_contributor = 0, n_hits = 0;
    bool done = !inside;

    for (uint32_t base = begin; base < end; base += CHUNK) {
        if (__syncthreads_and(done)) break;  // also fences LDS reuse
        const int n = (int)min((uint32_t)CHUNK, end - base);
        if (tid < n && !(p.dbg & 32)) stage_face(p, b, (int)p.face_list[base + tid], tx * TILE, ty * TILE, view_o,
                                                 s_cov[tid], s_shade[tid], nullptr);
        else if (tid < CHUNK) stage_null(s_cov[tid]);
        *reinterpret_cast<uint4*>(&s_pm[tid][0]) = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
        if (!(p.dbg & 16)) rasterize_faces<CHUNK>(s_cov, n, tid, s_pm);  // A
        __syncthreads();
        uint32_t m[WORDS];
        {
            const uint4 mm = *reinterpret_cast<const uint4*>(&s_pm[ly * TILE + lx][0]);
            m[0] = mm.x; m[1] = mm.y; m[2] = mm.z; m[3] = mm.w;
        }
#pragma unroll
        for (int w = 0; w < WORDS; w++) if (done || (p.dbg & 8)) m[w] = 0;
        if (__all(done)) continue;  // wave-uniform
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
            if (denom == 0.0f) continue;  // "edge case": counted, not blended (forward.cu:429-430)
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
            n_hits++;
            if (T < T_EPS) { done = true; break; }  // blend first, test after (Q9)
        }
    }

    if (begin != end) {  // blended (pixel, face) pairs of the tile: sizes the backward's hit-record buffer
#pragma unroll
        for (int dlt = 32; dlt > 0; dlt >>= 1) n_hits += __shfl_xor(n_hits, dlt, 64);
        if (lane == 0 && n_hits) atomicAdd(&p.tile_hits[tile], n_hits);
    }
    if (inside) {
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

// DPP lane moves (VALU, no LDS traffic).  dpp_i: a lane whose source is outside its 16-lane row
// (row_shr) or outside the written rows (row_bcast, row_mask) keeps `old`.  dpp_f_any: same move, but the
// value of such lanes is unspecified (callers ignore it) -- no register initialisation, no hazard nops.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xF, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f_any(float src) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(src), CTRL, ROW_MASK, 0xF, true));
}
// 1-ulp reciprocal (v_rcp_f32): gradients are checked to 1e-4, the forward keeps IEEE division
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
constexpr int DPP_ROW_SHR = 0x110;      // + n, n = 1..15
constexpr int DPP_ROW_BCAST15 = 0x142;  // lane 15 of each row -> every lane of the next row
constexpr int DPP_ROW_BCAST31 = 0x143;  // lane 31 -> every lane of rows 2 and 3

// One level of the segmented inclusive scan over the 23 components: lanes whose source lane carries the
// same face key add the source's partial sum.  Hand-scheduled: hipcc turns the obvious
// "g += same ? dpp(g) : 0" into mov + nop + mov_dpp + cndmask + add per value; here it is
// v_cndmask_b32_dpp (select 0 / shifted value on VCC = "different key") + v_add_f32 -- two VALU per value.
// Lanes without a valid source read 0 (bound_ctrl:0); rows excluded by row_mask never write `t`, which
// is zeroed once per level.
#define DMR_SEG2(N, DPP) "v_cndmask_b32_dpp %[t], %[g" #N "], %[z], vcc " DPP "\n\tv_add_f32 %[g" #N "], %[g" #N "], %[t]\n\t"
#define DMR_SEG_LEVEL(DPP)                                                                                     \
    asm volatile("s_mov_b64 vcc, %[ns]\n\tv_mov_b32 %[t], 0\n\ts_nop 1\n\t"                                     \
                 DMR_SEG2(0, DPP) DMR_SEG2(1, DPP) DMR_SEG2(2, DPP) DMR_SEG2(3, DPP) DMR_SEG2(4, DPP) DMR_SEG2(5, DPP)  \
                 DMR_SEG2(6, DPP) DMR_SEG2(7, DPP) DMR_SEG2(8, DPP) DMR_SEG2(9, DPP) DMR_SEG2(10, DPP) DMR_SEG2(11, DPP) \
                 DMR_SEG2(12, DPP) DMR_SEG2(13, DPP) DMR_SEG2(14, DPP) DMR_SEG2(15, DPP) DMR_SEG2(16, DPP)          \
                 DMR_SEG2(17, DPP) DMR_SEG2(18, DPP) DMR_SEG2(19, DPP) DMR_SEG2(20, DPP) DMR_SEG2(21, DPP)          \
                 DMR_SEG2(22, DPP)                                                                             \
                 : [t] "=&v"(t), [g0] "+v"(g[0]), [g1] "+v"(g[1]), [g2] "+v"(g[2]), [g3] "+v"(g[3]), [g4] "+v"(g[4]),  \
                   [g5] "+v"(g[5]), [g6] "+v"(g[6]), [g7] "+v"(g[7]), [g8] "+v"(g[8]), [g9] "+v"(g[9]),             \
                   [g10] "+v"(g[10]), [g11] "+v"(g[11]), [g12] "+v"(g[12]), [g13] "+v"(g[13]), [g14] "+v"(g[14]),   \
                   [g15] "+v"(g[15]), [g16] "+v"(g[16]), [g17] "+v"(g[17]), [g18] "+v"(g[18]), [g19] "+v"(g[19]),   \
                   [g20] "+v"(g[20]), [g21] "+v"(g[21]), [g22] "+v"(g[22])                                      \
                 : [z] "v"(0.0f), [ns] "s"(ns)                                                                 \
                 : "vcc")

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void seg_scan_level(int k, float (&g)[NACC]) {
    static_assert(NACC == 23, "the asm lists 23 registers");
    const int ko = dpp_i<CTRL, ROW_MASK>((int)0x80000000, k);
    const uint64_t ns = __ballot(ko != k);  // lanes that must NOT add (different face, or no source lane)
    float t;
    if (CTRL == DPP_ROW_SHR + 1) DMR_SEG_LEVEL("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0");
    else if (CTRL == DPP_ROW_SHR + 2) DMR_SEG_LEVEL("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0");
    else if (CTRL == DPP_ROW_SHR + 4) DMR_SEG_LEVEL("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0");
    else if (CTRL == DPP_ROW_SHR + 8) DMR_SEG_LEVEL("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0");
    else if (CTRL == DPP_ROW_BCAST15) DMR_SEG_LEVEL("row_bcast:15 row_mask:0xa bank_mask:0xf bound_ctrl:0");
    else DMR_SEG_LEVEL("row_bcast:31 row_mask:0xc bank_mask:0xf bound_ctrl:0");
}

// ---------------------------------------------------------------------------
// backward, kernel 1 of 2: k_tri_backward_pix -- the per-pixel sequential part.
//
// Per chunk of 64 list entries (walked from the back of the tile list):
//   A. coverage as in the forward; every pixel gets a 64-bit mask `rem` of the chunk faces covering it
//      (positions >= its n_contrib masked off, backward.cu:192-194).
//   B. in list order from the back: recover T (Q10), the running accum_rec terms and dL/dalpha
//      (backward.cu:244-308).  At most BWD_SLOTS hits per pixel per pass; (T, dL_dalpha) of hit #h is
//      parked in s_pool[h][pixel].
//   C. the pass's hits are listed FACE-major (ballot transposes + block scan + scatter) and written out
//      as 16-byte HitRecords at the tile's offset (scan of the forward's per-tile hit counts).
// Everything that needs 23 accumulators per face happens in kernel 2, so this kernel keeps the forward's
// register/LDS footprint.  History of the fused versions (all measured at C4, see profiles/r01):
// 23 ds_add_f32 per hit: 59 % of wave cycles stalled on LDS issue (1.26 ms); one thread per
// (face, quadrant) accumulating in registers: ~15 % lane utilisation (1.25 ms); hit-parallel phase with a
// segmented scan inside this kernel: 168 VGPRs + 50 KB LDS -> 3 waves/SIMD, 0.56 ms.
// ---------------------------------------------------------------------------
#ifndef DMR_BWD_SLOTS
#define DMR_BWD_SLOTS 8
#endif
constexpr int BWD_SLOTS = DMR_BWD_SLOTS;

__global__ void __launch_bounds__(256)
k_tri_backward_pix(TriParams p, const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
                   float4* __restrict__ pixrec, HitRecord* __restrict__ hits, uint32_t capacity) {
    constexpr int CHUNK = 64;
    static_assert(BWD_CHUNK == CHUNK, "64-bit per-pixel masks, one wave scans the 64 face counters");
    __shared__ CovRec s_cov[CHUNK];
    __shared__ ShadeRec s_shade[CHUNK];
    __shared__ uint32_t s_fcnt[CHUNK];              // hits per face this pass
    __shared__ uint32_t s_fcur[CHUNK];              // claim cursor per face
    __shared__ uint32_t s_fstart[CHUNK + 1];        // exclusive scan of s_fcnt
    __shared__ float2 s_pool[BWD_SLOTS][TILE_PIX];  // [hit ordinal from the back][pixel] = (T, dL_dalpha)
    __shared__ uint32_t s_pm[TILE_PIX][CHUNK / 32]; // [tile-local pixel y*16+x][32-face word]: coverage bits of the chunk
    __shared__ uint32_t s_max_last;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tx = blockIdx.x, ty = blockIdx.y + p.r0, b = blockIdx.z;
    const int lx = (wave & 1) * 8 + (lane & 7), ly = (wave >> 1) * 8 + (lane >> 3);
    const int px = tx * TILE + lx, py = ty * TILE + ly;
    const bool inside = px < p.W && py < p.H;
    const int64_t HW = (int64_t)p.H * p.W;
    const int64_t pix_id = (int64_t)p.W * py + px;
    const int64_t bpix = (int64_t)b * HW + pix_id;

    const int tile = (b * p.gy + ty) * p.gx + tx;
    const uint32_t begin = p.tile_offset[tile], end = p.tile_offset[tile + 1];
    if (begin == end) return;  // uniform
    uint32_t hit_cursor = p.hit_offset[tile];
    if (hit_cursor == p.hit_offset[tile + 1]) return;  // no pixel of the tile blended anything

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
        // what kernel 2 needs of this pixel: ray direction and upstream gradient, two 16-byte gathers per hit
        pixrec[2 * bpix] = make_float4(rd.x, rd.y, rd.z, dpd);
        pixrec[2 * bpix + 1] = make_float4(dpc0, dpc1, dpc2, 0.f);
    }
    // backward.cu:293-298 (loop invariant there)
    float bg_dot = 0.f;
    bg_dot += p.bg[0] * dpc0; bg_dot += p.bg[1] * dpc1; bg_dot += p.bg[2] * dpc2;
    const float bd_dot = 0.f + (float)(1.0 * (double)dpd);

    if (tid == 0) s_max_last = 0;
    if (tid < CHUNK) s_fcnt[tid] = 0u;
    __syncthreads();
    if (last_contributor) atomicMax(&s_max_last, last_contributor);
    __syncthreads();
    const uint32_t total = s_max_last;  // list positions >= total contribute to no pixel of the tile
    if (total == 0) return;

    // pixel-thread state of the reverse walk
    float T = prev_T_final;
    bool first_pass = true;
    float acr0 = 0, acr1 = 0, acr2 = 0, acrd = 0;
    float last_alpha = 0, lc0 = 0, lc1 = 0, lc2 = 0, last_depth = 0;


    const uint32_t nchunks = (total + CHUNK - 1) / CHUNK;
    for (uint32_t ci = 0; ci < nchunks; ci++) {
        const uint32_t hi = total - ci * CHUNK;  // chunk = list positions [lo, hi)
        const uint32_t lo = hi > (uint32_t)CHUNK ? hi - CHUNK : 0u;
        const int n = (int)(hi - lo);
        __syncthreads();  // previous chunk is done with the LDS records
        if (tid < n)
            stage_face(p, b, (int)p.face_list[begin + lo + tid], tx * TILE, ty * TILE, view_o,
                       s_cov[tid], s_shade[tid], nullptr);
        else if (tid < CHUNK) stage_null(s_cov[tid]);
        *reinterpret_cast<uint2*>(&s_pm[tid][0]) = make_uint2(0u, 0u);
        __syncthreads();
        rasterize_faces<CHUNK>(s_cov, n, tid, s_pm);  // ---- A
        __syncthreads();
        uint64_t rem;
        {
            const uint2 mm = *reinterpret_cast<const uint2*>(&s_pm[ly * TILE + lx][0]);
            rem = (uint64_t)mm.x | ((uint64_t)mm.y << 32);
            const int64_t lim = (int64_t)last_contributor - (int64_t)lo;  // keep positions < last_contributor
            if (lim <= 0) rem = 0;
            else if (lim < 64) rem &= (1ull << lim) - 1ull;
        }

        while (__syncthreads_or(rem != 0ull)) {
            // ---- B: up to BWD_SLOTS hits of this pixel, from the back
            uint64_t pm = 0;
            int h = 0;
            if (p.dbg & 4) rem = 0ull;
            while (rem != 0ull && h < BWD_SLOTS) {
                const int k = 63 - __clzll((long long)rem);
                const uint64_t bit = 1ull << k;
                rem &= ~bit;
                const ShadeRec& r = s_shade[k];
                const V3 E1 = {r.E1[0], r.E1[1], r.E1[2]}, E2 = {r.E2[0], r.E2[1], r.E2[2]};
                const V3 Tv = {r.T[0], r.T[1], r.T[2]}, Q = {r.Q[0], r.Q[1], r.Q[2]};
                const V3 Pv = cross(rd, E2);
                const float denom = dot(Pv, E1);
                if (denom == 0.0f) continue;  // "edge case": skipped entirely (backward.cu:215-216)
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
                s_pool[h][tid] = make_float2(T, dL_dalpha);
                atomicAdd(&s_fcnt[k], 1u);
                pm |= bit;
                h++;
            }
            // ---- C: face-major slots.  Integer LDS atomics run at full rate on gfx950 (4 cycles per wave
            // instruction when conflict-free; ds_add_f32 takes ~200), so the (pixel, face) -> face-major
            // transposition is: count per face (B did that), scan the 64 counts, then every pixel claims a
            // slot per hit with a returning ds_add and writes its 16-byte record straight to the tile's region.
            __syncthreads();  // all counts of this pass are in
            if (wave == 0) {
                const int c = (lane < n) ? (int)s_fcnt[lane] : 0;
                int incl = c;
#pragma unroll
                for (int dlt = 1; dlt < 64; dlt <<= 1) {
                    const int o = __shfl_up(incl, dlt, 64);
                    if (lane >= dlt) incl += o;
                }
                s_fstart[lane] = (uint32_t)(incl - c);
                if (lane == 63) s_fstart[64] = (uint32_t)incl;
                s_fcnt[lane] = 0u;  // clean for the next pass
                s_fcur[lane] = 0u;
            }
            __syncthreads();
            const uint32_t Hp = s_fstart[64];
            {
                uint64_t m = pm;
                int hh = 0;
                const uint32_t pixel = (uint32_t)bpix;
                while (m != 0ull) {
                    const int k = 63 - __clzll((long long)m);
                    m &= ~(1ull << k);
                    const uint32_t slot = s_fstart[k] + atomicAdd(&s_fcur[k], 1u);
                    const float2 rec = s_pool[hh][tid];
                    hh++;
                    HitRecord hr;
                    hr.entry = begin + lo + (uint32_t)k;
                    hr.pixel = pixel;
                    hr.T = rec.x; hr.dL_dalpha = rec.y;
                    if (hit_cursor + slot < capacity) hits[hit_cursor + slot] = hr;  // capacity < total only while a size guess is being refuted
                }
            }
            hit_cursor += Hp;
            __syncthreads();  // claims done before wave 0 clears / rescans the counters in the next pass
        }
    }
}

// ---------------------------------------------------------------------------
// backward, kernel 2 of 2: k_tri_backward_hits -- one lane per blended (pixel, face) pair.
//
// Flat over the hit records (face-major inside every (tile, chunk, pass) group), no workgroup barriers.
// Each lane gathers its face and pixel data, recomputes the pixel-dependent geometry and the 23 gradient
// components of backward.cu:313-382; a segmented DPP wave scan keyed by the list entry leaves each
// entry's total in the last lane of its segment; segment totals are staged in wave-private LDS and flushed
// with PACKED global atomics: 3 vertex rows + 1 face row per segment (4 memory-side requests) instead of
// the reference's 23 global atomics per (pixel, face) (backward.cu:389-418).
// ---------------------------------------------------------------------------
constexpr int STAGE_SEGS = 16;  // segment totals staged per flush round and wave

__global__ void __launch_bounds__(256)
k_tri_backward_hits(TriParams p, const float4* __restrict__ pixrec, const HitRecord* __restrict__ hits,
                    const unsigned long long* __restrict__ hit_total, uint32_t capacity,
                    float* __restrict__ vrow, float* __restrict__ frow) {
    const uint32_t nhits = (uint32_t)min((unsigned long long)capacity, *hit_total);
    __shared__ float s_stage[4][STAGE_SEGS][28];  // 23 sums, v0, v1, v2, face, view
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t HW = (int64_t)p.H * p.W;
    const uint32_t stride = gridDim.x * 256u;
    for (uint32_t base = blockIdx.x * 256u + wave * 64u; base < nhits; base += stride) {
        const uint32_t hi_idx = base + lane;
        const bool valid = hi_idx < nhits;
        int k = -1 - lane;  // invalid lanes: unique keys
        int v0 = 0, v1 = 0, v2 = 0, face = 0, b = 0;
        float g[NACC];
#pragma unroll
        for (int c = 0; c < NACC; c++) g[c] = 0.f;
        if (valid) {
            const HitRecord hr = hits[hi_idx];
            k = (int)hr.entry;
            face = (int)p.face_list[hr.entry];
            b = (int)(hr.pixel / (uint32_t)HW);
            v0 = p.faces[3 * face]; v1 = p.faces[3 * face + 1]; v2 = p.faces[3 * face + 2];
            const float4 pr0 = pixrec[2 * (int64_t)hr.pixel], pr1 = pixrec[2 * (int64_t)hr.pixel + 1];
            const V3 d = {pr0.x, pr0.y, pr0.z};
            const float gdp = pr0.w, g0 = pr1.x, g1 = pr1.y, g2 = pr1.z;
            const V3 p0 = load_v3(p.verts, v0), p1 = load_v3(p.verts, v1), p2 = load_v3(p.verts, v2);
            const V3 cc0 = load_v3(p.verts_color, v0), cc1 = load_v3(p.verts_color, v1), cc2 = load_v3(p.verts_color, v2);
            const float fd0 = p.vproj[(int64_t)b * p.P + v0].w, fd1 = p.vproj[(int64_t)b * p.P + v1].w,
                        fd2 = p.vproj[(int64_t)b * p.P + v2].w;
            const float alpha = p.faces_opacity[face], intense = p.faces_intense[(int64_t)b * p.F + face];
            const V3 view_o = {p.inv_mv[16 * b + 12], p.inv_mv[16 * b + 13], p.inv_mv[16 * b + 14]};
            const V3 Tv = view_o - p0, E1 = p1 - p0, E2 = p2 - p0;
            const V3 Q = cross(Tv, E1);
            const float c00 = cc0.x, c01 = cc0.y, c02 = cc0.z, c10 = cc1.x, c11 = cc1.y, c12 = cc1.z;
            const float c20 = cc2.x, c21 = cc2.y, c22 = cc2.z;
            const float Th = hr.T;

            const V3 Pv = cross(d, E2);
            const float denom = dot(Pv, E1);
            const float inv_denom = fast_rcp(denom);
            const float nu = dot(Pv, Tv);
            const float iu = nu * inv_denom;
            const float iv = dot(Q, d) * inv_denom;
            float iuc, ivc; int code;
            clamp_bary_uv(iu, iv, iuc, ivc, code);
            const float i0 = 1 - iuc - ivc, i1 = iuc, i2 = ivc;
            const float dic0 = g0 * alpha * Th, dic1 = g1 * alpha * Th, dic2 = g2 * alpha * Th;
            const float did = gdp * alpha * Th;

            float dL_di0 = 0, dL_di1 = 0, dL_di2 = 0, dfint = 0;
            dL_di0 += c00 * dic0 * intense; dL_di1 += c10 * dic0 * intense; dL_di2 += c20 * dic0 * intense;
            dfint += (i0 * c00 + i1 * c10 + i2 * c20) * dic0;
            dL_di0 += c01 * dic1 * intense; dL_di1 += c11 * dic1 * intense; dL_di2 += c21 * dic1 * intense;
            dfint += (i0 * c01 + i1 * c11 + i2 * c21) * dic1;
            dL_di0 += c02 * dic2 * intense; dL_di1 += c12 * dic2 * intense; dL_di2 += c22 * dic2 * intense;
            dfint += (i0 * c02 + i1 * c12 + i2 * c22) * dic2;
            dL_di0 += fd0 * did; dL_di1 += fd1 * did; dL_di2 += fd2 * did;

            float duc_du, duc_dv, dvc_du, dvc_dv;
            clamp_bary_uv_grad(code, duc_du, duc_dv, dvc_du, dvc_dv);
            const float di0_diu = -1.f * duc_du + -1.f * dvc_du, di0_div = -1.f * duc_dv + -1.f * dvc_dv;
            const float di1_diu = 1.f * duc_du + 0.f * dvc_du, di1_div = 1.f * duc_dv + 0.f * dvc_dv;
            const float di2_diu = 0.f * duc_du + 1.f * dvc_du, di2_div = 0.f * duc_dv + 1.f * dvc_dv;
            const float dL_diu = dL_di0 * di0_diu + dL_di1 * di1_diu + dL_di2 * di2_diu;
            const float dL_div = dL_di0 * di0_div + dL_di1 * di1_div + dL_di2 * di2_div;

            // ray_tri_intersection_grad (auxiliary.h:288-333), Q11/Q12 kept
            const float dsq = denom, den2 = dsq * dsq, dinv = fast_rcp(den2);
            const float w0 = nu, w1 = dsq, w2 = dot(Q, E2);
            const V3 du_dE1 = (-1.0f * Pv * w0) * dinv;
            const V3 du_dE2 = (cross(Tv, d) * w1 - w0 * cross(E1, d)) * dinv;
            const V3 du_dT = (Pv * w1) * dinv;
            const V3 dv_dE1 = ((cross(E2, Tv) * w1) - (w2 * Pv)) * dinv;
            const V3 dv_dE2 = ((Q * w1) - (w2 * cross(E1, d))) * dinv;
            const V3 dv_dT = cross(E1, E2) * w1 * dinv;
            const V3 du_dp0 = -du_dE1 - du_dE2 - du_dT, dv_dp0 = -dv_dE1 - dv_dE2 - dv_dT;
            const V3 dp0 = dL_diu * du_dp0 + dL_div * dv_dp0;
            const V3 dp1 = dL_diu * du_dE1 + dL_div * dv_dE1;
            const V3 dp2 = dL_diu * du_dE2 + dL_div * dv_dE2;

            g[0] = dp0.x; g[1] = dp0.y; g[2] = dp0.z;
            g[3] = dp1.x; g[4] = dp1.y; g[5] = dp1.z;
            g[6] = dp2.x; g[7] = dp2.y; g[8] = dp2.z;
            g[9] = i0 * dic0 * intense; g[10] = i0 * dic1 * intense; g[11] = i0 * dic2 * intense;
            g[12] = i1 * dic0 * intense; g[13] = i1 * dic1 * intense; g[14] = i1 * dic2 * intense;
            g[15] = i2 * dic0 * intense; g[16] = i2 * dic1 * intense; g[17] = i2 * dic2 * intense;
            g[18] = i0 * did; g[19] = i1 * did; g[20] = i2 * did;
            g[21] = hr.dL_dalpha; g[22] = dfint;
        }
        // segmented inclusive scan over the wave (hits of one entry are consecutive lanes): four row-local
        // DPP levels, then the two row-broadcast levels of the classic wave scan
        seg_scan_level<DPP_ROW_SHR + 1, 0xF>(k, g);
        seg_scan_level<DPP_ROW_SHR + 2, 0xF>(k, g);
        seg_scan_level<DPP_ROW_SHR + 4, 0xF>(k, g);
        seg_scan_level<DPP_ROW_SHR + 8, 0xF>(k, g);
        seg_scan_level<DPP_ROW_BCAST15, 0xA>(k, g);
        seg_scan_level<DPP_ROW_BCAST31, 0xC>(k, g);

        // segment tails hold the totals; stage them (wave-private LDS) and flush with packed atomics
        const int kn = __shfl_down(k, 1, 64);
        const bool tail = valid && (lane == 63 || kn != k);
        const uint64_t tmask = __ballot(tail);
        const int ntail = __popcll(tmask);
        const int rank = __popcll(tmask & ((1ull << lane) - 1ull));
        for (int r0 = 0; r0 < ntail; r0 += STAGE_SEGS) {
            if (tail && rank >= r0 && rank < r0 + STAGE_SEGS) {
                float* st = s_stage[wave][rank - r0];
#pragma unroll
                for (int c = 0; c < NACC; c++) st[c] = g[c];
                st[23] = __int_as_float(v0); st[24] = __int_as_float(v1); st[25] = __int_as_float(v2);
                st[26] = __int_as_float(face); st[27] = __int_as_float(b);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int nseg = min(STAGE_SEGS, ntail - r0);
            // 32 lanes per segment = 3 vertex rows x 8 + face row x 8 (2 used); 2 segments per sweep
            const int sub = lane & 31, grp = sub >> 3, comp = sub & 7;
            for (int s0 = 0; s0 < nseg; s0 += 2) {
                const int sg = s0 + (lane >> 5);
                if (sg >= nseg || (p.dbg & 512)) continue;
                const float* st = s_stage[wave][sg];
                const int sb = __float_as_int(st[27]);
                if (grp < 3) {
                    if (comp == 7) continue;
                    const int ai = comp < 3 ? grp * 3 + comp : (comp < 6 ? 9 + grp * 3 + (comp - 3) : 18 + grp);
                    atomicAdd(&vrow[((int64_t)sb * p.P + __float_as_int(st[23 + grp])) * VROW + comp], st[ai]);
                } else if (comp < 2) {
                    atomicAdd(&frow[((int64_t)sb * p.F + __float_as_int(st[26])) * FROW + comp], st[21 + comp]);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

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

static TriParams make_params(const dmr_scene& s, int gx, int gy, int r0, const float4* vproj,
                             const uint32_t* tile_offset, const uint32_t* face_list, TriImageState img) {
    TriParams p;
    p.B = s.B; p.P = s.P; p.F = s.F; p.W = s.W; p.H = s.H; p.gx = gx; p.gy = gy; p.r0 = r0;
    { static const int dbg = getenv("DMR_ABLATE") ? atoi(getenv("DMR_ABLATE")) : 0; p.dbg = dbg; }  // timing ablations only
    p.verts = s.verts; p.faces = s.faces; p.verts_color = s.verts_color; p.faces_opacity = s.faces_opacity;
    p.inv_mv = s.inv_mv_mats; p.inv_proj = s.inv_proj_mats; p.faces_intense = s.faces_intense; p.bg = s.background;
    p.vproj = vproj; p.tile_offset = tile_offset; p.face_list = face_list;
    p.final_T = img.final_T; p.final_prev_T = img.final_prev_T; p.n_contrib = img.n_contrib;
    p.tile_hits = img.tile_hits; p.hit_offset = img.hit_offset;
    return p;
}

void launch_tri_forward(const dmr_scene& s, int gx, int gy, int r0, int r1, const float4* vproj,
                        const uint32_t* tile_offset, const uint32_t* face_list, TriImageState img,
                        float* out_color, float* out_depth, hipStream_t st) {
    if (r1 <= r0) return;
    TriParams p = make_params(s, gx, gy, r0, vproj, tile_offset, face_list, img);
    StageScope t(DMR_STAGE_TRI_FORWARD, st);
    k_tri_forward<FWD_CHUNK><<<dim3(gx, r1 - r0, s.B), dim3(256), 0, st>>>(p, out_color, out_depth);
}

void launch_tri_backward_pix(const dmr_scene& s, int gx, int gy, int r0, int r1, const float4* vproj,
                             const uint32_t* tile_offset, const uint32_t* face_list, TriImageState img,
                             const float* dL_dcolor, const float* dL_ddepth, float4* pixrec, HitRecord* hits,
                             uint32_t capacity, hipStream_t st) {
    if (r1 <= r0) return;
    TriParams p = make_params(s, gx, gy, r0, vproj, tile_offset, face_list, img);
    StageScope t(DMR_STAGE_TRI_BACKWARD, st);
    k_tri_backward_pix<<<dim3(gx, r1 - r0, s.B), dim3(256), 0, st>>>(p, dL_dcolor, dL_ddepth, pixrec, hits, capacity);
}

void launch_tri_backward_hits(const dmr_scene& s, const float4* vproj, const uint32_t* face_list,
                              const float4* pixrec, const HitRecord* hits, const unsigned long long* hit_total,
                              uint32_t capacity, float* vrow, float* frow, hipStream_t st) {
    if (capacity == 0) return;
    const uint32_t nhits = capacity;  // grid size from the host-known bound
    TriImageState none{nullptr, nullptr, nullptr, nullptr, nullptr};
    TriParams p = make_params(s, 0, 0, 0, vproj, nullptr, face_list, none);
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((nhits + 255u) / 256u, 256u * 16u);
    StageScope t(DMR_STAGE_TRI_BACKWARD_HITS, st);
    k_tri_backward_hits<<<dim3(blocks), dim3(256), 0, st>>>(p, pixrec, hits, hit_total, capacity, vrow, frow);
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
