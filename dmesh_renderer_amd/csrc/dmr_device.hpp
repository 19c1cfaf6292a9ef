// dmr_device.hpp -- device helpers shared by the gfx950 kernels.
//
// FP contract: the whole library is compiled with -ffp-contract=off and every
// expression below keeps the evaluation order of the reference helper it replaces,
// so tile indices, sort keys and coverage are decided by the same float bits as in
// the CPU oracle.  The double-precision islands of the reference (Q1 in SURVEY.md)
// are kept.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

// Timing ablations (scripts/prof_ablate.sh) and the forced-fallback switch of tests/test_fallback_gpu.py exist only in
// the ABLATION build (`python -m dmesh_renderer_amd.build --ablation`: -DDMR_ABLATION, a separate
// libdmesh_renderer_hip_ablation.so that reads the DMR_ABLATE environment variable).  The product library has no
// such switch: DMR_DBG folds to false, the kernels' parameter blocks have no debug field, nothing calls getenv.
#ifdef DMR_ABLATION
#define DMR_DBG(p, bits) (((p).dbg & (bits)) != 0)
#else
#define DMR_DBG(p, bits) false
#endif

namespace dmr {

// "Are all threads of the workgroup done?" with ONE barrier.  __syncthreads_and() compiles to ds_write / s_barrier / ds_and /
// s_barrier / ds_read / s_barrier on this target: >= 1.5 k cycles at the top of every chunk of a tile's list (phase stamps,
// profiles/r02/phase_times_c4.txt).  Here a thread that is not done raises word `parity` of a two-word LDS flag before the
// barrier, everybody reads it after, and thread 0 clears the other word for the next round.  The caller zeroes both words and
// passes a barrier once before the first call, and has at least one more barrier between two calls (the clear must not
// overtake the next round's raise).
struct AllDone {
    uint32_t* live;   // __shared__ uint32_t[2]
    uint32_t parity;
    __device__ __forceinline__ void init(uint32_t* s_live) {
        live = s_live; parity = 0u;
        if (threadIdx.x == 0) { live[0] = 0u; live[1] = 0u; }
        __syncthreads();
    }
    __device__ __forceinline__ bool barrier(bool done) {
        if (!done) live[parity] = 1u;
        __syncthreads();
        const bool all = live[parity] == 0u;
        parity ^= 1u;
        if (threadIdx.x == 0) live[parity] = 0u;
        return all;
    }
};

constexpr int TILE = 16;            // cuda_*/config.h:5-6 (BLOCK_X = BLOCK_Y = 16)
constexpr int TILE_PIX = 256;
constexpr float T_EPS = 0.0001f;    // auxiliary.h:8

struct V2 { float x, y; };
struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };

__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float b) { return {a.x * b, a.y * b, a.z * b}; }
__device__ __forceinline__ V3 operator*(float b, V3 a) { return {b * a.x, b * a.y, b * a.z}; }
__device__ __forceinline__ V3 operator/(V3 a, float b) { return {a.x / b, a.y / b, a.z / b}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// float -> int32, truncating, saturating, NaN -> 0 (v_cvt_i32_f32; same as CUDA's cvt.rzi)
__device__ __forceinline__ int f2i(float v) { return __float2int_rz(v); }

// auxiliary.h:33-36
__device__ __forceinline__ float ndc2pix(float v, int S) {
    return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5);
}
// auxiliary.h:38-41
__device__ __forceinline__ float pix2ndc(float v, int S) {
    return (float)((((double)v * 2.0 + 1.0) / (double)S) - 1.0);
}
// auxiliary.h:245-253
__device__ __forceinline__ float clamp_w(float w) {
    const float eps = 1e-4f;
    if (w >= 0 && w < eps) return eps;
    else if (w < 0 && w > -eps) return -eps;
    else return w;
}
// auxiliary.h:71-79
__device__ __forceinline__ V3 xform4x3(V3 p, const float* __restrict__ m) {
    return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
            m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
            m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
}
// auxiliary.h:81-90
__device__ __forceinline__ V4 xform4x4(V3 p, const float* __restrict__ m) {
    return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
            m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
            m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14],
            m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]};
}

// Tile rect of a projected triangle, auxiliary.h:55-69 (truncate toward zero, clamp to
// [0, grid]); rows additionally clipped to the band [r0, r1) of this shard.
struct Rect { uint32_t minx, miny, maxx, maxy; };
__device__ __forceinline__ Rect tile_rect(V2 p0, V2 p1, V2 p2, int gx, int gy, int r0, int r1) {
    Rect r;
    r.minx = (uint32_t)min(gx, max(0, f2i(fminf(fminf(p0.x, p1.x), p2.x) / (float)TILE)));
    r.miny = (uint32_t)min(gy, max(0, f2i(fminf(fminf(p0.y, p1.y), p2.y) / (float)TILE)));
    r.maxx = (uint32_t)min(gx, max(0, (int)((uint32_t)f2i(fmaxf(fmaxf(p0.x, p1.x), p2.x) / (float)TILE) + 1u)));
    r.maxy = (uint32_t)min(gy, max(0, (int)((uint32_t)f2i(fmaxf(fmaxf(p0.y, p1.y), p2.y) / (float)TILE) + 1u)));
    r.miny = max(r.miny, (uint32_t)r0);
    r.maxy = min(r.maxy, (uint32_t)r1);
    if (r.maxy < r.miny) r.maxy = r.miny;
    return r;
}

// Per-pixel world-space ray, tri generateRaysCUDA (cuda_rasterizer/forward.cu:184-231) and the
// seed <= 0 branch of the tet one (cuda_renderer/forward.cu:90-145).  Recomputed inside the
// compositing kernels instead of being stored (saves 24 B/pixel each way).
// Seeded ray jitter of the tet renderer (cuda_renderer/forward.cu:82-88,120-123): the reference draws two
// cuRAND XORWOW uniforms per pixel from a per-call cudaMalloc'ed state array, curand_init(seed, idx, 0).
// XORWOW's skip-ahead tables are not reproducible here, so the bits are NOT the reference's (parity unpinned);
// what is kept is the distribution and the mapping: u in (0, 1] -> pixel - 0.5 + 0.5 u (Q20).  The generator is
// counter based (Philox-4x32-10, key = seed, counter = pixel index), so the three kernels that need a pixel's
// ray recompute the same one without any state in memory.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t k0, uint32_t& o0, uint32_t& o1) {
    uint32_t c2 = 0u, c3 = 0u, k1 = 0u;
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o0 = c0; o1 = c1;
}
// curand_uniform's mapping of 32 random bits to (0, 1]
__device__ __forceinline__ float uniform_01(uint32_t x) { return (float)x * 2.3283064365386963e-10f + 1.1641532182693481e-10f; }

template <bool TET>
__device__ __forceinline__ void pixel_ray(const float* __restrict__ inv_mv, const float* __restrict__ inv_proj,
                                          int px, int py, int W, int H, V3& o, V3& d, int seed = 0, uint64_t idx = 0) {
    o = {inv_mv[12], inv_mv[13], inv_mv[14]};
    V2 pixf = {px + 0.5f, py + 0.5f};
    if (TET && seed > 0) {  // idx = (view * H + y) * W + x, the reference's thread rank
        uint32_t r0, r1;
        philox4x32_10((uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)seed, r0, r1);
        pixf.x = (float)px - 0.5f + (0.5f * uniform_01(r0));
        pixf.y = (float)py - 0.5f + (0.5f * uniform_01(r1));
    }
    V2 nd = {pix2ndc(pixf.x, W), pix2ndc(pixf.y, H)};
    V4 pv = xform4x4({nd.x, nd.y, -1.0f}, inv_proj);
    V4 pw = xform4x4({pv.x, pv.y, pv.z}, inv_mv);
    d = V3{pw.x, pw.y, pw.z} - o;
    float len;
    if (TET) { len = sqrtf(dot(d, d)); len = fmaxf(len, 0.0001f); }
    else len = sqrtf(dot(d, d)) + 0.0000001f;
    d = d / len;
}

// auxiliary.h:335-372
__device__ __forceinline__ void clamp_bary_uv(float u, float v, float& u_c, float& v_c, int& code) {
    if (u >= 0.0f && v >= 0.0f && u + v <= 1.0f) { u_c = u; v_c = v; code = 0; }
    else if (u <= 0.0f && v <= 0.0f) { u_c = 0.0f; v_c = 0.0f; code = 1; }
    else if ((u >= 1.0f && v <= 0.0f) || (v >= 0.0f && v <= u - 1.0f)) { u_c = 1.0f; v_c = 0.0f; code = 2; }
    else if ((u <= 0.0f && v >= 1.0f) || (u >= 0.0f && v >= u + 1.0f)) { u_c = 0.0f; v_c = 1.0f; code = 3; }
    else if (u <= 0.0f && v <= 1.0f && v >= 0.0f) { u_c = 0.0f; v_c = v; code = 4; }
    else if (u <= 1.0f && u >= 0.0f && v <= 0.0f) { u_c = u; v_c = 0.0f; code = 5; }
    else { u_c = (1.0f + u - v) * 0.5f; v_c = (1.0f - u + v) * 0.5f; code = 6; }
}

// auxiliary.h:374-400
__device__ __forceinline__ void clamp_bary_uv_grad(int code, float& duc_du, float& duc_dv, float& dvc_du, float& dvc_dv) {
    dvc_du = 0.0f; duc_dv = 0.0f;
    if (code == 0) { duc_du = 1.0f; dvc_dv = 1.0f; }
    else if (code == 1 || code == 2 || code == 3) { duc_du = 0.0f; dvc_dv = 0.0f; }
    else if (code == 4) { duc_du = 0.0f; dvc_dv = 1.0f; }
    else if (code == 5) { duc_du = 1.0f; dvc_dv = 0.0f; }
    else { duc_du = 0.5f; dvc_du = -0.5f; duc_dv = -0.5f; dvc_dv = 0.5f; }
}

// Moeller-Trumbore, tet flavour with the real hit test (cuda_renderer/auxiliary.h:265-296)
__device__ __forceinline__ bool ray_tri_hit(V3 o, V3 d, V3 p0, V3 p1, V3 p2, V3& tuv) {
    V3 T = o - p0, E1 = p1 - p0, E2 = p2 - p0;
    V3 P = cross(d, E2), Q = cross(T, E1);
    float denom = dot(P, E1);
    if (denom == 0.0f) return false;
    float inv_denom = 1.0f / denom;
    tuv.x = dot(Q, E2) * inv_denom;
    tuv.y = dot(P, T) * inv_denom;
    tuv.z = dot(Q, d) * inv_denom;
    return (tuv.x >= 0.0f && tuv.y >= 0.0f && tuv.z >= 0.0f && tuv.y + tuv.z <= 1.0f);
}

// ... with the two edge vectors E1 = p1 - p0, E2 = p2 - p0 given (the tet march's records keep them: the same subtractions,
// done once per tet and face instead of once per ray and step)
__device__ __forceinline__ bool ray_tri_hit_edges(V3 o, V3 d, V3 p0, V3 E1, V3 E2, V3& tuv) {
    V3 T = o - p0;
    V3 P = cross(d, E2), Q = cross(T, E1);
    float denom = dot(P, E1);
    if (denom == 0.0f) return false;
    float inv_denom = 1.0f / denom;
    tuv.x = dot(Q, E2) * inv_denom;
    tuv.y = dot(P, T) * inv_denom;
    tuv.z = dot(Q, d) * inv_denom;
    return (tuv.x >= 0.0f && tuv.y >= 0.0f && tuv.z >= 0.0f && tuv.y + tuv.z <= 1.0f);
}

__device__ __forceinline__ V3 load_v3(const float* __restrict__ a, int id) {
    return {a[3 * id], a[3 * id + 1], a[3 * id + 2]};
}

// cuda_renderer/auxiliary.h:345-394, split so that a march step computes the tet centre once for the
// four faces it tests (same arithmetic per face, so results are unchanged)
__device__ __forceinline__ V3 tet_center(const float* __restrict__ verts, const int* __restrict__ tets, int tet_idx) {
    V3 q0 = load_v3(verts, tets[4 * tet_idx]), q1 = load_v3(verts, tets[4 * tet_idx + 1]);
    V3 q2 = load_v3(verts, tets[4 * tet_idx + 2]), q3 = load_v3(verts, tets[4 * tet_idx + 3]);
    return (q0 + q1 + q2 + q3) * 0.25f;
}
__device__ __forceinline__ V3 face_outward_normal(V3 p0, V3 p1, V3 p2, V3 center) {
    V3 n = cross(p1 - p0, p2 - p0);
    float n_norm = sqrtf(dot(n, n));
    n_norm = fmaxf(n_norm, 0.0001f);
    n = n / n_norm;
    if (dot(n, center - p0) > 0.0f) n = -n;
    return n;
}
__device__ __forceinline__ V3 tet_face_outward_normal(const float* __restrict__ verts, const int* __restrict__ faces,
                                                      const int* __restrict__ tets, int face_idx, int tet_idx) {
    return face_outward_normal(load_v3(verts, faces[3 * face_idx]), load_v3(verts, faces[3 * face_idx + 1]),
                               load_v3(verts, faces[3 * face_idx + 2]), tet_center(verts, tets, tet_idx));
}

// ---------------------------------------------------------------------------
// Coverage set-up: the reference's in_tri (auxiliary.h:179-243) evaluates three 28.4
// fixed-point edge functions per (pixel, face).  Everything that does not depend on the
// pixel is hoisted here, once per staged face: with px = 16*x + 8, py = 16*y + 8 (the
// pixel centre (x + 0.5) * 16, exact in fp32) each edge function is
//     s_i(x, y) = s0_i + bx_i * (x - x0) + by_i * (y - y0)      (mod 2^32, as int32 wraps, Q7)
// for a tile whose first pixel is (x0, y0).  `ok` is false for zero-area faces.
// ---------------------------------------------------------------------------
struct EdgeSetup { int32_t s0[3], bx[3], by[3]; bool ok; int x0, x1, y0, y1; };  // box: tile-local pixels, inclusive

// The tile-independent half of that set-up: the
// three vertices snapped to 28.4 fixed point (auxiliary.h:190-195), the zero-area test (:201-202) and the winding swap
// (:203-212).
struct alignas(32) FaceCov { int32_t x1, y1, x2, y2, x3, y3; int32_t ok; int32_t pad; };

__device__ __forceinline__ FaceCov snap_face(V2 p1, V2 p2, V2 p3) {
    const float sub = 16.0f;
    uint32_t x1 = (uint32_t)f2i(p1.x * sub), y1 = (uint32_t)f2i(p1.y * sub);
    uint32_t x2 = (uint32_t)f2i(p2.x * sub), y2 = (uint32_t)f2i(p2.y * sub);
    uint32_t x3 = (uint32_t)f2i(p3.x * sub), y3 = (uint32_t)f2i(p3.y * sub);
    const int32_t area = (int32_t)((x2 - x1) * (y3 - y1) - (x3 - x1) * (y2 - y1));
    if (area < 0) { uint32_t t = x2; x2 = x3; x3 = t; t = y2; y2 = y3; y3 = t; }
    return FaceCov{(int32_t)x1, (int32_t)y1, (int32_t)x2, (int32_t)y2, (int32_t)x3, (int32_t)y3, area != 0 ? 1 : 0, 0};
}

// The tile-dependent half: edge functions at the tile's first pixel (x0, y0), their steps, the pixel box.
__device__ __forceinline__ EdgeSetup edge_setup(const FaceCov& f, int x0, int y0) {
    const uint32_t x1 = (uint32_t)f.x1, y1 = (uint32_t)f.y1, x2 = (uint32_t)f.x2, y2 = (uint32_t)f.y2, x3 = (uint32_t)f.x3, y3 = (uint32_t)f.y3;
    EdgeSetup e;
    e.ok = f.ok != 0;
    const uint32_t cx[3] = {x1 - x2, x2 - x3, x3 - x1};
    const uint32_t cy[3] = {y1 - y2, y2 - y3, y3 - y1};
    const uint32_t vx[3] = {x1, x2, x3};
    const uint32_t vy[3] = {y1, y2, y3};
    const uint32_t px = (uint32_t)(16 * x0 + 8), py = (uint32_t)(16 * y0 + 8);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        uint32_t s = cx[i] * (py - vy[i]) - cy[i] * (px - vx[i]);
        if ((int32_t)cy[i] > 0 || ((int32_t)cy[i] == 0 && (int32_t)cx[i] > 0)) s -= 1u;  // top-left rule
        e.s0[i] = (int32_t)s;
        e.bx[i] = (int32_t)(0u - 16u * cy[i]);
        e.by[i] = (int32_t)(16u * cx[i]);
    }
    // Pixel box.  A pixel centre (16x+8, 16y+8) can only pass the three edge tests if it lies inside the closed box of
    // the snapped vertices -- provided no int32 product wrapped.  With every vertex within D = 2^14 sub-pixels (1024 px)
    // of the tile's first pixel centre: |c| = |v_i - v_j| < 2D, |p - v| < D + 240 (the tile's other pixels), so
    // |s| <= 2 * 2D * (D + 240) = 1.09e9 < 2^31 (at D = 2^15 it is 4.3e9: a triangle spanning ~4000 px around a tile
    // could wrap and, in the reference, cover pixels outside its box).  Otherwise the whole tile is rasterised.
    const int32_t X[3] = {(int32_t)x1, (int32_t)x2, (int32_t)x3}, Y[3] = {(int32_t)y1, (int32_t)y2, (int32_t)y3};
    const int32_t ox = 16 * x0 + 8, oy = 16 * y0 + 8;  // centre of the tile's first pixel
    bool near = true;
#pragma unroll
    for (int i = 0; i < 3; i++) near = near && abs(X[i] - ox) < (1 << 14) && abs(Y[i] - oy) < (1 << 14) && abs(X[i]) < (1 << 29) && abs(Y[i]) < (1 << 29);
    e.x0 = 0; e.x1 = TILE - 1; e.y0 = 0; e.y1 = TILE - 1;
    if (near) {
        const int32_t minx = min(min(X[0], X[1]), X[2]) - ox, maxx = max(max(X[0], X[1]), X[2]) - ox;
        const int32_t miny = min(min(Y[0], Y[1]), Y[2]) - oy, maxy = max(max(Y[0], Y[1]), Y[2]) - oy;
        e.x0 = max(0, (minx + 15) >> 4); e.x1 = min(TILE - 1, maxx >> 4);   // ceil / floor of the centre index
        e.y0 = max(0, (miny + 15) >> 4); e.y1 = min(TILE - 1, maxy >> 4);
    }
    return e;
}
__device__ __forceinline__ EdgeSetup edge_setup(V2 p1, V2 p2, V2 p3, int x0, int y0) { return edge_setup(snap_face(p1, p2, p3), x0, y0); }

static_assert(sizeof(FaceCov) == 32, "FaceCov");

}  // namespace dmr
