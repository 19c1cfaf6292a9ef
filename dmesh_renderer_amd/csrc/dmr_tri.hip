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
#include "dmr_kernels.hpp"

namespace dmr {

constexpr int FWD_CHUNK = 128;
constexpr int BWD_CHUNK = 128;

struct alignas(16) CovRec {
    int32_t s0[3]; int32_t ok;
    int32_t bx[3]; int32_t pad0;
    int32_t by[3]; int32_t pad1;
};
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
    int B, P, F, W, H, gx, gy, r0;
    const float* verts; const int* faces; const float* verts_color; const float* faces_opacity;
    const float* inv_mv; const float* inv_proj; const float* faces_intense; const float* bg;
    const float4* vproj; const uint32_t* tile_offset; const uint32_t* face_list;
    float* final_T; float* final_prev_T; uint32_t* n_contrib;
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
    cov.ok = e.ok ? 1 : 0;
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

// phase A for one 32-face word of the chunk
__device__ __forceinline__ uint32_t coverage_word(const CovRec* __restrict__ cov, int count, int lx, int ly) {
    uint32_t m = 0;
    for (int j = 0; j < count; j++) {
        const CovRec& c = cov[j];
        bool in = edge_inside(c.s0, c.bx, c.by, lx, ly) && (c.ok != 0);
        m |= in ? (1u << j) : 0u;
    }
    return m;
}

template <int CHUNK>
__global__ void __launch_bounds__(256)
k_tri_forward(TriParams p, float* __restrict__ out_color, float* __restrict__ out_depth) {
    constexpr int WORDS = CHUNK / 32;
    __shared__ CovRec s_cov[CHUNK];
    __shared__ ShadeRec s_shade[CHUNK];

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
    uint32_t last_contributor = 0;
    bool done = !inside;

    for (uint32_t base = begin; base < end; base += CHUNK) {
        if (__syncthreads_and(done)) break;  // also fences LDS reuse
        const int n = (int)min((uint32_t)CHUNK, end - base);
        if (tid < n) stage_face(p, b, (int)p.face_list[base + tid], tx * TILE, ty * TILE, view_o,
                                s_cov[tid], s_shade[tid], nullptr);
        __syncthreads();
        if (__all(done)) continue;  // wave-uniform

        uint32_t m[WORDS];
#pragma unroll
        for (int w = 0; w < WORDS; w++) {
            const int cnt = min(32, max(0, n - 32 * w));
            m[w] = coverage_word(s_cov + 32 * w, cnt, lx, ly);
            if (done) m[w] = 0;
        }
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
            if (T < T_EPS) { done = true; break; }  // blend first, test after (Q9)
        }
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
// backward
// ---------------------------------------------------------------------------
constexpr int NACC = 23;  // 9 dverts, 9 dvcolor, 3 dvdepth, dopacity, dintense

template <int CHUNK>
__global__ void __launch_bounds__(256)
k_tri_backward(TriParams p, const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
               float* __restrict__ vrow, float* __restrict__ frow) {
    constexpr int WORDS = CHUNK / 32;
    __shared__ CovRec s_cov[CHUNK];
    __shared__ ShadeRec s_shade[CHUNK];
    __shared__ int s_vid[CHUNK][4];
    __shared__ float s_acc[NACC][CHUNK];  // component-major: lanes on different faces hit different banks
    __shared__ uint32_t s_touched[CHUNK];
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
    // backward.cu:293-298 (loop invariant there)
    float bg_dot = 0.f;
    bg_dot += p.bg[0] * dpc0; bg_dot += p.bg[1] * dpc1; bg_dot += p.bg[2] * dpc2;
    const float bd_dot = 0.f + (float)(1.0 * (double)dpd);

    if (tid == 0) s_max_last = 0;
    __syncthreads();
    if (last_contributor) atomicMax(&s_max_last, last_contributor);
    __syncthreads();
    const uint32_t max_last = s_max_last;  // entries at list positions >= max_last contribute nowhere
    if (max_last == 0) return;

    float T = prev_T_final;
    bool first_pass = true;
    float acr0 = 0, acr1 = 0, acr2 = 0, acrd = 0;
    float last_alpha = 0, lc0 = 0, lc1 = 0, lc2 = 0, last_depth = 0;

    const uint32_t total = max_last;                       // process list positions [0, total)
    const uint32_t nchunks = (total + CHUNK - 1) / CHUNK;
    for (uint32_t ci = 0; ci < nchunks; ci++) {
        // chunk covers positions [lo, hi), taken from the back
        const uint32_t hi = total - ci * CHUNK;
        const uint32_t lo = hi > (uint32_t)CHUNK ? hi - CHUNK : 0u;
        const int n = (int)(hi - lo);
        __syncthreads();  // previous chunk's flush done before LDS reuse
        if (tid < n) {
            stage_face(p, b, (int)p.face_list[begin + lo + tid], tx * TILE, ty * TILE, view_o,
                       s_cov[tid], s_shade[tid], s_vid[tid]);
            s_touched[tid] = 0;
        }
        for (int i = tid; i < NACC * CHUNK; i += 256) (&s_acc[0][0])[i] = 0.f;
        __syncthreads();

        uint32_t m[WORDS];
#pragma unroll
        for (int w = 0; w < WORDS; w++) {
            const int cnt = min(32, max(0, n - 32 * w));
            uint32_t mw = coverage_word(s_cov + 32 * w, cnt, lx, ly);
            // keep only positions < last_contributor (backward.cu:192-194)
            const int64_t lim = (int64_t)last_contributor - (int64_t)(lo + 32 * w);
            if (lim <= 0) mw = 0;
            else if (lim < 32) mw &= (1u << lim) - 1u;
            m[w] = mw;
        }
        while (true) {
            int w = -1; uint32_t mw = 0;
#pragma unroll
            for (int q = 0; q < WORDS; q++) if (m[q]) { w = q; mw = m[q]; }
            if (w < 0) break;
            const int bit = 31 - __clz((int)mw);
            const uint32_t clr = mw & ~(1u << bit);
#pragma unroll
            for (int q = 0; q < WORDS; q++) if (q == w) m[q] = clr;
            const int k = 32 * w + bit;
            const ShadeRec& r = s_shade[k];

            const V3 E1 = {r.E1[0], r.E1[1], r.E1[2]}, E2 = {r.E2[0], r.E2[1], r.E2[2]};
            const V3 Tv = {r.T[0], r.T[1], r.T[2]}, Q = {r.Q[0], r.Q[1], r.Q[2]};
            const V3 Pv = cross(rd, E2);
            const float denom = dot(Pv, E1);
            if (denom == 0.0f) continue;
            const float inv_denom = 1.0f / denom;
            const float nu = dot(Pv, Tv);          // v0 of the grad helper
            const float iu = nu * inv_denom;
            const float iv = dot(Q, rd) * inv_denom;
            float iuc, ivc; int code;
            clamp_bary_uv(iu, iv, iuc, ivc, code);
            const float i0 = 1 - iuc - ivc, i1 = iuc, i2 = ivc;
            const float intense = r.intense;
            const float c00 = r.c0[0], c01 = r.c0[1], c02 = r.c0[2];
            const float c10 = r.c1[0], c11 = r.c1[1], c12 = r.c1[2];
            const float c20 = r.c2[0], c21 = r.c2[1], c22 = r.c2[2];
            const float iC0 = (i0 * c00 + i1 * c10 + i2 * c20) * intense;
            const float iC1 = (i0 * c01 + i1 * c11 + i2 * c21) * intense;
            const float iC2 = (i0 * c02 + i1 * c12 + i2 * c22) * intense;
            const float iD = i0 * r.d0 + i1 * r.d1 + i2 * r.d2;
            const float alpha = r.opacity;

            if (!first_pass) T = T / (1.f - alpha);  // Q10
            first_pass = false;

            float dL_dalpha = 0.0f;
            acr0 = last_alpha * lc0 + (1.f - last_alpha) * acr0; lc0 = iC0;
            const float dic0 = dpc0 * alpha * T; dL_dalpha += (iC0 - acr0) * dpc0;
            acr1 = last_alpha * lc1 + (1.f - last_alpha) * acr1; lc1 = iC1;
            const float dic1 = dpc1 * alpha * T; dL_dalpha += (iC1 - acr1) * dpc1;
            acr2 = last_alpha * lc2 + (1.f - last_alpha) * acr2; lc2 = iC2;
            const float dic2 = dpc2 * alpha * T; dL_dalpha += (iC2 - acr2) * dpc2;
            acrd = last_alpha * last_depth + (1.f - last_alpha) * acrd; last_depth = iD;
            const float did = dpd * alpha * T; dL_dalpha += (iD - acrd) * dpd;
            dL_dalpha *= T;
            last_alpha = alpha;
            if (alpha == 1.0f) {
                dL_dalpha += (-prev_T_final) * bg_dot;
                dL_dalpha += (-prev_T_final) * bd_dot;
            } else {
                dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;
                dL_dalpha += (-T_final / (1.f - alpha)) * bd_dot;
            }

            float dL_di0 = 0, dL_di1 = 0, dL_di2 = 0, dfint = 0;
            dL_di0 += c00 * dic0 * intense; dL_di1 += c10 * dic0 * intense; dL_di2 += c20 * dic0 * intense;
            const float g00 = i0 * dic0 * intense, g10 = i1 * dic0 * intense, g20 = i2 * dic0 * intense;
            dfint += (i0 * c00 + i1 * c10 + i2 * c20) * dic0;
            dL_di0 += c01 * dic1 * intense; dL_di1 += c11 * dic1 * intense; dL_di2 += c21 * dic1 * intense;
            const float g01 = i0 * dic1 * intense, g11 = i1 * dic1 * intense, g21 = i2 * dic1 * intense;
            dfint += (i0 * c01 + i1 * c11 + i2 * c21) * dic1;
            dL_di0 += c02 * dic2 * intense; dL_di1 += c12 * dic2 * intense; dL_di2 += c22 * dic2 * intense;
            const float g02 = i0 * dic2 * intense, g12 = i1 * dic2 * intense, g22 = i2 * dic2 * intense;
            dfint += (i0 * c02 + i1 * c12 + i2 * c22) * dic2;
            dL_di0 += r.d0 * did; dL_di1 += r.d1 * did; dL_di2 += r.d2 * did;
            const float gd0 = i0 * did, gd1 = i1 * did, gd2 = i2 * did;

            float duc_du, duc_dv, dvc_du, dvc_dv;
            clamp_bary_uv_grad(code, duc_du, duc_dv, dvc_du, dvc_dv);
            const float di0_diu = -1.f * duc_du + -1.f * dvc_du, di0_div = -1.f * duc_dv + -1.f * dvc_dv;
            const float di1_diu = 1.f * duc_du + 0.f * dvc_du, di1_div = 1.f * duc_dv + 0.f * dvc_dv;
            const float di2_diu = 0.f * duc_du + 1.f * dvc_du, di2_div = 0.f * duc_dv + 1.f * dvc_dv;
            const float dL_diu = dL_di0 * di0_diu + dL_di1 * di1_diu + dL_di2 * di2_diu;
            const float dL_div = dL_di0 * di0_div + dL_di1 * di1_div + dL_di2 * di2_div;

            // ray_tri_intersection_grad (auxiliary.h:288-333), Q11/Q12 kept
            const float dsq = denom, den2 = dsq * dsq, dinv = 1.0f / den2;
            const float v0 = nu, v1 = dsq, v2 = dot(Q, E2);
            const V3 du_dE1 = (-1.0f * Pv * v0) * dinv;
            const V3 du_dE2 = (cross(Tv, rd) * v1 - v0 * cross(E1, rd)) * dinv;
            const V3 du_dT = (Pv * v1) * dinv;
            const V3 dv_dE1 = ((cross(E2, Tv) * v1) - (v2 * Pv)) * dinv;
            const V3 dv_dE2 = ((Q * v1) - (v2 * cross(E1, rd))) * dinv;
            const V3 dv_dT = cross(E1, E2) * v1 * dinv;
            const V3 du_dp0 = -du_dE1 - du_dE2 - du_dT, dv_dp0 = -dv_dE1 - dv_dE2 - dv_dT;
            const V3 dp0 = dL_diu * du_dp0 + dL_div * dv_dp0;
            const V3 dp1 = dL_diu * du_dE1 + dL_div * dv_dE1;
            const V3 dp2 = dL_diu * du_dE2 + dL_div * dv_dE2;

            atomicAdd(&s_acc[0][k], dp0.x); atomicAdd(&s_acc[1][k], dp0.y); atomicAdd(&s_acc[2][k], dp0.z);
            atomicAdd(&s_acc[3][k], dp1.x); atomicAdd(&s_acc[4][k], dp1.y); atomicAdd(&s_acc[5][k], dp1.z);
            atomicAdd(&s_acc[6][k], dp2.x); atomicAdd(&s_acc[7][k], dp2.y); atomicAdd(&s_acc[8][k], dp2.z);
            atomicAdd(&s_acc[9][k], g00); atomicAdd(&s_acc[10][k], g01); atomicAdd(&s_acc[11][k], g02);
            atomicAdd(&s_acc[12][k], g10); atomicAdd(&s_acc[13][k], g11); atomicAdd(&s_acc[14][k], g12);
            atomicAdd(&s_acc[15][k], g20); atomicAdd(&s_acc[16][k], g21); atomicAdd(&s_acc[17][k], g22);
            atomicAdd(&s_acc[18][k], gd0); atomicAdd(&s_acc[19][k], gd1); atomicAdd(&s_acc[20][k], gd2);
            atomicAdd(&s_acc[21][k], dL_dalpha); atomicAdd(&s_acc[22][k], dfint);
            s_touched[k] = 1u;
        }
        __syncthreads();

        // flush: 32 lanes per face = 3 vertex rows x 8 + face row x 8 (2 used); 8 faces per sweep
        const int sub = tid & 31, grp = sub >> 3, comp = sub & 7;
        for (int f0 = 0; f0 < n; f0 += 8) {
            const int k = f0 + (tid >> 5);
            if (k >= n || !s_touched[k]) continue;
            if (grp < 3) {
                if (comp == 7) continue;
                const int ai = comp < 3 ? grp * 3 + comp : (comp < 6 ? 9 + grp * 3 + (comp - 3) : 18 + grp);
                const float v = s_acc[ai][k];
                atomicAdd(&vrow[((int64_t)b * p.P + s_vid[k][grp]) * VROW + comp], v);
            } else if (comp < 2) {
                const float v = s_acc[21 + comp][k];
                atomicAdd(&frow[((int64_t)b * p.F + s_vid[k][3]) * FROW + comp], v);
            }
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
    p.verts = s.verts; p.faces = s.faces; p.verts_color = s.verts_color; p.faces_opacity = s.faces_opacity;
    p.inv_mv = s.inv_mv_mats; p.inv_proj = s.inv_proj_mats; p.faces_intense = s.faces_intense; p.bg = s.background;
    p.vproj = vproj; p.tile_offset = tile_offset; p.face_list = face_list;
    p.final_T = img.final_T; p.final_prev_T = img.final_prev_T; p.n_contrib = img.n_contrib;
    return p;
}

void launch_tri_forward(const dmr_scene& s, int gx, int gy, int r0, int r1, const float4* vproj,
                        const uint32_t* tile_offset, const uint32_t* face_list, TriImageState img,
                        float* out_color, float* out_depth, hipStream_t st) {
    if (r1 <= r0) return;
    TriParams p = make_params(s, gx, gy, r0, vproj, tile_offset, face_list, img);
    k_tri_forward<FWD_CHUNK><<<dim3(gx, r1 - r0, s.B), dim3(256), 0, st>>>(p, out_color, out_depth);
}

void launch_tri_backward(const dmr_scene& s, int gx, int gy, int r0, int r1, const float4* vproj,
                         const uint32_t* tile_offset, const uint32_t* face_list, TriImageState img,
                         const float* dL_dcolor, const float* dL_ddepth, float* vrow, float* frow, hipStream_t st) {
    if (r1 <= r0) return;
    TriParams p = make_params(s, gx, gy, r0, vproj, tile_offset, face_list, img);
    k_tri_backward<BWD_CHUNK><<<dim3(gx, r1 - r0, s.B), dim3(256), 0, st>>>(p, dL_dcolor, dL_ddepth, vrow, frow);
}

void launch_tri_unpack(const dmr_scene& s, const float* vrow, const float* frow, float* dL_dverts,
                       float* dL_dvcolor, float* dL_dfopacity, float* dL_dvdepth, float* dL_dfintense,
                       hipStream_t st) {
    const int64_t n = s.P > s.F ? s.P : s.F;
    if (n == 0) return;
    k_tri_unpack<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(
        s.B, s.P, s.F, vrow, frow, dL_dverts, dL_dvcolor, dL_dfopacity, dL_dvdepth, dL_dfintense);
}

}  // namespace dmr
