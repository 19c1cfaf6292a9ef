// dmr_oracle.cpp -- CPU restatement of the dmesh_renderer hot path.
//
// TEST INFRASTRUCTURE ONLY (see dmr_oracle.h).  PARITY UNPINNED at kernel level:
// the reference has no tests / golden vectors and cannot be built here (CUDA only).
//
// Every function below restates one reference kernel/helper, stage for stage,
// with the same evaluation order.  Build with -ffp-contract=off so that no
// a*b+c is fused: the HIP kernels are written to the same contract, which is
// what makes tile/sort indices comparable bit for bit.
//
// Citations are relative to the reference tree (SonSang/dmesh_renderer).

#include "dmr_oracle.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <numeric>
#include <string>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

thread_local std::string g_err;

// ---------------------------------------------------------------------------
// own float2/float3 ops (same component order as the reference's math header:
// dot = x*x' + y*y' + z*z' left to right; cross = (ay*bz-az*by, az*bx-ax*bz, ax*by-ay*bx))
// ---------------------------------------------------------------------------
struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

inline f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }
inline f3 operator*(f3 a, float b) { return {a.x * b, a.y * b, a.z * b}; }
inline f3 operator*(float b, f3 a) { return {b * a.x, b * a.y, b * a.z}; }
inline f3 operator/(f3 a, float b) { return {a.x / b, a.y / b, a.z / b}; }
inline float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline f3 cross(f3 a, f3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

constexpr float T_EPS = 0.0001f;  // auxiliary.h:8
constexpr int BLOCK_X = 16, BLOCK_Y = 16;  // config.h:5-6

// CUDA float->int conversion (cvt.rzi.s32.f32): truncate, saturate, NaN -> 0.
inline int f2i(float v) {
    if (v != v) return 0;
    if (v >= 2147483648.0f) return INT_MAX;
    if (v <= -2147483648.0f) return INT_MIN;
    return (int)v;
}

// auxiliary.h:33-36 (double-precision island, Q1)
inline float ndc2Pix(float v, int S) { return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5); }
// auxiliary.h:38-41
inline float pix2Ndc(float v, int S) { return (float)((((double)v * 2.0 + 1.0) / (double)S) - 1.0); }

// auxiliary.h:245-253
inline float clamp_w(float w) {
    const float eps = 1e-4f;
    if (w >= 0 && w < eps) return eps;
    else if (w < 0 && w > -eps) return -eps;
    else return w;
}

// auxiliary.h:71-79
inline f3 transformPoint4x3(f3 p, const float* m) {
    return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
            m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
            m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
}
// auxiliary.h:81-90
inline f4 transformPoint4x4(f3 p, const float* m) {
    return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
            m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
            m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14],
            m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]};
}

struct Rect { uint32_t minx, miny, maxx, maxy; };

// auxiliary.h:55-69 (Q4: truncation toward zero, then clamp to [0, grid])
inline Rect getRectFromTri(f2 p0, f2 p1, f2 p2, int gx, int gy) {
    Rect r;
    r.minx = (uint32_t)std::min(gx, std::max(0, f2i(fminf(fminf(p0.x, p1.x), p2.x) / (float)BLOCK_X)));
    r.miny = (uint32_t)std::min(gy, std::max(0, f2i(fminf(fminf(p0.y, p1.y), p2.y) / (float)BLOCK_Y)));
    // f2i(...) + 1 wraps like the device's 32-bit add (INT_MAX + 1 -> INT_MIN -> clamped to 0)
    auto inc = [](int v) { return (int)((uint32_t)v + 1u); };
    r.maxx = (uint32_t)std::min(gx, std::max(0, inc(f2i(fmaxf(fmaxf(p0.x, p1.x), p2.x) / (float)BLOCK_X))));
    r.maxy = (uint32_t)std::min(gy, std::max(0, inc(f2i(fmaxf(fmaxf(p0.y, p1.y), p2.y) / (float)BLOCK_Y))));
    return r;
}

// auxiliary.h:179-243.  28.4 fixed point, top-left rule, winding agnostic.
// int32 products wrap (Q7): computed in uint32_t to avoid UB.
inline bool in_tri(f2 p, f2 p1, f2 p2, f2 p3) {
    const float subpixel = 16.0f;
    uint32_t px = (uint32_t)f2i(p.x * subpixel), py = (uint32_t)f2i(p.y * subpixel);
    uint32_t x1 = (uint32_t)f2i(p1.x * subpixel), y1 = (uint32_t)f2i(p1.y * subpixel);
    uint32_t x2 = (uint32_t)f2i(p2.x * subpixel), y2 = (uint32_t)f2i(p2.y * subpixel);
    uint32_t x3 = (uint32_t)f2i(p3.x * subpixel), y3 = (uint32_t)f2i(p3.y * subpixel);

    int32_t area = (int32_t)((x2 - x1) * (y3 - y1) - (x3 - x1) * (y2 - y1));
    if (area == 0) return false;
    else if (area < 0) { std::swap(x2, x3); std::swap(y2, y3); }

    uint32_t cx1 = x1 - x2, cy1 = y1 - y2;
    uint32_t cx2 = x2 - x3, cy2 = y2 - y3;
    uint32_t cx3 = x3 - x1, cy3 = y3 - y1;
    uint32_t px1 = px - x1, py1 = py - y1;
    uint32_t px2 = px - x2, py2 = py - y2;
    uint32_t px3 = px - x3, py3 = py - y3;

    uint32_t s1 = cx1 * py1 - cy1 * px1;
    uint32_t s2 = cx2 * py2 - cy2 * px2;
    uint32_t s3 = cx3 * py3 - cy3 * px3;

    if ((int32_t)cy1 > 0 || ((int32_t)cy1 == 0 && (int32_t)cx1 > 0)) s1 -= 1;
    if ((int32_t)cy2 > 0 || ((int32_t)cy2 == 0 && (int32_t)cx2 > 0)) s2 -= 1;
    if ((int32_t)cy3 > 0 || ((int32_t)cy3 == 0 && (int32_t)cx3 > 0)) s3 -= 1;
    return ((int32_t)s1 < 0) && ((int32_t)s2 < 0) && ((int32_t)s3 < 0);
}

// auxiliary.h:255-286 (tri: never rejects, Q8) / cuda_renderer/auxiliary.h:265-296 (tet: real test)
template <bool TET>
inline bool ray_tri_intersection(f3 o, f3 d, f3 p0, f3 p1, f3 p2, f3& tuv) {
    f3 T = o - p0, E1 = p1 - p0, E2 = p2 - p0;
    f3 P = cross(d, E2), Q = cross(T, E1);
    float denom = dot(P, E1);
    if (denom == 0.0f) return false;
    float inv_denom = 1.0f / denom;
    tuv.x = dot(Q, E2) * inv_denom;
    tuv.y = dot(P, T) * inv_denom;
    tuv.z = dot(Q, d) * inv_denom;
    if (TET) return (tuv.x >= 0.0f && tuv.y >= 0.0f && tuv.z >= 0.0f && tuv.y + tuv.z <= 1.0f);
    return true;
}

// auxiliary.h:288-333 (Q11: "dv" uses t's numerator; Q12: clamp of denom is dead code)
inline void ray_tri_intersection_grad(f3 o, f3 d, f3 p0, f3 p1, f3 p2,
                                      f3& du_dp0, f3& du_dp1, f3& du_dp2,
                                      f3& dv_dp0, f3& dv_dp1, f3& dv_dp2) {
    f3 T = o - p0, E1 = p1 - p0, E2 = p2 - p0;
    float denom_sqrt = dot(cross(d, E2), E1);
    float denom = denom_sqrt * denom_sqrt;
    float denom_inv = 1.0f / denom;
    float v0 = dot(cross(d, E2), T);
    float v1 = denom_sqrt;
    float v2 = dot(cross(T, E1), E2);
    f3 du_dE1 = (-1.0f * cross(d, E2) * v0) * denom_inv;
    f3 du_dE2 = (cross(T, d) * v1 - v0 * cross(E1, d)) * denom_inv;
    f3 du_dT = (cross(d, E2) * v1) * denom_inv;
    f3 dv_dE1 = ((cross(E2, T) * v1) - (v2 * cross(d, E2))) * denom_inv;
    f3 dv_dE2 = ((cross(T, E1) * v1) - (v2 * cross(E1, d))) * denom_inv;
    f3 dv_dT = cross(E1, E2) * v1 * denom_inv;
    du_dp0 = -du_dE1 - du_dE2 - du_dT;
    dv_dp0 = -dv_dE1 - dv_dE2 - dv_dT;
    du_dp1 = du_dE1; dv_dp1 = dv_dE1;
    du_dp2 = du_dE2; dv_dp2 = dv_dE2;
}

// The same function evaluated in double (inputs promoted, outputs rounded once): not a reference behaviour.  It
// measures how much of a dL_dverts entry is float rounding noise of the reference's own arithmetic -- cross(T, d) loses
// |T| / (distance of p0 from the ray) ~ 1e2..1e4 in relative precision -- which bounds how closely ANY other float
// evaluation order can agree with the restatement above (dmro_set_tri_grad_f64, tests/tools/grad_noise.py).
struct d3 { double x, y, z; };
inline d3 operator+(d3 a, d3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline d3 operator-(d3 a, d3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline d3 operator-(d3 a) { return {-a.x, -a.y, -a.z}; }
inline d3 operator*(d3 a, double b) { return {a.x * b, a.y * b, a.z * b}; }
inline d3 operator*(double b, d3 a) { return {b * a.x, b * a.y, b * a.z}; }
inline double dot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline d3 cross(d3 a, d3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline d3 up(f3 a) { return {a.x, a.y, a.z}; }
inline void tri_dverts_f64(f3 o_, f3 d_, f3 p0_, f3 p1_, f3 p2_, float dL_diu, float dL_div, d3& dp0, d3& dp1, d3& dp2) {
    const d3 o = up(o_), d = up(d_), p0 = up(p0_), p1 = up(p1_), p2 = up(p2_);
    const d3 T = o - p0, E1 = p1 - p0, E2 = p2 - p0;
    const double v1 = dot(cross(d, E2), E1), denom_inv = 1.0 / (v1 * v1);
    const double v0 = dot(cross(d, E2), T), v2 = dot(cross(T, E1), E2);
    const d3 du_dE1 = (-1.0 * cross(d, E2) * v0) * denom_inv;
    const d3 du_dE2 = (cross(T, d) * v1 - v0 * cross(E1, d)) * denom_inv;
    const d3 du_dT = (cross(d, E2) * v1) * denom_inv;
    const d3 dv_dE1 = ((cross(E2, T) * v1) - (v2 * cross(d, E2))) * denom_inv;
    const d3 dv_dE2 = ((cross(T, E1) * v1) - (v2 * cross(E1, d))) * denom_inv;
    const d3 dv_dT = cross(E1, E2) * v1 * denom_inv;
    const d3 du_dp0 = -du_dE1 - du_dE2 - du_dT, dv_dp0 = -dv_dE1 - dv_dE2 - dv_dT;
    dp0 = (double)dL_diu * du_dp0 + (double)dL_div * dv_dp0;
    dp1 = (double)dL_diu * du_dE1 + (double)dL_div * dv_dE1;
    dp2 = (double)dL_diu * du_dE2 + (double)dL_div * dv_dE2;
}
static int g_tri_grad_f64 = 0;

// auxiliary.h:335-372
inline void clamp_bary_uv(float u, float v, float& u_c, float& v_c, int& code) {
    if (u >= 0.0f && v >= 0.0f && u + v <= 1.0f) { u_c = u; v_c = v; code = 0; }
    else if (u <= 0.0f && v <= 0.0f) { u_c = 0.0f; v_c = 0.0f; code = 1; }
    else if ((u >= 1.0f && v <= 0.0f) || (v >= 0.0f && v <= u - 1.0f)) { u_c = 1.0f; v_c = 0.0f; code = 2; }
    else if ((u <= 0.0f && v >= 1.0f) || (u >= 0.0f && v >= u + 1.0f)) { u_c = 0.0f; v_c = 1.0f; code = 3; }
    else if (u <= 0.0f && v <= 1.0f && v >= 0.0f) { u_c = 0.0f; v_c = v; code = 4; }
    else if (u <= 1.0f && u >= 0.0f && v <= 0.0f) { u_c = u; v_c = 0.0f; code = 5; }
    else { u_c = (1.0f + u - v) * 0.5f; v_c = (1.0f - u + v) * 0.5f; code = 6; }
}

// auxiliary.h:374-400
inline void clamp_bary_uv_grad(int code, float& duc_du, float& duc_dv, float& dvc_du, float& dvc_dv) {
    dvc_du = 0.0f; duc_dv = 0.0f;
    if (code == 0) { duc_du = 1.0f; dvc_dv = 1.0f; }
    else if (code == 1 || code == 2 || code == 3) { duc_du = 0.0f; dvc_dv = 0.0f; }
    else if (code == 4) { duc_du = 0.0f; dvc_dv = 1.0f; }
    else if (code == 5) { duc_du = 1.0f; dvc_dv = 0.0f; }
    else { duc_du = 0.5f; dvc_du = -0.5f; duc_dv = -0.5f; dvc_dv = 0.5f; }
}

inline f3 get_vert(const float* verts, int id) { return {verts[3 * id], verts[3 * id + 1], verts[3 * id + 2]}; }
inline void get_face_vert(const float* verts, const int* faces, int id, f3& p0, f3& p1, f3& p2) {
    p0 = get_vert(verts, faces[3 * id]);
    p1 = get_vert(verts, faces[3 * id + 1]);
    p2 = get_vert(verts, faces[3 * id + 2]);
}

// cuda_renderer/auxiliary.h:345-394
inline f3 tet_face_outward_normal(const float* verts, const int* faces, const int* tets, int face_idx, int tet_idx) {
    f3 p0, p1, p2;
    get_face_vert(verts, faces, face_idx, p0, p1, p2);
    f3 d1 = p1 - p0, d2 = p2 - p0;
    f3 n = cross(d1, d2);
    float n_norm = sqrtf(dot(n, n));
    n_norm = fmaxf(n_norm, 0.0001f);
    n = n / n_norm;
    f3 q0 = get_vert(verts, tets[4 * tet_idx]), q1 = get_vert(verts, tets[4 * tet_idx + 1]);
    f3 q2 = get_vert(verts, tets[4 * tet_idx + 2]), q3 = get_vert(verts, tets[4 * tet_idx + 3]);
    f3 center = (q0 + q1 + q2 + q3) * 0.25f;
    f3 d = center - p0;
    if (dot(n, d) > 0.0f) n = -n;
    return n;
}

// rasterizer_impl.cu:25-40
inline uint32_t getHigherMsb(uint32_t n) {
    uint32_t msb = sizeof(n) * 4;
    uint32_t step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step;
        else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

}  // namespace

// ---------------------------------------------------------------------------
// state = the reference's four scratch buffers, as named arrays
// (rasterizer_impl.h:18-58, renderer_impl.h:18-66)
// ---------------------------------------------------------------------------
struct dmro_state {
    int B = 0, P = 0, F = 0, T = 0, W = 0, H = 0, gx = 0, gy = 0, row_begin = 0, row_end = 0;
    bool tet = false;
    int64_t R = 0;
    std::vector<float> ndc, image;                    // VertState
    std::vector<float> depths, min_depths, max_depths;  // FaceState
    std::vector<uint32_t> tiles_touched, face_offsets;
    std::vector<uint64_t> keys;                       // BinningState (sorted)
    std::vector<uint32_t> values;
    std::vector<uint32_t> ranges;                     // ImageState
    std::vector<float> ray_o, ray_d, final_T, final_prev_T;
    std::vector<uint32_t> n_contrib;
    std::vector<int32_t> first_face, first_tet, last_face, last_tet;
    std::vector<uint8_t> is_active;
};

namespace {

bool check_scene(const dmro_scene* s, bool tet) {
    if (!s) { g_err = "null scene"; return false; }
    if (s->B <= 0 || s->W <= 0 || s->H <= 0 || s->P < 0 || s->F < 0) { g_err = "bad dimensions"; return false; }
    return true;
}

void band_of(const dmro_scene* s, int gy, int& r0, int& r1) {
    r0 = s->row_begin; r1 = s->row_end;
    if (r0 == 0 && r1 == 0) r1 = gy;
    r0 = std::max(0, std::min(gy, r0));
    r1 = std::max(r0, std::min(gy, r1));
}

// preprocessPointCUDA: cuda_rasterizer/forward.cu:17-47 (tet copy cuda_renderer/forward.cu:21-52)
void preprocess_point(const dmro_scene* s, dmro_state* st) {
    const int B = s->B, P = s->P;
    st->ndc.assign((size_t)B * P * 3, 0.f);
    st->image.assign((size_t)B * P * 2, 0.f);
#pragma omp parallel for schedule(static)
    for (int64_t idx = 0; idx < (int64_t)B * P; idx++) {
        int b = (int)(idx / P), p = (int)(idx % P);
        const float* mv = s->mv_mats + 16 * b;
        const float* pr = s->proj_mats + 16 * b;
        f3 p_view = transformPoint4x3({s->verts[3 * p], s->verts[3 * p + 1], s->verts[3 * p + 2]}, mv);
        f4 p_proj = transformPoint4x4(p_view, pr);
        float p_w = (float)(1.0 / (double)clamp_w(p_proj.w));  // forward.cu:38 (double divide, Q1)
        f3 n = {p_proj.x * p_w, p_proj.y * p_w, p_proj.z * p_w};
        st->ndc[3 * idx] = n.x; st->ndc[3 * idx + 1] = n.y; st->ndc[3 * idx + 2] = n.z;
        st->image[2 * idx] = ndc2Pix(n.x, s->W);
        st->image[2 * idx + 1] = ndc2Pix(n.y, s->H);
    }
}

inline Rect band_clip(Rect r, int r0, int r1) {
    r.miny = std::max<uint32_t>(r.miny, (uint32_t)r0);
    r.maxy = std::min<uint32_t>(r.maxy, (uint32_t)r1);
    if (r.maxy < r.miny) r.maxy = r.miny;
    return r;
}

// preprocessFaceCUDA: tri forward.cu:76-149; tet cuda_renderer/forward.cu:178-260
void preprocess_face(const dmro_scene* s, dmro_state* st) {
    const int B = s->B, P = s->P, F = s->F;
    const size_t BF = (size_t)B * F;
    st->depths.assign(BF, 0.f);
    st->tiles_touched.assign(BF, 0u);
    if (st->tet) { st->min_depths.assign(BF, 0.f); st->max_depths.assign(BF, 0.f); }
#pragma omp parallel for schedule(static)
    for (int64_t idx = 0; idx < (int64_t)BF; idx++) {
        int b = (int)(idx / F), f = (int)(idx % F);
        float max_z = 0, min_z = 0, depth = 0;
        f2 img[3];
        for (int i = 0; i < 3; i++) {
            int v = s->faces[3 * f + i];
            size_t bv = (size_t)b * P + v;
            float z = st->ndc[3 * bv + 2];
            if (i == 0) { max_z = z; min_z = z; }
            else { max_z = fmaxf(max_z, z); min_z = fminf(min_z, z); }
            depth += z;
            img[i] = {st->image[2 * bv], st->image[2 * bv + 1]};
        }
        depth = depth / 3.0f;
        if (max_z < -1.0f || min_z > 1.0f) continue;  // Q3
        Rect r = band_clip(getRectFromTri(img[0], img[1], img[2], st->gx, st->gy), st->row_begin, st->row_end);
        if ((r.maxx - r.minx) * (r.maxy - r.miny) == 0) continue;
        st->tiles_touched[idx] = (r.maxy - r.miny) * (r.maxx - r.minx);
        auto map01 = [](float z) { float d = (z + 1.0f) * 0.5f; if (d < 0.0f) d = 0.0f; if (d > 1.0f) d = 1.0f; return d; };
        st->depths[idx] = map01(depth);  // Q5
        if (st->tet) { st->min_depths[idx] = map01(min_z); st->max_depths[idx] = map01(max_z); }
    }
}

// cub::DeviceScan::InclusiveSum (rasterizer_impl.cu:278-284) + D2H read (:287-292)
void scan_offsets(dmro_state* st) {
    st->face_offsets.resize(st->tiles_touched.size());
    uint32_t acc = 0;
    for (size_t i = 0; i < st->tiles_touched.size(); i++) { acc += st->tiles_touched[i]; st->face_offsets[i] = acc; }
    st->R = st->tiles_touched.empty() ? 0 : (int64_t)st->face_offsets.back();
}

// duplicateWithKeys (rasterizer_impl.cu:44-97; tet renderer_impl.cu:44-99 with min depth),
// cub::DeviceRadixSort::SortPairs on bits [0, 32+bit) (:316-324), identifyTileRanges (:102-124).
// Guarded quirk (Q25): the reference emits a culled face's rect too (no tiles_touched
// check), racing with the slots of the following faces; here a face emits exactly
// tiles_touched entries, i.e. none when it was culled.
void bin_and_sort(const dmro_scene* s, dmro_state* st) {
    const int B = s->B, P = s->P, F = s->F;
    const size_t BF = (size_t)B * F;
    const int64_t R = st->R;
    std::vector<uint64_t> keys_unsorted((size_t)R);
    std::vector<uint32_t> vals_unsorted((size_t)R);
    const std::vector<float>& key_depth = st->tet ? st->min_depths : st->depths;
    const int grid_size = st->gx * st->gy;
#pragma omp parallel for schedule(static)
    for (int64_t idx = 0; idx < (int64_t)BF; idx++) {
        if (st->tiles_touched[idx] == 0) continue;
        int b = (int)(idx / F), f = (int)(idx % F);
        uint32_t off = (idx == 0) ? 0 : st->face_offsets[idx - 1];
        f2 img[3];
        for (int i = 0; i < 3; i++) {
            size_t bv = (size_t)b * P + s->faces[3 * f + i];
            img[i] = {st->image[2 * bv], st->image[2 * bv + 1]};
        }
        Rect r = band_clip(getRectFromTri(img[0], img[1], img[2], st->gx, st->gy), st->row_begin, st->row_end);
        uint32_t dbits;
        memcpy(&dbits, &key_depth[idx], 4);
        for (uint32_t y = r.miny; y < r.maxy; y++)
            for (uint32_t x = r.minx; x < r.maxx; x++) {
                uint64_t key = (uint64_t)y * st->gx + x;
                key = key + (uint64_t)((int64_t)grid_size * b);
                key <<= 32;
                key |= dbits;
                keys_unsorted[off] = key;
                vals_unsorted[off] = (uint32_t)f;
                off++;
            }
    }
    const uint32_t bit = getHigherMsb((uint32_t)(B * st->gx * st->gy));
    const int nbits = 32 + (int)bit;
    const uint64_t mask = nbits >= 64 ? ~0ull : ((1ull << nbits) - 1ull);
    std::vector<uint32_t> order((size_t)R);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b2) {
        return (keys_unsorted[a] & mask) < (keys_unsorted[b2] & mask);
    });
    st->keys.resize((size_t)R);
    st->values.resize((size_t)R);
    for (int64_t i = 0; i < R; i++) { st->keys[i] = keys_unsorted[order[i]]; st->values[i] = vals_unsorted[order[i]]; }

    st->ranges.assign((size_t)B * grid_size * 2, 0u);  // cudaMemset (:330), sized per tile (Q22)
    for (int64_t idx = 0; idx < R; idx++) {
        uint32_t currtile = (uint32_t)(st->keys[idx] >> 32);
        if (idx == 0) st->ranges[2 * currtile] = 0;
        else {
            uint32_t prevtile = (uint32_t)(st->keys[idx - 1] >> 32);
            if (currtile != prevtile) { st->ranges[2 * prevtile + 1] = (uint32_t)idx; st->ranges[2 * currtile] = (uint32_t)idx; }
        }
        if (idx == R - 1) st->ranges[2 * currtile + 1] = (uint32_t)R;
    }
}

// Seeded jitter, cuda_renderer/forward.cu:82-88,120-123: pixf = pixel - 0.5 + 0.5 * curand_uniform(state[idx]),
// two draws per pixel.  PARITY UNPINNED: cuRAND's XORWOW sequence (curand_init(seed, idx, 0)) cannot be restated
// here (its skip-ahead tables are not in the reference tree); the draws come from Philox-4x32-10 with key = seed
// and counter = pixel index, mapped to (0, 1] exactly as curand_uniform maps 32 bits.
static void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t k0, uint32_t& o0, uint32_t& o1) {
    uint32_t c2 = 0u, c3 = 0u, k1 = 0u;
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o0 = c0; o1 = c1;
}
static float uniform_01(uint32_t x) { return (float)x * 2.3283064365386963e-10f + 1.1641532182693481e-10f; }

// generateRaysCUDA: tri forward.cu:184-231; tet cuda_renderer/forward.cu:90-145
void generate_rays(const dmro_scene* s, dmro_state* st) {
    const int B = s->B, W = s->W, H = s->H;
    const size_t N = (size_t)B * W * H;
    st->ray_o.assign(N * 3, 0.f);
    st->ray_d.assign(N * 3, 0.f);
#pragma omp parallel for schedule(static)
    for (int64_t idx = 0; idx < (int64_t)N; idx++) {
        int b = (int)(idx / ((int64_t)W * H));
        int pixel_id = (int)(idx % ((int64_t)W * H));
        const float* imv = s->inv_mv_mats + 16 * b;
        const float* ipr = s->inv_proj_mats + 16 * b;
        f3 o = {imv[12], imv[13], imv[14]};
        int pixel_x = pixel_id % W, pixel_y = pixel_id / W;
        f2 pixf = {pixel_x + 0.5f, pixel_y + 0.5f};
        if (st->tet && s->ray_random_seed > 0) {
            uint32_t r0, r1;
            philox4x32_10((uint32_t)idx, (uint32_t)((uint64_t)idx >> 32), (uint32_t)s->ray_random_seed, r0, r1);
            pixf.x = (float)pixel_x - 0.5f + (0.5f * uniform_01(r0));
            pixf.y = (float)pixel_y - 0.5f + (0.5f * uniform_01(r1));
        }
        f2 pix_ndc = {pix2Ndc(pixf.x, W), pix2Ndc(pixf.y, H)};
        f4 pix_view = transformPoint4x4({pix_ndc.x, pix_ndc.y, -1.0f}, ipr);
        f4 pix_world = transformPoint4x4({pix_view.x, pix_view.y, pix_view.z}, imv);
        f3 d = f3{pix_world.x, pix_world.y, pix_world.z} - o;
        float len;
        if (st->tet) { len = sqrtf(dot(d, d)); len = fmaxf(len, 0.0001f); }
        else len = sqrtf(dot(d, d)) + 0.0000001f;
        d = d / len;
        st->ray_o[3 * idx] = o.x; st->ray_o[3 * idx + 1] = o.y; st->ray_o[3 * idx + 2] = o.z;
        st->ray_d[3 * idx] = d.x; st->ray_d[3 * idx + 1] = d.y; st->ray_d[3 * idx + 2] = d.z;
    }
}

void common_front(const dmro_scene* s, dmro_state* st, bool tet) {
    st->B = s->B; st->P = s->P; st->F = s->F; st->T = s->T; st->W = s->W; st->H = s->H;
    st->gx = (s->W + BLOCK_X - 1) / BLOCK_X;
    st->gy = (s->H + BLOCK_Y - 1) / BLOCK_Y;
    band_of(s, st->gy, st->row_begin, st->row_end);
    st->tet = tet;
    const size_t N = (size_t)s->B * s->W * s->H;
    st->ranges.assign((size_t)s->B * st->gx * st->gy * 2, 0u);
    st->final_T.assign(N, 0.f); st->final_prev_T.assign(N, 0.f); st->n_contrib.assign(N, 0u);
    if (s->P == 0 || s->F == 0) { st->R = 0; return; }  // render.cu:105 (+Q16 guard for F == 0)
    preprocess_point(s, st);
    preprocess_face(s, st);
    scan_offsets(st);
    bin_and_sort(s, st);
    generate_rays(s, st);
}

struct FaceRec {  // what renderCUDA stages in shared memory (forward.cu:320-339)
    int id, v0, v1, v2;
    f3 p0, p1, p2;
    f2 i0, i1, i2;
    float c0[3], c1[3], c2[3];
    float d0, d1, d2, opacity, intense;
};

inline FaceRec fetch_face(const dmro_scene* s, const dmro_state* st, int b, int face_id) {
    FaceRec r;
    r.id = face_id;
    r.v0 = s->faces[3 * face_id]; r.v1 = s->faces[3 * face_id + 1]; r.v2 = s->faces[3 * face_id + 2];
    r.p0 = get_vert(s->verts, r.v0); r.p1 = get_vert(s->verts, r.v1); r.p2 = get_vert(s->verts, r.v2);
    size_t b0 = (size_t)b * s->P + r.v0, b1 = (size_t)b * s->P + r.v1, b2 = (size_t)b * s->P + r.v2;
    r.i0 = {st->image[2 * b0], st->image[2 * b0 + 1]};
    r.i1 = {st->image[2 * b1], st->image[2 * b1 + 1]};
    r.i2 = {st->image[2 * b2], st->image[2 * b2 + 1]};
    for (int ch = 0; ch < 3; ch++) {
        r.c0[ch] = s->verts_color[3 * r.v0 + ch];
        r.c1[ch] = s->verts_color[3 * r.v1 + ch];
        r.c2[ch] = s->verts_color[3 * r.v2 + ch];
    }
    r.d0 = s->verts_depth[b0]; r.d1 = s->verts_depth[b1]; r.d2 = s->verts_depth[b2];
    r.opacity = s->faces_opacity[face_id];
    r.intense = s->faces_intense[(size_t)b * s->F + face_id];
    return r;
}

// TRI_FORWARD::renderCUDA (forward.cu:257-489), one tile
void tri_render_tile(const dmro_scene* s, dmro_state* st, int b, int ty, int tx, float* out_color, float* out_depth) {
    const int W = s->W, H = s->H;
    const int tile = (b * st->gx * st->gy) + ty * st->gx + tx;
    const uint32_t r0 = st->ranges[2 * tile], r1 = st->ranges[2 * tile + 1];
    std::vector<FaceRec> recs;
    recs.reserve(r1 > r0 ? r1 - r0 : 0);
    for (uint32_t k = r0; k < r1; k++) recs.push_back(fetch_face(s, st, b, (int)st->values[k]));
    for (int ly = 0; ly < BLOCK_Y; ly++)
        for (int lx = 0; lx < BLOCK_X; lx++) {
            const int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
            if (px >= W || py >= H) continue;
            const size_t pix_id = (size_t)W * py + px;
            const size_t bpix = (size_t)b * W * H + pix_id;
            const f3 ro = {st->ray_o[3 * bpix], st->ray_o[3 * bpix + 1], st->ray_o[3 * bpix + 2]};
            const f3 rd = {st->ray_d[3 * bpix], st->ray_d[3 * bpix + 1], st->ray_d[3 * bpix + 2]};
            float pT = 1.0f, T = 1.0f;
            uint32_t contributor = 0, last_contributor = 0;
            float C[3] = {0, 0, 0}, D = 0;
            const f2 pixf = {px + 0.5f, py + 0.5f};
            for (size_t j = 0; j < recs.size(); j++) {
                const FaceRec& r = recs[j];
                contributor++;
                if (!in_tri(pixf, r.i0, r.i1, r.i2)) continue;
                f3 tuv = {0, 0, 0};
                if (!ray_tri_intersection<false>(ro, rd, r.p0, r.p1, r.p2, tuv)) continue;
                float iuc, ivc; int code;
                clamp_bary_uv(tuv.y, tuv.z, iuc, ivc, code);
                float i0 = 1 - iuc - ivc, i1 = iuc, i2 = ivc;
                float iC[3];
                for (int ch = 0; ch < 3; ch++) {
                    iC[ch] = i0 * r.c0[ch] + i1 * r.c1[ch] + i2 * r.c2[ch];
                    iC[ch] = iC[ch] * r.intense;
                }
                float iD = i0 * r.d0 + i1 * r.d1 + i2 * r.d2;
                float alpha = r.opacity;
                float test_T = T * (1 - alpha);
                for (int ch = 0; ch < 3; ch++) C[ch] += iC[ch] * alpha * T;
                D += iD * alpha * T;
                pT = T; T = test_T;
                last_contributor = contributor;
                if (T < T_EPS) break;  // Q9: blend first, then test
            }
            st->final_prev_T[bpix] = pT;
            st->final_T[bpix] = T;
            st->n_contrib[bpix] = last_contributor;
            for (int ch = 0; ch < 3; ch++)
                out_color[((size_t)b * 3 + ch) * H * W + pix_id] = C[ch] + T * s->background[ch];
            out_depth[(size_t)b * H * W + pix_id] = D + T * 1.0f;
        }
}

struct TriGradAcc {  // per-thread double accumulators (the reference uses float atomics in arbitrary order)
    std::vector<double> dverts, dvcolor, dfopacity, dvdepth, dfintense;
    void init(const dmro_scene* s) {
        dverts.assign((size_t)s->P * 3, 0.0); dvcolor.assign((size_t)s->P * 3, 0.0);
        dfopacity.assign((size_t)s->F, 0.0);
        dvdepth.assign((size_t)s->B * s->P, 0.0); dfintense.assign((size_t)s->B * s->F, 0.0);
    }
};

// TRI_BACKWARD::renderCUDA (backward.cu:9-421), one tile
void tri_backward_tile(const dmro_scene* s, const dmro_state* st, int b, int ty, int tx,
                       const float* dL_dcolor, const float* dL_ddepth, TriGradAcc& g) {
    const int W = s->W, H = s->H, P = s->P, F = s->F;
    const int tile = (b * st->gx * st->gy) + ty * st->gx + tx;
    const uint32_t r0 = st->ranges[2 * tile], r1 = st->ranges[2 * tile + 1];
    if (r1 <= r0) return;
    std::vector<FaceRec> recs;  // in list order; walked from the back
    recs.reserve(r1 - r0);
    for (uint32_t k = r0; k < r1; k++) recs.push_back(fetch_face(s, st, b, (int)st->values[k]));
    const uint32_t toDo = r1 - r0;
    for (int ly = 0; ly < BLOCK_Y; ly++)
        for (int lx = 0; lx < BLOCK_X; lx++) {
            const int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
            if (px >= W || py >= H) continue;
            const size_t pix_id = (size_t)W * py + px;
            const size_t bpix = (size_t)b * W * H + pix_id;
            const f3 ro = {st->ray_o[3 * bpix], st->ray_o[3 * bpix + 1], st->ray_o[3 * bpix + 2]};
            const f3 rd = {st->ray_d[3 * bpix], st->ray_d[3 * bpix + 1], st->ray_d[3 * bpix + 2]};
            const float T_final = st->final_T[bpix];
            const float prev_T_final = st->final_prev_T[bpix];
            float T = prev_T_final;
            bool T_first_pass = true;
            uint32_t contributor = toDo;
            const uint32_t last_contributor = st->n_contrib[bpix];
            float accum_rec[3] = {0, 0, 0}, accum_recd = 0;
            float dpc[3];
            for (int i = 0; i < 3; i++) dpc[i] = dL_dcolor[((size_t)b * 3 + i) * H * W + pix_id];
            const float dpd = dL_ddepth[(size_t)b * H * W + pix_id];
            float last_alpha = 0, last_color[3] = {0, 0, 0}, last_depth = 0;
            const f2 pixf = {px + 0.5f, py + 0.5f};
            for (uint32_t jj = 0; jj < toDo; jj++) {
                const FaceRec& r = recs[toDo - 1 - jj];
                contributor--;
                if (contributor >= last_contributor) continue;
                if (!in_tri(pixf, r.i0, r.i1, r.i2)) continue;
                f3 tuv = {0, 0, 0};
                if (!ray_tri_intersection<false>(ro, rd, r.p0, r.p1, r.p2, tuv)) continue;
                float iuc, ivc; int code;
                clamp_bary_uv(tuv.y, tuv.z, iuc, ivc, code);
                float i0 = 1 - iuc - ivc, i1 = iuc, i2 = ivc;
                float iC[3];
                for (int ch = 0; ch < 3; ch++)
                    iC[ch] = (i0 * r.c0[ch] + i1 * r.c1[ch] + i2 * r.c2[ch]) * r.intense;
                float iD = i0 * r.d0 + i1 * r.d1 + i2 * r.d2;
                float alpha = r.opacity;
                if (!T_first_pass) T = T / (1.f - alpha);  // Q10
                T_first_pass = false;

                float dL_dicolor[3] = {0, 0, 0}, dL_didepth = 0, dL_dalpha = 0.0f;
                for (int ch = 0; ch < 3; ch++) {
                    const float c = iC[ch];
                    accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                    last_color[ch] = c;
                    dL_dicolor[ch] = dpc[ch] * alpha * T;
                    dL_dalpha += (c - accum_rec[ch]) * dpc[ch];
                }
                {
                    const float c = iD;
                    accum_recd = last_alpha * last_depth + (1.f - last_alpha) * accum_recd;
                    last_depth = c;
                    dL_didepth = dpd * alpha * T;
                    dL_dalpha += (c - accum_recd) * dpd;
                }
                dL_dalpha *= T;
                last_alpha = alpha;
                float bg_dot = 0, bd_dot = 0;
                for (int i = 0; i < 3; i++) bg_dot += s->background[i] * dpc[i];
                bd_dot += (float)(1.0 * (double)dpd);
                if (alpha == 1.0f) {
                    dL_dalpha += (-prev_T_final) * bg_dot;
                    dL_dalpha += (-prev_T_final) * bd_dot;
                } else {
                    dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;
                    dL_dalpha += (-T_final / (1.f - alpha)) * bd_dot;
                }

                float dL_di0 = 0, dL_di1 = 0, dL_di2 = 0;
                float dvc0[3] = {0, 0, 0}, dvc1[3] = {0, 0, 0}, dvc2[3] = {0, 0, 0};
                float dfint = 0;
                for (int ch = 0; ch < 3; ch++) {
                    dL_di0 += r.c0[ch] * dL_dicolor[ch] * r.intense;
                    dL_di1 += r.c1[ch] * dL_dicolor[ch] * r.intense;
                    dL_di2 += r.c2[ch] * dL_dicolor[ch] * r.intense;
                    dvc0[ch] += i0 * dL_dicolor[ch] * r.intense;
                    dvc1[ch] += i1 * dL_dicolor[ch] * r.intense;
                    dvc2[ch] += i2 * dL_dicolor[ch] * r.intense;
                    dfint += (i0 * r.c0[ch] + i1 * r.c1[ch] + i2 * r.c2[ch]) * dL_dicolor[ch];
                }
                dL_di0 += r.d0 * dL_didepth;
                dL_di1 += r.d1 * dL_didepth;
                dL_di2 += r.d2 * dL_didepth;
                float dvd0 = i0 * dL_didepth, dvd1 = i1 * dL_didepth, dvd2 = i2 * dL_didepth;

                const float di0_diuc = -1, di0_divc = -1, di1_diuc = 1, di1_divc = 0, di2_diuc = 0, di2_divc = 1;
                float diuc_diu, diuc_div, divc_diu, divc_div;
                clamp_bary_uv_grad(code, diuc_diu, diuc_div, divc_diu, divc_div);
                float di0_diu = di0_diuc * diuc_diu + di0_divc * divc_diu;
                float di0_div = di0_diuc * diuc_div + di0_divc * divc_div;
                float di1_diu = di1_diuc * diuc_diu + di1_divc * divc_diu;
                float di1_div = di1_diuc * diuc_div + di1_divc * divc_div;
                float di2_diu = di2_diuc * diuc_diu + di2_divc * divc_diu;
                float di2_div = di2_diuc * diuc_div + di2_divc * divc_div;
                float dL_diu = dL_di0 * di0_diu + dL_di1 * di1_diu + dL_di2 * di2_diu;
                float dL_div = dL_di0 * di0_div + dL_di1 * di1_div + dL_di2 * di2_div;

                f3 du0, du1, du2, dv0, dv1, dv2;
                ray_tri_intersection_grad(ro, rd, r.p0, r.p1, r.p2, du0, du1, du2, dv0, dv1, dv2);
                f3 dp0 = dL_diu * du0 + dL_div * dv0;
                f3 dp1 = dL_diu * du1 + dL_div * dv1;
                f3 dp2 = dL_diu * du2 + dL_div * dv2;

                if (g_tri_grad_f64) {  // (noise measurement only, see tri_dverts_f64)
                    d3 q0, q1, q2;
                    tri_dverts_f64(ro, rd, r.p0, r.p1, r.p2, dL_diu, dL_div, q0, q1, q2);
                    g.dverts[3 * r.v0] += q0.x; g.dverts[3 * r.v0 + 1] += q0.y; g.dverts[3 * r.v0 + 2] += q0.z;
                    g.dverts[3 * r.v1] += q1.x; g.dverts[3 * r.v1 + 1] += q1.y; g.dverts[3 * r.v1 + 2] += q1.z;
                    g.dverts[3 * r.v2] += q2.x; g.dverts[3 * r.v2 + 1] += q2.y; g.dverts[3 * r.v2 + 2] += q2.z;
                } else {
                g.dverts[3 * r.v0] += dp0.x; g.dverts[3 * r.v0 + 1] += dp0.y; g.dverts[3 * r.v0 + 2] += dp0.z;
                g.dverts[3 * r.v1] += dp1.x; g.dverts[3 * r.v1 + 1] += dp1.y; g.dverts[3 * r.v1 + 2] += dp1.z;
                g.dverts[3 * r.v2] += dp2.x; g.dverts[3 * r.v2 + 1] += dp2.y; g.dverts[3 * r.v2 + 2] += dp2.z;
                }
                for (int k = 0; k < 3; k++) {
                    g.dvcolor[3 * r.v0 + k] += dvc0[k];
                    g.dvcolor[3 * r.v1 + k] += dvc1[k];
                    g.dvcolor[3 * r.v2 + k] += dvc2[k];
                }
                g.dvdepth[(size_t)b * P + r.v0] += dvd0;
                g.dvdepth[(size_t)b * P + r.v1] += dvd1;
                g.dvdepth[(size_t)b * P + r.v2] += dvd2;
                g.dfopacity[r.id] += dL_dalpha;
                g.dfintense[(size_t)b * F + r.id] += dfint;
            }
        }
}

// firstIntersectCUDA (cuda_renderer/forward.cu:298-445), one tile.  Q18 guarded:
// only pixels inside the image are touched.
void tet_first_intersect_tile(const dmro_scene* s, dmro_state* st, int b, int ty, int tx) {
    const int W = s->W, H = s->H, F = s->F;
    const int tile = (b * st->gx * st->gy) + ty * st->gx + tx;
    const uint32_t r0 = st->ranges[2 * tile], r1 = st->ranges[2 * tile + 1];
    for (int ly = 0; ly < BLOCK_Y; ly++)
        for (int lx = 0; lx < BLOCK_X; lx++) {
            const int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
            if (px >= W || py >= H) continue;
            const size_t bpix = (size_t)b * W * H + (size_t)W * py + px;
            const f3 ro = {st->ray_o[3 * bpix], st->ray_o[3 * bpix + 1], st->ray_o[3 * bpix + 2]};
            const f3 rd = {st->ray_d[3 * bpix], st->ray_d[3 * bpix + 1], st->ray_d[3 * bpix + 2]};
            int ff = -1, ft = -1;
            float min_T = -1.0f, min_T_max_depth = -1.0f;
            for (uint32_t k = r0; k < r1; k++) {
                const int face_id = (int)st->values[k];
                const size_t fb = (size_t)b * F + face_id;
                if (min_T >= 0.0f && st->min_depths[fb] > min_T_max_depth) break;
                f3 p0, p1, p2, tuv;
                get_face_vert(s->verts, s->faces, face_id, p0, p1, p2);
                if (!ray_tri_intersection<true>(ro, rd, p0, p1, p2, tuv)) continue;
                float curr_T = tuv.x;
                if (min_T < 0.0f || curr_T < min_T) {
                    min_T = curr_T;
                    min_T_max_depth = st->max_depths[fb];
                    ff = face_id;
                }
            }
            if (ff >= 0) {
                for (int i = 0; i < 2; i++) {
                    int tet_id = s->face_tets[2 * ff + i];
                    if (tet_id < 0) continue;
                    f3 n = tet_face_outward_normal(s->verts, s->faces, s->tets, ff, tet_id);
                    if (dot(n, rd) < 0.0f) ft = tet_id;
                }
            }
            st->first_face[bpix] = ff;
            st->first_tet[bpix] = ft;
        }
}

// TET_FORWARD::renderCUDA (cuda_renderer/forward.cu:485-815), one pixel
void tet_render_pixel(const dmro_scene* s, dmro_state* st, int b, int px, int py,
                      float* out_color, float* out_depth, float* out_active) {
    const int W = s->W, H = s->H, F = s->F;
    const size_t pix_id = (size_t)W * py + px;
    const size_t bpix = (size_t)b * W * H + pix_id;
    const f3 ro = {st->ray_o[3 * bpix], st->ray_o[3 * bpix + 1], st->ray_o[3 * bpix + 2]};
    const f3 rd = {st->ray_d[3 * bpix], st->ray_d[3 * bpix + 1], st->ray_d[3 * bpix + 2]};
    const int first_face = st->first_face[bpix], first_tet = st->first_tet[bpix];
    const float* mv = s->mv_mats + 16 * b;
    const float* pr = s->proj_mats + 16 * b;
    bool done = false;
    float rt = 0.0f, iu = 0.f, iv = 0.f;
    if (first_face == -1 || first_tet == -1) done = true;
    else {
        f3 tuv = {0, 0, 0}, p0, p1, p2;
        get_face_vert(s->verts, s->faces, first_face, p0, p1, p2);
        ray_tri_intersection<true>(ro, rd, p0, p1, p2, tuv);
        rt = tuv.x; iu = tuv.y; iv = tuv.z;
    }
    f3 C = {0, 0, 0};
    float D = 0.0f, log_T = 0.0f, prev_log_T = 0.0f;
    int last_face = -1, last_tet = -1;
    bool is_active = false;
    uint32_t n_contrib = 0;
    int curr_face = first_face, curr_tet = first_tet;
    float curr_rt = rt, curr_iu = iu, curr_iv = iv;
    while (!done) {
        f3 c0 = get_vert(s->verts_color, s->faces[3 * curr_face]);
        f3 c1 = get_vert(s->verts_color, s->faces[3 * curr_face + 1]);
        f3 c2 = get_vert(s->verts_color, s->faces[3 * curr_face + 2]);
        f3 col = (c0 + (c1 - c0) * curr_iu + (c2 - c0) * curr_iv);  // Q21
        float opacity = s->faces_opacity[curr_face];
        float intense = s->faces_intense[(size_t)b * F + curr_face];
        col = col * intense;
        float tmp_T = expf(log_T);
        C = C + tmp_T * opacity * col;
        f3 pt = ro + (rd * curr_rt);
        f4 pn = transformPoint4x4(transformPoint4x3(pt, mv), pr);
        float pw = 1.0f / clamp_w(pn.w);  // fp32 divide here (Q1)
        float pdepth = pn.z * pw;
        D += tmp_T * opacity * pdepth;
        prev_log_T = log_T;
        if (opacity < 1.0f) log_T += logf(1.0f - opacity);
        else log_T = logf(T_EPS * 0.1f);
        if (expf(log_T) < T_EPS) { done = true; is_active = true; }
        n_contrib++;
        last_face = curr_face;
        last_tet = curr_tet;

        int next_face = -1, next_tet = -1;
        float next_rt = 0, next_iu = 0, next_iv = 0;
        if (curr_tet == -1) { is_active = true; done = true; }
        if (!done) {
            int others[4];
            int cnt = 0;
            for (int i = 0; i < 4; i++) {
                int tf = s->tet_faces[4 * curr_tet + i];
                if (tf == curr_face) continue;
                others[cnt++] = tf;
            }
            if (cnt != 3) done = true;  // "Error case 1"; nothing below is observable once done
            f3 ncur = tet_face_outward_normal(s->verts, s->faces, s->tets, curr_face, curr_tet);
            if (dot(ncur, rd) >= 0.0f) done = true;  // "Error case 2"
            int next_cnt = 0;
            for (int i = 0; i < std::min(cnt, 3); i++) {
                f3 p0, p1, p2, tuv;
                get_face_vert(s->verts, s->faces, others[i], p0, p1, p2);
                bool hit = ray_tri_intersection<true>(ro, rd, p0, p1, p2, tuv);
                f3 n = tet_face_outward_normal(s->verts, s->faces, s->tets, others[i], curr_tet);
                if (hit && dot(n, rd) > 0.0f) {
                    next_face = others[i];
                    next_rt = tuv.x; next_iu = tuv.y; next_iv = tuv.z;
                    next_cnt++;
                }
            }
            if (next_cnt != 1) done = true;  // "Error case 3"
            else {
                for (int i = 0; i < 2; i++) {
                    int t = s->face_tets[2 * next_face + i];
                    if (t == curr_tet || t == -1) continue;
                    next_tet = t;
                    break;
                }
            }
            curr_face = next_face; curr_tet = next_tet;
            curr_rt = next_rt; curr_iu = next_iu; curr_iv = next_iv;
        }
    }
    st->final_T[bpix] = log_T;           // final_log_T
    st->final_prev_T[bpix] = prev_log_T;  // final_prev_log_T
    st->last_face[bpix] = last_face;
    st->last_tet[bpix] = last_tet;
    st->n_contrib[bpix] = n_contrib;
    st->is_active[bpix] = is_active ? 1 : 0;
    const size_t HW = (size_t)H * W;
    if (is_active) {
        float fT = expf(log_T);
        out_color[((size_t)b * 3 + 0) * HW + pix_id] = C.x + fT * s->background[0];
        out_color[((size_t)b * 3 + 1) * HW + pix_id] = C.y + fT * s->background[1];
        out_color[((size_t)b * 3 + 2) * HW + pix_id] = C.z + fT * s->background[2];
        out_depth[(size_t)b * HW + pix_id] = D + fT * 1.0f;
        out_active[(size_t)b * HW + pix_id] = 1.0f;
    } else {
        out_color[((size_t)b * 3 + 0) * HW + pix_id] = s->background[0];
        out_color[((size_t)b * 3 + 1) * HW + pix_id] = s->background[1];
        out_color[((size_t)b * 3 + 2) * HW + pix_id] = s->background[2];
        out_depth[(size_t)b * HW + pix_id] = 1.0f;
        out_active[(size_t)b * HW + pix_id] = 0.0f;
    }
}

// TET_BACKWARD::renderCUDA (cuda_renderer/backward.cu:86-487), one pixel
void tet_backward_pixel(const dmro_scene* s, const dmro_state* st, int b, int px, int py,
                        const float* dL_dcolor, const float* dL_ddepth,
                        std::vector<double>& dvcolor, std::vector<double>& dfopacity) {
    const int W = s->W, H = s->H, F = s->F;
    const size_t pix_id = (size_t)W * py + px;
    const size_t HW = (size_t)H * W;
    const size_t bpix = (size_t)b * HW + pix_id;
    const float fprev_log_T = st->final_prev_T[bpix], flog_T = st->final_T[bpix];
    const float final_prev_T = expf(fprev_log_T), final_T = expf(flog_T);
    float prev_log_T = fprev_log_T;
    const int last_face = st->last_face[bpix], last_tet = st->last_tet[bpix];
    bool done = false;
    if (!st->is_active[bpix]) done = true;
    const int first_face = st->first_face[bpix];
    float dpc[3];
    for (int i = 0; i < 3; i++) dpc[i] = dL_dcolor[((size_t)b * 3 + i) * HW + pix_id];
    const float dpd = dL_ddepth[(size_t)b * HW + pix_id];
    const float* mv = s->mv_mats + 16 * b;
    const float* pr = s->proj_mats + 16 * b;
    const f3 ro = {st->ray_o[3 * bpix], st->ray_o[3 * bpix + 1], st->ray_o[3 * bpix + 2]};
    const f3 rd = {st->ray_d[3 * bpix], st->ray_d[3 * bpix + 1], st->ray_d[3 * bpix + 2]};
    float rt = 0.0f, iu = 0, iv = 0;
    if (last_face == -1) done = true;
    else {
        f3 tuv = {0, 0, 0}, p0, p1, p2;
        get_face_vert(s->verts, s->faces, last_face, p0, p1, p2);
        ray_tri_intersection<true>(ro, rd, p0, p1, p2, tuv);
        rt = tuv.x; iu = tuv.y; iv = tuv.z;
    }
    int curr_face = last_face, curr_tet = last_tet;
    float curr_rt = rt, curr_iu = iu, curr_iv = iv;
    float last_alpha = 0.0f, last_color[3] = {0, 0, 0}, accum_rec[3] = {0, 0, 0};
    float last_depth = 0.0f, accum_recd = 0.0f;
    if (curr_face != -1) {
        for (int i = 0; i < 2; i++) {
            int t = s->face_tets[2 * curr_face + i];
            if (t == curr_tet) continue;
            curr_tet = t;
            break;
        }
    }
    bool first_iter = true;
    while (!done) {
        const int v0 = s->faces[3 * curr_face], v1 = s->faces[3 * curr_face + 1], v2 = s->faces[3 * curr_face + 2];
        f3 c0 = get_vert(s->verts_color, v0), c1 = get_vert(s->verts_color, v1), c2 = get_vert(s->verts_color, v2);
        float i0 = 1.0f - curr_iu - curr_iv, i1 = curr_iu, i2 = curr_iv;
        f3 col = (i0 * c0) + (i1 * c1) + (i2 * c2);  // Q21
        float opacity = s->faces_opacity[curr_face];
        float intense = s->faces_intense[(size_t)b * F + curr_face];
        col = col * intense;
        f3 pt = ro + (rd * curr_rt);
        f4 pn = transformPoint4x4(transformPoint4x3(pt, mv), pr);
        float pw = 1.0f / clamp_w(pn.w);
        float pdepth = pn.z * pw;
        if (!first_iter) prev_log_T = prev_log_T - logf(1.0f - opacity);
        first_iter = false;
        float prev_T = expf(prev_log_T);

        float dL_dcol[3], dL_dop = 0.0f;
        float tc[3] = {col.x, col.y, col.z};
        for (int ch = 0; ch < 3; ch++) {
            const float c = tc[ch];
            accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
            last_color[ch] = c;
            dL_dcol[ch] = dpc[ch] * opacity * prev_T;
            dL_dop += (c - accum_rec[ch]) * dpc[ch];
        }
        {
            const float c = pdepth;
            accum_recd = last_alpha * last_depth + (1.f - last_alpha) * accum_recd;
            last_depth = c;
            dL_dop += (c - accum_recd) * dpd;
        }
        dL_dop *= prev_T;
        last_alpha = opacity;
        float bg_dot = 0, bd_dot = 0;
        for (int i = 0; i < 3; i++) bg_dot += s->background[i] * dpc[i];
        bd_dot += (float)(1.0 * (double)dpd);
        if (opacity == 1.0f) {
            dL_dop += (-final_prev_T) * bg_dot;
            dL_dop += (-final_prev_T) * bd_dot;
        } else {
            dL_dop += (-final_T / (1.f - opacity)) * bg_dot;
            dL_dop += (-final_T / (1.f - opacity)) * bd_dot;
        }
        for (int ch = 0; ch < 3; ch++) {
            dvcolor[3 * v0 + ch] += i0 * dL_dcol[ch] * intense;
            dvcolor[3 * v1 + ch] += i1 * dL_dcol[ch] * intense;
            dvcolor[3 * v2 + ch] += i2 * dL_dcol[ch] * intense;
        }
        dfopacity[curr_face] += dL_dop;

        if (curr_face == first_face) done = true;
        if (!done) {
            if (curr_tet == -1) done = true;
            else {
                int prev_face = -1, prev_tet = -1;
                float prev_rt = 0, prev_iu = 0, prev_iv = 0;
                int others[4];
                int cnt = 0;
                for (int i = 0; i < 4; i++) {
                    int tf = s->tet_faces[4 * curr_tet + i];
                    if (tf == curr_face) continue;
                    others[cnt++] = tf;
                }
                if (cnt != 3) done = true;
                f3 ncur = tet_face_outward_normal(s->verts, s->faces, s->tets, curr_face, curr_tet);
                if (dot(ncur, rd) <= 0.0f) done = true;
                int prev_cnt = 0;
                for (int i = 0; i < std::min(cnt, 3); i++) {
                    f3 p0, p1, p2, tuv;
                    get_face_vert(s->verts, s->faces, others[i], p0, p1, p2);
                    bool hit = ray_tri_intersection<true>(ro, rd, p0, p1, p2, tuv);
                    f3 n = tet_face_outward_normal(s->verts, s->faces, s->tets, others[i], curr_tet);
                    if (hit && dot(n, rd) < 0.0f) {
                        prev_face = others[i];
                        prev_rt = tuv.x; prev_iu = tuv.y; prev_iv = tuv.z;
                        prev_cnt++;
                    }
                }
                if (prev_cnt != 1) done = true;
                else {
                    for (int i = 0; i < 2; i++) {
                        int t = s->face_tets[2 * prev_face + i];
                        if (t == curr_tet || t == -1) continue;
                        prev_tet = t;
                        break;
                    }
                }
                curr_face = prev_face; curr_tet = prev_tet;
                curr_rt = prev_rt; curr_iu = prev_iu; curr_iv = prev_iv;
            }
        }
    }
}

template <class T>
int64_t copy_out(const std::vector<T>& v, void* dst, int64_t cap) {
    int64_t n = (int64_t)(v.size() * sizeof(T));
    if (dst && n > 0) memcpy(dst, v.data(), (size_t)std::min(n, cap));
    return n;
}

}  // namespace

extern "C" {

const char* dmro_last_error(void) { return g_err.c_str(); }
void dmro_set_tri_grad_f64(int on) { g_tri_grad_f64 = on; }

int dmro_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

// Rasterizer::forward (rasterizer_impl.cu:175-383) behind RasterizeTrianglesCUDA (render.cu:29-132)
dmro_state* dmro_tri_forward(const dmro_scene* s, float* out_color, float* out_depth) {
    if (!check_scene(s, false)) return nullptr;
    dmro_state* st = new dmro_state();
    const size_t HW = (size_t)s->H * s->W;
    std::fill(out_color, out_color + (size_t)s->B * 3 * HW, 0.f);  // torch::full(0) render.cu:88-89
    std::fill(out_depth, out_depth + (size_t)s->B * HW, 0.f);
    common_front(s, st, false);
    if (s->P == 0 || s->F == 0) return st;
    const int ntiles = s->B * (st->row_end - st->row_begin) * st->gx;
#pragma omp parallel for schedule(dynamic, 1)
    for (int t = 0; t < ntiles; t++) {
        int rows = st->row_end - st->row_begin;
        int b = t / (rows * st->gx);
        int rem = t % (rows * st->gx);
        tri_render_tile(s, st, b, st->row_begin + rem / st->gx, rem % st->gx, out_color, out_depth);
    }
    return st;
}

// Rasterizer::backward (rasterizer_impl.cu:387-467) behind RasterizeTrianglesBackwardCUDA (render.cu:134-208)
int dmro_tri_backward(const dmro_scene* s, const dmro_state* st,
                      const float* dL_dcolor, const float* dL_ddepth,
                      float* dL_dverts, float* dL_dvcolor, float* dL_dfopacity,
                      float* dL_dvdepth, float* dL_dfintense) {
    if (!check_scene(s, false) || !st) return 1;
    const size_t P = (size_t)s->P, F = (size_t)s->F, B = (size_t)s->B;
    std::fill(dL_dverts, dL_dverts + 3 * P, 0.f);
    std::fill(dL_dvcolor, dL_dvcolor + 3 * P, 0.f);
    std::fill(dL_dfopacity, dL_dfopacity + F, 0.f);
    std::fill(dL_dvdepth, dL_dvdepth + B * P, 0.f);
    std::fill(dL_dfintense, dL_dfintense + B * F, 0.f);
    if (s->F == 0 || s->P == 0) return 0;
    const int nthreads = dmro_num_threads();
    std::vector<TriGradAcc> acc((size_t)nthreads);
    for (auto& a : acc) a.init(s);
    const int rows = st->row_end - st->row_begin;
    const int ntiles = s->B * rows * st->gx;
#pragma omp parallel for schedule(dynamic, 1)
    for (int t = 0; t < ntiles; t++) {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        int b = t / (rows * st->gx);
        int rem = t % (rows * st->gx);
        tri_backward_tile(s, st, b, st->row_begin + rem / st->gx, rem % st->gx, dL_dcolor, dL_ddepth, acc[tid]);
    }
    auto reduce = [&](float* dst, size_t n, std::vector<double> TriGradAcc::*m) {
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < (int64_t)n; i++) {
            double v = 0;
            for (auto& a : acc) v += (a.*m)[i];
            dst[i] = (float)v;
        }
    };
    reduce(dL_dverts, 3 * P, &TriGradAcc::dverts);
    reduce(dL_dvcolor, 3 * P, &TriGradAcc::dvcolor);
    reduce(dL_dfopacity, F, &TriGradAcc::dfopacity);
    reduce(dL_dvdepth, B * P, &TriGradAcc::dvdepth);
    reduce(dL_dfintense, B * F, &TriGradAcc::dfintense);
    return 0;
}

// Renderer::forward (cuda_renderer/renderer_impl.cu:193-410) behind RenderFTetsCUDA (render.cu:213-336)
dmro_state* dmro_tet_forward(const dmro_scene* s, float* out_color, float* out_depth, float* out_active) {
    if (!check_scene(s, true)) return nullptr;
    if (s->T < 0) { g_err = "bad T"; return nullptr; }
    dmro_state* st = new dmro_state();
    const size_t HW = (size_t)s->H * s->W, N = (size_t)s->B * HW;
    std::fill(out_color, out_color + (size_t)s->B * 3 * HW, 0.f);
    std::fill(out_depth, out_depth + N, 0.f);
    std::fill(out_active, out_active + N, 0.f);
    common_front(s, st, true);
    st->first_face.assign(N, -1); st->first_tet.assign(N, -1);
    st->last_face.assign(N, -1); st->last_tet.assign(N, -1);
    st->is_active.assign(N, 0);
    if (s->P == 0 || s->F == 0) {  // Q17 guard: nothing to march, every pixel inactive
        for (int b = 0; b < s->B; b++)
            for (size_t p = 0; p < HW; p++) {
                if ((int)(p / s->W) / BLOCK_Y < st->row_begin || (int)(p / s->W) / BLOCK_Y >= st->row_end) continue;
                for (int ch = 0; ch < 3; ch++) out_color[((size_t)b * 3 + ch) * HW + p] = s->background[ch];
                out_depth[(size_t)b * HW + p] = 1.0f;
            }
        return st;
    }
    const int rows = st->row_end - st->row_begin;
    const int ntiles = s->B * rows * st->gx;
#pragma omp parallel for schedule(dynamic, 1)
    for (int t = 0; t < ntiles; t++) {
        int b = t / (rows * st->gx);
        int rem = t % (rows * st->gx);
        int ty = st->row_begin + rem / st->gx, tx = rem % st->gx;
        tet_first_intersect_tile(s, st, b, ty, tx);
        for (int ly = 0; ly < BLOCK_Y; ly++)
            for (int lx = 0; lx < BLOCK_X; lx++) {
                int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
                if (px >= s->W || py >= s->H) continue;
                tet_render_pixel(s, st, b, px, py, out_color, out_depth, out_active);
            }
    }
    return st;
}

// Renderer::backward (renderer_impl.cu:413-498) behind RenderFTetsBackwardCUDA (render.cu:338-412)
int dmro_tet_backward(const dmro_scene* s, const dmro_state* st,
                      const float* dL_dcolor, const float* dL_ddepth,
                      float* dL_dvcolor, float* dL_dfopacity) {
    if (!check_scene(s, true) || !st) return 1;
    const size_t P = (size_t)s->P, F = (size_t)s->F;
    std::fill(dL_dvcolor, dL_dvcolor + 3 * P, 0.f);
    std::fill(dL_dfopacity, dL_dfopacity + F, 0.f);
    if (s->F == 0 || s->P == 0) return 0;
    const int nthreads = dmro_num_threads();
    std::vector<std::vector<double>> accc((size_t)nthreads, std::vector<double>(3 * P, 0.0));
    std::vector<std::vector<double>> acco((size_t)nthreads, std::vector<double>(F, 0.0));
    const int rows = st->row_end - st->row_begin;
    const int ntiles = s->B * rows * st->gx;
#pragma omp parallel for schedule(dynamic, 1)
    for (int t = 0; t < ntiles; t++) {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        int b = t / (rows * st->gx);
        int rem = t % (rows * st->gx);
        int ty = st->row_begin + rem / st->gx, tx = rem % st->gx;
        for (int ly = 0; ly < BLOCK_Y; ly++)
            for (int lx = 0; lx < BLOCK_X; lx++) {
                int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
                if (px >= s->W || py >= s->H) continue;
                tet_backward_pixel(s, st, b, px, py, dL_dcolor, dL_ddepth, accc[tid], acco[tid]);
            }
    }
    for (size_t i = 0; i < 3 * P; i++) { double v = 0; for (auto& a : accc) v += a[i]; dL_dvcolor[i] = (float)v; }
    for (size_t i = 0; i < F; i++) { double v = 0; for (auto& a : acco) v += a[i]; dL_dfopacity[i] = (float)v; }
    return 0;
}

int64_t dmro_num_rendered(const dmro_state* st) { return st ? st->R : -1; }

int64_t dmro_get(const dmro_state* st, const char* name, void* dst, int64_t cap) {
    if (!st || !name) return -1;
    std::string n(name);
    if (n == "ndc") return copy_out(st->ndc, dst, cap);
    if (n == "image") return copy_out(st->image, dst, cap);
    if (n == "depths") return copy_out(st->depths, dst, cap);
    if (n == "min_depths") return copy_out(st->min_depths, dst, cap);
    if (n == "max_depths") return copy_out(st->max_depths, dst, cap);
    if (n == "tiles_touched") return copy_out(st->tiles_touched, dst, cap);
    if (n == "face_offsets") return copy_out(st->face_offsets, dst, cap);
    if (n == "keys") return copy_out(st->keys, dst, cap);
    if (n == "values") return copy_out(st->values, dst, cap);
    if (n == "ranges") return copy_out(st->ranges, dst, cap);
    if (n == "ray_o") return copy_out(st->ray_o, dst, cap);
    if (n == "ray_d") return copy_out(st->ray_d, dst, cap);
    if (n == "final_T") return copy_out(st->final_T, dst, cap);
    if (n == "final_prev_T") return copy_out(st->final_prev_T, dst, cap);
    if (n == "n_contrib") return copy_out(st->n_contrib, dst, cap);
    if (n == "first_face") return copy_out(st->first_face, dst, cap);
    if (n == "first_tet") return copy_out(st->first_tet, dst, cap);
    if (n == "last_face") return copy_out(st->last_face, dst, cap);
    if (n == "last_tet") return copy_out(st->last_tet, dst, cap);
    if (n == "is_active") return copy_out(st->is_active, dst, cap);
    g_err = "unknown intermediate: " + n;
    return -1;
}

void dmro_free(dmro_state* st) { delete st; }

int dmro_in_tri(float px, float py, float x1, float y1, float x2, float y2, float x3, float y3) {
    return in_tri({px, py}, {x1, y1}, {x2, y2}, {x3, y3}) ? 1 : 0;
}
void dmro_clamp_bary_uv(float u, float v, float* uc, float* vc, int* code) { clamp_bary_uv(u, v, *uc, *vc, *code); }
int dmro_ray_tri(const float* o, const float* d, const float* p0, const float* p1, const float* p2,
                 int tet_flavour, float* tuv) {
    f3 r = {0, 0, 0};
    f3 O = {o[0], o[1], o[2]}, Dd = {d[0], d[1], d[2]};
    f3 A = {p0[0], p0[1], p0[2]}, Bq = {p1[0], p1[1], p1[2]}, Cq = {p2[0], p2[1], p2[2]};
    bool ok = tet_flavour ? ray_tri_intersection<true>(O, Dd, A, Bq, Cq, r) : ray_tri_intersection<false>(O, Dd, A, Bq, Cq, r);
    tuv[0] = r.x; tuv[1] = r.y; tuv[2] = r.z;
    return ok ? 1 : 0;
}
float dmro_ndc2pix(float v, int S) { return ndc2Pix(v, S); }
float dmro_pix2ndc(float v, int S) { return pix2Ndc(v, S); }
void dmro_rect_from_tri(const float* p0, const float* p1, const float* p2, int gx, int gy, uint32_t* rect) {
    Rect r = getRectFromTri({p0[0], p0[1]}, {p1[0], p1[1]}, {p2[0], p2[1]}, gx, gy);
    rect[0] = r.minx; rect[1] = r.miny; rect[2] = r.maxx; rect[3] = r.maxy;
}
uint32_t dmro_higher_msb(uint32_t n) { return getHigherMsb(n); }

}  // extern "C"
