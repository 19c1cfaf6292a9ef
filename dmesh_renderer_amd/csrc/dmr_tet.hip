// dmr_tet.hip -- tet renderer: first hit per pixel, then tet-to-tet ray march (gfx950).
//
// Replaces firstIntersectCUDA (cuda_renderer/forward.cu:298-445), TET_FORWARD::renderCUDA
// (cuda_renderer/forward.cu:485-815) and TET_BACKWARD::renderCUDA (cuda_renderer/backward.cu:86-487).
//
// first_intersect walks the tile's min-depth-sorted face list in LDS-staged chunks (one
// 16x16 tile per 256-thread workgroup, wave w = 8x8 quadrant w); the march kernels are one
// thread per pixel stepping through packed per-face / per-tet records (below) that every forward
// rebuilds.  The backward sums per face in a per-tile LDS hash table (DPP-quad pre-reduction,
// ds_add_f64) before touching HBM.  Rays are recomputed per pixel (generateRaysCUDA,
// forward.cu:90-145; seeded jitter from a counter-based generator, dmr_device.hpp).
// Guards: Q17 (no work when P/F/T == 0), Q18 (only pixels inside the image are touched),
// Q19 (the unread is_active_backward diagnostic is dropped).
#include <algorithm>
#include <cstdlib>

#include "dmr_kernels.hpp"
#include "dmr_sort.hpp"

namespace dmr {

constexpr int FI_CHUNK = 256;

#ifndef DMR_TET_FI_HOIST
#define DMR_TET_FI_HOIST 1
#endif
#if DMR_TET_FI_HOIST
// A staged face of the first-hit search with everything of ray_tri_hit (cuda_renderer/auxiliary.h:265-296) that does not depend
// on the ray's direction -- the origin is the same for all pixels of a view: T = o - p0, E1, E2, Q = T x E1, Q . E2 (the
// numerator of t) -- computed once per staged face with the reference's arithmetic instead of once per (pixel, face): 23 of a
// test's ~68 instructions in a kernel whose VALUs are busy 91 % of the time (profiles/r03/valu_mix_c3.txt).
struct alignas(16) HitRec { float T[3], E1[3], E2[3], Q[3]; float qe2, min_depth, max_depth; int face; };
static_assert(sizeof(HitRec) == 64, "HitRec");
#else
struct alignas(16) HitRec { float p0[3], p1[3], p2[3]; float min_depth, max_depth; int face; };
static_assert(sizeof(HitRec) == 48, "HitRec");
#endif

struct alignas(16) TetFaceRec { float p0[3], p1[3], p2[3], n[3]; int ft0, ft1; float opacity; int pad; };
static_assert(sizeof(TetFaceRec) == 64, "TetFaceRec");
// log1m = logf(1 - opacity), evaluated once per face by the same device function the march would call per step
struct alignas(16) TetColRec { float c0[3], c1[3], c2[3]; int v0, v1, v2; float opacity, log1m; int pad[2]; };
static_assert(sizeof(TetColRec) == 64, "TetColRec");

// Per tet: everything a march step needs of it, in ONE 224-byte record -- its four faces (id | slot of the face in the tet
// behind it << 29 | orientation flip << 31), the tets behind them (-1: none) and, per face, the three vertices and the unit
// normal before orientation (round 3: the vertex p0 and the edge vectors p1 - p0, p2 - p0).  A step used to be tet record -> three face records: two dependent levels of gathers; the march
// now carries the slot its current face has in its current tet, so the step's eleven loads go out together, one level.
struct alignas(16) TetBlock { int face[4]; int nbr[4]; float geo[4][12]; };
static_assert(sizeof(TetBlock) == 224, "TetBlock");
#ifndef DMR_TET_DUP_BIT
#define DMR_TET_DUP_BIT 1
#endif
#if DMR_TET_DUP_BIT
constexpr int TET_FACE_MASK = 0x0fffffff;  // (check_scene: F < 2^28)
// bit 28 of a face entry: the same face id sits in another slot of this tet as well (malformed tet_faces).  The march carries
// the slot of its current face, so the reference's "exactly one of the tet's four faces is the current one" (`cnt != 3`,
// forward.cu:716-722) is: the slot's entry is the current face and has no duplicate -- one compare per step instead of four
// masks, four compares and a count, all of the 4.5-cycle kind in a kernel whose VALUs are busy all the time.
constexpr int TET_FACE_DUP = 0x10000000;
#else
constexpr int TET_FACE_MASK = 0x1fffffff;  // (check_scene: F < 2^29)
constexpr int TET_FACE_DUP = 0;
#endif

struct TetParams {
    int B, P, F, W, H, gx, gy, r0;
#ifdef DMR_ABLATION
    int dbg;  // DMR_ABLATE bits (ablation build only)
#endif
    const float* verts; const int* faces; const float* verts_color; const float* faces_opacity;
    const float* mv; const float* proj; const float* inv_mv; const float* inv_proj;
    const float* faces_intense; const float* bg;
    const int* tets; const int* face_tets; const int* tet_faces;
    const int* seed;  // ray_random_seed of the forward, kept in the image buffer for the backward
    const TetFaceRec* facerec; const TetColRec* colrec; const TetBlock* tetrec;
    TetImageState img;
};

// ---------------------------------------------------------------------------
// Packed march records, rebuilt by every forward call (geometry and colours are inputs) and kept in the
// face buffer for the backward.  A march step of the reference (cuda_renderer/forward.cu:704-767) is a
// chain of narrow dependent gathers: tet_faces -> faces -> verts for four faces, tets -> verts for the
// centre, then four normalised outward normals (sqrt + divide each).  Everything that does not depend
// on the ray is hoisted here, with the reference's arithmetic, so the decisions stay bit-identical:
//   TetFaceRec : the three vertices, the UNIT normal before orientation, face_tets, opacity -- one 64-byte line
//   TetColRec  : the three vertex colours and ids, the opacity and logf(1 - opacity) -- one 64-byte line
//   TetBlock   : per tet (above); bit 31 of a face entry is set where tet_face_outward_normal flips the unit normal
//                (dot(n, centre - p0) > 0, cuda_renderer/auxiliary.h:386-392); dot(-n, d) == -dot(n, d) exactly; the tet
//                behind a face follows the reference's rule (the first entry of face_tets[face] that is neither this tet
//                nor -1, forward.cu:761-767).
// A step is then ONE level of 16-byte loads (TetBlock), then the chosen face's TetColRec for the shading.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_tet_prep_faces(int F, const float* __restrict__ verts, const int* __restrict__ faces,
                 const float* __restrict__ verts_color, const float* __restrict__ faces_opacity,
                 const int* __restrict__ face_tets, TetFaceRec* __restrict__ facerec, TetColRec* __restrict__ colrec,
                 int seed, int* __restrict__ seed_slot, TetSeq* __restrict__ seq, uint32_t seq_steps, unsigned long long seq_offset) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f == 0) {
        *seed_slot = seed;
        seq->max_steps = 0u; seq->cap_steps = seq_steps; seq->offset = seq_offset;  // the march sequence of this forward (dmr_kernels.hpp)
    }
    if (f >= F) return;
    const int v0 = faces[3 * f], v1 = faces[3 * f + 1], v2 = faces[3 * f + 2];
    const V3 p0 = load_v3(verts, v0), p1 = load_v3(verts, v1), p2 = load_v3(verts, v2);
    V3 n = cross(p1 - p0, p2 - p0);  // cuda_renderer/auxiliary.h:375-384
    float n_norm = sqrtf(dot(n, n));
    n_norm = fmaxf(n_norm, 0.0001f);
    n = n / n_norm;
    TetFaceRec r;
    r.p0[0] = p0.x; r.p0[1] = p0.y; r.p0[2] = p0.z;
    r.p1[0] = p1.x; r.p1[1] = p1.y; r.p1[2] = p1.z;
    r.p2[0] = p2.x; r.p2[1] = p2.y; r.p2[2] = p2.z;
    r.n[0] = n.x; r.n[1] = n.y; r.n[2] = n.z;
    r.ft0 = face_tets[2 * f]; r.ft1 = face_tets[2 * f + 1];
    r.opacity = faces_opacity[f]; r.pad = 0;
    facerec[f] = r;
    const V3 c0 = load_v3(verts_color, v0), c1 = load_v3(verts_color, v1), c2 = load_v3(verts_color, v2);
    TetColRec c;
    c.c0[0] = c0.x; c.c0[1] = c0.y; c.c0[2] = c0.z;
    c.c1[0] = c1.x; c.c1[1] = c1.y; c.c1[2] = c1.z;
    c.c2[0] = c2.x; c.c2[1] = c2.y; c.c2[2] = c2.z;
    c.v0 = v0; c.v1 = v1; c.v2 = v2;
    c.opacity = r.opacity; c.log1m = logf(1.0f - r.opacity); c.pad[0] = c.pad[1] = 0;
    colrec[f] = c;
}

__global__ void __launch_bounds__(256)
k_tet_prep_tets(int T, int F, const float* __restrict__ verts, const int* __restrict__ faces, const int* __restrict__ tets,
                const int* __restrict__ tet_faces, const int* __restrict__ face_tets, TetBlock* __restrict__ tetrec) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const V3 center = tet_center(verts, tets, t);
    TetBlock r;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int f = tet_faces[4 * t + i];
        int nbr = -1;
        V3 p0 = {0, 0, 0}, p1 = p0, p2 = p0, n = p0;
        if (f >= 0 && f < F) {
            p0 = load_v3(verts, faces[3 * f]); p1 = load_v3(verts, faces[3 * f + 1]); p2 = load_v3(verts, faces[3 * f + 2]);
            n = cross(p1 - p0, p2 - p0);  // as k_tet_prep_faces (cuda_renderer/auxiliary.h:375-384): the same bits
            float n_norm = sqrtf(dot(n, n));
            n_norm = fmaxf(n_norm, 0.0001f);
            n = n / n_norm;
            const int ft0 = face_tets[2 * f], ft1 = face_tets[2 * f + 1];
            if (!(ft0 == t || ft0 == -1)) nbr = ft0;
            else if (!(ft1 == t || ft1 == -1)) nbr = ft1;
            int nslot = 0;  // the slot of this face in the tet behind it (first match, like every look-up of the march)
            if (nbr >= 0 && nbr < T) {
                for (int q = 3; q >= 0; q--) if (tet_faces[4 * nbr + q] == f) nslot = q;
            }
            const bool flip = dot(n, center - p0) > 0.0f;
            bool dup = false;  // (as the reference counts: any other entry with the same id)
            for (int q = 0; q < 4; q++) if (q != i && tet_faces[4 * t + q] == f) dup = true;
            f = f | (nslot << 29) | (flip ? (int)0x80000000 : 0) | (dup ? TET_FACE_DUP : 0);
        }
        r.face[i] = f; r.nbr[i] = nbr;
        r.geo[i][0] = p0.x; r.geo[i][1] = p0.y; r.geo[i][2] = p0.z;
#ifndef DMR_TET_EDGES
#define DMR_TET_EDGES 1
#endif
#if DMR_TET_EDGES
        // (the edge vectors, not the vertices: ray_tri_hit's p1 - p0 and p2 - p0, taken here once per tet and face -- the march
        // is VALU-bound, profiles/r03/valu_mix_c3.txt, and these were 18 of a step's ~420 instructions)
        const V3 E1 = p1 - p0, E2 = p2 - p0;
        r.geo[i][3] = E1.x; r.geo[i][4] = E1.y; r.geo[i][5] = E1.z;
        r.geo[i][6] = E2.x; r.geo[i][7] = E2.y; r.geo[i][8] = E2.z;
#else
        r.geo[i][3] = p1.x; r.geo[i][4] = p1.y; r.geo[i][5] = p1.z;
        r.geo[i][6] = p2.x; r.geo[i][7] = p2.y; r.geo[i][8] = p2.z;
#endif
        r.geo[i][9] = n.x; r.geo[i][10] = n.y; r.geo[i][11] = n.z;
    }
    tetrec[t] = r;
}

// the slot `face` has in `tet` (first match; 0 if there is none: the march's own check stops it then)
__device__ __forceinline__ int tet_slot_of(const TetBlock* __restrict__ tetrec, int tet, int face) {
    const int4 hd = *reinterpret_cast<const int4*>(tetrec[tet].face);
    int slot = 0;
    if ((hd.w & TET_FACE_MASK) == face) slot = 3;
    if ((hd.z & TET_FACE_MASK) == face) slot = 2;
    if ((hd.y & TET_FACE_MASK) == face) slot = 1;
    if ((hd.x & TET_FACE_MASK) == face) slot = 0;
    return slot;
}

__device__ __forceinline__ TetFaceRec load_facerec(const TetFaceRec* __restrict__ a, int f) {
    const float4* q = reinterpret_cast<const float4*>(a + f);
    const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    TetFaceRec r;
    r.p0[0] = q0.x; r.p0[1] = q0.y; r.p0[2] = q0.z; r.p1[0] = q0.w;
    r.p1[1] = q1.x; r.p1[2] = q1.y; r.p2[0] = q1.z; r.p2[1] = q1.w;
    r.p2[2] = q2.x; r.n[0] = q2.y; r.n[1] = q2.z; r.n[2] = q2.w;
    r.ft0 = __float_as_int(q3.x); r.ft1 = __float_as_int(q3.y); r.opacity = q3.z; r.pad = 0;
    return r;
}

// dot(outward normal of `face` as seen from `tet`, d): flip = bit 31 of the tet's record entry
__device__ __forceinline__ float oriented_dot(const TetFaceRec& r, bool flip, V3 d) {
    const float v = dot(V3{r.n[0], r.n[1], r.n[2]}, d);
    return flip ? -v : v;
}

// SORT: the workgroup first sorts its tile's list (dmr_sort.hpp; keys = the scatter pass's unsorted entries), as the tri
// forward does: the sort kernel of its own was the serial chain of the longest tile on a nearly idle chip.
template <bool SORT>
__global__ void __launch_bounds__(256)
k_tet_first_intersect(TetParams p, const float* __restrict__ key_depth, const float* __restrict__ max_depth,
                      const uint32_t* __restrict__ tile_offset, uint32_t* __restrict__ face_list, uint64_t* __restrict__ keys,
                      uint32_t capacity) {
    constexpr int REC_BYTES = FI_CHUNK * (int)sizeof(HitRec);
    __shared__ __attribute__((aligned(16))) unsigned char s_mem[SORT && SORT_LDS_BYTES > REC_BYTES ? SORT_LDS_BYTES : REC_BYTES];
    HitRec* const s_rec = reinterpret_cast<HitRec*>(s_mem);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tx = blockIdx.x, ty = blockIdx.y + p.r0, b = blockIdx.z;
    const int lx = (wave & 1) * 8 + (lane & 7), ly = (wave >> 1) * 8 + (lane >> 3);
    const int px = tx * TILE + lx, py = ty * TILE + ly;
    const bool inside = px < p.W && py < p.H;
    const int64_t bpix = (int64_t)b * p.H * p.W + (int64_t)p.W * py + px;

    V3 ro = {0, 0, 0}, rd = {0, 0, 0};
    if (inside) pixel_ray<true>(p.inv_mv + 16 * b, p.inv_proj + 16 * b, px, py, p.W, p.H, ro, rd, *p.seed, (uint64_t)bpix);
    const V3 view_o = {p.inv_mv[16 * b + 12], p.inv_mv[16 * b + 13], p.inv_mv[16 * b + 14]};  // = ro of every pixel of the view

    const int tile = (b * p.gy + ty) * p.gx + tx;
    // (a list beyond the buffer -- only while a size guess is being refuted, the result is redone then -- holds no
    // valid face ids: the tile is treated as empty)
    uint32_t begin = tile_offset[tile], end = tile_offset[tile + 1];
    if (end > capacity) begin = end = 0u;
    if (SORT && begin != end) {  // uniform
        sort_tile(begin, end - begin, keys, face_list, reinterpret_cast<uint64_t*>(s_mem),
                  reinterpret_cast<uint32_t*>(s_mem + SORT_LDS_KEYS * sizeof(uint64_t)), (uint32_t)tid);
        __syncthreads();  // face_list[begin, end) is sorted and visible to this workgroup; the LDS is free
    }

    bool done = !inside;
    float min_T = -1.0f, min_T_max_depth = -1.0f;
    int ff = -1;
    __shared__ uint32_t s_live[2];
    AllDone all_done;
    all_done.init(s_live);
    for (uint32_t base = begin; base < end; base += FI_CHUNK) {
        if (all_done.barrier(done)) break;
        const int n = (int)min((uint32_t)FI_CHUNK, end - base);
        if (tid < n) {
            const int face = (int)face_list[base + tid];
            const V3 a = load_v3(p.verts, p.faces[3 * face]);
            const V3 c = load_v3(p.verts, p.faces[3 * face + 1]);
            const V3 e = load_v3(p.verts, p.faces[3 * face + 2]);
            HitRec& r = s_rec[tid];
#if DMR_TET_FI_HOIST
            const V3 hT = view_o - a, hE1 = c - a, hE2 = e - a;
            const V3 hQ = cross(hT, hE1);
            r.T[0] = hT.x; r.T[1] = hT.y; r.T[2] = hT.z;
            r.E1[0] = hE1.x; r.E1[1] = hE1.y; r.E1[2] = hE1.z;
            r.E2[0] = hE2.x; r.E2[1] = hE2.y; r.E2[2] = hE2.z;
            r.Q[0] = hQ.x; r.Q[1] = hQ.y; r.Q[2] = hQ.z;
            r.qe2 = dot(hQ, hE2);
#else
            r.p0[0] = a.x; r.p0[1] = a.y; r.p0[2] = a.z;
            r.p1[0] = c.x; r.p1[1] = c.y; r.p1[2] = c.z;
            r.p2[0] = e.x; r.p2[1] = e.y; r.p2[2] = e.z;
#endif
            r.min_depth = key_depth[(int64_t)b * p.F + face];
            r.max_depth = max_depth[(int64_t)b * p.F + face];
            r.face = face;
        }
        __syncthreads();
        for (int j = 0; !done && j < n; j++) {
            const HitRec& r = s_rec[j];
            if (min_T >= 0.0f && r.min_depth > min_T_max_depth) { done = true; continue; }
            V3 tuv;
#if DMR_TET_FI_HOIST
            {
                const V3 E1 = {r.E1[0], r.E1[1], r.E1[2]}, E2 = {r.E2[0], r.E2[1], r.E2[2]};
                const V3 P = cross(rd, E2);
                const float denom = dot(P, E1);
                if (denom == 0.0f) continue;
                const float inv_denom = 1.0f / denom;
                tuv.x = r.qe2 * inv_denom;
                tuv.y = dot(P, {r.T[0], r.T[1], r.T[2]}) * inv_denom;
                tuv.z = dot({r.Q[0], r.Q[1], r.Q[2]}, rd) * inv_denom;
                if (!(tuv.x >= 0.0f && tuv.y >= 0.0f && tuv.z >= 0.0f && tuv.y + tuv.z <= 1.0f)) continue;
            }
#else
            if (!ray_tri_hit(ro, rd, {r.p0[0], r.p0[1], r.p0[2]}, {r.p1[0], r.p1[1], r.p1[2]},
                             {r.p2[0], r.p2[1], r.p2[2]}, tuv))
                continue;
#endif
            if (min_T < 0.0f || tuv.x < min_T) { min_T = tuv.x; min_T_max_depth = r.max_depth; ff = r.face; }
        }
    }
    if (!inside) return;
    int ft = -1;
    if (ff >= 0) {
        const TetFaceRec fr = load_facerec(p.facerec, ff);
        for (int i = 0; i < 2; i++) {
            const int tet_id = i == 0 ? fr.ft0 : fr.ft1;
            if (tet_id < 0) continue;
            const int4 tr = *reinterpret_cast<const int4*>(p.tetrec[tet_id].face);
            const int e[4] = {tr.x, tr.y, tr.z, tr.w};
            bool flip = false;
#pragma unroll
            for (int q = 0; q < 4; q++) if ((e[q] & TET_FACE_MASK) == ff) flip = e[q] < 0;
            if (oriented_dot(fr, flip, rd) < 0.0f) ft = tet_id;
        }
    }
    p.img.first_face[bpix] = ff;
    p.img.first_tet[bpix] = ft;
}

// (t, u, v) of the ray on `face` and the face's unit normal (carried from step to step by the march)
__device__ __forceinline__ void face_tuv(const TetParams& p, V3 ro, V3 rd, int face, float& rt, float& iu, float& iv, V3& n) {
    V3 tuv = {0, 0, 0};
    const TetFaceRec r = load_facerec(p.facerec, face);
    ray_tri_hit(ro, rd, {r.p0[0], r.p0[1], r.p0[2]}, {r.p1[0], r.p1[1], r.p1[2]}, {r.p2[0], r.p2[1], r.p2[2]}, tuv);
    rt = tuv.x; iu = tuv.y; iv = tuv.z;
    n = {r.n[0], r.n[1], r.n[2]};
}

// One march step shared by forward (FWD: leave through the face whose outward normal follows
// the ray) and backward (enter face: normal against the ray).  Returns false when the march
// must stop ("error cases" 1-3 of the reference).
// FWD only, `back_amb`: one of the tet's other faces is hit with its outward normal AGAINST the ray, i.e. the reference's
// reverse march, arriving in this tet through the face chosen here, would find a second candidate next to the face the
// forward came in through and stop ("error case 3", backward.cu:456-460).  Same tests, same bits: the forward's entry face
// always qualifies there (it was accepted as a hit with this ray, and its normal is checked below), so this flag is all the
// backward needs to know of this tet.
template <bool FWD>
__device__ __forceinline__ bool march_step(const TetParams& p, V3 ro, V3 rd, int& curr_face, int& curr_tet, int& curr_slot,
                                           float& curr_rt, float& curr_iu, float& curr_iv, float& curr_dn, bool* back_amb = nullptr) {
    // The tet's record: header and the three faces other than the current one (whose slot the march carries), all requested
    // at once.  `others` keep the record's order, as the reference's loop over tet_faces does.
    const char* base = reinterpret_cast<const char*>(p.tetrec + curr_tet);
    const bool s1 = curr_slot == 0, s2 = curr_slot <= 1, s3 = curr_slot <= 2;  // entry i of `others` is slot i + (shift i)
    const float4* g0 = reinterpret_cast<const float4*>(base + 32 + 48 * (s1 ? 1 : 0));
    const float4* g1 = reinterpret_cast<const float4*>(base + 32 + 48 * (s2 ? 2 : 1));
    const float4* g2 = reinterpret_cast<const float4*>(base + 32 + 48 * (s3 ? 3 : 2));
    const int4 tr = reinterpret_cast<const int4*>(base)[0], nb = reinterpret_cast<const int4*>(base)[1];
    const float4 a0 = g0[0], a1 = g0[1], a2 = g0[2];
    const float4 b0 = g1[0], b1 = g1[1], b2 = g1[2];
    const float4 c0 = g2[0], c1 = g2[1], c2 = g2[2];
#if DMR_TET_DUP_BIT
    const int cur_e = s1 ? tr.x : (s2 ? tr.y : (s3 ? tr.z : tr.w));  // the entry of the slot the march carries
    if ((cur_e & (TET_FACE_MASK | TET_FACE_DUP)) != curr_face) return false;  // the reference's `cnt != 3` (see TET_FACE_DUP)
    const bool cur_flip = cur_e < 0;
#else
    const int t0 = tr.x & TET_FACE_MASK, t1 = tr.y & TET_FACE_MASK, t2 = tr.z & TET_FACE_MASK, t3 = tr.w & TET_FACE_MASK;
    const bool m0 = t0 == curr_face, m1 = t1 == curr_face, m2 = t2 == curr_face, m3 = t3 == curr_face;
    if ((int)m0 + (int)m1 + (int)m2 + (int)m3 != 1) return false;  // the reference's `cnt != 3` (then the slot is the match's)
    const bool cur_flip = (m0 ? tr.x : (m1 ? tr.y : (m2 ? tr.z : tr.w))) < 0;
#endif
    const int r0e = s1 ? tr.y : tr.x, r1e = s2 ? tr.z : tr.y, r2e = s3 ? tr.w : tr.z;
    const int nb0 = s1 ? nb.y : nb.x, nb1 = s2 ? nb.z : nb.y, nb2 = s3 ? nb.w : nb.z;
    bool ok = true;
    const float dcur0 = curr_dn;  // dot(the current face's unit normal, rd): the previous step computed it when it chose the face
    const float dcur = cur_flip ? -dcur0 : dcur0;
    if (FWD ? (dcur >= 0.0f) : (dcur <= 0.0f)) ok = false;
    int nf = -1, ncnt = 0, nt = -1, ns = 0;
    float nrt = 0, niu = 0, niv = 0;
    float ndn = 0.f;
    bool amb = false;
    // q0..q2: p0, p1, p2, unit normal of the candidate; e: its header entry; behind: the tet on its other side
    // (a face id outside [0, F) -- malformed tet_faces -- never hits)
    auto test = [&](float4 q0, float4 q1, float4 q2, int e, int behind) {
        const int of = e & TET_FACE_MASK;
        V3 tuv;
#if DMR_TET_EDGES
        const bool hit = ray_tri_hit_edges(ro, rd, {q0.x, q0.y, q0.z}, {q0.w, q1.x, q1.y}, {q1.z, q1.w, q2.x}, tuv) && (unsigned)of < (unsigned)p.F;
#else
        const bool hit = ray_tri_hit(ro, rd, {q0.x, q0.y, q0.z}, {q0.w, q1.x, q1.y}, {q1.z, q1.w, q2.x}, tuv) && (unsigned)of < (unsigned)p.F;
#endif
        const V3 n = {q2.y, q2.z, q2.w};
        const float dn0 = dot(n, rd);
        const float dn = e < 0 ? -dn0 : dn0;
        if (hit && (FWD ? (dn > 0.0f) : (dn < 0.0f))) {
            nf = of; nrt = tuv.x; niu = tuv.y; niv = tuv.z; nt = behind; ns = (e >> 29) & 3; ndn = dn0; ncnt++;
        }
        if (FWD && hit && dn < 0.0f) amb = true;
    };
    test(a0, a1, a2, r0e, nb0);
    test(b0, b1, b2, r1e, nb1);
    test(c0, c1, c2, r2e, nb2);
    if (ncnt != 1 || !ok) return false;
    if (FWD && back_amb) *back_amb = amb;
    curr_face = nf; curr_tet = nt; curr_slot = ns; curr_rt = nrt; curr_iu = niu; curr_iv = niv; curr_dn = ndn;
    return true;
}

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {  // all 64 lanes must be active
#pragma unroll
    for (int dlt = 32; dlt > 0; dlt >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, dlt, 64));
    return v;
}

#ifndef DMR_TET_FWD_WAVES
#define DMR_TET_FWD_WAVES 4
#endif
__global__ void __launch_bounds__(256, DMR_TET_FWD_WAVES)
k_tet_forward(TetParams p, float* __restrict__ out_color, float* __restrict__ out_depth, float* __restrict__ out_active) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tx = blockIdx.x, ty = blockIdx.y + p.r0, b = blockIdx.z;
    const int px = tx * TILE + (wave & 1) * 8 + (lane & 7), py = ty * TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < p.W && py < p.H;  // (lanes outside stay for the wave-level bookkeeping of the march sequence)
    const int64_t HW = (int64_t)p.H * p.W, pix_id = (int64_t)p.W * py + px, bpix = (int64_t)b * HW + pix_id;
    V3 ro = {0, 0, 0}, rd = {0, 0, 0};
    int first_face = -1, first_tet = -1;
    if (inside) {
        pixel_ray<true>(p.inv_mv + 16 * b, p.inv_proj + 16 * b, px, py, p.W, p.H, ro, rd, *p.seed, (uint64_t)bpix);
        first_face = p.img.first_face[bpix]; first_tet = p.img.first_tet[bpix];
    }
    const float* mv = p.mv + 16 * b;
    const float* pr = p.proj + 16 * b;

    bool done = false;
    int curr_face = first_face, curr_tet = first_tet;
    float curr_rt = 0.f, curr_iu = 0.f, curr_iv = 0.f;
    float curr_dn = 0.f;  // dot(curr_face's unit normal before orientation, rd)
    int curr_slot = 0;  // the slot of curr_face in curr_tet's record
    if (first_face == -1 || first_tet == -1) done = true;
    else {
        V3 curr_n = {0, 0, 0};
        face_tuv(p, ro, rd, first_face, curr_rt, curr_iu, curr_iv, curr_n);
        curr_dn = dot(curr_n, rd);
        curr_slot = tet_slot_of(p.tetrec, first_tet, first_face);
    }

    // the march sequence (dmr_kernels.hpp): this lane's 16-byte words, one per four steps, 64 words apart
    const uint32_t seq_cap = p.img.seq->cap_steps;
    uint4* const seq_row = reinterpret_cast<uint4*>(p.img.binning + p.img.seq->offset) +
                           ((((size_t)b * p.gy + ty) * p.gx + tx) * 4 + wave) * (size_t)(seq_cap / 4u) * 64u + (uint32_t)lane;
    bool back_amb = false;

    V3 C = {0, 0, 0};
    float D = 0.f, log_T = 0.f, prev_log_T = 0.f;
    float T_cur = expf(log_T);  // expf(log_T), carried from step to step (the reference evaluates it twice per step)
    int last_face = -1, last_tet = -1;
    bool active = false;
    uint32_t n_contrib = 0;
    // One 4-byte store per step, straight to the lane's dword of the row (measured at C3, k_tet_forward: no sequence 213 us,
    // this 225; four entries kept in registers and stored as one 16-byte word every fourth step: 272 -- with or without the
    // store itself, the selects that pick the register cost more than three stores).
    auto seq_put = [&](uint32_t s, int face, bool amb) {  // entry s of this lane
        if (s < seq_cap) reinterpret_cast<uint32_t*>(seq_row + (size_t)(s >> 2) * 64u)[s & 3u] = (uint32_t)face | (amb ? 0x80000000u : 0u);
    };
    while (!done) {
        seq_put(n_contrib, curr_face, back_amb);
        const float4* cq = reinterpret_cast<const float4*>(p.colrec + curr_face);
        const float4 cq0 = cq[0], cq1 = cq[1], cq2 = cq[2], cq3 = cq[3];
        const V3 c0 = {cq0.x, cq0.y, cq0.z}, c1 = {cq0.w, cq1.x, cq1.y}, c2 = {cq1.z, cq1.w, cq2.x};
        V3 col = (c0 + (c1 - c0) * curr_iu + (c2 - c0) * curr_iv);  // Q21
        const float opacity = cq3.x;
        const float intense = p.faces_intense[(int64_t)b * p.F + curr_face];
        col = col * intense;
        const float tmp_T = T_cur;  // expf(log_T): the value the previous step computed for its termination test
        C = C + tmp_T * opacity * col;
        const V3 pt = ro + (rd * curr_rt);
        const V4 pn = xform4x4(xform4x3(pt, mv), pr);
#ifndef DMR_TET_FWD_RCP
#define DMR_TET_FWD_RCP 1
#endif
        // (the hit point's depth decides nothing: 1-ulp reciprocal instead of the IEEE division's ten instructions -- the march is
        // VALU-bound; out_depth moves by ~1e-7 of its value, the tolerance is 1e-5)
        const float pw = DMR_TET_FWD_RCP ? __builtin_amdgcn_rcpf(clamp_w(pn.w)) : 1.0f / clamp_w(pn.w);
        D += tmp_T * opacity * (pn.z * pw);
        prev_log_T = log_T;
        if (opacity < 1.0f) log_T += cq3.y;  // logf(1 - opacity), per face (TetColRec)
        else log_T = logf(T_EPS * 0.1f);
        T_cur = expf(log_T);
        if (T_cur < T_EPS) { done = true; active = true; }
        n_contrib++;
        last_face = curr_face;
        last_tet = curr_tet;
        if (curr_tet == -1) { active = true; done = true; }
        if (!done && !march_step<true>(p, ro, rd, curr_face, curr_tet, curr_slot, curr_rt, curr_iu, curr_iv, curr_dn, &back_amb)) done = true;
    }
    {   // the wave's longest march (is the sequence complete? the next call's estimate)
        const uint32_t steps = wave_max_u32(n_contrib);
        if (lane == 0 && steps != 0u) atomicMax(&p.img.seq->max_steps, steps);
    }
    if (!inside) return;
    p.img.final_log_T[bpix] = log_T;
    p.img.final_prev_log_T[bpix] = prev_log_T;
    p.img.last_face[bpix] = last_face;
    p.img.last_tet[bpix] = last_tet;
    p.img.n_contrib[bpix] = n_contrib;
    p.img.is_active[bpix] = active ? 1 : 0;
    if (active) {
        const float fT = expf(log_T);
        out_color[((int64_t)b * 3 + 0) * HW + pix_id] = C.x + fT * p.bg[0];
        out_color[((int64_t)b * 3 + 1) * HW + pix_id] = C.y + fT * p.bg[1];
        out_color[((int64_t)b * 3 + 2) * HW + pix_id] = C.z + fT * p.bg[2];
        out_depth[bpix] = D + fT * 1.0f;
        out_active[bpix] = 1.0f;
    } else {
        out_color[((int64_t)b * 3 + 0) * HW + pix_id] = p.bg[0];
        out_color[((int64_t)b * 3 + 1) * HW + pix_id] = p.bg[1];
        out_color[((int64_t)b * 3 + 2) * HW + pix_id] = p.bg[2];
        out_depth[bpix] = 1.0f;
        out_active[bpix] = 0.0f;
    }
}

// Gradient accumulation of the tet backward.  The reference issues 10 scattered global atomics per
// (pixel, face) (cuda_renderer/backward.cu:353-360); at C3 that is 240 M memory-side requests (19.6 ms
// measured on MI355X).  The pixels of a 16x16 tile march through nearly the same faces, so the tile
// first sums per face in an LDS hash table (key = face id, 10 cells, ds_cmpst + ds_add_f64) and touches
// global memory once per (tile, face): 3 vertex-colour rows + the opacity.  A full table or a long probe
// sequence falls back to the direct atomics.  The cells are DOUBLES because of the atomic's rate, not its
// precision: ds_add_f32 retires one lane per ~3 cycles per CU, ds_add_f64 ten times that
// (scripts/micro/lds_atomics.hip: 194 vs 20 cycles per conflict-free wave instruction), and this kernel was
// bound by exactly those adds (10 per marched face and pixel).
#ifndef DMR_TET_TBL
#define DMR_TET_TBL 512
#endif
#ifndef DMR_TET_BWD_WAVES
#define DMR_TET_BWD_WAVES 1
#endif
constexpr int TET_TBL = DMR_TET_TBL;     // slots (a multiple of 16)
constexpr int TET_PROBES = 8;

struct TetAccum {
    int* key; double (*val)[TET_TBL];
    __device__ __forceinline__ int find(int face) const {
        uint32_t slot = __umulhi((uint32_t)face * 2654435761u, (uint32_t)TET_TBL);
        for (int i = 0; i < TET_PROBES; i++) {
            const int prev = atomicCAS(&key[slot], -1, face);
            if (prev == -1 || prev == face) return (int)slot;
            slot = slot + 1u == (uint32_t)TET_TBL ? 0u : slot + 1u;
        }
        return -1;
    }
};

// Arithmetic of k_tet_backward_seq that decides nothing (no index, no branch of the march depends on it): contracted to FMA,
// 1-ulp reciprocals and the hardware exponential.  Gradients are checked to 1e-4; the forward and the re-marching kernel,
// whose tests pick faces, keep the exact forms.
#pragma clang fp contract(fast)
namespace tfast {
struct F3 { float x, y, z; };
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ F3 operator*(float b, F3 a) { return {b * a.x, b * a.y, b * a.z}; }
__device__ __forceinline__ float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ F3 cross(F3 a, F3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// (t, u, v) of cuda_renderer/auxiliary.h:265-296 without the hit test (the march already decided)
__device__ __forceinline__ void tuv(F3 o, F3 d, F3 p0, F3 p1, F3 p2, float& t, float& u, float& v) {
    const F3 T = o - p0, E1 = p1 - p0, E2 = p2 - p0;
    const F3 P = cross(d, E2), Q = cross(T, E1);
    const float inv = rcp(dot(P, E1));
    t = dot(Q, E2) * inv; u = dot(P, T) * inv; v = dot(Q, d) * inv;
}
// ndc depth of a world point: rows z and w of proj * (mv * (pt, 1)) (auxiliary.h:71-90), z / clamp_w(w)
__device__ __forceinline__ float ndc_depth(F3 pt, const float* __restrict__ mv, const float* __restrict__ pr) {
    const float vx = mv[0] * pt.x + mv[4] * pt.y + mv[8] * pt.z + mv[12];
    const float vy = mv[1] * pt.x + mv[5] * pt.y + mv[9] * pt.z + mv[13];
    const float vz = mv[2] * pt.x + mv[6] * pt.y + mv[10] * pt.z + mv[14];
    const float cz = pr[2] * vx + pr[6] * vy + pr[10] * vz + pr[14];
    const float cw = pr[3] * vx + pr[7] * vy + pr[11] * vz + pr[15];
    return cz * rcp(clamp_w(cw));
}
}  // namespace tfast
#pragma clang fp contract(off)

// One pixel's state of the reverse walk and the gradient of one marched face (cuda_renderer/backward.cu:236-360),
// shared by the two backward kernels: k_tet_backward re-marches like the reference, k_tet_backward_seq takes the faces
// from the forward's march sequence.
struct TetBwdPixel {
    float dpc0, dpc1, dpc2, dpd, bg_dot, bd_dot, final_prev_T, final_T, prev_log_T;
    float last_alpha, lc0, lc1, lc2, ar0, ar1, ar2, last_depth, ard;
    bool first_iter;
    // -> g[0..8] = dL/d(vertex colours of the face), g[9] = dL/d(opacity); v0..v2: the face's vertices
    __device__ __forceinline__ void face_grad(const TetParams& p, int b, int face, V3 ro, V3 rd, const float* __restrict__ mv,
                                              const float* __restrict__ pr, float rt, float iu, float iv, float (&g)[10],
                                              int& v0, int& v1, int& v2) {
        const float4* cq = reinterpret_cast<const float4*>(p.colrec + face);
        face_grad(ro, rd, mv, pr, rt, iu, iv, cq[0], cq[1], cq[2], cq[3], p.faces_intense[(int64_t)b * p.F + face], g, v0, v1, v2);
    }
    __device__ __forceinline__ void face_grad(V3 ro, V3 rd, const float* __restrict__ mv, const float* __restrict__ pr, float rt,
                                              float iu, float iv, float4 cq0, float4 cq1, float4 cq2, float4 cq3, float intense,
                                              float (&g)[10], int& v0, int& v1, int& v2) {
        const V3 c0 = {cq0.x, cq0.y, cq0.z}, c1 = {cq0.w, cq1.x, cq1.y}, c2 = {cq1.z, cq1.w, cq2.x};
        v0 = __float_as_int(cq2.y); v1 = __float_as_int(cq2.z); v2 = __float_as_int(cq2.w);
        const float i0 = 1.0f - iu - iv, i1 = iu, i2 = iv;
        V3 col = (i0 * c0) + (i1 * c1) + (i2 * c2);  // Q21
        const float opacity = cq3.x;
        col = col * intense;
        const V3 pt = ro + (rd * rt);
        const V4 pn = xform4x4(xform4x3(pt, mv), pr);
        const float pw = 1.0f / clamp_w(pn.w);
        const float pdepth = pn.z * pw;
        if (!first_iter) prev_log_T = prev_log_T - cq3.y;  // logf(1 - opacity), per face (TetColRec)
        first_iter = false;
        const float prev_T = expf(prev_log_T);

        float dop = 0.f;
        ar0 = last_alpha * lc0 + (1.f - last_alpha) * ar0; lc0 = col.x;
        const float dc0 = dpc0 * opacity * prev_T; dop += (col.x - ar0) * dpc0;
        ar1 = last_alpha * lc1 + (1.f - last_alpha) * ar1; lc1 = col.y;
        const float dc1 = dpc1 * opacity * prev_T; dop += (col.y - ar1) * dpc1;
        ar2 = last_alpha * lc2 + (1.f - last_alpha) * ar2; lc2 = col.z;
        const float dc2 = dpc2 * opacity * prev_T; dop += (col.z - ar2) * dpc2;
        ard = last_alpha * last_depth + (1.f - last_alpha) * ard; last_depth = pdepth;
        dop += (pdepth - ard) * dpd;
        dop *= prev_T;
        last_alpha = opacity;
        if (opacity == 1.0f) {
            dop += (-final_prev_T) * bg_dot;
            dop += (-final_prev_T) * bd_dot;
        } else {
            dop += (-final_T / (1.f - opacity)) * bg_dot;
            dop += (-final_T / (1.f - opacity)) * bd_dot;
        }
        g[0] = i0 * dc0 * intense; g[1] = i0 * dc1 * intense; g[2] = i0 * dc2 * intense;
        g[3] = i1 * dc0 * intense; g[4] = i1 * dc1 * intense; g[5] = i1 * dc2 * intense;
        g[6] = i2 * dc0 * intense; g[7] = i2 * dc1 * intense; g[8] = i2 * dc2 * intense;
        g[9] = dop;
    }
    // the same with tfast's arithmetic, from the face's records (k_tet_backward_seq)
    __device__ __forceinline__ void face_grad_fast(V3 ro, V3 rd, const float* __restrict__ mv, const float* __restrict__ pr,
                                                   float4 f0, float4 f1, float4 f2, float4 cq0, float4 cq1, float4 cq2, float4 cq3,
                                                   float intense, float (&g)[10], int& v0, int& v1, int& v2) {
#pragma clang fp contract(fast)
        using namespace tfast;
        const F3 o = {ro.x, ro.y, ro.z}, d = {rd.x, rd.y, rd.z};
        float rt, iu, iv;
        tuv(o, d, {f0.x, f0.y, f0.z}, {f0.w, f1.x, f1.y}, {f1.z, f1.w, f2.x}, rt, iu, iv);
        const F3 c0 = {cq0.x, cq0.y, cq0.z}, c1 = {cq0.w, cq1.x, cq1.y}, c2 = {cq1.z, cq1.w, cq2.x};
        v0 = __float_as_int(cq2.y); v1 = __float_as_int(cq2.z); v2 = __float_as_int(cq2.w);
        const float i0 = 1.0f - iu - iv, i1 = iu, i2 = iv;
        const F3 col = intense * ((i0 * c0) + (i1 * c1) + (i2 * c2));  // Q21
        const float opacity = cq3.x;
        const float pdepth = ndc_depth(o + (rt * d), mv, pr);
        if (!first_iter) prev_log_T = prev_log_T - cq3.y;  // logf(1 - opacity), per face (TetColRec)
        first_iter = false;
        const float prev_T = __expf(prev_log_T);
        float dop = 0.f;
        ar0 = last_alpha * lc0 + (1.f - last_alpha) * ar0; lc0 = col.x;
        ar1 = last_alpha * lc1 + (1.f - last_alpha) * ar1; lc1 = col.y;
        ar2 = last_alpha * lc2 + (1.f - last_alpha) * ar2; lc2 = col.z;
        ard = last_alpha * last_depth + (1.f - last_alpha) * ard; last_depth = pdepth;
        dop = (col.x - ar0) * dpc0 + (col.y - ar1) * dpc1 + (col.z - ar2) * dpc2 + (pdepth - ard) * dpd;
        dop *= prev_T;
        last_alpha = opacity;
        const float tail = opacity == 1.0f ? -final_prev_T : -final_T * rcp(1.f - opacity);
        dop += tail * bg_dot + tail * bd_dot;
        const float sc = opacity * prev_T * intense;
        const float dc0 = dpc0 * sc, dc1 = dpc1 * sc, dc2 = dpc2 * sc;
        g[0] = i0 * dc0; g[1] = i0 * dc1; g[2] = i0 * dc2;
        g[3] = i1 * dc0; g[4] = i1 * dc1; g[5] = i1 * dc2;
        g[6] = i2 * dc0; g[7] = i2 * dc1; g[8] = i2 * dc2;
        g[9] = dop;
    }
};

// Adds the wave's (pixel, face) gradients of one step to the tile's table.  Called by ALL lanes of the wave (`act`: this lane
// has a gradient).  Neighbouring pixels march through the same faces in near lockstep, so the lanes of a wave pile onto a few
// table cells, and same-address LDS atomics serialise (profiles/r03: the LDS was busy 93 % of the kernel, 81 % of that
// in conflict cycles).  So lanes that hold the same face first merge pairwise, four butterfly levels inside a 16-lane
// DPP row (lane ^ 1, ^ 2 by quad_perm, + 4, + 8 by row shifts): at every level a live lane whose partner is alive with the
// same face takes the partner's ten values and the partner retires -- whatever the faces' layout over the lanes (round 2
// merged only aligned quads / octets / rows that were uniform).  What is still alive goes to the table.  Values are moved
// with selects, not multiplied by 0/1 masks: a non-finite gradient must not leak into another face's sums.
// One level: g += m * g[partner] with m = 1.0 where this lane takes its partner's values, else 0.0 -- one v_fmac_f32_dpp per
// value (a select + add is three instructions, and the DPP forms are half rate: 120 of this kernel's 530 VALU instructions per
// step were these merges).  Non-finite values never get here (tet_accumulate takes such lanes out first): 0 * inf would
// leak NaN into the partner's face.
#define DMR_TET_FMAC(N, DPP) "v_fmac_f32_dpp %[g" #N "], %[g" #N "], %[m] " DPP "\n\t"
#define DMR_TET_MERGE(DPP)                                                                                              \
    asm volatile("s_nop 1\n\t"                                                                                          \
                 DMR_TET_FMAC(0, DPP) DMR_TET_FMAC(1, DPP) DMR_TET_FMAC(2, DPP) DMR_TET_FMAC(3, DPP) DMR_TET_FMAC(4, DPP)  \
                 DMR_TET_FMAC(5, DPP) DMR_TET_FMAC(6, DPP) DMR_TET_FMAC(7, DPP) DMR_TET_FMAC(8, DPP) DMR_TET_FMAC(9, DPP)  \
                 : [g0] "+v"(g[0]), [g1] "+v"(g[1]), [g2] "+v"(g[2]), [g3] "+v"(g[3]), [g4] "+v"(g[4]),                  \
                   [g5] "+v"(g[5]), [g6] "+v"(g[6]), [g7] "+v"(g[7]), [g8] "+v"(g[8]), [g9] "+v"(g[9])                   \
                 : [m] "v"(m))
template <int LEVEL>
__device__ __forceinline__ void tet_merge_level(int lane, int& key, float (&g)[10]) {
    constexpr int offset = 1 << LEVEL;
    constexpr int FROM_UPPER = LEVEL == 0 ? 0xB1 : (LEVEL == 1 ? 0x4E : (LEVEL == 2 ? 0x104 : 0x108));  // quad_perm ^1, ^2; row_shl:4, :8
    constexpr int FROM_LOWER = LEVEL == 0 ? 0xB1 : (LEVEL == 1 ? 0x4E : (LEVEL == 2 ? 0x114 : 0x118));  // ... row_shr:4, :8
    const bool lower = (lane & offset) == 0;
    const int k_up = __builtin_amdgcn_update_dpp(-1, key, FROM_UPPER, 0xF, 0xF, false);   // key of lane + offset (as seen by lower lanes)
    const int k_lo = __builtin_amdgcn_update_dpp(-1, key, FROM_LOWER, 0xF, 0xF, false);   // key of lane - offset (as seen by upper lanes)
    const float m = (lower && key >= 0 && k_up == key) ? 1.0f : 0.0f;
    const bool retire = !lower && key >= 0 && k_lo == key;
    if (LEVEL == 0) DMR_TET_MERGE("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:0");
    else if (LEVEL == 1) DMR_TET_MERGE("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:0");
    else if (LEVEL == 2) DMR_TET_MERGE("row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:0");
    else DMR_TET_MERGE("row_shl:8 row_mask:0xf bank_mask:0xf bound_ctrl:0");
    if (retire) key = -1;  // its values went to the partner (what it still holds is finite and is taken by nobody: its key says so)
}

__device__ __forceinline__ void tet_direct_atomics(const float (&g)[10], int face, int v0, int v1, int v2, float* __restrict__ dL_dvcolor,
                                                   float* __restrict__ dL_dfopacity) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
        atomicAdd(&dL_dvcolor[3 * v0 + c], g[c]);
        atomicAdd(&dL_dvcolor[3 * v1 + c], g[3 + c]);
        atomicAdd(&dL_dvcolor[3 * v2 + c], g[6 + c]);
    }
    atomicAdd(&dL_dfopacity[face], g[9]);
}

__device__ __forceinline__ void tet_accumulate(const TetParams& p, const TetAccum& acc, double (*s_val)[TET_TBL], int lane, bool act,
                                               int face, float (&g)[10], int v0, int v1, int v2, float* __restrict__ dL_dvcolor,
                                               float* __restrict__ dL_dfopacity) {
    if (act) {  // a lane with a non-finite value adds its ten values the reference's way and takes no part in the merges
        float chk = 0.f;
#pragma unroll
        for (int c = 0; c < 10; c++) chk = fmaf(g[c], 0.f, chk);
        if (!(chk == 0.f)) { tet_direct_atomics(g, face, v0, v1, v2, dL_dvcolor, dL_dfopacity); act = false; }
    }
    int key = act ? face : -1;  // -1: nothing (left) in this lane
    if (!act) {
#pragma unroll
        for (int c = 0; c < 10; c++) g[c] = 0.f;
    }
    tet_merge_level<0>(lane, key, g);
    tet_merge_level<1>(lane, key, g);
    tet_merge_level<2>(lane, key, g);
    tet_merge_level<3>(lane, key, g);
    if (key < 0) return;
    // (ablation build, DMR_ABLATE bit 2048, tests only: odd faces are refused a slot, which exercises the direct-atomic fallback)
    const int slot = (DMR_DBG(p, 2048) && (face & 1)) ? -1 : acc.find(face);
    if (slot >= 0) {
#pragma unroll
        for (int c = 0; c < 10; c++) atomicAdd(&s_val[c][slot], (double)g[c]);
    } else {
        tet_direct_atomics(g, face, v0, v1, v2, dL_dvcolor, dL_dfopacity);
    }
}

// flush of the tile's table: 16 lanes per slot (10 used): lanes 0-8 -> the three vertex-colour rows, lane 9 -> opacity
__device__ __forceinline__ void tet_flush(const TetParams& p, const int* s_key, double (*s_val)[TET_TBL], int tid,
                                          float* __restrict__ dL_dvcolor, float* __restrict__ dL_dfopacity) {
    const int sub = tid & 15;
    for (int s0 = 0; s0 < TET_TBL; s0 += 16) {
        const int slot = s0 + (tid >> 4);
        const int face = s_key[slot];
        if (face < 0 || sub > 9) continue;
        const float v = (float)s_val[sub][slot];
        if (sub < 9) atomicAdd(&dL_dvcolor[3 * p.faces[3 * face + sub / 3] + sub % 3], v);
        else atomicAdd(&dL_dfopacity[face], v);
    }
}

// what a pixel's reverse walk starts from; false: the pixel has no gradient (outside, inactive, nothing marched)
__device__ __forceinline__ bool tet_bwd_begin(const TetParams& p, int b, int px, int py, const float* __restrict__ dL_dcolor,
                                              const float* __restrict__ dL_ddepth, TetBwdPixel& st, V3& ro, V3& rd,
                                              int& first_face, int& last_face) {
    const int64_t HW = (int64_t)p.H * p.W, pix_id = (int64_t)p.W * py + px, bpix = (int64_t)b * HW + pix_id;
    bool work = px < p.W && py < p.H;
    if (work) work = p.img.is_active[bpix] != 0;
    last_face = -1;
    if (work) { last_face = p.img.last_face[bpix]; work = last_face != -1; }
    if (!work) return false;
    first_face = p.img.first_face[bpix];
    const float fprev = p.img.final_prev_log_T[bpix], flog = p.img.final_log_T[bpix];
    st.final_prev_T = expf(fprev); st.final_T = expf(flog);
    st.prev_log_T = fprev;
    st.dpc0 = dL_dcolor[((int64_t)b * 3 + 0) * HW + pix_id];
    st.dpc1 = dL_dcolor[((int64_t)b * 3 + 1) * HW + pix_id];
    st.dpc2 = dL_dcolor[((int64_t)b * 3 + 2) * HW + pix_id];
    st.dpd = dL_ddepth[bpix];
    float bg_dot = 0.f;
    bg_dot += p.bg[0] * st.dpc0; bg_dot += p.bg[1] * st.dpc1; bg_dot += p.bg[2] * st.dpc2;
    st.bg_dot = bg_dot;
    st.bd_dot = 0.f + (float)(1.0 * (double)st.dpd);
    st.last_alpha = 0.f; st.lc0 = st.lc1 = st.lc2 = 0.f; st.ar0 = st.ar1 = st.ar2 = 0.f; st.last_depth = 0.f; st.ard = 0.f;
    st.first_iter = true;
    pixel_ray<true>(p.inv_mv + 16 * b, p.inv_proj + 16 * b, px, py, p.W, p.H, ro, rd, *p.seed, (uint64_t)bpix);
    return true;
}

// The re-marching backward (the reference's algorithm): the fallback when the forward's march sequence is not there --
// no capacity estimate yet (first call of a view configuration) or a scene that outgrew it.  Decided on the device:
// both kernels are always launched and one of them returns at once.
// (Launched with a few workgroups per CU that loop over the band's tiles, not one per tile: when it is the idle one of the
// two launches -- every call but the first of a view configuration -- 2 500 workgroups that only find that out cost 6.7 us
// at C3, 768 cost under 2.)
__global__ void __launch_bounds__(256)
k_tet_backward(TetParams p, int rows, const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
               float* __restrict__ dL_dvcolor, float* __restrict__ dL_dfopacity) {
    {
        const uint32_t cap = p.img.seq->cap_steps;
        if (cap != 0u && p.img.seq->max_steps <= cap && !DMR_DBG(p, 8192)) return;  // uniform: k_tet_backward_seq does this call's work
    }
    __shared__ int s_key[TET_TBL];
    __shared__ double s_val[10][TET_TBL];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const TetAccum acc{s_key, s_val};
    const int ntiles = p.gx * rows * p.B;
    for (int ti = blockIdx.x; ti < ntiles; ti += gridDim.x) {   // tiles of the band [r0, r0 + rows), all views
        __syncthreads();  // (the previous tile's flush has read the table)
        for (int i = tid; i < TET_TBL; i += 256) {
            s_key[i] = -1;
#pragma unroll
            for (int c = 0; c < 10; c++) s_val[c][i] = 0.0;
        }
        __syncthreads();
        const int tx = ti % p.gx, ty = (ti / p.gx) % rows + p.r0, b = ti / (p.gx * rows);
        const int px = tx * TILE + (wave & 1) * 8 + (lane & 7), py = ty * TILE + (wave >> 1) * 8 + (lane >> 3);
        TetBwdPixel st;
        V3 ro = {0, 0, 0}, rd = {0, 0, 0};
        int first_face = -1, last_face = -1;
        bool done = !tet_bwd_begin(p, b, px, py, dL_dcolor, dL_ddepth, st, ro, rd, first_face, last_face);
        const float* mv = p.mv + 16 * b;
        const float* pr = p.proj + 16 * b;
        int curr_face = last_face, curr_tet = -1, curr_slot = 0;
        float curr_rt = 0.f, curr_iu = 0.f, curr_iv = 0.f;
        V3 curr_n = {0, 0, 0};
        float curr_dn = 0.f;  // dot(curr_face's unit normal before orientation, rd), carried from step to step (march_step)
        if (!done) {
            curr_tet = p.img.last_tet[(int64_t)b * p.H * p.W + (int64_t)p.W * py + px];
            face_tuv(p, ro, rd, last_face, curr_rt, curr_iu, curr_iv, curr_n);
            curr_dn = dot(curr_n, rd);
            // step back across the last face (backward.cu:223-232)
            for (int i = 0; i < 2; i++) {
                const int t = p.face_tets[2 * curr_face + i];
                if (t == curr_tet) continue;
                curr_tet = t;
                break;
            }
            if (curr_tet >= 0) curr_slot = tet_slot_of(p.tetrec, curr_tet, curr_face);
        }
        while (!__all(done)) {  // the wave's lanes stay together: tet_accumulate merges lanes that hold the same face
            const bool act = !done;
            float g[10];
            int v0 = 0, v1 = 0, v2 = 0;
            const int face = curr_face;
            if (act) {
                st.face_grad(p, b, curr_face, ro, rd, mv, pr, curr_rt, curr_iu, curr_iv, g, v0, v1, v2);
                if (curr_face == first_face) done = true;
                if (!done) {
                    if (curr_tet == -1) done = true;
                    else if (!march_step<false>(p, ro, rd, curr_face, curr_tet, curr_slot, curr_rt, curr_iu, curr_iv, curr_dn)) done = true;
                }
            }
            tet_accumulate(p, acc, s_val, lane, act, face, g, v0, v1, v2, dL_dvcolor, dL_dfopacity);
        }
        __syncthreads();
        tet_flush(p, s_key, s_val, tid, dL_dvcolor, dL_dfopacity);
    }
}

// The backward on the forward's march sequence (dmr_kernels.hpp): a wave walks its rows from the back, step s of all its
// pixels in the same iteration (a pixel joins at its own last step), one contiguous kilobyte per four steps.  Per step and
// pixel: ONE ray-triangle evaluation for (t, u, v) -- the function the reference's reverse march evaluates for the face it
// picked, here with contracted arithmetic (tfast: nothing is decided by it) -- instead of the tet record, three candidate
// records, three tests and the orientation logic.  Stops where the reference stops: behind first_face, or behind an entry
// whose bit 31 says the reverse march would find two candidates there.
__global__ void __launch_bounds__(256, DMR_TET_BWD_WAVES)
k_tet_backward_seq(TetParams p, const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
                   float* __restrict__ dL_dvcolor, float* __restrict__ dL_dfopacity, uint32_t* __restrict__ host_seq_steps) {
    const uint32_t seq_cap = p.img.seq->cap_steps, longest = p.img.seq->max_steps;
    if (host_seq_steps && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)
        *host_seq_steps = longest;  // pinned: the next forward's capacity estimate (whichever kernel does the work)
    if (seq_cap == 0u || longest > seq_cap || DMR_DBG(p, 8192)) return;  // uniform: no complete sequence, k_tet_backward re-marches
    __shared__ int s_key[TET_TBL];
    __shared__ double s_val[10][TET_TBL];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < TET_TBL; i += 256) {
        s_key[i] = -1;
#pragma unroll
        for (int c = 0; c < 10; c++) s_val[c][i] = 0.0;
    }
    __syncthreads();
    const TetAccum acc{s_key, s_val};

    const int tx = blockIdx.x, ty = blockIdx.y + p.r0, b = blockIdx.z;
    const int px = tx * TILE + (wave & 1) * 8 + (lane & 7), py = ty * TILE + (wave >> 1) * 8 + (lane >> 3);
    TetBwdPixel st;
    V3 ro = {0, 0, 0}, rd = {0, 0, 0};
    int first_face = -1, last_face = -1;
    const bool work = tet_bwd_begin(p, b, px, py, dL_dcolor, dL_ddepth, st, ro, rd, first_face, last_face);
    const uint32_t n = work ? min(p.img.n_contrib[(int64_t)b * p.H * p.W + (int64_t)p.W * py + px], seq_cap) : 0u;
    const uint32_t smax = wave_max_u32(n);  // the wave's first step from the back
    const uint4* const seq_row = reinterpret_cast<const uint4*>(p.img.binning + p.img.seq->offset) +
                                 ((((size_t)b * p.gy + ty) * p.gx + tx) * 4 + wave) * (size_t)(seq_cap / 4u) * 64u + (uint32_t)lane;
    if (smax != 0u) {  // (uniform per wave)
        const float* mv = p.mv + 16 * b;
        const float* pr = p.proj + 16 * b;
        // The load stream runs ahead of the arithmetic: the sequence says which records step s - 1 needs while step s is
        // being computed (the reference's -- and the fallback's -- reverse march cannot know: there a step is a chain of
        // dependent gathers, tet record -> three face records -> tests -> colour record).
        struct Recs { float4 f0, f1, f2, c0, c1, c2, c3; float intense; };
        auto load_recs = [&](uint32_t e, Recs& r) {
            if (e == 0xffffffffu) return;  // this lane has no such step (not joined yet, or below step 0)
            const int face = (int)(e & 0x7fffffffu);
            const float4* fq = reinterpret_cast<const float4*>(p.facerec + face);
            const float4* cq = reinterpret_cast<const float4*>(p.colrec + face);
            r.f0 = fq[0]; r.f1 = fq[1]; r.f2 = fq[2];
            r.c0 = cq[0]; r.c1 = cq[1]; r.c2 = cq[2]; r.c3 = cq[3];
            r.intense = p.faces_intense[(int64_t)b * p.F + face];
        };
        // entry t of this lane out of the row word it belongs to; never an id that is not a face's (-> "no such step")
        auto entry = [&](const uint4& w, int t) -> uint32_t {
            if (t < 0 || (uint32_t)t >= n) return 0xffffffffu;
            const uint32_t q = (uint32_t)t & 3u;
            const uint32_t e = q == 0u ? w.x : (q == 1u ? w.y : (q == 2u ? w.z : w.w));
            return (e & 0x7fffffffu) < (uint32_t)p.F ? e : 0xffffffffu;
        };
        auto load_row = [&](int t) -> uint4 {  // the row word of step t (uniform t; only lanes that have a step in that row read)
            uint4 w = make_uint4(0u, 0u, 0u, 0u);
            if (t >= 0 && ((uint32_t)t & ~3u) < n) w = seq_row[(size_t)((uint32_t)t >> 2) * 64u];
            return w;
        };
        Recs rc = {}, rn = {};
        uint4 w_cur = load_row((int)smax - 1);                     // row of the step being prepared (s - 1 below)
        uint4 w_nxt = load_row((((int)smax - 1) & ~3) - 1);        // the row below it, requested a row ahead
        uint32_t e_cur = entry(w_cur, (int)smax - 1);
        load_recs(e_cur, rc);
        bool done = !work;
        // one step: computes step s from `cur` while the records of step s - 1 arrive in `nxt`; true: the wave is finished.
        // (Two record sets used alternately: rotating one into the other was 29 register moves per step.  A ring of three,
        // two steps ahead: 175 VGPRs, 367 -> 400 us at C3 -- the latency is hidden at distance one.)
        auto one = [&](int s, const Recs& cur, Recs& nxt) -> bool {
            // the entry of step s - 1 comes from w_cur, or from w_nxt when s - 1 is the last step of the row below
            if ((s & 3) == 0) { w_cur = w_nxt; w_nxt = load_row(s - 5); }
            const uint32_t e_nxt = entry(w_cur, s - 1);
            load_recs(done ? 0xffffffffu : e_nxt, nxt);             // in flight during step s
            const bool act = !done && e_cur != 0xffffffffu;
            const int face = (int)(e_cur & 0x7fffffffu);
            float g[10];
            int v0 = 0, v1 = 0, v2 = 0;
            if (act) {  // (t, u, v) of the ray on this face, then the face's gradient
                st.face_grad_fast(ro, rd, mv, pr, cur.f0, cur.f1, cur.f2, cur.c0, cur.c1, cur.c2, cur.c3, cur.intense, g, v0, v1, v2);
                if (face == first_face || (e_cur & 0x80000000u)) done = true;
            }
            tet_accumulate(p, acc, s_val, lane, act, face, g, v0, v1, v2, dL_dvcolor, dL_dfopacity);
            e_cur = e_nxt;
            return __all(done);
        };
        for (int s = (int)smax - 1; s >= 0; s -= 2) {
            if (one(s, rc, rn)) break;
            if (s == 0 || one(s - 1, rn, rc)) break;
        }
    }
    __syncthreads();
    tet_flush(p, s_key, s_val, tid, dL_dvcolor, dL_dfopacity);
}

static TetParams make_params(const dmr_scene& s, int gx, int gy, int r0, TetImageState img) {
    TetParams p;
    p.B = s.B; p.P = s.P; p.F = s.F; p.W = s.W; p.H = s.H; p.gx = gx; p.gy = gy; p.r0 = r0;
#ifdef DMR_ABLATION
    { static const int dbg = getenv("DMR_ABLATE") ? atoi(getenv("DMR_ABLATE")) : 0; p.dbg = dbg; }  // ablation build only
#endif
    p.verts = s.verts; p.faces = s.faces; p.verts_color = s.verts_color; p.faces_opacity = s.faces_opacity;
    p.mv = s.mv_mats; p.proj = s.proj_mats; p.inv_mv = s.inv_mv_mats; p.inv_proj = s.inv_proj_mats;
    p.faces_intense = s.faces_intense; p.bg = s.background;
    p.tets = s.tets; p.face_tets = s.face_tets; p.tet_faces = s.tet_faces;
    p.seed = img.seed;
    p.facerec = reinterpret_cast<const TetFaceRec*>(img.facerec);
    p.colrec = reinterpret_cast<const TetColRec*>(img.colrec);
    p.tetrec = reinterpret_cast<const TetBlock*>(img.tetrec);
    p.img = img;
    return p;
}

size_t tet_facerec_bytes() { return sizeof(TetFaceRec); }
size_t tet_colrec_bytes() { return sizeof(TetColRec); }
size_t tet_tetrec_bytes() { return sizeof(TetBlock); }

void launch_tet_prep(const dmr_scene& s, TetImageState img, uint32_t seq_steps, unsigned long long seq_offset, hipStream_t st) {
    k_tet_prep_faces<<<dim3((unsigned)std::max(1, (s.F + 255) / 256)), dim3(256), 0, st>>>(
        s.F, s.verts, s.faces, s.verts_color, s.faces_opacity, s.face_tets,
        reinterpret_cast<TetFaceRec*>(img.facerec), reinterpret_cast<TetColRec*>(img.colrec), s.ray_random_seed, img.seed,
        img.seq, seq_steps, seq_offset);
    if (s.T > 0)
        k_tet_prep_tets<<<dim3((unsigned)((s.T + 255) / 256)), dim3(256), 0, st>>>(
            s.T, s.F, s.verts, s.faces, s.tets, s.tet_faces, s.face_tets, reinterpret_cast<TetBlock*>(img.tetrec));
}

void launch_tet_first_intersect(const dmr_scene& s, int gx, int gy, int r0, int r1, const float* key_depth,
                                const float* max_depth, const uint32_t* tile_offset, uint64_t* keys, uint32_t* face_list,
                                uint32_t capacity, TetImageState img, hipStream_t st) {
    if (r1 <= r0) return;
    TetParams p = make_params(s, gx, gy, r0, img);
    StageScope t(DMR_STAGE_TET_FIRST, st);
    const dim3 grid(gx, r1 - r0, s.B), block(256);
    if (keys) k_tet_first_intersect<true><<<grid, block, 0, st>>>(p, key_depth, max_depth, tile_offset, face_list, keys, capacity);
    else k_tet_first_intersect<false><<<grid, block, 0, st>>>(p, key_depth, max_depth, tile_offset, face_list, nullptr, capacity);
}

void launch_tet_forward(const dmr_scene& s, int gx, int gy, int r0, int r1, TetImageState img,
                        float* out_color, float* out_depth, float* out_active, hipStream_t st) {
    if (r1 <= r0) return;
    TetParams p = make_params(s, gx, gy, r0, img);
    StageScope t(DMR_STAGE_TET_FORWARD, st);
    k_tet_forward<<<dim3(gx, r1 - r0, s.B), dim3(256), 0, st>>>(p, out_color, out_depth, out_active);
}

// both gradient tensors zeroed by one launch (two hipMemsetAsync are three fill kernels of ~4.6 us each)
__global__ void __launch_bounds__(256)
k_tet_zero_grads(float* __restrict__ a, int64_t na, float* __restrict__ b, int64_t nb) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < na) a[i] = 0.f;
    else if (i - na < nb) b[i - na] = 0.f;
}

void launch_tet_zero_grads(float* dL_dvcolor, int64_t n_vcolor, float* dL_dfopacity, int64_t n_fopacity, hipStream_t st) {
    const int64_t n = n_vcolor + n_fopacity;
    if (n > 0) k_tet_zero_grads<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(dL_dvcolor, n_vcolor, dL_dfopacity, n_fopacity);
}

void launch_tet_backward(const dmr_scene& s, int gx, int gy, int r0, int r1, TetImageState img,
                         const float* dL_dcolor, const float* dL_ddepth, float* dL_dvcolor, float* dL_dfopacity,
                         uint32_t* host_seq_steps, hipStream_t st) {
    if (r1 <= r0) return;
    TetParams p = make_params(s, gx, gy, r0, img);
    StageScope t(DMR_STAGE_TET_BACKWARD, st);
    k_tet_backward_seq<<<dim3(gx, r1 - r0, s.B), dim3(256), 0, st>>>(p, dL_dcolor, dL_ddepth, dL_dvcolor, dL_dfopacity, host_seq_steps);
    const int ntiles = gx * (r1 - r0) * s.B;
    k_tet_backward<<<dim3((unsigned)std::min(ntiles, 768)), dim3(256), 0, st>>>(p, r1 - r0, dL_dcolor, dL_ddepth, dL_dvcolor, dL_dfopacity);
}

}  // namespace dmr
