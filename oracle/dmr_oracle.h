/*
 * dmr_oracle.h -- C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product path (dmesh_renderer_amd) never does.
 *
 * PARITY UNPINNED (kernel level): the reference (SonSang/dmesh_renderer) ships
 * no tests, golden vectors or CPU path, and its CUDA sources cannot be built in
 * this image, so this restatement is pinned only by (i) line-by-line reading of
 * the reference files cited in dmr_oracle.cpp, (ii) independent brute-force and
 * finite-difference checks in tests/, (iii) the reference Python wrapper run
 * over this oracle (tests/golden/gen_wrapper_fixtures.py).
 */
#ifndef DMR_ORACLE_H
#define DMR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Inputs common to the tri and tet renderers: the arguments of
 * CudaRasterizer::Rasterizer::forward (cuda_rasterizer/rasterizer.h:13-42) and
 * CudaRenderer::Renderer::forward (cuda_renderer/renderer.h:12-46). Matrices
 * are column-major m[4*col+row] (auxiliary.h:71-90). */
typedef struct dmro_scene {
    int B, P, F, T, W, H;
    const float* background;     /* [3] */
    const float* verts;          /* [P,3] */
    const int32_t* faces;        /* [F,3] */
    const float* verts_color;    /* [P,3] */
    const float* faces_opacity;  /* [F] */
    const float* mv_mats;        /* [B,16] */
    const float* proj_mats;      /* [B,16] */
    const float* inv_mv_mats;    /* [B,16] */
    const float* inv_proj_mats;  /* [B,16] */
    const float* verts_depth;    /* [B,P] */
    const float* faces_intense;  /* [B,F] */
    const int32_t* tets;         /* [T,4]  (tet only) */
    const int32_t* face_tets;    /* [F,2]  (tet only, -1 = none) */
    const int32_t* tet_faces;    /* [T,4]  (tet only) */
    int ray_random_seed;         /* tet only; > 0: Philox ray jitter (parity unpinned vs cuRAND) */
    /* tile-row band [row_begin,row_end) rendered by this call; 0,0 = all rows.
     * Not a reference feature: mirrors the multi-GPU shard of the product. */
    int row_begin, row_end;
} dmro_scene;

typedef struct dmro_state dmro_state; /* forward->backward state (the 4 buffers) */

/* tri: returns NULL on error (dmro_last_error()). Writes out_color [B,3,H,W],
 * out_depth [B,1,H,W]. */
dmro_state* dmro_tri_forward(const dmro_scene* s, float* out_color, float* out_depth);
/* grads: dL_dverts [P,3], dL_dvcolor [P,3], dL_dfopacity [F], dL_dvdepth [B,P],
 * dL_dfintense [B,F]; all overwritten. */
int dmro_tri_backward(const dmro_scene* s, const dmro_state* st,
                      const float* dL_dcolor, const float* dL_ddepth,
                      float* dL_dverts, float* dL_dvcolor, float* dL_dfopacity,
                      float* dL_dvdepth, float* dL_dfintense);

/* tet: out_active [B,H,W] (1.0 / 0.0). */
dmro_state* dmro_tet_forward(const dmro_scene* s, float* out_color, float* out_depth,
                             float* out_active);
int dmro_tet_backward(const dmro_scene* s, const dmro_state* st,
                      const float* dL_dcolor, const float* dL_ddepth,
                      float* dL_dvcolor, float* dL_dfopacity);

/* Intermediates, for stage-by-stage parity checks. */
int64_t dmro_num_rendered(const dmro_state* st);
/* name: "ndc"(f32 BP*3) "image"(f32 BP*2) "depths"(f32 BF) "min_depths" "max_depths"
 * "tiles_touched"(u32 BF) "face_offsets"(u32 BF) "keys"(u64 R) "values"(u32 R)
 * "ranges"(u32 B*Nt*2) "ray_o"(f32 BWH*3) "ray_d"(f32 BWH*3) "final_T"(f32 BWH)
 * "final_prev_T"(f32 BWH) "n_contrib"(u32 BWH) "first_face" "first_tet"
 * "last_face" "last_tet"(i32 BWH) "is_active"(u8 BWH).
 * Returns the byte size; copies min(size, cap) bytes into dst when dst != NULL. */
int64_t dmro_get(const dmro_state* st, const char* name, void* dst, int64_t cap);
void dmro_free(dmro_state* st);

/* Helper-level entry points (unit tests of the restated device helpers). */
int dmro_in_tri(float px, float py, float x1, float y1, float x2, float y2, float x3, float y3);
void dmro_clamp_bary_uv(float u, float v, float* uc, float* vc, int* code);
int dmro_ray_tri(const float* o, const float* d, const float* p0, const float* p1,
                 const float* p2, int tet_flavour, float* tuv);
float dmro_ndc2pix(float v, int S);
float dmro_pix2ndc(float v, int S);
void dmro_rect_from_tri(const float* p0, const float* p1, const float* p2, int gx, int gy,
                        uint32_t* rect /* minx,miny,maxx,maxy */);
uint32_t dmro_higher_msb(uint32_t n);

const char* dmro_last_error(void);
/* Noise measurement only (tests/tools/grad_noise.py): non-zero = dmro_tri_backward evaluates the vertex-position gradient of
 * every (pixel, face) pair in double (same formula, float inputs) instead of in the reference's float arithmetic. */
void dmro_set_tri_grad_f64(int on);
int dmro_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
