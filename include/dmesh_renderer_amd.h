/*
 * dmesh_renderer_amd.h -- C ABI of libdmesh_renderer_hip.so (MI355X / gfx950).
 *
 * This is the drop-in boundary for the hot path of SonSang/dmesh_renderer: the four
 * entry points below replace what the reference's pybind module `_C` (ext.cpp:6-11)
 * reaches through render.cu:
 *
 *   dmr_tri_forward   <- CudaRasterizer::Rasterizer::forward  (cuda_rasterizer/rasterizer.h:13-42,
 *                        called from RasterizeTrianglesCUDA, render.cu:107-129)
 *   dmr_tri_backward  <- CudaRasterizer::Rasterizer::backward (rasterizer.h:44-67, render.cu:175-204)
 *   dmr_tet_forward   <- CudaRenderer::Renderer::forward      (cuda_renderer/renderer.h:12-46, render.cu:303-334)
 *   dmr_tet_backward  <- CudaRenderer::Renderer::backward     (renderer.h:48-75, render.cu:378-409)
 *
 * Plain pointers and sizes only: every pointer is a DEVICE pointer unless stated,
 * tensors are dense row-major fp32 / int32 exactly as render.cu hands them down
 * (`.contiguous().data<T>()`), matrices are [B,16] column-major m[4*col+row]
 * (auxiliary.h:71-90) unless dmr_scene.mats_transposed says otherwise.  `stream` is a hipStream_t (NULL = default stream).  All work
 * is enqueued on that stream; the only host synchronisation is the wait for the 4-byte num_rendered in the forward
 * calls and for the 8-byte record count in the tri backward (reference: rasterizer_impl.cu:287-292), and not even that
 * with DMR_FLAG_ASYNC or under stream capture ("Sizes only the device knows" below).
 *
 * Return value: 0 on success, non-zero on error; dmr_last_error() then returns a
 * message for the calling thread (the glue raises RuntimeError with it, as the
 * reference's CHECK_CUDA -> std::runtime_error does, auxiliary.h:425-432).
 */
#ifndef DMESH_RENDERER_AMD_H
#define DMESH_RENDERER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DMR_ABI_VERSION 4

/* Scratch buffers.  The first four are the reference's pointBuffer / faceBuffer /
 * binningBuffer / imageBuffer (rasterizer.h:14-17): opaque byte buffers that the
 * caller owns, returns from forward and passes back to backward.  DMR_BUF_WORK is a
 * transient workspace of the backward calls (packed gradient accumulators). */
enum { DMR_BUF_POINT = 0, DMR_BUF_FACE = 1, DMR_BUF_BINNING = 2, DMR_BUF_IMAGE = 3, DMR_BUF_WORK = 4 };

/* Footprints (bytes; B views, P verts, F faces, T tets, Nt = B * ceil(W/16) * ceil(H/16) tiles, R list entries; every
 * sub-array rounded up to 256): point 16 BP; face 16 BF (tet: 20 BF + 128 F + 224 T); image ~76 B + 40 Nt + 12 BWH (tet: 29 BWH);
 * binning 12 R', R' = R or, with a size estimate, 1.25 R_prev + 4096 -- plus, tri only, the coverage masks the forward
 * leaves for the backward: 4096 (R'/128 + min(Nt, R') + 1), i.e. 32 B per list entry and 4 KB per tile that can be busy (a
 * tile's first chunk has the slot of the tile's rank among the busy tiles, so a sparse frame of 1 M tiles pays for its list
 * entries, not 4 GiB for its tiles; the reference's binning buffer scales with R only) -- plus, tet only, the forward's march sequence for
 * the backward: 4 bytes per tile pixel (256 Nt of them) and step of capacity, capacity = the longest march of the previous
 * call with the same view configuration * 1.25 + 4 steps (0 in the first such call), the whole capped at 16 GiB;
 * work (tri backward) 32 BP + 8 BF + 8192 Nt + 16 per hit record. */

/* C equivalent of the reference's four std::function<char*(size_t)> allocators
 * (rasterizer.h:14-17, render.cu:18-24): must return a device pointer to at least
 * `nbytes` bytes (256-byte aligned), or NULL on failure.  Called from the calling thread,
 * normally once per buffer per call; a second request for the same buffer (larger size) replaces the
 * first one -- it happens when a size guess taken from the previous call turned out too small. */
typedef void* (*dmr_alloc_fn)(void* ctx, int which, size_t nbytes);

typedef struct dmr_scene {
    int32_t B, P, F, T, W, H;    /* views, verts, faces, tets (0 for tri), image size */
    const float* background;     /* [3] */
    const float* verts;          /* [P,3] */
    const int32_t* faces;        /* [F,3] */
    const float* verts_color;    /* [P,3] */
    const float* faces_opacity;  /* [F] */
    const float* mv_mats;        /* [B,16] column-major */
    const float* proj_mats;      /* [B,16] */
    const float* inv_mv_mats;    /* [B,16] */
    const float* inv_proj_mats;  /* [B,16] */
    const float* verts_depth;    /* [B,P] */
    const float* faces_intense;  /* [B,F] */
    const int32_t* tets;         /* [T,4]  tet renderer only */
    const int32_t* face_tets;    /* [F,2]  tet renderer only, -1 = no tet */
    const int32_t* tet_faces;    /* [T,4]  tet renderer only */
    int32_t ray_random_seed;     /* tet renderer only; <= 0: rays through pixel centres; > 0: jittered rays, pixel - 0.5 + 0.5 u
                                  * (cuda_renderer/forward.cu:120-123) with u from Philox-4x32-10(key = seed, counter = pixel):
                                  * same distribution as the reference, not its cuRAND XORWOW bits (parity unpinned) */
    /* Tile-row band [row_begin, row_end) this call renders (multi-GPU shard by tile
     * rows); 0,0 means all rows.  Pixels outside the band are left untouched. */
    int32_t row_begin, row_end;
    /* Bit i set (0 mv, 1 proj, 2 inv_mv, 3 inv_proj): matrix i is handed over with its 4x4 blocks transposed,
     * i.e. element k of the [B,16] contract above lives at 16*b + 4*(k & 3) + (k >> 2).  That is the storage
     * behind the `.transpose(1, 2)` views the reference wrapper passes (dmesh_renderer/__init__.py:219-220)
     * and behind th.inverse of such a view, so the glue need not launch the four `.contiguous()` copies of
     * render.cu:117-120.  The forward stores the matrices in contract layout in the image buffer; the
     * backward reads them from there (its matrix arguments are not dereferenced). */
    int32_t mats_transposed;
    /* DMR_FLAG_* bits.  DMR_FLAG_ASYNC: the call never waits for the device (see "Sizes only the device knows"). */
    int32_t flags;
} dmr_scene;

/* Sizes only the device knows.  R (the binning buffer's entries, `num_rendered`) and the number of blended (pixel,
 * face) pairs (the tri backward's record buffer) are results of kernels.  The reference stalls on a device->host copy of
 * R before it can go on (rasterizer_impl.cu:287-299).  Here a call sizes both buffers from the previous call with the
 * same view configuration (+25 %), enqueues everything, and
 *   - by default waits for the size to arrive in pinned host memory (the kernel that computes it stores it there; the host
 *     polls the word: no event, no stream synchronisation, the GPU keeps running), returns the exact R and redoes the
 *     affected stages if the estimate was too small;
 *   - with DMR_FLAG_ASYNC, or when `stream` is being captured into a HIP graph (hipStreamIsCapturing), does not wait at
 *     all: *num_rendered receives the CAPACITY it used (an upper bound that the backward accepts in R's place), every
 *     kernel clamps to it, and a scene that outgrew it sets a sticky per-device flag that dmr_overflowed() reports --
 *     the results of such a call are incomplete (tiles beyond the capacity render as empty) and the caller repeats the
 *     step with a default (waiting) call, which refreshes the estimate.  Needs one earlier default call with the same
 *     view configuration (the warm-up before a capture), else it fails. */
#define DMR_FLAG_ASYNC 1
/* 1 if an asynchronous / captured call on `device` (-1: the current one) overflowed its capacity since the flag was
 * last reset; call it after the stream (or the graph launch) has completed.  reset != 0 clears the flag. */
int dmr_overflowed(int device, int reset);
/* How often a default call had to enqueue stages a second time because its size estimate was too small (process-wide,
 * monotonic): 0 in a steady training loop; a figure that keeps growing says the estimates do not fit the workload (they are
 * kept per view configuration and power-of-two bucket of B * F). */
uint64_t dmr_redo_count(void);

/* out_color [B,3,H,W], out_depth [B,1,H,W]: every pixel of the rendered tile rows is written; the caller
 * zero-initialises them (render.cu:88-89) when a band leaves rows untouched or when P == 0 / F == 0
 * (nothing is launched, render.cu:105).  *num_rendered receives R = sum of tiles touched. */
int dmr_tri_forward(const dmr_scene* scene, float* out_color, float* out_depth,
                    dmr_alloc_fn alloc, void* alloc_ctx, void* stream, int* num_rendered);

/* Gradient outputs are fully overwritten: dL_dverts [P,3], dL_dvcolor [P,3],
 * dL_dfopacity [F], dL_dvdepth [B,P], dL_dfintense [B,F]. */
int dmr_tri_backward(const dmr_scene* scene, const float* dL_dcolor, const float* dL_ddepth,
                     int num_rendered, const void* point_buf, const void* face_buf,
                     const void* binning_buf, const void* image_buf,
                     float* dL_dverts, float* dL_dvcolor, float* dL_dfopacity,
                     float* dL_dvdepth, float* dL_dfintense,
                     dmr_alloc_fn alloc, void* alloc_ctx, void* stream);

/* out_active [B,H,W]: 1.0 where the ray marched to a valid end, else 0.0.  Like the tri forward, every pixel of the
 * rendered tile rows of all three outputs is written (background / 1 / 0 where the march fails). */
int dmr_tet_forward(const dmr_scene* scene, float* out_color, float* out_depth, float* out_active,
                    dmr_alloc_fn alloc, void* alloc_ctx, void* stream, int* num_rendered);

/* dL_dvcolor [P,3], dL_dfopacity [F], fully overwritten. */
int dmr_tet_backward(const dmr_scene* scene, const float* dL_dcolor, const float* dL_ddepth,
                     const void* point_buf, const void* face_buf,
                     const void* binning_buf, const void* image_buf,
                     float* dL_dvcolor, float* dL_dfopacity,
                     dmr_alloc_fn alloc, void* alloc_ctx, void* stream);

/* Caller-side step of the path: the reference wrapper computes th.inverse of the (transposed) model-view and
 * projection matrices on every forward (dmesh_renderer/__init__.py:62-63,298-299) -- on a GPU that is two batched
 * LU factorisations, ~0.12 ms of small kernels.  This inverts `count` 4x4 matrices with one kernel (adjugate /
 * determinant in double precision, rounded to fp32 once): out[16*m + 4*i + j] = inverse(A_m)[i][j] with
 * A_m[i][j] = in[16*m + 4*i + j], or in[16*m + i + 4*j] when `transposed` (the storage behind a .transpose(1, 2)
 * view).  A singular matrix gives inf/nan entries, as a division by a zero determinant does. */
int dmr_invert_mats(const float* in, int count, int transposed, float* out, void* stream);

/* Parity/debug export of forward intermediates held in the scratch buffers.
 * name: "image" (f32 [B*P,2]) "ndc_z" (f32 [B*P]) "key_depth" (f32 [B*F]) "max_depth" (tet, f32 [B*F])
 * "tiles_touched" (u32 [B*F]) "ranges" (u32 [B*Nt,2]) "face_list" (u32 [R]) "final_T" "final_prev_T"
 * (f32 [B*W*H]) "n_contrib" (u32 [B*W*H]) "tile_hits" (tri, u32 [B*Nt]: blended (pixel, face) pairs per tile) "first_face" "first_tet" "last_face" "last_tet" (i32, tet)
 * "is_active" (u8, tet) "tet_seq" (tet, u32 [2]: the longest march in steps, the steps per pixel the forward's march sequence had room for).  dst is a DEVICE pointer with room for `cap` bytes.  Returns the byte size
 * of the item (copying min(size, cap) when dst != NULL), or -1. */
int64_t dmr_export(const dmr_scene* scene, int is_tet, int num_rendered, const char* name,
                   const void* point_buf, const void* face_buf, const void* binning_buf,
                   const void* image_buf, void* dst, int64_t cap, void* stream);

/* Per-stage device timing with HIP events recorded on the caller's stream (bench.py's roofline
 * leg).  `mask` has bit i set to time stage i; 0 disables (the default: no events recorded). */
enum {
    DMR_STAGE_PROJECT = 0, DMR_STAGE_SETUP_FACES = 1, DMR_STAGE_SCAN = 2, DMR_STAGE_SCATTER = 3,
    DMR_STAGE_SORT = 4, DMR_STAGE_TRI_FORWARD = 5, DMR_STAGE_TRI_BACKWARD = 6, DMR_STAGE_TRI_UNPACK = 7,
    DMR_STAGE_TET_FIRST = 8, DMR_STAGE_TET_FORWARD = 9, DMR_STAGE_TET_BACKWARD = 10,
    DMR_STAGE_TRI_BACKWARD_HITS = 11, DMR_NUM_STAGES = 12
};
void dmr_profile_enable(uint32_t mask);
/* Waits for the recorded events, ADDS each stage's elapsed milliseconds / launch count into
 * ms[DMR_NUM_STAGES] / launches[DMR_NUM_STAGES] and clears the records.  Returns 0 on success. */
int dmr_profile_collect(double* ms, int64_t* launches);
const char* dmr_stage_name(int stage);

const char* dmr_last_error(void);
int dmr_abi_version(void);
/* Name of the code object architecture the library was built for ("gfx950"). */
const char* dmr_build_arch(void);

#ifdef __cplusplus
}
#endif
#endif /* DMESH_RENDERER_AMD_H */
