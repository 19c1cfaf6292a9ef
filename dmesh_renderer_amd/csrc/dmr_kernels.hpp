// dmr_kernels.hpp -- internal launcher declarations (host side of the gfx950 kernels).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dmesh_renderer_amd.h"
#include "dmr_device.hpp"

namespace dmr {

// Packed gradient accumulators of the backward passes (DMR_BUF_WORK):
//   vertex row  [B][P][8] = { dx, dy, dz, dr, dg, db, ddepth, - }   one 32-byte row per (view, vertex)
//   face row    [B][F][2] = { dopacity, dintense }
// so that one (tile, face) flush touches 3 vertex rows + 1 face row (4 memory-side atomic
// requests) instead of 23 scattered dwords.
constexpr int VROW = 8;
constexpr int FROW = 2;
// Records of one list entry (one face in one tile) form a run padded to a multiple of HIT_GROUP; the hit-parallel backward
// takes one group per lane (k_tri_backward_pix / k_tri_backward_hits, dmr_tri.hip).
constexpr int HIT_GROUP = 4;
// Inside a tile's region the records are stored in blocks of HIT_BLOCK = 64 groups, record q of group l of a block at
// block * HIT_BLOCK + q * 64 + l: the 64 lanes of a wave read record q of their groups as one contiguous kilobyte
// (16 cache lines; 64 bytes per lane in memory order would be 64 lines per load instruction, and the hit-parallel kernel
// is bound by exactly that: line accesses in the texture addresser).  Regions are whole blocks.
constexpr int HIT_BLOCK = 64 * HIT_GROUP;
__host__ __device__ inline uint32_t hit_slot_address(uint32_t rel) {  // rel: record index in the tile's face-major order
    return (rel & ~(uint32_t)(HIT_BLOCK - 1)) + (rel & (uint32_t)(HIT_GROUP - 1)) * 64u + ((rel & (uint32_t)(HIT_BLOCK - 1)) / (uint32_t)HIT_GROUP);
}

// Record region of a tile: a face's records form one run padded to a multiple of HIT_GROUP, so a tile with h blended pairs
// (counted by the forward) in a list of `len` entries needs at most h + (HIT_GROUP - 1) * min(len, h) records, rounded up to
// whole blocks.
__host__ __device__ inline uint32_t record_bound(uint32_t h, uint32_t len) {
    if (h == 0u) return 0u;
    const uint32_t b = h + (uint32_t)(HIT_GROUP - 1) * (len < h ? len : h);
    return (b + (uint32_t)(HIT_BLOCK - 1)) & ~(uint32_t)(HIT_BLOCK - 1);
}
// Up to this many tiles one workgroup scans them all (k_scan_tiles, k_scan_hits), and k_tri_backward_pix can find its tile's
// record region on its own (HitRegions below).
constexpr int SCAN_SINGLE_MAX = 8192;

// ---- per-stage HIP-event timing (dmr_api.hip); a no-op unless dmr_profile_enable() set the stage's bit
struct StageScope {
    int stage; hipStream_t st; void* rec;
    StageScope(int stage, hipStream_t st);
    ~StageScope();
};

// ---- binning (dmr_binning.hip)
// also zeroes counters[0, ncounters) (tile_count | tile_hits)
void launch_project_verts(const dmr_scene& s, float4* vproj, float* mats, uint32_t* counters, size_t ncounters,
                          hipStream_t st);
void launch_setup_faces(const dmr_scene& s, bool tet, const float4* vproj, int gx, int gy, int r0, int r1,
                        uint2* face_rect, float* key_depth, float* max_depth, uint32_t* tiles_touched,
                        uint32_t* tile_count, hipStream_t st);
// scan_tmp: scan_tmp_words(ntiles) u32 of scratch whose first SCAN_TMP_BUCKETS words are zero on entry
constexpr int SCAN_TMP_BUCKETS = 128;
size_t scan_tmp_words(int ntiles);
// A size the host waits for travels as ONE 8-byte word in pinned (coherent) host memory: the call's sequence number in the
// top 24 bits, the size (clamped to 40 bits) below.  The host polls the word until the sequence number is its own: no event
// packet in the stream (an event record between two kernels costs the device 2-7 us, profiles/r03/dead_ends.md), no ordering
// question between a value and a flag.
__host__ __device__ inline unsigned long long host_size_word(uint32_t seq, unsigned long long size) {
    return ((unsigned long long)(seq & 0xffffffu) << 40) | (size < (1ull << 40) ? size : (1ull << 40) - 1ull);
}
// host_num_rendered (pinned, may be null): receives host_size_word(host_seq, R).  overflow (pinned, may be null): set to 1 when
// R > capacity (asynchronous calls, which never read R on the host)
void launch_scan_tiles(int ntiles, const uint32_t* tile_count, uint32_t* tile_offset, uint32_t* tile_cursor,
                       int* num_rendered, unsigned long long* host_num_rendered, uint32_t host_seq, uint32_t* tile_order, uint32_t* scan_tmp,
                       uint32_t capacity, uint32_t* overflow, hipStream_t st);
void launch_scatter_faces(const dmr_scene& s, int gx, int gy, const uint2* face_rect, const float* key_depth,
                          const uint32_t* tiles_touched, uint32_t* tile_cursor, uint64_t* keys, uint32_t capacity,
                          unsigned long long* mask_offset_dst, unsigned long long mask_offset, unsigned long long mask_first, hipStream_t st);
// capacity: entries the binning buffer holds; it is below R only while a size guess is being refuted (dmr_api.hip):
// every kernel that walks the tile lists clamps to it, the results are then thrown away and redone.
// (frames above SCAN_SINGLE_MAX tiles: below, the tri forward / the tet first-hit kernel sort their tiles themselves)
void launch_sort_tiles(int ntiles, const uint32_t* tile_offset, const uint32_t* tile_order, uint64_t* keys,
                       uint32_t* face_list, uint32_t capacity, hipStream_t st);

// ---- tri compositing (dmr_tri.hip)
struct TriImageState {
    float* final_T; float* final_prev_T; uint32_t* n_contrib;
    uint32_t* tile_hits;    // covered (pixel, face) pairs below n_contrib per tile, counted by the forward
    uint32_t* tile_bound;   // record_bound(tile_hits, list length), written next to it (16-byte aligned, zero for empty tiles)
    const uint32_t* hit_offset;     // record regions: exclusive scan of the tiles' record bounds (k_scan_hits, backward)
    uint32_t* tile_used;            // records k_tri_backward_pix wrote into a tile's region (padded runs; <= the bound)
    const uint32_t* tile_order;  // all B * gx * gy tiles, longest list first (k_scan_tiles)
    // Coverage masks the forward keeps for the backward: one 4 KB slot (256 pixels x 128 face bits) per 128-entry chunk of a
    // tile's list.  Chunk 0 of the tile at position q of tile_order (busy tiles come first there: q < number of busy tiles
    // <= min(tiles, list entries)) is slot q of the first mask_first_slots() slots; chunk c >= 1 of a list that starts at entry
    // `begin` is slot mask_first_slots() + begin / 128 + c - 1 (the next busy tile's chunks start at or behind
    // (begin + len) / 128 >= begin / 128 + ceil(len / 128) - 1: no two chunks share a slot).  So the masks grow with the list
    // entries, not with the tiles of the frame (up to round 2: one slot per tile, empty or not -- 4 GiB for a frame of 1 M
    // tiles).  They live in the binning buffer behind the lists, whose capacity the backward does not know on the host
    // (speculative sizing): mask_offset[0] = their byte offset, mask_offset[1] = mask_first_slots(), kept on the device.
    const unsigned long long* mask_offset;
};
constexpr int MASK_CHUNK = 128;     // list entries per mask slot = the compositing kernels' chunk
inline size_t mask_first_slots(size_t list_capacity, size_t ntiles) { return list_capacity < ntiles ? list_capacity : ntiles; }
inline size_t mask_slots(size_t list_capacity, size_t ntiles) { return mask_first_slots(list_capacity, ntiles) + list_capacity / MASK_CHUNK + 1; }
// One blended (pixel, face) pair, written face-major per (tile, chunk, pass) by k_tri_backward_pix and
// consumed one per lane by k_tri_backward_hits.
// (A 32-byte record carrying face and vertex ids was tried: kernel 2 did not get faster, kernel 1 got 9 % slower.)
struct alignas(16) HitRecord { uint32_t id; uint32_t pixel; float T; float dL_dalpha; };  // id: word (slot mod HIT_GROUP) of {face, v0, v1, v2}; pixel: tile-local
// keys: the unsorted (depth_bits << 32 | face) list entries of the scatter pass: every tile's workgroup sorts its own list
// (dmr_sort.hpp) into face_list before compositing it, no launch_sort_tiles; null: face_list is sorted already
void launch_tri_forward(const dmr_scene& s, int gx, int gy, int r0, int r1, const float4* vproj,
                        const uint32_t* tile_offset, uint64_t* keys, uint32_t* face_list, uint32_t capacity, TriImageState img,
                        float* out_color, float* out_depth, hipStream_t st);
// host_*: pinned host memory the kernel also writes its total to (no separate device->host copy)
// hit_offset: every tile's region of the record buffer, sized by the bound h + (HIT_GROUP - 1) * min(list length, h) of
// its h blended pairs (tile_hits); tile_used is cleared (the per-pixel kernel fills it)
void launch_scan_hits(int ntiles, const uint32_t* tile_hits, const uint32_t* tile_offset, uint32_t* hit_offset, uint32_t* tile_used,
                      unsigned long long* hit_total, unsigned long long* host_hit_total, uint32_t host_seq, uint32_t* scan_tmp, uint32_t capacity,
                      uint32_t* overflow, hipStream_t st);
// Without launch_scan_hits (B * tiles <= SCAN_SINGLE_MAX): every workgroup of k_tri_backward_pix sums the record bounds of
// the tiles before its own (tile_bound: eight 16-byte loads per thread, all in flight at once) and publishes hit_offset[tile] /
// tile_used[tile] for the hit-parallel kernel; the last tile's workgroup also leaves the total (device, pinned host) and
// raises the overflow word when it exceeds the capacity.  One launch and ~8 us of single-workgroup latency less per step.
struct HitRegions {
    uint32_t* hit_offset;                 // null: the regions come from launch_scan_hits
    unsigned long long* hit_total; unsigned long long* host_hit_total; uint32_t* overflow;  // the last two may be null
    uint32_t host_seq;                    // host_hit_total receives host_size_word(host_seq, total)
};
// also zeroes work[0, work_floats) (the packed accumulators)
void launch_tri_backward_pix(const dmr_scene& s, int gx, int gy, int r0, int r1, const float4* vproj,
                             const uint32_t* tile_offset, const uint32_t* face_list, TriImageState img,
                             const float* dL_dcolor, const float* dL_ddepth, float4* pixrec, HitRecord* hits,
                             uint32_t capacity, float* work, size_t work_floats, HitRegions regions, hipStream_t st);
// one workgroup per tile (longest list first) over the tile's img.tile_used records
void launch_tri_backward_hits(const dmr_scene& s, int gx, int gy, const float4* vproj, const uint32_t* face_list, TriImageState img,
                              const float4* pixrec, const HitRecord* hits, uint32_t capacity, float* vrow, float* frow,
                              hipStream_t st);
void launch_tri_unpack(const dmr_scene& s, const float* vrow, const float* frow, float* dL_dverts,
                       float* dL_dvcolor, float* dL_dfopacity, float* dL_dvdepth, float* dL_dfintense,
                       hipStream_t st);

// ---- tet ray march (dmr_tet.hip)
// The MARCH SEQUENCE the forward leaves for the backward: the faces every pixel crossed, in order, so that the backward
// consumes them from the back instead of re-discovering them (the reference re-marches: three ray-triangle tests and four
// outward normals per step, cuda_renderer/backward.cu:372-476).  Static layout, no allocation on the device: wave w of the
// forward (tile * 4 + quadrant; its 64 pixels are at the same step in the same loop iteration) owns cap_steps / 4 rows of
// 64 x 16 bytes, step s of lane l is dword (s & 3) of the 16-byte word l of row s / 4 -- a lane stores four steps as one
// 16-byte word, a wave's store / load of a row is one contiguous kilobyte.  The region lies in the binning buffer behind
// the lists; cap_steps is the longest march of the previous call with the same view configuration + 25 % (dmr_api.hip; a
// dynamic allocation of 4-KB chunks from an atomic counter was measured first: its bookkeeping inside the march loop cost
// the forward 100 us at C3, more than the backward gained).  A sequence that did not fit (max_steps > cap_steps; cap_steps
// == 0: no estimate yet) is ignored: the backward then re-marches like the reference.
// Bit 31 of an entry: the reference's reverse march would stop behind this face ("error case 3": a second face of the
// tet the ray leaves through this one is hit from outside as well, backward.cu:456-460) -- the forward has those three
// tests in its hands anyway, so the backward reproduces the reference's decision without repeating them.
struct TetSeq {
    uint32_t max_steps;         // longest march of this forward (atomicMax, one per wave): complete iff <= cap_steps
    uint32_t cap_steps;         // steps per pixel the region has room for (a multiple of 4; 0: no region)
    unsigned long long offset;  // bytes from the binning buffer's start
};
inline size_t tet_seq_bytes(size_t ntiles, size_t cap_steps) { return ntiles * 4 * (cap_steps / 4) * 1024; }
struct TetImageState {
    float* final_log_T; float* final_prev_log_T; uint32_t* n_contrib;
    int32_t* first_face; int32_t* first_tet; int32_t* last_face; int32_t* last_tet; uint8_t* is_active;
    void* facerec; void* colrec; void* tetrec;  // packed march records (dmr_tet.hip), in the face buffer
    int* seed;                                  // ray_random_seed of the forward (image buffer)
    TetSeq* seq;                                // the march sequence's descriptor (image buffer)
    char* binning;                              // start of the binning buffer (the sequence region lies at seq->offset)
};
size_t tet_facerec_bytes();   // per face
size_t tet_colrec_bytes();    // per face
size_t tet_tetrec_bytes();    // per tet
// builds the packed march records from the scene (every forward call); also resets the march sequence's descriptor:
// room for seq_steps steps per pixel at byte seq_offset of the binning buffer
void launch_tet_prep(const dmr_scene& s, TetImageState img, uint32_t seq_steps, unsigned long long seq_offset, hipStream_t st);
// keys: as launch_tri_forward (non-null: every tile's workgroup sorts its list itself)
void launch_tet_first_intersect(const dmr_scene& s, int gx, int gy, int r0, int r1, const float* key_depth,
                                const float* max_depth, const uint32_t* tile_offset, uint64_t* keys, uint32_t* face_list,
                                uint32_t capacity, TetImageState img, hipStream_t st);
void launch_tet_forward(const dmr_scene& s, int gx, int gy, int r0, int r1, TetImageState img,
                        float* out_color, float* out_depth, float* out_active, hipStream_t st);
void launch_tet_zero_grads(float* dL_dvcolor, int64_t n_vcolor, float* dL_dfopacity, int64_t n_fopacity, hipStream_t st);
// Two launches, of which the device runs one: k_tet_backward_seq when the forward's march sequence is complete
// (seq->max_steps <= seq->cap_steps != 0), else the re-marching k_tet_backward; the other one returns at once.  No host read.
// host_seq_steps (pinned, may be null): receives seq->max_steps, the next forward's capacity estimate.
void launch_tet_backward(const dmr_scene& s, int gx, int gy, int r0, int r1, TetImageState img,
                         const float* dL_dcolor, const float* dL_ddepth, float* dL_dvcolor, float* dL_dfopacity,
                         uint32_t* host_seq_steps, hipStream_t st);

}  // namespace dmr
