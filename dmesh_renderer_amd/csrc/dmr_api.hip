// dmr_api.hip -- C ABI (include/dmesh_renderer_amd.h) and host orchestration.
//
// Replaces the host side of Rasterizer::forward/backward (cuda_rasterizer/rasterizer_impl.cu:175-467)
// and Renderer::forward/backward (cuda_renderer/renderer_impl.cu:193-498): scratch carving,
// stage sequencing, the single device->host read of num_rendered.  Unlike the reference
// (a cudaDeviceSynchronize after every stage on the legacy default stream, Q14) everything is
// enqueued on the caller's stream and the only host wait is that 4-byte read, which sizes the
// binning buffer.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "dmr_kernels.hpp"

namespace {

thread_local std::string g_err;

int fail(const std::string& m) { g_err = m; return 1; }

#define DMR_HIP(expr)                                                                           \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(e_));   \
    } while (0)

constexpr size_t ALIGN = 256;
inline size_t up(size_t n) { return (n + ALIGN - 1) & ~(ALIGN - 1); }

// Bump carving of a scratch buffer (the reference's obtain(), rasterizer_impl.h:10-16).
struct Carver {
    char* base; size_t off;
    explicit Carver(void* b) : base(reinterpret_cast<char*>(b)), off(0) {}
    template <class T> T* take(size_t count) {
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += up(count * sizeof(T));
        return p;
    }
};

struct PointState { float4* vproj; };
struct FaceState { uint2* rect; float* key_depth; float* max_depth; uint32_t* tiles_touched; void* facerec; void* colrec; void* tetrec; };
struct ImageState {
    uint32_t* tile_count; uint32_t* tile_offset; uint32_t* tile_cursor; int* num_rendered;
    float* final_T; float* final_prev_T; uint32_t* n_contrib;
    uint32_t* tile_hits; uint32_t* tile_bound; uint32_t* scan_tmp; uint32_t* hit_offset; uint32_t* tile_used; unsigned long long* hit_total; uint32_t* tile_order;
    int32_t* first_face; int32_t* first_tet; int32_t* last_face; int32_t* last_tet; uint8_t* is_active;
    float* mats;  // [mv | proj | inv_mv | inv_proj], [B,16] each, contract layout (written by k_project_verts)
    int* seed;    // tet: ray_random_seed of the forward (the backward recomputes the same jittered rays)
    unsigned long long* mask_offset;  // tri: {byte offset of the coverage masks inside the binning buffer, their first-chunk slots} (written on the device)
    dmr::TetSeq* seq;  // tet: the march sequence's descriptor (dmr_kernels.hpp)
};
struct BinningState {
    uint64_t* keys; uint32_t* face_list; uint32_t capacity; unsigned long long mask_offset, mask_first;
    char* base; unsigned long long seq_offset; uint32_t seq_steps;  // tet: the march sequence's region
};

size_t carve_point(void* b, size_t BP, PointState& s) { Carver c(b); s.vproj = c.take<float4>(BP); return c.off; }
size_t carve_face(void* b, size_t BF, size_t F, size_t T, bool tet, FaceState& s) {
    Carver c(b);
    s.rect = c.take<uint2>(BF); s.key_depth = c.take<float>(BF);
    s.max_depth = tet ? c.take<float>(BF) : nullptr;
    s.tiles_touched = c.take<uint32_t>(BF);
    s.facerec = s.colrec = s.tetrec = nullptr;
    if (tet) {  // packed march records (view independent)
        s.facerec = c.take<char>(F * dmr::tet_facerec_bytes());
        s.colrec = c.take<char>(F * dmr::tet_colrec_bytes());
        s.tetrec = c.take<char>(T * dmr::tet_tetrec_bytes());
    }
    return c.off;
}
size_t carve_image(void* b, size_t B, size_t ntiles, size_t npix, bool tet, ImageState& s) {
    Carver c(b);
    s.mats = c.take<float>(64 * B);
    s.seed = c.take<int>(1);
    s.mask_offset = c.take<unsigned long long>(2);
    // counters, zeroed by k_project_verts at the start of a forward: [tile_count | tile_hits | tile_bound | scan_tmp's buckets]
    s.tile_count = c.take<uint32_t>(ntiles); s.tile_hits = c.take<uint32_t>(ntiles); s.tile_bound = c.take<uint32_t>(ntiles);
    s.scan_tmp = c.take<uint32_t>(dmr::scan_tmp_words((int)ntiles));
    s.hit_offset = c.take<uint32_t>(ntiles + 1); s.tile_used = c.take<uint32_t>(ntiles); s.hit_total = c.take<unsigned long long>(1);
    s.tile_offset = c.take<uint32_t>(ntiles + 1);
    s.tile_cursor = c.take<uint32_t>(ntiles); s.num_rendered = c.take<int>(1);
    s.tile_order = c.take<uint32_t>(ntiles);
    s.final_T = c.take<float>(npix); s.final_prev_T = c.take<float>(npix); s.n_contrib = c.take<uint32_t>(npix);
    if (tet) {
        s.first_face = c.take<int32_t>(npix); s.first_tet = c.take<int32_t>(npix);
        s.last_face = c.take<int32_t>(npix); s.last_tet = c.take<int32_t>(npix);
        s.is_active = c.take<uint8_t>(npix);
        s.seq = c.take<dmr::TetSeq>(1);
    } else {
        s.first_face = s.first_tet = s.last_face = s.last_tet = nullptr; s.is_active = nullptr;
        s.seq = nullptr;
    }
    return c.off;
}
// face_list first: the backward re-derives it from the buffer start, whatever capacity the forward
// gave the buffer (speculative sizing allocates more than R entries)
// ... then, tri only, the coverage masks the forward leaves for the backward (dmr_kernels.hpp): their place depends on the
// capacity, so the kernels take it from the device (ImageState::mask_offset, written by the scatter pass).  mask_tiles = 0:
// no masks (tet; the backward, which never carves beyond the lists on the host).
// ... or, tet only, the forward's march sequence (seq_steps steps per pixel of seq_tiles tiles, dmr_kernels.hpp; its place is
// kept on the device too).
size_t carve_binning(void* b, size_t R, size_t mask_tiles, size_t seq_tiles, size_t seq_steps, BinningState& s) {
    Carver c(b);
    s.face_list = c.take<uint32_t>(R); s.keys = c.take<uint64_t>(R);
    s.capacity = (uint32_t)std::min<size_t>(R, 0xffffffffu);
    s.mask_offset = c.off;
    s.mask_first = dmr::mask_first_slots(R, mask_tiles);
    if (mask_tiles && R > 0) c.take<uint4>(dmr::mask_slots(R, mask_tiles) * 256);
    s.base = reinterpret_cast<char*>(b); s.seq_offset = c.off; s.seq_steps = (uint32_t)seq_steps;
    if (seq_steps) c.take<char>(dmr::tet_seq_bytes(seq_tiles, seq_steps));
    return c.off;
}

struct Dims { int gx, gy, r0, r1, ntiles; size_t BP, BF, npix; };

// The scene as every kernel behind k_project_verts sees it: matrices in contract layout, read from the image buffer.
dmr_scene canonical(const dmr_scene* s, const float* mats) {
    dmr_scene c = *s;
    const size_t n = 16 * (size_t)s->B;
    c.mv_mats = mats; c.proj_mats = mats + n; c.inv_mv_mats = mats + 2 * n; c.inv_proj_mats = mats + 3 * n;
    c.mats_transposed = 0;
    return c;
}

int check_scene(const dmr_scene* s, bool tet, Dims& d) {
    if (!s) return fail("null scene");
    if (s->B <= 0 || s->W <= 0 || s->H <= 0 || s->P < 0 || s->F < 0 || s->T < 0) return fail("bad dimensions");
    if ((int64_t)s->B * s->P > INT32_MAX || (int64_t)s->B * s->F > INT32_MAX || (int64_t)s->B * s->W * s->H > INT32_MAX)
        return fail("problem too large for 32-bit indexing");
    d.gx = (s->W + dmr::TILE - 1) / dmr::TILE;
    d.gy = (s->H + dmr::TILE - 1) / dmr::TILE;
    if (d.gx > 65535 || d.gy > 65535) return fail("image too large (tile coordinates must fit 16 bits)");
    d.r0 = s->row_begin; d.r1 = s->row_end;
    if (d.r0 == 0 && d.r1 == 0) d.r1 = d.gy;
    d.r0 = std::max(0, std::min(d.gy, d.r0));
    d.r1 = std::max(d.r0, std::min(d.gy, d.r1));
    d.ntiles = s->B * d.gx * d.gy;
    d.BP = (size_t)s->B * s->P; d.BF = (size_t)s->B * s->F; d.npix = (size_t)s->B * s->W * s->H;
    if (tet && s->F > 0 && s->P > 0 && (!s->tets || !s->face_tets || !s->tet_faces)) return fail("tet topology missing");
    if (tet && s->F >= (1 << 28)) return fail("too many faces for the tet renderer (face ids must fit 28 bits)");
    return 0;
}

// Pinned (coherent, device-visible) landing pad the kernels write the sizes to -- forward: the 8-byte word at byte 0,
// backward: the one at byte 8, each host_size_word(sequence number, size) (dmr_kernels.hpp) -- one per host thread and
// device.  The host POLLS the word until it carries the call's sequence number: an event recorded between two kernels costs the
// device 2-7 us (profiles/r03/dead_ends.md), a default call had two of them per step.
struct SizeRead {
    void* slot = nullptr; uint32_t seq = 0;
    volatile unsigned long long* word(int which) const { return reinterpret_cast<volatile unsigned long long*>(slot) + which; }
    uint32_t next_seq() { seq = (seq + 1u) & 0xffffffu; if (seq == 0u) seq = 1u; return seq; }
};
SizeRead* size_read() {
    thread_local std::map<int, SizeRead> per_device;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    SizeRead& sr = per_device[dev];
    if (!sr.slot) {
        if (hipHostMalloc(&sr.slot, 64, hipHostMallocCoherent | hipHostMallocPortable) != hipSuccess) { sr.slot = nullptr; return nullptr; }
        memset(sr.slot, 0, 64);
    }
    return &sr;
}
// Waits until *w carries `seq`; returns the size in it.  Spins on the pinned word; if it does not arrive within ~50 ms the
// stream is synchronised instead (a kernel error then surfaces as a HIP error, and a completed stream has written the word).
int wait_size(volatile unsigned long long* w, uint32_t seq, hipStream_t st, unsigned long long* size) {
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0;; spins++) {
        const unsigned long long v = *w;
        if ((uint32_t)(v >> 40) == (seq & 0xffffffu)) { std::atomic_thread_fence(std::memory_order_acquire); *size = v & ((1ull << 40) - 1ull); return 0; }
        if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) break;
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
    }
    DMR_HIP(hipStreamSynchronize(st));
    const unsigned long long v = *w;
    if ((uint32_t)(v >> 40) != (seq & 0xffffffu)) return fail("the size word was not written (internal error)");
    *size = v & ((1ull << 40) - 1ull);
    return 0;
}

// Sticky per-device overflow word of the asynchronous / captured calls (pinned host memory the scan kernels store 1
// into when a scene outgrew the capacity a call was enqueued with); read by dmr_overflowed().  Created by the first
// default (waiting) call on a device -- never inside a stream capture, where hipHostMalloc is not allowed.
std::mutex g_overflow_mu;
std::map<int, uint32_t*> g_overflow;
uint32_t* overflow_word(int dev, bool create) {
    std::lock_guard<std::mutex> lk(g_overflow_mu);
    auto it = g_overflow.find(dev);
    if (it != g_overflow.end()) return it->second;
    if (!create) return nullptr;
    void* p = nullptr;
    if (hipHostMalloc(&p, 64, hipHostMallocCoherent | hipHostMallocPortable) != hipSuccess) return nullptr;
    *reinterpret_cast<volatile uint32_t*>(p) = 0u;
    return g_overflow[dev] = reinterpret_cast<uint32_t*>(p);
}

bool stream_is_capturing(hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return false; }
    return cs != hipStreamCaptureStatusNone;
}

// Speculative sizing.  The sizes of the binning buffer (R) and of the hit-record buffer are
// only known on the device.  The reference stalls the pipeline on a device->host read before it can continue
// (rasterizer_impl.cu:287-299).  Here the previous call with the same view configuration provides a capacity guess
// (its sizes per face / per list entry, +25 %): the buffer is allocated and ALL remaining kernels are enqueued
// before the host waits -- and it waits for the word the size-writing kernel stores into pinned host memory (polled: no
// event packet, no stream synchronisation), so the GPU keeps running.  Kernels clamp their writes to the capacity; if
// the exact size turns out larger the affected stages are simply enqueued again with an exact buffer (first call,
// or a scene that grew by more than 25 %).
// The guess is keyed by the view configuration only and kept as RATIOS (list entries and hit records per (view, face)
// pair): a mesh whose face count changes every iteration (DMesh re-tetrahedralises) still gets a guess, and the cache
// stays a handful of entries.
// Asynchronous calls (DMR_FLAG_ASYNC, or a stream that is being captured into a HIP graph) use the same estimate and
// never wait: the size is not read back at all, the scan kernel compares it with the capacity on the device and sets
// the sticky overflow word (dmr_overflowed).  That is what makes forward + backward capturable as one graph.
struct SizeKey {
    int v[7];
    bool operator<(const SizeKey& o) const { return memcmp(v, o.v, sizeof(v)) < 0; }
};
// seq_steps: tet only -- pinned word the backward kernels leave the forward's longest march in (steps); read without any
// wait by the next forward as its capacity estimate (a stale value only costs that call the re-marching backward)
struct SizeGuess { double rendered_per_face = 0.0, hits_per_face = 0.0; uint32_t* seq_steps = nullptr; };
std::mutex g_size_mu;
std::map<SizeKey, SizeGuess> g_size_cache;
// ... and by the magnitude of the mesh (floor(log2(B * F))): two renderers with the same view configuration and meshes of very
// different size -- say a coarse and a fine level of one pipeline, alternating -- would otherwise trade one entry back and forth
// between over-allocation and the redo path ("what's weak" 8).  A mesh that crosses a power of two finds the
// neighbouring bucket's estimate (lookup tries b, b - 1, b + 1).
std::atomic<uint64_t> g_redo_count{0};  // stages enqueued a second time because an estimate was too small (dmr_redo_count)
int size_bucket(size_t BF) { int b = 0; while ((BF >> (b + 1)) != 0) b++; return b; }
SizeKey size_key(const dmr_scene* s, bool tet, const Dims& d) {
    return SizeKey{{s->B, s->W, s->H, d.r0, d.r1, tet ? 1 : 0, size_bucket(d.BF)}};
}
// the estimate `field` of this key's bucket or, failing that, of a neighbouring bucket (0: none); g_size_mu held
double lookup_estimate(const SizeKey& key, double SizeGuess::* field) {
    for (int db : {0, -1, 1}) {
        SizeKey k = key; k.v[6] += db;
        auto it = g_size_cache.find(k);
        if (it != g_size_cache.end() && it->second.*field > 0.0) return it->second.*field;
    }
    return 0.0;
}
uint64_t padded(uint64_t n) { return n + n / 4 + 4096; }

// Stages shared by both renderers up to the sorted per-tile lists, then `render` (the renderer's own
// kernels, which only need the binning state).  See "Speculative sizing" above for the control flow.
int run_forward(const dmr_scene* s, bool tet, const Dims& d, dmr_alloc_fn alloc, void* ctx, hipStream_t st,
                PointState& ps, FaceState& fs, ImageState& is, int* num_rendered,
                const std::function<void(const BinningState&)>& render) {
    PointState tp; FaceState tf; ImageState ti;
    void* pb = alloc(ctx, DMR_BUF_POINT, carve_point(nullptr, d.BP, tp));
    void* fb = alloc(ctx, DMR_BUF_FACE, carve_face(nullptr, d.BF, (size_t)s->F, (size_t)s->T, tet, tf));
    void* ib = alloc(ctx, DMR_BUF_IMAGE, carve_image(nullptr, (size_t)s->B, (size_t)d.ntiles, d.npix, tet, ti));
    if (!pb || !fb || !ib) return fail("scratch allocation failed");
    carve_point(pb, d.BP, ps);
    carve_face(fb, d.BF, (size_t)s->F, (size_t)s->T, tet, fs);
    carve_image(ib, (size_t)s->B, (size_t)d.ntiles, d.npix, tet, is);
    const bool async = (s->flags & DMR_FLAG_ASYNC) != 0 || stream_is_capturing(st);
    int dev = 0;
    DMR_HIP(hipGetDevice(&dev));
    uint32_t* overflow = overflow_word(dev, !async);

    // host_R: pinned slot the scan also stores R into (null: not needed).  Not on a redo pass: R is known by then, and
    // a late store could land in the slot after this call has returned and a later call (another stream) reuses it.
    auto front = [&](unsigned long long* host_R, uint32_t host_seq, uint64_t capacity, uint32_t* ovf) -> int {
        // (tile_count | tile_hits | tile_bound are contiguous: zeroed by k_project_verts, a slice per block)
        dmr::launch_project_verts(*s, ps.vproj, is.mats, is.tile_count, (size_t)(is.scan_tmp + dmr::SCAN_TMP_BUCKETS - is.tile_count), st);
        dmr::launch_setup_faces(*s, tet, ps.vproj, d.gx, d.gy, d.r0, d.r1, fs.rect, fs.key_depth, fs.max_depth,
                                fs.tiles_touched, is.tile_count, st);
        dmr::launch_scan_tiles(d.ntiles, is.tile_count, is.tile_offset, is.tile_cursor, is.num_rendered, host_R, host_seq, is.tile_order,
                               is.scan_tmp, (uint32_t)std::min<uint64_t>(capacity, 0xffffffffu), ovf, st);
        return 0;
    };
    // tet: room for the forward's march sequence = the longest march the last backward that has run reported, + 25 % (0: none
    // yet -- this call's backward re-marches), within a memory budget.  The pinned word is created by the first default call
    // of the view configuration.
    size_t seq_steps = 0;
    if (tet) {
        std::lock_guard<std::mutex> lk(g_size_mu);
        SizeGuess& g = g_size_cache[size_key(s, tet, d)];
        if (!g.seq_steps && !async) {
            void* w = nullptr;
            if (hipHostMalloc(&w, 64, hipHostMallocCoherent | hipHostMallocPortable) == hipSuccess) {
                g.seq_steps = reinterpret_cast<uint32_t*>(w);
                *reinterpret_cast<volatile uint32_t*>(g.seq_steps) = 0u;
            } else (void)hipGetLastError();
        }
        if (g.seq_steps) {
            const uint64_t longest = *reinterpret_cast<volatile uint32_t*>(g.seq_steps);
            if (longest) {
                const uint64_t budget = 16ull << 30;  // bytes: beyond it the longest rays do not fit and such a call re-marches
                const uint64_t per4 = dmr::tet_seq_bytes((size_t)d.ntiles, 4);
                seq_steps = (size_t)(std::min<uint64_t>((longest + longest / 4 + 4 + 3) / 4, std::max<uint64_t>(budget / std::max<uint64_t>(per4, 1), 1)) * 4);
            }
        }
    }
    auto rest = [&](uint64_t capacity) -> int {
        BinningState tb, bs;
        const size_t mask_tiles = tet ? 0 : (size_t)d.ntiles;
        void* bb = alloc(ctx, DMR_BUF_BINNING, carve_binning(nullptr, (size_t)capacity, mask_tiles, (size_t)d.ntiles, seq_steps, tb));
        if (!bb && (capacity > 0 || seq_steps > 0)) return fail("binning allocation failed");
        carve_binning(bb, (size_t)capacity, mask_tiles, (size_t)d.ntiles, seq_steps, bs);

        if (capacity > 0) {
            dmr::launch_scatter_faces(*s, d.gx, d.gy, fs.rect, fs.key_depth, fs.tiles_touched, is.tile_cursor, bs.keys,
                                      (uint32_t)capacity, tet ? nullptr : is.mask_offset, bs.mask_offset, bs.mask_first, st);
            // (up to SCAN_SINGLE_MAX tiles the tri forward / the tet first-hit kernel sort a tile's list at the start of its workgroup)
            if (d.ntiles > dmr::SCAN_SINGLE_MAX)
                dmr::launch_sort_tiles(d.ntiles, is.tile_offset, is.tile_order, bs.keys, bs.face_list, bs.capacity, st);
        }
        render(bs);
        return 0;
    };

    const SizeKey key = size_key(s, tet, d);
    uint64_t guess = 0;
    {
        std::lock_guard<std::mutex> lk(g_size_mu);
        const double per_face = lookup_estimate(key, &SizeGuess::rendered_per_face);
        if (per_face > 0.0) guess = std::min<uint64_t>(padded((uint64_t)(per_face * (double)d.BF)), 0x7fffffffu);
    }
    if (async) {  // no host wait at all: capacity from the estimate, overflow checked on the device
        if (!guess || !overflow)
            return fail("asynchronous / captured call without a size estimate: run one default (waiting) call with the same "
                        "view configuration first");
        if (front(nullptr, 0u, guess, overflow) || rest(guess)) return 1;
        *num_rendered = (int)guess;  // the capacity: an upper bound the backward accepts in R's place
        DMR_HIP(hipGetLastError());
        return 0;
    }
    SizeRead* sr = size_read();
    if (!sr) return fail("hipHostMalloc failed");
    const uint32_t seq = sr->next_seq();
    if (front(const_cast<unsigned long long*>(sr->word(0)), seq, ~0ull, nullptr)) return 1;
    if (guess && rest(guess)) return 1;
    unsigned long long R64 = 0;
    if (wait_size(sr->word(0), seq, st, &R64)) return 1;  // the forward's one host wait (rasterizer_impl.cu:287-292)
    if (R64 > 0x7fffffffull) return fail("num_rendered overflows 31 bits");
    const int R = (int)R64;
    *num_rendered = R;
    if (!guess) {
        if (rest((uint64_t)R)) return 1;
    } else if ((uint64_t)R > guess) {  // the guess was too small: redo binning + render with the exact size
        g_redo_count.fetch_add(1, std::memory_order_relaxed);
        DMR_HIP(hipStreamSynchronize(st));
        if (front(nullptr, 0u, ~0ull, nullptr) || rest((uint64_t)R)) return 1;
    }
    {
        std::lock_guard<std::mutex> lk(g_size_mu);
        g_size_cache[key].rendered_per_face = (double)std::max(R, 1) / (double)std::max<size_t>(d.BF, 1);
    }
    DMR_HIP(hipGetLastError());
    return 0;
}

// ---- per-stage timing ---------------------------------------------------------------------------
std::atomic<uint32_t> g_prof_mask{0};
struct ProfRec { int stage; hipEvent_t a, b; };
std::mutex g_prof_mu;
std::vector<ProfRec*> g_prof_live, g_prof_free;

}  // namespace

namespace dmr {
StageScope::StageScope(int stage_, hipStream_t st_) : stage(stage_), st(st_), rec(nullptr) {
    if (!(g_prof_mask.load(std::memory_order_relaxed) & (1u << stage))) return;
    ProfRec* r = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        if (!g_prof_free.empty()) { r = g_prof_free.back(); g_prof_free.pop_back(); }
    }
    if (!r) {
        r = new ProfRec();
        if (hipEventCreate(&r->a) != hipSuccess || hipEventCreate(&r->b) != hipSuccess) { delete r; return; }
    }
    r->stage = stage;
    (void)hipEventRecord(r->a, st);
    rec = r;
}
StageScope::~StageScope() {
    if (!rec) return;
    ProfRec* r = reinterpret_cast<ProfRec*>(rec);
    (void)hipEventRecord(r->b, st);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_live.push_back(r);
}
}  // namespace dmr

extern "C" {

void dmr_profile_enable(uint32_t mask) { g_prof_mask.store(mask); }

int dmr_profile_collect(double* ms, int64_t* launches) {
    std::vector<ProfRec*> live;
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        live.swap(g_prof_live);
    }
    int rc = 0;
    for (ProfRec* r : live) {
        float t = 0.f;
        if (hipEventSynchronize(r->b) != hipSuccess || hipEventElapsedTime(&t, r->a, r->b) != hipSuccess) {
            g_err = "profile event read failed";
            rc = 1;
        } else if (r->stage >= 0 && r->stage < DMR_NUM_STAGES) {
            if (ms) ms[r->stage] += (double)t;
            if (launches) launches[r->stage] += 1;
        }
    }
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (ProfRec* r : live) g_prof_free.push_back(r);
    return rc;
}

const char* dmr_stage_name(int stage) {
    static const char* names[DMR_NUM_STAGES] = {"k_project_verts", "k_setup_faces", "k_scan_tiles", "k_scatter_faces",
                                                "k_sort_tiles", "k_tri_forward", "k_tri_backward_pix", "k_tri_unpack",
                                                "k_tet_first_intersect", "k_tet_forward", "k_tet_backward",
                                                "k_tri_backward_hits"};
    return (stage >= 0 && stage < DMR_NUM_STAGES) ? names[stage] : "?";
}

const char* dmr_last_error(void) { return g_err.c_str(); }

int dmr_overflowed(int device, int reset) {
    int dev = device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return 0;
    uint32_t* w = overflow_word(dev, false);
    if (!w) return 0;
    volatile uint32_t* v = w;
    const int r = *v != 0u;
    if (reset) *v = 0u;
    return r;
}
uint64_t dmr_redo_count(void) { return g_redo_count.load(std::memory_order_relaxed); }
int dmr_abi_version(void) { return DMR_ABI_VERSION; }
const char* dmr_build_arch(void) { return "gfx950"; }

int dmr_tri_forward(const dmr_scene* s, float* out_color, float* out_depth, dmr_alloc_fn alloc, void* ctx,
                    void* stream, int* num_rendered) {
    Dims d;
    if (check_scene(s, false, d)) return 1;
    if (!alloc || !num_rendered || !out_color || !out_depth) return fail("null argument");
    *num_rendered = 0;
    if (s->P == 0 || s->F == 0) return 0;  // render.cu:105 (and Q16: F == 0)
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    PointState ps; FaceState fs; ImageState is;
    auto render = [&](const BinningState& bs) {
        dmr::TriImageState img{is.final_T, is.final_prev_T, is.n_contrib, is.tile_hits, is.tile_bound, is.hit_offset, is.tile_used, is.tile_order, is.mask_offset};
        const dmr_scene sc = canonical(s, is.mats);
        dmr::launch_tri_forward(sc, d.gx, d.gy, d.r0, d.r1, ps.vproj, is.tile_offset, d.ntiles > dmr::SCAN_SINGLE_MAX ? nullptr : bs.keys,
                                bs.face_list, bs.capacity, img, out_color,
                                out_depth, st);
    };
    return run_forward(s, false, d, alloc, ctx, st, ps, fs, is, num_rendered, render);
}

int dmr_tri_backward(const dmr_scene* s, const float* dL_dcolor, const float* dL_ddepth, int num_rendered,
                     const void* point_buf, const void* face_buf, const void* binning_buf, const void* image_buf,
                     float* dL_dverts, float* dL_dvcolor, float* dL_dfopacity, float* dL_dvdepth, float* dL_dfintense,
                     dmr_alloc_fn alloc, void* ctx, void* stream) {
    Dims d;
    if (check_scene(s, false, d)) return 1;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // Nothing to back-propagate: no geometry (render.cu:173), nothing on screen, or an empty tile-row band (a rank without
    // rows; the asynchronous forward of such a band returns its capacity as num_rendered, hence the band test of its own).
    if (s->P == 0 || s->F == 0 || num_rendered <= 0 || d.r1 <= d.r0) {
        if (s->P > 0) {
            DMR_HIP(hipMemsetAsync(dL_dverts, 0, sizeof(float) * 3 * (size_t)s->P, st));
            DMR_HIP(hipMemsetAsync(dL_dvcolor, 0, sizeof(float) * 3 * (size_t)s->P, st));
            DMR_HIP(hipMemsetAsync(dL_dvdepth, 0, sizeof(float) * d.BP, st));
        }
        if (s->F > 0) {
            DMR_HIP(hipMemsetAsync(dL_dfopacity, 0, sizeof(float) * (size_t)s->F, st));
            DMR_HIP(hipMemsetAsync(dL_dfintense, 0, sizeof(float) * d.BF, st));
        }
        if (s->P > 0 && s->F > 0) {  // a default call leaves an estimate behind even so: later asynchronous calls need one
            std::lock_guard<std::mutex> lk(g_size_mu);
            SizeGuess& g = g_size_cache[size_key(s, false, d)];
            if (!(g.hits_per_face > 0.0)) g.hits_per_face = 1.0 / (double)std::max<size_t>(d.BF, 1);
        }
        return 0;
    }
    if (!point_buf || !face_buf || !binning_buf || !image_buf || !alloc) return fail("null scratch buffer");
    PointState ps; FaceState fs; ImageState is; BinningState bs;
    carve_point(const_cast<void*>(point_buf), d.BP, ps);
    carve_face(const_cast<void*>(face_buf), d.BF, (size_t)s->F, (size_t)s->T, false, fs);
    carve_image(const_cast<void*>(image_buf), (size_t)s->B, (size_t)d.ntiles, d.npix, false, is);
    carve_binning(const_cast<void*>(binning_buf), (size_t)num_rendered, 0, 0, 0, bs);

    // The forward counted the blended (pixel, face) pairs per tile and in total; the total sizes the record
    // buffer (the backward's one 8-byte host read; speculative sizing as in the forward).
    const bool async = (s->flags & DMR_FLAG_ASYNC) != 0 || stream_is_capturing(st);
    int dev = 0;
    DMR_HIP(hipGetDevice(&dev));
    uint32_t* overflow = overflow_word(dev, !async);
    const size_t vbytes = up(sizeof(float) * dmr::VROW * d.BP), fbytes = up(sizeof(float) * dmr::FROW * d.BF);
    const size_t pbytes = up(sizeof(float4) * 2 * (size_t)d.ntiles * dmr::TILE_PIX);  // per tile: its 256 pixels' (ray, upstream gradient) records
    // regions: see dmr::HitRegions -- null hit_offset: launch_scan_hits has laid the record regions out
    auto rest = [&](uint64_t capacity, dmr::HitRegions regions) -> int {
        const size_t hbytes = up(sizeof(dmr::HitRecord) * (size_t)capacity);
        char* work = reinterpret_cast<char*>(alloc(ctx, DMR_BUF_WORK, vbytes + fbytes + pbytes + hbytes));
        if (!work) return fail("workspace allocation failed");
        float* vrow = reinterpret_cast<float*>(work);
        float* frow = reinterpret_cast<float*>(work + vbytes);
        float4* pixrec = reinterpret_cast<float4*>(work + vbytes + fbytes);
        dmr::HitRecord* hits = reinterpret_cast<dmr::HitRecord*>(work + vbytes + fbytes + pbytes);
        dmr::TriImageState img{is.final_T, is.final_prev_T, is.n_contrib, is.tile_hits, is.tile_bound, is.hit_offset, is.tile_used, is.tile_order, is.mask_offset};
        const dmr_scene sc = canonical(s, is.mats);
        // (Splitting the tiles into bands whose hit-parallel kernel runs on a second stream while the next band's
        // per-pixel kernel computes -- atomic unit and SIMDs busy at the same time -- was measured and lost: 0.56 ms
        // per step with 1 band, 0.64 with 2, 0.70 with 4 at C4; cross-stream event waits cost more than the overlap.)
        // k_tri_backward_pix also zeroes the packed accumulators (every block a slice).  (Handing out the tiles'
        // record regions from an atomic cursor: the 2.9 k returning same-address atomics stall their waves, in order
        // with every younger load -- 17 us.)
        dmr::launch_tri_backward_pix(sc, d.gx, d.gy, d.r0, d.r1, ps.vproj, is.tile_offset, bs.face_list, img,
                                     dL_dcolor, dL_ddepth, pixrec, hits, (uint32_t)capacity,
                                     reinterpret_cast<float*>(work), (vbytes + fbytes) / sizeof(float), regions, st);
        dmr::launch_tri_backward_hits(sc, d.gx, d.gy, ps.vproj, bs.face_list, img, pixrec, hits, (uint32_t)capacity, vrow, frow, st);
        dmr::launch_tri_unpack(*s, vrow, frow, dL_dverts, dL_dvcolor, dL_dfopacity, dL_dvdepth, dL_dfintense, st);
        return 0;
    };
    const dmr::HitRegions scanned{nullptr, nullptr, nullptr, nullptr, 0u};
    // With a size estimate and few enough tiles the per-pixel kernel lays the regions out itself: no scan launch.
    const bool self_regions = d.ntiles <= dmr::SCAN_SINGLE_MAX;
    const SizeKey key = size_key(s, false, d);
    uint64_t guess = 0;
    {
        std::lock_guard<std::mutex> lk(g_size_mu);
        const double per_face = lookup_estimate(key, &SizeGuess::hits_per_face);
        if (per_face > 0.0) guess = std::min<uint64_t>(padded((uint64_t)(per_face * (double)d.BF)), 0xfffffffeull);
    }
    if (async) {  // no host wait: see run_forward
        if (!guess || !overflow)
            return fail("asynchronous / captured call without a size estimate: run one default (waiting) backward with the "
                        "same view configuration first");
        if (self_regions) {
            if (rest(guess, dmr::HitRegions{is.hit_offset, is.hit_total, nullptr, overflow, 0u})) return 1;
        } else {
            dmr::launch_scan_hits(d.ntiles, is.tile_hits, is.tile_offset, is.hit_offset, is.tile_used, is.hit_total, nullptr, 0u, is.scan_tmp, (uint32_t)guess, overflow, st);
            if (rest(guess, scanned)) return 1;
        }
        DMR_HIP(hipGetLastError());
        return 0;
    }
    SizeRead* sr = size_read();
    if (!sr) return fail("hipHostMalloc failed");
    unsigned long long* host_total = const_cast<unsigned long long*>(sr->word(1));
    const uint32_t seq = sr->next_seq();
    if (guess && self_regions) {
        // everything is enqueued with the estimate; the total arrives behind the per-pixel kernel
        if (rest(guess, dmr::HitRegions{is.hit_offset, is.hit_total, host_total, nullptr, seq})) return 1;
    } else {
        dmr::launch_scan_hits(d.ntiles, is.tile_hits, is.tile_offset, is.hit_offset, is.tile_used, is.hit_total, host_total, seq, is.scan_tmp, 0xffffffffu, nullptr, st);
        if (guess && rest(guess, scanned)) return 1;
    }
    unsigned long long nhits = 0;
    if (wait_size(sr->word(1), seq, st, &nhits)) return 1;
    if (nhits >= 0xffffffffull) return fail("more than 2^32 blended (pixel, face) pairs");
    if (!guess) {
        if (rest(nhits, scanned)) return 1;
    } else if (nhits > guess) {  // (the redo pass does not store into the pinned slot: a later call may own it by then)
        g_redo_count.fetch_add(1, std::memory_order_relaxed);
        DMR_HIP(hipStreamSynchronize(st));
        if (self_regions) {
            if (rest(nhits, dmr::HitRegions{is.hit_offset, is.hit_total, nullptr, nullptr, 0u})) return 1;
        } else if (rest(nhits, scanned)) return 1;
    }
    {
        std::lock_guard<std::mutex> lk(g_size_mu);
        g_size_cache[key].hits_per_face = (double)std::max<uint64_t>(nhits, 1) / (double)std::max<size_t>(d.BF, 1);
    }
    DMR_HIP(hipGetLastError());
    return 0;
}

int dmr_tet_forward(const dmr_scene* s, float* out_color, float* out_depth, float* out_active, dmr_alloc_fn alloc,
                    void* ctx, void* stream, int* num_rendered) {
    Dims d;
    if (check_scene(s, true, d)) return 1;
    if (!alloc || !num_rendered || !out_color || !out_depth || !out_active) return fail("null argument");
    *num_rendered = 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    PointState ps; FaceState fs; ImageState is;
    auto render = [&](const BinningState& bs) {
        dmr::TetImageState img{is.final_T, is.final_prev_T, is.n_contrib, is.first_face, is.first_tet,
                               is.last_face, is.last_tet, is.is_active, fs.facerec, fs.colrec, fs.tetrec, is.seed,
                               is.seq, bs.base};
        const dmr_scene sc = canonical(s, is.mats);
        dmr::launch_tet_prep(sc, img, bs.seq_steps, bs.seq_offset, st);
        dmr::launch_tet_first_intersect(sc, d.gx, d.gy, d.r0, d.r1, fs.key_depth, fs.max_depth, is.tile_offset,
                                        d.ntiles > dmr::SCAN_SINGLE_MAX ? nullptr : bs.keys, bs.face_list, bs.capacity, img, st);
        dmr::launch_tet_forward(sc, d.gx, d.gy, d.r0, d.r1, img, out_color, out_depth, out_active, st);
    };
    return run_forward(s, true, d, alloc, ctx, st, ps, fs, is, num_rendered, render);
}

int dmr_tet_backward(const dmr_scene* s, const float* dL_dcolor, const float* dL_ddepth, const void* point_buf,
                     const void* face_buf, const void* binning_buf, const void* image_buf, float* dL_dvcolor,
                     float* dL_dfopacity, dmr_alloc_fn alloc, void* ctx, void* stream) {
    Dims d;
    if (check_scene(s, true, d)) return 1;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dmr::launch_tet_zero_grads(dL_dvcolor, 3 * (int64_t)s->P, dL_dfopacity, (int64_t)s->F, st);
    if (s->P == 0 || s->F == 0) return 0;
    if (!image_buf || !face_buf) return fail("null scratch buffer");
    ImageState is; FaceState fs;
    carve_image(const_cast<void*>(image_buf), (size_t)s->B, (size_t)d.ntiles, d.npix, true, is);
    carve_face(const_cast<void*>(face_buf), d.BF, (size_t)s->F, (size_t)s->T, true, fs);
    // binning_buf: the forward's march sequence lives there (where and how much of it: is.seq, on the device); a null
    // buffer is fine when the forward ran without one (its descriptor then says cap = 0 and the backward re-marches)
    dmr::TetImageState img{is.final_T, is.final_prev_T, is.n_contrib, is.first_face, is.first_tet,
                           is.last_face, is.last_tet, is.is_active, fs.facerec, fs.colrec, fs.tetrec, is.seed,
                           is.seq, reinterpret_cast<char*>(const_cast<void*>(binning_buf))};
    const dmr_scene sc = canonical(s, is.mats);
    uint32_t* host_seq_steps = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_size_mu);
        auto it = g_size_cache.find(size_key(s, true, d));
        if (it != g_size_cache.end()) host_seq_steps = it->second.seq_steps;
    }
    dmr::launch_tet_backward(sc, d.gx, d.gy, d.r0, d.r1, img, dL_dcolor, dL_ddepth, dL_dvcolor, dL_dfopacity, host_seq_steps, st);
    DMR_HIP(hipGetLastError());
    return 0;
}

// ---- 4x4 inverses (caller-side step) ---------------------------------------
namespace {
__global__ void k_invert_mats(const float* __restrict__ in, int count, int transposed, float* __restrict__ out) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= count) return;
    double a[16];
#pragma unroll
    for (int k = 0; k < 16; k++) a[k] = (double)in[16 * m + (transposed ? 4 * (k & 3) + (k >> 2) : k)];  // a[4*i + j]
    // cofactor expansion with the six 2x2 minors of the upper and of the lower two rows
    const double s0 = a[0] * a[5] - a[1] * a[4], s1 = a[0] * a[6] - a[2] * a[4], s2 = a[0] * a[7] - a[3] * a[4];
    const double s3 = a[1] * a[6] - a[2] * a[5], s4 = a[1] * a[7] - a[3] * a[5], s5 = a[2] * a[7] - a[3] * a[6];
    const double c5 = a[10] * a[15] - a[11] * a[14], c4 = a[9] * a[15] - a[11] * a[13], c3 = a[9] * a[14] - a[10] * a[13];
    const double c2 = a[8] * a[15] - a[11] * a[12], c1 = a[8] * a[14] - a[10] * a[12], c0 = a[8] * a[13] - a[9] * a[12];
    const double det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
    const double r = 1.0 / det;
    double b[16];
    b[0] = (a[5] * c5 - a[6] * c4 + a[7] * c3) * r;   b[1] = (-a[1] * c5 + a[2] * c4 - a[3] * c3) * r;
    b[2] = (a[13] * s5 - a[14] * s4 + a[15] * s3) * r; b[3] = (-a[9] * s5 + a[10] * s4 - a[11] * s3) * r;
    b[4] = (-a[4] * c5 + a[6] * c2 - a[7] * c1) * r;  b[5] = (a[0] * c5 - a[2] * c2 + a[3] * c1) * r;
    b[6] = (-a[12] * s5 + a[14] * s2 - a[15] * s1) * r; b[7] = (a[8] * s5 - a[10] * s2 + a[11] * s1) * r;
    b[8] = (a[4] * c4 - a[5] * c2 + a[7] * c0) * r;   b[9] = (-a[0] * c4 + a[1] * c2 - a[3] * c0) * r;
    b[10] = (a[12] * s4 - a[13] * s2 + a[15] * s0) * r; b[11] = (-a[8] * s4 + a[9] * s2 - a[11] * s0) * r;
    b[12] = (-a[4] * c3 + a[5] * c1 - a[6] * c0) * r; b[13] = (a[0] * c3 - a[1] * c1 + a[2] * c0) * r;
    b[14] = (-a[12] * s3 + a[13] * s1 - a[14] * s0) * r; b[15] = (a[8] * s3 - a[9] * s1 + a[10] * s0) * r;
#pragma unroll
    for (int k = 0; k < 16; k++) out[16 * m + k] = (float)b[k];
}
}  // namespace

int dmr_invert_mats(const float* in, int count, int transposed, float* out, void* stream) {
    if (count < 0 || (count > 0 && (!in || !out))) return fail("dmr_invert_mats: bad arguments");
    if (count == 0) return 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    k_invert_mats<<<dim3((unsigned)((count + 63) / 64)), dim3(64), 0, st>>>(in, count, transposed, out);
    DMR_HIP(hipGetLastError());
    return 0;
}

// ---- parity/debug export ---------------------------------------------------------------------
namespace {
__global__ void k_export_vproj(const float4* v, int64_t n, int what, float* dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (what == 0) { dst[2 * i] = v[i].x; dst[2 * i + 1] = v[i].y; }
    else dst[i] = v[i].z;
}
__global__ void k_export_ranges(const uint32_t* off, const uint32_t* cnt, int64_t n, uint32_t* dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // the reference leaves empty tiles at (0, 0) (cudaMemset, rasterizer_impl.cu:330)
    const bool empty = cnt[i] == 0;
    dst[2 * i] = empty ? 0u : off[i];
    dst[2 * i + 1] = empty ? 0u : off[i + 1];
}
}  // namespace

int64_t dmr_export(const dmr_scene* s, int is_tet, int num_rendered, const char* name, const void* point_buf,
                   const void* face_buf, const void* binning_buf, const void* image_buf, void* dst, int64_t cap,
                   void* stream) {
    Dims d;
    if (!name || check_scene(s, is_tet != 0, d)) return -1;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    PointState ps; FaceState fs; ImageState is; BinningState bs;
    carve_point(const_cast<void*>(point_buf), d.BP, ps);
    carve_face(const_cast<void*>(face_buf), d.BF, (size_t)s->F, (size_t)s->T, is_tet != 0, fs);
    carve_image(const_cast<void*>(image_buf), (size_t)s->B, (size_t)d.ntiles, d.npix, is_tet != 0, is);
    carve_binning(const_cast<void*>(binning_buf), (size_t)std::max(0, num_rendered), 0, 0, 0, bs);
    const std::string n(name);
    auto plain = [&](const void* src, size_t bytes) -> int64_t {
        if (dst && src && bytes) {
            if (hipMemcpyAsync(dst, src, std::min<size_t>(bytes, (size_t)cap), hipMemcpyDeviceToDevice, st) != hipSuccess) {
                g_err = "export copy failed";
                return -1;
            }
        }
        return (int64_t)bytes;
    };
    if (n == "image" || n == "ndc_z") {
        const size_t bytes = d.BP * (n == "image" ? 8 : 4);
        if (dst && d.BP && (size_t)cap >= bytes)
            k_export_vproj<<<dim3((unsigned)((d.BP + 255) / 256)), dim3(256), 0, st>>>(
                ps.vproj, (int64_t)d.BP, n == "image" ? 0 : 1, reinterpret_cast<float*>(dst));
        return (int64_t)bytes;
    }
    if (n == "ranges") {
        const size_t bytes = (size_t)d.ntiles * 8;
        if (dst && (size_t)cap >= bytes)
            k_export_ranges<<<dim3((unsigned)((d.ntiles + 255) / 256)), dim3(256), 0, st>>>(
                is.tile_offset, is.tile_count, d.ntiles, reinterpret_cast<uint32_t*>(dst));
        return (int64_t)bytes;
    }
    if (n == "key_depth") return plain(fs.key_depth, d.BF * 4);
    if (n == "max_depth") return is_tet ? plain(fs.max_depth, d.BF * 4) : -1;
    if (n == "tiles_touched") return plain(fs.tiles_touched, d.BF * 4);
    if (n == "face_list") return plain(bs.face_list, (size_t)std::max(0, num_rendered) * 4);
    if (n == "final_T") return plain(is.final_T, d.npix * 4);
    if (n == "final_prev_T") return plain(is.final_prev_T, d.npix * 4);
    if (n == "n_contrib") return plain(is.n_contrib, d.npix * 4);
    if (n == "tile_hits") return is_tet ? -1 : plain(is.tile_hits, (size_t)d.ntiles * 4);
    if (is_tet) {
        if (n == "first_face") return plain(is.first_face, d.npix * 4);
        if (n == "first_tet") return plain(is.first_tet, d.npix * 4);
        if (n == "last_face") return plain(is.last_face, d.npix * 4);
        if (n == "last_tet") return plain(is.last_tet, d.npix * 4);
        if (n == "is_active") return plain(is.is_active, d.npix);
        if (n == "tet_seq") return plain(is.seq, 8);  // {longest march, steps the forward's march sequence had room for}
    }
    g_err = "unknown export item: " + n;
    return -1;
}

}  // extern "C"
