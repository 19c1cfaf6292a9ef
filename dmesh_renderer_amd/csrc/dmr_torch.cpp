// dmr_torch.cpp -- `dmesh_renderer_amd._C`: the binding surface of the reference extension (ext.cpp:4-11) as a
// compiled PyTorch-ROCm module over the C ABI of libdmesh_renderer_hip.so (include/dmesh_renderer_amd.h).
//
//     render_tris            <- RasterizeTrianglesCUDA          (render.cu:29-132)
//     render_tris_backward   <- RasterizeTrianglesBackwardCUDA  (render.cu:134-208)
//     render_tets            <- RenderFTetsCUDA                 (render.cu:213-336)
//     render_tets_backward   <- RenderFTetsBackwardCUDA         (render.cu:338-412)
//
// Same positional arguments, tuple arity, dtypes / shapes and error behaviour (RuntimeError with the reference's
// messages, render.cu:49-79,237-277).  This file is what render.cu is in the reference: shape checks, output and
// scratch allocation (PyTorch's caching allocator: no device allocation in the steady state, and the graph-private
// pool under torch.cuda.graph capture), the current stream, raw-pointer hand-off.  It holds no compute and no HIP
// kernel; the C ABI library is loaded with dlopen (DMR_LIBRARY=<path> selects another build of it, e.g. the ablation
// build of build.py --ablation) and nothing here works without it: there is no fallback of any kind.
//
// Extensions over the reference (keyword arguments, all optional):
//   rows=(begin, end)        a band of 16-pixel tile rows (multi-GPU tile-row sharding); (0, 0) = everything
//   fill_outside=True        render_tris: zero the pixels outside the band (False: leave them uninitialised)
//   flat_out=None            backward: one caller-owned fp32 buffer receiving the gradients back to back (the payload
//                            of the one all-reduce in sharding.py); the returned tensors are views into it
//   set_async(True)          calls never wait for the device (DMR_FLAG_ASYNC; automatic under stream capture):
//                            `num_rendered` is then the capacity used, overflowed() reports a scene that outgrew it
#include <torch/extension.h>

#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <dlfcn.h>

#include <array>
#include <atomic>
#include <cstdlib>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/dmesh_renderer_amd.h"

namespace {

constexpr int NUM_CHANNELS = 3;  // cuda_*/config.h:4

[[noreturn]] void err(const std::string& m) { throw std::runtime_error(m); }

// ---- the C ABI, bound at import ---------------------------------------------------------------------------------
struct Abi {
    void* handle = nullptr;
    std::string path;
    decltype(&dmr_tri_forward) tri_forward = nullptr;
    decltype(&dmr_tri_backward) tri_backward = nullptr;
    decltype(&dmr_tet_forward) tet_forward = nullptr;
    decltype(&dmr_tet_backward) tet_backward = nullptr;
    decltype(&dmr_invert_mats) invert_mats = nullptr;
    decltype(&dmr_export) export_item = nullptr;
    decltype(&dmr_profile_enable) profile_enable = nullptr;
    decltype(&dmr_profile_collect) profile_collect = nullptr;
    decltype(&dmr_stage_name) stage_name = nullptr;
    decltype(&dmr_last_error) last_error = nullptr;
    decltype(&dmr_overflowed) overflowed = nullptr;
    decltype(&dmr_redo_count) redo_count = nullptr;
    decltype(&dmr_abi_version) abi_version = nullptr;
    decltype(&dmr_build_arch) build_arch = nullptr;
};

Abi g_abi;
std::atomic<int> g_async{0};

template <class F>
void bind(F& fn, const char* name) {
    void* p = dlsym(g_abi.handle, name);
    if (!p) throw std::runtime_error(g_abi.path + " does not export " + name);
    fn = reinterpret_cast<F>(p);
}

void load_abi() {
    const char* env = std::getenv("DMR_LIBRARY");
    std::string path;
    if (env && *env) path = env;
    else {
        Dl_info info;
        if (!dladdr(reinterpret_cast<void*>(&load_abi), &info) || !info.dli_fname)
            throw std::runtime_error("cannot locate the dmesh_renderer_amd package directory");
        path = info.dli_fname;
        const size_t slash = path.find_last_of('/');
        path = (slash == std::string::npos ? std::string(".") : path.substr(0, slash)) + "/libdmesh_renderer_hip.so";
    }
    g_abi.path = path;
    g_abi.handle = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!g_abi.handle)
        throw std::runtime_error(path + " cannot be loaded (" + std::string(dlerror() ? dlerror() : "?") +
                                 "): the HIP library is not built (run `python -m dmesh_renderer_amd.build`); there is no CPU fallback");
    bind(g_abi.tri_forward, "dmr_tri_forward"); bind(g_abi.tri_backward, "dmr_tri_backward");
    bind(g_abi.tet_forward, "dmr_tet_forward"); bind(g_abi.tet_backward, "dmr_tet_backward");
    bind(g_abi.invert_mats, "dmr_invert_mats"); bind(g_abi.export_item, "dmr_export");
    bind(g_abi.profile_enable, "dmr_profile_enable"); bind(g_abi.profile_collect, "dmr_profile_collect");
    bind(g_abi.stage_name, "dmr_stage_name"); bind(g_abi.last_error, "dmr_last_error");
    bind(g_abi.overflowed, "dmr_overflowed"); bind(g_abi.redo_count, "dmr_redo_count"); bind(g_abi.abi_version, "dmr_abi_version");
    bind(g_abi.build_arch, "dmr_build_arch");
    if (g_abi.abi_version() != DMR_ABI_VERSION)
        throw std::runtime_error("ABI mismatch: " + path + " is version " + std::to_string(g_abi.abi_version()) +
                                 ", the binding expects " + std::to_string(DMR_ABI_VERSION));
}

[[noreturn]] void raise_lib() { err(g_abi.last_error()); }

// ---- argument handling (render.cu:49-79,113-129,237-277) ------------------------------------------------------------
std::string dtype_name(const at::Tensor& t) { return std::string("torch.") + c10::toString(t.scalar_type()); }

at::Tensor f32(const at::Tensor& t, const char* name) {
    // render.cu:113-129: `.contiguous().data<float>()` throws for any other dtype
    if (t.scalar_type() != at::kFloat) err("expected scalar type Float but found " + dtype_name(t) + " (" + name + ")");
    return t.contiguous();
}
at::Tensor i32(const at::Tensor& t, const char* name) {
    if (t.scalar_type() != at::kInt) err("expected scalar type Int but found " + dtype_name(t) + " (" + name + ")");
    return t.contiguous();
}
// [B,4,4] matrix -> (tensor to keep alive, transposed-storage flag).  render.cu:117-120 makes the `.transpose(1, 2)`
// views of the wrapper contiguous with four copy kernels per call; the library reads that storage in place instead
// (dmr_scene.mats_transposed).
std::pair<at::Tensor, int> mat(const at::Tensor& t, const char* name) {
    if (t.scalar_type() != at::kFloat) err("expected scalar type Float but found " + dtype_name(t) + " (" + name + ")");
    if (t.is_contiguous()) return {t, 0};
    if (t.dim() == 3 && t.transpose(1, 2).is_contiguous()) return {t, 1};
    return {t.contiguous(), 0};
}

void check_common(const at::Tensor& verts, const at::Tensor& faces, const at::Tensor& verts_color,
                  const at::Tensor& faces_opacity, const at::Tensor& mv, const at::Tensor& proj, const at::Tensor& inv_mv,
                  const at::Tensor& inv_proj, const at::Tensor& verts_depth, const at::Tensor& faces_intense, bool tet) {
    // messages: render.cu:49-79 (tri) and :237-267 (tet)
    if (verts.dim() != 2 || verts.size(1) != 3) err("verts must have dimensions (num_points, 3)");
    if (faces.dim() != 2 || faces.size(1) != 3) err("faces must have dimensions (num_faces, 3)");
    if (tet) {
        if (verts_color.dim() != 2 || verts_color.size(0) != verts.size(0) || verts_color.size(1) != 3)
            err("vert_color must have dimensions (num_verts, 3)");
        if (faces_opacity.dim() != 1 || faces_opacity.size(0) != faces.size(0)) err("face_opacity must have dimensions (num_faces)");
    } else {
        if (verts_color.dim() != 2 || verts_color.size(0) != verts.size(0)) err("vert color must have dimensions (num_points, N)");
        if (verts_color.size(1) != NUM_CHANNELS) err("vert color must have dimensions (num_points, 3)");  // Q15
        if (faces_opacity.dim() != 1 || faces_opacity.size(0) != faces.size(0)) err("face opacity must have dimensions (num_faces,)");
    }
    const std::string bdim = tet ? "batch_size" : "B";
    const std::pair<const at::Tensor*, const char*> mats[4] = {{&mv, "mv_mats"}, {&proj, "proj_mats"}, {&inv_mv, "inv_mv_mats"},
                                                               {&inv_proj, "inv_proj_mats"}};
    for (auto& m : mats)
        if (m.first->dim() != 3 || m.first->size(1) != 4 || m.first->size(2) != 4)
            err(std::string(m.second) + " must have dimensions (" + bdim + ", 4, 4)");
    if (verts_depth.dim() != 2 || verts_depth.size(1) != verts.size(0))
        err(tet ? "verts_depth must have dimensions (batch_size, num_verts)" : "verts_depth must have dimensions (B, num_points,)");
    if (faces_intense.dim() != 2 || faces_intense.size(1) != faces.size(0))
        err(tet ? "faces_intense must have dimensions (batch_size, num_faces)" : "faces_intense must have dimensions (B, num_faces,)");
    const int64_t B = mv.size(0);
    const std::pair<const at::Tensor*, const char*> batched[5] = {{&proj, "proj_mats"}, {&inv_mv, "inv_mv_mats"},
                                                                  {&inv_proj, "inv_proj_mats"}, {&verts_depth, "verts_depth"},
                                                                  {&faces_intense, "faces_intense"}};
    for (auto& m : batched)
        if (m.first->size(0) != B)  // the reference would read out of bounds here
            err(std::string(m.second) + " must have the batch size of mv_mats (" + std::to_string(B) + ")");
}

c10::Device hip_device_of(const at::Tensor& verts) {
    if (!verts.is_cuda())
        err("dmesh_renderer_amd has no CPU path: tensors must be on a HIP device "
            "(the reference allocates on torch::kCUDA unconditionally, render.cu:91-96)");
    return verts.device();
}

// the caller-owned scratch buffers of one call: the C equivalent of the reference's four resizeFunctional lambdas
// (render.cu:18-24,91-100), plus the backward's transient workspace
struct Scratch {
    c10::Device dev;
    std::array<at::Tensor, 5> buf;
    explicit Scratch(c10::Device d) : dev(d) {}
    at::Tensor get(int which) const {
        return buf[which].defined() ? buf[which] : at::empty({0}, at::TensorOptions().dtype(at::kByte).device(dev));
    }
};
void* alloc_cb(void* ctx, int which, size_t nbytes) {
    auto* s = reinterpret_cast<Scratch*>(ctx);
    if (which < 0 || which >= 5) return nullptr;
    try {
        s->buf[which] = at::empty({(int64_t)std::max<size_t>(nbytes, 1)}, at::TensorOptions().dtype(at::kByte).device(s->dev));
        return s->buf[which].data_ptr();
    } catch (...) {  // reported by the library as an allocation failure
        return nullptr;
    }
}

// Owns the contiguous input tensors and the dmr_scene of one call.
struct Call {
    c10::Device dev;
    std::vector<at::Tensor> keep;
    dmr_scene sc{};
    Scratch scratch;

    template <class T>
    const T* ptr(const at::Tensor& t) { keep.push_back(t); return t.numel() ? reinterpret_cast<const T*>(t.data_ptr()) : nullptr; }

    Call(c10::Device dev_, const at::Tensor& bg, const at::Tensor& verts, const at::Tensor& faces, const at::Tensor& verts_color,
         const at::Tensor& faces_opacity, const at::Tensor& mv, const at::Tensor& proj, const at::Tensor& inv_mv,
         const at::Tensor& inv_proj, const at::Tensor& verts_depth, const at::Tensor& faces_intense, int64_t H, int64_t W,
         const at::Tensor* tets, const at::Tensor* face_tets, const at::Tensor* tet_faces, int64_t seed, std::pair<int, int> rows)
        : dev(dev_), scratch(dev_) {
        keep.reserve(16);
        auto m0 = mat(mv, "mv_mats"), m1 = mat(proj, "proj_mats"), m2 = mat(inv_mv, "inv_mv_mats"), m3 = mat(inv_proj, "inv_proj_mats");
        if (bg.numel() < NUM_CHANNELS) err("background must have 3 channels");
        sc.B = (int32_t)mv.size(0); sc.P = (int32_t)verts.size(0); sc.F = (int32_t)faces.size(0);
        sc.T = tets ? (int32_t)tets->size(0) : 0;
        sc.W = (int32_t)W; sc.H = (int32_t)H;
        sc.background = ptr<float>(f32(bg, "background"));
        sc.verts = ptr<float>(f32(verts, "verts"));
        sc.faces = ptr<int32_t>(i32(faces, "faces"));
        sc.verts_color = ptr<float>(f32(verts_color, "verts_color"));
        sc.faces_opacity = ptr<float>(f32(faces_opacity, "faces_opacity"));
        sc.mv_mats = ptr<float>(m0.first); sc.proj_mats = ptr<float>(m1.first);
        sc.inv_mv_mats = ptr<float>(m2.first); sc.inv_proj_mats = ptr<float>(m3.first);
        sc.verts_depth = ptr<float>(f32(verts_depth, "verts_depth"));
        sc.faces_intense = ptr<float>(f32(faces_intense, "faces_intense"));
        if (tets) {
            sc.tets = ptr<int32_t>(i32(*tets, "tets"));
            sc.face_tets = ptr<int32_t>(i32(*face_tets, "face_tets"));
            sc.tet_faces = ptr<int32_t>(i32(*tet_faces, "tet_faces"));
        }
        sc.ray_random_seed = (int32_t)seed;
        sc.row_begin = rows.first; sc.row_end = rows.second;
        sc.mats_transposed = m0.second | (m1.second << 1) | (m2.second << 2) | (m3.second << 3);
        sc.flags = g_async.load(std::memory_order_relaxed) ? DMR_FLAG_ASYNC : 0;
        for (const at::Tensor& t : keep)
            if (t.device() != dev) err("all tensors must be on " + dev.str() + " (one is on " + t.device().str() + ")");
    }

    void* stream() const { return reinterpret_cast<void*>(c10::hip::getCurrentHIPStream(dev.index()).stream()); }
};

template <class T> T* mptr(const at::Tensor& t) { return t.numel() ? reinterpret_cast<T*>(t.data_ptr()) : nullptr; }

at::TensorOptions f32_on(c10::Device dev) { return at::TensorOptions().dtype(at::kFloat).device(dev); }

// ---- the four functions of ext.cpp:6-11 -----------------------------------------------------------------------------
// -> (num_rendered:int, color [B,3,H,W], depth [B,1,H,W], pointBuffer, faceBuffer, binningBuffer, imgBuffer)
using TriFwdOut = std::tuple<int64_t, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor>;
TriFwdOut render_tris(const at::Tensor& background, const at::Tensor& verts, const at::Tensor& faces, const at::Tensor& verts_color,
                      const at::Tensor& faces_opacity, const at::Tensor& mv_mats, const at::Tensor& proj_mats,
                      const at::Tensor& inv_mv_mats, const at::Tensor& inv_proj_mats, const at::Tensor& verts_depth,
                      const at::Tensor& faces_intense, int64_t image_height, int64_t image_width, std::pair<int, int> rows,
                      bool fill_outside) {
    check_common(verts, faces, verts_color, faces_opacity, mv_mats, proj_mats, inv_mv_mats, inv_proj_mats, verts_depth, faces_intense, false);
    const c10::Device dev = hip_device_of(verts);
    c10::DeviceGuard guard(dev);
    Call call(dev, background, verts, faces, verts_color, faces_opacity, mv_mats, proj_mats, inv_mv_mats, inv_proj_mats, verts_depth,
              faces_intense, image_height, image_width, nullptr, nullptr, nullptr, 0, rows);
    // the kernels write every pixel of the rendered rows; zero-fill (render.cu:88-89) is only needed when nothing is
    // launched (P == 0 / F == 0, render.cu:105) or when a band leaves rows untouched
    const bool full = ((rows.first == 0 && rows.second == 0) || !fill_outside) && call.sc.P > 0 && call.sc.F > 0;
    const auto opt = f32_on(dev);
    at::Tensor color = full ? at::empty({call.sc.B, NUM_CHANNELS, image_height, image_width}, opt)
                            : at::zeros({call.sc.B, NUM_CHANNELS, image_height, image_width}, opt);
    at::Tensor depth = full ? at::empty({call.sc.B, 1, image_height, image_width}, opt) : at::zeros({call.sc.B, 1, image_height, image_width}, opt);
    int rendered = 0;
    // (the bindings release the GIL around this whole function: the default call waits for the size read-back)
    if (g_abi.tri_forward(&call.sc, mptr<float>(color), mptr<float>(depth), &alloc_cb, &call.scratch, call.stream(), &rendered)) raise_lib();
    return TriFwdOut((int64_t)rendered, color, depth, call.scratch.get(DMR_BUF_POINT), call.scratch.get(DMR_BUF_FACE),
                          call.scratch.get(DMR_BUF_BINNING), call.scratch.get(DMR_BUF_IMAGE));
}

// -> (dL_dverts [P,3], dL_dvcolor [P,3], dL_dfopacity [F], dL_dvdepth [B,P], dL_dfintense [B,F])
using TriBwdOut = std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor>;
TriBwdOut render_tris_backward(const at::Tensor& background, const at::Tensor& verts, const at::Tensor& faces, const at::Tensor& verts_color,
                               const at::Tensor& faces_opacity, const at::Tensor& mv_mats, const at::Tensor& proj_mats,
                               const at::Tensor& inv_mv_mats, const at::Tensor& inv_proj_mats, const at::Tensor& verts_depth,
                               const at::Tensor& faces_intense, const at::Tensor& dL_dout_color, const at::Tensor& dL_dout_depth,
                               int64_t R, const at::Tensor& pointBuffer, const at::Tensor& faceBuffer, const at::Tensor& binningBuffer,
                               const at::Tensor& imageBuffer, std::pair<int, int> rows, const std::optional<at::Tensor>& flat_out) {
    const c10::Device dev = hip_device_of(verts);
    c10::DeviceGuard guard(dev);
    if (dL_dout_color.dim() != 4) err("dL_dout_color must have dimensions (B, 3, H, W)");
    const int64_t H = dL_dout_color.size(2), W = dL_dout_color.size(3);  // render.cu:163-164
    Call call(dev, background, verts, faces, verts_color, faces_opacity, mv_mats, proj_mats, inv_mv_mats, inv_proj_mats, verts_depth,
              faces_intense, H, W, nullptr, nullptr, nullptr, 0, rows);
    const at::Tensor gc = f32(dL_dout_color, "dL_dout_color");  // may arrive non-contiguous / expanded (render.cu:197-198)
    const at::Tensor gd = f32(dL_dout_depth, "dL_dout_depth");
    const int64_t B = call.sc.B, P = call.sc.P, F = call.sc.F;
    at::Tensor g_verts, g_vcolor, g_fop, g_vdepth, g_fint;
    if (!flat_out.has_value()) {
        const auto opt = f32_on(dev);
        g_verts = at::empty({P, 3}, opt); g_vcolor = at::empty({P, NUM_CHANNELS}, opt); g_fop = at::empty({F}, opt);
        g_vdepth = at::empty({B, P}, opt); g_fint = at::empty({B, F}, opt);
    } else {
        const at::Tensor& fo = *flat_out;
        const int64_t total = 6 * P + F + B * (P + F);
        if (fo.scalar_type() != at::kFloat || fo.device() != dev || !fo.is_contiguous() || fo.numel() != total)
            err("flat_out must be a contiguous float32 tensor of " + std::to_string(total) + " elements on " + dev.str());
        const at::Tensor flat = fo.view({-1});
        int64_t o = 0;
        g_verts = flat.narrow(0, o, 3 * P).view({P, 3}); o += 3 * P;
        g_vcolor = flat.narrow(0, o, 3 * P).view({P, NUM_CHANNELS}); o += 3 * P;
        g_fop = flat.narrow(0, o, F); o += F;
        g_vdepth = flat.narrow(0, o, B * P).view({B, P}); o += B * P;
        g_fint = flat.narrow(0, o, B * F).view({B, F});
    }
    const at::Tensor pb = pointBuffer.contiguous(), fb = faceBuffer.contiguous(), bb = binningBuffer.contiguous(), ib = imageBuffer.contiguous();
    if (g_abi.tri_backward(&call.sc, mptr<const float>(gc), mptr<const float>(gd), (int)R, mptr<const void>(pb), mptr<const void>(fb),
                           mptr<const void>(bb), mptr<const void>(ib), mptr<float>(g_verts), mptr<float>(g_vcolor), mptr<float>(g_fop),
                           mptr<float>(g_vdepth), mptr<float>(g_fint), &alloc_cb, &call.scratch, call.stream()))
        raise_lib();
    return TriBwdOut(g_verts, g_vcolor, g_fop, g_vdepth, g_fint);
}

void check_tets(const at::Tensor& faces, const at::Tensor& tets, const at::Tensor& face_tets, const at::Tensor& tet_faces) {
    // render.cu:269-277
    if (tets.dim() != 2 || tets.size(1) != 4) err("tets must have dimensions (num_tets, 4)");
    if (face_tets.dim() != 2 || face_tets.size(0) != faces.size(0) || face_tets.size(1) != 2) err("face_tets must have dimensions (num_faces, 2)");
    if (tet_faces.dim() != 2 || tet_faces.size(0) != tets.size(0) || tet_faces.size(1) != 4) err("tet_faces must have dimensions (num_tets, 4)");
}

// -> (color [B,3,H,W], depth [B,1,H,W], active f32 [B,H,W], pointBuffer, faceBuffer, binningBuffer, imgBuffer)
using TetFwdOut = std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor>;
TetFwdOut render_tets(const at::Tensor& background, const at::Tensor& verts, const at::Tensor& faces, const at::Tensor& verts_color,
                      const at::Tensor& faces_opacity, const at::Tensor& mv_mats, const at::Tensor& proj_mats,
                      const at::Tensor& inv_mv_mats, const at::Tensor& inv_proj_mats, const at::Tensor& verts_depth,
                      const at::Tensor& faces_intense, const at::Tensor& tets, const at::Tensor& face_tets, const at::Tensor& tet_faces,
                      int64_t image_height, int64_t image_width, int64_t ray_random_seed, std::pair<int, int> rows) {
    check_common(verts, faces, verts_color, faces_opacity, mv_mats, proj_mats, inv_mv_mats, inv_proj_mats, verts_depth, faces_intense, true);
    check_tets(faces, tets, face_tets, tet_faces);
    const c10::Device dev = hip_device_of(verts);
    c10::DeviceGuard guard(dev);
    Call call(dev, background, verts, faces, verts_color, faces_opacity, mv_mats, proj_mats, inv_mv_mats, inv_proj_mats, verts_depth,
              faces_intense, image_height, image_width, &tets, &face_tets, &tet_faces, ray_random_seed, rows);
    // k_tet_forward writes every pixel of the rendered rows (background where the march fails); zero-fill
    // (render.cu:287-290) is only needed when a band leaves rows untouched or nothing is launched
    const bool full = rows.first == 0 && rows.second == 0 && call.sc.P > 0 && call.sc.F > 0;
    const auto opt = f32_on(dev);
    auto img = [&](at::IntArrayRef shape) { return full ? at::empty(shape, opt) : at::zeros(shape, opt); };
    at::Tensor color = img({call.sc.B, NUM_CHANNELS, image_height, image_width});
    at::Tensor depth = img({call.sc.B, 1, image_height, image_width});
    at::Tensor active = img({call.sc.B, image_height, image_width});
    int rendered = 0;
    if (g_abi.tet_forward(&call.sc, mptr<float>(color), mptr<float>(depth), mptr<float>(active), &alloc_cb, &call.scratch, call.stream(), &rendered))
        raise_lib();
    return TetFwdOut(color, depth, active, call.scratch.get(DMR_BUF_POINT), call.scratch.get(DMR_BUF_FACE),
                          call.scratch.get(DMR_BUF_BINNING), call.scratch.get(DMR_BUF_IMAGE));
}

// -> (dL_dverts_color [P,3], dL_dfaces_opacity [F])
using TetBwdOut = std::tuple<at::Tensor, at::Tensor>;
TetBwdOut render_tets_backward(const at::Tensor& background, const at::Tensor& verts, const at::Tensor& faces, const at::Tensor& verts_color,
                               const at::Tensor& faces_opacity, const at::Tensor& mv_mats, const at::Tensor& proj_mats,
                               const at::Tensor& inv_mv_mats, const at::Tensor& inv_proj_mats, const at::Tensor& verts_depth,
                               const at::Tensor& faces_intense, const at::Tensor& tets, const at::Tensor& face_tets,
                               const at::Tensor& tet_faces, const at::Tensor& grad_color, const at::Tensor& grad_depth,
                               const at::Tensor& pointBuffer, const at::Tensor& faceBuffer, const at::Tensor& binningBuffer,
                               const at::Tensor& imageBuffer, std::pair<int, int> rows, const std::optional<at::Tensor>& flat_out) {
    const c10::Device dev = hip_device_of(verts);
    c10::DeviceGuard guard(dev);
    if (grad_color.dim() != 4) err("grad_color must have dimensions (B, 3, H, W)");
    const int64_t H = grad_color.size(2), W = grad_color.size(3);  // render.cu:371-372
    Call call(dev, background, verts, faces, verts_color, faces_opacity, mv_mats, proj_mats, inv_mv_mats, inv_proj_mats, verts_depth,
              faces_intense, H, W, &tets, &face_tets, &tet_faces, 0, rows);
    const at::Tensor gc = f32(grad_color, "grad_color"), gd = f32(grad_depth, "grad_depth");
    const int64_t P = call.sc.P, F = call.sc.F;
    at::Tensor g_vcolor, g_fop;
    if (!flat_out.has_value()) {
        g_vcolor = at::empty({P, 3}, f32_on(dev)); g_fop = at::empty({F}, f32_on(dev));
    } else {
        const at::Tensor& fo = *flat_out;
        if (fo.scalar_type() != at::kFloat || fo.device() != dev || !fo.is_contiguous() || fo.numel() != 3 * P + F)
            err("flat_out must be a contiguous float32 tensor of " + std::to_string(3 * P + F) + " elements on " + dev.str());
        const at::Tensor flat = fo.view({-1});
        g_vcolor = flat.narrow(0, 0, 3 * P).view({P, 3}); g_fop = flat.narrow(0, 3 * P, F);
    }
    const at::Tensor pb = pointBuffer.contiguous(), fb = faceBuffer.contiguous(), bb = binningBuffer.contiguous(), ib = imageBuffer.contiguous();
    if (g_abi.tet_backward(&call.sc, mptr<const float>(gc), mptr<const float>(gd), mptr<const void>(pb), mptr<const void>(fb),
                           mptr<const void>(bb), mptr<const void>(ib), mptr<float>(g_vcolor), mptr<float>(g_fop), &alloc_cb,
                           &call.scratch, call.stream()))
        raise_lib();
    return TetBwdOut(g_vcolor, g_fop);
}

// ---- extensions -----------------------------------------------------------------------------------------------------
// th.inverse of [B,4,4] float32 HIP tensors with one small library kernel each (dmr_invert_mats: adjugate in double
// precision).  The reference wrapper calls th.inverse twice per forward (dmesh_renderer/__init__.py:62-63: two batched
// LU factorisations, ~0.12 ms of small kernels on the GPU).  Returns contiguous tensors.
py::tuple invert_mats(const py::args& mats) {
    if (mats.size() == 0) return py::tuple();
    std::vector<at::Tensor> in;
    for (const auto& h : mats) in.push_back(h.cast<at::Tensor>());
    const c10::Device dev = hip_device_of(in[0]);
    c10::DeviceGuard guard(dev);
    void* st = reinterpret_cast<void*>(c10::hip::getCurrentHIPStream(dev.index()).stream());
    py::tuple res(in.size());
    for (size_t i = 0; i < in.size(); i++) {
        const at::Tensor& m = in[i];
        if (m.dim() != 3 || m.size(1) != 4 || m.size(2) != 4) err("matrices must have dimensions (B, 4, 4)");
        if (m.device() != dev) err("all matrices must be on " + dev.str());
        auto t = mat(m, "matrix");
        at::Tensor out = at::empty({t.first.size(0), 4, 4}, f32_on(dev));
        if (t.first.size(0) && g_abi.invert_mats(mptr<const float>(t.first), (int)t.first.size(0), t.second, mptr<float>(out), st)) raise_lib();
        res[i] = out;
    }
    return res;
}

// Parity/debug helper: copy one forward intermediate out of the scratch buffers (dmr_export).  `call_args` are the 11
// (tri) / 14 (tet) leading tensors of render_*.
at::Tensor export_item(const std::string& name, const py::sequence& call_args, bool is_tet, int64_t num_rendered,
                       const py::sequence& buffers, int64_t H, int64_t W, const py::object& dtype) {
    std::vector<at::Tensor> a;
    for (const auto& h : call_args) a.push_back(h.cast<at::Tensor>());
    if (a.size() < (is_tet ? 14u : 11u)) err("export: call_args must hold the leading tensors of render_*");
    const c10::Device dev = hip_device_of(a[1]);
    c10::DeviceGuard guard(dev);
    Call call(dev, a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], H, W, is_tet ? &a[11] : nullptr,
              is_tet ? &a[12] : nullptr, is_tet ? &a[13] : nullptr, 0, {0, 0});
    std::vector<at::Tensor> b;
    for (const auto& h : buffers) b.push_back(h.cast<at::Tensor>().contiguous());
    if (b.size() != 4) err("export: buffers must be the four scratch tensors");
    void* st = call.stream();
    const int64_t n = g_abi.export_item(&call.sc, is_tet ? 1 : 0, (int)num_rendered, name.c_str(), mptr<const void>(b[0]), mptr<const void>(b[1]),
                                        mptr<const void>(b[2]), mptr<const void>(b[3]), nullptr, 0, st);
    if (n < 0) raise_lib();
    at::Tensor out = at::empty({std::max<int64_t>(n, 1)}, at::TensorOptions().dtype(at::kByte).device(dev));
    g_abi.export_item(&call.sc, is_tet ? 1 : 0, (int)num_rendered, name.c_str(), mptr<const void>(b[0]), mptr<const void>(b[1]),
                      mptr<const void>(b[2]), mptr<const void>(b[3]), out.data_ptr(), n, st);
    return out.narrow(0, 0, n).view(torch::python::detail::py_object_to_dtype(dtype));
}

py::tuple profile_collect() {
    std::vector<double> ms(DMR_NUM_STAGES, 0.0);
    std::vector<int64_t> cnt(DMR_NUM_STAGES, 0);
    if (g_abi.profile_collect(ms.data(), cnt.data())) raise_lib();
    return py::make_tuple(ms, cnt);
}

}  // namespace

PYBIND11_MODULE(_C, m) {
    load_abi();  // import fails loudly when the HIP library is missing or mismatched
    m.doc() = "dmesh_renderer_amd._C: the reference's four-function binding surface (ext.cpp:6-11) over libdmesh_renderer_hip.so";
    const auto no_rows = std::pair<int, int>(0, 0);
    m.def("render_tris", &render_tris, py::arg("background"), py::arg("verts"), py::arg("faces"), py::arg("verts_color"),
          py::arg("faces_opacity"), py::arg("mv_mats"), py::arg("proj_mats"), py::arg("inv_mv_mats"), py::arg("inv_proj_mats"),
          py::arg("verts_depth"), py::arg("faces_intense"), py::arg("image_height"), py::arg("image_width"),
          py::arg("rows") = no_rows, py::arg("fill_outside") = true, py::call_guard<py::gil_scoped_release>());
    m.def("render_tris_backward", &render_tris_backward, py::arg("background"), py::arg("verts"), py::arg("faces"), py::arg("verts_color"),
          py::arg("faces_opacity"), py::arg("mv_mats"), py::arg("proj_mats"), py::arg("inv_mv_mats"), py::arg("inv_proj_mats"),
          py::arg("verts_depth"), py::arg("faces_intense"), py::arg("dL_dout_color"), py::arg("dL_dout_depth"), py::arg("R"),
          py::arg("pointBuffer"), py::arg("faceBuffer"), py::arg("binningBuffer"), py::arg("imageBuffer"),
          py::arg("rows") = no_rows, py::arg("flat_out") = py::none(), py::call_guard<py::gil_scoped_release>());
    m.def("render_tets", &render_tets, py::arg("background"), py::arg("verts"), py::arg("faces"), py::arg("verts_color"),
          py::arg("faces_opacity"), py::arg("mv_mats"), py::arg("proj_mats"), py::arg("inv_mv_mats"), py::arg("inv_proj_mats"),
          py::arg("verts_depth"), py::arg("faces_intense"), py::arg("tets"), py::arg("face_tets"), py::arg("tet_faces"),
          py::arg("image_height"), py::arg("image_width"), py::arg("ray_random_seed"), py::arg("rows") = no_rows,
          py::call_guard<py::gil_scoped_release>());
    m.def("render_tets_backward", &render_tets_backward, py::arg("background"), py::arg("verts"), py::arg("faces"), py::arg("verts_color"),
          py::arg("faces_opacity"), py::arg("mv_mats"), py::arg("proj_mats"), py::arg("inv_mv_mats"), py::arg("inv_proj_mats"),
          py::arg("verts_depth"), py::arg("faces_intense"), py::arg("tets"), py::arg("face_tets"), py::arg("tet_faces"),
          py::arg("grad_color"), py::arg("grad_depth"), py::arg("pointBuffer"), py::arg("faceBuffer"), py::arg("binningBuffer"),
          py::arg("imageBuffer"), py::arg("rows") = no_rows, py::arg("flat_out") = py::none(), py::call_guard<py::gil_scoped_release>());
    m.def("invert_mats", &invert_mats);
    m.def("export", &export_item, py::arg("name"), py::arg("call_args"), py::arg("is_tet"), py::arg("num_rendered"), py::arg("buffers"),
          py::arg("H"), py::arg("W"), py::arg("dtype"));
    // asynchronous calls (include/dmesh_renderer_amd.h, "Sizes only the device knows")
    m.def("set_async", [](bool on) { g_async.store(on ? 1 : 0); }, py::arg("on"),
          "Calls never wait for the device: num_rendered is the capacity used; check overflowed() after synchronising.");
    m.def("is_async", []() { return g_async.load() != 0; });
    m.def("redo_count", []() { return (unsigned long long)g_abi.redo_count(); },
          "Default calls that had to enqueue stages twice because a size estimate was too small (process-wide).");
    m.def("overflowed", [](int device, bool reset) { return g_abi.overflowed(device, reset ? 1 : 0) != 0; }, py::arg("device") = -1,
          py::arg("reset") = true, "True if an asynchronous / graph-captured call outgrew its buffers since the last reset.");
    // per-stage HIP-event timing (bench.py's roofline leg)
    m.def("profile_enable", [](uint32_t mask) { g_abi.profile_enable(mask); }, py::arg("mask"));
    m.def("profile_collect", &profile_collect, "-> (ms per stage, launches per stage), accumulated since the last call");
    m.def("stage_name", [](int i) { return std::string(g_abi.stage_name(i)); });
    m.def("last_error", []() { return std::string(g_abi.last_error()); });
    m.def("library_path", []() { return g_abi.path; });
    m.def("build_arch", []() { return std::string(g_abi.build_arch()); });
    m.attr("NUM_STAGES") = (int)DMR_NUM_STAGES;
    m.attr("ABI_VERSION") = (int)DMR_ABI_VERSION;
    m.attr("NUM_CHANNELS") = NUM_CHANNELS;
    m.attr("SUPPORTS_FLAT_OUT") = true;  // render_*_backward(flat_out=...), used by sharding.py
    m.attr("STAGE_PROJECT") = (int)DMR_STAGE_PROJECT; m.attr("STAGE_SETUP_FACES") = (int)DMR_STAGE_SETUP_FACES;
    m.attr("STAGE_SCAN") = (int)DMR_STAGE_SCAN; m.attr("STAGE_SCATTER") = (int)DMR_STAGE_SCATTER; m.attr("STAGE_SORT") = (int)DMR_STAGE_SORT;
    m.attr("STAGE_TRI_FORWARD") = (int)DMR_STAGE_TRI_FORWARD; m.attr("STAGE_TRI_BACKWARD") = (int)DMR_STAGE_TRI_BACKWARD;
    m.attr("STAGE_TRI_UNPACK") = (int)DMR_STAGE_TRI_UNPACK; m.attr("STAGE_TET_FIRST") = (int)DMR_STAGE_TET_FIRST;
    m.attr("STAGE_TET_FORWARD") = (int)DMR_STAGE_TET_FORWARD; m.attr("STAGE_TET_BACKWARD") = (int)DMR_STAGE_TET_BACKWARD;
    m.attr("STAGE_TRI_BACKWARD_HITS") = (int)DMR_STAGE_TRI_BACKWARD_HITS;
}
