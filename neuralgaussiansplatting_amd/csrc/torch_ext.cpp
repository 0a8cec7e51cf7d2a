// The compiled torch extension behind diff_gaussian_rasterization._C: the three functions the reference's extension
// exports (submodules/diff-gaussian-rasterization/ext.cpp:15-19) with the signatures of rasterize_points.h:18-67
// and the marshalling of rasterize_points.cu:35-217 -- tensors in, raw pointers and sizes to the C ABI of
// libngs_rasterizer.so, state buffers grown through tensor.resize_() callbacks -- so that no Python runs between the
// kernels of a call.  Host-only code: the kernels live in the library, which is bound at run time (bind()).
#include <torch/extension.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/core/DeviceGuard.h>
#include <dlfcn.h>

#include <map>
#include <mutex>
#include <string>
#include <tuple>

#include "../../include/ngs_rasterizer.h"

namespace {

struct Api {
    void* handle = nullptr;
    decltype(&ngs_rasterize_forward_presized) forward = nullptr;
    decltype(&ngs_rasterize_backward) backward = nullptr;
    decltype(&ngs_mark_visible) mark_visible = nullptr;
    decltype(&ngs_last_error) last_error = nullptr;
} api;

// Instance count of the previous call per (device, W, H): consecutive calls of a training or rendering loop produce
// nearly the same number of instances, so the next call sizes its binning buffer from it up front and the library
// queues every stage without waiting for its own count (ngs_rasterize_forward_presized).
std::mutex hint_mutex;
std::map<std::tuple<int, int, int, int>, std::pair<long long, long long>> instance_hint;
long long forced_capacity = -1;  // tests: >= 0 replaces the estimate (0 = wait for the count, as the reference does)

template <typename F>
void resolve(F& fn, const char* name, const std::string& path)
{
    fn = reinterpret_cast<F>(dlsym(api.handle, name));
    TORCH_CHECK(fn != nullptr, "symbol ", name, " not found in ", path);
}

void bind(const std::string& path)
{
    void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    TORCH_CHECK(h != nullptr, "libngs_rasterizer.so is not built or cannot be loaded (", path, "): ", dlerror());
    api.handle = h;
    resolve(api.forward, "ngs_rasterize_forward_presized", path);
    resolve(api.backward, "ngs_rasterize_backward", path);
    resolve(api.mark_visible, "ngs_mark_visible", path);
    resolve(api.last_error, "ngs_last_error", path);
}

void require_bound() { TORCH_CHECK(api.handle != nullptr, "the rasterizer library has not been bound (call bind(path) first)"); }

void check(int rc) { TORCH_CHECK(rc == 0, api.last_error()); }

// == resizeFunctional (rasterize_points.cu:27-33): the library asks, the tensor grows
char* grow(void* ctx, size_t bytes)
{
    auto* t = static_cast<torch::Tensor*>(ctx);
    if ((int64_t)bytes > t->numel()) t->resize_({(int64_t)bytes});
    return reinterpret_cast<char*>(t->data_ptr());
}

void require_gpu(const torch::Tensor& t)
{
    TORCH_CHECK(t.is_cuda(), "the rasterizer runs on the GPU only: means3D is on ", t.device().str());
}

// A contiguous float32 tensor on `dev`, or an empty one (how the reference signals an absent optional input).
torch::Tensor f32(const torch::Tensor& t, const torch::Device& dev)
{
    if (t.numel() == 0) return t;
    TORCH_CHECK(t.device() == dev, "all rasterizer inputs must live on ", dev.str(), " (got ", t.device().str(), ")");
    TORCH_CHECK(t.scalar_type() == torch::kFloat32, "rasterizer inputs must be float32 (got ", t.scalar_type(), ")");
    return t.contiguous();
}

// torch on ROCm keeps its "cuda" device type and streams over HIP; this is the hipStream_t of torch.cuda.current_stream()
hipStream_t current_stream(const torch::Device& dev)
{
    return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream();
}

const float* fptr(const torch::Tensor& t) { return t.numel() ? t.data_ptr<float>() : nullptr; }
float* fptr_mut(torch::Tensor& t) { return t.numel() ? t.data_ptr<float>() : nullptr; }

// rasterize_points.h:18-37 / rasterize_points.cu:35-116
std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor> rasterize_gaussians(
    const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors,
    const torch::Tensor& opacity, const torch::Tensor& scales, const torch::Tensor& rotations,
    const float scale_modifier, const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
    const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy, const int image_height,
    const int image_width, const torch::Tensor& sh, const int degree, const torch::Tensor& campos,
    const bool prefiltered, const bool debug)
{
    TORCH_CHECK(means3D.dim() == 2 && means3D.size(1) == 3, "means3D must have dimensions (num_points, 3)");
    require_gpu(means3D);
    require_bound();
    const torch::Device dev = means3D.device();
    const int P = (int)means3D.size(0), H = image_height, W = image_width;
    const torch::Tensor means = f32(means3D, dev), col = f32(colors, dev), op = f32(opacity, dev), sc = f32(scales, dev),
                        rot = f32(rotations, dev), cov = f32(cov3D_precomp, dev), shs = f32(sh, dev),
                        bg = f32(background, dev), view = f32(viewmatrix, dev), proj = f32(projmatrix, dev),
                        cam = f32(campos, dev);
    const auto bytes = torch::TensorOptions().dtype(torch::kUInt8).device(dev);
    torch::Tensor geom = torch::empty({0}, bytes), binning = torch::empty({0}, bytes), img = torch::empty({0}, bytes);
    int rendered = 0;
    if (P == 0) {
        return std::make_tuple(rendered, torch::zeros({3, H, W}, means.options()),
                               torch::zeros({0}, means.options().dtype(torch::kInt32)), geom, binning, img);
    }
    long long capacity = 0;
    // (two scenes of one resolution that alternate in a process keep their own estimates unless their sizes are close)
    const auto key = std::make_tuple((int)dev.index(), W, H, 31 - __builtin_clz((unsigned)std::max(P, 1)));
    {
        std::lock_guard<std::mutex> lock(hint_mutex);
        const auto it = instance_hint.find(key);
        if (it != instance_hint.end()) {
            const long long hint_P = it->second.first, hint_R = it->second.second;
            if (hint_R > 0 && 2 * hint_P >= P && hint_P <= 2LL * P) {  // a scene of about the same size
                const long long est = hint_R * P / hint_P;
                capacity = std::min<long long>(est + est / 8 + 4096, 0x7fffffffLL);
            }
        }
        if (forced_capacity >= 0) capacity = forced_capacity;
    }
    torch::Tensor out_color = torch::empty({3, H, W}, means.options());
    torch::Tensor radii = torch::empty({P}, means.options().dtype(torch::kInt32));
    const int M = shs.numel() ? (int)shs.size(1) : 0;
    {
        const c10::DeviceGuard guard(dev);
        check(api.forward(grow, &geom, grow, &binning, grow, &img, P, degree, M, fptr(bg), W, H, fptr(means), fptr(shs),
                          fptr(col), fptr(op), fptr(sc), scale_modifier, fptr(rot), fptr(cov), fptr(view), fptr(proj),
                          fptr(cam), tan_fovx, tan_fovy, prefiltered ? 1 : 0, out_color.data_ptr<float>(),
                          radii.data_ptr<int>(), debug ? 1 : 0, current_stream(dev),
                          &rendered, capacity));
    }
    {
        std::lock_guard<std::mutex> lock(hint_mutex);
        if (instance_hint.size() > 256) instance_hint.clear();
        instance_hint[key] = std::make_pair((long long)P, (long long)rendered);
    }
    return std::make_tuple(rendered, out_color, radii, geom, binning, img);
}

// rasterize_points.h:39-61 / rasterize_points.cu:118-196.  Every output is written in full by the library (zeros
// for Gaussians that were not rendered), so nothing is zero-filled here.
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
           torch::Tensor>
rasterize_gaussians_backward(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& radii,
                             const torch::Tensor& colors, const torch::Tensor& scales, const torch::Tensor& rotations,
                             const float scale_modifier, const torch::Tensor& cov3D_precomp,
                             const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix, const float tan_fovx,
                             const float tan_fovy, const torch::Tensor& dL_dout_color, const torch::Tensor& sh,
                             const int degree, const torch::Tensor& campos, const torch::Tensor& geomBuffer, const int R,
                             const torch::Tensor& binningBuffer, const torch::Tensor& imageBuffer, const bool debug)
{
    require_gpu(means3D);
    require_bound();
    const torch::Device dev = means3D.device();
    const int P = (int)means3D.size(0);
    const int H = (int)dL_dout_color.size(1), W = (int)dL_dout_color.size(2);
    const torch::Tensor means = f32(means3D, dev), col = f32(colors, dev), sc = f32(scales, dev), rot = f32(rotations, dev),
                        cov = f32(cov3D_precomp, dev), shs = f32(sh, dev), dpix = f32(dL_dout_color, dev),
                        bg = f32(background, dev), view = f32(viewmatrix, dev), proj = f32(projmatrix, dev),
                        cam = f32(campos, dev);
    const int M = shs.numel() ? (int)shs.size(1) : 0;
    const auto make = [&](std::initializer_list<int64_t> shape) {
        return P != 0 ? torch::empty(shape, means.options()) : torch::zeros(shape, means.options());
    };
    torch::Tensor dL_dmeans3D = make({P, 3}), dL_dmeans2D = make({P, 3}), dL_dcolors = make({P, 3}),
                  dL_dopacity = make({P, 1}), dL_dcov3D = make({P, 6}), dL_dsh = make({P, M, 3}),
                  dL_dscales = make({P, 3}), dL_drotations = make({P, 4});
    if (P != 0) {
        const torch::Tensor rad = radii.contiguous();
        TORCH_CHECK(rad.scalar_type() == torch::kInt32 && rad.device() == dev, "radii must be int32 on ", dev.str());
        const c10::DeviceGuard guard(dev);
        check(api.backward(P, degree, M, R, fptr(bg), W, H, fptr(means), fptr(shs), fptr(col), fptr(sc), scale_modifier,
                           fptr(rot), fptr(cov), fptr(view), fptr(proj), fptr(cam), tan_fovx, tan_fovy,
                           rad.data_ptr<int>(), reinterpret_cast<char*>(geomBuffer.data_ptr()),
                           reinterpret_cast<char*>(binningBuffer.data_ptr()),
                           reinterpret_cast<char*>(imageBuffer.data_ptr()), fptr(dpix), fptr_mut(dL_dmeans2D), nullptr,
                           fptr_mut(dL_dopacity), fptr_mut(dL_dcolors), fptr_mut(dL_dmeans3D), fptr_mut(dL_dcov3D),
                           fptr_mut(dL_dsh), fptr_mut(dL_dscales), fptr_mut(dL_drotations), debug ? 1 : 0,
                           current_stream(dev)));
    }
    return std::make_tuple(dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales,
                           dL_drotations);
}

// rasterize_points.h:63-67 / rasterize_points.cu:198-217
torch::Tensor mark_visible(torch::Tensor& means3D, torch::Tensor& viewmatrix, torch::Tensor& projmatrix)
{
    require_gpu(means3D);
    require_bound();
    const torch::Device dev = means3D.device();
    const int P = (int)means3D.size(0);
    torch::Tensor present = torch::zeros({P}, means3D.options().dtype(torch::kBool));
    if (P != 0) {
        const torch::Tensor means = f32(means3D, dev), view = f32(viewmatrix, dev), proj = f32(projmatrix, dev);
        const c10::DeviceGuard guard(dev);
        check(api.mark_visible(P, fptr(means), fptr(view), fptr(proj),
                               reinterpret_cast<unsigned char*>(present.data_ptr<bool>()),
                               current_stream(dev)));
    }
    return present;
}

void set_forced_binning_capacity(long long capacity)
{
    std::lock_guard<std::mutex> lock(hint_mutex);
    forced_capacity = capacity;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    namespace py = pybind11;
    m.def("rasterize_gaussians", &rasterize_gaussians, py::arg("background"), py::arg("means3D"), py::arg("colors"),
          py::arg("opacity"), py::arg("scales"), py::arg("rotations"), py::arg("scale_modifier"),
          py::arg("cov3D_precomp"), py::arg("viewmatrix"), py::arg("projmatrix"), py::arg("tan_fovx"),
          py::arg("tan_fovy"), py::arg("image_height"), py::arg("image_width"), py::arg("sh"), py::arg("degree"),
          py::arg("campos"), py::arg("prefiltered"), py::arg("debug"));
    m.def("rasterize_gaussians_backward", &rasterize_gaussians_backward, py::arg("background"), py::arg("means3D"),
          py::arg("radii"), py::arg("colors"), py::arg("scales"), py::arg("rotations"), py::arg("scale_modifier"),
          py::arg("cov3D_precomp"), py::arg("viewmatrix"), py::arg("projmatrix"), py::arg("tan_fovx"),
          py::arg("tan_fovy"), py::arg("dL_dout_color"), py::arg("sh"), py::arg("degree"), py::arg("campos"),
          py::arg("geomBuffer"), py::arg("R"), py::arg("binningBuffer"), py::arg("imageBuffer"), py::arg("debug"));
    m.def("mark_visible", &mark_visible, py::arg("means3D"), py::arg("viewmatrix"), py::arg("projmatrix"));
    m.def("bind", &bind, py::arg("library_path"),
          "Load libngs_rasterizer.so from this path and resolve the entry points the three functions call.");
    m.def("set_forced_binning_capacity", &set_forced_binning_capacity, py::arg("capacity"),
          "Tests: a capacity >= 0 replaces the estimate of the binning buffer's size (0 = wait for the count); -1 "
          "restores the estimate.");
}
