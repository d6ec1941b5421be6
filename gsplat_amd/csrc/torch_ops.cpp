// Compiled op bodies: TORCH_LIBRARY_IMPL(gsplat, CUDA, ...) for the hot stage ops, calling the C-ABI of libgsplat_amd.so
// (INTEGRATION.md route B). Host C++ only - no device code here; the kernels stay behind include/gsplat_amd.h.
//
// What an op body does is what the reference's host functions do around their kernels (shape checks, output allocation
// from the inputs' options, current stream, device guard, error translation):
//   projection_ewa_3dgs_fused{,_bwd}   gsplat/cuda/csrc/Projection.cpp:366-440, 579-694
//   spherical_harmonics{,_bwd}         gsplat/cuda/csrc/SphericalHarmonics.cpp (ext.cpp:994-1014)
//   intersect_tile, intersect_offset   gsplat/cuda/csrc/Intersect.cpp:170-329
//   rasterize_to_pixels_3dgs{,_bwd}    gsplat/cuda/csrc/Rasterization.cpp:275-365, 484-587
// The Python bodies in gsplat_amd/_ops.py remain the implementation of every other op and of the private fast paths of
// gsplat_amd/rendering.py; _ops.py skips registering its own body for an op listed by gsx_torch_compiled_ops().
// Schemas are defined by _ops.py (verbatim from ext.cpp); an IMPL block may be loaded before or after the definitions.
#include <chrono>
#include <cstdlib>
#include <thread>
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/core/DeviceGuard.h>
#include <torch/library.h>

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <mutex>
#include <set>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/gsplat_amd.h"

namespace gsplat_amd {
namespace {

using at::Tensor;
using OptTensor = std::optional<Tensor>;

// ---- plumbing ----------------------------------------------------------------------------------------------------------
struct Launch { // device guard + the tensor's device's current stream (the reference's DEVICE_GUARD + getCurrentCUDAStream)
    c10::DeviceGuard guard;
    void *stream;
    explicit Launch(const Tensor &t)
        : guard(t.device()), stream((void *)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream())
    {
        TORCH_CHECK(t.is_cuda(), "gsplat_amd kernels only run on a ROCm device (got a CPU tensor); there is no CPU fallback");
    }
};

// ---- optional timing of the C-ABI calls made from here (what gsplat_amd._cabi.profile_begin / profile_end do for the Python
// bodies): a HIP event pair on the launch stream around every call whose entry point is selected. Off by default. ---------------
struct ProfRecord {
    std::string name;
    hipEvent_t a, b;
};
std::mutex g_prof_mutex;
bool g_prof_on = false;
std::set<std::string> g_prof_only; // empty = every entry point
std::vector<ProfRecord> g_prof;

struct Timed { // RAII: start event now, stop event at scope exit (after the C-ABI call has enqueued its kernels)
    hipEvent_t a = nullptr, b = nullptr;
    void *stream;
    const char *name;
    Timed(const char *fn, void *s) : stream(s), name(fn)
    {
        if (!g_prof_on) return;
        std::lock_guard<std::mutex> lock(g_prof_mutex);
        if (!g_prof_on || (!g_prof_only.empty() && !g_prof_only.count(fn))) return;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
        (void)hipEventRecord(a, (hipStream_t)stream);
    }
    ~Timed()
    {
        if (!a) return;
        (void)hipEventRecord(b, (hipStream_t)stream);
        std::lock_guard<std::mutex> lock(g_prof_mutex);
        g_prof.push_back({name, a, b});
    }
};

void check(int rc, const char *fn)
{
    if (rc == 0) return;
    const std::string msg = gsx_last_error();
    TORCH_CHECK(rc != -1, fn, ": ", msg); // GSX_ERR_ARG: an argument check, RuntimeError like the reference's TORCH_CHECK
    TORCH_CHECK(false, fn, " failed (code ", rc, "): ", msg);
}

Tensor contig(const Tensor &t) { return t.is_contiguous() ? t : t.contiguous(); }
OptTensor contig(const OptTensor &t) { return t.has_value() && t->defined() ? OptTensor(contig(*t)) : OptTensor(); }
bool has(const OptTensor &t) { return t.has_value() && t->defined(); }

void want_f32(const Tensor &t, const char *name)
{
    TORCH_CHECK(t.scalar_type() == at::kFloat, "gsplat_amd: ", name, " must be float32 (got ", t.scalar_type(),
                     "); the gfx950 kernels compute in fp32");
}
void want_f32(const OptTensor &t, const char *name)
{
    if (has(t)) want_f32(*t, name);
}

// The reference dispatches the projection ops over float AND double (AT_DISPATCH_FLOATING_TYPES, ProjectionEWA3DGSFused.cu:260,
// 686; ProjectionEWA3DGSPacked.cu:344, 733). In its double instantiation only the MEMORY type is double: the kernels load every
// value into glm float vectors / matrices (Common.h:65-70: vec3 = glm::vec<3, float>, mat3 = glm::mat<3, 3, float>), compute in
// float and widen the results on store. Same here: double tensors are narrowed, the fp32 kernels run, float outputs are widened.
bool is_f64(const Tensor &t) { return t.defined() && t.scalar_type() == at::kDouble; }
Tensor narrow32(const Tensor &t) { return t.defined() && t.scalar_type() == at::kDouble ? t.to(at::kFloat) : t; }
OptTensor narrow32(const OptTensor &t) { return has(t) ? OptTensor(narrow32(*t)) : t; }
Tensor widen64(const Tensor &t) { return t.defined() && t.scalar_type() == at::kFloat ? t.to(at::kDouble) : t; }
OptTensor widen64(const OptTensor &t) { return has(t) ? OptTensor(widen64(*t)) : t; }

template <class T> const T *cp(const Tensor &t) { return t.defined() && t.numel() ? t.const_data_ptr<T>() : nullptr; }
template <class T> const T *cp(const OptTensor &t) { return has(t) ? cp<T>(*t) : nullptr; }
template <class T> T *mp(Tensor &t) { return t.defined() && t.numel() ? t.mutable_data_ptr<T>() : nullptr; }
const float *fp(const Tensor &t) { return cp<float>(t); }
const float *fp(const OptTensor &t) { return cp<float>(t); }

// (tensor, row stride in floats) of a [..., width] float tensor whose rows are `width` contiguous floats at a uniform stride -
// e.g. a column view of the array-of-structures gradient rows returned by rasterize_to_pixels_3dgs_bwd - else a copy
std::pair<Tensor, uint32_t> row_view(const Tensor &t, int64_t width)
{
    if (t.is_contiguous()) return {t, (uint32_t)width};
    if (t.dim() >= 2 && t.size(-1) == width && t.stride(-1) == 1) {
        const int64_t rs = t.stride(-2);
        bool uniform = rs >= width;
        int64_t expect = rs * t.size(-2);
        for (int64_t d = t.dim() - 3; d >= 0 && uniform; --d) {
            if (t.size(d) != 1 && t.stride(d) != expect) uniform = false;
            expect *= t.size(d);
        }
        if (uniform) return {t, (uint32_t)rs};
    }
    return {t.contiguous(), (uint32_t)width};
}
std::pair<Tensor, uint32_t> row_view_1(const Tensor &t) // [...] scalars per row
{
    return {t.contiguous(), 1u};
}

int64_t prod(c10::IntArrayRef dims)
{
    int64_t p = 1;
    for (auto d : dims) p *= d;
    return p;
}

uint32_t bits_for_count(int64_t count) // MathUtils.h:25-35
{
    uint32_t b = 0;
    if (count <= 1) return 0;
    uint64_t v = (uint64_t)count - 1;
    while (v) {
        ++b;
        v >>= 1;
    }
    return b;
}

// ---- projection (dense) ------------------------------------------------------------------------------------------------
std::tuple<Tensor, Tensor, Tensor, Tensor, OptTensor>
projection_ewa_3dgs_fused(const Tensor &means_, const OptTensor &covars_, const OptTensor &quats_, const OptTensor &scales_,
                          const OptTensor &opacities_, const Tensor &viewmats_, const Tensor &Ks_, int64_t width,
                          int64_t height, double eps2d, double near_plane, double far_plane, double radius_clip,
                          bool calc_compensations, int64_t camera_model)
{
    if (is_f64(means_)) { // the double instantiation: double in memory, float arithmetic (see narrow32)
        auto [radii, m2, dep, con, comp] = projection_ewa_3dgs_fused(
            narrow32(means_), narrow32(covars_), narrow32(quats_), narrow32(scales_), narrow32(opacities_), narrow32(viewmats_),
            narrow32(Ks_), width, height, eps2d, near_plane, far_plane, radius_clip, calc_compensations, camera_model);
        return {radii, widen64(m2), widen64(dep), widen64(con), widen64(comp)};
    }
    want_f32(means_, "means"); want_f32(covars_, "covars"); want_f32(quats_, "quats"); want_f32(scales_, "scales");
    want_f32(viewmats_, "viewmats"); want_f32(Ks_, "Ks");
    TORCH_CHECK(has(covars_) || (has(quats_) && has(scales_)), "projection: either covars or (quats, scales) must be given");
    Launch L(means_);
    const Tensor means = contig(means_), viewmats = contig(viewmats_), Ks = contig(Ks_);
    const OptTensor covars = contig(covars_), opac = contig(opacities_);
    const OptTensor quats = has(covars) ? OptTensor() : contig(quats_), scales = has(covars) ? OptTensor() : contig(scales_);
    auto batch = means.sizes().slice(0, means.dim() - 2);
    const int64_t B = prod(batch), C = viewmats.size(-3), N = means.size(-2);
    std::vector<int64_t> shape(batch.begin(), batch.end());
    shape.push_back(C); shape.push_back(N);
    auto with = [&](int64_t last) { auto s = shape; s.push_back(last); return s; };
    Tensor radii = at::empty(with(2), means.options().dtype(at::kInt));
    Tensor means2d = at::empty(with(2), means.options()), depths = at::empty(shape, means.options());
    Tensor conics = at::empty(with(3), means.options());
    OptTensor comps;
    if (calc_compensations) comps = at::empty(shape, means.options());
    { Timed timed_("gsx_project_ewa_fwd", L.stream); check(gsx_project_ewa_fwd(fp(means), fp(covars), fp(quats), fp(scales), fp(opac), fp(viewmats), fp(Ks), (uint32_t)B,
                              (uint32_t)C, (uint32_t)N, (uint32_t)width, (uint32_t)height, (float)eps2d, (float)near_plane,
                              (float)far_plane, (float)radius_clip, (int)camera_model, mp<int32_t>(radii), mp<float>(means2d),
                              mp<float>(depths), mp<float>(conics), comps ? mp<float>(*comps) : nullptr, L.stream),
          "gsx_project_ewa_fwd"); }
    return {radii, means2d, depths, conics, comps};
}

std::tuple<Tensor, OptTensor, OptTensor, OptTensor, OptTensor>
projection_ewa_3dgs_fused_bwd(const Tensor &means_, const OptTensor &covars_, const OptTensor &quats_, const OptTensor &scales_,
                              const Tensor &viewmats_, const Tensor &Ks_, int64_t width, int64_t height, double eps2d,
                              int64_t camera_model, const Tensor &radii, const Tensor &conics, const OptTensor &compensations,
                              const Tensor &v_means2d_, const Tensor &v_depths_, const Tensor &v_conics_,
                              const OptTensor &v_compensations, bool viewmats_requires_grad)
{
    if (is_f64(means_)) { // the double instantiation: double in memory, float arithmetic (see narrow32)
        auto [vm, vc, vq, vs, vv] = projection_ewa_3dgs_fused_bwd(
            narrow32(means_), narrow32(covars_), narrow32(quats_), narrow32(scales_), narrow32(viewmats_), narrow32(Ks_), width,
            height, eps2d, camera_model, radii, narrow32(conics), narrow32(compensations), narrow32(v_means2d_),
            narrow32(v_depths_), narrow32(v_conics_), narrow32(v_compensations), viewmats_requires_grad);
        return {widen64(vm), widen64(vc), widen64(vq), widen64(vs), widen64(vv)};
    }
    Launch L(means_);
    const Tensor means = contig(means_), viewmats = contig(viewmats_), Ks = contig(Ks_);
    const OptTensor covars = contig(covars_);
    const OptTensor quats = has(covars) ? OptTensor() : contig(quats_), scales = has(covars) ? OptTensor() : contig(scales_);
    auto batch = means.sizes().slice(0, means.dim() - 2);
    const int64_t B = prod(batch), C = viewmats.size(-3), N = means.size(-2);
    Tensor v_means = at::empty_like(means);
    OptTensor v_covars, v_quats, v_scales, v_viewmats;
    if (has(covars)) v_covars = at::empty_like(*covars);
    else { v_quats = at::empty_like(*quats); v_scales = at::empty_like(*scales); }
    if (viewmats_requires_grad) v_viewmats = at::zeros_like(viewmats);
    auto [vm2, m2s] = row_view(v_means2d_, 2);
    auto [vcn, cns] = row_view(v_conics_, 3);
    const Tensor vdep = v_depths_.defined() ? contig(v_depths_) : Tensor();
    const Tensor rad = contig(radii), con = contig(conics);
    const OptTensor comp = contig(compensations), vcomp = contig(v_compensations);
    { Timed timed_("gsx_project_ewa_bwd", L.stream); check(gsx_project_ewa_bwd(fp(means), fp(covars), fp(quats), fp(scales), fp(viewmats), fp(Ks), (uint32_t)B, (uint32_t)C,
                              (uint32_t)N, (uint32_t)width, (uint32_t)height, (float)eps2d, (int)camera_model,
                              cp<int32_t>(rad), fp(con), fp(comp), vm2.const_data_ptr<float>(), m2s, fp(vdep),
                              vcn.const_data_ptr<float>(), cns, fp(vcomp), mp<float>(v_means),
                              v_covars ? mp<float>(*v_covars) : nullptr, v_quats ? mp<float>(*v_quats) : nullptr,
                              v_scales ? mp<float>(*v_scales) : nullptr, v_viewmats ? mp<float>(*v_viewmats) : nullptr,
                              L.stream),
          "gsx_project_ewa_bwd"); }
    return {v_means, v_covars, v_quats, v_scales, v_viewmats};
}

// ---- projection (packed rows) ----------------------------------------------------------------------------------------------
// Two passes (count -> scan -> write) and ONE host round trip for the exact number of rows (Projection.cpp:928-941). When the
// upper bound (every (image, Gaussian) pair visible) is small next to the model, the write pass is enqueued into row buffers
// of that size BEFORE the host learns nnz - it only needs the device-side offsets - and the first nnz rows are handed out:
// the GPU does not idle through the round trip, the allocations and the launch. Larger scenes allocate exact lengths
// (saving that memory is what packed rows are for).
constexpr int64_t kPackedRowBytes = 64, kPackedPreallocLimit = 1ll << 30;

std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, OptTensor>
projection_ewa_3dgs_packed(const Tensor &means_, const OptTensor &covars_, const OptTensor &quats_, const OptTensor &scales_,
                           const OptTensor &opacities_, const Tensor &viewmats_, const Tensor &Ks_, int64_t width,
                           int64_t height, double eps2d, double near_plane, double far_plane, double radius_clip,
                           bool sparse_grad, bool calc_compensations, int64_t camera_model)
{
    if (is_f64(means_)) { // the double instantiation: double in memory, float arithmetic (see narrow32)
        auto [bi, ci, gi, ip, radii, m2, dep, con, comp] = projection_ewa_3dgs_packed(
            narrow32(means_), narrow32(covars_), narrow32(quats_), narrow32(scales_), narrow32(opacities_), narrow32(viewmats_),
            narrow32(Ks_), width, height, eps2d, near_plane, far_plane, radius_clip, sparse_grad, calc_compensations, camera_model);
        return {bi, ci, gi, ip, radii, widen64(m2), widen64(dep), widen64(con), widen64(comp)};
    }
    (void)sparse_grad;
    want_f32(means_, "means"); want_f32(covars_, "covars"); want_f32(quats_, "quats"); want_f32(scales_, "scales");
    want_f32(viewmats_, "viewmats"); want_f32(Ks_, "Ks");
    TORCH_CHECK(has(covars_) || (has(quats_) && has(scales_)), "projection: either covars or (quats, scales) must be given");
    Launch L(means_);
    const Tensor means = contig(means_), viewmats = contig(viewmats_), Ks = contig(Ks_);
    const OptTensor covars = contig(covars_), opac = contig(opacities_);
    const OptTensor quats = has(covars) ? OptTensor() : contig(quats_), scales = has(covars) ? OptTensor() : contig(scales_);
    const int64_t B = prod(means.sizes().slice(0, means.dim() - 2)), C = viewmats.size(-3), N = means.size(-2);
    const int64_t total = B * C * N;
    const auto f32 = means.options(), i32 = means.options().dtype(at::kInt), i64 = means.options().dtype(at::kLong);
    auto outputs = [&](int64_t rows) {
        return std::make_tuple(at::empty({rows}, i64), at::empty({rows}, i64), at::empty({rows}, i64), at::zeros({B * C + 1}, i32),
                               at::empty({rows, 2}, i32), at::empty({rows, 2}, f32), at::empty({rows}, f32),
                               at::empty({rows, 3}, f32), calc_compensations ? OptTensor(at::empty({rows}, f32)) : OptTensor());
    };
    if (total == 0) return outputs(0);
    auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(means.device().index());
    // rows are placed from BLOCK counts (csrc/projection.hip: PackedBlocks): one int32 per 256 pairs, scanned by one workgroup
    // that stores the row count straight into the pinned word below - no per-pair flags, no cumsum tensor, no copy kernel
    const int64_t n_blocks = gsx_project_packed_blocks(total);
    Tensor blocks = at::empty({2, n_blocks}, i32);
    Tensor host_nnz = at::empty({1}, at::TensorOptions().dtype(at::kLong).pinned_memory(true));
    volatile int64_t *nnz_slot = host_nnz.mutable_data_ptr<int64_t>();
    *nnz_slot = -1; // sentinel: the scan kernel overwrites it
    { Timed timed_("gsx_project_ewa_packed_count", L.stream); check(gsx_project_ewa_packed_count_blocks(fp(means), fp(covars), fp(quats), fp(scales), fp(opac), fp(viewmats), fp(Ks), (uint32_t)B,
                                       (uint32_t)C, (uint32_t)N, (uint32_t)width, (uint32_t)height, (float)eps2d,
                                       (float)near_plane, (float)far_plane, (float)radius_clip, (int)camera_model,
                                       calc_compensations ? 1 : 0, mp<int32_t>(blocks), mp<int32_t>(blocks) + n_blocks, nullptr,
                                       host_nnz.mutable_data_ptr<int64_t>(), L.stream),
          "gsx_project_ewa_packed_count_blocks"); }
    // The row count is on the host once the SCAN has run - the write kernel enqueued after it does not have to finish first.
    // Polling the pinned word instead of synchronising the stream lets the caller slice the outputs and enqueue the next
    // kernels while the write kernel still runs (everything stays stream-ordered behind it).
    auto wait_nnz = [&]() -> int64_t {
        const auto t0 = std::chrono::steady_clock::now();
        for (uint64_t spin = 0;; ++spin) {
            const int64_t v = *nnz_slot;
            if (v >= 0 && v == *nnz_slot) return v; // two equal reads: a value caught half-written cannot pass
            if ((spin & 63u) == 63u) std::this_thread::yield();
            if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) break;
        }
        stream.synchronize(); // never seen; keeps the function correct if the copy is not host-visible before the stream drains
        return *nnz_slot;
    };
    auto write = [&](int64_t rows, decltype(outputs(0)) &o) {
        auto &[bi, ci, gi, indptr, radii, m2, dep, con, comp] = o;
        (void)rows;
        { Timed timed_("gsx_project_ewa_packed_write", L.stream); check(gsx_project_ewa_packed_write_blocks(fp(means), fp(covars), fp(quats), fp(scales), fp(opac), fp(viewmats), fp(Ks), (uint32_t)B,
                                           (uint32_t)C, (uint32_t)N, (uint32_t)width, (uint32_t)height, (float)eps2d,
                                           (float)near_plane, (float)far_plane, (float)radius_clip, (int)camera_model,
                                           cp<int32_t>(blocks) + n_blocks, mp<int64_t>(bi), mp<int64_t>(ci), mp<int64_t>(gi),
                                           mp<int32_t>(indptr), mp<int32_t>(radii), mp<float>(m2), mp<float>(dep), mp<float>(con),
                                           comp ? mp<float>(*comp) : nullptr, L.stream),
              "gsx_project_ewa_packed_write_blocks"); }
    };
    if (total * kPackedRowBytes <= kPackedPreallocLimit) {
        auto o = outputs(total);
        write(total, o);
        const int64_t nnz = wait_nnz(); // host round trip: exact-length COO outputs
        auto &[bi, ci, gi, indptr, radii, m2, dep, con, comp] = o;
        // a view pins the whole upper-bound buffer for as long as the step (and its autograd graph) holds the rows: copy the
        // heads out and let the big buffers go only when that is a real amount of memory (> 256 MiB) - the seven copies cost
        // 32 us of kernels and as much host time, which left the GPU idle behind the write pass (profiles/r08_ab.md #28)
        const bool compact = (total - nnz) * kPackedRowBytes > (int64_t(1) << 28);
        auto head = [&](const Tensor &t) { return compact ? t.narrow(0, 0, nnz).clone() : t.narrow(0, 0, nnz); };
        return {head(bi), head(ci), head(gi), indptr, head(radii), head(m2), head(dep), head(con),
                comp ? OptTensor(head(*comp)) : OptTensor()};
    }
    const int64_t nnz = wait_nnz();
    auto o = outputs(nnz);
    if (nnz > 0 || true) write(nnz, o);
    return o;
}

// ---- spherical harmonics ------------------------------------------------------------------------------------------------
struct ShDims {
    bool packed;
    int64_t B, C, N, K, D;
};
ShDims sh_dims(const Tensor &means, const Tensor &viewmats, const Tensor &coeffs, const OptTensor &gaussian_ids)
{
    ShDims d;
    d.packed = has(gaussian_ids);
    TORCH_CHECK(coeffs.dim() == 3, "coeffs must have shape [N, K, D] or [nnz, K, D], got ", coeffs.sizes());
    d.B = prod(means.sizes().slice(0, means.dim() - 2));
    d.C = viewmats.size(-3);
    d.N = means.size(-2);
    d.K = coeffs.size(-2);
    d.D = coeffs.size(-1);
    return d;
}

// The reference's input contract, message for message (SphericalHarmonics.cpp:38-131; Python twin: _ops._check_sh_inputs)
void check_sh_inputs(int64_t degrees_to_use, const Tensor &means, const Tensor &viewmats, const Tensor &coeffs,
                     const OptTensor &masks, const OptTensor &batch_ids, const OptTensor &camera_ids, const OptTensor &gaussian_ids)
{
    TORCH_CHECK(degrees_to_use >= 0 && degrees_to_use <= 4, "degrees_to_use must be between 0 and 4, got ", degrees_to_use);
    TORCH_CHECK(means.dim() >= 2 && means.size(-1) == 3, "means must have shape [..., N, 3], got ", means.sizes());
    TORCH_CHECK(viewmats.dim() == means.dim() + 1 && viewmats.size(-2) == 4 && viewmats.size(-1) == 4,
                "viewmats must have shape [..., C, 4, 4], got ", viewmats.sizes());
    TORCH_CHECK(means.sizes().slice(0, means.dim() - 2) == viewmats.sizes().slice(0, viewmats.dim() - 3),
                "means and viewmats batch dimensions must match");
    TORCH_CHECK(coeffs.dim() == 3, "coeffs must have shape [N, K, D] or [nnz, K, D], got ", coeffs.sizes());
    TORCH_CHECK(coeffs.size(-1) >= 1, "coeffs last dim D must be >= 1, got ", coeffs.size(-1));
    TORCH_CHECK((degrees_to_use + 1) * (degrees_to_use + 1) <= coeffs.size(-2),
                "degrees_to_use requires more SH coefficients than provided; degree ", degrees_to_use, ", coeffs shape ", coeffs.sizes());
    const bool packed = has(batch_ids) || has(camera_ids) || has(gaussian_ids);
    TORCH_CHECK(!packed || (has(batch_ids) && has(camera_ids) && has(gaussian_ids)),
                "batch_ids, camera_ids, and gaussian_ids must either all be provided or all be None");
    if (packed) {
        const int64_t nnz = coeffs.size(0);
        for (const OptTensor *ids : {&batch_ids, &camera_ids, &gaussian_ids}) {
            TORCH_CHECK((*ids)->dim() == 1 && (*ids)->numel() == nnz, "packed ID tensors must have shape [nnz]");
            TORCH_CHECK((*ids)->scalar_type() == at::kLong, "packed ID tensors must be int64");
        }
        if (has(masks)) TORCH_CHECK(masks->dim() == 1 && masks->numel() == nnz, "packed masks must have shape [nnz]");
    } else {
        TORCH_CHECK(means.size(-2) == coeffs.size(0), "means N must match coeffs N in dense mode");
        if (has(masks)) {
            at::DimVector mask_shape(viewmats.sizes().slice(0, viewmats.dim() - 2));
            mask_shape.push_back(means.size(-2));
            TORCH_CHECK(masks->sizes() == at::IntArrayRef(mask_shape), "dense masks must have shape [..., C, N]");
        }
    }
}

// Rolling-shutter SH (reference SphericalHarmonics.cuh:40-65): view direction = mean + offset, the camera offset R^T t
// AVERAGED over the two shutter endpoints. An equivalent global-shutter view matrix (identity rotation, t = that average) lets
// the kernels run unchanged (Python twin: _ops._sh_rs_viewmats / _sh_rs_split).
Tensor sh_rs_viewmats(const Tensor &viewmats, const Tensor &viewmats_rs)
{
    TORCH_CHECK(viewmats_rs.sizes() == viewmats.sizes(), "viewmats_rs must match viewmats shape");
    TORCH_CHECK(viewmats_rs.scalar_type() == at::kFloat, "viewmats_rs must be float32");
    auto offset = [](const Tensor &vm) {
        return at::matmul(vm.narrow(-2, 0, 3).narrow(-1, 0, 3).transpose(-1, -2), vm.narrow(-2, 0, 3).narrow(-1, 3, 1)).squeeze(-1);
    };
    Tensor syn = at::zeros_like(viewmats);
    syn.diagonal(0, -2, -1).fill_(1.0f);
    syn.narrow(-2, 0, 3).select(-1, 3).copy_(0.5f * (offset(viewmats) + offset(viewmats_rs)));
    return syn;
}
// v_syn[:3, 3] = S = sum over rows of v_dir; each endpoint weighs 1/2: v_R = t (x) S / 2, v_t = R S / 2
Tensor sh_rs_back(const Tensor &v_syn, const Tensor &vm)
{
    const Tensor S = v_syn.narrow(-2, 0, 3).select(-1, 3);
    Tensor g = at::zeros_like(vm);
    g.narrow(-2, 0, 3).narrow(-1, 0, 3).copy_(0.5f * vm.narrow(-2, 0, 3).select(-1, 3).unsqueeze(-1) * S.unsqueeze(-2));
    g.narrow(-2, 0, 3).select(-1, 3).copy_(0.5f * at::matmul(vm.narrow(-2, 0, 3).narrow(-1, 0, 3), S.unsqueeze(-1)).squeeze(-1));
    return g;
}

Tensor spherical_harmonics(int64_t degrees_to_use, const Tensor &means_, const Tensor &viewmats_, const Tensor &coeffs_,
                           const OptTensor &masks_, const OptTensor &batch_ids_, const OptTensor &camera_ids_,
                           const OptTensor &gaussian_ids_, const OptTensor &viewmats_rs)
{
    if (has(viewmats_rs))
        return spherical_harmonics(degrees_to_use, means_, sh_rs_viewmats(viewmats_, *viewmats_rs), coeffs_, masks_, batch_ids_,
                                   camera_ids_, gaussian_ids_, OptTensor());
    check_sh_inputs(degrees_to_use, means_, viewmats_, coeffs_, masks_, batch_ids_, camera_ids_, gaussian_ids_);
    want_f32(means_, "means"); want_f32(viewmats_, "viewmats");
    if (coeffs_.scalar_type() == at::kHalf) {
        // half coefficients, float arithmetic and colours (reference SphericalHarmonicsCUDA.cu:609-638): the band kernels of
        // csrc/sh_band.hip read dense [N, K, 3] half rows in place; gathered packed rows / D != 3 widen first (rare layouts)
        if (coeffs_.dim() == 3 && coeffs_.size(-1) == 3 && !has(gaussian_ids_)) {
            Launch L(means_);
            const Tensor means = contig(means_), viewmats = contig(viewmats_), coeffs = contig(coeffs_);
            const OptTensor masks = contig(masks_);
            const ShDims d = sh_dims(means, viewmats, coeffs, gaussian_ids_);
            TORCH_CHECK(coeffs.size(0) == d.N, "means N must match coeffs N in dense mode");
            std::vector<int64_t> shape(viewmats.sizes().begin(), viewmats.sizes().end() - 2);
            shape.push_back(d.N); shape.push_back(3);
            Tensor colors = at::empty(shape, means.options());
            { Timed timed_("gsx_sh_band_fwd", L.stream); check(gsx_sh_band_fwd((int)degrees_to_use, 0, 1, fp(means), fp(viewmats), coeffs.const_data_ptr(),
                                  has(masks) ? (const uint8_t *)masks->const_data_ptr<bool>() : nullptr, nullptr, nullptr, nullptr,
                                  (uint32_t)d.B, (uint32_t)d.C, (uint32_t)d.N, -1, (uint32_t)d.K, mp<float>(colors), L.stream),
                  "gsx_sh_band_fwd"); }
            return colors;
        }
        return spherical_harmonics(degrees_to_use, means_, viewmats_, coeffs_.to(at::kFloat), masks_, batch_ids_, camera_ids_,
                                   gaussian_ids_, viewmats_rs);
    }
    want_f32(coeffs_, "coeffs");
    Launch L(means_);
    const Tensor means = contig(means_), viewmats = contig(viewmats_), coeffs = contig(coeffs_);
    const OptTensor masks = contig(masks_), bi = contig(batch_ids_), ci = contig(camera_ids_), gi = contig(gaussian_ids_);
    const ShDims d = sh_dims(means, viewmats, coeffs, gi);
    Tensor colors;
    int64_t nnz = -1;
    if (d.packed) {
        nnz = gi->size(0);
        colors = at::empty({nnz, d.D}, means.options());
    } else {
        TORCH_CHECK(coeffs.size(0) == d.N, "means N must match coeffs N in dense mode");
        std::vector<int64_t> shape(viewmats.sizes().begin(), viewmats.sizes().end() - 2);
        shape.push_back(d.N); shape.push_back(d.D);
        colors = at::empty(shape, means.options());
    }
    { Timed timed_("gsx_sh_fwd", L.stream); check(gsx_sh_fwd((int)degrees_to_use, fp(means), fp(viewmats), fp(coeffs), has(masks) ? (const uint8_t *)masks->const_data_ptr<bool>() : nullptr,
                     cp<int64_t>(bi), cp<int64_t>(ci), cp<int64_t>(gi), (uint32_t)d.B, (uint32_t)d.C, (uint32_t)d.N, nnz, 1,
                     (uint32_t)d.K, (uint32_t)d.D, nullptr, 0, mp<float>(colors), L.stream),
          "gsx_sh_fwd"); }
    return colors;
}

std::tuple<Tensor, OptTensor, OptTensor, OptTensor>
spherical_harmonics_bwd(int64_t degrees_to_use, const Tensor &means_, const Tensor &viewmats_, const Tensor &coeffs_,
                        const OptTensor &masks_, const OptTensor &batch_ids_, const OptTensor &camera_ids_,
                        const OptTensor &gaussian_ids_, const OptTensor &viewmats_rs, const Tensor &v_colors_,
                        bool compute_v_means, bool compute_v_viewmats, bool compute_v_viewmats_rs)
{
    if (has(viewmats_rs)) {
        auto r = spherical_harmonics_bwd(degrees_to_use, means_, sh_rs_viewmats(viewmats_, *viewmats_rs), coeffs_, masks_, batch_ids_,
                                         camera_ids_, gaussian_ids_, OptTensor(), v_colors_, compute_v_means,
                                         compute_v_viewmats || compute_v_viewmats_rs, false);
        OptTensor v_vm, v_rs;
        if (std::get<2>(r).has_value()) {
            if (compute_v_viewmats) v_vm = sh_rs_back(*std::get<2>(r), viewmats_);
            if (compute_v_viewmats_rs) v_rs = sh_rs_back(*std::get<2>(r), *viewmats_rs);
        }
        return {std::get<0>(r), std::get<1>(r), v_vm, v_rs};
    }
    TORCH_CHECK(!compute_v_viewmats_rs, "compute_v_viewmats_rs needs viewmats_rs");
    if (coeffs_.scalar_type() == at::kHalf) { // see spherical_harmonics: v_coeffs comes back in the coefficients' own type
        if (coeffs_.dim() == 3 && coeffs_.size(-1) == 3 && !has(gaussian_ids_)) {
            Launch L(means_);
            const Tensor means = contig(means_), viewmats = contig(viewmats_), coeffs = contig(coeffs_), vcol = contig(v_colors_);
            const OptTensor masks = contig(masks_);
            const ShDims d = sh_dims(means, viewmats, coeffs, gaussian_ids_);
            Tensor v_coeffs = at::empty_like(coeffs);
            OptTensor v_means, v_viewmats;
            if (compute_v_means) v_means = at::empty_like(means);
            Tensor v_dirs;
            if (compute_v_viewmats) v_dirs = at::zeros({d.B * d.C * d.N, 3}, means.options());
            { Timed timed_("gsx_sh_band_bwd", L.stream); check(gsx_sh_band_bwd((int)degrees_to_use, 0, 1, fp(means), fp(viewmats), coeffs.const_data_ptr(),
                                  has(masks) ? (const uint8_t *)masks->const_data_ptr<bool>() : nullptr, (uint32_t)d.B, (uint32_t)d.C,
                                  (uint32_t)d.N, -1, (uint32_t)d.K, vcol.const_data_ptr<float>(), nullptr, v_coeffs.mutable_data_ptr(),
                                  v_means ? mp<float>(*v_means) : nullptr, v_dirs.defined() ? mp<float>(v_dirs) : nullptr, L.stream),
                  "gsx_sh_band_bwd"); }
            if (compute_v_viewmats) {
                const Tensor S = v_dirs.view({d.B * d.C, d.N, 3}).sum(1);
                const Tensor vm = viewmats.reshape({d.B * d.C, 4, 4});
                const Tensor R = vm.slice(1, 0, 3).slice(2, 0, 3), t = vm.slice(1, 0, 3).select(2, 3);
                Tensor v_vm = at::zeros_like(vm);
                v_vm.slice(1, 0, 3).slice(2, 0, 3).copy_(t.unsqueeze(2) * S.unsqueeze(1));
                v_vm.slice(1, 0, 3).select(2, 3).copy_(at::einsum("cij,cj->ci", {R, S}));
                v_viewmats = v_vm.reshape(viewmats.sizes());
            }
            return {v_coeffs, v_means, v_viewmats, OptTensor()};
        }
        auto r = spherical_harmonics_bwd(degrees_to_use, means_, viewmats_, coeffs_.to(at::kFloat), masks_, batch_ids_, camera_ids_,
                                         gaussian_ids_, viewmats_rs, v_colors_, compute_v_means, compute_v_viewmats, false);
        return {std::get<0>(r).to(at::kHalf), std::get<1>(r), std::get<2>(r), std::get<3>(r)};
    }
    Launch L(means_);
    const Tensor means = contig(means_), viewmats = contig(viewmats_), coeffs = contig(coeffs_);
    const OptTensor masks = contig(masks_), bi = contig(batch_ids_), ci = contig(camera_ids_), gi = contig(gaussian_ids_);
    const ShDims d = sh_dims(means, viewmats, coeffs, gi);
    auto [vcol, vstride] = row_view(v_colors_, d.D); // may be a column view of the compositing kernel's gradient rows
    Tensor v_coeffs = at::empty_like(coeffs);        // gathered coefficient rows: fully written by the kernel
    OptTensor v_means, v_viewmats;
    if (compute_v_means) {
        const bool full_write = d.D == 3 && d.N > 0 && !d.packed; // sh3_bwd_dense_kernel stores every (b, g)
        v_means = full_write ? at::empty_like(means) : at::zeros_like(means);
    }
    const int64_t nnz = d.packed ? gi->size(0) : -1;
    Tensor v_dirs;
    if (compute_v_viewmats) v_dirs = at::zeros({d.packed ? nnz : d.B * d.C * d.N, 3}, means.options());
    { Timed timed_("gsx_sh_bwd", L.stream); check(gsx_sh_bwd((int)degrees_to_use, fp(means), fp(viewmats), fp(coeffs), has(masks) ? (const uint8_t *)masks->const_data_ptr<bool>() : nullptr,
                     cp<int64_t>(bi), cp<int64_t>(ci), cp<int64_t>(gi), (uint32_t)d.B, (uint32_t)d.C, (uint32_t)d.N, nnz, 1,
                     (uint32_t)d.K, (uint32_t)d.D, nullptr, nullptr, vcol.const_data_ptr<float>(), vstride, nullptr,
                     mp<float>(v_coeffs), v_means ? mp<float>(*v_means) : nullptr, v_dirs.defined() ? mp<float>(v_dirs) : nullptr,
                     L.stream),
          "gsx_sh_bwd"); }
    if (compute_v_viewmats) {
        // dir = mean + R^T t  =>  v_R = t (x) sum_rows v_dir,  v_t = R sum_rows v_dir per camera (small host-side tensors)
        Tensor S;
        if (d.packed) {
            S = at::zeros({d.B * d.C, 3}, means.options());
            S.index_add_(0, *bi * d.C + *ci, v_dirs);
        } else {
            S = v_dirs.view({d.B * d.C, d.N, 3}).sum(1);
        }
        const Tensor vm = viewmats.reshape({d.B * d.C, 4, 4});
        const Tensor R = vm.slice(1, 0, 3).slice(2, 0, 3), t = vm.slice(1, 0, 3).select(2, 3);
        Tensor v_vm = at::zeros_like(vm);
        v_vm.slice(1, 0, 3).slice(2, 0, 3).copy_(t.unsqueeze(2) * S.unsqueeze(1));
        v_vm.slice(1, 0, 3).select(2, 3).copy_(at::einsum("cij,cj->ci", {R, S}));
        v_viewmats = v_vm.reshape(viewmats.sizes());
    }
    return {v_coeffs, v_means, v_viewmats, OptTensor()};
}

// ---- longest tile list of an intersection result, for the compositing calls that consume it ---------------------------
// A caller that drives the STAGE ops itself (the reference's isect_tiles -> isect_offset_encode -> rasterize_to_pixels, e.g.
// its own Python over this shim) has no orchestrator to carry the hint: the intersection notes the longest list of the result
// it returns under the address + length of `flatten_ids`, and a compositing call without a hint looks its `flatten_ids` up.
// The key is the IDENTITY OF THE STORAGE, not an address: a note holds a weak reference to the StorageImpl of the tensor it was
// taken for, which keeps that object's address from being handed out again for as long as the note exists - a later tensor
// that the allocator placed at the same device address has another StorageImpl and does not match (round 4 keyed the notes by
// data pointer + length: after allocator reuse a stage-level call could pick up another intersection's value, and which
// kernel ran - and so the float summation order - depended on the process' history).
struct LongestNote {
    std::optional<c10::weak_intrusive_ptr<c10::StorageImpl>> storage; // empty slot: no value
    int64_t offset = 0, n = 0, longest = 0;
    const c10::StorageImpl *target() const { return storage ? storage->_unsafe_get_target() : nullptr; }
};
static std::mutex g_notes_mu;
static LongestNote g_notes[16];
static unsigned g_notes_next = 0;
void note_longest(const Tensor &flat, int64_t longest)
{
    if (!flat.defined() || flat.numel() <= 0 || !flat.has_storage()) return;
    c10::StorageImpl *impl = flat.storage().unsafeGetStorageImpl();
    std::lock_guard<std::mutex> lock(g_notes_mu);
    LongestNote fresh;
    fresh.storage = c10::weak_intrusive_ptr<c10::StorageImpl>(c10::intrusive_ptr<c10::StorageImpl>::reclaim_copy(impl));
    fresh.offset = flat.storage_offset(); fresh.n = flat.numel(); fresh.longest = longest;
    for (auto &e : g_notes)
        if (e.target() == impl) { e = std::move(fresh); return; }
    g_notes[g_notes_next++ % 16u] = std::move(fresh);
}
int64_t lookup_longest(const Tensor &flat)
{
    if (!flat.defined() || flat.numel() <= 0 || !flat.has_storage()) return 0;
    const c10::StorageImpl *impl = flat.storage().unsafeGetStorageImpl();
    std::lock_guard<std::mutex> lock(g_notes_mu);
    for (const auto &e : g_notes)
        if (e.target() == impl && e.offset == flat.storage_offset() && e.n == flat.numel()) return e.longest;
    return 0;
}

// ---- the segment workspace of a compositing FORWARD, for the backward over the same lists ------------------------------
// gsx_raster3d_fwd_seg leaves every slice's colour sums and end transmittance in its workspace; a backward that still has it
// needs no pre-pass (gsx_raster3d_bwd_seg_reuse). The reference's op schemas have no slot for a workspace, so the forward
// body notes it under the identity of the `last_ids` it returns (the tensor every autograd formula - this package's and the
// reference's own - saves and hands to the backward op), the same way the longest list is noted above: a weak reference to
// the StorageImpl, never an address. The workspace itself is held STRONGLY (tens of MB for a 1080p scene) in a ring of four;
// a note whose last_ids died is dropped at the next call of either function.
struct SegWsNote {
    std::optional<c10::weak_intrusive_ptr<c10::StorageImpl>> storage;
    int64_t offset = 0, n = 0, n_isects = 0, cdim = 0, seg_len = 0;
    std::vector<int64_t> inputs; // (address, version) of every tensor the forward composited from: see seg_ws_inputs_key
    Tensor ws;
    const c10::StorageImpl *target() const { return storage ? storage->_unsafe_get_target() : nullptr; }
};
static std::mutex g_seg_ws_mu;
static SegWsNote g_seg_ws[4];
static unsigned g_seg_ws_next = 0;
// The sums in the workspace belong to the forward's INPUTS: a backward call that brings other tensors than the forward saw, or
// the same ones written to since (version counter), gets no workspace and runs its pre-pass on what it was given.
static std::vector<int64_t> seg_ws_inputs_key(at::TensorList inputs)
{
    std::vector<int64_t> k;
    for (const Tensor &t : inputs) {
        k.push_back(t.defined() ? (int64_t)reinterpret_cast<intptr_t>(t.const_data_ptr()) : 0);
        k.push_back(t.defined() && !t.is_inference() ? (int64_t)t._version() : 0);
    }
    return k;
}
static void seg_ws_purge_locked()
{
    for (auto &e : g_seg_ws)
        if (e.storage && e.storage->expired()) e = SegWsNote();
}
void note_seg_workspace(const Tensor &last_ids, const Tensor &ws, int64_t n_isects, int64_t cdim, int64_t seg_len,
                        at::TensorList inputs)
{
    if (!last_ids.defined() || last_ids.numel() <= 0 || !last_ids.has_storage()) return;
    c10::StorageImpl *impl = last_ids.storage().unsafeGetStorageImpl();
    std::lock_guard<std::mutex> lock(g_seg_ws_mu);
    seg_ws_purge_locked();
    SegWsNote fresh;
    fresh.storage = c10::weak_intrusive_ptr<c10::StorageImpl>(c10::intrusive_ptr<c10::StorageImpl>::reclaim_copy(impl));
    fresh.offset = last_ids.storage_offset(); fresh.n = last_ids.numel();
    fresh.n_isects = n_isects; fresh.cdim = cdim; fresh.seg_len = seg_len; fresh.ws = ws;
    fresh.inputs = seg_ws_inputs_key(inputs);
    for (auto &e : g_seg_ws)
        if (e.target() == impl) { e = std::move(fresh); return; }
    g_seg_ws[g_seg_ws_next++ % 4u] = std::move(fresh);
}
Tensor lookup_seg_workspace(const Tensor &last_ids, int64_t n_isects, int64_t cdim, int64_t seg_len, at::TensorList inputs)
{
    if (!last_ids.defined() || last_ids.numel() <= 0 || !last_ids.has_storage()) return Tensor();
    const c10::StorageImpl *impl = last_ids.storage().unsafeGetStorageImpl();
    std::lock_guard<std::mutex> lock(g_seg_ws_mu);
    seg_ws_purge_locked();
    const std::vector<int64_t> key = seg_ws_inputs_key(inputs);
    for (const auto &e : g_seg_ws)
        if (e.target() == impl && e.inputs == key && e.offset == last_ids.storage_offset() && e.n == last_ids.numel() && e.n_isects == n_isects
            && e.cdim == cdim && e.seg_len == seg_len && e.ws.defined() && e.ws.device() == last_ids.device())
            return e.ws;
    return Tensor();
}
static bool seg_reuse_env()
{
    static const bool v = [] {
        const char *e = std::getenv("GSPLAT_AMD_SEG_REUSE"); // 0: always the pre-pass (A/B)
        return !(e && (e[0] == '0' || e[0] == 0));
    }();
    return v;
}

// ---- tile intersection --------------------------------------------------------------------------------------------------
Tensor bytes(int64_t n, const Tensor &like) { return at::empty({n < 8 ? 8 : n}, like.options().dtype(at::kByte)); }

std::tuple<Tensor, Tensor, Tensor>
intersect_tile(const Tensor &means2d_, const Tensor &radii_, const Tensor &depths_, const OptTensor &conics_,
               const OptTensor &opacities_, const OptTensor &image_ids_, const OptTensor &gaussian_ids, std::optional<int64_t> n_images,
               int64_t tile_size, int64_t tile_w, int64_t tile_h, bool sort, bool segmented)
{
    (void)gaussian_ids; // the global sort is used (results are identical to the segmented one), but the reference's refusal of
    // segmented + packed (Intersect.cpp:207-211) is part of the contract
    TORCH_CHECK(!(has(image_ids_) && segmented), "segmented sort is not supported for packed inputs");
    const bool f64 = means2d_.scalar_type() == at::kDouble; // radius boxes in double, depth narrowed to float32 in the key
    if (f64) {
        TORCH_CHECK(!has(conics_) && !has(opacities_), "gsplat_amd: intersect_tile with float64 rows supports the "
                         "radius-box test only (conics / opacities select the exact test, which is computed in fp32)");
    } else {
        want_f32(means2d_, "means2d"); want_f32(depths_, "depths"); want_f32(conics_, "conics"); want_f32(opacities_, "opacities");
    }
    Launch L(means2d_);
    const bool packed = has(image_ids_);
    const Tensor means2d = contig(means2d_), depths = contig(f64 ? depths_.to(at::kDouble) : depths_);
    const Tensor radii = contig(radii_.scalar_type() == at::kInt ? radii_ : radii_.to(at::kInt));
    const OptTensor conics = contig(conics_), opac = contig(opacities_), image_ids = contig(image_ids_);
    int64_t rows, n_per, I;
    std::vector<int64_t> out_shape;
    if (packed) {
        TORCH_CHECK(n_images.has_value(), "n_images is required when packed");
        rows = means2d.size(0); n_per = 1; I = *n_images;
        out_shape = {rows};
    } else {
        auto image_dims = means2d.sizes().slice(0, means2d.dim() - 2);
        I = prod(image_dims); n_per = means2d.size(-2); rows = I * n_per;
        out_shape.assign(means2d.sizes().begin(), means2d.sizes().end() - 1);
    }
    const uint32_t tile_bits = bits_for_count(tile_w * tile_h), image_bits = bits_for_count(I);
    TORCH_CHECK(tile_bits + image_bits <= 32, "intersect_tile: tile id bits (", tile_bits, ") + image id bits (", image_bits,
                ") exceed the 32 bits available above the depth in the 64-bit sort key");
    Tensor tiles_per_gauss = at::empty(out_shape, means2d.options().dtype(at::kInt));
    auto none = [&]() {
        return std::make_tuple(tiles_per_gauss, at::empty({0}, means2d.options().dtype(at::kLong)),
                               at::empty({0}, means2d.options().dtype(at::kInt)));
    };
    if (rows == 0) return none();
    const uint32_t uI = (uint32_t)I, uts = (uint32_t)tile_size, utw = (uint32_t)tile_w, uth = (uint32_t)tile_h;
    // the grand total comes back through pinned host memory (the one host sync of this op: Intersect.cpp:258-259)
    Tensor host_total = at::empty({2}, at::TensorOptions().dtype(at::kLong).pinned_memory(true)); // [n_isects, longest tile list]
    host_total.mutable_data_ptr<int64_t>()[1] = 0;
    int64_t *const host_longest = host_total.mutable_data_ptr<int64_t>() + 1;
    auto hip_stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(means2d.device().index());
    if (sort && !f64 && gsx_isect_fused_supported(uI, utw, uth, packed ? 1 : 0)) {
        Tensor offsets = at::empty({I * tile_w * tile_h}, means2d.options().dtype(at::kInt));
        int64_t M = GSX_ISECT_RETRY;
        if (gsx_isect_binned_should_try(rows, uI, utw, uth, packed ? 1 : 0)) { // the one decision of this intersection
            // tile-owner-major path (csrc/isect_binned.hip); GSX_ISECT_RETRY = its entry workspace was too small
            Tensor count_ws = bytes(gsx_isect_binned_count_workspace_bytes(rows, uI, utw, uth), means2d);
            { Timed timed_("gsx_isect_binned_count", L.stream); check(gsx_isect_binned_count(fp(means2d), cp<int32_t>(radii), fp(depths), fp(conics), fp(opac), nullptr, rows, uI,
                                         uts, utw, uth, mp<int32_t>(tiles_per_gauss), mp<int32_t>(offsets),
                                         host_total.mutable_data_ptr<int64_t>(), host_longest, count_ws.mutable_data_ptr(), count_ws.numel(), L.stream),
                  "gsx_isect_binned_count"); }
            hip_stream.synchronize();
            M = *host_total.const_data_ptr<int64_t>();
            if (M != GSX_ISECT_RETRY) {
                TORCH_CHECK(M < (1ll << 31), "intersect_tile: ", M, " intersections overflow the int32 index space");
                Tensor ids = at::empty({M}, means2d.options().dtype(at::kLong)), flat = at::empty({M}, means2d.options().dtype(at::kInt));
                if (M == 0) return {tiles_per_gauss, ids, flat};
                Tensor ws = bytes(gsx_isect_binned_emit_workspace_bytes(M), means2d);
                { Timed timed_("gsx_isect_binned_emit_sort", L.stream); check(gsx_isect_binned_emit_sort(rows, uI, uts, utw, uth, count_ws.mutable_data_ptr(), count_ws.numel(),
                                                 cp<int32_t>(offsets), M, *host_longest, mp<int64_t>(ids), mp<int32_t>(flat), ws.mutable_data_ptr(), ws.numel(), L.stream),
                      "gsx_isect_binned_emit_sort"); }
                note_longest(flat, *host_longest);
                return {tiles_per_gauss, ids, flat};
            }
            gsx_isect_binned_note_retry(rows, uI, utw, uth); // sent back: not tried again for the next 63 calls of this shape
        }
        Tensor count_ws = bytes(gsx_isect_fused_count_workspace_bytes(rows, uI, utw, uth), means2d);
        { Timed timed_("gsx_isect_fused_count", L.stream); check(gsx_isect_fused_count(fp(means2d), cp<int32_t>(radii), fp(conics), fp(opac), nullptr, rows, uI, uts, utw, uth,
                                    mp<int32_t>(tiles_per_gauss), mp<int32_t>(offsets), host_total.mutable_data_ptr<int64_t>(), host_longest,
                                    count_ws.mutable_data_ptr(), count_ws.numel(), L.stream),
              "gsx_isect_fused_count"); }
        hip_stream.synchronize();
        M = *host_total.const_data_ptr<int64_t>();
        TORCH_CHECK(M < (1ll << 31), "intersect_tile: ", M, " intersections overflow the int32 index space");
        Tensor ids = at::empty({M}, means2d.options().dtype(at::kLong)), flat = at::empty({M}, means2d.options().dtype(at::kInt));
        if (M == 0) return {tiles_per_gauss, ids, flat};
        Tensor ws = bytes(gsx_isect_fused_emit_workspace_bytes(M, uI, utw, uth), means2d);
        { Timed timed_("gsx_isect_fused_emit_sort", L.stream); check(gsx_isect_fused_emit_sort(fp(means2d), cp<int32_t>(radii), fp(depths), fp(conics), fp(opac), nullptr, rows, uI, uts,
                                        utw, uth, count_ws.mutable_data_ptr(), count_ws.numel(), cp<int32_t>(offsets), M,
                                        mp<int64_t>(ids), mp<int32_t>(flat), ws.mutable_data_ptr(), ws.numel(), L.stream),
              "gsx_isect_fused_emit_sort"); }
        note_longest(flat, *host_longest);
        return {tiles_per_gauss, ids, flat};
    }
    if (f64) {
        Timed timed_("gsx_isect_count_f64", L.stream);
        check(gsx_isect_count_f64(cp<double>(means2d), cp<int32_t>(radii), cp<int64_t>(image_ids), rows, (uint32_t)n_per, uI, uts,
                                  utw, uth, mp<int32_t>(tiles_per_gauss), L.stream),
              "gsx_isect_count_f64");
    } else
    { Timed timed_("gsx_isect_count", L.stream); check(gsx_isect_count(fp(means2d), cp<int32_t>(radii), fp(conics), fp(opac), cp<int64_t>(image_ids), rows, (uint32_t)n_per, uI,
                          uts, utw, uth, mp<int32_t>(tiles_per_gauss), L.stream),
          "gsx_isect_count"); }
    Tensor cum = at::empty({rows}, means2d.options().dtype(at::kLong));
    {
        Tensor ws = bytes(gsx_scan_workspace_bytes(rows), means2d);
        { Timed timed_("gsx_scan_i32", L.stream); check(gsx_scan_i32(cp<int32_t>(tiles_per_gauss), rows, mp<int64_t>(cum), ws.mutable_data_ptr(), ws.numel(), L.stream),
              "gsx_scan_i32"); }
    }
    host_total.copy_(cum.slice(0, rows - 1, rows), /*non_blocking=*/true);
    hip_stream.synchronize();
    const int64_t M = *host_total.const_data_ptr<int64_t>();
    TORCH_CHECK(M < (1ll << 31), "intersect_tile: ", M, " intersections overflow the int32 index space");
    Tensor ids = at::empty({M}, means2d.options().dtype(at::kLong)), flat = at::empty({M}, means2d.options().dtype(at::kInt));
    if (M == 0) return {tiles_per_gauss, ids, flat};
    if (f64) {
        Timed timed_("gsx_isect_emit_f64", L.stream);
        check(gsx_isect_emit_f64(cp<double>(means2d), cp<int32_t>(radii), cp<double>(depths), cp<int64_t>(image_ids),
                                 cp<int64_t>(cum), rows, (uint32_t)n_per, uI, uts, utw, uth, mp<int64_t>(ids), mp<int32_t>(flat),
                                 L.stream),
              "gsx_isect_emit_f64");
    } else
    { Timed timed_("gsx_isect_emit", L.stream); check(gsx_isect_emit(fp(means2d), cp<int32_t>(radii), fp(depths), fp(conics), fp(opac), cp<int64_t>(image_ids),
                         cp<int64_t>(cum), rows, (uint32_t)n_per, uI, uts, utw, uth, mp<int64_t>(ids), mp<int32_t>(flat), L.stream),
          "gsx_isect_emit"); }
    if (!sort) return {tiles_per_gauss, ids, flat};
    Tensor ids2 = at::empty_like(ids), flat2 = at::empty_like(flat);
    if (gsx_isect_tile_sort_supported(uI, utw, uth)) {
        Tensor ws = bytes(gsx_isect_tile_sort_workspace_bytes(M, uI, utw, uth), means2d);
        { Timed timed_("gsx_isect_tile_sort", L.stream); check(gsx_isect_tile_sort(cp<int64_t>(ids), cp<int32_t>(flat), M, uI, utw, uth, mp<int64_t>(ids2), mp<int32_t>(flat2),
                                  ws.mutable_data_ptr(), ws.numel(), L.stream),
              "gsx_isect_tile_sort"); }
        return {tiles_per_gauss, ids2, flat2};
    }
    Tensor ws = bytes(gsx_sort_pairs_workspace_bytes(M), means2d);
    int in_alt = 0;
    { Timed timed_("gsx_sort_pairs", L.stream); check(gsx_sort_pairs(mp<int64_t>(ids), mp<int32_t>(flat), mp<int64_t>(ids2), mp<int32_t>(flat2), M,
                         (int)(32 + tile_bits + image_bits), ws.mutable_data_ptr(), ws.numel(), &in_alt, L.stream),
          "gsx_sort_pairs"); }
    if (in_alt) return {tiles_per_gauss, ids2, flat2};
    return {tiles_per_gauss, ids, flat};
}

Tensor intersect_offset(const Tensor &isect_ids_, int64_t I, int64_t tile_w, int64_t tile_h)
{
    Launch L(isect_ids_);
    const Tensor ids = contig(isect_ids_);
    Tensor offsets = at::empty({I, tile_h, tile_w}, ids.options().dtype(at::kInt));
    { Timed timed_("gsx_isect_offsets", L.stream); check(gsx_isect_offsets(cp<int64_t>(ids), ids.numel(), (uint32_t)I, (uint32_t)tile_w, (uint32_t)tile_h, mp<int32_t>(offsets),
                            L.stream),
          "gsx_isect_offsets"); }
    return offsets;
}

// ---- compositing --------------------------------------------------------------------------------------------------------
struct RasterDims {
    std::vector<int64_t> image_dims;
    int64_t I, th, tw, D;
};
RasterDims raster_dims(const Tensor &isect_offsets, const Tensor &colors)
{
    RasterDims r;
    r.image_dims.assign(isect_offsets.sizes().begin(), isect_offsets.sizes().end() - 2);
    r.I = prod(r.image_dims);
    r.th = isect_offsets.size(-2);
    r.tw = isect_offsets.size(-1);
    r.D = colors.size(-1);
    return r;
}

// Longest tile list of the intersection the NEXT compositing call consumes, set by the orchestrator (rendering.py knows it
// from the intersection's host word; the reference's op schema has no room for it). Above gsx_raster3d_seg_cut() the forward cuts
// long lists into segments (csrc/raster3d_seg.hip). 0 = unknown: one workgroup per tile. Consumed (reset) by the call.
thread_local int64_t g_long_tile_hint = 0;
// The compositing kernels' 48-byte array-of-structures rows of the NEXT compositing call (csrc/raster3d.hpp: splat_rows), announced
// by the wrapper like the hint above; `key` = the data pointer of the means2d they were written for: a body only uses rows
// announced for ITS means2d. Consumed (reset) by the call.
thread_local const float *g_splat_rows = nullptr;
thread_local const float *g_splat_rows_key = nullptr;
static const float *consume_splat_rows(const Tensor &means2d, int64_t D)
{
    const float *rows = g_splat_rows, *key = g_splat_rows_key;
    g_splat_rows = g_splat_rows_key = nullptr;
    return (rows && D == 3 && key == means2d.const_data_ptr<float>()) ? rows : nullptr;
}
// segment length / the longest list from which segmenting starts; GSPLAT_AMD_SEG_LEN overrides (A/B), 0 switches it off
static int64_t seg_len_env()
{
    static const int64_t v = [] {
        const char *e = std::getenv("GSPLAT_AMD_SEG_LEN");
        return e ? (int64_t)std::atoll(e) : (int64_t)768;
    }();
    return v;
}
#define kSegLen (seg_len_env())

std::tuple<Tensor, Tensor, Tensor, Tensor>
rasterize_to_pixels_3dgs(const Tensor &means2d_, const Tensor &conics_, const Tensor &colors_, const Tensor &opacities_,
                         const OptTensor &backgrounds_, const OptTensor &masks_, int64_t width, int64_t height, int64_t tile_size,
                         const Tensor &isect_offsets_, const Tensor &flatten_ids_, bool packed, bool absgrad)
{
    (void)packed;
    want_f32(means2d_, "means2d"); want_f32(conics_, "conics"); want_f32(colors_, "colors"); want_f32(opacities_, "opacities");
    want_f32(backgrounds_, "backgrounds");
    Launch L(means2d_);
    const RasterDims r = raster_dims(isect_offsets_, colors_);
    TORCH_CHECK(r.th * tile_size >= height && r.tw * tile_size >= width,
                      "rasterize_to_pixels: isect_offsets tile grid does not cover the image");
    TORCH_CHECK(!has(masks_) || masks_->scalar_type() == at::kBool, "masks must be a bool tensor");
    const Tensor means2d = contig(means2d_), conics = contig(conics_), colors = contig(colors_), opac = contig(opacities_);
    const OptTensor bg = contig(backgrounds_), masks = contig(masks_);
    const Tensor offsets = contig(isect_offsets_), flat = contig(flatten_ids_);
    auto shape = [&](std::initializer_list<int64_t> tail) {
        auto s = r.image_dims;
        s.insert(s.end(), tail);
        return s;
    };
    Tensor renders = at::empty(shape({height, width, r.D}), means2d.options());
    Tensor alphas = at::empty(shape({height, width, 1}), means2d.options());
    Tensor last_ids = at::empty(shape({height, width}), means2d.options().dtype(at::kInt));
    int64_t longest = g_long_tile_hint;
    g_long_tile_hint = 0;
    if (longest == 0) longest = lookup_longest(flatten_ids_); // stage-level caller: no orchestrator hint
    const float *splat_rows = consume_splat_rows(means2d, r.D);
    if (kSegLen > 0 && longest > gsx_raster3d_seg_cut(flat.numel(), (uint32_t)r.I, (uint32_t)r.tw, (uint32_t)r.th, (uint32_t)kSegLen)) {
        Tensor ws = at::empty({gsx_raster3d_seg_workspace_bytes(flat.numel(), (uint32_t)r.I, (uint32_t)r.tw, (uint32_t)r.th, (uint32_t)r.D,
                                                              (uint32_t)kSegLen)}, means2d.options().dtype(at::kByte));
        Timed timed_("gsx_raster3d_fwd", L.stream); // same stage name: it IS the compositing forward
        check(gsx_raster3d_fwd_seg(fp(means2d), fp(conics), fp(colors), fp(opac), fp(bg),
                                   masks ? (const uint8_t *)masks->const_data_ptr<bool>() : nullptr, cp<int32_t>(offsets),
                                   cp<int32_t>(flat), (uint32_t)r.I, (uint32_t)flat.numel(), (uint32_t)r.D, (uint32_t)width,
                                   (uint32_t)height, (uint32_t)tile_size, (uint32_t)r.tw, (uint32_t)r.th, mp<float>(renders),
                                   mp<float>(alphas), mp<int32_t>(last_ids), (uint32_t)kSegLen, ws.mutable_data_ptr(), ws.numel(), L.stream),
              "gsx_raster3d_fwd_seg");
        // the backward over these lists starts its slices from the sums this call left (no pre-pass): <= 4 channels only
        if (r.D <= 4 && tile_size == 16 && seg_reuse_env())
            note_seg_workspace(last_ids, ws, flat.numel(), r.D, kSegLen, {means2d_, conics_, colors_, opacities_, isect_offsets_, flatten_ids_});
    } else if (splat_rows)
    { Timed timed_("gsx_raster3d_fwd", L.stream); check(gsx_raster3d_fwd_rows(fp(means2d), fp(conics), fp(colors), fp(opac), splat_rows, fp(bg),
                           masks ? (const uint8_t *)masks->const_data_ptr<bool>() : nullptr, cp<int32_t>(offsets),
                           cp<int32_t>(flat), (uint32_t)r.I, (uint32_t)flat.numel(), (uint32_t)r.D, (uint32_t)width,
                           (uint32_t)height, (uint32_t)tile_size, (uint32_t)r.tw, (uint32_t)r.th, mp<float>(renders),
                           mp<float>(alphas), mp<int32_t>(last_ids), L.stream),
          "gsx_raster3d_fwd_rows"); }
    else
    { Timed timed_("gsx_raster3d_fwd", L.stream); check(gsx_raster3d_fwd(fp(means2d), fp(conics), fp(colors), fp(opac), fp(bg),
                           masks ? (const uint8_t *)masks->const_data_ptr<bool>() : nullptr, cp<int32_t>(offsets),
                           cp<int32_t>(flat), (uint32_t)r.I, (uint32_t)flat.numel(), (uint32_t)r.D, (uint32_t)width,
                           (uint32_t)height, (uint32_t)tile_size, (uint32_t)r.tw, (uint32_t)r.th, mp<float>(renders),
                           mp<float>(alphas), mp<int32_t>(last_ids), L.stream),
          "gsx_raster3d_fwd"); }
    Tensor holder = absgrad ? at::zeros_like(means2d) : at::empty({0}, means2d.options());
    return {renders, alphas, holder, last_ids};
}

std::tuple<OptTensor, Tensor, Tensor, Tensor, Tensor, OptTensor>
rasterize_to_pixels_3dgs_bwd(const Tensor &means2d_, const Tensor &conics_, const Tensor &colors_, const Tensor &opacities_,
                             const OptTensor &backgrounds_, const OptTensor &masks_, const Tensor &tile_offsets_,
                             const Tensor &flatten_ids_, const Tensor &render_alphas_, const Tensor &last_ids_, int64_t width,
                             int64_t height, int64_t tile_size, bool absgrad, const Tensor &v_render_colors_,
                             const Tensor &v_render_alphas_, bool compute_v_backgrounds)
{
    Launch L(means2d_);
    const RasterDims r = raster_dims(tile_offsets_, colors_);
    const Tensor means2d = contig(means2d_), conics = contig(conics_), colors = contig(colors_), opac = contig(opacities_);
    const OptTensor bg = contig(backgrounds_), masks = contig(masks_);
    const Tensor offsets = contig(tile_offsets_), flat = contig(flatten_ids_), ra = contig(render_alphas_), li = contig(last_ids_);
    const Tensor v_rc = contig(v_render_colors_);
    const Tensor v_ra = v_render_alphas_.defined() ? contig(v_render_alphas_) : Tensor(); // undefined = zeros
    // ONE zero-filled array-of-structures buffer [R][6 (+2) + D]; the gradients are COLUMN VIEWS of it (gsplat_amd.h)
    const int64_t R = opac.numel(), geo = absgrad ? 8 : 6;
    int64_t longest = g_long_tile_hint; // set by the autograd formula around this call (gsplat_amd/_autograd.py)
    g_long_tile_hint = 0;
    if (longest == 0) longest = lookup_longest(flatten_ids_); // e.g. the reference's own autograd formula
    const float *splat_rows = consume_splat_rows(means2d, r.D);
    const bool segmented = kSegLen > 0 && !absgrad && r.D <= 4 && tile_size == 16
        && longest > gsx_raster3d_seg_cut(flat.numel(), (uint32_t)r.I, (uint32_t)r.tw, (uint32_t)r.th, (uint32_t)kSegLen);
    // the per-tile launch zero-fills the rows itself (inside its tile-order kernel: gsx_raster3d_bwd_fill)
    Tensor rows = segmented ? at::zeros({R, geo + r.D}, means2d.options()) : at::empty({R, geo + r.D}, means2d.options());
    if (segmented) {
        Tensor ws = at::empty({gsx_raster3d_bwd_seg_workspace_bytes(flat.numel(), (uint32_t)r.I, (uint32_t)r.tw, (uint32_t)r.th, (uint32_t)r.D,
                                                                  (uint32_t)kSegLen)}, means2d.options().dtype(at::kByte));
        // the forward call's workspace, when this process still has it (noted under last_ids): no pre-pass
        const Tensor fws = seg_reuse_env() ? lookup_seg_workspace(last_ids_, flat.numel(), r.D, kSegLen,
                                                                  {means2d_, conics_, colors_, opacities_, tile_offsets_, flatten_ids_})
                                           : Tensor();
        Timed timed_("gsx_raster3d_bwd", L.stream);
        check(gsx_raster3d_bwd_seg_reuse(fp(means2d), fp(conics), fp(colors), fp(opac), fp(bg),
                                   masks ? (const uint8_t *)masks->const_data_ptr<bool>() : nullptr, cp<int32_t>(offsets),
                                   cp<int32_t>(flat), fp(ra), cp<int32_t>(li), fp(v_rc), fp(v_ra), (uint32_t)r.I, (uint32_t)flat.numel(),
                                   (uint32_t)r.D, (uint32_t)width, (uint32_t)height, (uint32_t)tile_size, (uint32_t)r.tw, (uint32_t)r.th,
                                   mp<float>(rows), (uint32_t)(geo + r.D), (uint32_t)kSegLen,
                                   fws.defined() ? fws.const_data_ptr() : nullptr, fws.defined() ? fws.numel() : 0,
                                   ws.mutable_data_ptr(), ws.numel(), L.stream),
              "gsx_raster3d_bwd_seg_reuse");
    } else
    {
        // workspace for the longest-first tile order of the launch (csrc/raster3d_bwd.hip: "longest tiles first")
        Tensor ws = at::empty({gsx_raster3d_bwd_workspace_bytes((uint32_t)r.I, (uint32_t)r.tw, (uint32_t)r.th)}, means2d.options().dtype(at::kByte));
        Timed timed_("gsx_raster3d_bwd", L.stream);
        check(gsx_raster3d_bwd_fill_rows(fp(means2d), fp(conics), fp(colors), fp(opac), splat_rows, fp(bg),
                                  masks ? (const uint8_t *)masks->const_data_ptr<bool>() : nullptr, cp<int32_t>(offsets),
                                  cp<int32_t>(flat), fp(ra), cp<int32_t>(li), fp(v_rc), fp(v_ra), (uint32_t)r.I, (uint32_t)flat.numel(),
                                  (uint32_t)r.D, (uint32_t)width, (uint32_t)height, (uint32_t)tile_size, (uint32_t)r.tw, (uint32_t)r.th,
                                  absgrad ? 1 : 0, mp<float>(rows), (uint32_t)(geo + r.D), R, (int64_t)-1, (int64_t)1, ws.mutable_data_ptr(), ws.numel(),
                                  L.stream), // splat_rows == NULL: the plain gsx_raster3d_bwd_fill
              "gsx_raster3d_bwd");
    }
    Tensor v_means2d = rows.slice(1, 0, 2).view(means2d.sizes()), v_conics = rows.slice(1, 2, 5).view(conics.sizes());
    Tensor v_opac = rows.select(1, 5).view(opac.sizes()), v_colors = rows.slice(1, geo, geo + r.D).view(colors.sizes());
    OptTensor v_abs, v_bg;
    if (absgrad) v_abs = rows.slice(1, 6, 8).view(means2d.sizes());
    if (has(bg) && compute_v_backgrounds) v_bg = (v_rc * (1.0 - ra)).sum(at::IntArrayRef({-3, -2})); // Rasterization.cpp:567-577
    return {v_abs, v_means2d, v_conics, v_colors, v_opac, v_bg};
}

// ---- the two halves of the fused intersection, for the orchestrators (rendering.py) --------------------------------------
// intersect_tile above blocks on the intersection count between its halves. The orchestrators enqueue the SH kernels in
// between instead (gsplat_amd/_ops.py: isect_begin / isect_finish); these are the same two halves as private ops
// (namespace gsplat_amd: not part of the reference's surface). The count travels through a pinned host word initialised to a
// sentinel: the second half polls it - no event, no stream synchronisation, and the kernels enqueued in between keep running.
std::tuple<Tensor, Tensor, Tensor, Tensor>
isect_fused_begin(const Tensor &means2d, const Tensor &radii, const Tensor &depths, const OptTensor &conics, const OptTensor &opac,
                  int64_t rows, int64_t I, int64_t tile_size, int64_t tile_w, int64_t tile_h, c10::IntArrayRef out_shape)
{
    Launch L(means2d);
    const uint32_t uI = (uint32_t)I, uts = (uint32_t)tile_size, utw = (uint32_t)tile_w, uth = (uint32_t)tile_h;
    Tensor tiles_per_gauss = at::empty(out_shape, means2d.options().dtype(at::kInt));
    // pinned host words: [0] n_isects (sentinel -1 until the count has run), [1] the longest tile list (written first)
    // [2] which count ran (1 = tile-owner-major): the second half reads the workspace laid out by THIS choice, whatever the
    // environment switches say by then
    Tensor host_total = at::empty({3}, at::TensorOptions().dtype(at::kLong).pinned_memory(true));
    host_total.mutable_data_ptr<int64_t>()[0] = -1;
    host_total.mutable_data_ptr<int64_t>()[1] = 0;
    const bool binned_path = gsx_isect_binned_should_try(rows, uI, utw, uth, 0) != 0; // the one decision; slot [2] carries it
    host_total.mutable_data_ptr<int64_t>()[2] = binned_path ? 1 : 0;
    Tensor offsets = at::empty({I * tile_w * tile_h}, means2d.options().dtype(at::kInt));
    if (binned_path) { // tile-owner-major path (csrc/isect_binned.hip)
        Tensor count_ws = bytes(gsx_isect_binned_count_workspace_bytes(rows, uI, utw, uth), means2d);
        { Timed timed_("gsx_isect_binned_count", L.stream); check(gsx_isect_binned_count(fp(means2d), cp<int32_t>(radii), fp(depths), fp(conics), fp(opac), nullptr, rows, uI, uts,
                                     utw, uth, mp<int32_t>(tiles_per_gauss), mp<int32_t>(offsets), host_total.mutable_data_ptr<int64_t>(),
                                     host_total.mutable_data_ptr<int64_t>() + 1, count_ws.mutable_data_ptr(), count_ws.numel(), L.stream),
              "gsx_isect_binned_count"); }
        return {tiles_per_gauss, offsets, count_ws, host_total};
    }
    Tensor count_ws = bytes(gsx_isect_fused_count_workspace_bytes(rows, uI, utw, uth), means2d);
    { Timed timed_("gsx_isect_fused_count", L.stream); check(gsx_isect_fused_count(fp(means2d), cp<int32_t>(radii), fp(conics), fp(opac), nullptr, rows, uI, uts, utw, uth,
                                mp<int32_t>(tiles_per_gauss), mp<int32_t>(offsets), host_total.mutable_data_ptr<int64_t>(),
                                host_total.mutable_data_ptr<int64_t>() + 1, count_ws.mutable_data_ptr(), count_ws.numel(), L.stream),
          "gsx_isect_fused_count"); }
    return {tiles_per_gauss, offsets, count_ws, host_total};
}

std::tuple<Tensor, Tensor>
isect_fused_finish(const Tensor &means2d, const Tensor &radii, const Tensor &depths, const OptTensor &conics, const OptTensor &opac,
                   int64_t rows, int64_t I, int64_t tile_size, int64_t tile_w, int64_t tile_h, Tensor count_ws,
                   const Tensor &offsets, const Tensor &host_total, Tensor tiles_per_gauss)
{
    Launch L(means2d);
    const uint32_t uI = (uint32_t)I, uts = (uint32_t)tile_size, utw = (uint32_t)tile_w, uth = (uint32_t)tile_h;
    volatile const int64_t *slot = host_total.const_data_ptr<int64_t>();
    auto hip_stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(means2d.device().index());
    int64_t M = -1;
    {
        const auto t0 = std::chrono::steady_clock::now();
        for (uint64_t spin = 0;; ++spin) {
            M = *slot;
            if (M != -1 && M == *slot) break; // two equal reads: a value caught half-written cannot pass
            if ((spin & 63u) == 63u) std::this_thread::yield(); // the wait is ~10-100 us: do not pin a core at 100 % for it
            if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) {
                hip_stream.synchronize();
                M = *slot;
                break;
            }
        }
    }
    TORCH_CHECK(host_total.numel() > 2, "isect_fused_finish: host_total must be the three pinned words of isect_fused_begin");
    bool binned = slot[2] != 0; // the choice isect_fused_begin made (never re-derived: a query here could answer differently)
    if (binned && M == GSX_ISECT_RETRY) {
        // the binned path's entry workspace was too small for this scene (very large Gaussians), or a bin too crowded: count
        // again Gaussian-major, and do not try this shape again for a while
        binned   = false;
        gsx_isect_binned_note_retry(rows, uI, utw, uth);
        count_ws = bytes(gsx_isect_fused_count_workspace_bytes(rows, uI, utw, uth), means2d);
        { Timed timed_("gsx_isect_fused_count", L.stream); check(gsx_isect_fused_count(fp(means2d), cp<int32_t>(radii), fp(conics), fp(opac), nullptr, rows, uI, uts, utw, uth,
                                    mp<int32_t>(tiles_per_gauss), offsets.mutable_data_ptr<int32_t>(),
                                    host_total.mutable_data_ptr<int64_t>(), host_total.mutable_data_ptr<int64_t>() + 1,
                                    count_ws.mutable_data_ptr(), count_ws.numel(), L.stream),
              "gsx_isect_fused_count"); }
        hip_stream.synchronize();
        M = *slot;
    }
    TORCH_CHECK(M >= 0, "intersect_tile: the intersection count never reached the host");
    TORCH_CHECK(M < (1ll << 31), "intersect_tile: ", M, " intersections overflow the int32 index space");
    Tensor ids = at::empty({M}, means2d.options().dtype(at::kLong)), flat = at::empty({M}, means2d.options().dtype(at::kInt));
    if (M == 0) return {ids, flat};
    if (binned) {
        Tensor ws = bytes(gsx_isect_binned_emit_workspace_bytes(M), means2d);
        { Timed timed_("gsx_isect_binned_emit_sort", L.stream); check(gsx_isect_binned_emit_sort(rows, uI, uts, utw, uth, count_ws.mutable_data_ptr(), count_ws.numel(),
                                         cp<int32_t>(offsets), M, (int64_t)slot[1], mp<int64_t>(ids), mp<int32_t>(flat), ws.mutable_data_ptr(), ws.numel(), L.stream),
              "gsx_isect_binned_emit_sort"); }
        note_longest(flat, (int64_t)slot[1]);
        return {ids, flat};
    }
    Tensor ws = bytes(gsx_isect_fused_emit_workspace_bytes(M, uI, utw, uth), means2d);
    { Timed timed_("gsx_isect_fused_emit_sort", L.stream); check(gsx_isect_fused_emit_sort(fp(means2d), cp<int32_t>(radii), fp(depths), fp(conics), fp(opac), nullptr, rows, uI, uts,
                                    utw, uth, count_ws.mutable_data_ptr(), count_ws.numel(), cp<int32_t>(offsets), M,
                                    mp<int64_t>(ids), mp<int32_t>(flat), ws.mutable_data_ptr(), ws.numel(), L.stream),
          "gsx_isect_fused_emit_sort"); }
    note_longest(flat, (int64_t)slot[1]);
    return {ids, flat};
}

// ---- 2DGS: the two forward ops on the critical host path of rasterization_2dgs (the backward bodies stay in _ops.py: the
// GPU has the whole compositing backward queued while they run) -------------------------------------------------------------
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor>
projection_2dgs_fused(const Tensor &means_, const Tensor &quats_, const Tensor &scales_, const Tensor &viewmats_, const Tensor &Ks_,
                      int64_t width, int64_t height, double eps2d, double near_plane, double far_plane, double radius_clip)
{
    (void)eps2d; // accepted and unused, as in the reference (Projection2DGSFused.cu evaluates the box at one sigma)
    want_f32(means_, "means"); want_f32(quats_, "quats"); want_f32(scales_, "scales"); want_f32(viewmats_, "viewmats");
    want_f32(Ks_, "Ks");
    const int64_t N = means_.size(-2);
    TORCH_CHECK(means_.size(-1) == 3 && quats_.dim() >= 2 && quats_.size(-2) == N && quats_.size(-1) == 4
                          && scales_.dim() >= 2 && scales_.size(-2) == N && scales_.size(-1) == 3,
                      "projection_2dgs: bad shapes means ", means_.sizes(), " quats ", quats_.sizes(), " scales ", scales_.sizes());
    Launch L(means_);
    const Tensor means = contig(means_), quats = contig(quats_), scales = contig(scales_), viewmats = contig(viewmats_),
                 Ks = contig(Ks_);
    const int64_t B = prod(means.sizes().slice(0, means.dim() - 2)), C = viewmats.size(-3);
    std::vector<int64_t> shape(means.sizes().begin(), means.sizes().end() - 2);
    shape.push_back(C); shape.push_back(N);
    auto with = [&](std::initializer_list<int64_t> tail) {
        auto s = shape;
        s.insert(s.end(), tail);
        return s;
    };
    Tensor radii = at::empty(with({2}), means.options().dtype(at::kInt)), means2d = at::empty(with({2}), means.options());
    Tensor depths = at::empty(shape, means.options()), rt = at::empty(with({3, 3}), means.options());
    Tensor normals = at::empty(with({3}), means.options());
    { Timed timed_("gsx_project_2dgs_fwd", L.stream); check(gsx_project_2dgs_fwd(fp(means), fp(quats), fp(scales), fp(viewmats), fp(Ks), (uint32_t)B, (uint32_t)C, (uint32_t)N,
                               (uint32_t)width, (uint32_t)height, (float)near_plane, (float)far_plane, (float)radius_clip,
                               mp<int32_t>(radii), mp<float>(means2d), mp<float>(depths), mp<float>(rt), mp<float>(normals),
                               L.stream),
          "gsx_project_2dgs_fwd"); }
    return {radii, means2d, depths, rt, normals};
}

std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor>
rasterize_to_pixels_2dgs(const Tensor &means2d_, const Tensor &ray_transforms_, const Tensor &colors_, const Tensor &opacities_,
                         const Tensor &normals_, const Tensor &densify, const OptTensor &backgrounds_, const OptTensor &masks_,
                         int64_t width, int64_t height, int64_t tile_size, const Tensor &tile_offsets_, const Tensor &flatten_ids_,
                         bool packed, bool absgrad, bool distloss)
{
    (void)packed; (void)densify;
    want_f32(means2d_, "means2d"); want_f32(ray_transforms_, "ray_transforms"); want_f32(colors_, "colors");
    want_f32(opacities_, "opacities"); want_f32(normals_, "normals"); want_f32(backgrounds_, "backgrounds");
    Launch L(means2d_);
    const RasterDims r = raster_dims(tile_offsets_, colors_);
    TORCH_CHECK(r.th * tile_size >= height && r.tw * tile_size >= width,
                      "rasterize_to_pixels_2dgs: tile grid does not cover the image");
    TORCH_CHECK(!has(masks_) || masks_->scalar_type() == at::kBool, "masks must be a bool tensor");
    const Tensor means2d = contig(means2d_), rt = contig(ray_transforms_), colors = contig(colors_), opac = contig(opacities_),
                 normals = contig(normals_);
    const OptTensor bg = contig(backgrounds_), masks = contig(masks_);
    const Tensor offsets = contig(tile_offsets_), flat = contig(flatten_ids_);
    auto shape = [&](std::initializer_list<int64_t> tail) {
        auto s = r.image_dims;
        s.insert(s.end(), tail);
        return s;
    };
    const auto f32 = means2d.options(), i32 = means2d.options().dtype(at::kInt);
    Tensor renders = at::empty(shape({height, width, r.D}), f32), alphas = at::empty(shape({height, width, 1}), f32);
    Tensor rnormals = at::empty(shape({height, width, 3}), f32), rdistort = at::empty(shape({height, width, 1}), f32);
    Tensor rmedian = at::empty(shape({height, width, 1}), f32);
    Tensor last_ids = at::empty(shape({height, width}), i32), median_ids = at::empty(shape({height, width}), i32);
    { Timed timed_("gsx_raster2d_fwd", L.stream); check(gsx_raster2d_fwd(fp(means2d), fp(rt), fp(colors), fp(opac), fp(normals), fp(bg),
                           masks ? (const uint8_t *)masks->const_data_ptr<bool>() : nullptr, cp<int32_t>(offsets),
                           cp<int32_t>(flat), (uint32_t)r.I, (uint32_t)flat.numel(), (uint32_t)r.D, (uint32_t)width,
                           (uint32_t)height, (uint32_t)tile_size, (uint32_t)r.tw, (uint32_t)r.th, distloss ? 1 : 0,
                           mp<float>(renders), mp<float>(alphas), mp<float>(rnormals), mp<float>(rdistort), mp<float>(rmedian),
                           mp<int32_t>(last_ids), mp<int32_t>(median_ids), L.stream),
          "gsx_raster2d_fwd"); }
    Tensor holder = absgrad ? at::zeros_like(means2d) : at::empty({0}, f32);
    return {renders, alphas, rnormals, rdistort, rmedian, holder, last_ids, median_ids};
}

} // namespace
void set_long_tile_hint(int64_t longest) { g_long_tile_hint = longest; }
void set_splat_rows(const float *rows, const float *key) { g_splat_rows = rows; g_splat_rows_key = key; }
void note_longest_op(const Tensor &flatten_ids, int64_t longest) { note_longest(flatten_ids, longest); }
int64_t lookup_longest_op(const Tensor &flatten_ids) { return lookup_longest(flatten_ids); }
void note_seg_workspace_op(const Tensor &last_ids, const Tensor &ws, int64_t n_isects, int64_t cdim, int64_t seg_len,
                           at::TensorList inputs)
{
    note_seg_workspace(last_ids, ws, n_isects, cdim, seg_len, inputs);
}
std::optional<Tensor> lookup_seg_workspace_op(const Tensor &last_ids, int64_t n_isects, int64_t cdim, int64_t seg_len,
                                              at::TensorList inputs)
{
    Tensor t = lookup_seg_workspace(last_ids, n_isects, cdim, seg_len, inputs);
    return t.defined() ? std::optional<Tensor>(t) : std::nullopt;
}
} // namespace gsplat_amd

// gsplat_amd/_ops.py (ctypes): the longest tile list of the intersection that the next compositing call of THIS thread consumes
extern "C" void gsx_torch_set_long_tile_hint(int64_t longest) { gsplat_amd::set_long_tile_hint(longest); }
extern "C" void gsx_torch_set_splat_rows(uint64_t rows, uint64_t means2d_key)
{
    gsplat_amd::set_splat_rows(reinterpret_cast<const float *>(rows), reinterpret_cast<const float *>(means2d_key));
}

TORCH_LIBRARY(gsplat_amd, m)
{
    m.def("isect_fused_begin(Tensor means2d, Tensor radii, Tensor depths, Tensor? conics, Tensor? opacities, int rows, int n_images, int tile_size, "
          "int tile_w, int tile_h, int[] out_shape) -> (Tensor, Tensor, Tensor, Tensor)");
    m.def("isect_fused_finish(Tensor means2d, Tensor radii, Tensor depths, Tensor? conics, Tensor? opacities, int rows, int n_images, "
          "int tile_size, int tile_w, int tile_h, Tensor count_ws, Tensor offsets, Tensor host_total, Tensor tiles_per_gauss) -> (Tensor, Tensor)");
    // the Python op bodies (GSPLAT_AMD_COMPILED_OPS=0, A/B kernel libraries) share the compiled bodies' notes
    m.def("note_longest(Tensor flatten_ids, int longest) -> ()");
    m.def("lookup_longest(Tensor flatten_ids) -> int");
    m.def("note_seg_workspace(Tensor last_ids, Tensor ws, int n_isects, int cdim, int seg_len, Tensor[] inputs) -> ()");
    m.def("lookup_seg_workspace(Tensor last_ids, int n_isects, int cdim, int seg_len, Tensor[] inputs) -> Tensor?");
}

TORCH_LIBRARY_IMPL(gsplat_amd, CUDA, m)
{
    m.impl("isect_fused_begin", &gsplat_amd::isect_fused_begin);
    m.impl("isect_fused_finish", &gsplat_amd::isect_fused_finish);
}

// the notes are keyed by storage identity: any backend (tools/dry_run.py drives the host paths with CPU tensors)
TORCH_LIBRARY_IMPL(gsplat_amd, CompositeExplicitAutograd, m)
{
    m.impl("note_longest", &gsplat_amd::note_longest_op);
    m.impl("lookup_longest", &gsplat_amd::lookup_longest_op);
    m.impl("note_seg_workspace", &gsplat_amd::note_seg_workspace_op);
    m.impl("lookup_seg_workspace", &gsplat_amd::lookup_seg_workspace_op);
}

TORCH_LIBRARY_IMPL(gsplat, CUDA, m)
{
    using namespace gsplat_amd;
    m.impl("projection_ewa_3dgs_fused", &projection_ewa_3dgs_fused);
    m.impl("projection_ewa_3dgs_fused_bwd", &projection_ewa_3dgs_fused_bwd);
    m.impl("projection_ewa_3dgs_packed", &projection_ewa_3dgs_packed);
    m.impl("spherical_harmonics", &spherical_harmonics);
    m.impl("spherical_harmonics_bwd", &spherical_harmonics_bwd);
    m.impl("intersect_tile", &intersect_tile);
    m.impl("intersect_offset", &intersect_offset);
    m.impl("rasterize_to_pixels_3dgs", &rasterize_to_pixels_3dgs);
    m.impl("rasterize_to_pixels_3dgs_bwd", &rasterize_to_pixels_3dgs_bwd);
    m.impl("projection_2dgs_fused", &projection_2dgs_fused);
    m.impl("rasterize_to_pixels_2dgs", &rasterize_to_pixels_2dgs);
}

// timing hooks for gsplat_amd/_cabi.py: begin(only = space-separated entry points or "" for all); end() returns
// "name ms\n" per timed call (synchronises the events) in a buffer owned by this library until the next call
extern "C" void gsx_torch_profile_begin(const char *only)
{
    using namespace gsplat_amd;
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    g_prof_only.clear();
    std::string w;
    for (const char *p = only ? only : ""; ; ++p) {
        if (*p == ' ' || *p == 0) {
            if (!w.empty()) g_prof_only.insert(w);
            w.clear();
            if (*p == 0) break;
        } else w.push_back(*p);
    }
    g_prof_on = true;
}

extern "C" const char *gsx_torch_profile_end()
{
    using namespace gsplat_amd;
    static std::string out;
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    g_prof_on = false;
    out.clear();
    for (auto &r : g_prof) {
        float ms = 0.0f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess)
            out += r.name + " " + std::to_string(ms) + "\n";
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_prof.clear();
    return out.c_str();
}

// the ops above, for gsplat_amd/_ops.py (which keeps its Python body for every op NOT named here)
extern "C" const char *gsx_torch_compiled_ops()
{
    return "projection_ewa_3dgs_fused projection_ewa_3dgs_fused_bwd projection_ewa_3dgs_packed spherical_harmonics spherical_harmonics_bwd "
           "intersect_tile intersect_offset rasterize_to_pixels_3dgs rasterize_to_pixels_3dgs_bwd "
           "projection_2dgs_fused rasterize_to_pixels_2dgs";
}
