// torch custom classes that the reference's operator schemas name by type.
//
// `gsplat::rasterization_3dgs` (reference gsplat/cuda/ext.cpp:1144-1159) takes
// `__torch__.torch.classes.gsplat.UnscentedTransformParameters`, `...FThetaCameraDistortionParameters` and two optional
// sensor classes as arguments, and the reference's Python constructs the first two on EVERY rasterization() call
// (gsplat/rendering.py:576-580, 636; gsplat/cuda/_wrapper.py:207-210). A schema string can only be parsed once those
// class types exist, and a torch custom class can only be registered from C++ - so this file (host C++, no device code,
// no kernels) is the one piece of the boundary that links against libtorch. Everything here is a plain parameter record:
// the classic 3DGS path that this backend implements never reads them (the 3DGUT / f-theta / lidar / windshield paths
// that do are out of scope), but the reference's Python must be able to build, pass and pickle them.
//
// Constructor argument names, defaults and attribute names follow ext.cpp:144-657 so that reference-side code such as
// `UnscentedTransformParameters(alpha=0.2)` or `lidar_coeffs.to_cpp()` (gsplat/cuda/_lidar.py) keeps working.
#include <torch/custom_class.h>
#include <torch/library.h>

#include <array>
#include <cmath>
#include <vector>

namespace gsplat_amd {

struct UnscentedTransformParameters : torch::CustomClassHolder {
    double alpha = 0.1, beta = 2.0, kappa = 0.0, in_image_margin_factor = 0.1;
    bool require_all_sigma_points_valid = false;
};

constexpr size_t kFThetaTerms = 6; // Cameras.h:103

struct FThetaCameraDistortionParameters : torch::CustomClassHolder {
    int64_t reference_poly = 0;
    std::vector<double> pixeldist_to_angle_poly = std::vector<double>(kFThetaTerms, 0.0);
    std::vector<double> angle_to_pixeldist_poly = std::vector<double>(kFThetaTerms, 0.0);
    double max_angle = 0.0;
    std::vector<double> linear_cde = std::vector<double>(3, 0.0);
};

struct BivariateWindshieldModelParameters : torch::CustomClassHolder {
    static constexpr int64_t kMaxOrder = 5, kMaxCoeffs = 21; // ExternalDistortion.h:44-45
    at::Tensor horizontal_poly, vertical_poly, horizontal_poly_inverse, vertical_poly_inverse; // float tensors, like the reference
    int64_t reference_poly = 1;                                                                // FORWARD = 1, BACKWARD = 2
};

struct FOV : torch::CustomClassHolder {
    double start = 0.0, span = 0.0;
};

struct RowOffsetStructuredSpinningLidarModelParametersExt : torch::CustomClassHolder {
    at::Tensor row_elevations_rad, column_azimuths_rad, row_azimuth_offsets_rad;
    int64_t spinning_direction = 0;
    double spinning_frequency_hz = 0.0;
    c10::intrusive_ptr<FOV> fov_vert_rad, fov_horiz_rad;
    double fov_eps_rad = 0.0;
    at::Tensor angles_to_columns_map;
    int64_t n_bins_azimuth = 0, n_bins_elevation = 0;
    at::Tensor cdf_elevation, cdf_dense_ray_mask, tiles_pack_info, tiles_to_elements_map;
};

template <class T>
static std::vector<double> fixed(const std::vector<double> &v, size_t n, const char *what)
{
    TORCH_CHECK(v.size() == n, what, ": expected ", n, " values, got ", v.size());
    return v;
}

} // namespace gsplat_amd

using namespace gsplat_amd;

// FRAGMENT: the operator schemas of the same namespace are defined from Python (gsplat_amd/_ops.py), after this library
// has been loaded.
TORCH_LIBRARY_FRAGMENT(gsplat, m)
{
    using UT = UnscentedTransformParameters;
    m.class_<UT>("UnscentedTransformParameters")
        .def(torch::init([](double alpha, double beta, double kappa, double in_image_margin_factor,
                            bool require_all_sigma_points_valid) {
                 // the sigma-point spread sqrt(alpha^2 (3 + kappa)) must be real (Cameras.h:72-80)
                 TORCH_CHECK(alpha * alpha * (3.0 + kappa) > 0.0,
                             "UnscentedTransformParameters: alpha^2 * (3 + kappa) must be positive");
                 auto p = c10::make_intrusive<UT>();
                 p->alpha = alpha; p->beta = beta; p->kappa = kappa;
                 p->in_image_margin_factor = in_image_margin_factor;
                 p->require_all_sigma_points_valid = require_all_sigma_points_valid;
                 return p;
             }),
             "parameter record of the unscented transform",
             {torch::arg("alpha") = 0.1, torch::arg("beta") = 2.0, torch::arg("kappa") = 0.0,
              torch::arg("in_image_margin_factor") = 0.1, torch::arg("require_all_sigma_points_valid") = false})
        .def_readwrite("alpha", &UT::alpha)
        .def_readwrite("beta", &UT::beta)
        .def_readwrite("kappa", &UT::kappa)
        .def_readwrite("in_image_margin_factor", &UT::in_image_margin_factor)
        .def_readwrite("require_all_sigma_points_valid", &UT::require_all_sigma_points_valid)
        .def_pickle(
            [](const c10::intrusive_ptr<UT> &s) -> std::vector<double> {
                return {s->alpha, s->beta, s->kappa, s->in_image_margin_factor,
                        s->require_all_sigma_points_valid ? 1.0 : 0.0};
            },
            [](std::vector<double> st) -> c10::intrusive_ptr<UT> {
                TORCH_CHECK(st.size() == 5, "UnscentedTransformParameters: bad pickle state");
                auto p = c10::make_intrusive<UT>();
                p->alpha = st[0]; p->beta = st[1]; p->kappa = st[2]; p->in_image_margin_factor = st[3];
                p->require_all_sigma_points_valid = st[4] != 0.0;
                return p;
            });

    using FT = FThetaCameraDistortionParameters;
    m.class_<FT>("FThetaCameraDistortionParameters")
        .def(torch::init([](int64_t reference_poly, std::vector<double> pixeldist_to_angle_poly,
                            std::vector<double> angle_to_pixeldist_poly, double max_angle, std::vector<double> linear_cde) {
                 auto p = c10::make_intrusive<FT>();
                 p->reference_poly = reference_poly;
                 p->pixeldist_to_angle_poly = fixed<FT>(pixeldist_to_angle_poly, kFThetaTerms, "pixeldist_to_angle_poly");
                 p->angle_to_pixeldist_poly = fixed<FT>(angle_to_pixeldist_poly, kFThetaTerms, "angle_to_pixeldist_poly");
                 p->max_angle = max_angle;
                 p->linear_cde = fixed<FT>(linear_cde, 3, "linear_cde");
                 return p;
             }),
             "f-theta distortion parameter record",
             {torch::arg("reference_poly") = 0,
              torch::arg("pixeldist_to_angle_poly") = std::vector<double>(kFThetaTerms, 0.0),
              torch::arg("angle_to_pixeldist_poly") = std::vector<double>(kFThetaTerms, 0.0),
              torch::arg("max_angle") = 0.0, torch::arg("linear_cde") = std::vector<double>(3, 0.0)})
        .def_readwrite("reference_poly", &FT::reference_poly)
        .def_readwrite("pixeldist_to_angle_poly", &FT::pixeldist_to_angle_poly)
        .def_readwrite("angle_to_pixeldist_poly", &FT::angle_to_pixeldist_poly)
        .def_readwrite("max_angle", &FT::max_angle)
        .def_readwrite("linear_cde", &FT::linear_cde);

    using BW = BivariateWindshieldModelParameters;
    m.class_<BW>("BivariateWindshieldModelParameters")
        .def(torch::init<>())
        .def_readwrite("horizontal_poly", &BW::horizontal_poly)
        .def_readwrite("vertical_poly", &BW::vertical_poly)
        .def_readwrite("horizontal_poly_inverse", &BW::horizontal_poly_inverse)
        .def_readwrite("vertical_poly_inverse", &BW::vertical_poly_inverse)
        .def_readwrite("reference_poly", &BW::reference_poly)
        .def_static("get_max_order", []() -> int64_t { return BW::kMaxOrder; })
        .def_static("get_max_coeffs", []() -> int64_t { return BW::kMaxCoeffs; });

    m.class_<FOV>("FOV")
        .def(torch::init([](double start, double span) {
                 auto p = c10::make_intrusive<FOV>();
                 p->start = start; p->span = span;
                 return p;
             }),
             "angular interval", {torch::arg("start") = 0.0, torch::arg("span") = 0.0})
        .def_readwrite("start", &FOV::start)
        .def_readwrite("span", &FOV::span);

    using LP = RowOffsetStructuredSpinningLidarModelParametersExt;
    m.class_<LP>("RowOffsetStructuredSpinningLidarModelParametersExt")
        .def(torch::init([](at::Tensor row_elevations_rad, at::Tensor column_azimuths_rad,
                            at::Tensor row_azimuth_offsets_rad, int64_t spinning_direction, double spinning_frequency_hz,
                            c10::intrusive_ptr<FOV> fov_vert_rad, c10::intrusive_ptr<FOV> fov_horiz_rad,
                            double fov_eps_rad, at::Tensor angles_to_columns_map, int64_t n_bins_azimuth,
                            int64_t n_bins_elevation, at::Tensor cdf_elevation, at::Tensor cdf_dense_ray_mask,
                            at::Tensor tiles_pack_info, at::Tensor tiles_to_elements_map) {
                 auto p = c10::make_intrusive<LP>();
                 p->row_elevations_rad = row_elevations_rad; p->column_azimuths_rad = column_azimuths_rad;
                 p->row_azimuth_offsets_rad = row_azimuth_offsets_rad; p->spinning_direction = spinning_direction;
                 p->spinning_frequency_hz = spinning_frequency_hz; p->fov_vert_rad = fov_vert_rad;
                 p->fov_horiz_rad = fov_horiz_rad; p->fov_eps_rad = fov_eps_rad;
                 p->angles_to_columns_map = angles_to_columns_map; p->n_bins_azimuth = n_bins_azimuth;
                 p->n_bins_elevation = n_bins_elevation; p->cdf_elevation = cdf_elevation;
                 p->cdf_dense_ray_mask = cdf_dense_ray_mask; p->tiles_pack_info = tiles_pack_info;
                 p->tiles_to_elements_map = tiles_to_elements_map;
                 return p;
             }),
             "spinning-lidar parameter record",
             {torch::arg("row_elevations_rad"), torch::arg("column_azimuths_rad"), torch::arg("row_azimuth_offsets_rad"),
              torch::arg("spinning_direction"), torch::arg("spinning_frequency_hz"), torch::arg("fov_vert_rad"),
              torch::arg("fov_horiz_rad"), torch::arg("fov_eps_rad"), torch::arg("angles_to_columns_map"),
              torch::arg("n_bins_azimuth"), torch::arg("n_bins_elevation"), torch::arg("cdf_elevation"),
              torch::arg("cdf_dense_ray_mask"), torch::arg("tiles_pack_info"), torch::arg("tiles_to_elements_map")})
        .def_readwrite("row_elevations_rad", &LP::row_elevations_rad)
        .def_readwrite("column_azimuths_rad", &LP::column_azimuths_rad)
        .def_readwrite("row_azimuth_offsets_rad", &LP::row_azimuth_offsets_rad)
        .def_readwrite("spinning_direction", &LP::spinning_direction)
        .def_readwrite("spinning_frequency_hz", &LP::spinning_frequency_hz)
        .def_readwrite("fov_eps_rad", &LP::fov_eps_rad)
        .def_readwrite("n_bins_azimuth", &LP::n_bins_azimuth)
        .def_readwrite("n_bins_elevation", &LP::n_bins_elevation)
        .def_readwrite("fov_vert_rad", &LP::fov_vert_rad)
        .def_readwrite("fov_horiz_rad", &LP::fov_horiz_rad)
        .def_readwrite("angles_to_columns_map", &LP::angles_to_columns_map)
        .def_readwrite("cdf_elevation", &LP::cdf_elevation)
        .def_readwrite("cdf_dense_ray_mask", &LP::cdf_dense_ray_mask)
        .def_readwrite("tiles_pack_info", &LP::tiles_pack_info)
        .def_readwrite("tiles_to_elements_map", &LP::tiles_to_elements_map);
}
