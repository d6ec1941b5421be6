"""CPU: input validation of rasterization() — the reference's own test table (tests/test_rasterization.py:723-800: case ->
message the error must match) plus the classic-path checks of Rendering.cpp:236-480. Validation runs before any kernel or
collective, so CPU tensors are enough and no process group is needed."""
import pytest
import torch


def _scene(N=6, C=2, width=16, height=12):
    means = torch.rand(N, 3)
    means[:, 2] += 2.0
    return {
        "means": means, "quats": torch.randn(N, 4), "scales": torch.rand(N, 3) * 0.05 + 0.01,
        "opacities": torch.rand(N), "colors": torch.rand(N, 3), "viewmats": torch.eye(4).expand(C, -1, -1).clone(),
        "Ks": torch.tensor([[20.0, 0.0, width / 2], [0.0, 20.0, height / 2], [0.0, 0.0, 1.0]]).expand(C, -1, -1).clone(),
        "width": width, "height": height,
    }


@pytest.mark.parametrize("case,match", [
    ("batch_dims", "batch dimensions"), ("with_eval3d", "with_eval3d=True"), ("with_ut", "with_ut=True"),
    ("camera_model", "camera_model"), ("sparse_grad", "sparse_grad=True"), ("absgrad", "absgrad=True"),
    ("rolling_shutter", "rolling shutter"), ("distortion", "camera distortion"), ("global_z_order", "global_z_order=False"),
    ("per_view_color", "per-Gaussian colors"), ("rays", "does not support rays"), ("return_normals", "return_normals=True"),
    ("lidar", "lidar coefficients"), ("wrong_n_color", "colors must have shape"),
])
def test_distributed_rejects_unsupported_configs(case, match):
    import gsplat_amd

    kw = _scene()
    C, N = kw["viewmats"].shape[0], kw["means"].shape[0]
    if case == "batch_dims":
        for k in ("means", "quats", "scales", "opacities", "colors"):
            kw[k] = kw[k].expand(2, *kw[k].shape).clone()
        kw["viewmats"] = kw["viewmats"].expand(2, C, 4, 4).clone()
        kw["Ks"] = kw["Ks"].expand(2, C, 3, 3).clone()
    elif case in ("with_eval3d", "with_ut", "sparse_grad", "absgrad", "return_normals"):
        kw[case] = True
    elif case == "camera_model":
        kw["camera_model"] = "ortho"
    elif case == "rays":
        kw["rays"] = torch.zeros(C, kw["height"], kw["width"], 6)
    elif case == "rolling_shutter":
        kw["rolling_shutter"] = 0  # RollingShutterType.ROLLING_TOP_TO_BOTTOM
        kw["viewmats_rs"] = kw["viewmats"].clone()
    elif case == "distortion":
        kw["radial_coeffs"] = torch.zeros(C, 6)
    elif case == "global_z_order":
        kw["global_z_order"] = False
    elif case == "per_view_color":
        kw["colors"] = torch.rand(C, N, 3)
    elif case == "lidar":
        kw["lidar_coeffs"] = object()
    elif case == "wrong_n_color":
        kw["colors"] = torch.rand(N + 1, 3)
    with pytest.raises(RuntimeError, match=match):
        gsplat_amd.rasterization(**kw, distributed=True)


def test_distributed_needs_a_process_group():
    import gsplat_amd

    with pytest.raises(ValueError, match="initialized default torch.distributed process group"):
        gsplat_amd.rasterization(**_scene(), distributed=True)


@pytest.mark.parametrize("over,exc,match", [
    (dict(external_distortion_coeffs=object()), RuntimeError, "with_ut=True"),  # test_rasterization.py:803-813
    (dict(render_mode="RGB-Ed"), RuntimeError, "hit-distance render modes require with_eval3d=True"),
    (dict(return_normals=True), RuntimeError, "return_normals=True requires with_eval3d=True"),
    (dict(global_z_order=False), RuntimeError, "global_z_order can be false only if with_ut=True"),
    (dict(camera_model="ftheta"), RuntimeError, "ftheta camera is only supported via UT"),
    (dict(camera_model="lidar"), RuntimeError, "Lidar coefficients must be given if and only if"),
    (dict(sparse_grad=True, packed=False), RuntimeError, "sparse_grad is only supported when packed is True"),
    (dict(channel_chunk=0), RuntimeError, "channel_chunk must be > 0"),
    (dict(rays=torch.zeros(2, 12, 16, 6)), RuntimeError, "Rays input is only supported with Eval3D"),
    (dict(radial_coeffs=torch.zeros(2, 6)), RuntimeError, "Radial distortion requires with_ut=True"),
    (dict(tangential_coeffs=torch.zeros(2, 2)), RuntimeError, "Tangential distortion requires with_ut=True"),
    (dict(thin_prism_coeffs=torch.zeros(2, 4)), RuntimeError, "Thin-prism distortion requires with_ut=True"),
    (dict(viewmats_rs=torch.eye(4).expand(2, 4, 4)), RuntimeError, "viewmats_rs should be None for global rolling shutter"),
    (dict(rolling_shutter=1), RuntimeError, "Rolling shutter requires with_ut=True"),
    (dict(rasterize_mode="antialiased", with_ut=True), ValueError, "only supports rasterize_mode='classic'"),
    (dict(render_mode="XYZ"), ValueError, "Unsupported render_mode"),
    (dict(opacities=torch.rand(5)), RuntimeError, r"opacities must have shape \[..., N\]"),
    (dict(quats=torch.randn(6, 3)), RuntimeError, r"quats must have shape \[..., N, 4\]"),
    (dict(scales=None), RuntimeError, "covars or scales is required"),
    (dict(Ks=torch.eye(3).expand(3, 3, 3)), RuntimeError, r"Ks must have shape \[..., C, 3, 3\]"),
    (dict(colors=None), RuntimeError, "colors must be provided for color render modes"),
    (dict(colors=torch.rand(6, 4, 3), sh_degree=2), RuntimeError, "sh_degree requires more color SH coefficients"),
    (dict(colors=torch.rand(5, 16, 3), sh_degree=1), RuntimeError, r"SH colors must have shape \[N, K, D\]"),
    (dict(covars=torch.rand(6, 2, 2)), RuntimeError, "covars must have shape"),
    (dict(with_ut=True), RuntimeError, "Packed mode is not supported with UT"),  # packed defaults to True
    (dict(with_ut=True, packed=False, covars=torch.rand(6, 3, 3), quats=None, scales=None), RuntimeError,
     "UT and Eval3D rasterization require quats and scales, not covars"),
    (dict(with_eval3d=True), RuntimeError, "Packed mode is not supported with Eval3D"),
    (dict(packed=False, camera_model="lidar", lidar_coeffs=object()), RuntimeError, "Lidar camera model requires with_ut=True"),
    (dict(with_ut=True, packed=False, camera_model="lidar"), RuntimeError,
     "Lidar coefficients must be given if and only if camera model is lidar"),
])
def test_classic_path_validation(over, exc, match):
    import gsplat_amd

    kw = _scene()
    kw.update(over)
    with pytest.raises(exc, match=match):
        gsplat_amd.rasterization(**kw)


@pytest.mark.parametrize("over,match", [
    (dict(quats=torch.randn(6, 3)), r"quats must have shape \[..., N, 4\]"),
    (dict(scales=torch.rand(5, 3)), r"scales must have shape \[..., N, 3\]"),
    (dict(opacities=torch.rand(7)), r"opacities must have shape \[..., N\]"),
    (dict(viewmats=torch.eye(3).expand(2, 3, 3)), r"viewmats must have shape \[..., C, 4, 4\]"),
    (dict(colors=torch.rand(6, 4, 3), sh_degree=2), "SH degree 2 too high for 4 coefficient bands"),
    (dict(colors=torch.rand(5, 9, 3), sh_degree=2), r"SH coefficients must have shape \[N, K, D\]"),
    (dict(distloss=True, render_mode="RGB"), "distloss requires a depth render mode"),
])
def test_2dgs_validation(over, match):
    """check_rasterization_2dgs_inputs (Rendering.cpp:1588-1636)."""
    import gsplat_amd

    kw = _scene()
    kw.update(over)
    with pytest.raises(RuntimeError, match=match):
        gsplat_amd.rasterization_2dgs(**kw)
