"""GPU: the whole-pipeline dispatcher ops gsplat::rasterization_3dgs / _2dgs (what the reference's gsplat.rasterization()
calls) against gsplat_amd.rasterization() on the same inputs — same kernels behind both, so values and gradients must be
IDENTICAL — and gsplat::assemble_proj_features_unpacked_fwd against the unfused chain and the CPU oracle."""
import pytest
import torch

from _util import assert_close_ratio, assert_grad_close, make_scene

pytestmark = pytest.mark.gpu
DEV = "cuda"
NAMES = ("means", "quats", "scales", "opacities", "colors")


@pytest.fixture(scope="module")
def G():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    import gsplat_amd
    from gsplat_amd import _ops

    assert _ops.COMPOSITE_UNAVAILABLE is None, _ops.COMPOSITE_UNAVAILABLE
    return gsplat_amd


def _composite_3dgs(leaves, sc, W, H, *, sh_degree, packed, render_mode, antialiased, absgrad, backgrounds=None,
                    covars=None, extra_signals=None, tile_size=16):
    """The call gsplat/rendering.py:601-650 makes."""
    c = torch.classes.gsplat
    has_color = render_mode.startswith("RGB")
    return torch.ops.gsplat.rasterization_3dgs(
        leaves["means"], covars, None if covars is not None else leaves["quats"],
        None if covars is not None else leaves["scales"], leaves["opacities"], leaves["colors"] if has_color else None,
        sc["viewmats"].to(DEV), sc["Ks"].to(DEV), W, H, tile_size, 0.3, 0.01, 1e10, 0.0, backgrounds, packed, False,
        absgrad, antialiased, not antialiased, 0, False, 32, has_color, -1 if sh_degree is None else sh_degree,
        extra_signals, -1, render_mode in ("D", "ED", "RGB+D", "RGB+ED"), render_mode in ("ED", "RGB+ED"), False, False,
        None, None, c.UnscentedTransformParameters(), 4, None, None, None, c.FThetaCameraDistortionParameters(), None,
        None, True, False, False, 0, None, 1)


@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("render_mode,sh_degree,antialiased,absgrad", [
    ("RGB", 3, False, False), ("RGB+ED", 3, True, True), ("ED", None, False, False)])
def test_rasterization_3dgs_op_equals_rasterization(G, packed, render_mode, sh_degree, antialiased, absgrad):
    sc, W, H = make_scene(N=4000, C=2, width=176, height=120, seed=11, sh_degree=sh_degree)
    nch = {"RGB": 3, "RGB+ED": 4, "ED": 1}[render_mode]
    g = torch.Generator().manual_seed(2)
    v_rc, v_ra = torch.randn(2, H, W, nch, generator=g).to(DEV), torch.randn(2, H, W, 1, generator=g).to(DEV)
    bg = torch.rand(2, 3, generator=g).to(DEV) if render_mode.startswith("RGB") else None

    a = {k: sc[k].to(DEV).clone().requires_grad_(True) for k in NAMES}
    rc, ra, meta = G.rasterization(a["means"], a["quats"], a["scales"], a["opacities"], a["colors"],
                                   sc["viewmats"].to(DEV), sc["Ks"].to(DEV), W, H, sh_degree=sh_degree, packed=packed,
                                   render_mode=render_mode, rasterize_mode="antialiased" if antialiased else "classic",
                                   absgrad=absgrad, backgrounds=bg)
    ((rc * v_rc).sum() + (ra * v_ra).sum()).backward()

    b = {k: sc[k].to(DEV).clone().requires_grad_(True) for k in NAMES}
    out = _composite_3dgs(b, sc, W, H, sh_degree=sh_degree, packed=packed, render_mode=render_mode,
                          antialiased=antialiased, absgrad=absgrad, backgrounds=bg)
    assert len(out) == 19
    (rc2, ra2, extra, normals, absg, batch_ids, camera_ids, gaussian_ids, radii, means2d, depths, conics, opac, tpg,
     isect_ids, flatten_ids, isect_offsets, tile_w, tile_h) = out
    ((rc2 * v_rc).sum() + (ra2 * v_ra).sum()).backward()

    assert torch.equal(rc2, rc) and torch.equal(ra2, ra)
    assert extra.numel() == 0 and normals.numel() == 0
    assert (tile_w, tile_h) == (meta["tile_width"], meta["tile_height"])
    for got, key in ((radii, "radii"), (means2d, "means2d"), (depths, "depths"), (conics, "conics"), (opac, "opacities"),
                     (tpg, "tiles_per_gauss"), (isect_ids, "isect_ids"), (flatten_ids, "flatten_ids"),
                     (isect_offsets, "isect_offsets")):
        assert torch.equal(got, meta[key]), key
    if packed:
        assert torch.equal(gaussian_ids, meta["gaussian_ids"]) and torch.equal(camera_ids, meta["camera_ids"])
        assert batch_ids.dtype == torch.int64
    else:
        assert batch_ids.numel() == camera_ids.numel() == gaussian_ids.numel() == 0
    for k in NAMES:
        if render_mode == "ED" and k == "colors":
            continue
        # atomics (packed projection backward of two cameras, compositing flush) accumulate in unspecified order
        assert_grad_close(b[k].grad, a[k].grad, rel=2e-4, name=f"v_{k}")
    if absgrad:
        assert absg.shape == means2d.shape and float(absg.abs().sum()) > 0
        assert_grad_close(absg, meta["means2d"].absgrad, rel=2e-4, name="absgrad")
    else:
        assert absg.numel() == 0


def test_rasterization_3dgs_op_covars_and_extra_signals(G):
    """covars arrive as the six upper-triangular entries (gsplat/rendering.py:540-544); extra signals come back as
    their own output."""
    sc, W, H = make_scene(N=3000, C=1, width=128, height=96, seed=4)
    cov3 = G.quat_scale_to_covar_preci(sc["quats"].to(DEV), sc["scales"].to(DEV), True, False, False)[0]
    cov6 = G.quat_scale_to_covar_preci(sc["quats"].to(DEV), sc["scales"].to(DEV), True, False, True)[0]
    extra = torch.rand(3000, 2, device=DEV)
    leaves = {k: sc[k].to(DEV) for k in NAMES}
    rc, ra, meta = G.rasterization(leaves["means"], None, None, leaves["opacities"], leaves["colors"],
                                   sc["viewmats"].to(DEV), sc["Ks"].to(DEV), W, H, packed=False, covars=cov3,
                                   extra_signals=extra, render_mode="RGB+D")
    out = _composite_3dgs(leaves, sc, W, H, sh_degree=None, packed=False, render_mode="RGB+D", antialiased=False,
                          absgrad=False, covars=cov6, extra_signals=extra)
    assert torch.equal(out[0], rc) and torch.equal(out[1], ra)
    assert out[0].shape[-1] == 4 and out[2].shape == (1, H, W, 2)
    assert torch.equal(out[2], meta["render_extra_signals"])


def test_rasterization_3dgs_op_rejects_out_of_scope_arguments(G):
    sc, W, H = make_scene(N=64, C=1, width=32, height=32, seed=1)
    leaves = {k: sc[k].to(DEV) for k in NAMES}
    c = torch.classes.gsplat
    base = dict(sh_degree=None, packed=False, render_mode="RGB", antialiased=False, absgrad=False)
    _composite_3dgs(leaves, sc, W, H, **base)  # the plain call works
    args = list(torch.ops.gsplat.rasterization_3dgs.default._schema.arguments)
    names = [a.name for a in args]

    def call(**over):
        vals = [leaves["means"], None, leaves["quats"], leaves["scales"], leaves["opacities"], leaves["colors"],
                sc["viewmats"].to(DEV), sc["Ks"].to(DEV), W, H, 16, 0.3, 0.01, 1e10, 0.0, None, False, False, False,
                False, True, 0, False, 32, True, -1, None, -1, False, False, False, False, None, None,
                c.UnscentedTransformParameters(), 4, None, None, None, c.FThetaCameraDistortionParameters(), None, None,
                True, False, False, 0, None, 1]
        for k, v in over.items():
            vals[names.index(k)] = v
        return torch.ops.gsplat.rasterization_3dgs(*vals)

    # 3DGUT pieces that are built render (UT projection, from-world compositing on a perfect pinhole camera) ...
    for over in (dict(with_ut=True), dict(with_eval3d=True), dict(with_ut=True, with_eval3d=True)):
        out = call(**over)
        assert out[0].shape == (1, H, W, 3) and bool(torch.isfinite(out[0]).all())
    # ... so do the hit-distance depth channel and the normals of the from-world rasterizer (round 6)
    out = call(with_eval3d=True, use_hit_distance=True, append_depth=True)
    assert out[0].shape == (1, H, W, 4) and bool(torch.isfinite(out[0]).all())
    out = call(with_eval3d=True, return_normals=True)
    assert out[3].shape == (1, H, W, 3) and bool(torch.isfinite(out[3]).all())
    # ... what is not built, or is invalid in the reference too (Rendering.cpp:120-480), is refused, never approximated
    for over in (dict(with_ut=True, packed=True), dict(with_eval3d=True, packed=True), dict(rolling_shutter=0),
                 dict(use_hit_distance=True, append_depth=True), dict(return_normals=True),
                 dict(radial_coeffs=torch.zeros(1, 6, device=DEV)), dict(camera_model=3), dict(camera_model=4),
                 dict(with_eval3d=True, camera_model=4),
                 dict(rays=torch.zeros(1, H, W, 6, device=DEV))):
        with pytest.raises((RuntimeError, ValueError, NotImplementedError)):
            call(**over)


@pytest.mark.parametrize("packed", [False, True])
def test_rasterization_2dgs_op_equals_rasterization_2dgs(G, packed):
    sc, W, H = make_scene(N=3000, C=2, width=144, height=112, seed=8, sh_degree=2)
    g = torch.Generator().manual_seed(3)
    v = [torch.randn(2, H, W, n, generator=g).to(DEV) for n in (4, 1, 3, 1)]

    def loss(rc, ra, rn, rd):
        return (rc * v[0]).sum() + (ra * v[1]).sum() + (rn * v[2]).sum() + (rd * v[3]).sum()

    a = {k: sc[k].to(DEV).clone().requires_grad_(True) for k in NAMES}
    rc, ra, rn, sn, rd, rm, meta = G.rasterization_2dgs(
        a["means"], a["quats"], a["scales"], a["opacities"], a["colors"], sc["viewmats"].to(DEV), sc["Ks"].to(DEV), W, H,
        sh_degree=2, packed=packed, render_mode="RGB+ED", distloss=True)
    loss(rc, ra, rn, rd).backward()
    b = {k: sc[k].to(DEV).clone().requires_grad_(True) for k in NAMES}
    out = torch.ops.gsplat.rasterization_2dgs(
        b["means"], b["quats"], b["scales"], b["opacities"], b["colors"], sc["viewmats"].to(DEV), sc["Ks"].to(DEV), W, H,
        16, 0.3, 0.01, 1e10, 0.0, None, packed, False, False, True, 2, "RGB+ED", "expected")
    assert len(out) == 23 and out[20:] == (meta["tile_width"], meta["tile_height"], 2)
    loss(out[0], out[1], out[2], out[4]).backward()
    for got, want in ((out[0], rc), (out[1], ra), (out[2], rn), (out[3], sn), (out[4], rd), (out[5], rm),
                      (out[9], meta["radii"]), (out[12], meta["ray_transforms"]), (out[16], meta["isect_ids"])):
        assert torch.equal(got, want)
    assert out[6].numel() == 0 and (out[8] is None) == (not packed)
    for k in NAMES:
        assert_grad_close(b[k].grad, a[k].grad, rel=2e-4, name=f"v_{k}")


@pytest.mark.parametrize("Dc,degree,E,has_depth,extra_has_c,color_post", [
    (3, 3, 0, False, False, 2), (3, 3, 2, True, True, 2), (3, 0, 1, True, False, 1), (3, 4, 0, True, False, 0),
    (5, 2, 3, True, True, 2), (1, 1, 0, False, False, 1)])
def test_assemble_proj_features(G, Dc, degree, E, has_depth, extra_has_c, color_post):
    """out = [post(SH) | extra (+0.5 when extra_post == 1) | depth] (reference SphericalHarmonicsCUDA.cu:1100-1250):
    identical to the unfused chain on the GPU, and within fp32 rounding of the torch-CPU oracle."""
    from oracle import oracle

    B, C, N, K = 1, 2, 1500, 25
    g = torch.Generator().manual_seed(Dc * 10 + degree)
    means = torch.randn(N, 3, generator=g)
    sc, _, _ = make_scene(N=N, C=C, seed=1)
    viewmats = sc["viewmats"]
    coeffs = torch.randn(N, K, Dc, generator=g) * 0.4
    masks = torch.rand(C, N, generator=g) > 0.2
    extra = torch.randn((C, N, E) if extra_has_c else (N, E), generator=g) if E else None
    depths = torch.rand(C, N, generator=g) * 5 if has_depth else None
    width = Dc + E + int(has_depth)
    out = torch.full((C, N, width), float("nan"), device=DEV)
    relu = torch.zeros((C, N, Dc), dtype=torch.bool, device=DEV) if color_post == 2 else None
    extra_post = 1 if (E and Dc == 3) else 0
    d = lambda t: None if t is None else t.to(DEV)  # noqa: E731
    torch.ops.gsplat.assemble_proj_features_unpacked_fwd(
        degree, B, C, N, Dc, E, color_post, extra_post, has_depth, False, extra_has_c, d(means), d(viewmats), None,
        d(coeffs), d(extra), d(depths), d(masks), out, relu)
    assert not torch.isnan(out).any()

    def chain(sh):  # the unfused sequence of ops on whatever device `sh` lives on
        to = lambda t: None if t is None else t.to(sh.device)  # noqa: E731
        col = sh if color_post == 0 else sh + 0.5
        if color_post == 2:
            col = col.clamp_min(0)
        col = torch.where(to(masks)[..., None], col, torch.zeros_like(col))
        parts = [col]
        if E:
            e = to(extra) + (0.5 if extra_post == 1 else 0.0)
            parts.append(e if extra_has_c else e[None].expand(C, N, E))
        if has_depth:
            parts.append(to(depths)[..., None])
        return torch.cat(parts, -1)

    want = chain(G.spherical_harmonics(degree, d(means), d(viewmats), d(coeffs)))
    assert torch.equal(out[..., Dc:], want[..., Dc:])
    assert_close_ratio(out[..., :Dc], want[..., :Dc], 1e-5, 1e-6, name="colours vs unfused GPU chain")
    ref = oracle.assemble_proj_features(degree, means[None], viewmats[None], coeffs, None if extra is None else extra[None],
                                        None if depths is None else depths[None], masks[None], color_post, extra_post,
                                        has_depth, extra_has_c)[0]
    assert_close_ratio(out.cpu(), ref, 1e-4, 2e-5, name="vs oracle")
    if relu is not None:
        live = masks.to(DEV)[..., None].expand(C, N, Dc)
        assert torch.equal(relu[live], out[..., :Dc][live] > 0) and not relu[~live].any()
    # depth_is_zero: the depth column is written as zeros without reading `depths`
    if has_depth:
        out0 = torch.full_like(out, float("nan"))
        torch.ops.gsplat.assemble_proj_features_unpacked_fwd(
            degree, B, C, N, Dc, E, color_post, extra_post, True, True, extra_has_c, d(means), d(viewmats), None,
            d(coeffs), d(extra), None, d(masks), out0, None)
        assert torch.equal(out0[..., :-1], out[..., :-1]) and float(out0[..., -1].abs().max()) == 0.0


def test_assemble_proj_features_checks(G):
    z = lambda *s: torch.zeros(*s, device=DEV)  # noqa: E731
    op = torch.ops.gsplat.assemble_proj_features_unpacked_fwd
    with pytest.raises((ValueError, RuntimeError)):  # wrong output width
        op(0, 1, 1, 4, 3, 0, 0, 0, False, False, False, z(4, 3), torch.eye(4, device=DEV)[None], None, z(4, 1, 3), None,
           None, None, z(1, 4, 4), None)
    with pytest.raises((ValueError, RuntimeError)):  # relu mask without shift + relu
        op(0, 1, 1, 4, 3, 0, 1, 0, False, False, False, z(4, 3), torch.eye(4, device=DEV)[None], None, z(4, 1, 3), None,
           None, None, z(1, 4, 3), torch.zeros(1, 4, 3, dtype=torch.bool, device=DEV))
    with pytest.raises((ValueError, RuntimeError)):  # E > 0 without extra
        op(0, 1, 1, 4, 3, 2, 0, 0, False, False, False, z(4, 3), torch.eye(4, device=DEV)[None], None, z(4, 1, 3), None,
           None, None, z(1, 4, 5), None)
