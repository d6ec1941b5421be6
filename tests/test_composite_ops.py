"""CPU: the dispatcher boundary of the whole-pipeline ops — torch custom classes (csrc/torch_classes.cpp), operator
schemas verbatim against the reference's ext.cpp, the argument mapping of gsplat::rasterization_3dgs / _2dgs onto the
orchestrator, and the reference's own gsplat.rasterization() Python driving them through the shim. No kernels run here
(the orchestrator is stubbed); the numerics of the composite ops are covered by tests/test_gpu_composite.py."""
import os
import pickle
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.fixture(scope="module")
def ops():
    lib = os.path.join(ROOT, "gsplat_amd", "csrc", "libgsplat_amd_torch.so")
    if not os.path.exists(lib):
        import __graft_entry__ as g

        g.build()
    from gsplat_amd import _ops

    assert _ops.COMPOSITE_UNAVAILABLE is None, _ops.COMPOSITE_UNAVAILABLE
    return _ops


def test_custom_classes(ops):
    c = torch.classes.gsplat
    u = c.UnscentedTransformParameters()
    # defaults of the reference record (Cameras.h:59-64)
    assert (u.alpha, u.beta, u.kappa, u.in_image_margin_factor, u.require_all_sigma_points_valid) == (0.1, 2.0, 0.0, 0.1, False)
    u2 = c.UnscentedTransformParameters(alpha=0.5, require_all_sigma_points_valid=True)
    r = pickle.loads(pickle.dumps(u2))
    assert r.alpha == 0.5 and r.require_all_sigma_points_valid is True
    with pytest.raises(RuntimeError):
        c.UnscentedTransformParameters(alpha=0.0)
    f = c.FThetaCameraDistortionParameters()
    assert len(f.pixeldist_to_angle_poly) == 6 and len(f.angle_to_pixeldist_poly) == 6 and len(f.linear_cde) == 3
    with pytest.raises(RuntimeError):
        c.FThetaCameraDistortionParameters(pixeldist_to_angle_poly=[0.0] * 5)
    b = c.BivariateWindshieldModelParameters()
    assert c.BivariateWindshieldModelParameters.get_max_order() == 5
    assert c.BivariateWindshieldModelParameters.get_max_coeffs() == 21
    b.horizontal_poly = torch.tensor([0.0, 1.0, 0.0])  # tensor fields, like the reference's class (ext.cpp:430-440)
    assert b.horizontal_poly.tolist() == [0.0, 1.0, 0.0] and b.reference_poly == 1
    fov = c.FOV(0.25, 1.5)
    lp = c.RowOffsetStructuredSpinningLidarModelParametersExt(
        torch.zeros(4), torch.zeros(8), torch.zeros(4), 1, 10.0, fov, c.FOV(), 1e-3, torch.zeros(2, 2), 8, 4,
        torch.zeros(4), torch.zeros(4, dtype=torch.bool), torch.zeros(2, 2, dtype=torch.int32),
        torch.zeros(3, dtype=torch.int32))
    assert lp.n_bins_azimuth == 8 and lp.spinning_frequency_hz == 10.0


def _reference_schemas():
    src = open(os.path.join(REF, "gsplat", "cuda", "ext.cpp")).read()
    out = {}
    for m in re.finditer(r"m\.def\(\s*((?:\"[^\"]*\"\s*)+)\)", src):
        text = "".join(re.findall(r"\"([^\"]*)\"", m.group(1)))
        name = text.split("(", 1)[0].strip()
        out[name] = text
    return out


def test_schemas_are_the_references(ops):
    """Every operator this backend defines has, character for character after parsing, the schema the reference's
    TORCH_LIBRARY(gsplat) block declares (ext.cpp:983-1258)."""
    if not os.path.isdir(os.path.join(REF, "gsplat")):
        pytest.skip("reference checkout not present")
    ref = _reference_schemas()
    ours = dict(ops.SCHEMAS, **ops.COMPOSITE_SCHEMAS, **ops.CLASS_SCHEMAS)
    assert {"rasterization_3dgs", "rasterization_2dgs", "assemble_proj_features_unpacked_fwd"} <= set(ours)
    missing = [n for n in ours if n not in ref]
    assert not missing, f"ops without a reference schema: {missing}"
    for name, schema in ours.items():
        a = torch._C.parse_schema("gsplat::" + name + schema)
        b = torch._C.parse_schema("gsplat::" + ref[name])
        assert str(a) == str(b), name
        assert str(getattr(torch.ops.gsplat, name).default._schema) == str(b), name


def test_python_kernels_carry_the_schema_defaults(ops):
    """The dispatcher strips trailing arguments equal to the schema default before it calls a Python kernel, so every
    defaulted schema argument needs the same default on the Python function."""
    import inspect

    for name in dict(ops.SCHEMAS, **ops.COMPOSITE_SCHEMAS, **ops.CLASS_SCHEMAS):
        schema = getattr(torch.ops.gsplat, name).default._schema
        params = list(inspect.signature(ops.impl(name)).parameters.values())
        for i, arg in enumerate(schema.arguments):
            if arg.has_default_value():
                assert params[i].name == arg.name and params[i].default == arg.default_value, (name, arg.name)


class _Stub:
    """Stands in for rendering.rasterization(): records the call, returns tensors of the right kinds."""

    def __init__(self, packed=False, extra=False):
        self.calls, self.packed, self.extra = [], packed, extra

    def __call__(self, *args, **kw):
        self.calls.append((args, kw))
        means = args[0]
        f = lambda *s: torch.zeros(*s, dtype=means.dtype, device=means.device)  # noqa: E731
        ids = torch.zeros(3, dtype=torch.long) if self.packed else None
        m2 = f(3, 2)
        if kw.get("absgrad"):
            m2.absgrad = f(3, 2) + 7
        meta = dict(batch_ids=ids, camera_ids=ids, gaussian_ids=ids, radii=torch.zeros(3, 2, dtype=torch.int32),
                    means2d=m2, depths=f(3), conics=f(3, 3), opacities=f(3), tiles_per_gauss=torch.zeros(3, dtype=torch.int32),
                    isect_ids=torch.zeros(5, dtype=torch.long), flatten_ids=torch.zeros(5, dtype=torch.int32),
                    isect_offsets=torch.zeros(1, 1, 1, dtype=torch.int32), tile_width=1, tile_height=1)
        if self.extra:
            meta["render_extra_signals"] = f(1, 8, 8, 2)
        return f(1, 8, 8, 3), f(1, 8, 8, 1), meta


def _args3d(**over):
    N = 3
    u, ft = torch.classes.gsplat.UnscentedTransformParameters(), torch.classes.gsplat.FThetaCameraDistortionParameters()
    a = dict(means=torch.zeros(N, 3), covars=None, quats=torch.zeros(N, 4), scales=torch.zeros(N, 3),
             opacities=torch.zeros(N), colors=torch.zeros(N, 3), viewmats=torch.eye(4)[None], Ks=torch.eye(3)[None],
             image_width=8, image_height=8, tile_size=16, eps2d=0.3, near_plane=0.01, far_plane=1e10, radius_clip=0.0,
             backgrounds=None, packed=False, sparse_grad=False, absgrad=False, calc_compensations=False,
             rasterize_mode_is_classic=True, camera_model=0, segmented=False, channel_chunk=32, has_color=True,
             sh_degree=-1, extra_signals=None, extra_signals_sh_degree=-1, append_depth=False, expected_depth=False,
             with_eval3d=False, with_ut=False, rays=None, viewmats_rs=None, ut_params=u, rolling_shutter=4,
             radial_coeffs=None, tangential_coeffs=None, thin_prism_coeffs=None, ftheta_coeffs=ft, lidar_coeffs=None,
             external_distortion_params=None, global_z_order=True, use_hit_distance=False, return_normals=False,
             renderer_config=0, process_group_name=None, world_size=1)
    a.update(over)
    return list(a.values())


def test_rasterization_3dgs_argument_mapping(ops, monkeypatch):
    from gsplat_amd import rendering

    fn = ops.impl("rasterization_3dgs")
    modes = {(True, False, False, False): "RGB", (True, True, False, False): "RGB+D", (True, True, True, False): "RGB+ED",
             (False, True, False, False): "D", (False, True, True, False): "ED", (True, True, False, True): "RGB-d",
             (True, True, True, True): "RGB-Ed", (False, True, False, True): "d", (False, True, True, True): "Ed"}
    for (has_color, append_depth, expected, hit), mode in modes.items():
        stub = _Stub()
        monkeypatch.setattr(rendering, "rasterization", stub)
        out = fn(*_args3d(has_color=has_color, append_depth=append_depth, expected_depth=expected, use_hit_distance=hit))
        kw = stub.calls[0][1]
        assert kw["render_mode"] == mode
        assert (stub.calls[0][0][4] is None) == (not has_color)  # colors are dropped when the mode has none
        assert len(out) == 19 and out[17:] == (1, 1)
        # outputs the mode does not produce are empty tensors, never None (the schema has no optional outputs)
        assert all(isinstance(t, torch.Tensor) for t in out[:17])
        assert out[2].numel() == out[3].numel() == out[4].numel() == 0
        assert all(t.numel() == 0 and t.dtype == torch.long for t in out[5:8])
    stub = _Stub(packed=True, extra=True)
    monkeypatch.setattr(rendering, "rasterization", stub)
    cov = torch.zeros(3, 6)
    out = fn(*_args3d(covars=cov, quats=None, scales=None, sh_degree=2, extra_signals=torch.zeros(3, 2),
                      extra_signals_sh_degree=-1, calc_compensations=True, absgrad=True, packed=True, camera_model=2,
                      process_group_name="0", channel_chunk=7, tile_size=8))
    kw = stub.calls[0][1]
    assert kw["covars"] is cov and kw["_covars_triu"] and kw["sh_degree"] == 2 and kw["extra_signals_sh_degree"] is None
    assert kw["rasterize_mode"] == "antialiased" and kw["camera_model"] == "fisheye" and kw["distributed"] is True
    assert kw["channel_chunk"] == 7 and kw["tile_size"] == 8 and kw["packed"] is True
    assert out[2].shape == (1, 8, 8, 2) and float(out[4].sum()) == 42.0 and out[5].numel() == 3
    with pytest.raises(ValueError):
        fn(*_args3d(renderer_config=1))
    stub.calls.clear()
    fn(*_args3d(rolling_shutter=0))  # a rolling shutter is handed on (the orchestrator validates it: with_ut, viewmats_rs)
    assert stub.calls[0][1]["rolling_shutter"] == 0


def test_covars_triu_layout_is_what_the_reference_python_sends():
    """gsplat/rendering.py:540-544 flattens [..., 3, 3] to the six upper-triangular entries before the op call; the
    orchestrator takes them as they are: [N, 6] passes validation under `_covars_triu` (and only there), after which the
    call reaches the kernels - which refuse CPU tensors."""
    import gsplat_amd
    from gsplat_amd._cabi import GsplatAmdError

    N = 4
    args = (torch.rand(N, 3), None, None, torch.rand(N), torch.rand(N, 3), torch.eye(4)[None], torch.eye(3)[None], 8, 8)
    with pytest.raises((GsplatAmdError, NotImplementedError), match="ROCm device|CPU"):
        gsplat_amd.rasterization(*args, covars=torch.rand(N, 6), _covars_triu=True)
    with pytest.raises(RuntimeError, match="covars must have shape"):
        gsplat_amd.rasterization(*args, covars=torch.rand(N, 3, 3), _covars_triu=True)
    with pytest.raises(RuntimeError, match="covars must have shape"):
        gsplat_amd.rasterization(*args, covars=torch.rand(N, 6))


def test_reference_rasterization_python_drives_the_composite_op(ops):
    """INTEGRATION.md route A, end to end on the host side: the reference's own gsplat.rasterization() and
    rasterization_2dgs() (its Python, unmodified) run against the shim, build the custom-class records, call
    torch.ops.gsplat.rasterization_3dgs / _2dgs with their 50 / 22 arguments and unpack our return values into `meta`.
    The kernels cannot run without a GPU, so the orchestrator behind the op is stubbed and the composite is
    additionally registered for the CPU key inside the subprocess."""
    if not os.path.isdir(os.path.join(REF, "gsplat")):
        pytest.skip("reference checkout not present")
    code = r'''
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(ref)r); sys.path.insert(0, %(tests)r)
import torch
import gsplat_amd.csrc_shim as shim
sys.modules["gsplat.csrc"] = shim
import gsplat
from gsplat.cuda._backend import _C
assert _C is shim
from gsplat_amd import _ops, rendering
from test_composite_ops import _Stub
cpu = torch.library.Library("gsplat", "IMPL", "CPU")
for name in ("rasterization_3dgs", "rasterization_2dgs"):
    cpu.impl(name, _ops.impl(name))
stub = _Stub(packed=True)
rendering.rasterization = stub
N = 3
rc, ra, meta = gsplat.rasterization(torch.zeros(N, 3), torch.zeros(N, 4), torch.zeros(N, 3), torch.zeros(N),
                                    torch.zeros(N, 16, 3), torch.eye(4)[None], torch.eye(3)[None], 8, 8, sh_degree=3,
                                    render_mode="RGB+ED", absgrad=True, rasterize_mode="antialiased",
                                    covars=torch.eye(3).expand(N, 3, 3))
kw = stub.calls[0][1]
assert kw["render_mode"] == "RGB+ED" and kw["sh_degree"] == 3 and kw["absgrad"] and kw["packed"]
assert kw["rasterize_mode"] == "antialiased" and kw["covars"].shape == (N, 6) and kw["_covars_triu"]
assert kw["camera_model"] == "pinhole" and kw["distributed"] is False and kw["tile_size"] == 16
assert rc.shape == (1, 8, 8, 3) and meta["gaussian_ids"].numel() == 3 and meta["tile_width"] == 1
assert float(meta["means2d"].absgrad.sum()) == 42.0

def stub2d(*a, **k):
    f = lambda *s: torch.zeros(*s)
    m2 = f(3, 2)
    meta = dict(camera_ids=None, gaussian_ids=None, radii=torch.zeros(3, 2, dtype=torch.int32), means2d=m2, depths=f(3),
                ray_transforms=f(3, 3, 3), opacities=f(3), normals=f(3, 3), tiles_per_gauss=torch.zeros(3, dtype=torch.int32),
                isect_ids=torch.zeros(5, dtype=torch.long), flatten_ids=torch.zeros(5, dtype=torch.int32),
                isect_offsets=torch.zeros(1, 1, 1, dtype=torch.int32), gradient_2dgs=f(3, 2), tile_width=1, tile_height=1,
                n_cameras=1)
    stub2d.kw = k
    return f(1, 8, 8, 4), f(1, 8, 8, 1), f(1, 8, 8, 3), f(8, 8, 3), f(1, 8, 8, 1), f(1, 8, 8, 1), meta
rendering.rasterization_2dgs = stub2d
out = gsplat.rasterization_2dgs(torch.zeros(N, 3), torch.zeros(N, 4), torch.zeros(N, 3), torch.zeros(N), torch.zeros(N, 3),
                                torch.eye(4)[None], torch.eye(3)[None], 8, 8, render_mode="RGB+ED", distloss=True)
assert len(out) == 7 and out[3].shape == (8, 8, 3) and out[6]["n_cameras"] == 1 and out[6]["gradient_2dgs"].shape == (3, 2)
assert stub2d.kw["distloss"] is True and stub2d.kw["render_mode"] == "RGB+ED" and stub2d.kw["depth_mode"] == "expected"
print("OK")
''' % {"root": ROOT, "ref": REF, "tests": os.path.join(ROOT, "tests")}
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp", env=env, timeout=600)
    assert out.returncode == 0 and "OK" in out.stdout, (out.stdout[-2000:], out.stderr[-3000:])
