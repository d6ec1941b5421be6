"""GPU: long tile lists composited in SEGMENTS (csrc/raster3d_seg.hip) against the one-workgroup-per-tile walk of the same
lists, which the other suites pin to the oracle - and against the oracle directly on a small scene. Scenes: faint Gaussians
(no pixel saturates: every slice contributes), opaque ones (early termination fires inside the first slices: the later ones
carry transmittance 0 and stop at once), a mix, backgrounds, tile masks, more than 32 channels (two channel chunks),
several images. The backward (per-slice state from the forward's workspace, or pre-pass + per-pixel prefix, then slices walked
back to front) against the per-tile backward."""
import math

import pytest
import torch

from _util import assert_close_ratio, assert_grad_close, make_scene

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def G():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    import gsplat_amd

    return gsplat_amd


def _lists(G, N, C, W, H, opacity, seed, shrink=0.25):
    """Projected scene squeezed into the image centre so that tile lists run to several thousand entries."""
    sc, W, H = make_scene(N=N, C=C, width=W, height=H, seed=seed)
    a = {k: v.to(DEV) for k, v in sc.items()}
    a["means"][:, :2] *= shrink
    if opacity is not None:
        a["opacities"] = torch.full_like(a["opacities"], opacity) if not callable(opacity) else opacity(a["opacities"])
    rad, m2, d, con, _ = G.fully_fused_projection(a["means"], None, a["quats"], a["scales"], a["viewmats"], a["Ks"], W, H,
                                                  opacities=a["opacities"])
    op = a["opacities"][None].expand(C, -1).contiguous()
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    _, ids, fl = G.isect_tiles(m2, rad, d, 16, tw, th, conics=con, opacities=op)
    off = G.isect_offset_encode(ids, C, tw, th)
    counts = torch.diff(torch.cat([off.flatten(), torch.tensor([fl.numel()], device=DEV, dtype=off.dtype)]))
    from gsplat_amd import _ops

    # the scene must really take the segment path: its longest list is an outlier by the library's own rule
    assert int(counts.max()) > _ops._seg_cut(fl.numel(), C, tw, th), (int(counts.max()), _ops._seg_cut(fl.numel(), C, tw, th))
    return m2, con, op, off, fl, int(counts.max()), W, H, tw, th


@pytest.mark.parametrize("kind", ["faint", "opaque", "mixed"])
@pytest.mark.parametrize("D", [3, 35])
def test_segmented_forward_matches_per_tile_walk(G, kind, D):
    opacity = {"faint": 0.004 * 2, "opaque": 0.9, "mixed": (lambda o: torch.where(torch.rand_like(o) < 0.5, o * 0.02 + 0.004, o))}[kind]
    m2, con, op, off, fl, longest, W, H, tw, th = _lists(G, 40000, 2, 320, 192, opacity, seed=21)
    assert longest > 3000, longest  # several segments per long tile
    g = torch.Generator().manual_seed(D)
    colors = torch.rand(m2.shape[:-1] + (D,), generator=g).to(DEV)
    bg = torch.rand(2, D, generator=g).to(DEV)
    masks = torch.ones(2, th, tw, dtype=torch.bool, device=DEV)
    masks[0, th // 2, tw // 2] = False  # a masked tile in the crowded centre
    for kw in (dict(), dict(backgrounds=bg), dict(backgrounds=bg, masks=masks)):
        ref_c, ref_a = G.rasterize_to_pixels(m2, con, colors, op, W, H, 16, off, fl, _longest_tile_list=-1, **kw)  # per tile
        seg_c, seg_a = G.rasterize_to_pixels(m2, con, colors, op, W, H, 16, off, fl, _longest_tile_list=longest, **kw)
        # the two differ only in the association order of the transmittance products (and, on saturating tiles, not at all)
        assert_close_ratio(seg_c.cpu(), ref_c.cpu(), 2e-5, 2e-6, max_bad_ratio=1e-5, name=f"{kind} colours {sorted(kw)}")
        assert_close_ratio(seg_a.cpu(), ref_a.cpu(), 2e-5, 2e-6, max_bad_ratio=1e-5, name=f"{kind} alphas {sorted(kw)}")


def test_segmented_forward_last_ids_and_oracle(G):
    """last_ids (consumed by the backward) must be the per-tile walk's; and the segmented render against the CPU oracle."""
    from oracle import oracle as O

    m2, con, op, off, fl, longest, W, H, tw, th = _lists(G, 30000, 1, 256, 160, 0.01, seed=5)
    assert longest > 2500
    colors = torch.rand(m2.shape[:-1] + (3,), generator=torch.Generator().manual_seed(1)).to(DEV)
    from gsplat_amd import _ops

    outs = {}
    for name, hint in (("tile", -1), ("seg", longest)):  # -1: per tile, whatever the intersection noted
        _ops.set_long_tile_hint(hint)
        outs[name] = torch.ops.gsplat.rasterize_to_pixels_3dgs(m2, con, colors, op, None, None, W, H, 16, off, fl, False, False)
        _ops.set_long_tile_hint(0)
    assert torch.equal(outs["seg"][3], outs["tile"][3]), "last_ids differ"
    cpu = lambda t: t.detach().cpu()  # noqa: E731
    rc_o, ra_o = O.rasterize_to_pixels(cpu(m2), cpu(con), cpu(colors), cpu(op), W, H, 16, cpu(off), cpu(fl))[:2]
    assert_close_ratio(cpu(outs["seg"][0]), rc_o, 1e-3, 1e-4, max_bad_ratio=1e-3, name="segmented colours vs oracle")
    assert_close_ratio(cpu(outs["seg"][1]), ra_o, 1e-3, 1e-4, max_bad_ratio=1e-3, name="segmented alphas vs oracle")


@pytest.mark.parametrize("kind", ["faint", "opaque", "mixed"])
@pytest.mark.parametrize("D", [1, 3, 4])
def test_segmented_backward_matches_per_tile_walk(G, kind, D):
    opacity = {"faint": 0.008, "opaque": 0.9, "mixed": (lambda o: torch.where(torch.rand_like(o) < 0.5, o * 0.02 + 0.004, o))}[kind]
    m2, con, op, off, fl, longest, W, H, tw, th = _lists(G, 40000, 2, 320, 192, opacity, seed=22)
    assert longest > 3000, longest
    g = torch.Generator().manual_seed(10 + D)
    colors = torch.rand(m2.shape[:-1] + (D,), generator=g).to(DEV)
    bg = torch.rand(2, D, generator=g).to(DEV)
    masks = torch.ones(2, th, tw, dtype=torch.bool, device=DEV)
    masks[1, th // 2, tw // 2] = False
    w_c = torch.randn(2, H, W, D, generator=g).to(DEV)
    w_a = torch.randn(2, H, W, 1, generator=g).to(DEV)
    for kw in (dict(), dict(backgrounds=bg, masks=masks)):
        grads = {}
        for name, hint in (("tile", -1), ("seg", longest)):  # -1: per tile, whatever the intersection noted
            leaves = [t.detach().clone().requires_grad_(True) for t in (m2, con, colors, op)]
            extra = dict(kw)
            if "backgrounds" in extra:
                extra["backgrounds"] = bg.detach().clone().requires_grad_(True)
                leaves.append(extra["backgrounds"])
            rc, ra = G.rasterize_to_pixels(*leaves[:4], W, H, 16, off, fl, _longest_tile_list=hint, **extra)
            ((rc * w_c).sum() + (ra * w_a).sum()).backward()
            grads[name] = [t.grad.cpu() for t in leaves]
        for nm, a, b in zip(("means2d", "conics", "colors", "opacities", "backgrounds"), grads["seg"], grads["tile"]):
            # same sums in another association order (float atomics make either side run-to-run noisy at this level too)
            assert_grad_close(a, b, name=f"{kind} D={D} {sorted(kw)} v_{nm}")


def test_segmented_backward_through_rasterization(G):
    """rendering.py hands the intersection's longest list to forward AND backward; the per-tile side of the comparison
    is the same call with that report suppressed."""
    from gsplat_amd import rendering

    sc, W, H = make_scene(N=30000, C=1, width=256, height=160, seed=9)
    a = {k: v.to(DEV) for k, v in sc.items()}
    a["means"][:, :2] *= 0.25
    a["opacities"] = a["opacities"] * 0.05 + 0.004
    names = ("means", "quats", "scales", "opacities", "colors")
    out = {}
    seen = []
    saved = rendering._isect_max_tile_len
    for mode in ("seg", "tile"):
        rendering._isect_max_tile_len = (lambda st: seen.append(saved(st)) or seen[-1]) if mode == "seg" else (lambda st: -1)
        try:
            leaves = {k: a[k].detach().clone().requires_grad_(True) for k in names}
            rc, ra, info = G.rasterization(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"], leaves["colors"],
                                           a["viewmats"], a["Ks"], W, H)
            (rc.square().sum() + ra.sum()).backward()
            out[mode] = (rc.detach().cpu(), [leaves[k].grad.cpu() for k in names])
        finally:
            rendering._isect_max_tile_len = saved
    assert seen and seen[0] > 2500, seen  # the segment path really ran
    assert_close_ratio(out["seg"][0], out["tile"][0], 2e-5, 2e-6, max_bad_ratio=1e-5, name="render")
    for nm, x, y in zip(names, out["seg"][1], out["tile"][1]):
        assert_grad_close(x, y, name=f"v_{nm}")


def test_uniformly_long_lists_stay_on_the_per_tile_walk(G):
    """Segments are for outliers: when EVERY list is long (the c4 regime) the cut moves up with the mean and the plain
    entries run - bit-identical results with and without the hint."""
    from gsplat_amd import _ops

    sc, W, H = make_scene(N=60000, C=1, width=64, height=64, seed=3, scale_range=(0.05, 0.2))
    a = {k: v.to(DEV) for k, v in sc.items()}
    a["opacities"] = torch.full_like(a["opacities"], 0.01)
    rad, m2, d, con, _ = G.fully_fused_projection(a["means"], None, a["quats"], a["scales"], a["viewmats"], a["Ks"], W, H,
                                                  opacities=a["opacities"])
    op = a["opacities"][None].contiguous()
    _, ids, fl = G.isect_tiles(m2, rad, d, 16, 4, 4, conics=con, opacities=op)
    off = G.isect_offset_encode(ids, 1, 4, 4)
    counts = torch.diff(torch.cat([off.flatten(), torch.tensor([fl.numel()], device=DEV, dtype=off.dtype)]))
    longest = int(counts.max())
    assert longest > _ops.SEG_MIN_LONGEST and longest <= _ops._seg_cut(fl.numel(), 1, 4, 4), (longest, fl.numel())
    colors = torch.rand(m2.shape[:-1] + (3,), generator=torch.Generator().manual_seed(1)).to(DEV)
    ref = G.rasterize_to_pixels(m2, con, colors, op, W, H, 16, off, fl, _longest_tile_list=-1)
    hinted = G.rasterize_to_pixels(m2, con, colors, op, W, H, 16, off, fl, _longest_tile_list=longest)
    assert torch.equal(ref[0], hinted[0]) and torch.equal(ref[1], hinted[1])


@pytest.mark.parametrize("case", ["absgrad", "five_channels", "tile8"])
def test_segment_hint_with_backward_fallbacks(G, case):
    """Configurations whose BACKWARD has no segment form (absgrad, more than four channels, tiles smaller than 16): the forward
    may run in segments, the backward takes the per-tile kernel on the segmented forward's alphas / last_ids - gradients must
    still agree with the plain pair of launches."""
    m2, con, op, off, fl, longest, W, H, tw, th = _lists(G, 40000, 1, 320, 192, 0.01, seed=31)
    D = 5 if case == "five_channels" else 3
    ts = 16
    if case == "tile8":
        # tile size 8 needs its own lists; the hint is then above the cut of THAT grid or not - either way results must agree
        ts = 8
        sc, _, _ = make_scene(N=40000, C=1, width=W, height=H, seed=31)
        a = {k: v.to(DEV) for k, v in sc.items()}
        a["means"][:, :2] *= 0.25
        a["opacities"] = torch.full_like(a["opacities"], 0.01)
        rad, m2, d, con, _ = G.fully_fused_projection(a["means"], None, a["quats"], a["scales"], a["viewmats"], a["Ks"], W, H,
                                                      opacities=a["opacities"])
        op = a["opacities"][None].contiguous()
        tw, th = math.ceil(W / ts), math.ceil(H / ts)
        _, ids, fl = G.isect_tiles(m2, rad, d, ts, tw, th, conics=con, opacities=op)
        off = G.isect_offset_encode(ids, 1, tw, th)
        counts = torch.diff(torch.cat([off.flatten(), torch.tensor([fl.numel()], device=DEV, dtype=off.dtype)]))
        longest = int(counts.max())
    g = torch.Generator().manual_seed(3)
    colors = torch.rand(m2.shape[:-1] + (D,), generator=g).to(DEV)
    w_c = torch.randn(1, H, W, D, generator=g).to(DEV)
    grads = {}
    for name, hint in (("tile", -1), ("seg", longest)):  # -1: per tile, whatever the intersection noted
        leaves = [t.detach().clone().requires_grad_(True) for t in (m2, con, colors, op)]
        rc, ra = G.rasterize_to_pixels(*leaves, W, H, ts, off, fl, absgrad=(case == "absgrad"), _longest_tile_list=hint)
        ((rc * w_c).sum() + ra.sum()).backward()
        grads[name] = [t.grad.cpu() for t in leaves] + ([leaves[0].absgrad.cpu()] if case == "absgrad" else [])
    for nm, x, y in zip(("means2d", "conics", "colors", "opacities", "absgrad"), grads["seg"], grads["tile"]):
        assert_grad_close(x, y, name=f"{case} v_{nm}")


def test_stage_level_callers_get_segments_without_a_hint(G):
    """isect_tiles -> isect_offset_encode -> rasterize_to_pixels driven by hand (the reference's stage API, no rasterization()
    in between): the intersection notes the longest list of its result under flatten_ids, the compositing ops look it up -
    forward AND backward take the segment kernels, bit-identical to the explicitly hinted call and not to the per-tile walk."""
    m2, con, op, off, fl, longest, W, H, tw, th = _lists(G, 40000, 1, 320, 192, 0.01, seed=41)
    colors = torch.rand(m2.shape[:-1] + (3,), generator=torch.Generator().manual_seed(2)).to(DEV)
    w_c = torch.randn(1, H, W, 3, generator=torch.Generator().manual_seed(3)).to(DEV)
    res = {}
    for name, kw in (("noted", {}), ("hinted", dict(_longest_tile_list=longest)), ("tile", dict(_longest_tile_list=-1))):
        leaves = [t.detach().clone().requires_grad_(True) for t in (m2, con, colors, op)]
        rc, ra = G.rasterize_to_pixels(*leaves, W, H, 16, off, fl, **kw)
        ((rc * w_c).sum() + ra.sum()).backward()
        res[name] = [rc.detach(), ra.detach()] + [t.grad for t in leaves]
    assert all(torch.equal(a, b) for a, b in zip(res["noted"][:2], res["hinted"][:2])), "forward did not take the segments"
    assert not torch.equal(res["noted"][0], res["tile"][0]), "segments and per-tile walk are expected to differ in rounding"
    # atomics make the backward's last bits vary from run to run: compare it to both at the segment tolerance instead
    for a, b in zip(res["noted"][2:], res["hinted"][2:]):
        assert_grad_close(a.cpu(), b.cpu(), name="noted vs hinted")
    # through the raw torch ops too (what the reference's Python calls)
    out = torch.ops.gsplat.rasterize_to_pixels_3dgs(m2, con, colors, op, None, None, W, H, 16, off, fl, False, False)
    assert torch.equal(out[0], res["hinted"][0])


@pytest.mark.parametrize("kind", ["faint", "opaque", "mixed"])
def test_segmented_backward_from_the_forward_workspace_equals_the_prepass(G, kind):
    """gsx_raster3d_bwd_seg_reuse: the forward op notes its segment workspace under the identity of the last_ids it returns;
    a backward op that receives THAT tensor starts its slices from the forward's per-slice sums (no pre-pass), one that
    receives a copy of it (another storage: no note) runs the pre-pass. Same gradients up to the association order of the
    "behind" sums; the workspace is only read, so a second backward over the same forward gives the same answer."""
    from gsplat_amd import _ops

    opacity = {"faint": 0.008, "opaque": 0.9, "mixed": (lambda o: torch.where(torch.rand_like(o) < 0.5, o * 0.02 + 0.004, o))}[kind]
    m2, con, op, off, fl, longest, W, H, tw, th = _lists(G, 40000, 2, 320, 192, opacity, seed=23)
    g = torch.Generator().manual_seed(77)
    colors = torch.rand(m2.shape[:-1] + (3,), generator=g).to(DEV)
    bg = torch.rand(2, 3, generator=g).to(DEV)
    w_c = torch.randn(2, H, W, 3, generator=g).to(DEV)
    w_a = torch.randn(2, H, W, 1, generator=g).to(DEV)
    _ops.set_long_tile_hint(longest)
    rc, ra, _, last_ids = torch.ops.gsplat.rasterize_to_pixels_3dgs(m2, con, colors, op, bg, None, W, H, 16, off, fl, False, False)
    _ops.set_long_tile_hint(0)
    ins = [m2, con, colors, op, off, fl]
    lookup = torch.ops.gsplat_amd.lookup_seg_workspace
    noted = lookup(last_ids, fl.numel(), 3, _ops.SEG_LEN, ins)
    assert noted is not None, "the forward did not note its segment workspace"
    assert lookup(last_ids.clone(), fl.numel(), 3, _ops.SEG_LEN, ins) is None
    assert lookup(last_ids, fl.numel() - 1, 3, _ops.SEG_LEN, ins) is None  # other lists
    # the sums belong to the forward's inputs: other tensors, or the same ones written to since, get no workspace
    assert lookup(last_ids, fl.numel(), 3, _ops.SEG_LEN, [m2, con, colors.clone(), op, off, fl]) is None
    op.mul_(1.0)
    assert lookup(last_ids, fl.numel(), 3, _ops.SEG_LEN, ins) is None, "an input was written to after the forward"
    _ops.set_long_tile_hint(longest)
    rc, ra, _, last_ids = torch.ops.gsplat.rasterize_to_pixels_3dgs(m2, con, colors, op, bg, None, W, H, 16, off, fl, False, False)
    _ops.set_long_tile_hint(0)
    noted = lookup(last_ids, fl.numel(), 3, _ops.SEG_LEN, ins)
    assert noted is not None
    before = noted.clone()

    def backward(li):
        _ops.set_long_tile_hint(longest)
        try:
            return torch.ops.gsplat.rasterize_to_pixels_3dgs_bwd(m2, con, colors, op, bg, None, off, fl, ra, li, W, H, 16, False,
                                                                 w_c, w_a, True)
        finally:
            _ops.set_long_tile_hint(0)

    reuse, again, prepass = backward(last_ids), backward(last_ids), backward(last_ids.clone())
    assert torch.equal(noted, before), "the backward wrote into the forward's workspace"
    for nm, a, b, c in zip(("abs", "means2d", "conics", "colors", "opacities", "backgrounds"), reuse, again, prepass):
        if a is None:
            assert b is None and c is None
            continue
        assert_grad_close(a.cpu(), c.cpu(), name=f"{kind} reuse vs pre-pass v_{nm}")
        assert_grad_close(a.cpu(), b.cpu(), name=f"{kind} reuse, second backward v_{nm}")
    del rc


def test_segmented_backward_on_the_one_wave_kernel():
    """GSX_RASTER3D_BWD_SEG=w (read once per process): the slices walked by variant W - half-length slices, short tiles
    longest-first - instead of the default four-wave kernel; the same comparisons as above, in a subprocess."""
    import os
    import subprocess
    import sys

    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GSX_RASTER3D_BWD_SEG="w")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_segments.py"), "-q", "-m", "gpu", "-x",
                        "-p", "no:cacheprovider", "-k", "segmented_backward_matches_per_tile_walk and (mixed or opaque) and not 1]"],
                       capture_output=True, text=True, env=env, timeout=900, cwd=root)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
