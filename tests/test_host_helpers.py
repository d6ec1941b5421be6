import os
"""CPU: host-side helpers and the algebra the kernels rely on (no kernels run here)."""
import numpy as np
import pytest
import torch


def test_world_to_cam_matches_definition():
    import gsplat_amd

    g = torch.Generator().manual_seed(0)
    means, covars = torch.randn(2, 7, 3, generator=g), torch.randn(2, 7, 3, 3, generator=g)
    covars = covars @ covars.transpose(-1, -2)
    viewmats = torch.randn(2, 3, 4, 4, generator=g)
    mc, cc = gsplat_amd.world_to_cam(means, covars, viewmats)
    assert mc.shape == (2, 3, 7, 3) and cc.shape == (2, 3, 7, 3, 3)
    R, t = viewmats[1, 2, :3, :3], viewmats[1, 2, :3, 3]
    assert torch.allclose(mc[1, 2, 4], R @ means[1, 4] + t, atol=1e-5)
    assert torch.allclose(cc[1, 2, 4], R @ covars[1, 4] @ R.T, atol=1e-4)


def test_feature_probes_follow_build_config():
    import gsplat_amd

    cfg = gsplat_amd.build_config()
    assert set(cfg) == {"3dgs", "2dgs", "3dgut", "adam", "reloc", "losses", "camera_wrappers"}  # ext.cpp:83-97
    assert gsplat_amd.has_3dgs() and gsplat_amd.has_2dgs() and gsplat_amd.has_adam() and gsplat_amd.has_reloc()
    # 3DGUT (UT projection + from-world rasterizer for every camera model incl. the spinning lidar, global / rolling shutter,
    # lidar tiling) is built: the flag is True like a full build of the reference; GSPLAT_AMD_3DGUT=0 switches it off
    assert gsplat_amd.has_3dgut() and not gsplat_amd.has_losses() and not gsplat_amd.has_camera_wrappers()
    from gsplat_amd import csrc_shim

    assert csrc_shim.built_3dgut_subset()
    os.environ["GSPLAT_AMD_3DGUT"] = "0"
    try:
        assert csrc_shim.build_config()["3dgut"] is False
    finally:
        del os.environ["GSPLAT_AMD_3DGUT"]


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_tile_origin_moments_give_the_mean_relative_moments(seed):
    """The compositing backward (variant T, csrc/raster3d_bwd.hip) sums w, w u, w v, w u^2, w u v, w v^2 over the pixels of a
    tile with (u, v) measured from the TILE origin, and turns them into the moments of d = mean - pixel = a - (u, v) once per
    (tile, Gaussian). This is that conversion, in float32 like the kernel, against the direct sums in float64."""
    rng = np.random.default_rng(seed)
    u, v = np.meshgrid(np.arange(16, dtype=np.float32), np.arange(16, dtype=np.float32))
    w = rng.normal(size=(16, 16)).astype(np.float32) * np.exp(-rng.uniform(0, 4, size=(16, 16))).astype(np.float32)
    for ax, ay in ((3.7, 9.2), (-250.5, 40.25), (700.0, -320.0)):
        ax32, ay32 = np.float32(ax), np.float32(ay)
        S0, Su, Sv = w.sum(dtype=np.float32), (w * u).sum(dtype=np.float32), (w * v).sum(dtype=np.float32)
        Suu, Suv, Svv = (w * u * u).sum(dtype=np.float32), (w * u * v).sum(dtype=np.float32), (w * v * v).sum(dtype=np.float32)
        got = dict(x=ax32 * S0 - Su, y=ay32 * S0 - Sv, xx=ax32 * (ax32 * S0 - np.float32(2) * Su) + Suu,
                   xy=ax32 * (ay32 * S0 - Sv) - ay32 * Su + Suv, yy=ay32 * (ay32 * S0 - np.float32(2) * Sv) + Svv)
        w64, dx, dy = w.astype(np.float64), ax - u.astype(np.float64), ay - v.astype(np.float64)
        want = dict(x=(w64 * dx).sum(), y=(w64 * dy).sum(), xx=(w64 * dx * dx).sum(), xy=(w64 * dx * dy).sum(),
                    yy=(w64 * dy * dy).sum())
        scale = dict(x=(np.abs(w64) * np.abs(dx)).sum(), y=(np.abs(w64) * np.abs(dy)).sum(),
                     xx=(np.abs(w64) * dx * dx).sum(), xy=(np.abs(w64) * np.abs(dx * dy)).sum(),
                     yy=(np.abs(w64) * dy * dy).sum())
        for k in want:  # error relative to the size of the terms that are summed (what an fp32 direct sum would also see)
            assert abs(float(got[k]) - want[k]) <= 2e-5 * scale[k] + 1e-6, (k, ax, ay, float(got[k]), want[k])


def test_sort_words_order_like_doubles_where_the_kernels_say_so():
    """csrc/bitonic64.hpp sorts the 64-bit words (depth bits << 32 | row) with v_min_f64 / v_max_f64: a word whose high dword
    lies in [0x00100000, 0x7FF00000) is a positive normal double, and those order exactly like their bit patterns; negating
    both words (the network's descending blocks) reverses the order; +inf (the pad) is behind every such word. Everything
    else (bt_key_is_odd) takes the integer network. This is that claim, on the edges and on random words."""
    lo_edge, hi_edge = 0x00100000, 0x7FF00000

    def is_odd(depth_bits):  # bt_key_is_odd
        return ((depth_bits - lo_edge) & 0xFFFFFFFF) >= (hi_edge - lo_edge)

    for bits, odd in ((0x00000000, True), (0x000FFFFF, True), (0x00100000, False), (0x3F800000, False), (0x7F7FFFFF, False),
                      (0x7F800000, False), (0x7FC00000, False), (0x7FEFFFFF, False), (0x7FF00000, True), (0x80000000, True),
                      (0xBF800000, True), (0xFFFFFFFF, True)):
        assert is_odd(bits) == odd, hex(bits)
    rng = np.random.default_rng(0)
    hi = rng.integers(lo_edge, hi_edge, size=20000, dtype=np.uint64)
    hi[:6] = [lo_edge, lo_edge, hi_edge - 1, hi_edge - 1, 0x3F800000, 0x3F800000]  # ties on the depth: the row decides
    lo = rng.integers(0, 1 << 32, size=20000, dtype=np.uint64)
    words = np.unique((hi << np.uint64(32)) | lo)
    as_f64 = words.view(np.float64)
    assert np.all(np.isfinite(as_f64)) and np.all(as_f64 > 0)
    assert np.all(np.diff(as_f64) > 0), "ascending words are ascending doubles (np.unique sorted the integers)"
    assert np.all(np.diff((words ^ np.uint64(1 << 63)).view(np.float64)) < 0), "negated: descending"
    assert np.uint64(0x7FF0000000000000).view(np.float64) == np.inf and np.all(as_f64 < np.inf)
    pair = np.array([words[5], words[4]])
    assert np.minimum(*pair.view(np.float64)).view(np.uint64) == words[4], "min returns an operand's bits unchanged"


def test_pixel_linear_strides_recognises_the_views_autograd_hands_over():
    """_ops._pixel_linear_strides: which v_render_colors layouts the compositing backward reads in place."""
    from gsplat_amd._ops import _pixel_linear_strides as f

    I, H, W, D = 2, 5, 7, 3
    assert f(torch.zeros(I, H, W, D)) is None  # contiguous: the plain path
    assert f(torch.tensor(1.0).expand(I, H, W, D)) == (0, 0)  # gradient of sum()
    assert f(torch.zeros(D).expand(I, H, W, D)) == (0, 1)  # one value per channel
    assert f(torch.zeros(I, H, W, 5)[..., 1:4]) == (5, 1)  # channels of a wider image
    assert f(torch.zeros(I, H, W, 1).expand(I, H, W, D)) == (1, 0)  # one value per pixel
    assert f(torch.zeros(I, W, H, D).transpose(1, 2)) is None  # not linear in (i H + y) W + x: copied
    assert f(torch.zeros(I, H, 2 * W, D)[:, :, ::2]) == (6, 1)  # every other column of a 2 W image: still linear in the pixel
    assert f(torch.zeros(I, H, W + 2, D)[:, :, :W]) is None  # padded rows: row stride != W * pixel stride
    assert f(torch.zeros(1, H, W, D)[0].expand(I, H, W, D)) is None  # images on top of each other
    assert f(torch.zeros(H, W, 6)[..., ::2]) == (6, 2)  # no image dimension


@pytest.mark.parametrize("tool", ["check_binwalk", "check_spans"])
def test_tile_walk_host_checks(tool, tmp_path):
    """The walk headers are host + device code: tools/check_binwalk.cpp (clipped walks over a partition of the grid = the walk)
    and tools/check_spans.cpp (a row's 16-byte span record replays as the walk, or says it does not fit) are compiled for the
    HOST and run over 300 k random and adversarial Gaussians."""
    import os
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / tool)
    subprocess.run([hipcc, "-O2", "-std=c++17", "-ffp-contract=off", "-x", "hip", "--cuda-host-only", "-w", "-o", exe,
                    os.path.join(root, "tools", tool + ".cpp")], check=True, timeout=600)
    r = subprocess.run([exe, "300000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and " bad 0" in r.stdout, r.stdout + r.stderr


def test_elem_view_reads_single_columns_in_place():
    """_ops._elem_view: a contiguous tensor or ONE column of an array-of-structures buffer is handed to the kernels in place
    (with its element stride); anything else is made contiguous."""
    import torch
    from gsplat_amd._ops import _elem_view

    rows = torch.arange(5 * 9, dtype=torch.float32).reshape(5, 9)
    t, st = _elem_view(rows[:, 5])
    assert st == 9 and t.data_ptr() == rows[:, 5].data_ptr()
    t, st = _elem_view(rows[:, 5].view(1, 5))  # [C, N] over the same column
    assert st == 9 and t.data_ptr() == rows[:, 5].data_ptr()
    big = torch.arange(2 * 3 * 9, dtype=torch.float32).reshape(6, 9)
    t, st = _elem_view(big[:, 4].view(2, 3))  # two views, rows view-major
    assert st == 9
    t, st = _elem_view(torch.zeros(4, 3))
    assert st == 1
    t, st = _elem_view(torch.zeros(3, 4).t())  # not one uniform stride
    assert st == 1 and t.is_contiguous()
    t, st = _elem_view(torch.zeros(1).expand(4))  # stride 0: a copy
    assert st == 1 and t.is_contiguous() and t.numel() == 4


def test_longest_list_notes_python_fallback_is_keyed_by_storage_and_view():
    """`_ops._note_longest`'s Python fallback (compiled shim absent, or GSPLAT_AMD_LIB set) keys a note by the StorageImpl's
    address + the view (offset, numel) and keeps the noted tensor alive in its 16-entry ring, so the address cannot be handed
    out again while the note exists and nothing depends on torch preserving a storage's Python wrapper between calls."""
    import torch

    from gsplat_amd import _ops

    was = _ops._notes_compiled
    _ops._notes_compiled = False
    try:
        t = torch.arange(16, dtype=torch.int32)
        _ops._note_longest(t, 7)
        assert _ops._lookup_longest(t) == 7 and _ops._lookup_longest(t.view(16)) == 7
        assert _ops._lookup_longest(t[2:]) == 0 and _ops._lookup_longest(torch.arange(16, dtype=torch.int32)) == 0
        _ops._note_longest(t, 9)  # a newer note of the same view replaces the old one
        assert _ops._lookup_longest(t) == 9
        for i in range(20):  # the ring holds 16 notes
            _ops._note_longest(torch.zeros(4 + i, dtype=torch.int32), i)
        assert len(_ops._notes_py) == 16 and _ops._lookup_longest(t) == 0
    finally:
        _ops._notes_py.clear()
        _ops._notes_compiled = was


def test_intersection_path_memory_is_per_caller_and_per_thread():
    """Which intersection kernel runs depends on retry notes (a clustered scene that was sent back is not tried again for 63
    calls). The notes belong to a caller-owned object or to the calling THREAD's private default - never to the process
    (include/gsplat_amd.h: gsx_isect_path_memory_*)."""
    import threading

    import gsplat_amd
    from gsplat_amd import _cabi

    L = _cabi._lib
    shape = (1_000_000, 1, 120, 68)
    assert L.gsx_isect_binned_should_try(*shape, 0) == 1
    mine, other = gsplat_amd.IsectPathMemory(), gsplat_amd.IsectPathMemory()
    with mine:
        L.gsx_isect_binned_note_retry(*shape)
        assert L.gsx_isect_binned_supported(*shape, 0) == 0  # this caller was sent back: skip the attempt
        with other:
            assert L.gsx_isect_binned_supported(*shape, 0) == 1  # another caller's history is its own
        assert L.gsx_isect_binned_supported(*shape, 0) == 0
    assert L.gsx_isect_binned_supported(*shape, 0) == 1  # the thread's default never saw the retry
    L.gsx_isect_binned_note_retry(*shape)  # ... until it is sent back itself
    assert L.gsx_isect_binned_supported(*shape, 0) == 0
    seen = []
    t = threading.Thread(target=lambda: seen.append(L.gsx_isect_binned_supported(*shape, 0)))
    t.start()
    t.join()
    assert seen == [1]  # another thread has its own default
    skipped = sum(1 for _ in range(70) if L.gsx_isect_binned_should_try(*shape, 0) == 0)
    assert skipped == 63  # 63 intersections skip the attempt, the 64th probes again
