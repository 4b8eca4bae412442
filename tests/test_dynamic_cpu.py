"""Row a15 (per-frame dynamic Gaussian evaluation): the C oracle pinned to vectors produced by the reference's own
get_position / get_rotation / get_opacity / get_scaling (tests/golden/make_golden_dynamic.py)."""
import os

import numpy as np
import pytest

import oracle

G = os.path.join(os.path.dirname(__file__), "golden", "dynamic_400x50.npz")


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(G))


def _scalars(g, t):
    return oracle.dynamic_time_scalars(int(t), int(g["T"]), g["intervals"], int(g["start_frame_id"]), int(g["time_len"]))


def test_segment_lookup_matches_reference_knots(gold):
    # frames that sit exactly on a knot belong to the segment that ENDS there (searchsorted(nt-1e-7) - 1), frame 0 to segment 0
    T, I = int(gold["T"]), int(gold["I"])
    knots = np.linspace(0, T - 1, I + 1).astype(np.int64)
    for t in range(T):
        seg, d, _, _ = _scalars(gold, t)
        want = 0 if t == 0 else int(np.searchsorted(knots, t, side="left")) - 1
        assert seg == want, (t, seg, want)
        assert 0.0 <= d <= 1.0


@pytest.mark.parametrize("t", [0, 1, 5, 24, 25, 44, 45, 49])
def test_forward_matches_reference(gold, t):
    g = gold
    seg, d, poly, fourier = _scalars(g, t)
    pos, rot, opa, scl = oracle.dynamic_eval_forward(g["position"], g["pos_cubic_node"], g["rotation"], g["rot_poly_feat"],
                                                     g["rot_fourier_feat"], g["opacity"], g["scaling"], seg, d, poly, fourier)
    pre = f"t{t}_"
    np.testing.assert_allclose(pos, g[pre + "pos"], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(rot, g[pre + "rot"], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(opa, g[pre + "opa"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(scl, g[pre + "scl"], rtol=2e-6, atol=1e-9)


@pytest.mark.parametrize("t", [0, 1, 5, 24, 25, 44, 45, 49])
def test_backward_matches_reference(gold, t):
    g = gold
    N, I = g["position"].shape[0], int(g["I"])
    seg, d, poly, fourier = _scalars(g, t)
    pre = f"t{t}_"
    dpos, dcub, drot, dopa, dscl = oracle.dynamic_eval_backward(
        (N, 4, I, 3), g["rotation"], g["rot_poly_feat"], g["rot_fourier_feat"], g["opacity"], g["scaling"], seg, d, poly,
        fourier, g[pre + "g_pos"], g[pre + "g_rot"], g[pre + "g_opa"], g[pre + "g_scl"])
    np.testing.assert_allclose(dpos, g[pre + "d_position"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(dcub.reshape(N, -1), g[pre + "d_cubic"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(drot, g[pre + "d_rotation"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(dopa, g[pre + "d_opacity"], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(dscl, g[pre + "d_scaling"], rtol=2e-6, atol=1e-9)
    # only the active segment of the spline table receives gradient
    mask = np.ones(I, bool); mask[seg] = False
    assert not dcub[:, :, mask, :].any()


def test_frame_clock_matches_oracle_scalars(gold):
    """host logic of the product path (splatter_a_video_amd.dynamics.FrameClock) against the oracle's restatement"""
    from splatter_a_video_amd.dynamics import FrameClock
    T = int(gold["T"])
    clock = FrameClock(T)                                   # builds the knots the way the reference does
    np.testing.assert_array_equal(clock.intervals, gold["intervals"])
    assert clock.interval_num == int(gold["I"])
    for t in range(T):
        seg, d, basis = clock.scalars(t)
        oseg, od, opoly, ofour = _scalars(gold, t)
        assert seg == oseg
        assert np.float32(d) == np.float32(od)
        np.testing.assert_array_equal(np.array(list(basis), np.float32), np.concatenate([opoly, ofour]))
    for T2 in (2, 7, 11, 250):
        c = FrameClock(T2)
        assert c.intervals[0] == 0.0 and c.intervals[-1] == 1.0 and np.all(np.diff(c.intervals) > 0)
        for t in range(T2):
            seg, d, _ = c.scalars(t)
            assert 0 <= seg < c.interval_num and d >= 0.0


def test_dynamic_eval_fails_loudly_without_gpu():
    import torch
    from splatter_a_video_amd.dynamics import FrameClock, evaluate
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises((ValueError, RuntimeError)):
        evaluate(FrameClock(10), 3, opacity=torch.zeros(4, 1))


def test_poly_fourier_position_restatement_matches_reference():
    """the second dynamic point cloud's position model (src/dynamic_gaussian_points.py:169-186): numpy restatement against
    the vectors of the reference's own get_position (tests/golden/make_golden_polyfourier.py), values and gradients"""
    import os
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "polyfourier_300x40.npz")))
    for t in g["times"]:
        b = oracle.time_basis(int(t), int(g["start_frame_id"]), int(g["time_len"]))
        pos = oracle.position_poly_fourier_forward(g["position"], g["pos_poly_feat"], g["pos_fourier_feat"], b)
        np.testing.assert_allclose(pos, g[f"t{t}_pos"], rtol=2e-6, atol=2e-6)
        dp, dpoly, dfour = oracle.position_poly_fourier_backward(g[f"t{t}_g_pos"], b)
        np.testing.assert_allclose(dp, g[f"t{t}_d_position"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(dpoly, g[f"t{t}_d_pos_poly"], rtol=2e-6, atol=1e-6)
        np.testing.assert_allclose(dfour, g[f"t{t}_d_pos_fourier"], rtol=2e-6, atol=1e-6)
        assert float(np.abs(g[f"t{t}_det_d_position"]).max()) == 0.0        # detach_pos: no gradient to the base position


def test_frame_clock_default_knots_follow_torch_linspace():
    """the default knots are the reference's float32 torch.linspace(...).long() for every clip length"""
    import math
    import torch
    from splatter_a_video_amd.dynamics import FrameClock
    for n in range(2, 400):
        k = math.ceil(n / 5)
        want = torch.linspace(0, n - 1, k + 1).long().numpy().astype(np.float32) / np.float32(n - 1)
        np.testing.assert_array_equal(FrameClock(n).intervals, want)
