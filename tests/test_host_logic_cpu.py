"""Host-side logic of round 5 that needs no GPU: the sets / feature-source parser of the frame batch, the learning-rate segments
with a pattern inside a tensor, the library options' Python mirror."""
import pytest
import torch

from splatter_a_video_amd import frames as FR
from splatter_a_video_amd.optim import MAX_SEGMENTS, PatternLR, lr_segments


def test_parse_sets_flattens_feature_lists_and_marks_per_frame_tensors():
    F, P = 3, 10
    rgb, track, attrs = torch.rand(P, 3), torch.rand(F, P, 3), torch.rand(P, 16)
    sets = [dict(feature=rgb, bg=0.1, taps=True), dict(feature="depth", bg=1.0), dict(feature=[track, attrs], detach_opacity=True)]
    meta, parts, feats = FR._parse_sets(sets, F, P)
    assert meta == ((3, 0.1, False, True), ("depth", 1.0, False, False), (19, 0.0, True, False))
    assert parts == (((3, False),), (), ((3, True), (16, False)))
    assert [tuple(t.shape) for t in feats] == [(P, 3), (F, P, 3), (P, 16)]
    assert FR._has_sources(parts)
    assert not FR._has_sources((((3, False),), (), ((19, False),)))
    widths = [1 if m[0] == "depth" else m[0] for m in meta]
    plan = FR._one_pass_plan(meta, widths, 23)
    assert plan[0] == [0, 3, 4] and plan[1] == [3, 1, 19] and plan[3] == 3 and plan[4] == 0
    with pytest.raises(ValueError, match="per-frame feature tensor"):
        FR._parse_sets([dict(feature=torch.rand(F + 1, P, 3))], F, P)
    with pytest.raises(ValueError, match=r"\[P=10, c\]"):
        FR._parse_sets([dict(feature=torch.rand(P + 1, 3))], F, P)
    with pytest.raises(ValueError, match="depth"):
        FR._parse_sets([dict(feature="colour")], F, P)


def test_source_tensor_keeps_a_strided_per_frame_view_only_on_the_device():
    F, P = 2, 5
    pairs = torch.rand(F, 2, P, 3)
    with pytest.raises(ValueError, match="CUDA"):          # no CPU fallback: the gate is the usual device check
        FR._source_tensor(pairs[:, 1], True, F, P)


def test_lr_segments_carry_a_pattern_for_interleaved_groups():
    sl = {"cubic": (0, 120), "rotation": (120, 124), "shs": (124, 172), "attrs": (172, 188)}
    lrs = {"cubic": 6e-5, "rotation": 1e-3, "shs": PatternLR(1.25e-4, head_lr=2.5e-3, period=48, head=3), "attrs": 1e-3}
    ends, rates, pat = lr_segments(sl, lrs, patterns=True)
    assert ends == [120, 124, 172, 188] and rates == [6e-5, 1e-3, 1.25e-4, 1e-3]
    assert pat == [(0, 0, 0.0), (0, 0, 0.0), (48, 3, 2.5e-3), (0, 0, 0.0)]          # (a pattern group never merges with a neighbour)
    with pytest.raises(ValueError, match="pattern-aware"):
        lr_segments(sl, lrs)
    with pytest.raises(ValueError):
        PatternLR(1e-3, 1e-2, period=4, head=5)
    assert PatternLR(1e-3, 1e-2, 48, 3) == PatternLR(1e-3, 1e-2, 48, 3) != PatternLR(1e-3, 1e-2, 48, 4)
    many = {f"p{i}": (i, i + 1) for i in range(MAX_SEGMENTS + 1)}
    with pytest.raises(ValueError):
        lr_segments(many, {k: PatternLR(1e-3, 1e-2, 1, 1) for k in many}, patterns=True)


def test_library_options_through_the_abi(built_lib=None):
    """splat_set_option / splat_get_option need no GPU (process-wide integers read at launch time)"""
    from splatter_a_video_amd import _lib as L
    assert L.get_option("bwd_quarters") == 1 and L.get_option("sets_std") == 1 and L.get_option("bin_slot_keys") == 0
    with L.option("bwd_quarters", 0):
        assert L.get_option("bwd_quarters") == 0
    assert L.get_option("bwd_quarters") == 1
    L.set_option("deterministic", 1)
    assert L.deterministic()
    L.set_option("deterministic", 0)
    with pytest.raises(L.SplatError, match="unknown key"):
        L.set_option("no_such_option", 1)
    i3 = __import__("ctypes").c_int32 * 3
    q = L.lib().splat_blend_sets_uses_forward_pack
    assert q(L.ci(23), i3(0, 3, 4), i3(3, 1, 19), L.ci(1)) == 1 and q(L.ci(23), i3(0, 3, 4), i3(3, 1, 19), L.ci(0)) == 0
    assert q(L.ci(8), i3(0, 3, 4), i3(3, 1, 4), L.ci(1)) == 0
    with L.option("sets_std", 0):
        assert q(L.ci(23), i3(0, 3, 4), i3(3, 1, 19), L.ci(1)) == 0
