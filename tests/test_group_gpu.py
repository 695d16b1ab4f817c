"""GPU suite: a lock-step GROUP of sequences (cofusion_group_*, host/CoFusion.cpp CoFusionGroup) -- several independent RGB-D sequences
on one GPU sharing one context and one set of tracking launches -- gives every sequence the results of a CoFusion instance of its own,
bit for bit: poses, model lists, label masks, surfel buffers, frame after frame."""
import warnings

import numpy as np
import pytest

from co_fusion_amd import synth

pytestmark = pytest.mark.gpu
warnings.filterwarnings("ignore", category=RuntimeWarning)


def _compare(group, singles, t, what):
    for s, (a, b) in enumerate(zip(group.sequences, singles)):
        assert a.num_models == b.num_models, f"{what} frame {t} sequence {s}: {a.num_models} vs {b.num_models} models"
        if t > 0:
            assert np.array_equal(a.mask(), b.mask()), f"{what} frame {t} sequence {s}: label mask"
        for i in range(b.num_models):
            x, y = a.model_info(i), b.model_info(i)
            assert x["id"] == y["id"] and x["count"] == y["count"], f"{what} frame {t} sequence {s} model {i}: id / count"
            assert x["pose"].tobytes() == y["pose"].tobytes(), f"{what} frame {t} sequence {s} model {i}: pose"
            assert x["conf_threshold"] == y["conf_threshold"]
            assert a.model_download(i).tobytes() == b.model_download(i).tobytes(), f"{what} frame {t} sequence {s} model {i}: surfels"


def _run(S, W, H, n_obj, frames, use_gt=False, device_frames=False, crf=None, **kw):
    import torch
    from co_fusion_amd import facade
    cam = synth.Camera.scaled(W, H)
    scenes = [synth.Scene(n_obj=n_obj, seed=1234 + 17 * s) for s in range(S)]
    group = facade.CoFusionGroup(S, W, H, cam.fx, cam.fy, cam.cx, cam.cy, **kw)
    singles = [facade.CoFusion(W, H, cam.fx, cam.fy, cam.cx, cam.cy, **kw) for _ in range(S)]
    for s, c in enumerate(crf or []):   # per-sequence CRF settings (cofusion_set_crf on the borrowed handle and on the separate instance)
        if c:
            group.sequences[s].set_crf(**c); singles[s].set_crf(**c)
    most = 0
    for t in range(frames):
        rendered = [sc.render(cam, t, noise=True) for sc in scenes]
        depths = [r[0] for r in rendered]; rgbs = [r[1] for r in rendered]
        masks = [(r[2] * 40).astype(np.uint8) for r in rendered] if use_gt else None
        if device_frames:
            dts = [torch.from_numpy(d).cuda() for d in depths]
            cts = [torch.from_numpy(synth.rgb_to_rgba(c)).cuda() for c in rgbs]
            group.process_frames_device(dts, cts, timestamp=t)
            for s in range(S):
                singles[s].process_frame_device(dts[s], cts[s], timestamp=t)
        else:
            group.process_frames(depths, rgbs, masks, timestamp=t)
            for s in range(S):
                singles[s].process_frame(depths[s], rgbs[s], mask=None if masks is None else masks[s], timestamp=t)
        _compare(group, singles, t, f"{S} x {n_obj} objects")
        most = max(most, max(q.num_models for q in singles))
    # a sequence of a group is stepped by the group only: its borrowed handle refuses a frame of its own
    with pytest.raises(facade.CoFusionError, match="lock-step group"):
        group.sequences[0].process_frame(depths[0], rgbs[0], timestamp=frames)
    group.close()
    for q in singles:
        q.close()
    return most


def test_static_sequences_in_lockstep_match_separate_instances():
    """configs[1] x 4: four -static sequences (different scenes) through ONE set of tracking launches per frame"""
    _run(4, 320, 240, 0, 8, enable_multiple_models=0, max_surfels=1 << 19)


def test_static_sequences_device_frames():
    """the device-resident entry point (frames already in HBM), 640x480"""
    _run(3, 640, 480, 0, 4, device_frames=True, enable_multiple_models=0, max_surfels=1 << 21, device_frames_complete=1)


def test_multi_object_sequences_in_lockstep_match_separate_instances():
    """two multi-object sequences with the motion CRF: spawning happens at different frames in the two sequences, each keeps its own
    segmentation and model list"""
    most = _run(2, 320, 240, 3, 12, conf_global_init=0.5, model_spawn_offset=2, enable_multiple_models=1, max_surfels=1 << 19)
    assert most >= 2, "no object model was spawned"


def test_sequences_with_different_crf_settings_keep_their_own():
    """the group's batched segmentation chain took sequence 0's CRF parameters for everybody.  Two sequences whose settings
    differ (weights, new-label threshold, 6 instead of 10 mean-field steps) still equal separate instances bit for bit: sequences with
    different settings get chains of their own"""
    most = _run(2, 320, 240, 3, 12, conf_global_init=0.5, model_spawn_offset=2, enable_multiple_models=1, max_surfels=1 << 19,
                crf=[None, dict(weight_appearance=5.0, weight_smoothness=3.0, threshold_new=4.5, iterations=6)])
    assert most >= 2, "no object model was spawned"


def test_more_trackers_than_one_launch_holds():
    """five sequences x (background + 3 objects from ground-truth masks) = 20 trackers > 16 per lock-step launch (kMaxBatch): the
    tracking launches are issued in two chunks"""
    most = _run(5, 160, 128, 3, 5, use_gt=True, conf_global_init=0.5, model_spawn_offset=1, enable_multiple_models=1, max_surfels=1 << 18)
    assert most >= 4
