"""Scenarios of the frame-loop pin: the reference's own CoFusion::processFrame text (tests/refcofusion.py) and the
oracle's restatement of it (tests/orc_multi.MultiPipeline) play the same synthetic stream; everything the frame loop decides -- model
list, ids, poses, surfel buffers, confidence thresholds, unseen counters, label masks -- is digested per frame.  Test infrastructure."""
from __future__ import annotations

import hashlib

import numpy as np

W, H = 160, 128
# name -> (objects in the scene, frames, conf_global, spawn offset, ground-truth masks, multiple models)
SCENARIOS = {
    "crf_two_objects": (2, 20, 0.5, 3, False, True),
    "crf_four_objects": (4, 26, 0.5, 1, False, True),
    "gt_masks_three_objects": (3, 10, 0.5, 2, True, True),
    "gt_masks_fill_in_tracking": (3, 8, 10.0, 2, True, True),
    "static": (2, 5, 10.0, 20, False, False),
    # ground-truth odometry (processFrame's inPose, CoFusion.cpp:341-343) on frames 2-4 and 7, tracking on the others
    "static_injected_poses": (0, 9, 0.5, 20, False, False),
}
INJECTED = {"static_injected_poses": (2, 3, 4, 7)}
# tracking switches of the frame loop (CoFusion::setFrameToFrameRGB / setFastOdom / setSo3 / setPyramid / setIcpWeight): these scenarios
# compare the reference's loop with the C++ FACADE (tests/test_configs_gpu.py); the oracle's Python loop does not have the switches
SCENARIOS.update({
    "frame_to_frame_rgb": (2, 10, 10.0, 3, False, True),
    "fast_odom_no_so3_no_pyramid": (2, 10, 0.5, 3, False, True),
    "icp_only": (1, 8, 0.5, 3, False, True),
})
OPTIONS = {
    "frame_to_frame_rgb": dict(frame_to_frame_rgb=True),
    "fast_odom_no_so3_no_pyramid": dict(fast_odom=True, so3=False, pyramid=False),
    "icp_only": dict(icp_weight=100.0),
}
FACADE_ONLY = set(OPTIONS)
# CoFusion's `reloc` constructor argument (-rl; CoFusion.cpp:225, 301-338, 463, 495): frames whose background tracking is out of bounds
# are not fused, after more than ten in a row the camera is lost (no fusion, the clock stops).  BLACKOUT: frames in which the sensor is
# covered (no depth, black image) -- the tracker finds no correspondence, lastICPError is NaN, the covariance of a zero matrix is NaN / inf.
SCENARIOS.update({
    "reloc_static_lost": (0, 22, 0.5, 20, False, False),    # 13 covered frames: lost from the 11th on, and for good (no fern database)
    "reloc_static_recovers": (0, 14, 0.5, 20, False, False),  # 4 covered frames: skipped, then tracking and fusion resume
    "reloc_two_objects": (2, 14, 0.5, 3, False, True),        # the switch on a multi-model run
})
RELOC = {"reloc_static_lost", "reloc_static_recovers", "reloc_two_objects"}
BLACKOUT = {"reloc_static_lost": range(5, 18), "reloc_static_recovers": range(5, 9), "reloc_two_objects": range(9, 11)}


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:24]


def frames_of(name):
    from co_fusion_amd import synth
    n_obj, n_frames, _, _, use_gt, _ = SCENARIOS[name]
    cam = synth.Camera.scaled(W, H)
    sc = synth.Scene(n_obj=n_obj)
    out = []
    for t in range(n_frames):
        d, rgb, label, _ = sc.render(cam, t, noise=True)
        if t in BLACKOUT.get(name, ()):
            d = np.zeros_like(d); rgb = np.zeros_like(rgb)
        out.append((d, rgb, synth.rgb_to_rgba(rgb), (label * 40).astype(np.uint8) if use_gt else None))
    return cam, out


def injected_pose(name, t):
    """the generator's camera pose for the frames a scenario hands to processFrame as inPose, else None"""
    if t not in INJECTED.get(name, ()):
        return None
    from co_fusion_amd import synth
    return synth.Scene(n_obj=SCENARIOS[name][0]).camera_pose(t).astype(np.float32)


def run_reference(name):
    import refcofusion
    _, _, conf_global, spawn, _, multi = SCENARIOS[name]
    cam, frames = frames_of(name)
    cf = refcofusion.RefCoFusion(cam, conf_global=conf_global, spawn_offset=spawn, multi=multi, reloc=name in RELOC, **OPTIONS.get(name, {}))
    rows = []
    for t, (d, rgb, _, gt) in enumerate(frames):
        cf.process_frame(d, rgb, gt_mask=gt, timestamp=t, in_pose=injected_pose(name, t))
        ms = [cf.model(i) for i in range(cf.num_models)]
        rows.append(dict(ids=[m["id"] for m in ms], counts=[m["count"] for m in ms], poses=[_sha(m["pose"]) for m in ms],
                         surfels=[_sha(m["surfels"]) for m in ms], conf=[float(m["conf_threshold"]) for m in ms],
                         unseen=[m["unseen"] for m in ms], mask=_sha(cf.mask()), tick=cf.tick, lost=int(cf.lost),
                         pose_log=[m["last_pose_log"].copy() for m in ms]))
    return rows


def run_oracle(name):
    import orc_multi as om
    _, _, conf_global, spawn, _, multi = SCENARIOS[name]
    cam, frames = frames_of(name)
    rows = []
    if multi:
        cf = om.MultiPipeline(cam, conf_global=conf_global, spawn_offset=spawn, reloc=name in RELOC)
        for d, _, rgba, gt in frames:
            cf.process_frame(d, rgba, gt_mask=gt)
            ms = cf.models
            rows.append(dict(ids=[m.id for m in ms], counts=[m.surfels.shape[0] for m in ms], poses=[_sha(m.pose) for m in ms],
                             surfels=[_sha(m.surfels) for m in ms], conf=[float(np.float32(m.conf_threshold)) for m in ms],
                             unseen=[m.unseen for m in ms], mask=_sha(cf.mask), tick=cf.tick, lost=int(cf.reloc.lost)))
    else:
        import orc_pipeline as op
        cf = op.StaticPipeline(cam, conf_global=conf_global, reloc=name in RELOC)
        for t, (d, _, rgba, _) in enumerate(frames):
            cf.process_frame(d, rgba, in_pose=injected_pose(name, t))
            rows.append(dict(ids=[0], counts=[cf.surfels.shape[0]], poses=[_sha(cf.pose)], surfels=[_sha(cf.surfels)],
                             conf=[float(np.float32(conf_global))], unseen=[0], mask=_sha(cf.mask), tick=cf.tick, lost=int(cf.reloc.lost)))
    return rows


# The confidence threshold of an object model follows SegmentationResult::avgConfidence, which the reference's Segmentation computes from f32
# running sums per superpixel and the oracle from exact sums (pinned to 2e-5 on its own, ref_seg_v1.npz): a tolerance, everything else bits
CONF_RTOL = 1e-4


def run_reference_isolated(name, glpin_mutation=0):
    """run_reference in a process of its own.  Core/Segmentation/Segmentation.cpp:64 keeps the ground-truth label -> model id table
    in a FUNCTION-STATIC vector and indexes an uninitialised `modelIdToIndex[256]` with whatever it finds there: a second CoFusion
    instance in the same process inherits the first one's table, writes labels of models it does not have and then runs off the
    end of `modelData` (found by AddressSanitizer while building this pin).  The reference program has one instance per process."""
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r); import cfpin; rows = cfpin.run_reference(%r); "
            "[r.__setitem__('pose_log', [list(map(float, p)) for p in r['pose_log']]) for r in rows]; print('CFPIN_JSON' + json.dumps(rows))"
            % (here, os.path.dirname(here), name))
    env = dict(os.environ)
    env.pop("COFUSION_GLPIN_MUTATE", None)
    if glpin_mutation:   # corrupt the recorded OpenGL state behind the pasted Model::fuse / Model::clean text (ref_cofusion.cpp)
        env["COFUSION_GLPIN_MUTATE"] = str(int(glpin_mutation))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True, env=env).stdout
    line = [l for l in out.split("\n") if l.startswith("CFPIN_JSON")][-1]
    return json.loads(line[len("CFPIN_JSON"):])


def differences(ref_rows, orc_rows, keys=("ids", "counts", "poses", "surfels", "unseen", "mask", "tick", "lost")):
    out = []
    for t, (a, b) in enumerate(zip(ref_rows, orc_rows)):
        for k in keys:
            if a[k] != b[k]:
                out.append(f"frame {t}: {k}: reference {a[k]} vs oracle {b[k]}")
        if len(a["conf"]) != len(b["conf"]) or any(abs(x - y) > CONF_RTOL * max(abs(x), 1e-3) for x, y in zip(a["conf"], b["conf"])):
            out.append(f"frame {t}: conf: reference {a['conf']} vs oracle {b['conf']}")
    if len(ref_rows) != len(orc_rows):
        out.append("different number of frames")
    return out
