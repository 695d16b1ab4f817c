"""ctypes binding of the C++ facade (libcofusion.so, include/cofusion.h): the reference's
`CoFusion` object as GUI/MainController.cpp drives it (construct, processFrame, getters)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import lib as _libmod
from .api import Profile


class Config(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("device", C.c_int), ("max_surfels", C.c_int), ("max_models", C.c_int), ("conf_global_init", C.c_float),
                ("conf_object_init", C.c_float), ("depth_cutoff", C.c_float), ("icp_weight", C.c_float),
                ("outlier_coefficient", C.c_float), ("fast_odom", C.c_int), ("so3", C.c_int), ("frame_to_frame_rgb", C.c_int),
                ("pyramid", C.c_int), ("rgb_only", C.c_int), ("model_spawn_offset", C.c_uint), ("enable_multiple_models", C.c_int),
                ("enable_pose_logging", C.c_int), ("rank", C.c_int), ("world", C.c_int), ("device_frames_complete", C.c_int),
                ("mid_frame_predict", C.c_int), ("shard_background", C.c_int), ("enqueue_threads", C.c_int),
                ("colocate_background", C.c_int), ("reloc", C.c_int), ("early_index_maps", C.c_int)]


class CoFusionError(RuntimeError):
    pass


class _DevWords:
    """a device address as an int64 array (the __cuda_array_interface__ protocol): torch.as_tensor turns it into a tensor VIEW of the
    library's buffer, so that torch.distributed reduces it in place -- no staging copies"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = dict(shape=(int(n),), typestr="<i8", data=(int(ptr), False), version=2)


def rccl_unique_id():
    """ncclGetUniqueId through the facade (128 bytes): rank 0 creates it, every rank passes it to CoFusion.init_rccl"""
    lib = _libmod.load_host()
    buf = (C.c_ubyte * 128)()
    if lib.cofusion_rccl_unique_id(buf) != 0:
        raise CoFusionError(f"cofusion_rccl_unique_id: {lib.cofusion_last_error().decode()}")
    return bytes(buf)


def _make_config(lib, width, height, fx, fy, cx, cy, device, kw):
    cfg = Config()
    lib.cofusion_default_config(C.byref(cfg))
    cfg.width, cfg.height, cfg.fx, cfg.fy, cfg.cx, cfg.cy, cfg.device = width, height, fx, fy, cx, cy, device
    for k, v in kw.items():
        if not hasattr(cfg, k):
            raise TypeError(f"unknown CoFusion option {k}")
        setattr(cfg, k, v)
    return cfg


class CoFusion:
    def __init__(self, width=640, height=480, fx=528.0, fy=528.0, cx=320.0, cy=240.0, device=0, _borrowed=None, **kw):
        if not torch.cuda.is_available():
            raise CoFusionError("no GPU visible: the Co-Fusion hot path has no CPU fallback")
        self.lib = _libmod.load_host()
        self.abi = _libmod.load()
        cfg = _make_config(self.lib, width, height, fx, fy, cx, cy, device, kw)
        self.cfg = cfg
        self.width, self.height = width, height
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self._owned = _borrowed is None
        if _borrowed is not None:   # a sequence of a CoFusionGroup: the handle belongs to the group
            self.h = C.c_void_p(_borrowed)
            return
        self.h = C.c_void_p()
        self._check(self.lib.cofusion_create(C.byref(cfg), C.byref(self.h)))
        self._check(self.lib.cofusion_set_stream(self.h, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    def _check(self, rc):
        if rc != 0:
            raise CoFusionError(f"cofusion error {rc}: {self.lib.cofusion_last_error().decode()}")

    def init_rccl(self, unique_id=None):
        """The library's own RCCL communicator as this instance's collective (cofusion_init_rccl): ncclAllReduce / ncclBroadcast in
        place on the context's stream, no Python in the frame loop.  unique_id: the 128 bytes rank 0 got from rccl_unique_id(); None:
        created on rank 0 and distributed through the default torch.distributed process group.  Collective call (every rank)."""
        if unique_id is None:
            import torch.distributed as dist
            cuda = dist.get_backend() == "nccl"
            t = torch.zeros(128, dtype=torch.uint8)
            if self.cfg.rank == 0:
                t = torch.frombuffer(bytearray(rccl_unique_id()), dtype=torch.uint8).clone()
            if cuda:
                t = t.to(self.device)
            dist.broadcast(t, src=0)
            unique_id = bytes(t.cpu().numpy().tobytes())
        assert len(unique_id) == 128
        self._check(self.lib.cofusion_init_rccl(self.h, C.c_char_p(unique_id)))
        self.rccl = True

    def broadcast(self, tensor, root=0):
        """a device tensor (a frame: depth + colour in one buffer) from rank `root` to every rank: ncclBroadcast on the context's stream"""
        self._check(self.lib.cofusion_broadcast(self.h, C.c_void_p(tensor.data_ptr()), C.c_uint64(tensor.numel() * tensor.element_size()), int(root)))

    def set_allreduce(self, fn=None):
        """model-parallel mode (rank / world given at construction): register the SUM all-reduce of int64 buffers.
        Default: torch.distributed.all_reduce on the default process group (gloo: CPU tensor, nccl: staged through the GPU)."""
        import torch.distributed as dist

        def default(arr):
            t = torch.from_numpy(arr)
            if dist.get_backend() == "nccl":
                g = t.to(self.device)
                dist.all_reduce(g, op=dist.ReduceOp.SUM)
                t.copy_(g.cpu())
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM)

        impl = fn or default

        def thunk(buf, n, _user):
            try:
                impl(np.ctypeslib.as_array(buf, shape=(n,)))
                return 0
            except Exception:  # noqa: BLE001 -- reported through the C return code
                return -1

        self._allreduce_cb = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_int64), C.c_uint64, C.c_void_p)(thunk)  # keep alive
        self._check(self.lib.cofusion_set_allreduce(self.h, self._allreduce_cb, None))
        if fn is None:
            self.set_allreduce_device()   # nccl: RCCL on the stream; gloo (dry runs): the same path, staged by gloo itself
            if self.cfg.shard_background:
                self.set_collective()

    def set_collective(self):
        """C-ABI level collective (cf_set_collective) for a background split over the ranks (shard_background=1): op 0 = SUM of int64 words
        (the normal-equation accumulators, after every launch of the Gauss-Newton loop), op 1 = MIN of unsigned 64-bit words (the z-keys
        of the index map).  torch.distributed in place on a tensor view of the library's buffer, on the stream the library hands over."""
        import torch.distributed as dist
        sign = torch.tensor(-2 ** 63, dtype=torch.int64, device=self.device)

        def thunk(_user, op, dev_buf, words, hip_stream):
            try:
                t = torch.as_tensor(_DevWords(dev_buf, words), device=self.device)   # a view of the library's buffer: reduced in place
                stream = torch.cuda.ExternalStream(int(hip_stream), device=self.device) if hip_stream else torch.cuda.current_stream(self.device)
                with torch.cuda.stream(stream):   # everything on the stream the library enqueues this model's work on
                    if op == 0:
                        dist.all_reduce(t, op=dist.ReduceOp.SUM)
                    else:  # unsigned order through the signed collective: flip the sign bit, MIN, flip back
                        t.bitwise_xor_(sign)
                        dist.all_reduce(t, op=dist.ReduceOp.MIN)
                        t.bitwise_xor_(sign)
                return 0
            except Exception:  # noqa: BLE001 -- reported through the C return code
                import traceback
                traceback.print_exc()
                return -1

        self._collective_cb = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p)(thunk)  # keep alive
        assert self.abi.cf_set_collective(self._ctx(), self._collective_cb, None) == 0

    def set_allreduce_device(self):
        """The collective for buffers that live in HBM (per-superpixel segmentation sums): RCCL all-reduce of an int64 torch tensor,
        enqueued after the library's work on the context's stream -- no host visit.  torch.distributed takes tensors, not raw
        addresses: the library's buffer is wrapped as a tensor VIEW (_DevWords) and reduced in place.  This is the path of process groups
        that are not RCCL (the gloo tests on a one-GPU box); with RCCL use init_rccl (the library's own communicator)."""
        import torch.distributed as dist

        def thunk(dev_buf, n, hip_stream, _user):
            try:
                t = torch.as_tensor(_DevWords(dev_buf, n), device=self.device)   # a view of the library's buffer: reduced in place
                # the library enqueues on the torch stream it was handed (set_stream / the current stream at construction)
                stream = torch.cuda.ExternalStream(int(hip_stream), device=self.device) if hip_stream else torch.cuda.current_stream(self.device)
                with torch.cuda.stream(stream):
                    dist.all_reduce(t, op=dist.ReduceOp.SUM)
                return 0
            except Exception:  # noqa: BLE001 -- reported through the C return code
                import traceback
                traceback.print_exc()
                return -1

        self._allreduce_dev_cb = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p)(thunk)  # keep alive
        self._check(self.lib.cofusion_set_allreduce_device(self.h, self._allreduce_dev_cb, None))

    def model_owned(self, index):
        return self.lib.cofusion_model_owned(self.h, index) == 1

    def set_stream(self, stream):
        """enqueue all work of this instance on a torch.cuda.Stream (or a raw hipStream_t value)"""
        ptr = stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream)
        self._check(self.lib.cofusion_set_stream(self.h, C.c_void_p(ptr)))

    def save_ply(self, prefix):
        """CoFusion::savePly: <prefix>cloud-<id>.ply per model; returns the number of files"""
        n = self.lib.cofusion_save_ply(self.h, str(prefix).encode())
        if n < 0:
            raise CoFusionError(self.lib.cofusion_last_error().decode())
        return n

    def set_export_segmentation(self, prefix):
        """every segmented frame writes <prefix>Segmentation<tick>.png (labels, rejected superpixels as 0); '' switches it off"""
        self._check(self.lib.cofusion_set_export_segmentation(self.h, str(prefix or "").encode()))

    def export_poses(self, prefix):
        """CoFusion::exportPoses: <prefix>poses-<id>.txt per logged model (needs enable_pose_logging=1)"""
        n = self.lib.cofusion_export_poses(self.h, str(prefix).encode())
        if n < 0:
            raise CoFusionError(self.lib.cofusion_last_error().decode())
        return n

    def close(self):
        if getattr(self, "h", None):
            if getattr(self, "_owned", True):
                self.lib.cofusion_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_crf(self, unary_weight_error=75.0, unary_k_error=0.0375, threshold_new=5.5, weight_appearance=7.0, weight_smoothness=2.0,
                sigma_rgb=10.0, sigma_depth=0.9, sigma_pos=1.8, min_rel_size_new=0.015, max_rel_size_new=0.4, iterations=10):
        f = C.c_float
        self._check(self.lib.cofusion_set_crf(self.h, f(unary_weight_error), f(unary_k_error), f(threshold_new), f(weight_appearance),
                                              f(weight_smoothness), f(sigma_rgb), f(sigma_depth), f(sigma_pos), f(min_rel_size_new),
                                              f(max_rel_size_new), C.c_uint(iterations)))

    def process_frame(self, depth, rgb, mask=None, in_pose=None, timestamp=0):
        """Host numpy inputs: depth f32 [H,W] metres, rgb u8 [H,W,3], optional GT mask u8 [H,W]."""
        d = np.ascontiguousarray(depth, np.float32); c = np.ascontiguousarray(rgb, np.uint8)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        p = None if in_pose is None else np.ascontiguousarray(in_pose, np.float32).reshape(16)
        self._check(self.lib.cofusion_process_frame(self.h, C.c_int64(timestamp), c.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p),
                                                    None if m is None else m.ctypes.data_as(C.c_void_p),
                                                    None if p is None else p.ctypes.data_as(C.c_void_p)))

    def process_frame_device(self, depth_t, rgba_t, in_pose=None, timestamp=0):
        """Frame already resident in HBM: torch CUDA tensors depth f32 [H,W], rgba u8 [H,W,4]."""
        p = None if in_pose is None else np.ascontiguousarray(in_pose, np.float32).reshape(16)
        self._check(self.lib.cofusion_process_frame_device(self.h, C.c_int64(timestamp), C.c_void_p(depth_t.data_ptr()),
                                                           C.c_void_p(rgba_t.data_ptr()),
                                                           None if p is None else p.ctypes.data_as(C.c_void_p)))

    @property
    def num_models(self):
        return self.lib.cofusion_num_models(self.h)

    @property
    def tick(self):
        return self.lib.cofusion_tick(self.h)

    @property
    def lost(self):
        """CoFusion::getLost: the camera is lost (option reloc=1)"""
        return bool(self.lib.cofusion_is_lost(self.h))

    def model_info(self, index):
        mid = C.c_uint(); cnt = C.c_uint(); conf = C.c_float(); pose = (C.c_float * 16)()
        self._check(self.lib.cofusion_model_info(self.h, index, C.byref(mid), C.byref(cnt), pose, C.byref(conf)))
        return dict(id=mid.value, count=cnt.value, pose=np.array(pose, np.float32).reshape(4, 4), conf_threshold=conf.value)

    def model_cull_box(self, index):
        b = (C.c_int * 4)()
        self._check(self.lib.cofusion_model_cull_box(self.h, index, b))
        return list(b)

    def model_level0_visited(self, index):
        """(ICP pixels, residual pixels) the level-0 launch of the model's last tracking call visited for it (include/cofusion.h)"""
        a = C.c_uint64(); b = C.c_uint64()
        self._check(self.lib.cofusion_model_level0_visited(self.h, index, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def model_icp_stats(self, index):
        e = C.c_float(); c = C.c_float()
        self._check(self.lib.cofusion_model_icp_stats(self.h, index, C.byref(e), C.byref(c)))
        return e.value, c.value

    def model_tracking_inputs(self, index):
        """host copies of the prediction the next frame's tracking of this model reads: vertex4, normal4 (f32 [H,W,4]), image (u8 [H,W,4])"""
        H, W = self.height, self.width
        v = np.empty((H, W, 4), np.float32); n = np.empty((H, W, 4), np.float32); img = np.empty((H, W, 4), np.uint8)
        self._check(self.lib.cofusion_model_tracking_inputs(self.h, index, v.ctypes.data_as(C.c_void_p), n.ctypes.data_as(C.c_void_p),
                                                            img.ctypes.data_as(C.c_void_p)))
        return v, n, img

    def set_gn_mode(self, mode):
        assert self.abi.cf_set_gn_mode(self._ctx(), int(mode)) == 0

    def model_download(self, index):
        n = self.model_info(index)["count"]
        out = np.zeros((max(n, 1), 12), np.float32)
        c = C.c_uint32()
        self._check(self.lib.cofusion_model_download(self.h, index, out.ctypes.data_as(C.c_void_p), n, C.byref(c)))
        return out[:n]

    def mask(self):
        ptr = self.lib.cofusion_mask_device(self.h)
        host = np.empty(self.width * self.height, np.uint8)
        ctx = C.c_void_p(self.lib.cofusion_context(self.h))
        rc = self.abi.cf_memcpy_d2h(ctx, host.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_uint64(host.size))
        if rc != 0:
            raise CoFusionError("mask readback failed")
        return host.reshape(self.height, self.width)

    # profiling hooks of the underlying C-ABI context
    def _ctx(self):
        return C.c_void_p(self.lib.cofusion_context(self.h))

    def set_icp_launch(self, threads, ppt):
        assert self.abi.cf_set_icp_launch(self._ctx(), threads, ppt) == 0

    def set_icp_arith(self, mode):
        """rounding specification of the ICP sums: 0 / "product" (default), 1 / "gram" or 2 / "reference" (the reference's own f32 trees and host loop; include/cofusion_hip.h: cf_set_icp_arith)"""
        assert self.abi.cf_set_icp_arith(self._ctx(), {"product": 0, "gram": 1, "reference": 2}.get(mode, mode)) == 0

    def profile_enable(self, on=True):
        """on: False / True, or N > 1 = events on the level-0 launches of every N-th tracking call"""
        assert self.abi.cf_profile_enable(self._ctx(), int(on)) == 0

    def profile_read(self, reset=True):
        p = Profile()
        assert self.abi.cf_profile_read(self._ctx(), C.byref(p), int(reset)) == 0
        return p


class CoFusionGroup:
    """Several independent RGB-D sequences on ONE GPU in lock-step (include/cofusion.h: cofusion_group_*): one context, one set of
    tracking launches for the trackers of all sequences.  `sequences[s]` is a CoFusion view of sequence s for the getters (model_info,
    model_download, mask, ...); frames are fed to all sequences at once with process_frames / process_frames_device."""

    def __init__(self, n_sequences, width=640, height=480, fx=528.0, fy=528.0, cx=320.0, cy=240.0, device=0, **kw):
        if not torch.cuda.is_available():
            raise CoFusionError("no GPU visible: the Co-Fusion hot path has no CPU fallback")
        self.lib = _libmod.load_host()
        cfg = _make_config(self.lib, width, height, fx, fy, cx, cy, device, kw)
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.n = int(n_sequences)
        self.g = C.c_void_p()
        if self.lib.cofusion_group_create(C.byref(cfg), self.n, C.byref(self.g)) != 0:
            raise CoFusionError(f"cofusion_group_create: {self.lib.cofusion_last_error().decode()}")
        self._check(self.lib.cofusion_group_set_stream(self.g, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        self.sequences = [CoFusion(width, height, fx, fy, cx, cy, device, _borrowed=self.lib.cofusion_group_sequence(self.g, s), **kw)
                          for s in range(self.n)]

    def _check(self, rc):
        if rc != 0:
            raise CoFusionError(f"cofusion error {rc}: {self.lib.cofusion_last_error().decode()}")

    def set_stream(self, stream):
        ptr = stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream)
        self._check(self.lib.cofusion_group_set_stream(self.g, C.c_void_p(ptr)))

    def process_frames(self, depths, rgbs, masks=None, timestamp=0):
        """one frame per sequence from host arrays: depths[s] f32 [H,W] metres, rgbs[s] u8 [H,W,3], masks[s] u8 [H,W] or None"""
        keep = [np.ascontiguousarray(d, np.float32) for d in depths] + [np.ascontiguousarray(r, np.uint8) for r in rgbs]
        mk = [None if (masks is None or m is None) else np.ascontiguousarray(m, np.uint8) for m in (masks or [None] * self.n)]
        P = C.c_void_p * self.n
        ts = (C.c_int64 * self.n)(*([timestamp] * self.n))
        self._check(self.lib.cofusion_group_process_frames(
            self.g, ts, P(*[r.ctypes.data for r in keep[self.n:]]), P(*[d.ctypes.data for d in keep[:self.n]]),
            P(*[None if m is None else m.ctypes.data for m in mk]) if masks is not None else None))

    def process_frames_device(self, depth_ts, rgba_ts, timestamp=0):
        """one frame per sequence already resident in HBM: torch CUDA tensors depth f32 [H,W], rgba u8 [H,W,4]"""
        P = C.c_void_p * self.n
        ts = (C.c_int64 * self.n)(*([timestamp] * self.n))
        self._check(self.lib.cofusion_group_process_frames_device(self.g, ts, P(*[t.data_ptr() for t in depth_ts]),
                                                                  P(*[t.data_ptr() for t in rgba_ts])))

    def close(self):
        if getattr(self, "g", None):
            for s in self.sequences:
                s.h = None
            self.lib.cofusion_group_destroy(self.g)
            self.g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
