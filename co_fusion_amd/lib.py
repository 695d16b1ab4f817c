"""ctypes loader for the C-ABI library (co_fusion_amd/lib/libcofusion_hip.so).

The product path fails loudly when the HIP extension is missing or no GPU is present:
there is no CPU fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# CF_HIP_LIB: another build of the C-ABI library (A/B micro-benchmarks against an earlier build; diagnostics only)
# CF_LIB_DIR: a whole diagnostics build (make ABLATE=1 LIBDIR=../lib_ablate in csrc/ and host/: both libraries side by side)
_LIB_DIR = os.environ.get("CF_LIB_DIR") or os.path.join(_HERE, "lib")
LIB_PATH = os.environ.get("CF_HIP_LIB") or os.path.join(_LIB_DIR, "libcofusion_hip.so")

# every symbol include/cofusion_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "cf_create", "cf_destroy", "cf_last_error", "cf_set_stream", "cf_use_own_stream", "cf_get_stream", "cf_synchronize", "cf_fork", "cf_main", "cf_join", "cf_join_lane", "cf_thread_lane", "cf_mark", "cf_event_wait_host", "cf_fork_after", "cf_malloc",
    "cf_free", "cf_memcpy_h2d", "cf_memcpy_d2h", "cf_malloc_host", "cf_free_host", "cf_memcpy_h2d_async", "cf_memcpy_d2h_async", "cf_memcpy_d2d_async", "cf_rgb_to_rgba", "cf_create_vmap", "cf_create_nmap", "cf_copy_maps", "cf_resize_map",
    "cf_transform_maps", "cf_vertices_to_depth", "cf_pyrdown_gauss_f32", "cf_pyrdown_gauss_u8",
    "cf_rgba_to_intensity", "cf_sobel", "cf_project_cloud", "cf_icp_step", "cf_icp_step_band", "cf_rgb_residual", "cf_rgb_step",
    "cf_so3_step", "cf_odom_create", "cf_odom_destroy", "cf_odom_init_icp_model", "cf_odom_init_rgb_model",
    "cf_odom_init_rgb", "cf_odom_init_models_batch", "cf_odom_init_models_batch_frames", "cf_odom_init_models_batch_select", "cf_model_fill_ratio_device", "cf_odom_init_first_rgb", "cf_odom_init_icp", "cf_odom_get_incremental_transformation",
    "cf_odom_track_batch_async", "cf_odom_fetch_result", "cf_odom_get_covariance", "cf_odom_bind_frame_maps", "cf_odom_share_frame_maps", "cf_odom_set_culling", "cf_odom_set_band", "cf_set_collective", "cf_model_predict_indices_sharded", "cf_odom_buffer",
    "cf_bilateral", "cf_model_create", "cf_model_destroy", "cf_model_initialise", "cf_model_count",
    "cf_model_predict_indices", "cf_model_index_keys", "cf_model_index_resolve", "cf_model_combined_predict", "cf_model_prefetch_fill_ratio", "cf_model_perform_fill_in", "cf_model_requires_fill_in",
    "cf_model_fuse", "cf_model_clean", "cf_models_frame_passes", "cf_models_preindex", "cf_model_download_map", "cf_model_upload_map", "cf_model_buffer",
    "cf_fusion_weight", "cf_seg_create", "cf_seg_destroy", "cf_seg_slic", "cf_seg_accumulate", "cf_seg_crf", "cf_seg_upsample", "cf_seg_sums", "cf_seg_infer", "cf_seg_run_batch", "cf_seg_fetch", "cf_seg_publish_poses", "cf_seg_fetch_poses",
    "cf_seg_labels",
    "cf_depth_pyramid", "cf_set_icp_launch", "cf_set_icp_arith", "cf_get_icp_arith", "cf_set_gn_mode", "cf_profile_enable", "cf_profile_read", "cf_odom_bench_icp", "cf_odom_level0_visited",
    "cf_rccl_unique_id", "cf_rccl_init", "cf_rccl_allreduce", "cf_rccl_broadcast", "cf_rccl_info", "cf_rccl_destroy",
]


def build(verbose: bool = False) -> str:
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    # host-side C++ facade (g++), links against the C-ABI library
    cmd = ["make", "-C", os.path.join(_HERE, "host")]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return LIB_PATH


HOST_LIB_PATH = os.path.join(_LIB_DIR, "libcofusion.so")
HOST_SYMBOLS = [
    "cofusion_default_config", "cofusion_create", "cofusion_destroy", "cofusion_last_error", "cofusion_set_stream",
    "cofusion_process_frame", "cofusion_process_frame_device", "cofusion_num_models", "cofusion_tick", "cofusion_model_info",
    "cofusion_model_download", "cofusion_model_icp_stats", "cofusion_model_cull_box", "cofusion_model_level0_visited", "cofusion_model_tracking_inputs", "cofusion_mask_device", "cofusion_context", "cofusion_set_crf",
    "cofusion_save_ply", "cofusion_export_poses", "cofusion_set_export_segmentation", "cofusion_klg_open", "cofusion_klg_next", "cofusion_klg_set_reference_compatible", "cofusion_klg_close",
    "cofusion_klg_create", "cofusion_klg_write", "cofusion_klg_finish", "cofusion_debug_phase_ms", "cofusion_set_allreduce", "cofusion_set_allreduce_device", "cofusion_group_create", "cofusion_group_destroy", "cofusion_group_size", "cofusion_group_sequence", "cofusion_group_set_stream", "cofusion_group_process_frames", "cofusion_group_process_frames_device", "cofusion_rccl_unique_id", "cofusion_init_rccl", "cofusion_broadcast", "cofusion_model_owned", "cofusion_is_lost",
]
_host = None


def load_host() -> C.CDLL:
    """libcofusion.so: the C++ CoFusion/Model/Segmentation facade behind a flat C wrapper (include/cofusion.h)."""
    global _host
    if _host is None:
        load()  # torch first, then the C-ABI library (same HIP runtime instance)
        if not os.path.exists(HOST_LIB_PATH):
            raise RuntimeError(f"{HOST_LIB_PATH} is missing: run __graft_entry__.build()")
        _host = C.CDLL(HOST_LIB_PATH)
        _host.cofusion_last_error.restype = C.c_char_p
        _host.cofusion_mask_device.restype = C.c_void_p
        _host.cofusion_context.restype = C.c_void_p
        _host.cofusion_group_sequence.restype = C.c_void_p
    return _host


_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback for the product path)")
        # PyTorch bundles its own libamdhip64/libhsa-runtime64.  Device pointers and streams are only
        # interchangeable when both sides run on ONE HIP runtime instance, so torch must be loaded
        # first; the library's libamdhip64.so.7 dependency then resolves to the already-loaded copy.
        import torch  # noqa: F401
        _lib = C.CDLL(LIB_PATH)
        _lib.cf_last_error.restype = C.c_char_p
        _lib.cf_get_stream.restype = C.c_void_p
    return _lib
