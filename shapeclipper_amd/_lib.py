"""ctypes binding of libshapeclipper_hip.so (the C ABI declared in include/shapeclipper_hip.h).

The product path has no CPU fallback: if the shared library is missing or a tensor is not on a
ROCm device, calls raise.  Tensors are passed as raw device pointers (``tensor.data_ptr()``) plus
explicit sizes; kernels are enqueued on torch's *current* stream.
"""
from __future__ import annotations

import ctypes
import os
import threading
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# SHAPECLIPPER_HIP_LIB: load another build of the same C ABI (tuning / A-B experiments, e.g. tools/micro variants)
LIB_PATH = os.environ.get("SHAPECLIPPER_HIP_LIB") or os.path.join(_HERE, "lib", "libshapeclipper_hip.so")

SYMBOLS = (
    "sc_chamfer3d_forward", "sc_chamfer3d_forward_split", "sc_chamfer3d_forward_grid", "sc_chamfer3d_backward", "sc_sdf_forward", "sc_rgb_composite_forward",
    "sc_rgb_composite_backward", "sc_sdf_backward", "sc_wgrad", "sc_partial_reduce", "sc_tbl_sum", "sc_loss_fused_forward",
    "sc_clip_vit_forward", "sc_gemm_bf16", "sc_f32_to_bf16", "sc_clip_vit_forward_f16", "sc_gemm_f16", "sc_f32_to_f16",
    "sc_clip_cluster_supported", "sc_clip_cluster_pack", "sc_clip_vit_forward_packed", "sc_clip_cluster_set_batch_range",
    "sc_ray_sample_forward", "sc_ray_sample_backward", "sc_ray_sample_forward_eik", "sc_ray_sample_backward_eik", "sc_render_forward", "sc_sdf_grid_forward", "sc_sdf_grid_forward_split", "sc_sdf_value_forward_split", "sc_sdf_stream_pack_bytes", "sc_sdf_stream_pack", "sc_sdf_forward_stream", "sc_loss_fused_backward",
    "sc_bn_splits", "sc_bn_act_forward", "sc_bn_act_backward", "sc_bn_relu_pool_forward", "sc_bn_relu_pool_backward",
    "sc_isosurface_count", "sc_isosurface_emit", "sc_marching_cubes_count", "sc_marching_cubes_emit",
    "sc_isosurface_blocks_per_image", "sc_isosurface_block_scan", "sc_isosurface_block_count", "sc_isosurface_block_emit", "sc_marching_cubes_block_count",
    "sc_marching_cubes_block_emit", "sc_marching_cubes_block_count_masks", "sc_marching_cubes_block_emit_masks",
    "sc_camera_rays_forward", "sc_camera_rays_backward", "sc_pose_from_trig_forward", "sc_pose_from_trig_backward",
    "sc_estimator_head_forward", "sc_estimator_head_backward", "sc_camera_prior_forward", "sc_camera_prior_backward",
    "sc_camera_prior_max_images", "sc_transform_normal_forward", "sc_transform_normal_backward", "sc_loss_total_forward",
    "sc_loss_total_backward",
    "sc_render_backward", "sc_sdf_backward_fused", "sc_sdf_backward_fused_parts", "sc_sdf_backward_fused_partial_floats", "sc_tbl_sum_blocks", "sc_conv3x3_pack", "sc_conv3x3_forward", "sc_conv3x3_pack_multi", "sc_conv3x3_pack_multi_units", "sc_conv3x3_tile_channels", "sc_conv3x3_wgrad", "sc_conv3x3_wgrad_split", "sc_conv3x3_forward_split", "sc_conv3x3_forward_add", "sc_conv3x3_tile_channels_split", "sc_conv_stem_forward", "sc_conv_stem_wgrad", "sc_conv1x1s2_forward", "sc_conv1x1s2_backward_data", "sc_conv1x1s2_wgrad", "sc_conv3x3s2_forward", "sc_conv3x3s2_bd_pack", "sc_conv3x3s2_backward_data", "sc_conv3x3s2_wgrad",
    "sc_basic_block_forward", "sc_basic_block_backward", "sc_rgb_composite_backward_v3", "sc_set_reserved_cus", "sc_grid_cus", "sc_conv3x3_release_tables",
    "sc_linear_bn_supported", "sc_linear_bn_forward", "sc_linear_bn_backward", "sc_linear_backward_data",
    "sc_latent_bias_forward", "sc_latent_bias_backward", "sc_rgb_composite_backward_fused", "sc_rgb_composite_backward_fused_stash", "sc_rgb_composite_backward_fused_split", "sc_rgb_composite_forward_stash", "sc_rgb_composite_forward_split", "sc_rgb_composite_backward_fused_parts",
    "sc_rgb_composite_backward_fused_partial_floats",
)
# entry points that do not return an int status
SYMBOLS_OTHER = ("sc_clip_cluster_pack_elems", "sc_render_backward_workspace_bytes", "sc_chamfer3d_grid_workspace_bytes", "sc_clip_vit_workspace_bytes", "sc_conv3x3_pack_floats", "sc_conv3x3_workspace_floats", "sc_conv3x3_wgrad_workspace_floats", "sc_conv3x3_pack_floats_split", "sc_conv3x3_workspace_floats_split", "sc_conv_stem_wgrad_workspace_floats", "sc_conv1x1s2_wgrad_workspace_floats", "sc_conv3x3s2_pack_floats", "sc_conv3x3s2_workspace_floats", "sc_conv3x3s2_bd_pack_floats", "sc_conv3x3s2_bd_workspace_floats")

_lib: Optional[ctypes.CDLL] = None

# Optional per-entry-point GPU timing (bench.py): when TIMING is a dict every C-ABI call is bracketed by
# two events on torch's current stream (the stream the kernels are enqueued on).
TIMING = None
TIMING_SKIP = ("sc_sdf_stream_pack", "sc_sdf_stream_pack_bytes", "sc_rgb_composite_backward_fused_parts", "sc_rgb_composite_backward_fused_partial_floats", "sc_latent_bias_forward", "sc_latent_bias_backward", "sc_set_reserved_cus", "sc_grid_cus", "sc_conv3x3_release_tables", "sc_linear_bn_supported", "sc_linear_bn_forward", "sc_linear_bn_backward", "sc_linear_backward_data", "sc_bn_splits", "sc_isosurface_blocks_per_image", "sc_bn_act_forward", "sc_bn_act_backward", "sc_bn_relu_pool_forward", "sc_bn_relu_pool_backward",
               # the trunks' convolutions: ~250 calls per step -- two events each cost the step ~1 ms (rocprofv3 covers them: profiles/)
               "sc_conv3x3_forward", "sc_conv3x3_forward_split", "sc_conv3x3_forward_add", "sc_conv3x3_wgrad", "sc_conv3x3_wgrad_split", "sc_conv3x3_pack", "sc_conv3x3_pack_multi", "sc_conv3x3_pack_multi_units",
               "sc_conv_stem_forward", "sc_conv_stem_wgrad", "sc_conv1x1s2_forward", "sc_conv1x1s2_backward_data", "sc_conv1x1s2_wgrad",
               "sc_conv3x3s2_forward", "sc_conv3x3s2_bd_pack", "sc_conv3x3s2_backward_data", "sc_conv3x3s2_wgrad", "sc_conv3x3_tile_channels", "sc_conv3x3_tile_channels_split",
               "sc_basic_block_forward", "sc_basic_block_backward")

class _Timed:
    """Wraps a ctypes function: records (start, end) events around the call when TIMING is enabled."""

    def __init__(self, name, fn):
        self.name, self.fn = name, fn

    def __call__(self, *args):
        if TIMING is None or self.name in TIMING_SKIP:
            return self.fn(*args)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        code = self.fn(*args)
        e.record()
        TIMING.setdefault(self.name, []).append((s, e, args))
        return code


class _LibProxy:
    def __init__(self, cdll):
        self._cdll = cdll
        for name in SYMBOLS:
            fn = getattr(cdll, name)
            fn.restype = ctypes.c_int
            setattr(self, name, _Timed(name, fn))
        for name in SYMBOLS_OTHER:          # size queries: long long result, host-only
            fn = getattr(cdll, name)
            fn.restype = ctypes.c_longlong
            setattr(self, name, fn)


class HipLibraryMissing(RuntimeError):
    pass


def load():
    """Load the library once.  Raises HipLibraryMissing (never falls back)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f"{LIB_PATH} not found -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C shapeclipper_amd/csrc`; there is no CPU fallback for the HIP hot path")
        _lib = _LibProxy(ctypes.CDLL(LIB_PATH))
    return _lib


_tls = threading.local()    # .dev: device index of the tensor ptr() saw last ON THIS THREAD (autograd and side-stream threads have their own)


def ptr(t: Optional[torch.Tensor]):
    """Device pointer of a contiguous CUDA/ROCm tensor (NULL for None)."""
    if t is None:
        return ctypes.c_void_p(0)
    if not t.is_cuda:
        raise RuntimeError("shapeclipper_amd: HIP kernels need device tensors (got a CPU tensor); "
                           "the product path has no CPU fallback")
    if not t.is_contiguous():
        raise RuntimeError("shapeclipper_amd: tensor must be contiguous")
    _tls.dev = t.get_device()
    return ctypes.c_void_p(t.data_ptr())


def raw_stream(device: Optional[int] = None) -> int:
    """hipStream_t of torch's current stream on the current device as an integer.  (torch.cuda.current_stream() builds a Python Stream
    object per call, ~8 us: at ~280 entry-point calls per training step that was 2.4 ms of host time per step.)

    Device guard: the kernels are launched on the CURRENT device's stream, so a tensor that lives on another device
    (rank != device index, a stray `cuda:0` default in a multi-GPU process) would be dereferenced by the wrong GPU.  `device` = the
    device index of the call's tensors: callers that take raw data_ptr()s (BatchNorm, block and workspace paths) pass it explicitly;
    entry points that pass their tensors through ptr() leave it None and the device of the last ptr() of THIS thread is checked.  The
    remembered device is consumed by the check, so a later, unrelated call is never judged by a stale tensor."""
    cur = torch._C._cuda_getDevice()
    want = getattr(_tls, "dev", -1) if device is None else device
    _tls.dev = -1
    if want is not None and want >= 0 and want != cur:
        raise RuntimeError("shapeclipper_amd: tensor on cuda:%d but the current device is cuda:%d -- call torch.cuda.set_device(rank's device) "
                           "(or wrap the call in `with torch.cuda.device(t.device)`); kernels launch on the current device's stream"
                           % (want, cur))
    return torch._C._cuda_getCurrentRawStream(cur)


def stream():
    return ctypes.c_void_p(raw_stream())


def check(code: int, what: str):
    if code != 0:
        raise RuntimeError(f"shapeclipper_amd: {what} failed with hipError_t {code}")
