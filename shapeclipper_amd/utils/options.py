"""YAML + `--a.b.c=value` command-line configuration, same semantics as the reference's
utils/options.py (parse_arguments :16-34, set :36-44, _parent_ inheritance :46-60, process_options
:79-95) with two deliberate differences for unattended runs: an unknown CLI key and a changed
options.yaml never block on input() unless stdin is a TTY (benchmarks / CI must not hang)."""
from __future__ import annotations

import os
import random
import string
import sys

import numpy as np
import torch
import yaml

from . import util
from .util import EasyDict as edict
from .util import log

torch.backends.cudnn.benchmark = False

# defaults for keys this build adds (a reference YAML without them still loads)
# deterministic_conv: the reference sets cudnn.deterministic=True globally (utils/options.py:14); on ROCm that
# restricts MIOpen to GEMM-based backward solvers (measured 437 ms of 640 ms per bs32 step), so it is opt-in here.
HIP_DEFAULTS = dict(hip=dict(device_rng=False, device_choice=True, fused_backward=True, deterministic_conv=False, fused_loss=True, fused_adam=True, guarded_step=True, batched_encoders=True, two_streams=True, overlap_allreduce=False, reserve_cus=0, fused_block=True, fused_bottleneck=True, fused_rgb_wgrad=True, rgb_stash=True, value_split=True, rgb_split=True, rgb_bwd_split=True, sdf_stream=True, upload_stream=True, rocblas=True, conv3x3=True, conv3x3_split=True, conv_stem=True, conv1x1=True, conv3x3s2=True, conv3x3s2_grads=True))


def parse_arguments(args):
    """--key1.key2=value ; --flag (true) ; --flag! (false)"""
    opt_cmd = {}
    for arg in args:
        assert arg.startswith("--")
        if "=" not in arg[2:]:
            key_str, value = (arg[2:-1], "false") if arg[-1] == "!" else (arg[2:], "true")
        else:
            key_str, value = arg[2:].split("=", 1)
        keys = key_str.split(".")
        sub = opt_cmd
        for k in keys[:-1]:
            sub = sub.setdefault(k, {})
        assert keys[-1] not in sub, keys[-1]
        sub[keys[-1]] = yaml.safe_load(value)
    return edict(opt_cmd)


def load_options(fname):
    with open(fname) as f:
        opt = edict(yaml.safe_load(f))
    if "_parent_" in opt:
        parents = opt.pop("_parent_")
        parents = [parents] if isinstance(parents, str) else parents
        for parent in parents:
            opt = override_options(load_options(parent), opt, key_stack=[])
    print("loading {}...".format(fname))
    return opt


def _ask(question):
    if not sys.stdin or not sys.stdin.isatty():
        return "y"
    ans = None
    while ans not in ("y", "n"):
        ans = input(question)
    return ans


def override_options(opt, opt_over, key_stack=None, safe_check=False):
    for key, value in opt_over.items():
        if isinstance(value, dict):
            opt[key] = override_options(opt.get(key, edict()), value, key_stack=key_stack + [key], safe_check=safe_check)
        else:
            if safe_check and key not in opt:
                if _ask("\"{}\" not found in original opt, add? (y/n) ".format(".".join(key_stack + [key]))) == "n":
                    print("safe exiting...")
                    sys.exit()
            opt[key] = value
    return opt


def set(opt_cmd={}, verbose=True):
    if verbose:
        log.info("setting configurations...")
    opt = load_options(opt_cmd.yaml)
    opt = override_options(opt, opt_cmd, key_stack=[], safe_check=True)
    for k, v in HIP_DEFAULTS.items():
        cur = opt.get(k, edict())
        for kk, vv in v.items():
            cur.setdefault(kk, vv)
        opt[k] = edict(cur)
    process_options(opt)
    if verbose:
        log.options(opt)
    return opt


def process_options(opt):
    if opt.seed is not None:
        random.seed(opt.seed)
        np.random.seed(opt.seed)
        torch.manual_seed(opt.seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(opt.seed)
    else:
        opt.name += "_{}".format("".join(random.choice(string.ascii_uppercase) for _ in range(4)))
    opt.output_path = "{0}/{1}/{2}".format(opt.output_root, opt.group, opt.name)
    os.makedirs(opt.output_path, exist_ok=True)
    assert isinstance(opt.gpu, int)
    opt.device = "cpu" if opt.cpu or not torch.cuda.is_available() else "cuda:{}".format(opt.gpu)
    opt.H, opt.W = opt.image_size
    if "data" in opt and "dataset" in opt.data and opt.data.dataset not in opt.data and "pix3d" in opt.data:
        opt.data[opt.data.dataset] = opt.data.pix3d      # e.g. --data.dataset=synthetic reuses the Pix3D camera ranges
    torch.backends.cudnn.deterministic = bool(opt.get("hip", {}).get("deterministic_conv", False))
    from ..model import resnet
    resnet.HIP_CONV3X3 = bool(opt.get("hip", {}).get("conv3x3", True))
    resnet.HIP_CONV3X3_SPLIT = bool(opt.get("hip", {}).get("conv3x3_split", True))
    resnet.HIP_CONV_STEM = bool(opt.get("hip", {}).get("conv_stem", True))
    resnet.HIP_CONV_1X1 = bool(opt.get("hip", {}).get("conv1x1", True))
    resnet.HIP_CONV3X3_S2 = bool(opt.get("hip", {}).get("conv3x3s2", True))
    resnet.HIP_CONV3X3_S2_GRADS = bool(opt.get("hip", {}).get("conv3x3s2_grads", True))
    resnet.FUSED_BLOCK = bool(opt.get("hip", {}).get("fused_block", True))
    from .. import ops as _ops
    _ops.FUSED_RGB_WGRAD = bool(opt.get("hip", {}).get("fused_rgb_wgrad", True))
    _ops.RGB_STASH = bool(opt.get("hip", {}).get("rgb_stash", True))
    _ops.SDF_FWD_STREAM = bool(opt.get("hip", {}).get("sdf_stream", True))       # SDF forward (value, feature, d sdf/dx) from streamed pre-split fragments
    _ops.RGB_BWD_SPLIT = bool(opt.get("hip", {}).get("rgb_bwd_split", True)) and _ops.RGB_STASH and _ops.FUSED_RGB_WGRAD
    _ops.RGB_FWD_SPLIT = bool(opt.get("hip", {}).get("rgb_split", True))         # forward RGB network from pre-split bf16x3 fragments
    _ops.SDF_VALUE_SPLIT = bool(opt.get("hip", {}).get("value_split", True))     # evaluation grid: the pre-split bf16x3 value chain
    from ..model import renderer as _renderer
    _renderer.UPLOAD_STREAM = bool(opt.get("hip", {}).get("upload_stream", True))
    from ..model import view_estimator
    view_estimator.HIP_BOTTLENECK = bool(opt.get("hip", {}).get("fused_bottleneck", True))
    # (Round 5: the bottleneck blocks and the per-image latent biases no longer go through a BLAS at all -- csrc/bottleneck.hip,
    # latent_bias.hip -- so what this switch still touches are the ~20 remaining small products of a step: final projector / head Linears.)
    # The ~83 small fp32 GEMMs of a step (estimator heads, latent projectors: [B..3B, 256..512] x [C, C]) through rocBLAS instead of torch's
    # default hipBLASLt: 7 instead of 18 us of host time per call (a host-paced B=8 step 16.3 -> 14.9 ms); process-global
    # like cudnn.deterministic above, `--hip.rocblas!` leaves torch's choice alone.
    if bool(opt.get("hip", {}).get("rocblas", True)) and torch.cuda.is_available():
        try:
            torch.backends.cuda.preferred_blas_library("cublas")
        except Exception:       # noqa: BLE001  (older torch: no such switch)
            pass


def save_options_file(opt):
    fname = "{}/options.yaml".format(opt.output_path)
    if os.path.isfile(fname):
        with open(fname) as f:
            old = yaml.safe_load(f)
        if util.to_dict(opt) != old:
            print("existing options file found (different from current one)...")
            if _ask("override? (y/n) ") == "n":
                print("safe exiting...")
                sys.exit()
        else:
            print("existing options file found (identical)")
    else:
        print("(creating new options file...)")
    with open(fname, "w") as f:
        yaml.safe_dump(util.to_dict(opt), f, default_flow_style=False, indent=4)
