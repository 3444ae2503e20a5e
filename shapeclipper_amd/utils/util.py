"""Small host-side utilities with the reference's names (utils/util.py): EasyDict, logging,
checkpoint save/restore, distributed setup.  Third-party niceties (termcolor, vigra) are optional."""
from __future__ import annotations

import contextlib
import os
import shutil
import socket
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

try:  # colours are cosmetic; never a hard dependency of the hot path
    import termcolor

    def _paint(msg, color, **kw):
        return termcolor.colored(str(msg), color=color, attrs=[k for k, v in kw.items() if v is True])
except Exception:  # pragma: no cover
    def _paint(msg, color, **kw):
        return str(msg)


def red(m, **k): return _paint(m, "red", **k)
def green(m, **k): return _paint(m, "green", **k)
def blue(m, **k): return _paint(m, "blue", **k)
def cyan(m, **k): return _paint(m, "cyan", **k)
def yellow(m, **k): return _paint(m, "yellow", **k)
def magenta(m, **k): return _paint(m, "magenta", **k)
def grey(m, **k): return _paint(m, "grey", **k)


class EasyDict(dict):
    """dict with attribute access; nested dicts (also inside lists) are converted on assignment."""

    def __init__(self, d=None, **kwargs):
        super().__init__()
        src = dict(d) if d is not None else {}
        src.update(kwargs)
        for k, v in src.items():
            self[k] = v

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, cls):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return [cls(x) if isinstance(x, dict) and not isinstance(x, cls) else x for x in v]
        return v

    def __setitem__(self, k, v):
        v = self._wrap(v)
        super().__setitem__(k, v)
        object.__setattr__(self, k, v)

    __setattr__ = __setitem__

    def __delattr__(self, k):
        super().__delitem__(k)
        object.__delattr__(self, k)

    def update(self, e=None, **f):
        src = dict(e or {})
        src.update(f)
        for k, v in src.items():
            self[k] = v

    def pop(self, k, d=None):
        if k in self:
            v = self[k]
            delattr(self, k)
            return v
        return d

    def __deepcopy__(self, memo):
        import copy
        return type(self)({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def __reduce__(self):
        return (type(self), (dict(self),))


def get_time(sec):
    return int(sec // 86400), int(sec // 3600 % 24), int((sec // 60) % 60), int(sec % 60)


class Log:
    def process(self, pid): print(grey("Process ID: {}".format(pid), bold=True))
    def title(self, message): print(yellow(message, bold=True, underline=True))
    def info(self, message): print(magenta(message, bold=True))
    def warn(self, message): print(red(message, bold=True), file=sys.stderr, flush=True)

    def options(self, opt, level=0):
        for key, value in sorted(opt.items()):
            if isinstance(value, dict):
                print("   " * level + cyan("* ") + green(key) + ":")
                self.options(value, level + 1)
            else:
                print("   " * level + cyan("* ") + green(key) + ":", yellow(value))

    def loss_train(self, opt, ep, lr, loss, timer):
        msg = grey("[train] ", bold=True) + "epoch {}/{}".format(cyan(ep, bold=True), opt.max_epoch)
        msg += ", lr:{}".format(yellow("{:.2e}".format(lr), bold=True))
        msg += ", loss:{}".format(red("{:.3e}".format(float(loss.all)), bold=True))
        msg += ", time:{}".format(blue("{0}-{1:02d}:{2:02d}:{3:02d}".format(*get_time(timer.elapsed)), bold=True))
        msg += " (ETA:{})".format(blue("{0}-{1:02d}:{2:02d}:{3:02d}".format(*get_time(timer.arrival))))
        print(msg)

    def loss_eval(self, opt, loss=None, chamfer=None):
        msg = grey("[eval] ", bold=True)
        if loss is not None:
            msg += "loss:{}".format(red("{:.3e}".format(float(loss.all)), bold=True))
        if chamfer is not None:
            a, c = float(chamfer[0]), float(chamfer[1])
            msg += " chamfer:{}|{}|{}".format(green("{:.4f}".format(a), bold=True), green("{:.4f}".format(c), bold=True),
                                              green("{:.4f}".format((a + c) / 2), bold=True))
        print(msg)


log = Log()


def update_timer(opt, timer, ep, it_per_ep):
    momentum = 0.99
    timer.elapsed = time.time() - timer.start
    timer.it = timer.it_end - timer.it_start
    timer.it_mean = timer.it if timer.it_mean is None else timer.it_mean * momentum + timer.it * (1 - momentum)
    timer.arrival = timer.it_mean * it_per_ep * (opt.max_epoch - ep)


def _map_tensors(X, fn):
    if isinstance(X, dict):
        for k, v in X.items():
            X[k] = _map_tensors(v, fn)
    elif isinstance(X, list):
        for i, e in enumerate(X):
            X[i] = _map_tensors(e, fn)
    elif isinstance(X, tuple) and hasattr(X, "_fields"):
        return type(X)(**_map_tensors(X._asdict(), fn))
    elif isinstance(X, torch.Tensor):
        return fn(X)
    return X


def move_to_device(X, device):
    return _map_tensors(X, lambda t: t.to(device=device))


def detach_tensors(X):
    return _map_tensors(X, lambda t: t.detach())


def to_dict(D, dict_type=dict):
    D = dict_type(D)
    for k, v in D.items():
        if isinstance(v, dict):
            D[k] = to_dict(v, dict_type)
    return D


def get_child_state_dict(state_dict, key):
    out = {}
    for k, v in state_dict.items():
        name = k[7:] if k.startswith("module.") else k
        if name.startswith(key + "."):
            out[name[len(key) + 1:]] = v
    return out


def restore_checkpoint(opt, model, load_name=None, resume=False, best=False, evaluate=False):
    """Same file layout / key names as the reference (utils/util.py:123-169)."""
    assert (load_name is None) == (resume is not False)
    graph = model.graph.module if hasattr(model.graph, "module") else model.graph
    map_loc = torch.device(opt.device) if not isinstance(opt.device, int) else torch.device("cuda", opt.device)
    if resume:
        if best:
            load_name = "{0}/best.ckpt".format(opt.output_path)
        elif resume is True:
            load_name = "{0}/latest.ckpt".format(opt.output_path)
        else:
            load_name = "{0}/checkpoint/ep{1}.ckpt".format(opt.output_path, opt.resume)
        checkpoint = torch.load(load_name, map_location=map_loc)
        if evaluate:
            sd = {k: v for k, v in checkpoint["graph"].items() if "discriminator" not in k}
            missing, unexpected = graph.load_state_dict(sd, strict=False)
            print("Missing keys:", missing)
            print("Unexpected keys:", unexpected)
        else:
            graph.load_state_dict(checkpoint["graph"], strict=True)
    else:
        checkpoint = torch.load(load_name, map_location=map_loc)
        for name, child in graph.named_children():
            child_sd = get_child_state_dict(checkpoint["graph"], name)
            if child_sd:
                print("restoring {} on device {}...".format(name, opt.device))
                child.load_state_dict(child_sd)
            else:
                print("skipping {} on device {}...".format(name, opt.device))
    for key in model.__dict__:
        if key.split("_")[0] in ["optim", "sched"] and key in checkpoint and resume:
            print("restoring {} on device {}...".format(key, opt.device))
            getattr(model, key).load_state_dict(checkpoint[key])
    if resume:
        if resume is not True:
            assert resume == checkpoint["epoch"]
        ep, it, best_val = checkpoint["epoch"], checkpoint["iter"], checkpoint["best_val"]
        print("resuming from epoch {0} (iteration {1})".format(ep, it))
        return ep, it, best_val
    return None, None, None


def save_checkpoint(opt, model, ep, it, best_val, latest=False, best=False, children=None):
    os.makedirs("{0}/checkpoint".format(opt.output_path), exist_ok=True)
    graph = model.graph.module if hasattr(model.graph, "module") else model.graph
    sd = graph.state_dict()
    if children is not None:
        sd = {k: v for k, v in sd.items() if k.startswith(children)}
    checkpoint = dict(epoch=ep, iter=it, best_val=best_val, graph=sd)
    for key in model.__dict__:
        if key.split("_")[0] in ["optim", "sched"]:
            checkpoint[key] = getattr(model, key).state_dict()
    torch.save(checkpoint, "{0}/latest.ckpt".format(opt.output_path))
    if best:
        shutil.copy("{0}/latest.ckpt".format(opt.output_path), "{0}/best.ckpt".format(opt.output_path))
    if not latest:
        shutil.copy("{0}/latest.ckpt".format(opt.output_path), "{0}/checkpoint/ep{1}.ckpt".format(opt.output_path, ep))


@contextlib.contextmanager
def suppress(stdout=False, stderr=False):
    with open(os.devnull, "w") as devnull:
        old_out, old_err = sys.stdout, sys.stderr
        if stdout: sys.stdout = devnull
        if stderr: sys.stderr = devnull
        try:
            yield
        finally:
            sys.stdout, sys.stderr = old_out, old_err


def toggle_grad(model, requires_grad):
    for p in model.parameters():
        p.requires_grad_(requires_grad)


def compute_sampling_prob(opt, mask, uniform_fac=3):
    """Importance ray sampling around the silhouette (reference utils/util.py:237-248).
    Uses vigra's boundary distance transform when available, scipy's EDT otherwise."""
    assert len(mask.shape) == 2
    h, w = mask.shape
    assert opt.H == h
    binary = (mask > 0.5).float().cpu().numpy()
    try:
        import vigra
        sdf_2D = vigra.filters.boundaryDistanceTransform(binary)
    except Exception:
        from scipy import ndimage
        inside = ndimage.distance_transform_edt(binary > 0.5)
        outside = ndimage.distance_transform_edt(binary <= 0.5)
        sdf_2D = np.maximum(inside + outside - 0.5, 0).astype(np.float32)
    prob = 1 / (torch.from_numpy(np.asarray(sdf_2D)).float() + uniform_fac)
    prob = torch.nn.functional.normalize(prob.view(h * w), dim=-1, p=1).cpu().numpy().astype(np.float64)
    prob = prob / prob.sum()
    return torch.tensor(np.random.choice(h * w, opt.render.rand_sample, p=prob, replace=False))


def dist_backend():
    """'nccl' (= RCCL on ROCm, xGMI inside a node).  SHAPECLIPPER_DIST_BACKEND=gloo runs the same multi-rank code with several ranks
    sharing a GPU, which RCCL refuses (tests on a one-GPU box)."""
    return os.environ.get("SHAPECLIPPER_DIST_BACKEND", "nccl")


def get_rank(opt):
    """Rank of this process: `opt.rank` when the launcher set one, otherwise the reference's single-node convention rank == device
    index (train.py:37-41 spawns worker i on GPU i)."""
    r = opt.get("rank", None) if hasattr(opt, "get") else None
    if r is not None:
        return int(r)
    return opt.device if isinstance(opt.device, int) else 0


def setup(rank, world_size, port_no):
    """One process per GPU; backend 'nccl' is RCCL on ROCm (xGMI inside a node)."""
    dist.init_process_group(dist_backend(), init_method="tcp://127.0.0.1:" + str(port_no), rank=rank, world_size=world_size)


def cleanup():
    dist.destroy_process_group()


def is_port_in_use(port):
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        return s.connect_ex(("127.0.0.1", port)) == 0


class AverageMeter:
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count
