"""Training / evaluation engine -- call surface of the reference's model/runner.py
(Runner: load_dataset, build_networks, setup_optimizer, restore_checkpoint, setup_visualizer,
train, evaluate, ...).  Differences that matter on MI355X:

  * data parallelism is one process per GPU with ONE flat RCCL all-reduce of all gradients per step
    (shapeclipper_amd/parallel.py) instead of DistributedDataParallel's bucketed reducer;
  * the per-loss `.mean()` + isnan/isinf asserts (two host syncs per loss key, runner.py:294-305) are
    folded into a single device-side check that is read once per step;
  * TensorBoard / image dumps are optional (used when importable), never required by the step.
"""
from __future__ import annotations

import contextlib
import importlib
import os
import shutil
import time
from copy import deepcopy

import numpy as np
import torch
import tqdm

from ..parallel import FlatGradAllReduce, ModuleHolder
from ..utils import eval_3D, util
from ..utils.util import EasyDict as edict
from ..utils.util import cleanup, log, setup

try:
    from ..utils import util_vis
except Exception:  # pragma: no cover
    util_vis = None


def _rank0(opt):
    if opt.get("rank", None) is not None:      # set by the launcher when ranks and device indices are numbered differently
        return int(opt.rank) == 0
    return opt.device == 0 or opt.device in ("cuda:0", "cpu")


class Runner:

    def __init__(self, opt):
        if os.path.isdir(opt.output_path) and opt.resume is False and _rank0(opt):
            for fn in os.listdir(opt.output_path):
                if "tfevents" in fn:
                    os.remove(os.path.join(opt.output_path, fn))
                if "vis" in fn:
                    shutil.rmtree(os.path.join(opt.output_path, fn), ignore_errors=True)
        if _rank0(opt):
            os.makedirs(opt.output_path, exist_ok=True)
        if "world_size" not in opt:
            opt.world_size = 1
        if opt.world_size > 1 and not torch.distributed.is_initialized():
            setup(opt.device, opt.world_size, opt.port)
        opt.batch_size = opt.batch_size // opt.world_size
        self.optimizer = getattr(torch.optim, opt.optim.algo)
        self.tb = None
        self.reducer = None

    # ---- data -----------------------------------------------------------------------------------------
    def load_dataset(self, opt, eval_split="val"):
        data = importlib.import_module("data.{}".format(opt.data.dataset))
        if _rank0(opt): log.info("loading training data...")
        self.train_data = data.Dataset(opt, split="train")
        self.train_loader = self.train_data.setup_loader(opt, shuffle=True)
        self.num_batches = len(self.train_loader)
        if _rank0(opt): log.info("loading test data...")
        self.test_data = data.Dataset(opt, split=eval_split)
        self.test_loader = self.test_data.setup_loader(opt, shuffle=False, drop_last=False, batch_size=opt.eval.batch_size)
        self.viz_data = []

    # ---- networks / optimisers ------------------------------------------------------------------------
    def build_networks(self, opt):
        if _rank0(opt): log.info("building networks...")
        name = "pretrainer" if opt.pretrain else "graph"
        module = importlib.import_module("shapeclipper_amd.model.{}".format(name))
        dev = torch.device("cuda", opt.device) if isinstance(opt.device, int) else torch.device(opt.device)
        # `--hip.reserve_cus=N` (default 0): in a multi-GPU step the persistent one-workgroup-per-CU grids of the trunk convolutions are sized
        # for (CUs - N), leaving N compute units to RCCL's all-reduce kernels.  Set before any workspace is sized;
        # single-GPU runs keep every CU.
        reserve = int(opt.get("hip", {}).get("reserve_cus", 0) or 0)
        if dev.type == "cuda":
            from .. import ops
            ops.set_reserved_cus(reserve if opt.world_size > 1 else 0)
        self.graph = ModuleHolder(module.Graph(opt).to(dev))
        if opt.world_size > 1:
            # Default (round 4): north_star's SINGLE flat all-reduce after backward.  `--hip.overlap_allreduce` lays the buffer out
            # [early 94.5 % | late] and issues the early segment from inside backward on a side stream; on one rank its hooks and stream
            # juggling cost 1.1 ms of host time per step more than the single collective (profiles/r04_overlap_1rank.txt) on a step that
            # is host-paced, against ~0.7-1 ms of xGMI time it could hide: not the default until a multi-GPU run says otherwise.
            self.reducer = FlatGradAllReduce(self.graph.module, opt.world_size,
                                             overlap=None if opt.get("hip", {}).get("overlap_allreduce", False) else False)

    def setup_optimizer(self, opt):
        if _rank0(opt): log.info("setting up optimizers...")
        kwargs = {k: (tuple(v) if k == "betas" else v) for k, v in opt.optim.params.items()}
        on_gpu = next(self.graph.parameters()).is_cuda
        if on_gpu and self.optimizer in (torch.optim.Adam, torch.optim.AdamW) and opt.get("hip", {}).get("fused_adam", True):
            kwargs.setdefault("fused", True)       # one multi-tensor kernel instead of ~90 foreach launches per step
        full, view = [], []
        for k, v in self.graph.named_parameters():
            full.append(v)
            if "estimator" in k:
                view.append(v)
        self.optim_full = self.optimizer([dict(params=full, lr=opt.optim.lr)], **kwargs)
        self.optim_V = self.optimizer([dict(params=view, lr=opt.optim.lr)], **kwargs)
        # hip.guarded_step: the fused optimizer takes the step's NaN / Inf flag as its `found_inf` tensor (the hook torch's GradScaler uses) and
        # skips the update ON THE DEVICE; the host then reads the flag one step late instead of waiting for it before every optim.step()
        self._guarded_step = bool(kwargs.get("fused")) and bool(opt.get("hip", {}).get("guarded_step", True))

    def restore_checkpoint(self, opt, best=False, evaluate=False):
        epoch_start, iter_start = None, None
        if opt.resume:
            if _rank0(opt): log.info("resuming from previous checkpoint...")
            epoch_start, iter_start, self.best_val = util.restore_checkpoint(
                opt, self, resume=opt.resume, best=best if opt.data.dataset != "openimage" else False, evaluate=evaluate)
        elif opt.load is not None:
            if _rank0(opt): log.info("loading weights from checkpoint {}...".format(opt.load))
            epoch_start, iter_start, _ = util.restore_checkpoint(opt, self, load_name=opt.load)
        elif _rank0(opt):
            log.info("initializing weights from scratch...")
        self.epoch_start = epoch_start or 0
        self.iter_start = iter_start or 0

    def setup_visualizer(self, opt):
        if _rank0(opt) and opt.tb:
            try:
                from torch.utils.tensorboard import SummaryWriter
                self.tb = SummaryWriter(log_dir=opt.output_path, flush_secs=10)
            except Exception:
                log.info("tensorboard not available; scalar logging goes to stdout only")

    # ---- training -------------------------------------------------------------------------------------
    def train(self, opt):
        if _rank0(opt): log.title("TRAINING START")
        self.graph.module.estimator.reset_scales()
        self.timer = edict(start=time.time(), it_mean=None)
        self.iter_skip = self.iter_start % len(self.train_loader)
        self.it = self.iter_start
        if not opt.resume:
            self.best_val, self.best_ep = np.inf, 1
        if self.iter_start == 0 and _rank0(opt) and opt.freq.eval:
            self.evaluate(opt, ep=0, training=True)
        for self.ep in range(self.epoch_start, opt.max_epoch):
            self.train_epoch(opt)
        if _rank0(opt):
            self.save_checkpoint(opt, ep=self.ep + 1, it=self.it, best_val=self.best_val)
            if self.tb is not None:
                self.tb.flush(); self.tb.close()
            log.title("TRAINING DONE")
            log.info("Best CD: %.4f @ epoch %d" % (self.best_val, self.best_ep))
        if opt.world_size > 1:
            cleanup()

    def train_epoch(self, opt):
        if opt.world_size > 1:
            torch.distributed.barrier()
            if hasattr(self.train_loader, "sampler") and hasattr(self.train_loader.sampler, "set_epoch"):
                self.train_loader.sampler.set_epoch(self.ep)
        progress = tqdm.tqdm(range(self.num_batches), desc="training epoch {}".format(self.ep + 1), leave=False) \
            if _rank0(opt) else range(self.num_batches)
        self.graph.train()
        loader = iter(self.train_loader)
        loss = None
        for _ in progress:
            if self.iter_skip > 0:          # fast-forward after --resume
                self.iter_skip -= 1
                continue
            var = edict(next(loader))
            opt.H, opt.W = opt.image_size
            var = util.move_to_device(var, opt.device)
            loss = self.train_iteration(opt, var, progress)
        self.check_finite()                   # the last iteration's deferred NaN/Inf check
        if _rank0(opt) and loss is not None:
            log.loss_train(opt, self.ep + 1, opt.optim.lr, loss, self.timer)
        if (self.ep + 1) % opt.freq.eval == 0 and _rank0(opt):
            val = self.evaluate(opt, ep=self.ep + 1, training=True)
            if val < self.best_val:
                self.best_val, self.best_ep = val, self.ep + 1
                self.save_checkpoint(opt, ep=self.ep + 1, it=self.it, best_val=self.best_val, best=True, latest=True)

    def train_iteration(self, opt, var, loader=None):
        self.timer.it_start = time.time()
        frozen_keys = []
        if self.it > opt.optim.iter_camera:
            optim = self.optim_full
        else:                                  # camera warm-up: only the estimator learns
            for m in self.graph.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.eval()
            optim = self.optim_V
            frozen_keys = ["nearest_img", "nearest_mask", "nearest_normal", "eikonal"]
        if self.reducer is not None:
            self.reducer.zero_grad()
            self.reducer.broadcast_buffers()
        else:
            optim.zero_grad()
        var, loss = self.graph.forward(opt, var, training=True, get_loss=True)
        loss = self.summarize_loss(opt, var, loss, non_act_loss_key=frozen_keys, defer_check=True)
        loss.all.backward()
        if self.reducer is not None:
            self.reducer.all_reduce()
        # The reference asserts on NaN / Inf losses before it back-propagates (runner.py:296-302).  Here the flag left for pinned memory
        # right after the forward pass.  Guarded step (default with the fused optimizer): the flag is also the optimizer's `found_inf`, so a
        # poisoned step leaves the weights, the Adam moments and the step counters untouched without the host knowing yet; the host reads
        # the flag of the PREVIOUS step here (complete long ago: no wait -- the per-step wait cost 1.0 ms of 36.6 at bs32, 0.3 of 22.0 at
        # bs16, measured round 4 with tools/step_times.py) and raises the reference's assertion one step late, before this step's update is applied.
        # Otherwise (foreach / CPU optimizers): wait for this step's flag now, before optim.step().
        if getattr(self, "_guarded_step", False):
            optim.grad_scale, optim.found_inf = None, getattr(self, "_step_found_inf", None)
            self.check_finite(keep_last=True)
        else:
            self.check_finite()
        optim.step()

        # Never checkpoint a step whose losses were not verified -- and EVERY rank drains its pending flags at a checkpoint iteration, not
        # only rank 0: the flag is MAX-reduced over the ranks, so all of them raise here together instead of rank 0 dying alone
        # and the others walking into the next step's collectives against a dead peer.
        if (self.it + 1) % opt.freq.ckpt_latest == 0:
            self.check_finite()
        if _rank0(opt):
            if (self.it + 1) % opt.freq.ckpt_latest == 0:
                self.save_checkpoint(opt, ep=self.ep, it=self.it + 1, best_val=self.best_val, latest=True)
            if opt.freq.scalar and self.it % opt.freq.scalar == 0 and self.tb is not None:
                self.log_scalars(opt, var, loss, step=self.it, split="train")
                self.tb.add_scalar("train/beta", self.graph.module.renderer.density.beta, global_step=self.it)
        self.it += 1
        if loader is not None and hasattr(loader, "set_postfix") and self.it % 10 == 0:
            loader.set_postfix(it=self.it, loss="{:.3f}".format(float(loss.all)))
        self.timer.it_end = time.time()
        util.update_timer(opt, self.timer, self.ep if hasattr(self, "ep") else 0, len(loader) if loader is not None else 1)
        return loss

    def summarize_loss(self, opt, var, loss, non_act_loss_key=[], defer_check=False):
        """all = sum_k float(w_k) * loss_k (keys in non_act_loss_key weigh 0).  NaN/Inf are checked with ONE
        host read of a device flag instead of two syncs per key (reference runner.py:296-302).  With defer_check the
        flag travels to pinned host memory asynchronously and `check_finite(loss)` raises the same assertions later
        (train_iteration calls it after the backward pass is queued and before the optimizer step)."""
        assert "all" not in loss
        total, bad = 0., None
        keys = []
        for key in loss:
            assert key in opt.loss_weight
            if opt.loss_weight[key] is not None:
                keys.append(key)
        weights = tuple(0.0 if key in non_act_loss_key else float(opt.loss_weight[key]) for key in keys)
        values = [loss[key] if loss[key].dim() == 0 else loss[key].mean() for key in keys]
        from .. import ops
        if values and all(v.is_cuda for v in values) and len(values) <= ops.LOSS_TOTAL_MAX_TERMS:
            # the weighted sum (same order of additions) and the NaN/Inf flag of every term in one launch (csrc/camera_prior.hip)
            from ..functional import LossTotalFunction
            total, bad = LossTotalFunction.apply(weights, *values)
        else:
            for w, value in zip(weights, values):
                flag = ~torch.isfinite(value)
                bad = flag if bad is None else (bad | flag)
                total = total + w * value
        if "_bad_choice" in var and bad is not None:          # NaN neighbour probabilities (np.random.choice would have raised)
            bad = bad | var.pop("_bad_choice").to(bad.device)
        self._step_found_inf = None
        if bad is not None and opt.get("check_finite", True):
            reducer = getattr(self, "reducer", None)
            if defer_check and reducer is not None and reducer.comm:
                # multi-rank: MAX of the flag over the ranks, so that every rank raises (or none does) before optim.step()
                flag = reducer.reduce_flag(bad)
                stream = reducer._comm_stream
                if flag.is_cuda:
                    host = torch.empty((1,), dtype=torch.float32, pin_memory=True)
                    with torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext():
                        host.copy_(flag, non_blocking=True)
                        done = torch.cuda.Event()
                        done.record()
                    self._pending().append((host, done, {k: v.detach() for k, v in loss.items()}))
                    self._step_found_inf = flag.reshape(())  # exactly 1.0 where any rank's loss is NaN / Inf (MAX of 0 / 1 flags)
                elif bool(flag.item() > 0):
                    self._raise_not_finite(loss, any_rank=True)
            elif defer_check and bad.is_cuda:
                host = torch.empty((), dtype=torch.bool, pin_memory=True)
                host.copy_(bad, non_blocking=True)
                done = torch.cuda.Event()
                done.record()
                self._pending().append((host, done, {k: v.detach() for k, v in loss.items()}))
                self._step_found_inf = bad.detach().reshape(()).to(torch.float32)      # 0-dim, as GradScaler's
            elif bool(bad):
                self._raise_not_finite(loss)
        loss.update(all=total)
        return loss

    @staticmethod
    def _raise_not_finite(loss, any_rank=False):
        for key in loss:
            if key == "all":
                continue
            v = loss[key].mean()
            assert not torch.isinf(v), "loss {} is Inf".format(key)
            assert not torch.isnan(v), "loss {} is NaN".format(key)
        # multi-rank: the flag is the MAX over the ranks -- this rank's losses are finite, another rank's are not; every rank stops here
        assert not any_rank, "a loss is NaN / Inf on another rank"

    def _pending(self):
        if not hasattr(self, "_pending_checks"):
            self._pending_checks = []
        return self._pending_checks

    def check_finite(self, loss=None, keep_last=False):
        """Raise the reference's NaN/Inf assertions (runner.py:296-302) for the steps whose flags are pending, oldest first.
        train_iteration calls this between the (queued) backward pass and the optimizer step: each flag was copied to pinned memory
        right after its forward pass.  keep_last (guarded step): the newest flag stays pending -- its step is protected on the device
        by the optimizer's `found_inf` -- so the wait is for the previous step's event, which completed long ago.  Without keep_last
        (end of an epoch, before a checkpoint, unguarded optimizers) every pending flag is waited for."""
        queue = self._pending()
        while len(queue) > (1 if keep_last else 0):
            host, done, pending_loss = queue.pop(0)
            done.synchronize()
            if bool(host.reshape(-1)[0] > 0):
                queue.clear()
                self._raise_not_finite(loss if loss is not None else pending_loss, any_rank=host.dtype != torch.bool)

    # ---- evaluation -----------------------------------------------------------------------------------
    @torch.no_grad()
    def evaluate(self, opt, ep, training=False):
        self.graph.eval()
        opt.H, opt.W = opt.eval.image_size
        f_scores = []
        metric = dict(dist_acc=0., dist_cov=0.)
        C = opt.data.num_classes
        acc_cat, comp_cat, counts = [0.] * C, [0.] * C, [0.001] * C
        loader = tqdm.tqdm(self.test_loader, desc="evaluating", leave=False)
        for it, batch in enumerate(loader):
            var = self.evaluate_batch(opt, edict(batch), ep, it, single_gpu=True)
            dist_acc, dist_cov = eval_3D.eval_metrics(opt, var, self.graph.module.sdf_network)
            f_scores.append(var.f_score)
            for i in range(len(var.idx)):
                c = var.category_label[i].item()
                counts[c] += 1; acc_cat[c] += var.cd_acc[i].item(); comp_cat[c] += var.cd_comp[i].item()
            metric["dist_acc"] += dist_acc * len(var.idx)
            metric["dist_cov"] += dist_cov * len(var.idx)
            loader.set_postfix(CD="{:.3f}".format(float((dist_acc + dist_cov) / 2)))
            if not training:
                self.dump_results(opt, var, ep, write_new=(it == 0))
        if not training:
            with open(os.path.join(opt.output_path, "cd_cat.txt"), "w") as f:
                f.write("CD     Acc    Comp   Count Cat\n")
                names = getattr(self.test_data, "label2cat", {i: str(i) for i in range(C)})
                for i in range(C):
                    a, c = acc_cat[i] / counts[i], comp_cat[i] / counts[i]
                    f.write("%.4f %.4f %.4f %5d %s\n" % ((a + c) / 2, a, c, counts[i], names[i]))
            fs = torch.cat(f_scores, dim=0).mean(dim=0)
            with open(os.path.join(opt.output_path, "f_score.txt"), "w") as f:
                for i, th in enumerate(opt.eval.f_thresholds):
                    line = "F-score @ %.2f: %.4f" % (th * 100, fs[i].item())
                    print(line); f.write(line + "\n")
        n = max(len(self.test_data), 1)
        for k in metric:
            metric[k] /= n
        log.loss_eval(opt, loss=None, chamfer=(metric["dist_acc"], metric["dist_cov"]))
        opt.H, opt.W = opt.image_size
        return float((metric["dist_acc"] + metric["dist_cov"]) / 2)

    @torch.no_grad()
    def evaluate_sharded(self, opt, ep=0):
        """Evaluation over all ranks (BASELINE config[4]; the reference evaluates on one GPU, evaluate.py:16-18):
        rank r takes the test samples with index % world == r (eval.batch_size = 1), per-sample records are
        gathered once, rank 0 writes chamfer.txt / cd_cat.txt / f_score.txt in sample order."""
        from ..parallel import gather_eval_records
        self.graph.eval()
        opt.H, opt.W = opt.eval.image_size
        # the reference's single-node convention is rank == device index; the process group's rank is the same number there and stays
        # right when ranks and devices are numbered differently (several nodes; tests/test_gpu_two_ranks.py: two ranks on one GPU)
        rank = torch.distributed.get_rank() if torch.distributed.is_initialized() else util.get_rank(opt)
        recs = []
        for it in range(rank, len(self.test_data), opt.world_size):
            sample = self.test_data[it]
            batch = {k: ({kk: vv[None] for kk, vv in v.items()} if isinstance(v, dict) else torch.as_tensor(v)[None]) for k, v in sample.items()}
            var = self.evaluate_batch(opt, edict(batch), ep, it, single_gpu=True)
            eval_3D.eval_metrics(opt, var, self.graph.module.sdf_network)
            recs.append(torch.cat([var.idx.float().view(1), var.cd_acc.view(1), var.cd_comp.view(1), var.f_score.view(-1),
                                   var.category_label.float().view(1)]))
        dev = next(self.graph.parameters()).device
        records = torch.stack(recs) if recs else torch.zeros(0, 10, device=dev)
        allr = gather_eval_records(records.to(dev), opt.world_size).cpu()
        opt.H, opt.W = opt.image_size
        if rank == 0:
            with open("{}/chamfer.txt".format(opt.output_path), "w") as f:
                for r in allr:
                    f.write("{} {:.8f} {:.8f}\n".format(int(r[0]), r[1], r[2]))
            C = opt.data.num_classes
            names = getattr(self.test_data, "label2cat", {i: str(i) for i in range(C)})
            with open(os.path.join(opt.output_path, "cd_cat.txt"), "w") as f:
                f.write("CD     Acc    Comp   Count Cat\n")
                for c in range(C):
                    sel = allr[allr[:, 9] == c]
                    n = sel.shape[0] + 0.001
                    a_, c_ = float(sel[:, 1].sum()) / n, float(sel[:, 2].sum()) / n
                    f.write("%.4f %.4f %.4f %5d %s\n" % ((a_ + c_) / 2, a_, c_, n, names[c]))
            fs = allr[:, 3:9].mean(dim=0)
            with open(os.path.join(opt.output_path, "f_score.txt"), "w") as f:
                for i, th in enumerate(opt.eval.f_thresholds):
                    f.write("F-score @ %.2f: %.4f\n" % (th * 100, fs[i].item()))
            log.loss_eval(opt, loss=None, chamfer=(allr[:, 1].mean(), allr[:, 2].mean()))
        return float((allr[:, 1].mean() + allr[:, 2].mean()) / 2) if allr.shape[0] else float("nan")

    def evaluate_batch(self, opt, var, ep=None, it=None, single_gpu=False, visualize=False):
        var = util.move_to_device(var, opt.device)
        return self.graph.module(opt, var, training=False, visualize=visualize, get_loss=False)

    def vis_rotate(self, opt, var, n_views=50, vis_NN=False):
        B = len(var.idx)
        imgs, masks, normals = [], [], []
        as_map = lambda x, c: x.view(B, opt.H, opt.W, c).permute(0, 3, 1, 2).contiguous()
        for i in range(n_views):
            pose_i = var.vis_pose[i].unsqueeze(0).expand(B, -1, -1)
            rgb, mask, _, _, normal, _ = self.graph.module.renderer(
                opt, pose_i, var.intr, torch.ones_like(var.scale_dist), var.proj_latent_sdf,
                var.proj_latent_rgb_NN if vis_NN else var.proj_latent_rgb, training=False)
            imgs.append(as_map(rgb, 3)); masks.append(as_map(mask, 1)); normals.append(as_map(normal, 3) / 2 + 0.5)
        var.rotating_imgs, var.rotating_masks, var.rotating_normals = imgs, masks, normals

    @torch.no_grad()
    def log_scalars(self, opt, var, loss, metric=None, step=0, split="train"):
        if self.tb is None:
            return
        for key, value in loss.items():
            if key != "all":
                self.tb.add_scalar("{0}/loss_{1}".format(split, key), value.mean(), step)
        for key, value in (metric or {}).items():
            self.tb.add_scalar("{0}/{1}".format(split, key), value, step)

    @torch.no_grad()
    def dump_results(self, opt, var, ep, write_new=False, train=False):
        folder = "dump" if not train else "vis_{}".format(ep)
        os.makedirs("{}/{}/".format(opt.output_path, folder), exist_ok=True)
        if util_vis is not None:
            util_vis.dump_images(opt, var.idx, "image_recon", var.rgb_recon_map, masks=var.mask_hard_map, folder=folder)
            util_vis.dump_images(opt, var.idx, "mask_recon", var.mask_recon_map, folder=folder)
        if not train:
            with open("{}/chamfer.txt".format(opt.output_path), "w" if write_new else "a") as f:
                for i, acc, comp in zip(var.idx, var.cd_acc, var.cd_comp):
                    f.write("{} {:.8f} {:.8f}\n".format(i, acc, comp))

    def save_checkpoint(self, opt, ep=0, it=0, best_val=np.inf, latest=False, best=False):
        assert _rank0(opt)
        util.save_checkpoint(opt, self, ep=ep, it=it, best_val=best_val, latest=latest, best=best)
        if not latest:
            log.info("checkpoint saved: ({0}) {1}, epoch {2} (iteration {3})".format(opt.group, opt.name, ep, it))
        if best:
            log.info("Saving the current model as the best...")
