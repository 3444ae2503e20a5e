"""Sphere-SDF pre-training (reference model/pretrainer.py): random latents -> SDF MLP fitted to
|x| - radius on uniform points, plus an azimuth-uniformity prior on the estimator.  The only hot-path
call is SDFNetwork.get_conditional_output(compute_grad=False) -> sdf_fwd.hip / sdf_bwd.hip."""
from __future__ import annotations

import os
import shutil
import time

import numpy as np
import torch
import tqdm

from ..parallel import ModuleHolder
from ..utils import util
from ..utils.util import EasyDict as edict
from ..utils.util import log
from . import graph


class Graph(graph.Graph):

    def forward(self, opt, var):
        dev = next(self.parameters()).device
        var.latent_raw = torch.randn(opt.batch_size, opt.arch.latent_dim_shape + opt.arch.latent_dim_rgb, device=dev) * opt.pre.latent_std
        var.latent_shape = var.latent_raw[:, :opt.arch.latent_dim_shape]
        var.latent_rgb = var.latent_raw[:, opt.arch.latent_dim_shape:]
        var.proj_latent_sdf = self.latent_proj_shape(var.latent_shape)
        if opt.pre.viewpoint:
            # empirical azimuth distribution vs uniform grid, p=1 sliced Wasserstein (pretrainer.py:130-158)
            saved = opt.reg.emd_p
            opt.reg.emd_p = 1
            var.w_dist = self.loss_fns.cam_uniform_loss(opt, self.estimator(var.rgb_input_map)[0])
            opt.reg.emd_p = saved
        return var, self.compute_loss(opt, var)

    def compute_loss(self, opt, var):
        loss = edict(all=0)
        if opt.pre.density:
            loss.all = loss.all + self.get_sdf_loss(opt, var.proj_latent_sdf.device, var.proj_latent_sdf)
        if opt.pre.viewpoint:
            loss.all = loss.all + var.w_dist
        return loss

    def get_sdf_loss(self, opt, device, proj_latent_sdf):
        lo, hi = opt.pre.sample_range
        pts = torch.rand(opt.batch_size * opt.pre.sample_points, 3, device=device) * (hi - lo) + lo
        sdf = self.sdf_network.get_conditional_output(opt, opt.batch_size, pts, proj_latent_sdf, compute_grad=False)[0]
        return self.loss_fns.MSE_loss(sdf, pts.norm(dim=-1, keepdim=True) - opt.pre.radius)


class Runner:

    def __init__(self, opt):
        if os.path.isdir(opt.output_path) and opt.resume is False:
            for fn in os.listdir(opt.output_path):
                if "vis" in fn:
                    shutil.rmtree(os.path.join(opt.output_path, fn), ignore_errors=True)
        os.makedirs(opt.output_path, exist_ok=True)
        self.optimizer = getattr(torch.optim, opt.optim.algo)

    def load_dataset(self, opt, eval_split="train"):
        import importlib
        data = importlib.import_module("data.{}".format(opt.data.dataset))
        log.info("loading pretrain data...")
        self.pretrain_data = data.Dataset(opt, split="train")
        self.pretrain_loader = self.pretrain_data.setup_loader(opt, shuffle=True, batch_size=opt.batch_size, allow_ddp=False)

    def build_networks(self, opt):
        log.info("building networks...")
        self.graph = ModuleHolder(Graph(opt).to(opt.device))

    def setup_optimizer(self, opt):
        kwargs = {k: (tuple(v) if k == "betas" else v) for k, v in opt.optim.params.items()}
        params = [v for k, v in self.graph.named_parameters() if not ("estimator" in k and "fc" not in k)]
        self.optim = self.optimizer([dict(params=params, lr=opt.optim.lr)], **kwargs)

    def train(self, opt):
        log.title("TRAINING START")
        self.timer = edict(start=time.time(), it_mean=None)
        self.ep, self.it = 0, 0
        self.graph.train()
        self.save_checkpoint(opt, ep=self.ep, it=self.it + 1, latest=True)
        loader = iter(self.pretrain_loader)
        progress = tqdm.trange(opt.pre.iter, desc="pretraining", leave=False)
        for _ in progress:
            try:
                batch = next(loader)
            except StopIteration:
                loader = iter(self.pretrain_loader)
                batch = next(loader)
            var = util.move_to_device(edict(batch), opt.device)
            self.train_iteration(opt, var, progress)
        self.save_checkpoint(opt, ep=1, it=self.it)
        log.title("TRAINING DONE")

    def train_iteration(self, opt, var, loader):
        self.timer.it_start = time.time()
        self.optim.zero_grad()
        var, loss = self.graph.forward(opt, var)
        loss.all.backward()
        self.optim.step()
        if (self.it + 1) % opt.freq.ckpt_latest == 0:
            self.save_checkpoint(opt, ep=self.ep, it=self.it + 1, latest=True)
        self.it += 1
        if hasattr(loader, "set_postfix"):
            loader.set_postfix(it=self.it, loss="{:.3f}".format(float(loss.all)))
        self.timer.it_end = time.time()
        util.update_timer(opt, self.timer, self.ep, len(loader))
        return loss

    def save_checkpoint(self, opt, ep=0, it=0, latest=False):
        children = ("reconstructor", "sdf_network", "latent_proj_shape") + (("estimator",) if opt.pre.viewpoint else ())
        util.save_checkpoint(opt, self, ep=ep, it=it, best_val=np.inf, children=children)
        if not latest:
            log.info("checkpoint saved: ({0}) {1}, epoch {2} (iteration {3})".format(opt.group, opt.name, ep, it))
