"""Synthetic stand-in dataset with the reference's batch schema (data/pix3d.py:110-228) for offline
smoke runs: `--data.dataset=synthetic`.  The real Pix3D loader is out of scope."""
import torch

from .. import synthetic
from ..utils import util


class Dataset(torch.utils.data.Dataset):
    label2cat = {0: "synthetic"}

    def __init__(self, opt, split):
        self.opt, self.split = opt, split
        self.n = int(opt.data.get("synthetic_len", 64 if split == "train" else 4))
        self.training = split == "train"
        synthetic.cap_host_threads()         # once, from the thread that builds the loaders (see synthetic.make_batch)

    def __len__(self):
        return self.n

    def __getitem__(self, idx):
        b = synthetic.make_batch(self.opt, 1, seed=idx + (0 if self.training else 100000), training=True,
                                 n_gt_points=self.opt.eval.num_points if not self.training else 1024)
        out = {}
        for k, v in b.items():
            out[k] = {kk: vv[0] for kk, vv in v.items()} if isinstance(v, dict) else v[0]
        out["idx"] = idx
        if not self.training:   # eval renders the full image: [H*W] targets instead of sampled rays
            H, W = self.opt.image_size
            flat = lambda m: m.flatten(1).permute(1, 0).contiguous()
            out["rgb_input"], out["mask_input"], out["normal_input"] = flat(out["rgb_input_map"]), flat(out["mask_input_map"]), flat(out["normal_input_map"])
            out.pop("ray_idx", None)
        return out

    def setup_loader(self, opt, shuffle=False, drop_last=True, subcat=None, batch_size=None, allow_ddp=True):
        sampler = None
        if self.training and allow_ddp and opt.get("world_size", 1) > 1:
            sampler = torch.utils.data.distributed.DistributedSampler(self, num_replicas=opt.world_size,
                                                                      rank=util.get_rank(opt))
        return torch.utils.data.DataLoader(self, batch_size=batch_size or opt.batch_size, num_workers=0,
                                           shuffle=shuffle if sampler is None else False, drop_last=drop_last, sampler=sampler)

    def id_filename_mapping(self, opt, outpath):
        with open(outpath, "w") as f:
            for i in range(self.n):
                f.write("{} synthetic_{}\n".format(i, i))
