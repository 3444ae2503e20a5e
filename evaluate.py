"""python evaluate.py --yaml=options/pix3d/config.yaml --name=<run> --resume --eval.vox_res=100

Single process: same flow as the reference (evaluate.py:11-30).  Under torchrun (RANK / WORLD_SIZE set) the test
set is sharded over the ranks and the per-sample metrics are gathered once (Runner.evaluate_sharded)."""
import contextlib
import os
import sys

import torch

import utils.options as options
from utils.util import is_port_in_use, log
import model.runner

log.process(os.getpid())
log.title("[{}] (evaluating)".format(sys.argv[0]))
opt = options.set(opt_cmd=options.parse_arguments(sys.argv[1:]))
port = 34567
while is_port_in_use(port):
    port += 1
opt.device, opt.world_size, opt.port = 0, 1, port
sharded = "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1
if sharded:
    import torch.distributed as dist
    opt.device, opt.world_size = int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(opt.device)
    dist.init_process_group("nccl")

with (torch.cuda.device(opt.device) if torch.cuda.is_available() else contextlib.nullcontext()):
    evaluator = model.runner.Runner(opt)
    evaluator.load_dataset(opt, eval_split="test")
    evaluator.test_data.id_filename_mapping(opt, os.path.join(opt.output_path, "data_list.txt"))
    evaluator.build_networks(opt)
    evaluator.restore_checkpoint(opt, best=True, evaluate=True)
    evaluator.setup_visualizer(opt)
    if sharded:
        evaluator.reducer = None
        evaluator.evaluate_sharded(opt, ep=0)
        dist.destroy_process_group()
    else:
        evaluator.evaluate(opt, ep=0)
