"""Host-side profile of evaluate.py's per-sample loop (cProfile, top cumulative entries): python tools/prof_eval_host.py [vox_res=100] [n=8]"""
import cProfile, os, pstats, sys, io
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from shapeclipper_amd.model.runner import Runner
from shapeclipper_amd.utils import options
vox = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
o = options.set(options.parse_arguments(["--yaml=options/pix3d/config.yaml", "--name=prof_eval", "--output_root=/tmp/sc_prof_eval", "--arch.enc_pretrained!",
                                         "--data.dataset=synthetic", "--data.synthetic_len=%d" % n, "--eval.vox_res=%d" % vox, "--tb!"]), verbose=False)
o.device, o.world_size, o.port = 0, 1, 0
torch.manual_seed(0)
r = Runner(o)
r.load_dataset(o, eval_split="test")
r.build_networks(o)
r.evaluate(o, ep=0)            # warm-up (allocator, MIOpen find)
pr = cProfile.Profile()
pr.enable()
r.evaluate(o, ep=0)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(35)
print("\n".join(l[:200] for l in s.getvalue().splitlines()))
