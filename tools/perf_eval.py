"""Per-sample evaluation time (BASELINE config[4]: evaluate.py --eval.vox_res=100, eval.batch_size=1): graph forward
(encoders + full-frame render), level grid, iso-surface + 100k surface samples, Chamfer/F-score vs a 100k-point GT cloud.
Usage (GPU box): python tools/perf_eval.py [vox_res]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
import torch
from shapeclipper_amd import synthetic
from shapeclipper_amd.model.graph import Graph
from shapeclipper_amd.utils import eval_3D, options, util
from shapeclipper_amd.utils.util import EasyDict as edict

vox = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
opt = options.set(options.parse_arguments(["--yaml=%s/options/pix3d/config.yaml" % ROOT, "--name=perf_eval", "--output_root=/tmp/sc_perf_eval",
                                           "--tb!", "--arch.enc_pretrained!", "--eval.vox_res=%d" % vox]), verbose=False)
opt.device = 0
torch.manual_seed(0)
graph = Graph(opt).cuda().eval()
synthetic.cap_host_threads()
batch = util.move_to_device(synthetic.make_batch(opt, 1, seed=1, training=False, n_gt_points=100000), "cuda:0")
opt.H, opt.W = opt.eval.image_size

def timed(fn):
    torch.cuda.synchronize(); t0 = time.time(); r = fn(); torch.cuda.synchronize(); return r, (time.time() - t0) * 1e3

def sample():
    t = {}
    with torch.no_grad():
        var, t["graph forward (encoders + %dx%d render)" % (opt.H, opt.W)] = timed(lambda: graph(opt, edict(batch), training=False, get_loss=False))
        pts, t["dense grid points"] = timed(lambda: eval_3D.get_dense_3D_grid(opt, var))
        level, t["level grid %d^3" % (vox + 1)] = timed(lambda: eval_3D.compute_level_grid(opt, graph.sdf_network, var.proj_latent_sdf, pts))
        (dpc, meshes), t["iso-surface + %d samples" % opt.eval.num_points] = timed(
            lambda: eval_3D.surface_points_device(level, opt.eval.range[0], opt.eval.range[1], opt.eval.num_points, seed=0))
        _, t["eval_metrics total (grid + surface + chamfer + f-score)"] = timed(lambda: eval_3D.eval_metrics(opt, var, graph.sdf_network))
    return t, meshes[0].shape[0]

for _ in range(2):
    sample()
# the stages inside eval_metrics, timed through wrappers (each wrapper synchronises: the sum exceeds the un-instrumented total a little)
if os.environ.get("PARTS"):
    parts = {}
    def wrap(mod, name):
        fn = getattr(mod, name)
        def w(*a, **k):
            r, ms = timed(lambda: fn(*a, **k))
            parts[name] = parts.get(name, 0.0) + ms
            return r
        setattr(mod, name, w)
    for name in ("get_dense_3D_grid", "compute_level_grid", "surface_points_device", "normalize_pc", "chamfer_distance", "compute_fscore"):
        wrap(eval_3D, name)
    for _ in range(3):
        parts.clear()
        with torch.no_grad():
            var = graph(opt, edict(batch), training=False, get_loss=False)
            _, tot = timed(lambda: eval_3D.eval_metrics(opt, var, graph.sdf_network))
    print("eval_metrics %.2f ms:" % tot, ", ".join("%s %.2f" % kv for kv in parts.items()))
    sys.exit(0)
acc = {}
N = 5
t0 = time.time()
for _ in range(N):
    t, ntri = sample()
    for k, v in t.items():
        acc[k] = acc.get(k, 0.0) + v / N
torch.cuda.synchronize()
for k, v in acc.items():
    print("%8.2f ms  %s" % (v, k))
print("triangles:", ntri, " => one evaluation sample = %.1f ms (graph forward + eval_metrics)" %
      (acc[[k for k in acc if k.startswith("graph")][0]] + acc[[k for k in acc if k.startswith("eval_metrics")][0]]))
