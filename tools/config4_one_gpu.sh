# BASELINE config[4] in function: evaluate.py --eval.vox_res=100 sharded over 8 ranks -- on a one-GPU box the ranks share cuda:0 over gloo
# (SHAPECLIPPER_DIST_BACKEND).  Trains a 1-epoch checkpoint on the synthetic set first, then evaluates it in one process and in 8 ranks and
# compares chamfer.txt.  Usage (GPU box): bash tools/config4_one_gpu.sh ; results under gpurun_out/config4/
export MIOPEN_LOG_LEVEL=1 MIOPEN_FIND_MODE=FAST
R=$PWD; O=$R/gpurun_out/config4; mkdir -p $O
A="--yaml=options/pix3d/config.yaml --name=config4 --output_root=/tmp/sc_config4 --data.dataset=synthetic --data.synthetic_len=32 --batch_size=8 --max_epoch=1 --freq.eval=1 --tb! --arch.enc_pretrained!"
timeout 600 python train.py $A --eval.vox_res=16 --eval.num_points=1000 > $O/train.log 2>&1 || { tail -20 $O/train.log; exit 1; }
D=/tmp/sc_config4/pix3d_output/config4
( time timeout 600 python evaluate.py $A --eval.vox_res=100 --resume ) > $O/eval_1rank.log 2>&1 || { tail -20 $O/eval_1rank.log; exit 1; }
cp $D/chamfer.txt $O/chamfer_1rank.txt; cp $D/f_score.txt $O/f_score_1rank.txt; rm $D/chamfer.txt
export SHAPECLIPPER_DIST_BACKEND=gloo
( time timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29633 evaluate.py $A --eval.vox_res=100 --resume ) > $O/eval_8ranks.log 2>&1 || { tail -20 $O/eval_8ranks.log; exit 1; }
cp $D/chamfer.txt $O/chamfer_8ranks.txt; cp $D/f_score.txt $O/f_score_8ranks.txt
python - <<P
import numpy as np
a, b = np.loadtxt("$O/chamfer_1rank.txt"), np.loadtxt("$O/chamfer_8ranks.txt")
print("samples", a.shape[0], b.shape[0], " max |diff| of (idx, acc, comp) rows: %.3g" % np.abs(a - b).max(), " CD 1 rank %.6f  8 ranks %.6f" % ((a[:, 1].mean() + a[:, 2].mean()) / 2, (b[:, 1].mean() + b[:, 2].mean()) / 2))
print("f_score files identical:", open("$O/f_score_1rank.txt").read() == open("$O/f_score_8ranks.txt").read())
P
grep real $O/eval_1rank.log $O/eval_8ranks.log
