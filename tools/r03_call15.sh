set -e
export TMPDIR=/tmp
O=gpurun_out/r3
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_optim.py tests/test_gpu_train.py -m gpu -x -q -s > $O/t_opt2.log 2>&1 || { tail -60 $O/t_opt2.log; exit 1; }
CTSI_CONV_RING=1 python3 tools/profile_ops.py --net unet > $O/ops_unet_ring.log 2>&1
python3 tools/profile_ops.py --net unet > $O/ops_unet_noring.log 2>&1
echo done
