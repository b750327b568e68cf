set -e
export TMPDIR=/tmp
O=gpurun_out/r3
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "downsample or conv_family or split_k" > $O/t_ds.log 2>&1 || { tail -40 $O/t_ds.log; exit 1; }
timeout -k 10 900 python3 -m pytest tests/test_gpu_configs45.py tests/test_gpu_network.py -m gpu -x -q -s > $O/t_new.log 2>&1 || { tail -40 $O/t_new.log; exit 1; }
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q --deselect tests/test_gpu_configs45.py --deselect tests/test_gpu_network.py > $O/t_all.log 2>&1 || { tail -40 $O/t_all.log; exit 1; }
python3 tools/vae_wall.py > $O/vae_wall.log 2>&1
python3 tools/profile_ops.py --net unet > $O/ops_unet_ds.log 2>&1
python3 tools/profile_ops.py --net enc > $O/ops_enc_ds.log 2>&1
echo done
