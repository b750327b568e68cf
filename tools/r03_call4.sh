set -e
export TMPDIR=/tmp
O=gpurun_out/r3
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "downsample or attention or split_k" > $O/t_ds2.log 2>&1 || { tail -40 $O/t_ds2.log; exit 1; }
python3 tools/profile_ops.py --net unet > $O/ops_unet_v2.log 2>&1
python3 tools/profile_ops.py --net dec > $O/ops_dec_v2.log 2>&1
python3 tools/profile_ops.py --net enc > $O/ops_enc_v2.log 2>&1
python3 bench.py --steps 20 --warmup 5 > $O/bench_v2.log 2>&1
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/t_all2.log 2>&1 || { tail -40 $O/t_all2.log; exit 1; }
echo done
