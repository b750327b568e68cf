set -e
export TMPDIR=/tmp
O=gpurun_out/r3
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_train_ops.py -m gpu -x -q -s -k "wgrad" > $O/t_wh2.log 2>&1 || { tail -60 $O/t_wh2.log; exit 1; }
python3 tools/wgrad_bench.py > $O/wgrad_halo2.log 2>&1
timeout -k 10 900 python3 -m pytest tests/test_gpu_train.py tests/test_gpu_optim.py -m gpu -x -q > $O/t_train2.log 2>&1 || { tail -60 $O/t_train2.log; exit 1; }
python3 bench.py --mode train --steps 5 --warmup 2 > $O/bench_train_v4.log 2>&1
echo done
