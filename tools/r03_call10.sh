set -e
export TMPDIR=/tmp
O=gpurun_out/r3
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_train.py tests/test_gpu_train_ops.py tests/test_gpu_optim.py -m gpu -x -q > $O/t_train.log 2>&1 || { tail -60 $O/t_train.log; exit 1; }
python3 bench.py --mode train --steps 5 --warmup 2 > $O/bench_train_v3.log 2>&1
python3 tools/profile_train.py > $O/profile_train_v3.log 2>&1
timeout -k 10 600 python3 -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "training" > $O/t_train_full.log 2>&1 || { tail -60 $O/t_train_full.log; exit 1; }
echo done
