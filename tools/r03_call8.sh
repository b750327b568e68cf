set -e
export TMPDIR=/tmp
O=gpurun_out/r3
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_optim.py tests/test_gpu_train.py -m gpu -x -q -s > $O/t_opt.log 2>&1 || { tail -60 $O/t_opt.log; exit 1; }
python3 bench.py --mode train --steps 5 --warmup 2 > $O/bench_train_v2.log 2>&1
echo done
