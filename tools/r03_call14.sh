set -e
export TMPDIR=/tmp
O=gpurun_out/r3
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_train_ops.py tests/test_gpu_train.py -m gpu -x -q > $O/t_wh3.log 2>&1 || { tail -60 $O/t_wh3.log; exit 1; }
python3 tools/wgrad_bench.py > $O/wgrad_halo3.log 2>&1
python3 tools/repack_bench.py > $O/repack_bench.log 2>&1
python3 bench.py --mode train --steps 5 --warmup 2 > $O/bench_train_v5.log 2>&1
echo done
