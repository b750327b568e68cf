set -e
export TMPDIR=/tmp
O=gpurun_out/r3
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_train_ops.py -m gpu -x -q -s -k "wgrad" > $O/t_wh.log 2>&1 || { tail -60 $O/t_wh.log; exit 1; }
python3 tools/wgrad_bench.py > $O/wgrad_halo.log 2>&1
CTSI_WGRAD_HALO_TARGET=128 python3 tools/wgrad_bench.py > $O/wgrad_halo_t128.log 2>&1
CTSI_WGRAD_HALO_TARGET=512 python3 tools/wgrad_bench.py > $O/wgrad_halo_t512.log 2>&1
python3 tools/wgrad_bench.py --zero > $O/wgrad_halo_zero.log 2>&1
echo done
