set -e
mkdir -p gpurun_out/r3b
timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r3b/t_ops2.log 2>&1 || { tail -40 gpurun_out/r3b/t_ops2.log; exit 1; }
tail -2 gpurun_out/r3b/t_ops2.log
for a in "--cin 128 --cout 128 --dhw 48 128 128" "--cin 256 --cout 256 --dhw 48 64 64"; do python3 tools/k32_stamps.py $a 2>&1 | grep -v "amdgpu.ids\|whole launch"; done > gpurun_out/r3b/stamps_v2.log
cat gpurun_out/r3b/stamps_v2.log
B=$PWD/tools/evidence/libctsi_base.so
python3 bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/r3b/ab2_new1.log 2>&1
CTSI_LIB=$B python3 bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/r3b/ab2_base1.log 2>&1
python3 bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/r3b/ab2_new2.log 2>&1
CTSI_LIB=$B python3 bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/r3b/ab2_base2.log 2>&1
