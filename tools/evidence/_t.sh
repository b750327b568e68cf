set -e
mkdir -p gpurun_out/r3b
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "gn or groupnorm or resblock or train or norm" > gpurun_out/r3b/t_gn.log 2>&1 || { tail -40 gpurun_out/r3b/t_gn.log; exit 1; }
tail -3 gpurun_out/r3b/t_gn.log
python3 tools/gn_bench.py > gpurun_out/r3b/gn_bench.log 2>&1
cat gpurun_out/r3b/gn_bench.log | grep -v amdgpu.ids
python3 tools/profile_train.py > gpurun_out/r3b/pt_gn.log 2>&1
head -12 gpurun_out/r3b/pt_gn.log
