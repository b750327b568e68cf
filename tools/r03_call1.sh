set -e
export TMPDIR=/tmp
O=gpurun_out/r3
mkdir -p $O
python3 tools/profile_ops.py --net dec > $O/ops_dec.log 2>&1
python3 tools/profile_ops.py --net enc > $O/ops_enc.log 2>&1
python3 tools/profile_ops.py --net unet > $O/ops_unet.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_dec -o dec -- python3 tools/profile_ops.py --net dec --repeats 1 > $O/prof_dec.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_enc -o enc -- python3 tools/profile_ops.py --net enc --repeats 1 > $O/prof_enc.log 2>&1
python3 bench.py --model legacy163 --steps 20 --warmup 5 > $O/bench_legacy163.log 2>&1
CTSI_SHARD_CAPTURE=1 python3 bench.py --mode shard --steps 20 --warmup 5 > $O/bench_shard1.log 2>&1
python3 bench.py --batch 4 --steps 10 --warmup 3 --no-cpu > $O/bench_batch4.log 2>&1
echo done
