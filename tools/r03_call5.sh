set -e
export TMPDIR=/tmp
O=gpurun_out/r3
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_optim.py -m gpu -x -q -s -k "downsample or attention or split_k or optim or repack" > $O/t_ds2.log 2>&1 || { tail -60 $O/t_ds2.log; exit 1; }
python3 tools/profile_ops.py --net unet > $O/ops_unet_v2.log 2>&1
python3 tools/profile_ops.py --net dec > $O/ops_dec_v2.log 2>&1
python3 tools/profile_ops.py --net enc > $O/ops_enc_v2.log 2>&1
python3 bench.py --steps 20 --warmup 5 > $O/bench_v2.log 2>&1
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/t_all2.log 2>&1 || { tail -40 $O/t_all2.log; exit 1; }
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/fetch_calib.hip -o gpurun_out/fetch_calib > $O/calib_build.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/calib -o calib -- gpurun_out/fetch_calib > $O/calib.log 2>&1
python3 tools/fetch_calib_report.py $O/calib/calib_counter_collection.csv $O/fetch_calibration.json > $O/calib_report.log 2>&1
rm -f gpurun_out/fetch_calib
echo done
