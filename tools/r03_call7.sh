set -e
export TMPDIR=/tmp
O=gpurun_out/r3
mkdir -p $O
C=${CTSI_COMMIT:-41099d4+}
# training path: bench line, kernel stats of the same command, PMC traffic / MFMA busy over tools/profile_train.py
python3 bench.py --mode train --steps 5 --warmup 2 > $O/bench_train_v1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o train -- python3 bench.py --mode train --steps 5 --warmup 2 --no-roofline > $O/prof_train.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_tfetch -o fetch -- python3 tools/profile_train.py --repeats 1 > $O/pmc_tfetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_twrite -o write -- python3 tools/profile_train.py --repeats 1 > $O/pmc_twrite.log 2>&1
python3 tools/pmc_traffic.py $O/pmc_tfetch/fetch_counter_collection.csv $O/pmc_twrite/write_counter_collection.csv $O/r03_pmc_traffic_train.json $C "python tools/profile_train.py --repeats 1 (config-3 U-Net forward + backward, B = 4, latent 48^3)" > $O/pmc_traffic_train.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_tmfma -o mfma -- python3 tools/profile_train.py --repeats 1 > $O/pmc_tmfma.log 2>&1
python3 tools/pmc_mfma.py $O/pmc_tmfma/mfma_counter_collection.csv $O/r03_pmc_mfma_busy_train.json $C > $O/pmc_tmfma_report.log 2>&1
python3 tools/profile_train.py > $O/profile_train.log 2>&1
python3 bench.py --model legacy163 --steps 20 --warmup 5 > $O/bench_legacy163_v2.log 2>&1
CTSI_SHARD_CAPTURE=1 python3 bench.py --mode shard --steps 20 --warmup 5 > $O/bench_shard1_v2.log 2>&1
python3 bench.py --batch 4 --steps 10 --warmup 3 --no-cpu > $O/bench_batch4_v2.log 2>&1
echo done
