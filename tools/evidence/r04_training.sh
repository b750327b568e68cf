# Evidence collection of round 4 (run on the GPU box through gpurun): the bench command under rocprofv3 --kernel-trace --stats and the
# separate --pmc passes; the summaries under gpurun_out/r4 were copied into profiles/ by hand.
set -e
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
C=${CTSI_COMMIT:-unknown}   # git is not available on the GPU box: pass the short hash
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train2 -o train -- python3 bench.py --mode train --steps 5 --warmup 2 --no-roofline > $O/prof_train2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_tfetch2 -o fetch -- python3 tools/profile_train.py --repeats 1 > $O/pmc_tfetch2.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_twrite2 -o write -- python3 tools/profile_train.py --repeats 1 > $O/pmc_twrite2.log 2>&1
python3 tools/pmc_traffic.py $O/pmc_tfetch2/fetch_counter_collection.csv $O/pmc_twrite2/write_counter_collection.csv $O/r04_pmc_traffic_train.json $C "python tools/profile_train.py --repeats 1 (config-3 U-Net forward + backward, B = 4, latent 48^3)" > $O/pmc_traffic_train2.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_tmfma2 -o mfma -- python3 tools/profile_train.py --repeats 1 > $O/pmc_tmfma2.log 2>&1
python3 tools/pmc_mfma.py $O/pmc_tmfma2/mfma_counter_collection.csv $O/r04_pmc_mfma_busy_train.json $C > $O/pmc_tmfma_report2.log 2>&1

python3 bench.py --hw 192 --ddim-steps 10 --cpu-config1 > $O/bench_config1_cpu_phases.log 2>&1
echo done
