# Evidence collection of round 4 (run on the GPU box through gpurun): the bench command under rocprofv3 --kernel-trace --stats and the
# separate --pmc passes; the summaries under gpurun_out/r4 were copied into profiles/ by hand.
set -e
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
C=${CTSI_COMMIT:-unknown}   # git is not available on the GPU box: pass the short hash
# 1. the bench command under rocprofv3 --kernel-trace --stats (same command as the bench log next to it)
python3 bench.py --steps 20 --warmup 5 > $O/bench_ev.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python3 bench.py --steps 20 --warmup 5 --no-cpu --no-volume > $O/prof_bench.log 2>&1
# 2. PMC traffic (separate passes) + MFMA busy / LDS conflicts over the eager U-Net evaluations
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o fetch -- python3 tools/profile_ops.py --repeats 1 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o write -- python3 tools/profile_ops.py --repeats 1 > $O/pmc_write.log 2>&1
python3 tools/pmc_traffic.py $O/pmc_fetch/fetch_counter_collection.csv $O/pmc_write/write_counter_collection.csv $O/r04_pmc_traffic.json $C > $O/pmc_traffic.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -o mfma -- python3 tools/profile_ops.py --repeats 1 > $O/pmc_mfma.log 2>&1
python3 tools/pmc_mfma.py $O/pmc_mfma/mfma_counter_collection.csv $O/r04_pmc_mfma_busy.json $C > $O/pmc_mfma_report.log 2>&1
# 3. the VAE legs: kernel stats + traffic on real data
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_dec2 -o dec -- python3 tools/profile_ops.py --net dec --repeats 1 > $O/prof_dec2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_enc2 -o enc -- python3 tools/profile_ops.py --net enc --repeats 1 > $O/prof_enc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_dec -o fetch -- python3 tools/profile_ops.py --net dec --repeats 1 > $O/pmc_fetch_dec.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_dec -o write -- python3 tools/profile_ops.py --net dec --repeats 1 > $O/pmc_write_dec.log 2>&1
python3 tools/pmc_traffic.py $O/pmc_fetch_dec/fetch_counter_collection.csv $O/pmc_write_dec/write_counter_collection.csv $O/r04_pmc_traffic_vae_dec.json $C "python tools/profile_ops.py --net dec --repeats 1 (2 eager VAE decodes of a random latent (1,8,48,128,128) -> 512x512)" > $O/pmc_traffic_dec.log 2>&1
echo done
