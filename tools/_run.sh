rm -rf gpurun_out/prof_* 
bash tools/run_profiles.sh r05a > gpurun_out/r05a_profiles.log 2>&1
bash tools/run_profiles_aux.sh r05 > gpurun_out/r05_aux_profiles.log 2>&1
python bench.py > gpurun_out/r05a_full_bench.json 2> gpurun_out/r05a_full_bench.err
python -m pytest tests -q -m gpu 2>&1 | tail -12 > gpurun_out/r05_t10.log
python tools/parity_campaign.py > gpurun_out/r05_parity_campaign.txt 2>&1
cat gpurun_out/r05_t10.log; tail -3 gpurun_out/r05a_profiles.log; tail -12 gpurun_out/r05_parity_campaign.txt
