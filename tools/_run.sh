rm -rf gpurun_out/prof_* 
bash tools/run_profiles.sh r05b > gpurun_out/r05b_profiles.log 2>&1
python bench.py > gpurun_out/r05b_full_bench.json 2> gpurun_out/r05b_full_bench.err
python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/r05_t11.log
python tools/fast_detect_report.py 1048576 nb nb63 rach ext mixed > gpurun_out/r05_fast_report.txt 2>&1
cat gpurun_out/r05_t11.log; tail -3 gpurun_out/r05b_profiles.log; grep -v amdgpu gpurun_out/r05_fast_report.txt | cut -c1-220
