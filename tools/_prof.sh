cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python bench.py > /tmp/bench.log 2>&1; grep -v amdgpu.ids /tmp/bench.log | tail -1 > gpurun_out/r01_bench_line.json
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python bench.py --steps 1 --warmup 1 > /tmp/prof.log 2>&1
cp $(find /tmp/prof -name "*kernel_stats.csv" | head -1) gpurun_out/r01_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc/f -- python bench.py --steps 1 --warmup 0 > /tmp/pmcf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc/w -- python bench.py --steps 1 --warmup 0 > /tmp/pmcw.log 2>&1
python tools/pmc_traffic.py /tmp/pmc > gpurun_out/r01_pmc_traffic.json
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python bench.py --steps 1 --warmup 0 > /tmp/kt.log 2>&1
python tools/gap_analysis.py /tmp/kt > gpurun_out/r01_gap_analysis.json
head -c 400 gpurun_out/r01_bench_line.json; echo; head -5 gpurun_out/r01_kernel_stats.csv | cut -c1-150; head -c 600 gpurun_out/r01_pmc_traffic.json
