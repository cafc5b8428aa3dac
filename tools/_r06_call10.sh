R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(TCC|TCP|TA|TD)_[A-Za-z0-9_]+" | sort -u | tr '\n' ' ' | fold -w 220 > $OUT/r06_counters_avail.txt; wc -c $OUT/r06_counters_avail.txt
export FRIDO_TUNE_CACHE=$R/profiles/tune_cache.json FRIDO_TUNE_CACHE_READONLY=1 FRIDO_TUNE_ON_MISS=tune
pass() { n=$1; shift; timeout 400 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_$n -- python $R/tools/profile_forward.py --precision bf16x3 > /tmp/pmc_$n.log 2>&1; tail -2 /tmp/pmc_$n.log; }
pass a TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pass b TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr
pass c GRBM_GUI_ACTIVE TCC_TAG_STALL_sum TCC_EA0_RDREQ_32B_sum TCC_BUSY_avr
python $R/tools/pmc_instances.py /tmp/pmc_a /tmp/pmc_b /tmp/pmc_c --top 45 > $OUT/r06_pmc_instances.json 2> $OUT/r06_pmc_instances.err; head -c 3000 $OUT/r06_pmc_instances.json; tail -3 $OUT/r06_pmc_instances.err
