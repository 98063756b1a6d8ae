export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp
python3 $R/bench.py --steps 5 --warmup 2 2>/dev/null | grep '^{' > $O/bench_r01_v4.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_v4 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_v4.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc4_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fed4k > $O/pmc4_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc4_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fed4k > $O/pmc4_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/cal_fetch -- python3 $R/tools/pmc_calib.py > $O/cal_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/cal_write -- python3 $R/tools/pmc_calib.py > $O/cal_write.log 2>&1
cat $O/bench_r01_v4.json | cut -c1-1500
