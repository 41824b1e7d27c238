set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06_prof}; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --no-bf16 > $O/stats.log 2>&1
find $O/stats -name "*kernel_trace.csv" -delete
for w in fmt fmtb dec; do for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${c}_$w -o p -- python3 tools/profile_hotpath.py --what $w > $O/pmc_${c}_$w.log 2>&1
  find $O/pmc_${c}_$w -name "*kernel_trace.csv" -delete
done; done
for w in fmt fmtb dec; do
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq_$w -o p -- python3 tools/profile_hotpath.py --what $w > $O/pmc_sq_$w.log 2>&1
  find $O/pmc_sq_$w -name "*kernel_trace.csv" -delete
done
python tools/make_mfma_json.py $O/pmc_sq_fmt,$O/pmc_sq_fmtb,$O/pmc_sq_dec $O/pmc_mfma.json
python tools/make_traffic_json.py $O/pmc_FETCH_SIZE_fmt,$O/pmc_FETCH_SIZE_fmtb,$O/pmc_FETCH_SIZE_dec $O/pmc_WRITE_SIZE_fmt,$O/pmc_WRITE_SIZE_fmtb,$O/pmc_WRITE_SIZE_dec $O/pmc_traffic.json
# the bench line that goes to profiles/: run with THIS round's counter files in place, so its counter-derived fields are live
P=profiles/${2:-r06}
cp $O/pmc_mfma.json ${P}_pmc_mfma.json; cp $O/pmc_traffic.json ${P}_pmc_traffic.json
python bench.py > $O/bench_final.json 2> $O/bench_final.err; cut -c1-300 $O/bench_final.json
du -sh $O
