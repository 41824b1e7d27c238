#!/bin/sh
# Copy the summaries of a tools/profile_round.sh run (gpurun_out/<dir>, scratch) into profiles/ (tracked).
#   sh tools/collect_profiles.sh r03_prof r03
set -e
S=gpurun_out/$1; P=profiles/$2
if [ -s $S/bench_final.json ]; then cp $S/bench_final.json ${P}_bench.json; else cp $S/bench.json ${P}_bench.json; fi
cp $S/stats/bench_kernel_stats.csv ${P}_bench_kernel_stats.csv
cp $S/pmc_mfma.json ${P}_pmc_mfma.json
cp $S/pmc_traffic.json ${P}_pmc_traffic.json
for w in fmt fmtb dec; do
  python3 tools/summarize_pmc.py $S/pmc_FETCH_SIZE_$w > ${P}_pmc_fetch_size_$w.csv
  python3 tools/summarize_pmc.py $S/pmc_WRITE_SIZE_$w > ${P}_pmc_write_size_$w.csv
  python3 tools/summarize_pmc.py $S/pmc_sq_$w > ${P}_pmc_sq_$w.csv
done
ls -la profiles | grep $2
