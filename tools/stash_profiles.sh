#!/bin/bash
# Development tool: copies what tools/collect_profiles.sh left under gpurun_out/<tag>/ into profiles/<round>/<prefix>_* (the
# summaries that are kept in the repository) and installs its latest_traffic.json.   usage: tools/stash_profiles.sh <tag> <round> <prefix>
set -e
tag=$1; round=$2; pre=$3
src=gpurun_out/$tag; dst=profiles/$round
mkdir -p $dst
for c in c2 c3_1gpu c4 c5; do [ -f $src/bench_$c.json ] && cp $src/bench_$c.json $dst/${pre}_bench_$c.json; done
for c in c2 c4 c5; do f=$(find $src/stats_$c -name "*kernel_stats.csv" -printf '%T@ %p\n' | sort -rn | head -1 | cut -d' ' -f2-); [ -n "$f" ] && cp $f $dst/${pre}_kernel_stats_$c.csv; done
for d in pmc_fetch pmc_write pmc_fetch_C2 pmc_write_C2 pmc_fetch_C4 pmc_write_C4 pmc_fetch_C5 pmc_write_C5; do
  f=$(find $src/$d -name "*counter_collection.csv" -printf '%T@ %p\n' | sort -rn | head -1 | cut -d' ' -f2-); [ -n "$f" ] && cp $f $dst/${pre}_${d}_counter_collection.csv
done
cp $src/pmc_traffic.json $dst/${pre}_pmc_traffic.json
cp $src/pmc_traffic_all.json $dst/${pre}_pmc_traffic_all.json
for f in pmc_sq pmc_sq_c4 pmc_sq_c5; do [ -f $src/$f.txt ] && cp $src/$f.txt $dst/${pre}_$f.txt; done
for f in bench_c3_single_process_rccl.json bench_c3_single_process_peer2.json bench_readme.json reserve_cus.txt stream_rate.txt latency.txt dfa.txt wide_alphabets.txt longest_shapes.txt shapes.txt readme_shapes.txt map_flavours.txt; do [ -f $src/$f ] && grep -v amdgpu.ids $src/$f > $dst/${pre}_$f; done
cp $src/latest_traffic.json profiles/latest_traffic.json
ls $dst | grep "^${pre}_" | wc -l
