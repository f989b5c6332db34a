#!/bin/bash
# Runs on the GPU box (through gpurun): the bench lines and rocprofv3 evidence that profiles/<round>/ keeps.
# usage: [PART=A|B] bash tools/collect_profiles.sh <out-dir under gpurun_out/> <round dir under profiles/> [<file prefix in it>]
# (the last two only name, inside latest_traffic.json, the file tools/stash_profiles.sh will copy the traffic summary to)
# PART: a gpurun call lasts 20 minutes at most -- A = bench lines, kernel statistics, counters; B = the tools' tables and the
# bench lines once more with the traffic attached (needs A's latest_traffic.json, which travels in gpurun_out/ only: B copies
# profiles/latest_traffic.json from where the caller has put A's); unset: everything in one call.
set -u
OUT=gpurun_out/${1:-final}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
PART=${PART:-all}
if [ "$PART" != B ]; then
python3 bench.py --steps 20 --warmup 3 > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.err"
python3 bench.py --config C4 --steps 10 --warmup 2 > "$OUT/bench_c4.json" 2> "$OUT/bench_c4.err"
python3 bench.py --config C5 --steps 20 --warmup 3 > "$OUT/bench_c5.json" 2> "$OUT/bench_c5.err"
# per-kernel time: the program itself after "--"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_c2" -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-separate-launches > "$OUT/stats_c2.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_c4" -- python3 bench.py --config C4 --steps 3 --warmup 1 --no-cpu-baseline --no-separate-launches > "$OUT/stats_c4.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_c5" -- python3 bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline --no-separate-launches > "$OUT/stats_c5.log" 2>&1
# HBM traffic counters: separate passes, counters only (no other trace domains).  The no-verify / stream-only variants are
# ablation switches, which only the -DACGPU_ABLATION build has (tools/build_variant.sh abl -DACGPU_ABLATION, built before the
# gpurun call: ahocorasick_amd/lib_abl/ travels with the snapshot); its full build is the product kernel plus the switches.
export ACGPU_LIB=ahocorasick_amd/lib_abl/libacgpu.so
[ -f "$ACGPU_LIB" ] || { echo "missing $ACGPU_LIB: run tools/build_variant.sh abl -DACGPU_ABLATION first (the no-verify / stream-only variants would silently measure the product kernel)"; exit 3; }
V='{"tile":{"force_kernel":2},"noverify":{"force_kernel":2,"tile_debug":1},"stream":{"force_kernel":2,"tile_debug":5}}'
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 tools/kbench.py --rounds 1 --variants "$V" > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 tools/kbench.py --rounds 1 --variants "$V" > "$OUT/pmc_write.log" 2>&1
python3 tools/pmc_traffic.py "$OUT" "$OUT/pmc_traffic.json"
# SQ instruction / busy counters of the same three builds
bash tools/pmc_sq.sh "${1:-final}/pmc_sq" "$V" > "$OUT/pmc_sq.txt" 2>&1
unset ACGPU_LIB
# config 3's share on one GPU (Set records, R = 8), and the sibling kernels' counters (SQ + HBM traffic)
python3 bench.py --config C3 --steps 20 --warmup 3 > "$OUT/bench_c3_1gpu.json" 2> "$OUT/bench_c3_1gpu.err"
bash tools/pmc_sq.sh "${1:-final}/pmc_sq_c4" '{"c4":{}}' "--config C4 --set" k_longest > "$OUT/pmc_sq_c4.txt" 2>&1
bash tools/pmc_sq.sh "${1:-final}/pmc_sq_c5" '{"c5":{}}' "--config C5" k_ww_ > "$OUT/pmc_sq_c5.txt" 2>&1
for c in C2 C4 C5; do
  extra=""; [ $c = C4 ] && extra="--set"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_$c" -- python3 tools/kbench.py --rounds 1 --config $c $extra --variants '{"k":{}}' > "$OUT/pmc_fetch_$c.log" 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_$c" -- python3 tools/kbench.py --rounds 1 --config $c $extra --variants '{"k":{}}' > "$OUT/pmc_write_$c.log" 2>&1
done
python3 tools/pmc_traffic_all.py "$OUT" --c2-calibrated "$OUT/pmc_traffic.json" --latest "$OUT/latest_traffic.json" --tag "profiles/${2:-rNN}/${3:-${1:-final}}_pmc_traffic_all.json" > "$OUT/pmc_traffic_all.json" 2> "$OUT/pmc_traffic_all.err"
fi
if [ "$PART" = A ]; then cat "$OUT/bench_c2.json"; exit 0; fi
# round 4: the multi-device entries through ONE host process (a one-GPU box: RCCL over a world of one; peer copies over the
# device named twice), the reference's own published workload, the CU reserve, the chunked entry
python3 bench.py --gpus 1 --single-process --config C3 --steps 20 --warmup 3 > "$OUT/bench_c3_single_process_rccl.json" 2> "$OUT/bench_c3_sp.err"
python3 bench.py --gpus 2 --devices 0,0 --single-process --units-log2 28 --steps 10 --warmup 2 > "$OUT/bench_c3_single_process_peer2.json" 2>> "$OUT/bench_c3_sp.err"
python3 bench.py --config README --steps 5 --warmup 1 > "$OUT/bench_readme.json" 2> "$OUT/bench_readme.err"
python3 tools/kbench.py --rounds 9 --variants '{"reserve_cus=0":{},"reserve_cus=4":{"reserve_cus":4},"reserve_cus=8":{"reserve_cus":8},"reserve_cus=16":{"reserve_cus":16},"reserve_cus=32":{"reserve_cus":32}}' > "$OUT/reserve_cus.txt" 2>&1
python3 tools/stream_rate.py > "$OUT/stream_rate.txt" 2>&1
python3 tools/latency.py > "$OUT/latency.txt" 2>&1
# round 4, later: the DFA chunk scan (k_ac_dfa against the one-chain kernel of rounds 1-3), wide alphabets, other dictionary shapes
python3 tools/kbench.py --rounds 5 --variants '{"tile":{},"dfa":{"force_kernel":1},"dfa_one_chain_r3":{"force_kernel":1,"tile_debug":8796093022208,"lds_table_bytes":98304}}' > "$OUT/dfa.txt" 2>&1
python3 tools/wide_alphabets.py > "$OUT/wide_alphabets.txt" 2>&1
python3 tools/longest_shapes.py > "$OUT/longest_shapes.txt" 2>&1
python3 tools/shapes.py > "$OUT/shapes.txt" 2>&1
# round 5: the README word list through the tile kernel, the DFA chunk scan and k_ac_states; Set against Map records
python3 tools/readme_shapes.py 2>&1 | grep -v amdgpu.ids > "$OUT/readme_shapes.txt"
python3 tools/map_flavours.py 2>&1 | grep -v amdgpu.ids > "$OUT/map_flavours.txt"
# the three bench lines once more with this collection's traffic attached (bench.py attaches profiles/latest_traffic.json only
# when it was measured on the sources it runs; the copy of the repository on this box is scratch)
[ -f "$OUT/latest_traffic.json" ] && cp "$OUT/latest_traffic.json" profiles/latest_traffic.json
python3 bench.py --steps 20 --warmup 3 > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.err"
python3 bench.py --config C4 --steps 10 --warmup 2 > "$OUT/bench_c4.json" 2> "$OUT/bench_c4.err"
python3 bench.py --config C5 --steps 20 --warmup 3 > "$OUT/bench_c5.json" 2> "$OUT/bench_c5.err"
cat "$OUT/bench_c2.json"
