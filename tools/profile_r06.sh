#!/bin/bash
# Round-6 measurement record for profiles/ (run on the GPU box from the repository root: bash tools/profile_r06.sh):
#   * rocprofv3 --kernel-trace --stats of the eager bench step of C2 (default), C3, C4 (64 draws), C5 (128 chains; c5b: 2 of
#     them at a conditioning score of 1e6 -- the robust route of the GP):
#     per-kernel average durations;
#   * PMC passes -- each counter set in its OWN run, with --kernel-trace only, as gpurun requires:
#     FETCH_SIZE, WRITE_SIZE (HBM traffic; FETCH x 2 on gfx950, MI355X_MICROARCH.md) and SQ_INSTS_VALU + SQ_WAVES,
#     SQ_ACTIVE_INST_VALU + SQ_BUSY_CYCLES + SQ_WAVE_CYCLES (the fp64-VALU view) for every kernel of those steps, and
#     for the sparse / chi2 legs of the C2 sweep and the standalone Ops kepler / quad_solution_vector at n = 1.5e8
#     (tools/run_leg.py);
#   -> gpurun_out/r06/*, summarised by tools/make_profile_r06.py into profiles/r06_counters.json,
#      profiles/r06_bench_rocprof_summary.txt (which bench.py quotes when the kernel sources are unchanged).
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r06
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="--steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-stats --no-graph"
declare -A CMD
CMD[c2]="python $R/bench.py $B"
CMD[c3]="python $R/bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-stats --no-graph"
CMD[c4]="python $R/bench.py --config c4 --global-draws 64 $B"
CMD[c5]="python $R/bench.py --config c5 --global-draws 128 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-stats --no-graph"
CMD[c5b]="python $R/bench.py --config c5 --global-draws 128 --c5-bright 2 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-stats --no-graph"
CMD[j8]="python $R/tools/wide_step.py sho4 128 3"          # the C5 shape with a J = 8 kernel (four SHO terms): lane groups of eight
CMD[j10]="python $R/tools/wide_step.py rot2_sho 128 3"     # ... J = 10 (two RotationTerms + SHO): a DPP row of sixteen, wide trees
CMD[sparse]="python $R/tools/run_leg.py sparse 1024 10"
CMD[chi2]="python $R/tools/run_leg.py chi2 1024 10"
CMD[kepler]="python $R/tools/run_leg.py kepler 150000000 6"
CMD[quadsv]="python $R/tools/run_leg.py quadsv 150000000 4"
for leg in c2 c3 c4 c5 c5b j8 j10 sparse chi2 kepler quadsv; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/${leg}_trace -o p -- ${CMD[$leg]} > $out/${leg}_trace.log 2>&1
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    if [ "$leg" = c4 ] && [ "$c" != "FETCH_SIZE" ] && [ "$c" != "WRITE_SIZE" ]; then continue; fi
    if [ "$leg" = c5b ] && [ "$c" != "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU" ]; then continue; fi   # (the robust route's kernels: durations + instructions)
    if { [ "$leg" = j8 ] || [ "$leg" = j10 ]; } && [ "$c" = "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" ]; then continue; fi
    d=$out/${leg}_pmc_$(echo $c | tr ' ' '_' | cut -c1-30)
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o p -- ${CMD[$leg]} > /dev/null 2>&1
  done
done
python $R/bench.py --steps 5 --no-extras --no-cpu-baseline --no-stats > $out/bench_pre.json 2> /dev/null
python $R/tools/make_profile_r06.py $out $R/profiles
# gpurun merges at most 64 MiB of gpurun_out back: keep the summaries (and the kernel-stats tables), drop the raw traces
mkdir -p $out/profiles && cp $R/profiles/r06_* $out/profiles/
for leg in c2 c3 c4 c5 c5b j8 j10 sparse chi2 kepler quadsv; do cp $(find $out/${leg}_trace -name "*kernel_stats.csv" | head -1) $out/profiles/r06_${leg}_kernel_stats.csv 2>/dev/null; done
rm -rf $out/*_trace $out/*_pmc_*
# the un-profiled bench line last, with this record's counters in place (roofline.traffic / .valu filled in from it)
python $R/bench.py > $out/bench.json 2> $out/bench.err
cp $R/gpurun_out/bench_full.json $out/profiles/r06_bench_full.json 2>/dev/null
tail -1 $out/bench.json > $out/profiles/r06_bench.json
python $R/bench.py --config c4 --global-draws 64 --steps 50 --no-cpu-baseline > $out/bench_c4_64.json 2> /dev/null
python $R/bench.py --config c5 --global-draws 128 --steps 20 --no-cpu-baseline > $out/bench_c5_128.json 2> /dev/null
python $R/bench.py --config c3 --steps 20 --no-cpu-baseline > $out/bench_c3.json 2> /dev/null
