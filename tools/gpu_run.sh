#!/bin/bash
# One entry point for everything run on the GPU box:  gpurun -- 'bash tools/gpu_run.sh <recipe> [args]'
# (rounds 1-5 kept one gpu_r*.sh per call — 60 files; they are in the git history).  Output: gpurun_out/<recipe>/.
#   driver [tag]        smoke() + the driver's own command line (bench.py --gpus 1 --steps 20 --warmup 5): line + full record kept
#   profile <tag>       rocprofv3 passes over the TIMED STEP ONLY (--no-variants: one instantiation per kernel row) -> gpurun_out/prof_<tag>
#   profile-variant <tag> <variant>   the same over one variant's step (--variants <variant>, timed step's own passes excluded by launch order)
#   suite               the whole GPU test suite
#   tests <pytest args> a subset
#   fuzz [seconds]      the three differential fuzzers
#   bench <args>        bench.py with the given arguments (line + full record kept under gpurun_out/bench/)
#   py <script> [args]  any python script (microbenchmarks under tools/microbench/)
#   csv [rows]          CSV / materialisation tests + tools/microbench/csv_ingest.py and pipeline.py (default 5e7 rows)
export TMPDIR=/tmp
R=${1:-driver}; shift
OUT=gpurun_out/$R
mkdir -p $OUT
case $R in
driver)
  TAG=${1:-r06}
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
  T0=$(date +%s)
  timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --extras $OUT/${TAG}_bench_full.json > $OUT/bench.out 2> $OUT/bench.err
  echo "bench rc=$? wall $(( $(date +%s) - T0 )) s"; tail -3 $OUT/bench.err
  tail -1 $OUT/bench.out > $OUT/${TAG}_bench_line.json
  wc -c $OUT/${TAG}_bench_line.json
  python -c "import json,sys; d=json.load(open('$OUT/${TAG}_bench_line.json')); print(json.dumps(d, indent=1)[:6000])"
  ;;
profile)
  TAG=${1:-r06}; shift
  timeout 1200 bash tools/gpu_profile.sh $TAG --no-variants "$@" 2>&1 | tail -40
  ;;
profile-variant)
  TAG=${1:-r06}; V=$2
  timeout 1200 bash tools/gpu_profile.sh ${TAG}_$V --variants $V 2>&1 | tail -40
  ;;
suite)
  timeout 2400 python -m pytest tests -m gpu -q "$@" > $OUT/pytest_full.txt 2>&1
  grep -n "Fatal Python\|^tests/.*::\|File \"/root\|File \".*/tests/\|File \".*/csvplus_amd/" $OUT/pytest_full.txt | head -20
  tail -25 $OUT/pytest_full.txt | tee $OUT/pytest.txt
  ;;
tests)
  timeout 2400 python -m pytest -m gpu -q "$@" 2>&1 | tail -40 | tee $OUT/pytest.txt
  ;;
fuzz)
  S=${1:-60}
  timeout $((S*3+100)) python tools/fuzz_round6.py $S 604 2>&1 | grep -v amdgpu.ids | tee $OUT/fuzz_round6.txt | tail -4
  timeout $((S*3+100)) python tools/fuzz_round5.py $S 601 2>&1 | grep -v amdgpu.ids | tee $OUT/fuzz_round5.txt | tail -4
  timeout $((S*3+100)) python tools/fuzz_gpu.py $S 602 2>&1 | grep -v amdgpu.ids | tee $OUT/fuzz_gpu.txt | tail -3
  timeout $((S*3+100)) python tools/fuzz_builds.py $S 603 2>&1 | grep -v amdgpu.ids | tee $OUT/fuzz_builds.txt | tail -3
  ;;
bench)
  timeout 1500 python3 bench.py --extras $OUT/bench_full.json "$@" > $OUT/bench.out 2> $OUT/bench.err
  echo "bench rc=$?"; tail -3 $OUT/bench.err; tail -1 $OUT/bench.out | tee $OUT/bench_line.json | head -c 7000; echo
  ;;
py)
  timeout 1500 python "$@" 2>&1 | grep -v amdgpu.ids | tee $OUT/$(basename $1 .py).txt | tail -60
  ;;
csv)
  ROWS=${1:-5e7}
  timeout 900 python -m pytest tests/test_csv_ingest.py tests/test_materialize.py -m gpu -x -q 2>&1 | tail -3
  timeout 600 python tools/microbench/csv_ingest.py $ROWS 2>&1 | grep -v amdgpu.ids | tee $OUT/csv_ingest.txt | tail -12
  timeout 600 python tools/microbench/pipeline.py $ROWS 2>&1 | grep -v amdgpu.ids | tee $OUT/pipeline.txt | tail -24
  ;;
*)
  echo "unknown recipe $R"; exit 2;;
esac
true
