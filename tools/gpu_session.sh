#!/bin/bash
# One GPU session on a gpurun box: tools/gpu_session.sh <round> <tag> <action> [<action> ...]
#   (run as: gpurun --timeout S -- 'bash tools/gpu_session.sh r06 a "tests tests/test_gpu_ops.py -k aggregate" "prof step_b16384 tools/step_profile.py 16384 8 2 epoch"')
# Every action is ONE quoted string; outputs go to gpurun_out/<round><tag>/ (merged back by gpurun; what is kept moves to profiles/<round>/).
#   tests <pytest args>          python -m pytest <args> -m gpu -q            -> tests_<k>.txt
#   smoke                        __graft_entry__.smoke()                      -> smoke.txt
#   bench <bench.py args>        python bench.py <args>                       -> bench_<k>.json / .err
#   prof <name> <script + args>  rocprofv3 --kernel-trace --stats of the command -> <name>_kernel_stats.csv (+ <name>_trace.csv.gz with TRACE=1)
#   pmc <name> <counter> <cmd>   one rocprofv3 --pmc pass (no trace flags)    -> <name>_pmc_<counter>.csv.gz
#   run <name> <command line>    any command, env assignments allowed in front -> <name>.txt
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
ROUND=$1; TAG=$2; shift 2
O=$R/gpurun_out/$ROUND$TAG
mkdir -p $O
cd $R
k=0
for action in "$@"; do
  k=$((k + 1))
  verb=${action%% *}; rest=${action#* }
  [ "$verb" = "$action" ] && rest=""
  echo "== [$k] $action"
  case $verb in
    tests)
      eval "timeout ${T:-1500} python -X faulthandler -m pytest $rest -m gpu -q" > $O/tests_$k.txt 2>&1; tail -n 4 $O/tests_$k.txt ;;  # (eval: -k \"a or b\" keeps its quotes)
    smoke)
      python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -n 2 $O/smoke.txt ;;
    bench)
      timeout ${T:-900} python bench.py $rest > $O/bench_$k.json 2> $O/bench_$k.err; echo "bench rc=$?"; head -c 600 $O/bench_$k.json; echo ;;
    prof)
      name=${rest%% *}; cmd=${rest#* }
      (cd /tmp && export TMPDIR=/tmp && timeout ${T:-900} rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o $name -- python $R/$cmd > $O/$name.log 2>&1)
      cp $(find $O/prof_$name -name "*kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv
      if [ -n "$TRACE" ]; then cp $(find $O/prof_$name -name "*kernel_trace.csv" | head -1) $O/${name}_trace.csv; gzip -f $O/${name}_trace.csv; fi
      rm -rf $O/prof_$name
      python tools/kstats.py $O/${name}_kernel_stats.csv ${TOP:-12}
      tail -n 1 $O/$name.log | cut -c1-300 ;;
    pmc)
      name=${rest%% *}; r2=${rest#* }; ctr=${r2%% *}; cmd=${r2#* }
      (cd /tmp && export TMPDIR=/tmp && timeout ${T:-900} rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/pmc_$name -o $name -- python $R/$cmd > $O/${name}_pmc_$ctr.log 2>&1)
      cp $(find $O/pmc_$name -name "*counter_collection.csv" | head -1) $O/${name}_pmc_$ctr.csv; gzip -f $O/${name}_pmc_$ctr.csv
      rm -rf $O/pmc_$name ;;
    run)
      name=${rest%% *}; cmd=${rest#* }
      bash -c "timeout ${T:-900} env $cmd" > $O/$name.txt 2>&1; tail -n ${TAIL:-12} $O/$name.txt ;;
    *) echo "unknown action: $verb" ;;
  esac
done
