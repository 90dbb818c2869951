#!/bin/bash
# Round-3 controls for the free-running Dice comparison, CPU only, low priority, one thread per run:
#   standin  = the product's host logic on tests/cpu_backend.py (torch-CPU arithmetic under the product's Python)
#   oracle2  = a second realisation of the oracle (ORACLE_THREADS=1: another summation order than the round-2 runs)
# usage: tools/dice_cpu_queue.sh <workers> <out-dir>      (jobs alternate standin / oracle2 so that a partial run is still balanced)
W=${1:-5}; OUT=${2:-gpurun_out/dice3}; mkdir -p "$OUT"
jobs=()
for s in $(seq 0 24); do
  jobs+=("standin $s")
  if [ "$s" -lt 12 ]; then jobs+=("oracle2 $s"); fi
done
printf '%s\n' "${jobs[@]}" | xargs -P "$W" -L 1 bash -c '
  kind=$0; seed=$1; out='"$OUT"'/r03_dice_seed${seed}_${kind}_cpu.log
  [ -s "$out" ] && grep -q RESULT "$out" && exit 0
  side=$kind; [ "$kind" = oracle2 ] && side=oracle
  ORACLE_THREADS=1 OMP_NUM_THREADS=1 MKL_NUM_THREADS=1 DICE_LABEL=$kind nice -n 19 python tools/dice_seeds.py $seed $side > "$out" 2>&1
'
