#!/bin/bash
# Runs a list of GPU steps on the gpurun box.  A step that fails normally lets the next one run;
# a step that is KILLED by its timeout (124/137) stops the whole call (no further GPU work after a hang).
# usage: tools/gpu_steps.sh "name|seconds|command" ...
mkdir -p gpurun_out
rc_all=0
for spec in "$@"; do
  name="${spec%%|*}"; rest="${spec#*|}"; secs="${rest%%|*}"; cmd="${rest#*|}"
  echo "=== step $name (limit ${secs}s): $cmd"
  start=$(date +%s)
  timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "=== step $name rc=$rc in $(( $(date +%s) - start ))s"
  tail -n 15 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "=== step $name was killed at its limit: stopping"; exit $rc; fi
  [ $rc -ne 0 ] && rc_all=$rc
done
exit $rc_all
