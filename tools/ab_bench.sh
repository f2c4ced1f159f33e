#!/bin/bash
# back-to-back frames (bench.py) with two builds, interleaved
for i in 1 2 3; do for L in build/ab/lib_head.so build/ab/lib_new.so; do
  for w in c2 c3; do printf "%s %s " $w $(basename $L); FR_LIB_PATH=$PWD/$L python bench.py --steps 60 --no-cpu-baseline --workload $w 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; done; done; done
FR_LIB_PATH=$PWD/build/ab/lib_head.so python tools/interactive_time.py 2>/dev/null | grep -E "1280x720|1920x1080 max_iter  256" ; FR_LIB_PATH=$PWD/build/ab/lib_new.so python tools/interactive_time.py 2>/dev/null | grep -E "1280x720|1920x1080 max_iter  256"
