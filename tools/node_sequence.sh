#!/bin/bash
# fr_node on ONE card: the C-ABI host's frames-in-flight rate next to bench.py --pipelined (the yardstick) and the
# one-frame-at-a-time default; 1 part and 4 parts on device 0.  Writes gpurun_out/r04_node_sequence.txt.
set -o pipefail
out=gpurun_out/r04_node_sequence.txt
mkdir -p gpurun_out
: > $out
for wl in c2 c3; do
  echo "== $wl: default command legs (one frame at a time; --pipelined = two contexts alternating) ==" >> $out
  python bench.py --workload $wl --steps 60 --warmup 10 --pipelined --no-cpu-baseline 2>>$out | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])   # (RCCL prints a banner on stdout)
print('one at a time  ms/frame', d['ms_per_step'], 'Mpx/s', d['value'])
print('pipelined      ms/frame', d['pipelined']['ms_per_step'], 'Mpx/s', d['pipelined']['value'])
" >> $out || exit 1
  for parts in 1 4; do
    echo "== $wl: bench.py --host node --gpus 1 --node-parts $parts ==" >> $out
    python bench.py --workload $wl --steps 60 --warmup 10 --host node --gpus 1 --node-parts $parts 2>>$out | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])   # (RCCL prints a banner on stdout)
print('headline', d['config']['mode'], d['ms_per_step'], 'ms/frame', d['value'], 'Mpx/s  verified', d['exchange_verified'])
for r in d['node']['runs']:
    print('  ', {k: r.get(k) for k in ('mode','gather','gather_asked','parts','slots','lanes','ms_per_step','value','exchange_verified','error','note')})
" >> $out || exit 1
  done
done
for sl in "1 1" "2 1" "2 2" "4 2" "4 4" "8 4"; do
  set -- $sl
  echo "== c2: --host node --node-parts 4 --node-slots $1 --node-lanes $2 ==" >> $out
  python bench.py --workload c2 --steps 60 --warmup 10 --host node --gpus 1 --node-parts 4 --node-slots $1 --node-lanes $2 2>>$out | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])   # (RCCL prints a banner on stdout)
for r in d['node']['runs']:
    if r.get('mode') == 'sequence': print('  ', r.get('slots'), r.get('lanes'), r.get('ms_per_step'), 'ms/frame', r.get('value'), 'Mpx/s', r.get('exchange_verified'), r.get('error'))
" >> $out || exit 1
done
cat $out
