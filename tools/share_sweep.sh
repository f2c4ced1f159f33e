#!/bin/bash
# round 4: what a 1/8 share of C2 costs (8 shares one after the other) against queue / region geometry of the lane pool
P="--opt periodicity=-1 --only 8"
run() { printf "%-70s " "$*"; python tools/shard_cost.py $P "$@" 2>/dev/null | grep "plane=rgba" | sed 's/2 streams.*//'; }
run
run --opt pool_items_per_wg=16
run --opt pool_items_per_wg=16 --opt stream_probes=8
run --opt regions=8
run --opt regions=8 --opt stream_probes=8
run --opt regions=8 --opt pool_items_per_wg=16
run --opt regions=8 --opt pool_items_per_wg=16 --opt stream_probes=8
run --opt shards=8
run --opt shards=8 --opt pool_items_per_wg=16 --opt stream_probes=8
run --opt pool_refill_at=8
run --opt pool_refill_at=4 --opt pool_items_per_wg=16
run --opt pool_refill_at=8 --opt pool_items_per_wg=16 --opt regions=8 --opt stream_probes=8
run --opt stream_rotate=1
