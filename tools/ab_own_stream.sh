#!/bin/sh
# One lane on the default stream (rounds enqueued launch by launch) against one lane on a stream of its own (rounds
# replayed from a HIP graph): sh tools/ab_own_stream.sh   (on the GPU box, from the repo root)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
for g in 64 256; do for v in 0 1; do
  AZ_LANE_STREAM=$v timeout -k 10 200 python3 $R/tools/move_time.py $g 192 12 256 48 1 graph 2>/dev/null | tail -1 | sed "s/^/own stream $v: /" || exit 1
done; done
for v in 0 1; do
  AZ_LANE_STREAM=$v timeout -k 10 200 python3 $R/tools/move_time.py 512 64 4 192 48 1 graph 2>/dev/null | tail -1 | sed "s/^/own stream $v: /" || exit 1
done
