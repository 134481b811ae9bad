#!/bin/sh
# Wall time per lockstep move over moves 6-126 of the games (deep trees included), small configurations:
# sh tools/ab_whole_game.sh   (on the GPU box, from the repo root)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
timeout -k 10 200 python3 $R/tools/move_time.py 64 192 12 256 48 1 2>/dev/null | tail -1 || exit 1
timeout -k 10 200 python3 $R/tools/move_time.py 256 192 12 256 48 1 2>/dev/null | tail -1 || exit 1
timeout -k 10 200 python3 $R/tools/move_time.py 512 64 4 192 48 1 2>/dev/null | tail -1 || exit 1
timeout -k 10 200 python3 $R/tools/move_time.py 512 64 4 192 48 2 graph 2>/dev/null | tail -1 || exit 1
