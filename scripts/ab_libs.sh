#!/bin/bash
# Alternating A/B of two builds of libmnrf_hip.so on ONE box:  scripts/ab_libs.sh <lib_a|default> <lib_b|default> [rounds] -- <command...>
# (MNRF_LIB selects the library; "default" = the in-tree one).  The command must print one JSON line with "ms_per_step".
A=$1; B=$2; ROUNDS=${3:-3}; shift 3; [ "$1" == "--" ] && shift
for r in $(seq $ROUNDS); do
  for L in "$A" "$B"; do
    if [ "$L" == "default" ]; then unset MNRF_LIB; else export MNRF_LIB=$L; fi
    "$@" 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$L', round(d['ms_per_step'],3))"
  done
done
