#!/bin/bash
# policy-kernel time (graph-replayed launches, HIP events) per config for one or more library builds:
#   bash tools/polbench.sh "pp_map10,pp_map30" lib1.so lib2.so ...
CFGS=${1:-pp_map10}; shift
for lib in "$@"; do
  for cfg in ${CFGS//,/ }; do
    echo -n "$(basename $lib) $cfg  "
    COMMARL_LIB=$lib COMMARL_FWD_STOP=0 python tools/fwdphases.py child $cfg 2>&1 | grep "stop="
  done
done
