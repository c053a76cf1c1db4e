#!/bin/bash
# Same-call A/B timing of MFCC kernel variants (GPU box): every library in turn, three rounds, so clock drift hits all alike.
#   tools/ab_mfcc.sh tools/bin/libkws_a.so tools/bin/libkws_b.so ...      ("default" = the in-tree library)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for round in 1 2 3; do
  for lib in "$@"; do
    if [ "$lib" = default ]; then env -u KWS_HIP_LIB python tools/time_mfcc.py; else KWS_HIP_LIB=$PWD/$lib python tools/time_mfcc.py; fi
  done
done
