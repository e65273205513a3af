#!/bin/bash
# kernel stats of several builds of the library on one box: tools/ab_many.sh <grep pattern> <bench args...> -- lib1.so lib2.so ...
pat=$1; shift; args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
cd $GRAFT_REPO_ROOT
for l in "$@"; do
  cp $l crass_amd/libcrass_hip.so; echo "== $l"
  bash tools/stats_one.sh many_$(basename $l .so) "${args[@]}" 2>&1 | grep -E "$pat"
done
