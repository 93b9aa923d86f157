#!/bin/bash
# usage: bash tools/build_old_lib.sh <commit>   -- builds that commit's library as jrl-walkgen_amd/lib/libwg_mpc_old.so (in a scratch
# copy under /tmp) for same-box A/B runs: bash tools/ab.sh old default   (runs in the development container: it needs the git history)
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=/tmp/wg_old_build
rm -rf "$W"; mkdir -p "$W"
git -C "$ROOT" --work-tree="$W" checkout "$1" -- jrl-walkgen_amd/csrc jrl-walkgen_amd/Makefile include
git -C "$ROOT" reset -q
make -s -C "$W/jrl-walkgen_amd" lib/libwg_mpc.so
cp "$W/jrl-walkgen_amd/lib/libwg_mpc.so" "$ROOT/jrl-walkgen_amd/lib/libwg_mpc_old.so"
echo "built $1 -> lib/libwg_mpc_old.so"
