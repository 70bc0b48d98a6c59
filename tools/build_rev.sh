#!/bin/bash
# Builds the kernel library of a PAST revision next to the current one, for same-box A/B timing through ASR_HIP_LIB
# (box-to-box spread is ~2 %, more than most single changes):
#   bash tools/build_rev.sh <git-rev>      ->  build_ab/<rev>/asr_chinese_e2e_amd/libasr_hip.so   (same ABI version required)
#   bash tools/rt_env_sweep.sh ASR_HIP_LIB=$PWD/build_ab/<rev>/asr_chinese_e2e_amd/libasr_hip.so
set -e
rev=$1
root=$(cd "$(dirname "$0")/.." && pwd)
dst=$root/build_ab/$rev
rm -rf "$dst"; mkdir -p "$dst"
git -C "$root" archive "$rev" asr_chinese_e2e_amd/csrc include | tar -x -C "$dst"
make -C "$dst/asr_chinese_e2e_amd/csrc" -j8 ../libasr_hip.so > "$dst/build.log" 2>&1 || { tail -20 "$dst/build.log"; exit 1; }
rm -rf "$dst/asr_chinese_e2e_amd/csrc/build"
ls -la "$dst/asr_chinese_e2e_amd/libasr_hip.so"
