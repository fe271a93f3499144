#!/bin/bash
# build the CURRENT tree's libsonde_mi355.so as a named variant for interleaved A/B runs: tools/mkvariant.sh <name> [EXTRA flags]
# -> tools/ab_libs/lib_<name>.so (git-ignored; travels to the GPU box); then tools/ab_wb.sh / ab_repeat.sh with SONDE_MI355_LIB
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d /tmp/variant_XXXX)
mkdir -p $T/a/b $R/tools/ab_libs
cp -r $R/sdrpp_radiosonde_amd/csrc $T/a/b/csrc && cp -r $R/include $T/a/include
rm -rf $T/a/b/csrc/_obj
make -s -C $T/a/b/csrc EXTRA="$*" OUT=$R/tools/ab_libs/lib_$name.so 2>&1 | grep -E "error|warning: unused" || true
rm -rf $T
ls -la $R/tools/ab_libs/lib_$name.so
