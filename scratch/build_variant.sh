#!/bin/bash
# build a VARIANT of the library from a patched copy of the sources (experiments stay out of the product tree):
#   scratch/build_variant.sh <name> '<sed expression>' [file relative to csrc, default msm_impl.h] ...   → icicle-snark_amd/lib_<name>/libicicle_snark_hip.so
# A/B against the shipped library with scratch/ab_lib.sh (lib vs lib_b) or scratch/ab_many.sh.
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
v=$root/build/var_$name
rm -rf $v; mkdir -p $v/pkg
cp -r $root/icicle-snark_amd/csrc $v/pkg/csrc
ln -s $root/include $v/include
while [ $# -gt 0 ]; do
  expr=$1; file=${2:-msm_impl.h}; shift; shift || true
  sed -i "$expr" $v/pkg/csrc/$file
done
cd $root
make -j8 SRC=$v/pkg/csrc OBJDIR=build/obj_$name LIBDIR=icicle-snark_amd/lib_$name icicle-snark_amd/lib_$name/libicicle_snark_hip.so 2>&1 | grep -E "error|Error" -A5 || true
ls -la icicle-snark_amd/lib_$name/libicicle_snark_hip.so
