#!/bin/bash
# like build_variant.sh, but the patch is a python script that edits files under the copied csrc tree:
#   scratch/build_variant_py.sh <name> <patch.py>     (patch.py receives the path of the copied csrc as argv[1])
set -e
name=$1; patch=$2
root=$(cd "$(dirname "$0")/.." && pwd)
v=$root/build/var_$name
rm -rf $v; mkdir -p $v/pkg
cp -r $root/icicle-snark_amd/csrc $v/pkg/csrc
ln -s $root/include $v/include
python3 $patch $v/pkg/csrc
cd $root
make -j8 SRC=$v/pkg/csrc OBJDIR=build/obj_$name LIBDIR=icicle-snark_amd/lib_$name icicle-snark_amd/lib_$name/libicicle_snark_hip.so 2>&1 | grep -E "error|Error" -A5 || true
ls -la icicle-snark_amd/lib_$name/libicicle_snark_hip.so
