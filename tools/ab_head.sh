#!/bin/bash
# builds ../lib/libkmerhip_base.so from the committed kernels (HEAD) beside the working tree's build, for a same-box A/B:
#   bash tools/ab_head.sh && gpurun -- 'bash tools/ab_libs.sh "libkmerhip_base.so libkmerhip.so libkmerhip_base.so libkmerhip.so" "--k 21|--hg"'
set -e
cd "$(dirname "$0")/.."
tmp=$(mktemp -d)
for f in $(git diff --name-only -- krust_amd/csrc); do cp $f $tmp/$(basename $f); git show HEAD:$f > $f; done
make -C krust_amd/csrc -j8 VARIANT=_base 2>&1 | grep -i "error" || true
for f in $(ls $tmp); do cp $tmp/$f krust_amd/csrc/$f; touch krust_amd/csrc/$f; done
make -C krust_amd/csrc -j8 2>&1 | grep -i "error" || true
rm -rf $tmp krust_amd/lib/obj_base
ls -la krust_amd/lib/*.so
