#!/bin/bash
# tools/libir_prev.so = the library with ONE source taken from a git revision (default HEAD), everything else from the in-tree objects:
# the "prev" arm of tools/ab_lib.sh.   usage: tools/build_prev.sh igemm.hip [rev]
set -e
cd "$(dirname "$0")/.."
SRC=$1; REV=${2:-HEAD}
C=instarevive_amd/csrc
TMP=$C/_prev_$SRC
git show "$REV:$C/$SRC" > "$TMP"
EXTRA=""; case "$SRC" in *.cpp) EXTRA="-x hip";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=fast $EXTRA -c "$TMP" -o tools/_prev.o
OBJS=""
for o in igemm conv_s1 conv_s1_fp8 norm attention attn_d512 attn_fp8 attn_d512_fp8 swin_fused elementwise vae_io t5 unet api; do
  if [ "$o" = "${SRC%.*}" ]; then OBJS="$OBJS tools/_prev.o"; else OBJS="$OBJS $C/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/libir_prev.so $OBJS
rm -f "$TMP" tools/_prev.o
echo tools/libir_prev.so
