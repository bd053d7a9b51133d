#!/bin/bash
# oracle/build_ref.sh -- builds oracle/_ref/ from the reference's own sources where they lie.
#
# Exactly one source on the (widened) path is self-contained: src/audiofilters/g711.c includes
# nothing but its own g711.h, so gcc compiles it unmodified.  Every other file on the path needs
# bctoolbox / oRTP / generated-config headers this image lacks and is not built.
# Output goes to oracle/_ref/ only (git-ignored, travels to the GPU box with the snapshot).
set -e
here=$(cd "$(dirname "$0")" && pwd)
src=/root/reference/src/audiofilters/g711.c
if [ ! -f "$src" ]; then
	echo "build_ref.sh: $src not present (GPU box): keeping the prebuilt oracle/_ref" >&2
	exit 0
fi
mkdir -p "$here/_ref"
gcc -O2 -fPIC -shared -I/root/reference/src/audiofilters -o "$here/_ref/libg711_ref.so" "$src"
echo "build_ref.sh: built $here/_ref/libg711_ref.so from $src" >&2
