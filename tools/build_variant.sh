#!/bin/bash
# Builds libfractalrenderer_amd.so from a git ref (or "work" = the working tree) into build/ab/<name>.so, for A/B runs
# of two kernel versions in ONE gpurun session (tools/ab_libs.sh; the library is picked with FR_LIB_PATH).
# usage: tools/build_variant.sh <git-ref|work> <name>
set -e
REF="$1"; NAME="$2"; ROOT="$(cd "$(dirname "$0")/.." && pwd)"
TMP="$(mktemp -d)"; mkdir -p "$ROOT/build/ab" "$TMP/include" "$TMP/fractalrenderer_amd/csrc"
if [ "$REF" = work ]; then
  cp "$ROOT"/include/*.h "$TMP/include/"; cp "$ROOT"/fractalrenderer_amd/csrc/{*.c,*.cpp,*.h,*.hip,*.inc,Makefile} "$TMP/fractalrenderer_amd/csrc/"
else
  for f in $(git -C "$ROOT" ls-tree -r --name-only "$REF" include fractalrenderer_amd/csrc); do git -C "$ROOT" show "$REF:$f" > "$TMP/$f"; done
fi
make -C "$TMP/fractalrenderer_amd/csrc" -s ARCH=gfx950 OUT="$ROOT/build/ab/$NAME.so" ROOT="$TMP" "$ROOT/build/ab/$NAME.so"
rm -rf "$TMP"; ls -la "$ROOT/build/ab/$NAME.so"
