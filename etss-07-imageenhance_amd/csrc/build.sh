#!/bin/bash
# Build libie_hip.so (gfx950) in-tree: one hipcc per source in parallel, then link.
# Usage: build.sh [extra hipcc flags];  IE_OUT=path overrides the output, IE_JOBS the parallelism.
# Objects are cached in csrc/_obj/<flags-hash>/ and rebuilt only when the source, a header or the flags changed.
set -e
cd "$(dirname "$0")"
OUT=${IE_OUT:-../libie_hip.so}
SRCS=$(ls ie_*.hip)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function $*"
KEY=$(printf '%s' "$FLAGS" | md5sum | cut -c1-10)
OBJ=_obj/$KEY
mkdir -p "$OBJ"
# signature of the csrc/ie_*.h headers a source includes (transitively) + the LAYOUT part of the public ABI header include/ie_hip.h:
# its struct / enum bodies and #define lines (a changed ie_mconv_desc must recompile every object that reads one), not its function
# declarations (a new entry point would otherwise recompile every file, ~4 minutes).
ABI_SIG=$(python3 - <<'PY'
import hashlib, re
src = open("../../include/ie_hip.h").read()
src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
src = re.sub(r"//[^\n]*", "", src)
parts = re.findall(r"typedef\s+(?:struct|enum)\b[^{;]*\{.*?\}\s*\w+\s*;", src, flags=re.S) + re.findall(r"^[ \t]*#[ \t]*define[^\n]*", src, flags=re.M)
print(hashlib.md5(" ".join(" ".join(p.split()) for p in parts).encode()).hexdigest()[:16])
PY
)
hdr_sig() {
    local seen="" todo="$1" f inc
    while [ -n "$todo" ]; do
        set -- $todo; f=$1; shift; todo="$*"
        case " $seen " in *" $f "*) continue;; esac
        seen="$seen $f"
        for inc in $(grep -o '#include "[^"]*"' "$f" | sed 's/#include "//; s/"//'); do
            case "$inc" in */include/ie_hip.h) continue;; esac
            [ -f "$inc" ] && todo="$todo $inc"
        done
    done
    { cat $seen; echo "$ABI_SIG"; } | md5sum | cut -c1-16
}
todo=""
for s in $SRCS; do
    sig="$(hdr_sig "$s")"
    if [ -n "$IE_FORCE" ] || [ ! -f "$OBJ/$s.o" ] || [ "$(cat "$OBJ/$s.sig" 2>/dev/null)" != "$sig" ]; then
        todo="$todo $s"; rm -f "$OBJ/$s.sig"; echo "$sig" > "$OBJ/$s.sig.new"   # the signature of what is compiled NOW (edits during the build re-trigger)
    fi
done
if [ -n "$todo" ]; then
    printf '%s\n' $todo | xargs -P "${IE_JOBS:-8}" -I{} sh -c "cp {} $OBJ/{}.src.hip && hipcc $FLAGS -I. -c $OBJ/{}.src.hip -o $OBJ/{}.o.tmp && mv $OBJ/{}.o.tmp $OBJ/{}.o && mv $OBJ/{}.sig.new $OBJ/{}.sig"
fi
objs=""
for s in $SRCS; do objs="$objs $OBJ/$s.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" $objs
echo "built $(readlink -f $OUT) (recompiled:${todo:- nothing})"
