#!/bin/bash
# Build container only (needs /root/reference): the binding of INTEGRATION.md section 1 -- the replacement body of
# CLZMatcher::do_matching -- compiled against the reference's own headers (-fsyntax-only), so that the stub cannot rot.
# -D_FILE_WRAPPER_H pre-empts the include guard of the gz reader header, whose zlib-ng submodule is not vendored (the
# stub does not touch it); nothing of the reference is copied or written.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
REF=${1:-/root/reference}
[ -d "$REF/src" ] || { echo "no reference at $REF: skipped"; exit 77; }
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
python3 - "$ROOT/INTEGRATION.md" "$TMP/stub.cpp" <<'PY'
import re, sys
md = open(sys.argv[1]).read()
sec = md[md.index("## 1. The stub"):]
code = re.search(r"```cpp\n(.*?)```", sec, re.S).group(1)
open(sys.argv[2], "w").write('#include <algorithm>\n#include <iostream>\n#include <vector>\n#include "lz_matcher.h"\n' + code)
PY
g++ -std=c++20 -fsyntax-only -fpermissive -w -D_FILE_WRAPPER_H -DARCH_X64 -I"$REF/src" -I"$ROOT/include" "$TMP/stub.cpp"
echo "INTEGRATION.md section 1 stub: compiles against the reference's headers"
