"""Generate the `ffi.cdef` block of lua/facegen_hip.lua from include/facegen_hip.h (the single source of truth) and splice
it between the `-- BEGIN GENERATED CDEF` / `-- END GENERATED CDEF` markers.  `--check` only reports whether the file is
up to date (tests/test_lua_binding.py runs it)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "facegen_hip.h")
LUA = os.path.join(ROOT, "lua", "facegen_hip.lua")
BEGIN, END = "-- BEGIN GENERATED CDEF (scripts/gen_lua_cdef.py)", "-- END GENERATED CDEF"


def cdef_text():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    body = src[src.index('extern "C" {') + len('extern "C" {'):]
    body = body[:body.index("#ifdef __cplusplus")]
    body = "\n".join(l for l in body.splitlines() if not l.lstrip().startswith("#"))
    out = []
    for stmt in body.split(";"):
        stmt = " ".join(stmt.split())
        if not stmt:
            continue
        out.append(stmt + ";")
    # a struct / enum body contains ';' or ',' -- re-join the pieces that were split inside braces
    joined, depth, cur = [], 0, ""
    for piece in out:
        cur = (cur + " " + piece).strip() if cur else piece
        depth += piece.count("{") - piece.count("}")
        if depth == 0:
            joined.append(cur)
            cur = ""
    return "\n".join(joined)


def main():
    text = open(LUA).read()
    a, b = text.index(BEGIN), text.index(END)
    new = text[:a] + BEGIN + "\nffi.cdef[[\n" + cdef_text() + "\n]]\n" + text[b:]
    if "--check" in sys.argv:
        sys.exit(0 if new == text else 1)
    open(LUA, "w").write(new)


if __name__ == "__main__":
    main()
