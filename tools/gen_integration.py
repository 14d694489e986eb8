"""Regenerate the machine-made parts of INTEGRATION.md from include/avec_hip.h and avec_amd/lib.py: the symbol count and the ctypes struct stubs of the
reference-side binding (between the GENERATED markers).  `python tools/gen_integration.py` rewrites the file; `--check` exits 1 when it is stale
(tests/test_nnet_api.py runs the check and compares the stub's ctypes.sizeof with avec_amd.lib's)."""
import ctypes
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DOC = os.path.join(ROOT, "INTEGRATION.md")
BEGIN, END = "<!-- BEGIN GENERATED: struct stubs (tools/gen_integration.py) -->", "<!-- END GENERATED -->"
NAMES = {ctypes.c_void_p: "ctypes.c_void_p", ctypes.c_longlong: "ctypes.c_longlong", ctypes.c_int: "ctypes.c_int", ctypes.c_float: "ctypes.c_float", ctypes.c_uint: "ctypes.c_uint"}


def struct_stub(cls, cname):
    fields = ['("%s", %s)' % (n, NAMES[t]) for n, t in cls._fields_]
    lines, cur = [], "    _fields_ = ["
    for f in fields:
        if len(cur) + len(f) > 128:
            lines.append(cur.rstrip())
            cur = "                "
        cur += f + ", "
    lines.append(cur.rstrip(", ") + "]")
    return "class %s(ctypes.Structure):      # %s, %d bytes (field order = include/avec_hip.h)\n%s" % (cls.__name__, cname, ctypes.sizeof(cls), "\n".join(lines))


def generated_block():
    from avec_amd.lib import Epilogue, Rows
    return BEGIN + "\n```python\n" + struct_stub(Rows, "avec_rows_t") + "\n" + struct_stub(Epilogue, "avec_epilogue_t") + "\n```\n" + END


def render(text):
    from avec_amd.lib import declared_functions
    n = len(declared_functions())
    text = re.sub(r"`include/avec_hip\.h` \(\d+ `extern \"C\"` symbols", "`include/avec_hip.h` (%d `extern \"C\"` symbols" % n, text)
    i, j = text.index(BEGIN), text.index(END) + len(END)
    return text[:i] + generated_block() + text[j:]


if __name__ == "__main__":
    old = open(DOC).read()
    new = render(old)
    if "--check" in sys.argv:
        sys.exit(0 if new == old else 1)
    open(DOC, "w").write(new)
    print("INTEGRATION.md", "unchanged" if new == old else "rewritten")
