#!/usr/bin/env python3
"""Dev tool (runs only in the build container, where /root/reference exists): how close are the host modules that keep the
reference's call signatures to the reference files of the same name?  Token sequences with comments and docstrings removed,
difflib ratio, plus the share of the reference's tokens covered by matching blocks.
   python tools/similarity_check.py"""
import difflib, io, os, sys, tokenize
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SKIP = (tokenize.NL, tokenize.NEWLINE, tokenize.INDENT, tokenize.DEDENT, tokenize.ENDMARKER)


def code_tokens(src):
    out, prev = [], None
    for t in tokenize.generate_tokens(io.StringIO(src).readline):
        if t.type == tokenize.COMMENT:
            continue
        if t.type == tokenize.STRING and (prev is None or prev.type in SKIP):     # docstring / bare string statement
            prev = t
            continue
        if t.type in SKIP:
            prev = t
            continue
        out.append(t.string)
        prev = t
    return out


ref_dir = "/root/reference/code"
for name in sorted(os.listdir(os.path.join(REPO, "stark-anatomy_amd"))):
    ref = os.path.join(ref_dir, name)
    if not name.endswith(".py") or not os.path.exists(ref):
        continue
    a = code_tokens(open(os.path.join(REPO, "stark-anatomy_amd", name)).read())
    b = code_tokens(open(ref).read())
    sm = difflib.SequenceMatcher(None, a, b, autojunk=False)
    covered = sum(blk.size for blk in sm.get_matching_blocks()) / max(1, len(b))
    raw = difflib.SequenceMatcher(None, open(os.path.join(REPO, "stark-anatomy_amd", name)).read().split(), open(ref).read().split(), autojunk=False).ratio()
    print("%-18s tokens %5d vs %5d   code-token ratio %.3f   reference covered %.3f   whitespace-token ratio %.3f" % (name, len(a), len(b), sm.ratio(), covered, raw))
