#!/usr/bin/env python3
"""Instruction histogram per kernel from a hipcc -save-temps gfx950 .s file (tools only)."""
import collections
import re
import sys

s = open(sys.argv[1]).read()
starts = [(m.start(), m.group(1)) for m in re.finditer(r"^(_ZN2vg\w+):", s, flags=re.M)]
for (pos, name), nxt in zip(starts, starts[1:] + [(len(s), None)]):
    body = s[pos:nxt[0]].split("s_endpgm")[0]
    ins = [l.split()[0] for l in body.split("\n") if l.startswith("\t") and l.strip() and not l.startswith("\t.") and not l.startswith("\t;")]
    c = collections.Counter(ins)
    pick = lambda f: sum(v for k, v in c.items() if f(k))
    print("%-58s n=%5d f64=%5d div=%3d rcp=%3d sqrt=%3d ds=%3d gst=%3d gld=%3d sld=%3d br=%3d waitcnt=%3d" % (
        name[:58], len(ins), pick(lambda k: "f64" in k), c.get("v_div_scale_f64", 0) // 2, c.get("v_rcp_f64_e32", 0),
        c.get("v_sqrt_f64_e32", 0) + c.get("v_rsq_f64_e32", 0), pick(lambda k: k.startswith("ds_")),
        pick(lambda k: k.startswith("global_store")), pick(lambda k: k.startswith("global_load")),
        pick(lambda k: k.startswith("s_load")), pick(lambda k: k.startswith("s_cbranch")), c.get("s_waitcnt", 0)))
