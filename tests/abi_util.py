# -*- coding: utf-8 -*-
"""Parse include/vqvae_hip.h into {function: signature-code} (i/f/p/l/u) for ABI consistency tests."""
import os
import re

HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "vqvae_hip.h")


def header_protos():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(int|const char\*)\s+(vqh_\w+)\s*\(([^)]*)\)\s*;", src):
        name, args = m.group(2), m.group(3).strip()
        code = ""
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a or a.startswith("vqh_stream_t"):
                    code += "p"
                elif a.startswith("long long"):
                    code += "l"
                elif a.startswith("unsigned"):
                    code += "u"
                elif a.startswith("float"):
                    code += "f"
                elif a.startswith("int"):
                    code += "i"
                else:
                    raise ValueError(f"unparsed argument {a!r} in {name}")
        out[name] = code
    return out
