#!/usr/bin/env python3
"""Renders the gate table of libakaze_hip.so (csrc/akz_gates.hpp through akz_debug_gates) as the markdown table of DESIGN.md
section 6.1 -- between the markers <!-- gates:begin --> and <!-- gates:end -->.  `--check`: exit 1 if DESIGN.md is out of date
(a CPU test runs this).  Needs the built library, no GPU."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "akaze-rust_amd", "python"))
import akaze_amd  # noqa: E402


def fmt(v):
    v = float(v)
    if v >= 1e6 and v % 1e5 == 0:
        return f"{v / 1e6:g} M"
    if v >= 1 << 20 and v % (1 << 20) == 0:
        return f"{int(v) >> 20} Mi"
    return f"{int(v):,}".replace(",", " ")


def table():
    lines = ["| gate | value | counts | below / from there on |", "|---|---|---|---|"]
    for g in akaze_amd.gates():
        lines.append(f"| `{g['name']}` | {fmt(g['value'])} | {g['unit']} | {g['meaning']} |")
    return "\n".join(lines)


def main():
    path = os.path.join(ROOT, "DESIGN.md")
    text = open(path).read()
    a, b = "<!-- gates:begin -->", "<!-- gates:end -->"
    if a not in text or b not in text:
        print("DESIGN.md has no gate markers", file=sys.stderr)
        return 1
    new = text[:text.index(a) + len(a)] + "\n" + table() + "\n" + text[text.index(b):]
    if "--check" in sys.argv:
        if new != text:
            print("DESIGN.md's gate table is out of date: run tools/render_gates.py", file=sys.stderr)
            return 1
        return 0
    open(path, "w").write(new)
    print(table())
    return 0


if __name__ == "__main__":
    sys.exit(main())
