#!/usr/bin/env python3
"""Does the Rust shim (akaze-rust_amd/rust/src/lib.rs) cover the reference crate's public API by name and signature?

There is no Rust toolchain in the build image, so the shim cannot be compiled; this script is the next best check:

  1. the public items of the reference crate (`pub fn` with parameter names / types / return type, `pub struct` with
     its public fields, trait methods) are read from the reference sources when they are available
     (`--reference DIR`, default /root/reference) and written to tests/golden/reference_api.json; without the
     sources that snapshot is used;
  2. the same items are read from the shim, module path by module path;
  3. every `extern "C"` declaration of the shim must name a function of include/akaze_hip.h with the same number
     of parameters.

  4. the BODIES: every function of the shim (public or not, methods included) is reduced to the ordered list of
     `ffi::akz_*` functions it names; tests/shim_twin/shim_twin.cpp — a C++ restatement of the same bodies that IS
     compiled against the header and run against the oracle on a GPU (tests/test_gpu_shim_twin.py) — is reduced the same
     way, tag by tag (`// shim: <path>`).  Every shim function that names an `akz_*` function must have a twin with the
     same list, and no twin may exist without its function: a body cannot change without its tested twin changing too.

Exit status 0 = every reference item has a counterpart with an identical signature (type paths are compared by their
last segment: `types::evolution::Config` == `Config`)."""
import argparse
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def strip_comments(s):
    s = re.sub(r"/\*.*?\*/", "", s, flags=re.S)
    return re.sub(r"//[^\n]*", "", s)


def split_top(s, sep=","):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "<([{":
            depth += 1
        elif ch in ">)]}":
            depth -= 1
        if ch == sep and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def norm_type(t):
    t = re.sub(r"\s+", "", t)
    t = re.sub(r"(\b\w+::)+", "", t)  # type paths by their last segment
    return t


def norm_params(p):
    out = []
    for a in split_top(p):
        a = a.strip()
        if not a or a in ("&self", "&mut self", "self"):
            out.append(a.replace(" ", "")) if a else None
            continue
        name, _, ty = a.partition(":")
        name = re.sub(r"\bmut\s+", "", name).strip()
        out.append(f"{name}:{norm_type(ty)}")
    return out


FN = re.compile(r"\bpub\s+fn\s+(\w+)\s*(?:<[^>{]*>)?\s*\(", re.S)
TRAIT_FN = re.compile(r"\bfn\s+(\w+)\s*\(", re.S)


def parse_fn_at(src, m):
    """src[m.end():] starts inside the parameter list"""
    i, depth = m.end(), 1
    while depth and i < len(src):
        depth += src[i] == "("
        depth -= src[i] == ")"
        i += 1
    params = src[m.end():i - 1]
    rest = src[i:]
    j = min([k for k in (rest.find("{"), rest.find(";")) if k >= 0] or [0])
    ret = rest[:j].strip()
    ret = norm_type(ret[2:]) if ret.startswith("->") else ""
    return {"params": norm_params(params), "ret": ret}


def items_of_block(src, mod, api):
    """public functions, structs and traits declared directly in `src` (a module body), recursing into `pub mod`"""
    i = 0
    while i < len(src):
        m = re.compile(r"\b(pub(?:\(crate\))?\s+)?(mod|fn|struct|trait|impl)\b").search(src, i)
        if not m:
            break
        vis, kind = (m.group(1) or "").strip(), m.group(2)
        # body extent
        k = src.find("{", m.end())
        semi = src.find(";", m.end())
        if kind in ("fn",) and False:
            pass
        if k < 0 or (0 <= semi < k and kind in ("mod", "struct")):
            i = (semi if semi >= 0 else m.end()) + 1
            continue
        if kind == "fn":
            fm = FN.match(src, m.start()) if vis == "pub" else None
            # skip the body
            depth, j = 0, src.find("{", m.end())
            p = src.find("(", m.end())
            # find the body's opening brace after the parameter list
            depth_p, q = 1, p + 1
            while depth_p and q < len(src):
                depth_p += src[q] == "("
                depth_p -= src[q] == ")"
                q += 1
            j = src.find("{", q)
            semi2 = src.find(";", q)
            if j < 0 or (0 <= semi2 < j):
                i = semi2 + 1
                continue
            depth, e = 1, j + 1
            while depth and e < len(src):
                depth += src[e] == "{"
                depth -= src[e] == "}"
                e += 1
            if fm:
                api[f"{mod}::{fm.group(1)}" if mod else fm.group(1)] = dict(kind="fn", **parse_fn_at(src, fm))
            i = e
            continue
        depth, e = 1, k + 1
        while depth and e < len(src):
            depth += src[e] == "{"
            depth -= src[e] == "}"
            e += 1
        body = src[k + 1:e - 1]
        head = src[m.end():k]
        if kind == "mod":
            name = head.strip()
            if vis == "pub":
                items_of_block(body, f"{mod}::{name}" if mod else name, api)
        elif kind == "struct" and vis == "pub":
            name = re.match(r"\s*(\w+)", head).group(1)
            fields = [f"{fm.group(1)}:{norm_type(fm.group(2))}" for f in split_top(body)
                      for fm in [re.match(r"\s*(?:#\[[^\]]*\]\s*)*pub\s+(\w+)\s*:\s*(.+)", f.strip(), re.S)] if fm]
            api[f"{mod}::{name}" if mod else name] = {"kind": "struct", "fields": fields}
        elif kind == "trait" and vis == "pub":
            name = re.match(r"\s*(\w+)", head).group(1)
            methods = {}
            for tm in TRAIT_FN.finditer(body):
                methods[tm.group(1)] = parse_fn_at(body, tm)
            api[f"{mod}::{name}" if mod else name] = {"kind": "trait", "methods": methods}
        i = e
    return api


def reference_api(ref_root):
    api = {}
    src_root = os.path.join(ref_root, "akaze", "src")
    for dirpath, _, files in os.walk(src_root):
        for f in sorted(files):
            if not f.endswith(".rs"):
                continue
            rel = os.path.relpath(os.path.join(dirpath, f), src_root)[:-3]
            parts = [p for p in rel.split(os.sep) if p not in ("lib", "mod")]
            text = strip_comments(open(os.path.join(dirpath, f)).read())
            while True:  # unit-test modules (anywhere in the file): drop the brace block
                tm = re.search(r"#\[cfg\(test\)\]\s*mod\s+\w+\s*\{", text)
                if not tm:
                    break
                depth, e = 1, tm.end()
                while depth and e < len(text):
                    depth += text[e] == "{"
                    depth -= text[e] == "}"
                    e += 1
                text = text[:tm.start()] + text[e:]
            items_of_block(text, "::".join(parts), api)
    return api


def shim_api(path):
    return items_of_block(strip_comments(open(path).read()), "", {})


def header_functions(path):
    h = strip_comments(open(path).read())
    out = {}
    for m in re.finditer(r"\b(akz_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", h, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else len(split_top(args))
    return out


def shim_externs(path):
    s = strip_comments(open(path).read())
    out = {}
    for blk in re.finditer(r'extern\s+"C"\s*\{(.*?)\n\s*\}', s, flags=re.S):
        for m in re.finditer(r"\bfn\s+(akz_\w+)\s*\((.*?)\)\s*(?:->\s*[\w:*\s]+)?;", blk.group(1), flags=re.S):
            out[m.group(1)] = len(split_top(m.group(2)))
    return out


def _block_end(src, k):
    """src[k] == '{': index just after the matching '}'"""
    depth, e = 1, k + 1
    while depth and e < len(src):
        depth += src[e] == "{"
        depth -= src[e] == "}"
        e += 1
    return e


def shim_bodies(path):
    """{path of every fn / method / thread_local of the shim: ordered `ffi::akz_*` names in its body}"""
    src = strip_comments(open(path).read())
    out = {}

    def walk(block, prefix):
        i = 0
        pat = re.compile(r"\b(mod\s+(\w+)\s*\{|impl\b([^{;]*)\{|trait\s+\w+[^{;]*\{|extern\s+\"C\"\s*\{|fn\s+(\w+)\s*(?:<[^>{]*>)?\s*\(|"
                         r"thread_local!\s*\{)")
        while True:
            m = pat.search(block, i)
            if not m:
                return
            text = m.group(0)
            if text.startswith("fn"):
                q, depth = m.end(), 1
                while depth and q < len(block):
                    depth += block[q] == "("
                    depth -= block[q] == ")"
                    q += 1
                j, semi = block.find("{", q), block.find(";", q)
                if j < 0 or 0 <= semi < j:
                    i = (semi if semi >= 0 else q) + 1
                    continue
                e = _block_end(block, j)
                name = "::".join(prefix + [m.group(4)])
                out[name] = re.findall(r"\bffi::(akz_\w+)", block[j:e])
                i = e
                continue
            k = m.end() - 1
            e = _block_end(block, k)
            body = block[k + 1:e - 1]
            if text.startswith("mod"):
                walk(body, prefix + [m.group(2)])
            elif text.startswith("impl"):
                head = m.group(3).strip()
                ty = head.split(" for ")[-1].strip()
                ty = re.sub(r"<.*", "", ty)
                # `impl Drop for X { fn drop }` -> X::drop ; `impl Default for Config` etc. likewise
                walk(body, prefix + [ty])
            elif text.startswith("thread_local"):
                tm = re.search(r"static\s+(\w+)", body)
                out["::".join(prefix + [tm.group(1)])] = re.findall(r"\bffi::(akz_\w+)", body)
            # trait declarations and the extern block hold no bodies
            i = e

    walk(src, [])
    return out


def twin_bodies(path, known):
    """{tag: ordered names of C-ABI functions used between `// shim: tag` and the next tag}"""
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"\s*// shim: (\S+)", line)
        if m:
            cur = None if m.group(1) == "end" else m.group(1)
            if cur is not None:
                if cur in out:
                    raise SystemExit(f"twin tag {cur} appears twice")
                out[cur] = []
            continue
        if cur is None:
            continue
        code = re.sub(r"//.*", "", line)
        out[cur] += [t for t in re.findall(r"\bakz_\w+\b", code) if t in known]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--shim", default=os.path.join(ROOT, "akaze-rust_amd", "rust", "src", "lib.rs"))
    ap.add_argument("--snapshot", default=os.path.join(ROOT, "tests", "golden", "reference_api.json"))
    ap.add_argument("--twin", default=os.path.join(ROOT, "tests", "shim_twin", "shim_twin.cpp"))
    ap.add_argument("--quiet", action="store_true")
    args = ap.parse_args()
    if os.path.isdir(os.path.join(args.reference, "akaze", "src")):
        ref = reference_api(args.reference)
        os.makedirs(os.path.dirname(args.snapshot), exist_ok=True)
        json.dump(ref, open(args.snapshot, "w"), indent=1, sort_keys=True)
    else:
        ref = json.load(open(args.snapshot))
    shim = shim_api(args.shim)
    problems = []
    for name, item in sorted(ref.items()):
        got = shim.get(name)
        if got is None:
            problems.append(f"missing: {item['kind']} {name}")
            continue
        if item["kind"] == "fn" and (got.get("params") != item["params"] or got.get("ret") != item["ret"]):
            problems.append(f"signature of {name}: reference ({', '.join(item['params'])}) -> {item['ret'] or '()'}; "
                            f"shim ({', '.join(got.get('params', []))}) -> {got.get('ret') or '()'}")
        if item["kind"] == "struct":
            miss = [f for f in item["fields"] if f not in got.get("fields", [])]
            if miss:
                problems.append(f"struct {name}: public fields missing or of another type: {miss}")
        if item["kind"] == "trait":
            for mn, sig in item["methods"].items():
                if got.get("methods", {}).get(mn) != sig:
                    problems.append(f"trait {name}::{mn}: {got.get('methods', {}).get(mn)} != {sig}")
    hdr = header_functions(os.path.join(ROOT, "include", "akaze_hip.h"))
    ext = shim_externs(args.shim)
    for fn, n in sorted(ext.items()):
        if fn not in hdr:
            problems.append(f'extern "C" {fn}: not declared in include/akaze_hip.h')
        elif hdr[fn] != n:
            problems.append(f'extern "C" {fn}: {n} parameters in the shim, {hdr[fn]} in include/akaze_hip.h')
    bodies = shim_bodies(args.shim)
    twins = twin_bodies(args.twin, set(hdr))
    for name, calls in sorted(bodies.items()):
        if not calls and name not in twins:
            continue  # host-only helper without a twin: nothing crosses the ABI
        if name not in twins:
            problems.append(f"body of {name} calls {calls} but tests/shim_twin has no `// shim: {name}`")
        elif twins[name] != calls:
            problems.append(f"body of {name}: shim calls {calls}, its tested twin calls {twins[name]}")
    for name in sorted(set(twins) - set(bodies)):
        problems.append(f"twin `// shim: {name}` has no function of that path in the shim")
    if not args.quiet:
        print(f"bodies: {len(bodies)} shim functions, {sum(1 for c in bodies.values() if c)} of them cross the ABI; "
              f"{len(twins)} twins ({sum(1 for n in twins if n in bodies and twins[n] == bodies[n])} with identical call lists)")
        print(f"reference public items: {len(ref)} ({sum(1 for v in ref.values() if v['kind'] == 'fn')} functions, "
              f"{sum(1 for v in ref.values() if v['kind'] == 'struct')} structs, "
              f"{sum(1 for v in ref.values() if v['kind'] == 'trait')} traits); shim items: {len(shim)}; "
              f'extern "C" declarations: {len(ext)}')
        for p in problems:
            print("  PROBLEM", p)
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
