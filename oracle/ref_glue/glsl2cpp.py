#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- part of the oracle/_ref recipe.  Mechanical GLSL -> C++ rewriter.

Reads the reference's shader sources WHERE THEY LIE (argv[1], normally /root/reference/shaders) and writes
compilable C++ into a scratch directory (argv[2], outside the repository): no reference source is copied into the
tree, only oracle/_ref/libref.so is kept.  Line numbers are preserved (comments are blanked, never removed), so a
compiler diagnostic in <scratch>/pathtrace.glsl:213 points at /root/reference/shaders/pathtrace.glsl:213.

The rewrite is purely lexical -- the arithmetic, the control flow and the call order of the shaders are untouched:
  R1  comments blanked; `#version`, `#extension`, `precision ...;` lines blanked
  R2  `#ifdef __cplusplus` -> `#if 0` (the shaders' GLSL branch is the one compiled; the host branch of
      compress.glsl / host_device.h is compiled separately, unmodified, by ref_host.cpp)
  R3  floating literals without suffix get `f` (GLSL `1.0` is a 32-bit float, C++ `1.0` is a double)
  R4  parameter qualifiers: `in T x` -> `T x`, `out T x` / `inout T x` -> `T& x`
  R5  interface declarations:
        layout(..) uniform Block { T name; };            -> T name;            (filled by ref_glue before a dispatch)
        layout(..) buffer  Block { T name[]; };          -> const T* name;
        layout(..) uniform sampler2D name[];             -> const sampler2D* name;
        layout(..) uniform <opaque type> name;           -> <opaque type> name;
        layout(buffer_reference, ..) buffer B { T m[]; };-> struct B { const T* m; B(uint64_t a) : m((const T*)a) {} };
        layout(location = n) in|out T name;              -> thread_local T name;
        layout(location = n) rayPayload[In]EXT T name;   -> thread_local T name;   (ref_rtx_*.cpp move the payload bytes between stages)
        hitAttributeEXT T name;                          -> thread_local T name;
        ignoreIntersectionEXT;                           -> { gl_IgnoreIntersection = true; return; }
        layout(local_size_x = ..) in;                    -> (blank)
  R6  `void main()` -> `void shader_main()`;  shader-global mutable variables (`PtPayload prd;` ...) become thread_local
  R7  array constructor `T[n](a, b, ..)` -> `{a, b, ..}`
  R8  GLSL evaluates call arguments left to right (4.60 section 6.1.1); C++ leaves the order unspecified except inside a braced
      initialiser.  Constructor calls whose arguments call rand() more than once -- vec3(rand(s), rand(s), rand(s)) -- therefore
      become vec3{rand(s), rand(s), rand(s)}  (env_sampling.glsl:129, pathtrace.glsl:353, random.glsl:106)
  R9  GLSL-only spellings with no C++ counterpart, rewritten in place (each listed in PATCHES below with its file:line)
"""
import os
import re
import sys

PATCHES = {
    # GLSL allows a swizzle on a scalar (`uint.x`); C++ does not.      shade_state.glsl:103-105
    "shade_state.glsl": [(".tangent.x)", ".tangent)")],
    # `.xy` of an ivec2 passed to vec2(): identity swizzle on an integer vector.      pathtrace.glsl:357
    "pathtrace.glsl": [("vec2(sizeImage.xy)", "vec2(sizeImage)")],
}


def blank_comments(src):
    out = []
    i, n = 0, len(src)
    while i < n:
        if src.startswith("//", i):
            j = src.find("\n", i)
            j = n if j < 0 else j
            i = j
        elif src.startswith("/*", i):
            j = src.find("*/", i + 2)
            j = n if j < 0 else j + 2
            out.append("".join(ch if ch == "\n" else " " for ch in src[i:j]))
            i = j
        else:
            out.append(src[i])
            i += 1
    return "".join(out)


FLOAT_LIT = re.compile(r"(?<![\w.])((?:\d+\.\d*|\.\d+)(?:[eE][-+]?\d+)?|\d+[eE][-+]?\d+)(?![\w.])")


def keep_lines(m, repl):
    """replacement text padded with the newlines of the matched text (line numbers stay aligned with the reference)"""
    return repl + "\n" * m.group(0).count("\n")


def brace_ordered_constructors(s):
    out, i = [], 0
    for m in re.finditer(r"\b(vec[234])\s*\(", s):
        if m.start() < i:
            continue
        depth, j = 1, m.end()
        while depth:
            depth += {"(": 1, ")": -1}.get(s[j], 0)
            j += 1
        args = s[m.end():j - 1]
        if args.count("rand(") >= 2:
            out.append(s[i:m.start()] + m.group(1) + "{" + args + "}")
            i = j
    out.append(s[i:])
    return "".join(out)


def rewrite(name, src):
    s = blank_comments(src)
    for a, b in PATCHES.get(name, []):
        assert a in s, f"{name}: patch target {a!r} not found -- the reference changed, review PATCHES"
        s = s.replace(a, b)
    # R1
    s = re.sub(r"^[ \t]*#[ \t]*(version|extension)\b[^\n]*", "", s, flags=re.M)
    s = re.sub(r"^[ \t]*precision\s+\w+\s+\w+\s*;", "", s, flags=re.M)
    # R2
    s = re.sub(r"^[ \t]*#[ \t]*ifdef[ \t]+__cplusplus\b[^\n]*", "#if 0", s, flags=re.M)
    # R5 (before R4: `in` / `out` of interface declarations are not parameter qualifiers)
    s = re.sub(r"layout\s*\(\s*local_size[^)]*\)\s*in\s*;", lambda m: keep_lines(m, ""), s)
    s = re.sub(r"layout\s*\(\s*buffer_reference[^)]*\)\s*buffer\s+(\w+)\s*\{\s*(\w+)\s+(\w+)\s*\[\s*\]\s*;\s*\}\s*;",
               lambda m: keep_lines(m, f"struct {m.group(1)} {{ const {m.group(2)}* {m.group(3)}; {m.group(1)}(uint64_t a_) : {m.group(3)}((const {m.group(2)}*)a_) {{}} }};"), s)
    s = re.sub(r"layout\s*\([^)]*\)\s*uniform\s+\w+\s*\{\s*(\w+)\s+(\w+)\s*;\s*\}\s*;", lambda m: keep_lines(m, f"{m.group(1)} {m.group(2)};"), s)
    s = re.sub(r"layout\s*\([^)]*\)\s*buffer\s+\w+\s*\{\s*(\w+)\s+(\w+)\s*\[\s*\]\s*;\s*\}\s*;", lambda m: keep_lines(m, f"const {m.group(1)}* {m.group(2)};"), s)
    s = re.sub(r"layout\s*\([^)]*\)\s*uniform\s+(\w+)\s+(\w+)\s*\[\s*\]\s*;", lambda m: keep_lines(m, f"const {m.group(1)}* {m.group(2)};"), s)
    s = re.sub(r"layout\s*\([^)]*\)\s*uniform\s+(\w+)\s+(\w+)\s*;", lambda m: keep_lines(m, f"{m.group(1)} {m.group(2)};"), s)
    s = re.sub(r"layout\s*\(\s*location[^)]*\)\s*(?:in|out)\s+(\w+)\s+(\w+)\s*;", lambda m: keep_lines(m, f"thread_local {m.group(1)} {m.group(2)};"), s)
    s = re.sub(r"layout\s*\(\s*location[^)]*\)\s*rayPayload(?:In)?EXT\s+(\w+)\s+(\w+)\s*;", lambda m: keep_lines(m, f"thread_local {m.group(1)} {m.group(2)};"), s)
    s = re.sub(r"^[ \t]*hitAttributeEXT\s+(\w+)\s+(\w+)\s*;", r"thread_local \1 \2;", s, flags=re.M)
    s = re.sub(r"\bignoreIntersectionEXT\s*;", "{ gl_IgnoreIntersection = true; return; }", s)
    left = re.search(r"\blayout\s*\(", s)
    assert not left, f"{name}: unhandled layout declaration: " + s[left.start():left.start() + 120]
    # R3
    s = FLOAT_LIT.sub(lambda m: m.group(1) + "f", s)
    # R4
    s = re.sub(r"\b(?:inout|out)\s+(\w+)\s+(\w+)", r"\1& \2", s)
    s = re.sub(r"\bin\s+(\w+)\s+(\w+)", r"\1 \2", s)
    # R6
    s = re.sub(r"\bvoid\s+main\s*\(\s*\)", "void shader_main()", s)
    s = re.sub(r"^(PtPayload|ShadowHitPayload)(\s+\w+\s*;)", r"thread_local \1\2", s, flags=re.M)
    # R8
    s = brace_ordered_constructors(s)
    # R7
    s = re.sub(r"=\s*\w+\s*\[\s*\d+\s*\]\s*\(([^;]*)\)\s*;", r"= {\1};", s)
    return s


def main():
    src_dir, out_dir = sys.argv[1], sys.argv[2]
    os.makedirs(out_dir, exist_ok=True)
    for f in sorted(os.listdir(src_dir)):
        if not f.endswith((".glsl", ".h", ".comp", ".frag", ".rgen", ".rchit", ".rahit", ".rmiss")):
            continue
        with open(os.path.join(src_dir, f)) as fh:
            txt = fh.read()
        with open(os.path.join(out_dir, f), "w") as fh:
            fh.write(rewrite(f, txt))


if __name__ == "__main__":
    main()
