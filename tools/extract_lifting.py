#!/usr/bin/env python3
"""Extract the normative integer lifting networks of Daala's 1-D DCTs.

The sequence of lifting steps (constant, rounding offset, shift, operand
order) in the reference's src/dct.c *is* the definition of the transform:
there is no closed form that reproduces its rounding.  This dev-time tool
preprocesses the reference file with gcc -E, parses the bodies of

    od_bin_fdct{4,8,16,32,64} / od_bin_idct{4,8,16,32,64}
    (src/dct.c:87,127,166,286,366,659,4219,4321,4422,4622)

with a small C-subset parser (scoped declarations, do{}while(0) blocks,
=, +=, -= assignments over + - * >> and the OD_DCT_RSHIFT idiom,
src/filter.h:38-41) and re-emits each network in two neutral forms:

  * oracle/od_lifting_tables.h      - op tables for the CPU oracle's
                                      table-driven interpreter
  * daala_amd/csrc/gen/od_lifting_gen.h - straight-line __device__ code over
                                      renamed registers for the HIP kernels

It runs only where /root/reference exists (the dev container); its outputs
are committed.  Nothing at run time reads the reference.

Usage: python tools/extract_lifting.py [--ref /root/reference]
"""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RS1_RE = re.compile(
    r"\(\(\(int32_t\)\(\(\(uint32_t\)\((\w+)\) >> \(32 - \(1\)\)\) \+ "
    r"\(\1\)\)\) >> \(1\)\)")

TOK_RE = re.compile(r"\s*(>>|\+=|-=|[A-Za-z_]\w*|\d+|[-+*/=;(){}\[\],])")


def tokenize(s):
    toks = []
    pos = 0
    s = s.strip()
    while pos < len(s):
        m = TOK_RE.match(s, pos)
        if not m:
            raise SyntaxError("bad token at: " + s[pos:pos + 40])
        toks.append(m.group(1))
        pos = m.end()
    return toks


class Parser:
    """Parses one function body into a flat list of statements

    ('assign', lhs, op, expr) with lhs/expr trees over uniquely renamed
    registers.  Tree nodes: ('reg', id) ('const', v) ('in', k) ('out', k)
    ('add', a, b) ('sub', a, b) ('mul', a, b) ('shr', a, b) ('neg', a)
    ('rs1', a).
    """

    def __init__(self, toks, in_name, out_name):
        self.t = toks
        self.i = 0
        self.scopes = [{}]
        self.nregs = 0
        self.stmts = []
        self.in_name = in_name
        self.out_name = out_name

    def peek(self):
        return self.t[self.i] if self.i < len(self.t) else None

    def eat(self, tok=None):
        cur = self.t[self.i]
        if tok is not None and cur != tok:
            raise SyntaxError("expected %r got %r at %d" % (tok, cur, self.i))
        self.i += 1
        return cur

    def lookup(self, name):
        for sc in reversed(self.scopes):
            if name in sc:
                return sc[name]
        raise NameError(name)

    # ---- statements -------------------------------------------------
    def block(self):
        self.eat("{")
        self.scopes.append({})
        while self.peek() != "}":
            self.statement()
        self.eat("}")
        self.scopes.pop()

    def statement(self):
        p = self.peek()
        if p == "int":
            self.eat()
            name = self.eat()
            self.eat(";")
            self.scopes[-1][name] = self.nregs
            self.nregs += 1
        elif p == "do":
            self.eat()
            self.block()
            self.eat("while")
            self.eat("(")
            self.eat("0")
            self.eat(")")
            self.eat(";")
        elif p == "{":
            self.block()
        elif p == ";":
            self.eat()
        else:
            lhs = self.lvalue()
            op = self.eat()
            if op not in ("=", "+=", "-="):
                raise SyntaxError("assign op " + op)
            e = self.expr()
            self.eat(";")
            self.stmts.append(("assign", lhs, op, e))

    def const_index(self):
        e = self.expr()
        v = self.fold(e)
        if v is None:
            raise SyntaxError("non-constant index")
        return v

    def fold(self, e):
        k = e[0]
        if k == "const":
            return e[1]
        if k == "stride":
            return 1
        if k in ("add", "sub", "mul"):
            a, b = self.fold(e[1]), self.fold(e[2])
            if a is None or b is None:
                return None
            return {"add": a + b, "sub": a - b, "mul": a * b}[k]
        return None

    def lvalue(self):
        p = self.peek()
        if p == "*":
            self.eat()
            self.eat("(")
            name = self.eat()
            self.eat("+")
            k = self.const_index()
            self.eat(")")
            assert name == self.out_name, name
            return ("out", k)
        name = self.eat()
        if self.peek() == "[":
            self.eat()
            k = self.const_index()
            self.eat("]")
            assert name == self.out_name, name
            return ("out", k)
        return ("reg", self.lookup(name))

    # ---- expressions: shift < additive < multiplicative < unary ------
    def expr(self):
        a = self.additive()
        while self.peek() == ">>":
            self.eat()
            b = self.additive()
            a = ("shr", a, b)
        return a

    def additive(self):
        a = self.mult()
        while self.peek() in ("+", "-"):
            op = self.eat()
            b = self.mult()
            a = ("add" if op == "+" else "sub", a, b)
        return a

    def mult(self):
        a = self.unary()
        while self.peek() == "*":
            self.eat()
            b = self.unary()
            a = ("mul", a, b)
        return a

    def unary(self):
        p = self.peek()
        if p == "-":
            self.eat()
            return ("neg", self.unary())
        if p == "(":
            # cast or parenthesised expression
            if self.t[self.i + 1] in ("od_coeff", "int32_t", "int") \
               and self.t[self.i + 2] == ")":
                self.i += 3
                return self.unary()
            self.eat("(")
            e = self.expr()
            self.eat(")")
            return e
        if p == "*":
            self.eat()
            self.eat("(")
            name = self.eat()
            self.eat("+")
            k = self.const_index()
            self.eat(")")
            assert name == self.in_name, name
            return ("in", k)
        if p == "RS1":
            self.eat()
            self.eat("(")
            e = self.expr()
            self.eat(")")
            return ("rs1", e)
        tok = self.eat()
        if tok.isdigit():
            return ("const", int(tok))
        if tok == "xstride":
            return ("stride",)
        if self.peek() == "[":
            self.eat()
            k = self.const_index()
            self.eat("]")
            assert tok == self.in_name, tok
            return ("in", k)
        return ("reg", self.lookup(tok))


def extract_function(src, fname):
    m = re.search(r"^void %s\(([^)]*)\) \{" % fname, src, re.M)
    if not m:
        raise KeyError(fname)
    start = m.end() - 1
    depth = 0
    i = start
    while True:
        c = src[i]
        if c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0:
                break
        i += 1
    body = src[start:i + 1]
    body = RS1_RE.sub(r"RS1(\1)", body)
    assert "uint32_t" not in body, "unmatched OD_DCT_RSHIFT form in " + fname
    fwd = "fdct" in fname
    in_name, out_name = ("x", "y") if fwd else ("y", "x")
    p = Parser(tokenize(body), in_name, out_name)
    p.block()
    assert p.i == len(p.t)
    return p


# ---- emit 1: op tables for the oracle's interpreter -------------------
# ops: 0 LOAD d,k | 1 STORE k,a | 2 ADD d,a,b | 3 SUB d,a,b | 4 RS1 d,a
#      5 MULSH d,a,C,R,S  (d = (a*C + R) >> S) | 6 NEG d,a | 7 SHR d,a,S
#      8 MOV d,a
OPN = dict(LOAD=0, STORE=1, ADD=2, SUB=3, RS1=4, MULSH=5, NEG=6, SHR=7, MOV=8)


class OpEmitter:
    def __init__(self, nregs):
        self.nregs = nregs
        self.ops = []

    def tmp(self):
        r = self.nregs
        self.nregs += 1
        return r

    def ev(self, e):
        k = e[0]
        if k == "reg":
            return e[1]
        if k == "in":
            d = self.tmp()
            self.ops.append((OPN["LOAD"], d, 0, 0, e[1], 0, 0))
            return d
        if k == "rs1":
            a = self.ev(e[1])
            d = self.tmp()
            self.ops.append((OPN["RS1"], d, a, 0, 0, 0, 0))
            return d
        if k == "neg":
            a = self.ev(e[1])
            d = self.tmp()
            self.ops.append((OPN["NEG"], d, a, 0, 0, 0, 0))
            return d
        if k == "shr":
            s = e[2]
            assert s[0] == "const"
            x = e[1]
            # (a*C + R) >> S
            if x[0] == "add" and x[1][0] == "mul" and x[2][0] == "const" \
               and x[1][2][0] == "const":
                a = self.ev(x[1][1])
                d = self.tmp()
                self.ops.append((OPN["MULSH"], d, a, 0,
                                 x[1][2][1], x[2][1], s[1]))
                return d
            a = self.ev(x)
            d = self.tmp()
            self.ops.append((OPN["SHR"], d, a, 0, 0, 0, s[1]))
            return d
        if k in ("add", "sub"):
            a = self.ev(e[1])
            b = self.ev(e[2])
            d = self.tmp()
            self.ops.append((OPN["ADD" if k == "add" else "SUB"],
                             d, a, b, 0, 0, 0))
            return d
        raise ValueError("cannot lower %r" % (e,))

    def stmt(self, st):
        _, lhs, op, e = st
        v = self.ev(e)
        if lhs[0] == "out":
            assert op == "="
            self.ops.append((OPN["STORE"], 0, v, 0, lhs[1], 0, 0))
            return
        d = lhs[1]
        if op == "=":
            # retarget the producing op when it wrote a fresh temp
            if self.ops and self.ops[-1][1] == v and v >= self.base_tmp \
               and self.ops[-1][0] != OPN["STORE"]:
                o = self.ops[-1]
                self.ops[-1] = (o[0], d) + o[2:]
            else:
                self.ops.append((OPN["MOV"], d, v, 0, 0, 0, 0))
        else:
            self.ops.append((OPN["ADD" if op == "+=" else "SUB"],
                             d, d, v, 0, 0, 0))


def lower(p):
    em = OpEmitter(p.nregs)
    em.base_tmp = p.nregs
    for st in p.stmts:
        em.stmt(st)
    return em.ops, em.nregs


# ---- emit 2: straight-line device code --------------------------------
def cxx(e):
    k = e[0]
    if k == "reg":
        return "r%d" % e[1]
    if k == "const":
        return str(e[1])
    if k == "in":
        return "in[%d]" % e[1]
    if k == "rs1":
        return "od_rs1(%s)" % cxx(e[1])
    if k == "neg":
        return "(-%s)" % cxx(e[1])
    if k == "shr":
        x = e[1]
        if x[0] == "add" and x[1][0] == "mul" and x[2][0] == "const" \
           and x[1][2][0] == "const":
            return "od_lift(%s, %d, %d, %d)" % (cxx(x[1][1]), x[1][2][1],
                                                 x[2][1], e[2][1])
        return "((%s) >> %s)" % (cxx(e[1]), cxx(e[2]))
    if k == "add":
        return "(%s + %s)" % (cxx(e[1]), cxx(e[2]))
    if k == "sub":
        return "(%s - %s)" % (cxx(e[1]), cxx(e[2]))
    if k == "mul":
        return "(%s*%s)" % (cxx(e[1]), cxx(e[2]))
    raise ValueError(e)


def strip_parens(s):
    if s.startswith("(") and s.endswith(")"):
        depth = 0
        for i, c in enumerate(s):
            if c == "(":
                depth += 1
            elif c == ")":
                depth -= 1
                if depth == 0 and i != len(s) - 1:
                    return s
        return s[1:-1]
    return s


def regs_of(e, acc):
    if e[0] == "reg":
        acc.add(e[1])
    elif e[0] not in ("const", "in", "stride", "out"):
        for c in e[1:]:
            if isinstance(c, tuple):
                regs_of(c, acc)
    return acc


def slice_outputs(p, want):
    """Backward slice: the statements needed to produce outputs `want`."""
    live = set()
    keep = [False] * len(p.stmts)
    for idx in range(len(p.stmts) - 1, -1, -1):
        _, lhs, op, e = p.stmts[idx]
        if lhs[0] == "out":
            if lhs[1] in want:
                keep[idx] = True
                regs_of(e, live)
        elif lhs[1] in live:
            keep[idx] = True
            if op == "=":
                live.discard(lhs[1])
            regs_of(e, live)
    return keep


def emit_device_half(name, n, p, parity):
    """The half network producing outputs of one parity (out[k] = y[2k + parity]).
    After the first butterfly stage an n-point DCT separates into an n/2-point
    asymmetric DCT on the sums (even outputs) and an n/2-point asymmetric DST on
    the differences (odd outputs) that share nothing else: two lanes can each run
    one half on the same input column."""
    want = set(range(parity, n, 2))
    keep = slice_outputs(p, want)
    used = set()
    for k, st in zip(keep, p.stmts):
        if k:
            if st[1][0] == "reg":
                used.add(st[1][1])
            regs_of(st[3], used)
    lines = []
    lines.append("template <typename T> __device__ __forceinline__ void "
                 "%s(T (&out)[%d], const T (&in)[%d]) {" % (name, n // 2, n))
    lines.append("  T " + ", ".join("r%d" % i for i in sorted(used)) + ";")
    for k, (_, lhs, op, e) in zip(keep, p.stmts):
        if not k:
            continue
        rhs = strip_parens(cxx(e))
        if lhs[0] == "out":
            lines.append("  out[%d] = %s;" % (lhs[1] // 2, rhs))
        else:
            lines.append("  r%d %s %s;" % (lhs[1], op, rhs))
    lines.append("}")
    return "\n".join(lines), sum(keep)


def emit_device_inverse_split(n, p):
    """An n-point INVERSE network as two independent parts and a joining stage (round 5).

    Read backwards, the forward network's structure (an n/2-point asymmetric DCT of the sums, an
    n/2-point asymmetric DST of the differences, then butterflies) makes the inverse start with two
    sub-networks that share nothing - one fed by the even-indexed inputs only, one by the odd-indexed
    ones - and end in a joining stage (the butterflies) that needs both.  Nothing here assumes that
    structure: the statements are put in SSA form and classified by the parity of the inputs they
    transitively depend on; part P holds the statements that depend on inputs of parity P alone,
    the join the rest.  Two lanes can each run one part on the same input line, exchange the
    values that cross the cut and produce half of the outputs each:

      od_idctN_lift_part<P>(mid[M_P], in_P[n/2])       in_P[k] = in[2k + P]
      od_idctN_lift_join<H>(out_H[n/2], mid0, mid1)    out_H[k] = out[H*n/2 + k]

    Returns (code, stats)."""
    ver = {}
    cur = {}

    def rename(e):
        k = e[0]
        if k == "reg":
            return ("ssa", cur[e[1]])
        if k in ("const", "in", "stride", "out"):
            return e
        return (k,) + tuple(rename(c) if isinstance(c, tuple) else c for c in e[1:])

    ssa = []      # (name or ('out', i), expr)
    for _, lhs, op, e in p.stmts:
        rhs = rename(e)
        if lhs[0] == "out":
            ssa.append((lhs, rhs))
            continue
        r = lhs[1]
        if op == "+=":
            rhs = ("add", ("ssa", cur[r]), rhs)
        elif op == "-=":
            rhs = ("sub", ("ssa", cur[r]), rhs)
        elif op != "=":
            raise ValueError(op)
        ver[r] = ver.get(r, -1) + 1
        name = "v%d_%d" % (r, ver[r])
        cur[r] = name
        ssa.append((name, rhs))

    dep = {}

    def deps(e):
        k = e[0]
        if k == "in":
            return {e[1] & 1}
        if k == "ssa":
            return set(dep[e[1]])
        if k in ("const", "stride", "out"):
            return set()
        d = set()
        for c in e[1:]:
            if isinstance(c, tuple):
                d |= deps(c)
        return d

    def uses(e, acc):
        if e[0] == "ssa":
            acc.add(e[1])
        elif e[0] not in ("const", "in", "stride", "out"):
            for c in e[1:]:
                if isinstance(c, tuple):
                    uses(c, acc)
        return acc

    cls = []
    for lhs, rhs in ssa:
        d = deps(rhs)
        if isinstance(lhs, tuple):
            cls.append(2)             # an output assignment belongs to the join
        else:
            dep[lhs] = d
            if not d:
                raise ValueError("statement without an input: %s" % lhs)
            cls.append(2 if len(d) == 2 else next(iter(d)))
    # values that cross the cut, in the order the join first reads them
    mids = [[], []]
    seen = set()
    for (lhs, rhs), c in zip(ssa, cls):
        if c != 2:
            continue
        for u in sorted(uses(rhs, set())):
            if u not in seen and len(dep[u]) == 1:
                seen.add(u)
                mids[next(iter(dep[u]))].append(u)

    def cx(e, part):
        k = e[0]
        if k == "ssa":
            return e[1]
        if k == "in":
            assert part is not None and (e[1] & 1) == part
            return "in[%d]" % (e[1] >> 1)
        if k == "const":
            return str(e[1])
        if k == "rs1":
            return "od_rs1(%s)" % cx(e[1], part)
        if k == "neg":
            return "(-%s)" % cx(e[1], part)
        if k == "shr":
            x = e[1]
            if x[0] == "add" and x[1][0] == "mul" and x[2][0] == "const" and x[1][2][0] == "const":
                return "od_lift(%s, %d, %d, %d)" % (cx(x[1][1], part), x[1][2][1], x[2][1], e[2][1])
            return "((%s) >> %s)" % (cx(e[1], part), cx(e[2], part))
        if k == "add":
            return "(%s + %s)" % (cx(e[1], part), cx(e[2], part))
        if k == "sub":
            return "(%s - %s)" % (cx(e[1], part), cx(e[2], part))
        if k == "mul":
            return "(%s*%s)" % (cx(e[1], part), cx(e[2], part))
        raise ValueError(e)

    out = []
    stats = []
    h = n // 2
    out.append("/* od_idct%d_lift as two independent parts (inputs of one parity each) and a joining stage:\n"
               "   see tools/extract_lifting.py emit_device_inverse_split.  %d / %d values cross the cut. */"
               % (n, len(mids[0]), len(mids[1])))
    out.append("constexpr int kIdct%dMid0 = %d;\nconstexpr int kIdct%dMid1 = %d;" % (n, len(mids[0]), n, len(mids[1])))
    for part in (0, 1):
        lines = ["template <typename T> __device__ __forceinline__ void od_idct%d_lift_part%d("
                 "T (&mid)[%d], const T (&in)[%d]) {" % (n, part, len(mids[part]), h)]
        cnt = 0
        for (lhs, rhs), c in zip(ssa, cls):
            if c == part:
                lines.append("  const T %s = %s;" % (lhs, strip_parens(cx(rhs, part))))
                cnt += 1
        for i, m in enumerate(mids[part]):
            lines.append("  mid[%d] = %s;" % (i, m))
        lines.append("}")
        out.append("\n".join(lines))
        stats.append(("  idct%d part %d" % (n, part), cnt, len(mids[part]), 0))
    index = {m: ("m%d[%d]" % (pp, i)) for pp in (0, 1) for i, m in enumerate(mids[pp])}
    for half in (0, 1):
        want = set(range(half*h, half*h + h))
        # backward slice of the join for these outputs
        live = set()
        keep = [False]*len(ssa)
        for idx in range(len(ssa) - 1, -1, -1):
            lhs, rhs = ssa[idx]
            if cls[idx] != 2:
                continue
            if isinstance(lhs, tuple):
                if lhs[1] in want:
                    keep[idx] = True
                    uses(rhs, live)
            elif lhs in live:
                keep[idx] = True
                uses(rhs, live)
        lines = ["template <typename T> __device__ __forceinline__ void od_idct%d_lift_join%d("
                 "T (&out)[%d], const T (&m0)[%d], const T (&m1)[%d]) {" % (n, half, h, len(mids[0]), len(mids[1]))]
        cnt = 0
        used_mids = sorted(u for u in live if u in index)
        for u in used_mids:
            lines.append("  const T %s = %s;" % (u, index[u]))
        for idx, ((lhs, rhs), k) in enumerate(zip(ssa, keep)):
            if not k:
                continue
            if isinstance(lhs, tuple):
                lines.append("  out[%d] = %s;" % (lhs[1] - half*h, strip_parens(cx(rhs, None))))
            else:
                lines.append("  const T %s = %s;" % (lhs, strip_parens(cx(rhs, None))))
                cnt += 1
        lines.append("}")
        out.append("\n".join(lines))
        stats.append(("  idct%d join %d" % (n, half), cnt, len(used_mids), 0))
        # what this join needs from the OTHER part: indices into that part's mid[] (the lane that ran
        # part 1 - half exports exactly these to the lane that runs this join)
        other = 1 - half
        need = [i for i, m in enumerate(mids[other]) if m in live]
        out.append("/* join%d reads these entries of part%d's mid[] (the other lane's values that cross the cut) */\n"
                   "constexpr int kIdct%dNeed%d = %d;\n"
                   "static constexpr unsigned char kIdct%dNeedIdx%d[%d] = {%s};"
                   % (half, other, n, half, len(need), n, half, len(need), ", ".join(map(str, need))))
    return "\n\n".join(out), stats


def emit_device(name, n, p):
    lines = []
    lines.append("template <typename T> __device__ __forceinline__ void "
                 "%s(T (&out)[%d], const T (&in)[%d]) {" % (name, n, n))
    lines.append("  T " + ", ".join("r%d" % i for i in range(p.nregs)) + ";")
    for _, lhs, op, e in p.stmts:
        rhs = strip_parens(cxx(e))
        if lhs[0] == "out":
            lines.append("  out[%d] = %s;" % (lhs[1], rhs))
        else:
            lines.append("  r%d %s %s;" % (lhs[1], op, rhs))
    lines.append("}")
    return "\n".join(lines)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    src = subprocess.run(
        ["gcc", "-E", "-P", "-I", args.ref + "/include", "-I",
         args.ref + "/src", args.ref + "/src/dct.c"],
        check=True, capture_output=True, text=True).stdout

    tables = []
    device = []
    stats = []
    for kind in ("fdct", "idct"):
        for n in (4, 8, 16, 32, 64):
            fname = "od_bin_%s%d" % (kind, n)
            p = extract_function(src, fname)
            ops, nregs = lower(p)
            nmul = sum(1 for o in ops if o[0] == OPN["MULSH"])
            stats.append((fname, len(ops), nregs, nmul))
            tname = "OD_LIFT_%s%d" % (kind.upper(), n)
            rows = ",\n".join("  {%d,%d,%d,%d,%d,%d,%d}" % o for o in ops)
            tables.append(
                "/* %s: %d ops, %d registers, %d multiplies "
                "(restates %s, src/dct.c). */\n"
                "#define %s_NREGS %d\n#define %s_NOPS %d\n"
                "static const od_lift_op %s[%d] = {\n%s\n};\n"
                % (fname, len(ops), nregs, nmul, fname, tname, nregs,
                   tname, len(ops), tname, len(ops), rows))
            device.append(emit_device("od_%s%d_lift" % (kind, n), n, p))
            if kind == "fdct" and n >= 16:
                for parity, tag in ((0, "even"), (1, "odd")):
                    code, cnt = emit_device_half("od_fdct%d_lift_%s" % (n, tag), n, p, parity)
                    device.append(code)
                    stats.append(("  fdct%d %s half" % (n, tag), cnt, 0, 0))
            if kind == "idct" and n >= 32:
                code, st = emit_device_inverse_split(n, p)
                device.append(code)
                stats.extend(st)

    hdr = (
        "/* GENERATED by tools/extract_lifting.py - do not edit.\n"
        "   Lifting networks of the reference's 1-D transforms\n"
        "   (src/dct.c od_bin_fdctN, od_bin_idctN), restated as op tables.\n"
        "   ops: 0 LOAD d<-in[C] | 1 STORE out[C]<-a | 2 ADD | 3 SUB |\n"
        "        4 RS1 (OD_DCT_RSHIFT(a,1), src/filter.h:38-41) |\n"
        "        5 MULSH d=(a*C+R)>>S | 6 NEG | 7 SHR | 8 MOV */\n"
        "#ifndef OD_LIFTING_TABLES_H\n#define OD_LIFTING_TABLES_H\n"
        "typedef struct { short op, d, a, b; int c, r, s; } od_lift_op;\n\n")
    # LOAD/STORE carry the element index in field c.
    fixed = []
    for t in tables:
        fixed.append(t)
    with open(os.path.join(ROOT, "oracle", "od_lifting_tables.h"), "w") as f:
        f.write(hdr + "\n".join(fixed) + "\n#endif\n")

    os.makedirs(os.path.join(ROOT, "daala_amd", "csrc", "gen"), exist_ok=True)
    with open(os.path.join(ROOT, "daala_amd", "csrc", "gen",
                           "od_lifting_gen.h"), "w") as f:
        f.write(
            "/* GENERATED by tools/extract_lifting.py - do not edit.\n"
            "   Straight-line lifting networks of the reference's 1-D\n"
            "   transforms (src/dct.c od_bin_fdctN, od_bin_idctN) over renamed\n"
            "   registers.  in[]/out[] are in natural order; every array index\n"
            "   is a compile-time constant so the arrays live in VGPRs.\n"
            "   od_rs1  = OD_DCT_RSHIFT(a,1) (src/filter.h:38-41)\n"
            "   od_lift = (a*C + R) >> S, 32-bit wrapping product like the C. */\n"
            "#pragma once\n\n" + "\n\n".join(device) + "\n")
    for s in stats:
        print("%-16s ops=%4d regs=%4d muls=%3d" % s)


if __name__ == "__main__":
    sys.exit(main())
