"""A Lua 5.1 lexer + recursive-descent parser (test infrastructure).

There is no Lua / LuaJIT in the image, so the Lua the repo ships (lua/*.lua, and the reference's scripts after lua/patches/*.patch)
is syntax-checked by this parser instead: the complete Lua 5.1 grammar (reference manual section 8) -- statements, the operator
precedence table, function bodies with varargs, method calls, string-call and table-call sugar, table constructors, long strings
and long comments with levels, numeric literals incl. hex and LuaJIT's LL / ULL / i suffixes.  A syntax error raises
LuaSyntaxError(line, message).

Besides accepting / rejecting, `parse` resolves every name against the lexical scopes and reports the FREE names of the chunk:
`Chunk.globals_read` / `Chunk.globals_written` (name -> first line).  tests/test_lua_binding.py checks them against what
train.lua defines before it requires the re-hosted files."""
import re

KEYWORDS = {"and", "break", "do", "else", "elseif", "end", "false", "for", "function", "if", "in", "local", "nil", "not", "or",
            "repeat", "return", "then", "true", "until", "while"}
# longest first
SYMBOLS = ["...", "..", "==", "~=", "<=", ">=", "+", "-", "*", "/", "%", "^", "#", "<", ">", "=", "(", ")", "{", "}", "[", "]", ";",
           ":", ",", "."]


class LuaSyntaxError(Exception):
    def __init__(self, line, msg):
        Exception.__init__(self, "line %d: %s" % (line, msg))
        self.line = line


class Tok(object):
    __slots__ = ("kind", "val", "line")

    def __init__(self, kind, val, line):
        self.kind, self.val, self.line = kind, val, line

    def __repr__(self):
        return "%s(%r)@%d" % (self.kind, self.val, self.line)


_NUM = re.compile(r"0[xX][0-9a-fA-F]+(?:\.[0-9a-fA-F]*)?(?:[pP][+-]?\d+)?(?:ULL|LL|ull|ll|i)?|"
                  r"(?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?(?:ULL|LL|ull|ll|i)?")
_NAME = re.compile(r"[A-Za-z_][A-Za-z_0-9]*")


def tokenize(src):
    toks, i, line, n = [], 0, 1, len(src)
    if src.startswith("#"):                      # shebang line
        i = src.index("\n") if "\n" in src else n
    while i < n:
        c = src[i]
        if c == "\n":
            line += 1; i += 1; continue
        if c in " \t\r":
            i += 1; continue
        if src.startswith("--", i):
            m = re.match(r"--\[(=*)\[", src[i:])
            if m:                                # long comment
                close = "]" + m.group(1) + "]"
                j = src.find(close, i + m.end())
                if j < 0:
                    raise LuaSyntaxError(line, "unfinished long comment")
                line += src.count("\n", i, j); i = j + len(close); continue
            j = src.find("\n", i)
            i = n if j < 0 else j
            continue
        m = re.match(r"\[(=*)\[", src[i:])
        if m:                                    # long string
            close = "]" + m.group(1) + "]"
            j = src.find(close, i + m.end())
            if j < 0:
                raise LuaSyntaxError(line, "unfinished long string")
            toks.append(Tok("string", src[i + m.end():j], line))
            line += src.count("\n", i, j); i = j + len(close); continue
        if c in "'\"":
            j, buf = i + 1, []
            while True:
                if j >= n or src[j] == "\n":
                    raise LuaSyntaxError(line, "unfinished string")
                if src[j] == "\\":
                    if j + 1 < n and src[j + 1] == "\n":
                        line += 1
                    buf.append(src[j:j + 2]); j += 2; continue
                if src[j] == c:
                    break
                buf.append(src[j]); j += 1
            toks.append(Tok("string", "".join(buf), line)); i = j + 1; continue
        if c.isdigit() or (c == "." and i + 1 < n and src[i + 1].isdigit()):
            m = _NUM.match(src, i)
            j = m.end()
            if j < n and (src[j].isalnum() or src[j] == "_"):
                raise LuaSyntaxError(line, "malformed number near %r" % src[i:j + 1])
            toks.append(Tok("number", m.group(0), line)); i = j; continue
        m = _NAME.match(src, i)
        if m:
            w = m.group(0)
            toks.append(Tok("keyword" if w in KEYWORDS else "name", w, line)); i = m.end(); continue
        for s in SYMBOLS:
            if src.startswith(s, i):
                toks.append(Tok("sym", s, line)); i += len(s); break
        else:
            raise LuaSyntaxError(line, "unexpected character %r" % c)
    toks.append(Tok("eof", None, line))
    return toks


class Chunk(object):
    def __init__(self):
        self.globals_read, self.globals_written = {}, {}
        self.n_statements = 0
        self.functions = []          # (line, qualified name or None, n_params, is_vararg)
        self.calls = []              # (line, callee as a dotted / colon path when it is one, n_args or None for string / table sugar)


BINPRI = {"or": (1, 1), "and": (2, 2), "<": (3, 3), ">": (3, 3), "<=": (3, 3), ">=": (3, 3), "~=": (3, 3), "==": (3, 3),
          "..": (5, 4), "+": (6, 6), "-": (6, 6), "*": (7, 7), "/": (7, 7), "%": (7, 7), "^": (10, 9)}      # (left, right)
UNARY_PRI = 8


class Parser(object):
    def __init__(self, src):
        self.t = tokenize(src)
        self.p = 0
        self.chunk = Chunk()
        self.scopes = [set()]
        self.loop_depth = [0]        # per function: `break` needs an enclosing loop

    # ---- token helpers
    @property
    def tok(self):
        return self.t[self.p]

    def check(self, val, kind=None):
        t = self.tok
        return t.val == val and t.kind in (("sym", "keyword") if kind is None else (kind,))

    def accept(self, val):
        if self.check(val):
            self.p += 1
            return True
        return False

    def expect(self, val, what=None):
        if not self.accept(val):
            t = self.tok
            raise LuaSyntaxError(t.line, "'%s' expected%s near %s" % (val, " (%s)" % what if what else "",
                                                                      "<eof>" if t.kind == "eof" else repr(t.val)))

    def name(self):
        t = self.tok
        if t.kind != "name":
            raise LuaSyntaxError(t.line, "<name> expected near %s" % ("<eof>" if t.kind == "eof" else repr(t.val)))
        self.p += 1
        return t.val

    # ---- scopes
    def declare(self, n):
        self.scopes[-1].add(n)

    def is_local(self, n):
        return any(n in s for s in self.scopes)

    def read(self, n, line):
        if not self.is_local(n):
            self.chunk.globals_read.setdefault(n, line)

    def write(self, n, line):
        if not self.is_local(n):
            self.chunk.globals_written.setdefault(n, line)

    # ---- grammar
    def parse_chunk(self):
        self.block()
        if self.tok.kind != "eof":
            raise LuaSyntaxError(self.tok.line, "'<eof>' expected near %r" % self.tok.val)
        return self.chunk

    def block_ends(self):
        t = self.tok
        return t.kind == "eof" or (t.kind == "keyword" and t.val in ("end", "else", "elseif", "until"))

    def block(self, scope=True):
        if scope:
            self.scopes.append(set())
        while not self.block_ends():
            if self.check("return", "keyword"):
                self.p += 1
                if not self.block_ends() and not self.check(";"):
                    self.explist()
                self.accept(";")
                if not self.block_ends():
                    raise LuaSyntaxError(self.tok.line, "'return' must be the last statement of a block")
                break
            if self.check("break", "keyword"):
                if self.loop_depth[-1] == 0:
                    raise LuaSyntaxError(self.tok.line, "no loop to break")
                self.p += 1
                self.accept(";")
                if not self.block_ends():
                    raise LuaSyntaxError(self.tok.line, "'break' must be the last statement of a block")
                break
            self.statement()
            self.accept(";")
        if scope:
            self.scopes.pop()

    def statement(self):
        self.chunk.n_statements += 1
        t = self.tok
        if t.kind == "keyword":
            k = t.val
            if k == "if":
                self.p += 1; self.exp(); self.expect("then"); self.block()
                while self.accept("elseif"):
                    self.exp(); self.expect("then"); self.block()
                if self.accept("else"):
                    self.block()
                self.expect("end", "to close 'if' at line %d" % t.line)
                return
            if k == "while":
                self.p += 1; self.exp(); self.expect("do")
                self.loop_depth[-1] += 1; self.block(); self.loop_depth[-1] -= 1
                self.expect("end", "to close 'while' at line %d" % t.line)
                return
            if k == "do":
                self.p += 1; self.block(); self.expect("end", "to close 'do' at line %d" % t.line)
                return
            if k == "for":
                self.p += 1
                names = [self.name()]
                self.scopes.append(set())
                if self.accept("="):
                    self.exp(); self.expect(","); self.exp()
                    if self.accept(","):
                        self.exp()
                else:
                    while self.accept(","):
                        names.append(self.name())
                    self.expect("in"); self.explist()
                self.expect("do")
                for nm in names:
                    self.declare(nm)
                self.loop_depth[-1] += 1; self.block(); self.loop_depth[-1] -= 1
                self.scopes.pop()
                self.expect("end", "to close 'for' at line %d" % t.line)
                return
            if k == "repeat":
                self.p += 1
                self.scopes.append(set())
                self.loop_depth[-1] += 1; self.block(scope=False); self.loop_depth[-1] -= 1
                self.expect("until", "to close 'repeat' at line %d" % t.line)
                self.exp()                       # the condition sees the block's locals
                self.scopes.pop()
                return
            if k == "function":
                self.p += 1
                line = self.tok.line
                first = self.name()
                path, method = first, False
                self.read(first, line) if (self.check(".") or self.check(":")) else self.write(first, line)
                while self.accept("."):
                    path += "." + self.name()
                if self.accept(":"):
                    path += ":" + self.name(); method = True
                self.funcbody(path, method, t.line)
                return
            if k == "local":
                self.p += 1
                if self.accept("function"):
                    nm = self.name()
                    self.declare(nm)             # visible inside its own body (recursion)
                    self.funcbody(nm, False, t.line)
                    return
                names = [self.name()]
                while self.accept(","):
                    names.append(self.name())
                if self.accept("="):
                    self.explist()
                for nm in names:                 # ... but not inside its own initialiser
                    self.declare(nm)
                return
            raise LuaSyntaxError(t.line, "unexpected symbol near %r" % t.val)
        # exprstat: assignment or call
        kind, nm = self.suffixedexp(target=True)
        if self.check("=") or self.check(","):
            targets = [(kind, nm, t.line)]
            while self.accept(","):
                l2 = self.tok.line
                k2, n2 = self.suffixedexp(target=True)
                targets.append((k2, n2, l2))
            self.expect("=")
            self.explist()
            for k2, n2, l2 in targets:
                if k2 == "call":
                    raise LuaSyntaxError(l2, "cannot assign to a function call")
                if k2 == "paren":
                    raise LuaSyntaxError(l2, "cannot assign to a parenthesised expression")
                if k2 == "name":
                    self.write(n2, l2)
            return
        if kind != "call":
            raise LuaSyntaxError(t.line, "syntax error: an expression is not a statement (near %r)" % (self.tok.val,))

    def funcbody(self, qualified, method, line):
        self.expect("(")
        self.scopes.append(set())
        self.loop_depth.append(0)
        nparams, vararg = 0, False
        if method:
            self.declare("self")
        if not self.check(")"):
            while True:
                if self.accept("..."):
                    vararg = True
                    break
                self.declare(self.name()); nparams += 1
                if not self.accept(","):
                    break
        self.expect(")")
        self.chunk.functions.append((line, qualified, nparams, vararg))
        self.vararg_ok = getattr(self, "vararg_ok", [True]) + [vararg]
        self.block(scope=False)
        self.vararg_ok.pop()
        self.expect("end", "to close 'function' at line %d" % line)
        self.loop_depth.pop()
        self.scopes.pop()

    def explist(self):
        n = 1
        self.exp()
        while self.accept(","):
            self.exp(); n += 1
        return n

    def primaryexp(self, target):
        t = self.tok
        if t.kind == "name":
            self.p += 1
            return "name", t.val
        if self.accept("("):
            self.exp(); self.expect(")", "to close '(' at line %d" % t.line)
            return "paren", None
        raise LuaSyntaxError(t.line, "unexpected symbol near %s" % ("<eof>" if t.kind == "eof" else repr(t.val)))

    def suffixedexp(self, target=False):
        """Returns (kind, name): kind in name / index / call / paren; a bare name is resolved by the caller when `target`."""
        line = self.tok.line
        kind, nm = self.primaryexp(target)
        path = nm
        first = True
        while True:
            t = self.tok
            if t.kind == "sym" and t.val in (".", "[", ":", "(", "{") or t.kind == "string":
                if first and kind == "name":
                    self.read(nm, line)          # a name that is indexed / called is READ, whatever follows
                first = False
            if self.accept("."):
                f = self.name(); kind = "index"
                path = path + "." + f if path else None
            elif self.accept("["):
                self.exp(); self.expect("]"); kind = "index"; path = None
            elif self.accept(":"):
                f = self.name()
                path = path + ":" + f if path else None
                self.chunk.calls.append((t.line, path, self.callargs()))
                kind = "call"; path = None
            elif self.check("(") or self.check("{") or self.tok.kind == "string":
                if self.check("(") and t.line != self.t[self.p - 1].line:
                    raise LuaSyntaxError(t.line, "ambiguous syntax (function call x new statement)")
                self.chunk.calls.append((t.line, path, self.callargs()))
                kind = "call"; path = None
            else:
                break
        if kind == "name" and not target:
            self.read(nm, line)
        return kind, nm

    def callargs(self):
        t = self.tok
        if t.kind == "string":
            self.p += 1
            return None
        if self.check("{"):
            self.table()
            return None
        self.expect("(")
        n = 0
        if not self.check(")"):
            n = self.explist()
        self.expect(")", "to close '(' at line %d" % t.line)
        return n

    def table(self):
        t = self.tok
        self.expect("{")
        while not self.check("}"):
            if self.check("["):
                self.p += 1; self.exp(); self.expect("]"); self.expect("="); self.exp()
            elif self.tok.kind == "name" and self.t[self.p + 1].kind == "sym" and self.t[self.p + 1].val == "=":
                self.p += 2; self.exp()
            else:
                self.exp()
            if not (self.accept(",") or self.accept(";")):
                break
        self.expect("}", "to close '{' at line %d" % t.line)

    def simpleexp(self):
        t = self.tok
        if t.kind in ("number", "string"):
            self.p += 1
            return
        if t.kind == "keyword" and t.val in ("nil", "true", "false"):
            self.p += 1
            return
        if self.check("..."):
            if not getattr(self, "vararg_ok", [True])[-1]:
                raise LuaSyntaxError(t.line, "cannot use '...' outside a vararg function")
            self.p += 1
            return
        if self.check("{"):
            self.table()
            return
        if self.check("function", "keyword"):
            self.p += 1
            self.funcbody(None, False, t.line)
            return
        self.suffixedexp()

    def exp(self, limit=0):
        t = self.tok
        if (t.kind == "keyword" and t.val == "not") or (t.kind == "sym" and t.val in ("-", "#")):
            self.p += 1
            self.exp(UNARY_PRI)
        else:
            self.simpleexp()
        while True:
            t = self.tok
            op = t.val if (t.kind == "sym" or (t.kind == "keyword" and t.val in ("and", "or"))) else None
            if op not in BINPRI or BINPRI[op][0] <= limit:
                break
            self.p += 1
            self.exp(BINPRI[op][1])


def parse(src):
    """Parse a Lua 5.1 chunk; returns a Chunk or raises LuaSyntaxError."""
    return Parser(src).parse_chunk()
