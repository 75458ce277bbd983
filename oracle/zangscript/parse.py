"""zangscript parser (src/zangscript/parse.zig).  Produces the same tree the reference's codegen
consumes: globals, curves, tracks, modules (builtins first), scopes of statements, expressions."""
import math
from dataclasses import dataclass, field
from typing import List, Optional

from .builtins import BOOLEAN, BUFFER, COB, CONSTANT, CURVE, ModuleParam, ParamType
from .errors import ScriptError, SourceRange
from .tokenize import Token, Tokenizer, f32

RESERVED_NAMES = ("abs", "cos", "max", "min", "pi", "pow", "sample_rate", "sin", "sqrt")      # parse.zig:185-195
UNARY_FUNCTIONS = {"abs": "abs", "cos": "cos", "sin": "sin", "sqrt": "sqrt"}
BINARY_FUNCTIONS = {"max": "max", "min": "min", "pow": "pow"}
BINARY_OPERATORS = (("sym_plus", 1, "add"), ("sym_minus", 1, "sub"), ("sym_asterisk", 2, "mul"), ("sym_slash", 2, "div"))   # :513-518


@dataclass
class NumberLiteral:
    value: float
    verbatim: str          # the literal as written, so 0.7 does not become 0.699999988079071 (:131-136)


@dataclass
class Expr:
    """kind: call | track_call | delay | literal_boolean | literal_number | literal_enum_value |
    literal_curve | literal_track | literal_module | un_arith | bin_arith | local | feedback | name
    (parse.zig:143-158)."""
    kind: str
    sr: SourceRange
    a: object = None
    b: object = None
    op: str = None
    value: object = None
    args: list = None
    scope: object = None
    token: Token = None


@dataclass
class CallArg:
    param_name: str
    param_name_token: Token
    value: Expr


@dataclass
class Statement:
    kind: str              # let_assignment | output | feedback
    expr: Expr
    local_index: int = -1


@dataclass
class Scope:
    parent: Optional["Scope"]
    statements: List[Statement] = field(default_factory=list)


@dataclass
class Curve:
    points: list           # [(NumberLiteral t, NumberLiteral value)]


@dataclass
class TrackNote:
    t: NumberLiteral
    args_sr: SourceRange
    args: List[CallArg]


@dataclass
class Track:
    params: List[ModuleParam]
    notes: List[TrackNote]


@dataclass
class Module:
    params: List[ModuleParam]
    builtin_name: Optional[str] = None
    zig_package_name: Optional[str] = None
    scope: Optional[Scope] = None          # None for builtins
    locals: Optional[List[str]] = None


@dataclass
class Global:
    name: str
    value: Expr


@dataclass
class ParseResult:
    globals: List[Global]
    curves: List[Curve]
    tracks: List[Track]
    modules: List[Module]


class _ModuleState:
    def __init__(self, params):
        self.params = params
        self.locals = []


class Parser:
    def __init__(self, source, packages):
        self.source = source
        self.tok = Tokenizer(source)
        self.globals, self.curves, self.tracks, self.modules = [], [], [], []
        self.enums = []
        for pkg in packages:                                    # parse.zig:766-787
            self.enums.extend(pkg.enums)
            for b in pkg.builtins:
                idx = len(self.modules)
                self.modules.append(Module(list(b.params), b.name, pkg.zig_package_name))
                bogus = SourceRange((0, 0), (0, 0))
                self.globals.append(Global(b.name, Expr("literal_module", bogus, value=idx)))

    # ---- helpers
    def fail(self, sr, msg):
        return ScriptError(self.source, sr, msg)

    def text(self, sr):
        return self.source.text(sr)

    def expr(self, loc0, kind, **kw):                            # createExpr: ends at the tokenizer's position (:452-456)
        return Expr(kind, SourceRange(loc0, self.tok.loc()), **kw)

    # ---- definitions
    def define_curve(self):                                      # :197-231
        points, last_t = [], None
        while True:
            token = self.tok.next()
            if token.tt == "kw_end":
                break
            if token.tt != "number":
                raise self.tok.fail_expected("number or `end`", token)
            if last_t is not None and token.number <= last_t:
                raise self.fail(token.sr, "time value must be greater than the previous time value")
            last_t = token.number
            vt = self.tok.next()
            if vt.tt != "number":
                raise self.tok.fail_expected("number", vt)
            points.append((NumberLiteral(token.number, self.text(token.sr)), NumberLiteral(vt.number, self.text(vt.sr))))
        self.curves.append(Curve(points))
        return len(self.curves) - 1

    def expect_param_type(self, for_track):                      # :233-256
        token = self.tok.next()
        if token.tt != "name":
            raise self.tok.fail_expected("param type", token)
        name = self.text(token.sr)
        simple = {"boolean": BOOLEAN, "constant": CONSTANT, "waveform": BUFFER, "cob": COB, "curve": CURVE}
        if name in simple:
            pt = simple[name]
        else:
            for e in self.enums:
                if e.name == name:
                    pt = ParamType("one_of", e)
                    break
            else:
                raise self.tok.fail_expected("param type", token)
        if for_track and pt.kind in ("buffer", "constant_or_buffer"):
            raise self.fail(token.sr, "track param cannot be cob or waveform")
        return pt

    def parse_param_declarations(self, params, for_track):       # :258-288
        while True:
            token = self.tok.next()
            if token.tt == "kw_begin":
                return
            if token.tt != "name":
                raise self.tok.fail_expected("param declaration or `begin`", token)
            name = self.text(token.sr)
            if name in RESERVED_NAMES:
                raise self.fail(token.sr, "`%s` is a reserved name" % name)
            if any(p.name == name for p in params):
                raise self.fail(token.sr, "redeclaration of param `%s`" % name)
            self.tok.expect_next("sym_colon")
            pt = self.expect_param_type(for_track)
            self.tok.expect_next("sym_comma")
            params.append(ModuleParam(name, pt))

    def define_track(self):                                      # :290-326
        params = []
        self.parse_param_declarations(params, True)
        notes, last_t = [], None
        while True:
            token = self.tok.next()
            if token.tt == "kw_end":
                break
            if token.tt != "number":
                raise self.tok.fail_expected("number or `end`", token)
            if last_t is not None and token.number <= last_t:
                raise self.fail(token.sr, "time value must be greater than the previous time value")
            last_t = token.number
            loc0 = self.tok.loc()
            args = self.parse_call_args(None)
            notes.append(TrackNote(NumberLiteral(token.number, self.text(token.sr)), SourceRange(loc0, self.tok.loc()), args))
        self.tracks.append(Track(params, notes))
        return len(self.tracks) - 1

    def define_module(self):                                     # :328-359
        params = [ModuleParam("sample_rate", CONSTANT)]          # implicitly declared
        self.parse_param_declarations(params, False)
        ms = _ModuleState(params)
        scope = self.parse_statements(ms, None)
        self.modules.append(Module(params, scope=scope, locals=ms.locals))
        return len(self.modules) - 1

    # ---- expressions; pc = None (global context) or (module_state, scope)
    def parse_call_args(self, pc):                               # :366-410
        self.tok.expect_next("sym_left_paren")
        args = []
        token = self.tok.next()
        while token.tt != "sym_right_paren":
            if args:
                if token.tt != "sym_comma":
                    raise self.tok.fail_expected("`,` or `)`", token)
                token = self.tok.next()
            if token.tt != "name":
                raise self.tok.fail_expected("callee param name", token)
            name = self.text(token.sr)
            eq = self.tok.next()
            if eq.tt == "sym_equals":
                args.append(CallArg(name, token, self.expect_expression(pc)))
                token = self.tok.next()
            else:
                if pc is not None:                               # shorthand: `val` expands to `val=val`
                    args.append(CallArg(name, token, self.resolve_name(pc, token)))
                    token = eq
                # (in a global context the reference falls through without consuming: it then loops on the
                # same `token`, i.e. it never terminates on malformed track notes; we report the error)
                else:
                    raise self.tok.fail_expected("`=`", eq)
        return args

    def resolve_name(self, pc, token):                           # :458-493
        if pc is not None:
            ms, scope = pc
            name = self.text(token.sr)
            sc = scope
            while sc is not None:
                for st in reversed(sc.statements):               # later declarations shadow earlier ones
                    if st.kind == "let_assignment" and ms.locals[st.local_index] == name:
                        return Expr("local", token.sr, value=st.local_index)
                sc = sc.parent
        return Expr("name", token.sr, token=token)               # a param or a global: resolved in codegen

    def expect_expression(self, pc, priority=0):                 # :520-565
        negate = False
        if self.tok.peek().tt == "sym_minus":
            self.tok.next()
            negate = True
        a = self.expect_term(pc)
        loc0 = a.sr.loc0
        if self.tok.peek().tt == "sym_left_paren":
            if pc is None:
                raise self.fail(a.sr, "not a function")
            args = self.parse_call_args(pc)
            a = self.expr(loc0, "call", a=a, args=args)
        if negate:
            a = self.expr(loc0, "un_arith", op="neg", a=a)
        while True:
            token = self.tok.peek()
            for sym, prio, op in BINARY_OPERATORS:
                if token.tt == sym and priority < prio:
                    self.tok.next()
                    b = self.expect_expression(pc, prio)
                    a = self.expr(loc0, "bin_arith", op=op, a=a, b=b)
                    break
            else:
                return a

    def _unary(self, pc, loc0, op):
        self.tok.expect_next("sym_left_paren")
        a = self.expect_expression(pc)
        self.tok.expect_next("sym_right_paren")
        return self.expr(loc0, "un_arith", op=op, a=a)

    def _binary(self, pc, loc0, op):
        self.tok.expect_next("sym_left_paren")
        a = self.expect_expression(pc)
        self.tok.expect_next("sym_comma")
        b = self.expect_expression(pc)
        self.tok.expect_next("sym_right_paren")
        return self.expr(loc0, "bin_arith", op=op, a=a, b=b)

    def expect_term(self, pc):                                   # :584-690
        token = self.tok.next()
        loc0 = token.sr.loc0
        tt = token.tt
        if tt == "sym_left_paren":
            a = self.expect_expression(pc)
            self.tok.expect_next("sym_right_paren")
            return a
        if tt == "kw_defmodule":
            return self.expr(loc0, "literal_module", value=self.define_module())
        if tt == "kw_defcurve":
            return self.expr(loc0, "literal_curve", value=self.define_curve())
        if tt == "kw_deftrack":
            return self.expr(loc0, "literal_track", value=self.define_track())
        if tt == "kw_from":
            if pc is None:
                raise self.fail(token.sr, "cannot call track outside of module context")
            ms, scope = pc                                       # parseTrackCall :412-424
            track_expr = self.expect_expression(pc)
            self.tok.expect_next("sym_comma")
            speed = self.expect_expression(pc)
            self.tok.expect_next("kw_begin")
            inner = self.parse_statements(ms, scope)
            return self.expr(loc0, "track_call", a=track_expr, b=speed, scope=inner)
        if tt == "name":
            s = self.text(token.sr)
            if s in UNARY_FUNCTIONS:
                return self._unary(pc, loc0, UNARY_FUNCTIONS[s])
            if s in BINARY_FUNCTIONS:
                return self._binary(pc, loc0, BINARY_FUNCTIONS[s])
            if s == "pi":
                return self.expr(loc0, "literal_number", value=NumberLiteral(f32(math.pi), "std.math.pi"))
            r = self.resolve_name(pc, token)
            return Expr(r.kind, SourceRange(loc0, self.tok.loc()), value=r.value, token=r.token)
        if tt == "kw_false":
            return self.expr(loc0, "literal_boolean", value=False)
        if tt == "kw_true":
            return self.expr(loc0, "literal_boolean", value=True)
        if tt == "number":
            return self.expr(loc0, "literal_number", value=NumberLiteral(token.number, self.text(token.sr)))
        if tt == "enum_value":
            label = self.text(token.sr)
            if self.tok.peek().tt == "sym_left_paren":
                self.tok.next()
                payload = self.expect_expression(pc)
                self.tok.expect_next("sym_right_paren")
                return self.expr(loc0, "literal_enum_value", value=label, a=payload)
            return Expr("literal_enum_value", token.sr, value=label, a=None)
        if tt == "kw_delay":
            if pc is None:
                raise self.fail(token.sr, "cannot use delay outside of module context")
            ms, scope = pc                                       # parseDelay :426-446
            nt = self.tok.next()
            if nt.tt != "number":
                raise self.tok.fail_expected("number", nt)
            s = self.text(nt.sr)
            if not s.isdigit():
                raise self.fail(nt.sr, "malformatted integer")
            self.tok.expect_next("kw_begin")
            inner = self.parse_statements(ms, scope)
            return self.expr(loc0, "delay", value=int(s), scope=inner)
        if tt == "kw_feedback":
            if pc is None:
                raise self.fail(token.sr, "cannot use feedback outside of module context")
            return self.expr(loc0, "feedback")
        raise self.tok.fail_expected("expression", token)

    # ---- statements
    def parse_statements(self, ms, parent):                      # :734-764
        scope = Scope(parent)
        pc = (ms, scope)
        while True:
            token = self.tok.next()
            if token.tt == "kw_end":
                return scope
            if token.tt == "name":                               # parseLocalDecl :692-713
                name = self.text(token.sr)
                self.tok.expect_next("sym_equals")
                if name in RESERVED_NAMES:
                    raise self.fail(token.sr, "`%s` is a reserved name" % name)
                e = self.expect_expression(pc)                   # the new local is not yet visible to its own initialiser
                ms.locals.append(name)
                scope.statements.append(Statement("let_assignment", e, len(ms.locals) - 1))
            elif token.tt == "kw_out":
                scope.statements.append(Statement("output", self.expect_expression(pc)))
            elif token.tt == "kw_feedback":
                scope.statements.append(Statement("feedback", self.expect_expression(pc)))
            else:
                raise self.tok.fail_expected("local declaration, `out`, `feedback` or `end`", token)

    def parse_global_decl(self, token):                          # :715-732
        name = self.text(token.sr)
        self.tok.expect_next("sym_equals")
        if name in RESERVED_NAMES:
            raise self.fail(token.sr, "`%s` is a reserved name" % name)
        if any(g.name == name for g in self.globals):
            raise self.fail(token.sr, "redeclaration of global `%s`" % name)
        self.globals.append(Global(name, self.expect_expression(None)))

    def parse(self):                                             # :789-796
        while True:
            token = self.tok.next()
            if token.tt == "end_of_file":
                break
            if token.tt != "name":
                raise self.tok.fail_expected("declaration or end of file", token)
            self.parse_global_decl(token)
        return ParseResult(self.globals, self.curves, self.tracks, self.modules)


def parse(source, packages):
    return Parser(source, packages).parse()
