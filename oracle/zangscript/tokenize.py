"""zangscript tokenizer (src/zangscript/tokenize.zig:7-223).

Token kinds carry the reference's names: name, number, enum_value, sym_*, kw_*, illegal,
end_of_file.  A source location is (line, index) as in context.zig:14-20."""
import struct
from dataclasses import dataclass

from .errors import ScriptError, SourceRange

SYMBOLS = [            # tried in this order (tokenize.zig:186-199 / the union's field order :13-21)
    ("sym_asterisk", "*"), ("sym_colon", ":"), ("sym_comma", ","), ("sym_equals", "="),
    ("sym_left_paren", "("), ("sym_minus", "-"), ("sym_plus", "+"), ("sym_right_paren", ")"),
    ("sym_slash", "/"),
]
KEYWORDS = {k: "kw_" + k for k in
            ("begin", "defcurve", "defmodule", "deftrack", "delay", "end", "false", "feedback", "from", "out", "true")}
SYMBOL_TEXT = dict(SYMBOLS)
KEYWORD_TEXT = {v: k for k, v in KEYWORDS.items()}


def f32(x):
    """Round a Python float to f32 (what std.fmt.parseFloat(f32, ...) returns)."""
    return struct.unpack("<f", struct.pack("<f", x))[0]


@dataclass
class Token:
    tt: str
    sr: SourceRange
    number: float = 0.0


def _head(ch):
    return ("a" <= ch <= "z") or ("A" <= ch <= "Z")            # leading underscore is not allowed (:164-167)


def _tail(ch):
    return _head(ch) or ("0" <= ch <= "9") or ch == "_"


class Tokenizer:
    def __init__(self, source):
        self.source = source           # errors.Source
        self.line = 0
        self.index = 0

    def loc(self):
        return (self.line, self.index)

    def next(self):
        src = self.source.contents
        line, i = self.line, self.index
        try:
            while True:
                while i < len(src) and src[i] in " \t\r\n":
                    if src[i] == "\r":
                        i += 1
                        if i == len(src) or src[i] != "\n":     # a lone CR ends a line (:68-74)
                            line += 1
                            continue
                    if src[i] == "\n":
                        line += 1
                    i += 1
                if i + 2 < len(src) and src[i] == "/" and src[i + 1] == "/":
                    while i < len(src) and src[i] not in "\r\n":
                        i += 1
                    continue
                if i == len(src):
                    return Token("end_of_file", SourceRange((line, i), (line, i)))
                start = (line, i)
                for tt, text in SYMBOLS:
                    if src.startswith(text, i):
                        i += len(text)
                        return Token(tt, SourceRange(start, (line, i)))
                if src[i] == ".":
                    i += 1
                    start2 = (line, i)
                    if i == len(src) or not _head(src[i]):
                        raise ScriptError(self.source, SourceRange(start, start2), "dot must be followed by an identifier")
                    i += 1
                    while i < len(src) and _tail(src[i]):
                        i += 1
                    return Token("enum_value", SourceRange(start2, (line, i)))
                if "0" <= src[i] <= "9":
                    j = i + 1
                    while j < len(src) and (("0" <= src[j] <= "9") or src[j] == "."):
                        j += 1
                    text = src[i:j]
                    i = j
                    if text.count(".") > 1:
                        raise ScriptError(self.source, SourceRange(start, (line, i)), "malformatted number")
                    return Token("number", SourceRange(start, (line, i)), f32(float(text)))
                if _head(src[i]):
                    i += 1
                    while i < len(src) and _tail(src[i]):
                        i += 1
                    text = src[start[1]:i]
                    return Token(KEYWORDS.get(text, "name"), SourceRange(start, (line, i)))
                i += 1
                return Token("illegal", SourceRange(start, (line, i)))
        finally:
            self.line, self.index = line, i

    def peek(self):
        saved = (self.line, self.index)
        try:
            return self.next()
        finally:
            self.line, self.index = saved

    def fail_expected(self, desc, found):                       # tokenize.zig:140-146
        if found.tt == "end_of_file":
            return ScriptError(self.source, found.sr, "expected %s, found end of file" % desc)
        return ScriptError(self.source, found.sr, "expected %s, found `%s`" % (desc, self.source.text(found.sr)))

    def expect_next(self, tt):                                   # :149-158
        token = self.next()
        if token.tt == tt:
            return token
        text = SYMBOL_TEXT.get(tt) or KEYWORD_TEXT[tt]
        raise self.fail_expected("`%s`" % text, token)
