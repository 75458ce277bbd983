"""Source, locations and compile errors (src/zangscript/context.zig:4-25, fail.zig:47-116)."""
from dataclasses import dataclass


@dataclass(frozen=True)
class SourceRange:
    loc0: tuple            # (line, index), line counted from 0
    loc1: tuple


@dataclass
class Source:
    filename: str
    contents: str

    def text(self, sr):
        return self.contents[sr.loc0[1]:sr.loc1[1]]


class ScriptError(Exception):
    """error.Failed with the message fail() would have printed: `file:line:col: message`, the
    offending source line and a caret underline (fail.zig:47-111)."""

    def __init__(self, source, sr, message):
        self.source, self.sr, self.message = source, sr, message
        super().__init__(self.render())

    def render(self):
        src, sr = self.source, self.sr
        if sr is None:
            return "%s: %s" % (src.filename, self.message)
        c = src.contents
        start = sr.loc0[1]
        while start > 0 and c[start - 1] != "\n":
            start -= 1
        end = sr.loc0[1]
        while end < len(c) and c[end] not in "\r\n":
            end += 1
        head = "%s:%d:%d: %s" % (src.filename, sr.loc0[0] + 1, sr.loc0[1] - start + 1, self.message)
        if sr.loc0[1] == sr.loc1[1]:
            return head
        carets = " " * (sr.loc0[1] - start) + "^" * (min(end, sr.loc1[1]) - sr.loc0[1])
        return "%s\n\n%s\n%s" % (head, c[start:end], carets)
