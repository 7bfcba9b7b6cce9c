"""Protobuf *text format* reader for Caffe ``.prototxt`` network definitions.

The reference reads these files with libprotobuf
(``ReadNetParamsFromTextFileOrDie``, caffe_3d/src/caffe/net.cpp:31-36 ->
``google::protobuf::TextFormat::Parse``) against the schema in
caffe_3d/src/caffe/proto/caffe.proto.  There is no ``protoc`` in this image and
the hot path only needs a handful of fields, so this is a small schema-less
recursive-descent parser for the text format itself:

* ``name: value`` scalars (numbers, quoted strings, bare enum identifiers,
  ``true``/``false``),
* ``name { ... }`` and ``name: { ... }`` sub-messages,
* ``name: [a, b, c]`` short-hand for repeated scalars (the ECO files use it
  for ``pad``/``kernel_size``/``stride``/``order``),
* ``#`` comments.

Every field is kept as *repeated* (a list, in file order); typed accessors
apply the proto2 "last one wins"/default rules the callers need.  The
schema-aware cross-check against the reference's own generated
``caffe_pb2.py`` lives in tests/test_prototxt.py (test-only, container-only).
"""
from __future__ import annotations

import re
from typing import Any, Iterator, List, Optional, Tuple

__all__ = ["Message", "parse", "parse_file", "ParseError"]


class ParseError(ValueError):
    pass


class Enum(str):
    """A bare identifier value (enum constant such as ``MAX`` or ``TEST``)."""

    __slots__ = ()


class Message:
    """Ordered multi-map of field name -> list of values."""

    __slots__ = ("_fields", "_order")

    def __init__(self) -> None:
        self._fields: dict = {}
        self._order: List[Tuple[str, Any]] = []

    # -- construction -------------------------------------------------------
    def add(self, name: str, value: Any) -> None:
        self._fields.setdefault(name, []).append(value)
        self._order.append((name, value))

    # -- access -------------------------------------------------------------
    def has(self, name: str) -> bool:
        return name in self._fields

    def getall(self, name: str) -> list:
        return list(self._fields.get(name, ()))

    def get(self, name: str, default: Any = None) -> Any:
        """proto2 optional-field semantics: the last occurrence wins."""
        vals = self._fields.get(name)
        return vals[-1] if vals else default

    def msg(self, name: str) -> "Message":
        """Sub-message accessor; an absent sub-message reads as empty."""
        v = self.get(name)
        if v is None:
            return Message()
        if not isinstance(v, Message):
            raise ParseError(f"field {name!r} is a scalar, not a message")
        return v

    def items(self) -> Iterator[Tuple[str, Any]]:
        return iter(self._order)

    def names(self) -> List[str]:
        return list(self._fields)

    def __contains__(self, name: str) -> bool:
        return name in self._fields

    def __repr__(self) -> str:
        return "Message(%s)" % ", ".join(f"{k}={v!r}" for k, v in self._order)

    # -- serialisation (round-trip, used by tests and Net.save-style tools) --
    def to_text(self, indent: int = 0) -> str:
        pad = "  " * indent
        out = []
        for k, v in self._order:
            if isinstance(v, Message):
                out.append(f"{pad}{k} {{\n{v.to_text(indent + 1)}{pad}}}\n")
            elif isinstance(v, Enum):
                out.append(f"{pad}{k}: {v}\n")
            elif isinstance(v, str):
                esc = v.replace("\\", "\\\\").replace('"', '\\"')
                out.append(f'{pad}{k}: "{esc}"\n')
            elif isinstance(v, bool):
                out.append(f"{pad}{k}: {'true' if v else 'false'}\n")
            else:
                out.append(f"{pad}{k}: {v!r}\n")
        return "".join(out)


_TOKEN = re.compile(
    r"""
    (?P<ws>\s+|\#[^\n]*)                              |
    (?P<str>"(?:\\.|[^"\\])*"|'(?:\\.|[^'\\])*')       |
    (?P<num>[-+]?(?:inf|nan|(?:\d+\.?\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?)f?)\b) |
    (?P<ident>[A-Za-z_][A-Za-z0-9_.]*)                 |
    (?P<punct>[{}\[\]:,;<>])
    """,
    re.VERBOSE,
)

_ESCAPES = {"n": "\n", "t": "\t", "r": "\r", "\\": "\\", '"': '"', "'": "'", "0": "\0"}


def _unescape(s: str) -> str:
    body = s[1:-1]
    if "\\" not in body:
        return body
    out = []
    i = 0
    while i < len(body):
        c = body[i]
        if c == "\\" and i + 1 < len(body):
            out.append(_ESCAPES.get(body[i + 1], body[i + 1]))
            i += 2
        else:
            out.append(c)
            i += 1
    return "".join(out)


def _tokenize(text: str) -> List[Tuple[str, Any, int]]:
    toks = []
    pos = 0
    n = len(text)
    while pos < n:
        m = _TOKEN.match(text, pos)
        if m is None:
            line = text.count("\n", 0, pos) + 1
            raise ParseError(f"line {line}: unexpected character {text[pos]!r}")
        pos = m.end()
        kind = m.lastgroup
        if kind == "ws":
            continue
        raw = m.group(kind)
        if kind == "str":
            toks.append(("str", _unescape(raw), m.start()))
        elif kind == "num":
            r = raw.rstrip("f") if raw[-1:] == "f" and raw not in ("inf", "-inf", "+inf") else raw
            try:
                val: Any = int(r)
            except ValueError:
                val = float(r)
            toks.append(("num", val, m.start()))
        elif kind == "ident":
            toks.append(("ident", raw, m.start()))
        else:
            toks.append((raw, raw, m.start()))
    return toks


class _Parser:
    def __init__(self, text: str) -> None:
        self.text = text
        self.toks = _tokenize(text)
        self.i = 0

    def _line(self, tokpos: int) -> int:
        return self.text.count("\n", 0, tokpos) + 1

    def _peek(self) -> Optional[Tuple[str, Any, int]]:
        return self.toks[self.i] if self.i < len(self.toks) else None

    def _next(self) -> Tuple[str, Any, int]:
        if self.i >= len(self.toks):
            raise ParseError("unexpected end of input")
        t = self.toks[self.i]
        self.i += 1
        return t

    def _err(self, tok, what: str) -> ParseError:
        return ParseError(f"line {self._line(tok[2])}: {what}, got {tok[1]!r}")

    def parse_message(self, closer: Optional[str]) -> Message:
        msg = Message()
        while True:
            t = self._peek()
            if t is None:
                if closer is not None:
                    raise ParseError(f"unexpected end of input, expected {closer!r}")
                return msg
            if t[0] == closer:
                self.i += 1
                return msg
            if t[0] in (",", ";"):  # optional field separators
                self.i += 1
                continue
            if t[0] != "ident":
                raise self._err(t, "expected a field name")
            name = self._next()[1]
            t = self._peek()
            if t is None:
                raise ParseError(f"field {name!r}: unexpected end of input")
            if t[0] == ":":
                self.i += 1
                t = self._peek()
                if t is None:
                    raise ParseError(f"field {name!r}: unexpected end of input")
            if t[0] == "{":
                self.i += 1
                msg.add(name, self.parse_message("}"))
            elif t[0] == "<":
                self.i += 1
                msg.add(name, self.parse_message(">"))
            elif t[0] == "[":
                self.i += 1
                for v in self.parse_list():
                    msg.add(name, v)
            else:
                msg.add(name, self.parse_scalar())

    def parse_list(self) -> list:
        vals = []
        while True:
            t = self._peek()
            if t is None:
                raise ParseError("unexpected end of input inside [...]")
            if t[0] == "]":
                self.i += 1
                return vals
            if t[0] == ",":
                self.i += 1
                continue
            if t[0] == "{":
                self.i += 1
                vals.append(self.parse_message("}"))
            else:
                vals.append(self.parse_scalar())

    def parse_scalar(self) -> Any:
        t = self._next()
        if t[0] == "num":
            return t[1]
        if t[0] == "str":
            # adjacent string literals concatenate, as in protobuf text format
            s = t[1]
            while (p := self._peek()) is not None and p[0] == "str":
                s += self._next()[1]
            return s
        if t[0] == "ident":
            if t[1] == "true":
                return True
            if t[1] == "false":
                return False
            if t[1] in ("inf", "nan"):
                return float(t[1])
            return Enum(t[1])
        raise self._err(t, "expected a value")


def parse(text: str) -> Message:
    """Parse protobuf text format into a :class:`Message` tree."""
    return _Parser(text).parse_message(None)


def parse_file(path: str) -> Message:
    with open(path, "r") as f:
        return parse(f.read())
