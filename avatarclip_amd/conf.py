"""Minimal HOCON-subset parser for the reference's .conf files (pyhocon is what the reference uses,
AvatarGen/AppearanceGen/main.py:39-42; it is not installable here).  Supports exactly what the shipped confs use
(SURVEY.md Appendix B.16): nested `name { ... }` blocks, `key = value` with optional trailing comma, `#` comments,
lists `[a, b]`, unquoted strings with spaces and `{}` (prompts), paths, ints, floats (`5e-4`), True/False.
Accessors mirror pyhocon's ConfigTree: conf['a.b'], get_int/get_float/get_bool/get_string(key, default=...),
`**conf['model.sdf_network']`.
"""
import re
from collections import OrderedDict

_MISSING = object()


class ConfigMissingException(KeyError):
    pass


class ConfigTree(OrderedDict):
    def _lookup(self, key):
        node = self
        for part in key.split("."):
            if not isinstance(node, ConfigTree) or not OrderedDict.__contains__(node, part):
                raise ConfigMissingException("No configuration setting found for key %s" % key)
            node = OrderedDict.__getitem__(node, part)
        return node

    def __getitem__(self, key):
        return self._lookup(key)

    def __contains__(self, key):
        try:
            self._lookup(key)
            return True
        except ConfigMissingException:
            return False

    def get(self, key, default=_MISSING):
        try:
            return self._lookup(key)
        except ConfigMissingException:
            if default is _MISSING:
                raise
            return default

    def get_int(self, key, default=_MISSING):
        v = self.get(key, default)
        return v if v is default else int(v)

    def get_float(self, key, default=_MISSING):
        v = self.get(key, default)
        return v if v is default else float(v)

    def get_string(self, key, default=_MISSING):
        v = self.get(key, default)
        return v if v is default else str(v)

    def get_bool(self, key, default=_MISSING):
        v = self.get(key, default)
        if v is default:
            return v
        if isinstance(v, bool):
            return v
        if isinstance(v, str):
            if v.lower() in ("true", "yes", "on"):
                return True
            if v.lower() in ("false", "no", "off"):
                return False
        raise ValueError("%s is not a boolean: %r" % (key, v))

    def get_list(self, key, default=_MISSING):
        v = self.get(key, default)
        return v if v is default else list(v)

    def put(self, key, value):
        node = self
        parts = key.split(".")
        for part in parts[:-1]:
            if not OrderedDict.__contains__(node, part):
                OrderedDict.__setitem__(node, part, ConfigTree())
            node = OrderedDict.__getitem__(node, part)
        OrderedDict.__setitem__(node, parts[-1], value)


def _scalar(tok):
    tok = tok.strip()
    if len(tok) >= 2 and tok[0] == tok[-1] and tok[0] in "\"'":
        return tok[1:-1]
    if tok in ("True", "true"):
        return True
    if tok in ("False", "false"):
        return False
    if tok in ("null", "None"):
        return None
    if re.fullmatch(r"[+-]?\d+", tok):
        return int(tok)
    try:
        if re.fullmatch(r"[+-]?(\d+\.?\d*|\.\d+)([eE][+-]?\d+)?", tok):
            return float(tok)
    except ValueError:
        pass
    return tok


def _strip_comment(line):
    out, quote = [], None
    for i, ch in enumerate(line):
        if quote:
            if ch == quote:
                quote = None
        elif ch in "\"'":
            quote = ch
        elif ch == "#" or (ch == "/" and line[i:i + 2] == "//" and (i == 0 or line[i - 1].isspace())):
            break
        out.append(ch)
    return "".join(out).rstrip()


def _parse_list(text):
    inner = text.strip()[1:-1]
    items = [t for t in re.split(r"[,\n]", inner) if t.strip()]
    return [_scalar(t) for t in items]


class ConfigFactory:
    @staticmethod
    def parse_string(text):
        root = ConfigTree()
        stack = [root]
        lines = []
        for raw in text.splitlines():
            l = _strip_comment(raw).strip()
            m = re.match(r"^([\w.\-]+)\s*[=:]?\s*\{\s*(\S.*)$", l)      # `name { key = value` on one line
            if m and not l.endswith("{"):
                lines.append(m.group(1) + " {")
                l = m.group(2).strip()
            while l.endswith("}") and l != "}" and l.count("}") > l.count("{"):   # `key = value }`
                lines.append(l[:-1].strip())
                l = "}"
            lines.append(l)
        i = 0
        while i < len(lines):
            line = lines[i].strip()
            i += 1
            if not line:
                continue
            if line in ("}", "},"):
                stack.pop()
                continue
            m = re.fullmatch(r"([\w.\-]+)\s*[=:]?\s*\{", line)
            if m:
                node = ConfigTree()
                stack[-1].put(m.group(1), node)
                stack.append(node)
                continue
            m = re.match(r"([\w.\-]+)\s*[=:]\s*(.*)$", line)
            if not m:
                raise ValueError("cannot parse conf line: %r" % line)
            key, val = m.group(1), m.group(2).strip()
            if val.startswith("["):
                while val.count("[") > val.count("]"):
                    val += "\n" + lines[i].strip()
                    i += 1
                val = val.rstrip(",").strip()
                stack[-1].put(key, _parse_list(val))
                continue
            if val.endswith(","):
                val = val[:-1]
            stack[-1].put(key, _scalar(val))
        if len(stack) != 1:
            raise ValueError("unbalanced braces in conf")
        return root

    @staticmethod
    def parse_file(path):
        with open(path) as f:
            return ConfigFactory.parse_string(f.read())
