"""Test shim for the `commentjson` package imported by the reference's scripts/run.py (not installed here; no network):
json with // and /* */ comments stripped.  Only what run.py / common.py call: load, loads, dump, dumps."""
import json
import re

_COMMENT = re.compile(r'("(?:\\.|[^"\\])*")|//[^\n]*|/\*.*?\*/', re.S)


def _strip(text):
    return _COMMENT.sub(lambda m: m.group(1) or "", text)


def loads(text, **kw):
    return json.loads(_strip(text), **kw)


def load(fp, **kw):
    return loads(fp.read(), **kw)


dump = json.dump
dumps = json.dumps
JSONDecodeError = json.JSONDecodeError
