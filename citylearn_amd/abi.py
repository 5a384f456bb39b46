"""Slot numbers and flag bits of the C-ABI, parsed from ``include/citylearn_amd.h``.

The header is the single source of truth for the table layouts shared by the HIP kernels
(``citylearn_amd/csrc``), the host loader (``schema.py``) and the test oracle; parsing it here keeps the
Python side from drifting.
"""
from __future__ import annotations

import re
from pathlib import Path

HEADER = Path(__file__).resolve().parent.parent / 'include' / 'citylearn_amd.h'


def _strip_comments(text: str) -> str:
    text = re.sub(r'/\*.*?\*/', ' ', text, flags=re.S)
    return re.sub(r'//[^\n]*', ' ', text)


def _parse(text: str) -> dict:
    text = _strip_comments(text)
    out: dict = {}
    for m in re.finditer(r'#define\s+(CL\w+)\s+(.+)', text):
        name, expr = m.group(1), m.group(2).strip()
        expr = re.sub(r'(0x[0-9a-fA-F]+|\d+)[uU]\b', r'\1', expr)
        try:
            out[name] = int(eval(expr, {'__builtins__': {}}, out))  # noqa: S307 - header constants only
        except Exception:
            pass
    for m in re.finditer(r'enum\s+(\w+)\s*\{(.*?)\}', text, flags=re.S):
        nxt = 0
        for item in m.group(2).split(','):
            item = item.strip()
            if not item:
                continue
            if '=' in item:
                name, val = [s.strip() for s in item.split('=')]
                nxt = int(eval(val, {'__builtins__': {}}, out))  # noqa: S307
            else:
                name = item
            out[name] = nxt
            nxt += 1
    return out


_C = _parse(HEADER.read_text())
globals().update(_C)
CONSTANTS = dict(_C)

# every function the header declares (used by the "library exports every symbol" test)
EXPORTED_SYMBOLS = sorted(set(re.findall(r'\b(cl_\w+)\s*\(', _strip_comments(HEADER.read_text()))))

assert _C['CLP_USED'] <= _C['CLP_F_FIRST'] and _C['CLP_F_LAST'] < _C['CLP_D_FIRST'] and _C['CLP_D_LAST'] < _C['CL_NP'], 'cl_param overflows CL_NP'
assert _C['CLPD_USED'] * 2 <= _C['CLP_D_LAST'] - _C['CLP_D_FIRST'] + 1 and _C['CLP_D_FIRST'] % 2 == 0 and _C['CL_NP'] % 2 == 0
assert _C['CLP_D_LAST'] < _C['CLP_C_FIRST'] and _C['CLP_C_LAST'] < _C['CL_NP'] and _C['CLP_C_FIRST'] % 2 == 0 and _C['CLPC_USED'] * 2 <= _C['CLP_C_LAST'] - _C['CLP_C_FIRST'] + 1
