import re

_NUM = re.compile(r"^[+-]?(\d+\.?\d*|\.\d+)([edfslEDFSL][+-]?\d+)?$")
_RAT = re.compile(r"^([+-]?\d+)/(\d+)$")


def parse_number(tok):
    m = _RAT.match(tok)
    if m:
        return int(m.group(1)) / int(m.group(2))
    if _NUM.match(tok):
        return float(re.sub(r"[dDfFsSlL]", "e", tok))
    return None
