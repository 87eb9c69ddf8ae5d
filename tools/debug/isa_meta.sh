#!/bin/bash
# Device assembly of one kernel family + registers / scratch / code size of the stepping kernels
# (no GPU needed).  Usage: tools/debug/isa_meta.sh [w16|w8] [out.s] [extra -D flags...]
FAM=${1:-w16}; OUT=${2:-/tmp/isa/$FAM.s}; shift 2 2>/dev/null
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p "$(dirname "$OUT")"
WPG=${FAM#w}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
  -Wno-unused-function -Wno-pass-failed -DMHX_WPG=$WPG -DMHX_FAMILY=$FAM "$@" --cuda-device-only -S \
  -o "$OUT" "$ROOT/lisp-mcmc_amd/csrc/mhx_kernels.hip" 2>&1 | grep -E "error|warning: .*spill" 
python3 - "$OUT" <<'PY'
import re, sys
t = open(sys.argv[1]).read()
for m in re.finditer(r'^(_ZN3mhx3w\d+10k_adaptiveI[^:\n]*):', t, re.M):
    name = m.group(1)
    seg = t[m.end():]
    e = seg.find('.Lfunc_end')
    meta = seg[e:e + 6000]
    g = lambda k: re.search(r'; %s: (\d+)' % k, meta).group(1)
    short = re.sub(r'^_ZN3mhx3w\d+10k_adaptiveINS0_', '', name)[:62]
    print('%-64s lines %6d vgpr %3s scratch %4s code %6s' % (
        short, seg[:e].count('\n'), g('NumVgprs'), g('ScratchSize'),
        re.search(r'codeLenInByte = (\d+)', meta).group(1)))
PY
