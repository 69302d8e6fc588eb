"""svoxt_kernels.hip with its four kernel headers inlined: one text the experiment builds
(trace_build.py, trace_march.py, march_variants.py) can patch by substitution."""
import os
import re

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "svox_t_amd", "csrc")
PARTS = ["svoxt_lists.h", "svoxt_fwd_kernels.h", "svoxt_bwd_kernels.h", "svoxt_misc_kernels.h"]


def flat_source() -> str:
    src = open(os.path.join(CSRC, "svoxt_kernels.hip")).read()
    for name in PARTS:
        body = open(os.path.join(CSRC, name)).read().replace("#pragma once\n", "")
        body = re.sub(r'#include "svoxt_(lists|fwd_kernels|bwd_kernels|misc_kernels)\.h"\n', "", body)
        inc = f'#include "{name}"\n'
        assert src.count(inc) == 1, name
        src = src.replace(inc, body)
    for name in ("svoxt_tile_reduce.inc",):          # textual includes inside kernels
        src = src.replace(f'#include "{name}"\n', open(os.path.join(CSRC, name)).read())
    return src
