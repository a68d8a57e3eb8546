#!/usr/bin/env python3
"""Build experimental copies of libbrov2.so (rollout.hip compiled with extra -D flags) under build_variants/<name>/.

    python tools/build_variants.py name1=-DA=1,-DB=1 name2=-DC=1 ...

Run them on the GPU box with tools/attic/run_variants.sh (BROV2_LIBRARY selects the copy)."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bluerov2_dynamics_amd import _build  # noqa: E402

_build.build_library()
specs = []
for a in sys.argv[1:]:
    name, _, flags = a.partition("=")
    src = "rollout.hip"
    if ":" in name:                      # file:name=flags
        src, name = name.split(":", 1)
    specs.append((name, src, [f for f in flags.split(",") if f]))


def one(spec):
    name, src, flags = spec
    lib = _build.variant(name, {src: flags})
    return name, lib


with ThreadPoolExecutor(max_workers=3) as ex:
    for name, lib in ex.map(one, specs):
        print(name, lib, flush=True)
